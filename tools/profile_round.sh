#!/bin/bash
# One round's evidence in one go (run on the GPU box from the repo root):  bash tools/profile_round.sh r02a [bench args]
# Extra bench arguments (e.g. --model baseline_raw_gru --batch 256) select another workload for all three parts; the bench
# line of such a run is the --quick one (main region + stage pass).
#   gpurun_out/<tag>_bench.json          the bench line (default run)
#   gpurun_out/prof_<tag>/               rocprofv3 --kernel-trace --stats of the same command (csv)
#   gpurun_out/pmc_fetch, pmc_write      PMC passes (tools/collect_pmc.sh) -> gpurun_out/<tag>_pmc_traffic.json
set -e
tag=$1
shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
if [ $# -gt 0 ]; then extra="--quick --cpu-chunks 0"; else extra=""; fi
python3 bench.py --steps 20 --warmup 3 $extra "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rm -rf gpurun_out/prof_${tag}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python3 bench.py --steps 10 --warmup 3 --cpu-chunks 0 --quick "$@" > gpurun_out/prof_${tag}.log 2>&1
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
bash tools/collect_pmc.sh "$@"
python3 tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/${tag}_pmc_traffic.json
find gpurun_out/prof_${tag} -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_kernel_stats.csv
head -12 gpurun_out/${tag}_kernel_stats.csv
