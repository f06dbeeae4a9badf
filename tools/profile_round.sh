#!/bin/bash
# One round's evidence in one go (run on the GPU box from the repo root):  bash tools/profile_round.sh r02a
#   gpurun_out/<tag>_bench.json          the bench line (default run)
#   gpurun_out/prof_<tag>/               rocprofv3 --kernel-trace --stats of the same command (csv)
#   gpurun_out/pmc_fetch, pmc_write      PMC passes (tools/collect_pmc.sh) -> gpurun_out/<tag>_pmc_traffic.json
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
python3 bench.py --steps 20 --warmup 3 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rm -rf gpurun_out/prof_${tag}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python3 bench.py --steps 10 --warmup 3 --cpu-chunks 0 --exact-steps 0 --overlap-steps 0 --small-batch-steps 0 > gpurun_out/prof_${tag}.log 2>&1
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
bash tools/collect_pmc.sh
python3 tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/${tag}_pmc_traffic.json
find gpurun_out/prof_${tag} -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_kernel_stats.csv
head -12 gpurun_out/${tag}_kernel_stats.csv
