#!/bin/bash
# Collect HBM traffic of every kernel of the bench workload with rocprofv3 PMC counters, as prescribed by
# MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), no tracing domains besides
# --kernel-trace.  Run on the GPU box from the repo root:   bash tools/collect_pmc.sh [extra bench args]
# Then:  python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/pmc_traffic.json
set -e
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --cpu-chunks 0 --quick --no-stage-timing "$@" > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --cpu-chunks 0 --quick --no-stage-timing "$@" > gpurun_out/pmc_write.log 2>&1
