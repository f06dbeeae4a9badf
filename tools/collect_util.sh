#!/bin/bash
# Unit utilisation of every kernel of the bench workload (rocprofv3 derived counters, one pass each; --kernel-trace only):
#   MfmaUtil  LdsUtil  VALUBusy  LDSBankConflict      ->  gpurun_out/util_<name>/ ;  summary: python tools/util_summary.py gpurun_out
set -e
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
for c in MfmaUtil LdsUtil VALUBusy LDSBankConflict; do
  rm -rf gpurun_out/util_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/util_$c -- python3 bench.py --steps 2 --warmup 1 --cpu-chunks 0 --quick --no-stage-timing "$@" > gpurun_out/util_$c.log 2>&1
done
