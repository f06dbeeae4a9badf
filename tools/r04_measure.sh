#!/bin/bash
# Round-4 evidence in one GPU call (run on the GPU box from the repo root):  bash tools/r04_measure.sh <tag> [parts...]
#   parts: stamps   per-section shader-clock stamps + ablation launches of gru_bar16 (diagnostic library, tools/build_diag_lib.sh)
#          util     MfmaUtil / VALUBusy / LdsUtil / LDSBankConflict per kernel for the five bench workloads the review names
#          sq       raw SQ counters of the default workload (wave cycles, waits, LDS conflicts, MFMA busy cycles)
#          bench    the --quick bench line of the default workload
# Everything lands under gpurun_out/<tag>_*; copy what is to be judged into profiles/.
tag=$1; shift
parts=${@:-stamps util sq bench}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
has() { [[ " $parts " == *" $1 "* ]]; }

if has bench; then
  python3 bench.py --steps 20 --warmup 3 --quick --cpu-chunks 0 > gpurun_out/${tag}_quick_bench.json 2> gpurun_out/${tag}_quick_bench.err
fi
if has stamps; then
  SLOIKA_AMD_LIB=$PWD/tools/_build/libsloika_amd_diag.so timeout 900 python3 tools/bar16_check.py 96x96 > gpurun_out/${tag}_bar16_check.txt 2>&1
fi
if has util; then
  for wl in "b1024:" "b1024x4:--streams 4" "b256_rgrgr:--batch 256" "b256_baseline:--model baseline_raw_gru --batch 256" "train:--train"; do
    name=${wl%%:*}; args=${wl#*:}
    if [ -n "$UTIL_WL" ] && [[ " $UTIL_WL " != *" $name "* ]]; then continue; fi
    for c in MfmaUtil LdsUtil VALUBusy LDSBankConflict; do
      rm -rf gpurun_out/util_$c
      rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/util_$c -- python3 bench.py --steps 2 --warmup 1 --cpu-chunks 0 --quick --no-stage-timing $args > gpurun_out/util_$c.log 2>&1
    done
    python3 tools/util_summary.py gpurun_out > gpurun_out/${tag}_${name}_unit_utilisation.json
  done
fi
if has sq; then
  rocprofv3 -L > gpurun_out/${tag}_counters_list.txt 2>&1
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES" \
             "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY"; do
    rm -rf gpurun_out/sq_$i
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/sq_$i -- python3 bench.py --steps 2 --warmup 1 --cpu-chunks 0 --quick --no-stage-timing > gpurun_out/sq_$i.log 2>&1
    i=$((i+1))
  done
  python3 tools/sq_summary.py gpurun_out > gpurun_out/${tag}_sq_counters.json
fi
ls gpurun_out | head -100 > /dev/null
