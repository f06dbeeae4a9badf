#!/bin/bash
# Round-6 evidence (round 5: tools/r05_measure.sh, the same without the `pretrained` workload and the driver-style bench) in one GPU call
# (run on the GPU box from the repo root):  bash tools/r06_measure.sh <tag> [parts...]
#   parts: stamps   per-section shader-clock stamps + ablation launches of gru_bar16 (diagnostic library, tools/build_diag_lib.sh)
#          util     MfmaUtil / VALUBusy / LdsUtil / LDSBankConflict per kernel for the five bench workloads the review names
#          sq       raw SQ counters of the default workload (wave cycles, waits, LDS conflicts, MFMA busy cycles)
#          pmc      HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) of the default workload, the two batch-256 ones, the training step
#          stats    rocprofv3 --kernel-trace --stats of the default workload and of the training step
#          bench    the full default bench line (and the --train line)
# Everything lands under gpurun_out/<tag>_*; copy what is to be judged into profiles/.  Every command runs under `timeout`.
tag=$1; shift
parts=${@:-stamps util sq pmc stats bench}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
has() { [[ " $parts " == *" $1 "* ]]; }
Q="--steps 2 --warmup 1 --cpu-chunks 0 --quick --no-stage-timing"

if has bench; then
  # the driver's own command first (a fresh process on an idle device: its 20 steps behind 5), then the default run
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driver_style.json 2> gpurun_out/${tag}_bench_driver_style.err
  cp bench_detail.json gpurun_out/${tag}_bench_driver_style_detail.json
  timeout 600 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
  cp bench_detail.json gpurun_out/${tag}_bench_detail.json
  timeout 300 python3 bench.py --train --steps 10 --warmup 3 > gpurun_out/${tag}_train_bench.json 2> gpurun_out/${tag}_train_bench.err
  cp bench_detail.json gpurun_out/${tag}_train_bench_detail.json
  timeout 120 python3 tools/warmup_curve.py 60 --no-probe > gpurun_out/${tag}_warmup_curve.txt 2>&1
  timeout 120 python3 tools/warmup_kernel_only.py > gpurun_out/${tag}_warmup_kernel_only.txt 2>&1
fi
if has stamps; then
  SLOIKA_AMD_LIB=$PWD/tools/_build/libsloika_amd_diag.so timeout 900 python3 tools/bar16_check.py 96x96 > gpurun_out/${tag}_bar16_check.txt 2>&1
fi
if has stats; then
  rm -rf gpurun_out/prof_${tag} gpurun_out/prof_${tag}_train
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python3 bench.py --steps 10 --warmup 3 --cpu-chunks 0 --quick > gpurun_out/prof_${tag}.log 2>&1
  find gpurun_out/prof_${tag} -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_kernel_stats.csv
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_train -- python3 bench.py --train --steps 5 --warmup 2 --cpu-chunks 0 > gpurun_out/prof_${tag}_train.log 2>&1
  find gpurun_out/prof_${tag}_train -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_train_kernel_stats.csv
  rm -rf gpurun_out/prof_${tag}_pre
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_pre -- python3 bench.py --model pretrained --steps 10 --warmup 3 --cpu-chunks 0 --quick > gpurun_out/prof_${tag}_pre.log 2>&1
  find gpurun_out/prof_${tag}_pre -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_pretrained_kernel_stats.csv
  rm -rf gpurun_out/prof_${tag} gpurun_out/prof_${tag}_train gpurun_out/prof_${tag}_pre
fi
if has util; then
  for wl in "b1024:" "b1024x4:--streams 4" "b256_rgrgr:--batch 256" "b256_baseline:--model baseline_raw_gru --batch 256" "train:--train" "pretrained:--model pretrained"; do
    name=${wl%%:*}; args=${wl#*:}
    if [ -n "$UTIL_WL" ] && [[ " $UTIL_WL " != *" $name "* ]]; then continue; fi
    for c in MfmaUtil LdsUtil VALUBusy LDSBankConflict; do
      rm -rf gpurun_out/util_$c
      timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/util_$c -- python3 bench.py $Q $args > gpurun_out/util_$c.log 2>&1
    done
    python3 tools/util_summary.py gpurun_out > gpurun_out/${tag}_${name}_unit_utilisation.json
    rm -rf gpurun_out/util_MfmaUtil gpurun_out/util_LdsUtil gpurun_out/util_VALUBusy gpurun_out/util_LDSBankConflict
  done
fi
if has pmc; then
  for wl in "b1024:" "b256_rgrgr:--batch 256" "b256_baseline:--model baseline_raw_gru --batch 256" "train:--train" "pretrained:--model pretrained"; do
    name=${wl%%:*}; args=${wl#*:}
    rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
    timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py $Q $args > gpurun_out/pmc_fetch.log 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py $Q $args > gpurun_out/pmc_write.log 2>&1
    python3 tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/${tag}_${name}_pmc_traffic.json
    rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
  done
fi
if has sq; then
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES" \
             "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY"; do
    rm -rf gpurun_out/sq_$i
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/sq_$i -- python3 bench.py $Q > gpurun_out/sq_$i.log 2>&1
    i=$((i+1))
  done
  python3 tools/sq_summary.py gpurun_out > gpurun_out/${tag}_sq_counters.json
fi
# every counter file carries the hash of the kernels it was taken on (bench.py: "stale" when the tree differs)
python3 tools/stamp_profiles.py gpurun_out/${tag}_*unit_utilisation.json gpurun_out/${tag}_*pmc_traffic.json gpurun_out/${tag}_sq_counters.json > gpurun_out/${tag}_stamp.log 2>&1
true
