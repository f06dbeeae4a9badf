#!/bin/bash
# The evidence of tools/r05_measure.sh <tag> from gpurun_out/ into profiles/ (tracked), and the tables bench.py reads:
#     bash tools/install_profiles.sh r05a
set -e
cd "$(dirname "$0")/.."
tag=$1
for f in bench.json train_bench.json kernel_stats.csv train_kernel_stats.csv b1024_unit_utilisation.json b1024x4_unit_utilisation.json \
         b256_rgrgr_unit_utilisation.json b256_baseline_unit_utilisation.json train_unit_utilisation.json b1024_pmc_traffic.json \
         b256_rgrgr_pmc_traffic.json b256_baseline_pmc_traffic.json train_pmc_traffic.json sq_counters.json \
         pretrained_unit_utilisation.json pretrained_pmc_traffic.json pretrained_kernel_stats.csv bench_detail.json bench_driver_style.json \
         bench_driver_style_detail.json train_bench_detail.json warmup_curve.txt warmup_kernel_only.txt; do
  [ -s gpurun_out/${tag}_$f ] && cp gpurun_out/${tag}_$f profiles/${tag}_$f
done
python3 tools/util_merge.py $tag profiles > profiles/unit_utilisation.json
[ -s profiles/${tag}_b1024_pmc_traffic.json ] && python3 tools/pmc_summary.py --stages profiles/${tag}_b1024_pmc_traffic.json > profiles/pmc_traffic.json
[ -s profiles/${tag}_b256_rgrgr_pmc_traffic.json ] && python3 tools/pmc_summary.py --stages profiles/${tag}_b256_rgrgr_pmc_traffic.json raw_0.98_rgrgr 256 4000 > profiles/pmc_traffic_raw_0.98_rgrgr_b256.json
[ -s profiles/${tag}_b256_baseline_pmc_traffic.json ] && python3 tools/pmc_summary.py --stages profiles/${tag}_b256_baseline_pmc_traffic.json baseline_raw_gru 256 4000 > profiles/pmc_traffic_baseline_raw_gru_b256.json
[ -s profiles/${tag}_pretrained_pmc_traffic.json ] && python3 tools/pmc_summary.py --stages profiles/${tag}_pretrained_pmc_traffic.json pretrained 1024 4000 > profiles/pmc_traffic_pretrained_b1024.json
[ -s profiles/${tag}_train_pmc_traffic.json ] && python3 tools/pmc_summary.py --train-stages profiles/${tag}_train_pmc_traffic.json > profiles/pmc_traffic_train.json
echo "tree now: $(python3 tools/stamp_profiles.py --print); files: $(grep -ho '"csrc": "[^"]*"' profiles/pmc_traffic.json profiles/unit_utilisation.json | sort | uniq -c | tr '\n' ' ')"
