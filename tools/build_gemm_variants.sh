#!/bin/bash
# Build variants of csrc/gemm_rows_f16x3.hip into ONE library for tools/gemm_variants.py (A/B inside one process).
#   usage: tools/build_gemm_variants.sh "<flags of v0>" "<flags of v1>" ...     e.g.  "" "-DGH_NOSTORE=1"
set -e
cd "$(dirname "$0")/.."
V=tools/_build/gv; mkdir -p $V
objs=(); i=0
for flags in "$@"; do
  sed -e "s|#include \"common.h\"|#include \"../../../sloika_amd/csrc/common.h\"|" \
      -e "s/split_f16x2_kernel/spk_v$i/g" -e "s/slk_split_f16x2_f32/slk_sp_v$i/g" -e "s/gemm_rows_f16x3_kernel/grk_v$i/g" \
      -e "s/slk_linear_rowstats_f16x3/slk_lr_v$i/g" -e "s/slk_gemm_bias_act_f16x3/slk_gb_v$i/g" sloika_amd/csrc/gemm_rows_f16x3.hip > $V/g$i.hip
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=off $flags -c $V/g$i.hip -o $V/g$i.o
  objs+=($V/g$i.o); i=$((i+1))
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_build/libgemm_variants.so "${objs[@]}"
echo built $i variants
