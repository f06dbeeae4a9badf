#!/bin/bash
# Variants of csrc/gru_bwd16.hip in ONE shared library for tools/bwd16_variants.py (in-process A/B).
# usage: tools/build_bwd16_variants.sh "" "-DGW_ABL=4" ...
set -e
cd "$(dirname "$0")/.."
V=tools/_build/variants; mkdir -p $V
objs=(); i=0
for flags in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=off -Iinclude $flags \
      -Dslk_gru_backward16_f32=slk_gw_v$i -Dslk_gru_backward16_dx_f32=slk_gwdx_v$i -Dgru_bwd16_kernel=gru_gw_k$i -Dgru_bwd16_entry=gru_gw_e$i -c sloika_amd/csrc/gru_bwd16.hip -o $V/gw_$i.o &
  objs+=($V/gw_$i.o); i=$((i+1))
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_build/libbwd16_variants.so "${objs[@]}"
echo built $i variants
