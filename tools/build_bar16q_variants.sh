#!/bin/bash
# Variants of csrc/gru_bar16q.hip in ONE shared library for tools/bar16q_variants.py (in-process A/B; timings of one binary differ
# by ~10 % between boxes).   usage: tools/build_bar16q_variants.sh "<flags of v0>" "<flags of v1>" ...   e.g. "" "-DBAR16D_ABL=1"
set -e
cd "$(dirname "$0")/.."
V=tools/_build/variants; mkdir -p $V
objs=(); i=0
for flags in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form $flags \
      -Dslk_gru_bar16q_launch=slk_q_v$i -Dgru_bar16q_kernel=gru_q_k$i -c sloika_amd/csrc/gru_bar16q.hip -o $V/q_$i.o &
  objs+=($V/q_$i.o); i=$((i+1))
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_build/libbar16q_variants.so "${objs[@]}"
echo built $i variants
