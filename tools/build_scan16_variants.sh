#!/bin/bash
# Variants of csrc/gru_scan16.hip in ONE shared library for tools/scan16_variants.py.   usage: tools/build_scan16_variants.sh "" "-DSCAN16_ABL=1" ...
set -e
cd "$(dirname "$0")/.."
V=tools/_build/variants; mkdir -p $V
objs=(); i=0
for flags in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=off $flags \
      -Dslk_gru_scan16_f32=slk_s16_v$i -Dgru_scan16_kernel=gru_s16_k$i -c sloika_amd/csrc/gru_scan16.hip -o $V/s16_$i.o &
  objs+=($V/s16_$i.o); i=$((i+1))
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_build/libscan16_variants.so "${objs[@]}"
echo built $i variants
