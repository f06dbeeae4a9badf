#!/bin/bash
# Build gru_fused16 variants into ONE shared library for tools/gf16_variants.py (A/B timing inside one process; timings of
# the same binary differ by ~10 % between boxes of this pool).   usage: [SRC=gru_bar16] tools/build_gf16_variants.sh name=<git rev|work> ...
# SRC names the kernel file under sloika_amd/csrc (default gru_fused16); a spec may override it:  name=<rev|work>:<src>
set -e
cd "$(dirname "$0")/.."
V=tools/_build/variants; mkdir -p $V
objs=()
for spec in "$@"; do
  name=${spec%%=*}; rev=${spec#*=}; src=${SRC:-gru_fused16}
  case "$rev" in *:*) src=${rev#*:}; rev=${rev%%:*};; esac
  if [ "$rev" = work ]; then cat sloika_amd/csrc/$src.hip > $V/g_$name.hip; else git show $rev:sloika_amd/csrc/$src.hip > $V/g_$name.hip; fi
  sed -e "s|#include \"lds_flags.h\"|#include \"../../../sloika_amd/csrc/lds_flags.h\"|" \
      -e "s|#include \"f16split.h\"|#include \"../../../sloika_amd/csrc/f16split.h\"|" \
      -e "s/gru_fused16_kernel/gf16k_$name/g" -e "s/slk_gru_fused16_f32/slk_gf16_$name/g" \
      -e "s/gru_bar16_kernel/gf16k_$name/g" -e "s/slk_gru_bar16_f32/slk_gf16_$name/g" \
      -e "s/slk_dbg_/slk_dbg_${name}_/g" -e "s/slk_debug_read_/slk_debug_read_${name}_/g" $V/g_$name.hip > $V/v_$name.hip
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=off -c $V/v_$name.hip -o $V/v_$name.o
  objs+=($V/v_$name.o)
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_build/libgf16_variants.so "${objs[@]}"
echo built: "$@"
