#!/bin/bash
# Build variants of csrc/softmax_viterbi.hip (forward kernel only, no backtrace) into ONE library for tools/sv_variants.py
# (A/B timing inside one process).   usage: tools/build_sv_variants.sh "<flags of v0>" "<flags of v1>" ...   e.g. "" "-DSV_ABL=1"
set -e
cd "$(dirname "$0")/.."
V=tools/_build/sv; mkdir -p $V
objs=(); i=0
for flags in "$@"; do
  src=sloika_amd/csrc/softmax_viterbi.hip
  if [[ "$flags" == @* ]]; then src="${flags%% *}"; src="${src#@}"; if [[ "$flags" == *" "* ]]; then flags="${flags#* }"; else flags=""; fi; fi   # "@other.hip -Dflags": another source file
  sed -e "s|#include \"f16split.h\"|#include \"../../../sloika_amd/csrc/f16split.h\"|" \
      -e "s|#include \"decode_internal.h\"|#include \"../../../sloika_amd/csrc/decode_internal.h\"|" \
      -e "s/softmax_viterbi_kernel/svk_v$i/g" -e "s/sv_pack_kernel/svp_v$i/g" \
      -e "s/slk_softmax_viterbi_pack_bytes/slk_svpb_v$i/g" -e "s/slk_softmax_viterbi_pack_f32/slk_svp_v$i/g" \
      -e "s/slk_softmax_viterbi_f32/slk_sv_v$i/g" $src > $V/s$i.hip
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=off -DSV_NO_BACKTRACE -DSV_ONLY_KS=${SV_KS:-6} $flags -c $V/s$i.hip -o $V/s$i.o &
  objs+=($V/s$i.o); i=$((i+1))
  if (( i % 4 == 0 )); then wait; fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_build/libsv_variants.so "${objs[@]}" sloika_amd/_build/decode.o sloika_amd/_build/capi.o
echo built $i variants
