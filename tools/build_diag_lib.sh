#!/bin/bash
# The diagnostic build of the library for the scripts under tools/ (per-section s_memtime stamps, ablation launches, workgroup
# clocks of the Gru kernels: -DSLK_DIAG), next to the production one:
#     tools/build_diag_lib.sh && SLOIKA_AMD_LIB=$PWD/tools/_build/libsloika_amd_diag.so python tools/bar16_check.py
set -e
cd "$(dirname "$0")/.."
V=tools/_build/diag; mkdir -p $V
objs=()
for src in sloika_amd/csrc/*.hip; do
  o=$V/$(basename ${src%.hip}).o
  /opt/rocm/bin/hipcc $(python3 -c "import sys; sys.path.insert(0, '.'); from sloika_amd import build; print(' '.join(f for f in build.flags_for('$src') if f != '-Wall'))") -DSLK_DIAG -c $src -o $o &
  objs+=($o)
  if (( ${#objs[@]} % 4 == 0 )); then wait; fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_build/libsloika_amd_diag.so "${objs[@]}"
echo tools/_build/libsloika_amd_diag.so
