#!/bin/bash
# The library as of a git revision, next to the working tree's, for in-process A/B timing (devices of the pool differ by ~10 %):
#     tools/build_ref_lib.sh <rev>     ->  tools/_build/libref_<rev>.so   (sources: git archive <rev>, flags: that revision's build.py)
set -e
cd "$(dirname "$0")/.."
rev=$1
V=tools/_build/ref_$rev; rm -rf $V; mkdir -p $V
git archive $rev sloika_amd/csrc sloika_amd/build.py include | tar -x -C $V
extra=$(grep -q "amdgpu-mfma-vgpr-form" $V/sloika_amd/build.py && echo yes || echo no)
objs=()
for src in $V/sloika_amd/csrc/*.hip; do
  o=${src%.hip}.o
  f=""
  if [ $extra = yes ]; then f=$(cd $V && python3 -c "from sloika_amd import build; print(' '.join(x for x in build.flags_for('$(basename $src)') if x.startswith('-m') or x.startswith('-amdgpu')))" 2>/dev/null || true); fi
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=off -fvisibility=hidden $f -I$V/include -c $src -o $o &
  objs+=($o)
  if (( ${#objs[@]} % 6 == 0 )); then wait; fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_build/libref_$rev.so "${objs[@]}"
echo tools/_build/libref_$rev.so
