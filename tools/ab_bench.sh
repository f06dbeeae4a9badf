#!/bin/bash
# A/B of two library builds (tools/_build/lib_old.so, lib_new.so: copies of sloika_amd/_build/libsloika_amd.so) on ONE device:
# alternate bench runs, print ms_per_step and the stage times of each (devices differ by ~10 %, runs on one device by ~2 %)
L=sloika_amd/_build/libsloika_amd.so
cp $L /tmp/lib_keep.so
for r in 1 2 3; do
 for v in old new; do
  cp tools/_build/lib_$v.so $L
  python3 bench.py --steps 20 --warmup 3 --cpu-chunks 0 --exact-steps 0 --overlap-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stages_ms_per_step'].items()})"
 done
done
cp /tmp/lib_keep.so $L
