#!/bin/bash
# bash tools/bt_prof.sh REF.so TAG: kernel durations of the backtrace of REF.so and of the built library, one process
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_bt_$2 -o bt -- python3 tools/bt_ab.py $1 > gpurun_out/prof_bt_$2.log 2>&1
python3 tools/bt_prof.py gpurun_out/prof_bt_$2/bt_results.db
