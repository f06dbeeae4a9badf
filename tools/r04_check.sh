#!/bin/bash
# One GPU call: the quick bench line, the GPU test suite, the race screen of the Gru plans.   bash tools/r04_check.sh <tag> [pytest args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
python3 bench.py --steps 20 --warmup 3 --quick --cpu-chunks 0 > gpurun_out/${tag}_quick_bench.json 2> gpurun_out/${tag}_quick_bench.err
python3 -c "import json; d=json.load(open('gpurun_out/${tag}_quick_bench.json')); print('BENCH', d['value'], d['ms_per_step'], d['stages_ms_per_step'], d['roofline']['ms_per_launch'])"
timeout 3000 python3 -m pytest tests -m gpu -q "$@" > gpurun_out/${tag}_pytest.txt 2>&1; echo "pytest rc=$?"; grep -c FAILED gpurun_out/${tag}_pytest.txt; grep FAILED gpurun_out/${tag}_pytest.txt | sed 's/\[.*//' | sort | uniq -c | head -30; tail -2 gpurun_out/${tag}_pytest.txt
timeout 900 python3 tools/soak_new_kernels.py 30 > gpurun_out/${tag}_soak.txt 2>&1; tail -2 gpurun_out/${tag}_soak.txt
