#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_variants.py -m gpu -x -q -k "round6" > $O/gpu_sub.txt 2>&1; grep -aE "passed|failed" $O/gpu_sub.txt | tail -1; grep -a "Error\|assert" $O/gpu_sub.txt | head -8
for v in 2 -1 2 -1; do python3 bench.py --no-cpu-baseline --no-dyncore-compare --opt convec_nsingle=$v 2>$O/bench.err | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('nsingle', $v, round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), d['config']['state_crc'], round(d['stages_ms']['convec'],3), round(d['stages_ms']['mxlayr'],3), round(d['spunup']['ms_per_step'],3), round(d['spunup']['stages_ms']['convec'],3), round(d['spunup']['stages_ms']['mxlayr'],3), d['spunup'].get('state_crc'))"; done
