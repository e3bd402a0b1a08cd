#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_variants.py tests/test_gpu_golden.py tests/test_gpu_stage_parity.py -m gpu -x -q > $O/gpu_sub.txt 2>&1; grep -aE "passed|failed" $O/gpu_sub.txt | tail -1; grep -a "Error\|assert" $O/gpu_sub.txt | head -8
for v in 1 0 1 0; do timeout 600 python3 bench.py --no-cpu-baseline --no-dyncore-compare --spunup-steps 0 --opt barotp_block=$v 2>$O/bench_$v.err | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('barotp_block', $v, round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), d['config']['state_crc'], round(d['stages_ms']['barotp'],3))"; done
