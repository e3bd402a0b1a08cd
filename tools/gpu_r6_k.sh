#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_variants.py -m gpu -x -q -k "round6" > $O/gpu_sub.txt 2>&1; grep -aE "passed|failed" $O/gpu_sub.txt | tail -1
for rep in 1 2 3; do for v in 0 2; do python3 bench.py --no-cpu-baseline --no-dyncore-compare --spunup-steps 0 --opt mom_aw_split=$v 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('split', $v, round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), d['config']['state_crc'], round(d['stages_ms']['momtum'],3))"; done; done
