#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_variants.py tests/test_gpu_stage_parity.py tests/test_gpu_golden.py -m gpu -x -q > $O/gpu_sub.txt 2>&1; grep -aE "passed|failed" $O/gpu_sub.txt | tail -1
trace() {
  n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dyncore-compare --blocks 1 --spunup-steps 0 "$@" > $O/kt_$n.log 2>&1
  python3 tools/kstats.py $O/kt_$n k_mom_ > $O/kstats_$n.txt 2>&1
}
trace split0 --opt overlap=0 --opt mom_aw_split=0
trace split1 --opt overlap=0
grep -h "k_mom_visc" $O/kstats_*.txt
for rep in 1 2; do for v in 0 1; do python3 bench.py --no-cpu-baseline --no-dyncore-compare --spunup-steps 0 --opt mom_aw_split=$v 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('split', $v, round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), d['config']['state_crc'], round(d['stages_ms']['momtum'],3))"; done; done
for cfg in tnx2v1s; do python3 bench.py --config $cfg --no-cpu-baseline --steps 10 --spunup-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$cfg', round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), d['config']['state_crc'])"; done
