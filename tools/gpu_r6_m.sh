#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_xcheck_difest.py tests/test_gpu_golden.py tests/test_gpu_fortran_host.py -m gpu -x -q > $O/gpu_sub.txt 2>&1; grep -aE "passed|failed" $O/gpu_sub.txt | tail -1; grep -a "Error\|assert" $O/gpu_sub.txt | head -5
trace() {
  n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dyncore-compare --blocks 1 --spunup-steps 0 "$@" > $O/kt_$n.log 2>&1
  python3 tools/kstats.py $O/kt_$n k_ > $O/kstats_$n.txt 2>&1
}
trace rhs1 --opt overlap=0
grep -h "k_dfi_" $O/kstats_*.txt
for v in 1 0; do python3 bench.py --no-cpu-baseline --no-dyncore-compare --spunup-steps 0 --opt rhsctp=$v 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('rhsctp', $v, round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), d['config']['state_crc'], round(d['stages_ms']['difest'],3))"; done
