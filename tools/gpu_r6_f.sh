#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
trace() {
  n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dyncore-compare --blocks 1 --spunup-steps 0 "$@" > $O/kt_$n.log 2>&1
  python3 tools/kstats.py $O/kt_$n k_pgf_uv > $O/kstats_$n.txt 2>&1
}
for v in 0 5 6 7 8; do trace ring$v --opt overlap=0 --opt pgf_uv_ring=$v; done
for v in 0 5 7 8; do trace sp_ring$v --spinup 600 --opt overlap=0 --opt pgf_uv_ring=$v; done
grep -h "k_pgf_uv" $O/kstats_*.txt | grep -v "calls     1 "
for v in 7 8; do
  BLOMGPU_OPTS=pgf_uv_ring=$v timeout 900 python3 -m pytest tests/test_gpu_stage_parity.py tests/test_gpu_golden.py -m gpu -x -q > $O/gpu_ring$v.txt 2>&1
  grep -aE "passed|failed" $O/gpu_ring$v.txt | tail -1
done
export BLOMGPU_LIB=$GRAFT_REPO_ROOT/tools/probes/libblomgpu_kprof.so
for sp in 0 600; do
  python3 tools/kprof_waves.py --spinup $sp --sel 2 --nt 6 --labels "setup,limiter,solver,mixing,massless layers" 2>/dev/null | tee -a $O/kprof_diapfl.txt
done
