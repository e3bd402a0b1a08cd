#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_r6_ab.sh <tag> [suite]
# Round 6, first measurement: what the global-address-space pointers (blomgpu_internal.h, PtrTable) and the k_pgf_uv_ring variants do,
# kernel by kernel, on ONE box: kernel traces (one kernel at a time: --opt overlap=0) of round 5's library (tools/probes/libblomgpu_r05.so,
# built from the round-5 commit), of this round's with the old k_pgf_uv, and of the four ring variants; plain bench lines alternating
# between the two libraries; then (with `suite`) the GPU suite and the parity subset under the ring variants.
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
tools/probes/copy_rate > $O/copy_rate.txt 2>&1
export TMPDIR=/tmp
trace() {   # trace <name> <bench opts...>
  n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dyncore-compare "$@" > $O/kt_$n.log 2>&1
  python3 tools/kstats.py $O/kt_$n k_ > $O/kstats_$n.txt 2>&1
}
if [ -f tools/probes/libblomgpu_r05.so ]; then
  export BLOMGPU_LIB=$GRAFT_REPO_ROOT/tools/probes/libblomgpu_r05.so
  trace r05 --opt overlap=0
  unset BLOMGPU_LIB
fi
trace new --opt overlap=0
for v in 1 2 3 4; do trace ring$v --opt overlap=0 --opt pgf_uv_ring=$v; done
for rep in 1 2; do
  if [ -f tools/probes/libblomgpu_r05.so ]; then
    BLOMGPU_LIB=$GRAFT_REPO_ROOT/tools/probes/libblomgpu_r05.so python3 bench.py --steps 20 --no-cpu-baseline > $O/bench_r05_$rep.json 2>/dev/null
  fi
  python3 bench.py --steps 20 --no-cpu-baseline > $O/bench_new_$rep.json 2>/dev/null
  python3 bench.py --steps 20 --no-cpu-baseline --opt pgf_uv_ring=3 > $O/bench_ring3_$rep.json 2>/dev/null
  python3 bench.py --steps 20 --no-cpu-baseline --opt pgf_uv_ring=2 > $O/bench_ring2_$rep.json 2>/dev/null
done
for f in $O/bench_*.json; do python3 - "$f" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l); print(sys.argv[1].split('/')[-1], round(d['ms_per_step'], 3), (d.get('dyncore_only') or {}).get('ms_per_step'), d['config']['state_crc'], {k: round(v, 3) for k, v in d['stages_ms'].items()})
PY
done | tee $O/bench_summary.txt
if [ "$2" = "suite" ]; then
  timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/gpu_suite_full.txt 2>&1
  grep -aE "passed|failed" $O/gpu_suite_full.txt | tail -1 | tee $O/gpu_suite.txt
  for v in 2 3; do
    BLOMGPU_OPTS=pgf_uv_ring=$v timeout 900 python3 -m pytest tests/test_gpu_stage_parity.py tests/test_gpu_golden.py -m gpu -x -q > $O/gpu_ring$v.txt 2>&1
    grep -aE "passed|failed" $O/gpu_ring$v.txt | tail -1 | tee -a $O/gpu_suite.txt
  done
fi
tools/probes/copy_rate >> $O/copy_rate.txt 2>&1
cat $O/copy_rate.txt
for n in r05 new; do echo "== $n"; sort -t' ' -k1,1 $O/kstats_$n.txt | awk '{print $1, $3, $5}' | head -100; done > $O/kstats_pairs.txt
grep -h "k_pgf_uv" $O/kstats_*.txt
if [ "$2" = "suite" ]; then
  python3 tools/longrun_full_physics.py --steps 600 --every 1 --from 588 --budget-step 300 --golden tests/golden/channel_tke_live_long_crc.json > $O/longrun.txt 2> $O/longrun.err
  tail -25 $O/longrun.txt
fi
