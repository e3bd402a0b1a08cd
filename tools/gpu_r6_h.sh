#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
trace() {
  n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dyncore-compare --blocks 1 --spunup-steps 0 "$@" > $O/kt_$n.log 2>&1
  python3 tools/kstats.py $O/kt_$n k_convec_velocity > $O/kstats_$n.txt 2>&1
}
for v in 1000 0 1 2 3; do trace ns$v --opt overlap=0 --opt convec_nsingle=$v; trace sp_ns$v --spinup 600 --opt overlap=0 --opt convec_nsingle=$v; done
for f in $O/kstats_*.txt; do echo "$f $(grep -h k_convec_velocity $f | head -2 | awk '{printf "%s x%s | ", $5, $3}')"; done
timeout 1200 python3 -m pytest tests/test_gpu_variants.py -m gpu -x -q -k "round6 or variants_bit_identical" > $O/variants.txt 2>&1; grep -aE "passed|failed" $O/variants.txt | tail -1
for cfg in tnx2v1s tnx1v4s; do python3 bench.py --config $cfg --no-cpu-baseline --steps 10 > $O/bench_$cfg.json 2>/dev/null; done
python3 bench.py --config tnx1v4s --tracers 24 --no-cpu-baseline --steps 10 > $O/bench_tnx1v4s_24.json 2>/dev/null
python3 bench.py --config hybrid --no-cpu-baseline --steps 10 > $O/bench_hybrid.json 2>/dev/null
for f in $O/bench_*.json; do python3 - $f <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l); print(sys.argv[1].split('/')[-1], round(d['ms_per_step'], 3), d.get('ms_per_step_median'), {k: round(v, 3) for k, v in d.get('stages_ms', {}).items()})
PY
done
