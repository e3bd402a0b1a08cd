#!/bin/bash
# usage (on the GPU box): tools/gpu_r6_b.sh <tag>  -- k_pgf_uv variants (reuse of the previous level's EOS values, 4 waves per SIMD), the
# re-associated pressure scan, the new tests and the bench line with blocks / spunup
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
tools/probes/copy_rate > $O/copy_rate.txt 2>&1
export TMPDIR=/tmp
trace() {
  n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dyncore-compare --blocks 1 --spunup-steps 0 "$@" > $O/kt_$n.log 2>&1
  python3 tools/kstats.py $O/kt_$n k_ > $O/kstats_$n.txt 2>&1
}
trace old --opt overlap=0
trace oldreuse --opt overlap=0 --opt pgf_reuse=1
for v in 11 12 13 14 111 112 113 114 101 103; do trace ring$v --opt overlap=0 --opt pgf_uv_ring=$v; done
trace scan --opt overlap=0 --opt scan_reassoc=1
grep -h "k_pgf_uv\|k_pscan" $O/kstats_*.txt | awk '{print FILENAME, $0}' 
for f in $O/kstats_*.txt; do echo "$f: $(grep -h 'k_pgf_uv\|k_pscan' $f | awk '{printf "%s %s us x%s | ", $2, $6, $4}')"; done | tee $O/pgf_summary.txt
timeout 900 python3 -m pytest tests/test_gpu_fortran_host.py tests/test_xcheck_difest.py -m gpu -x -q -k "difest_live or spun_up" > $O/newtests.txt 2>&1
tail -3 $O/newtests.txt
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 - $O/bench.json <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l)
        print({k: d.get(k) for k in ("ms_per_step", "ms_per_step_blocks", "ms_per_step_median", "ms_per_step_min", "ms_per_step_max")})
        print("spunup", d.get("spunup")); print("cpu", {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k != "stages_ms"})
PY
tools/probes/copy_rate >> $O/copy_rate.txt 2>&1
cat $O/copy_rate.txt
