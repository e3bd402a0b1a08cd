#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
trace() {
  n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dyncore-compare --blocks 1 --spunup-steps 0 "$@" > $O/kt_$n.log 2>&1
  python3 tools/kstats.py $O/kt_$n k_ > $O/kstats_$n.txt 2>&1
}
trace new --opt overlap=0
trace sp_new --spinup 600 --opt overlap=0
export BLOMGPU_LIB=$GRAFT_REPO_ROOT/tools/probes/libblomgpu_r05.so
trace sp_r05 --spinup 600 --opt overlap=0
unset BLOMGPU_LIB
grep -h "k_convec_velocity" $O/kstats_*.txt
timeout 900 python3 -m pytest tests/test_gpu_stage_parity.py tests/test_gpu_golden.py tests/test_convec.py tests/test_xcheck_mxlayr.py -m gpu -x -q > $O/gpu_sub.txt 2>&1
grep -aE "passed|failed" $O/gpu_sub.txt | tail -1
python3 - $O <<'PY'
import re, sys
O = sys.argv[1]
def load(fn):
    d = {}
    for l in open(fn):
        m = re.match(r"(.{46}) calls\s+(\d+) avg_us\s+([\d.]+)", l)
        if m: d[m.group(1).strip().replace("void ", "")] = (int(m.group(2)), float(m.group(3)))
    return d
a, b = load(f"{O}/kstats_sp_r05.txt"), load(f"{O}/kstats_sp_new.txt")
rows = sorted(((a[k][0] * a[k][1], k) for k in a if k in b), reverse=True)
print("spun-up state (600 steps): kernel, calls, r05 us, now us")
for t, k in rows[:32]:
    print(f"{k:46s} {a[k][0]:6d} {a[k][1]:9.1f} {b[k][1]:9.1f}")
PY
