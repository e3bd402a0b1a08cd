#!/bin/bash
# usage (on the GPU box): tools/gpu_ab_libs.sh <tag> <name=lib.so> <name=lib.so> ... [-- bench options]
# The same short bench run under rocprofv3 --kernel-trace once per library (BLOMGPU_LIB), then per-kernel average durations side by
# side, and the bench's own ms/step (no profiler) for each, alternating twice.  Same box, same job: what an A/B between libraries needs.
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:?tag}; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" == "--" ] && shift
BOPT="$*"
export TMPDIR=/tmp
for nl in "${LIBS[@]}"; do
  n=${nl%%=*}; l=${nl#*=}
  export BLOMGPU_LIB=$GRAFT_REPO_ROOT/$l
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --blocks 1 --spunup-steps 0 --no-dyncore-compare $BOPT > $O/kt_$n.log 2>&1)
done
for rep in 1 2; do
  for nl in "${LIBS[@]}"; do
    n=${nl%%=*}; l=${nl#*=}
    BLOMGPU_LIB=$GRAFT_REPO_ROOT/$l python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --spunup-steps 0 --no-dyncore-compare $BOPT 2>/dev/null | grep "^{" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$n', round(d['ms_per_step'],4), d['ms_per_step_blocks'], d.get('state_crc'))"
  done
done
python3 - "$O" "${LIBS[@]}" <<'PY'
import csv, glob, sys
O, libs = sys.argv[1], [x.split("=")[0] for x in sys.argv[2:]]
tab = {}
for n in libs:
    f = glob.glob(f"{O}/kt_{n}/**/*kernel_stats.csv", recursive=True)
    if not f: print("no stats for", n); continue
    for r in csv.DictReader(open(f[0])):
        nm = r["Name"].split("(")[0].replace("void ", "")
        tab.setdefault(nm, {})[n] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3)
rows = sorted(tab.items(), key=lambda kv: -max(v[2] for v in kv[1].values()))
print(f"{'kernel':46s}" + "".join(f"{n:>22s}" for n in libs))
for nm, d in rows[:45]:
    print(f"{nm[:46]:46s}" + "".join((f"{d[n][0]:7d} x {d[n][1]:9.1f} us" if n in d else f"{'-':>22s}") for n in libs))
print("total kernel ms:", {n: round(sum(d[n][2] for d in tab.values() if n in d) / 1e3, 2) for n in libs})
PY
