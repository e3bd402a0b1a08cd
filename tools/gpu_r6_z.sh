#!/bin/bash
# usage (on the GPU box): tools/gpu_r6_z.sh <tag> [pytest -k expression]
# the golden and stage-parity tests that cover what was touched, then the bench line's stage times and the kernel-trace averages
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:?tag}; K=${2:-}
O=gpurun_out/$T; mkdir -p $O
python -m pytest tests/test_gpu_golden.py -x -q > $O/t_golden.log 2>&1; tail -2 $O/t_golden.log
if [ -n "$K" ]; then python -m pytest tests -m gpu -x -q -k "$K" > $O/t_k.log 2>&1; grep -a "passed\|failed" $O/t_k.log | tail -2; fi
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/b.err | grep "^{" > $O/b.json
python3 - <<PY
import json
d=json.loads(open("$O/b.json").read())
print(d["ms_per_step"], d["ms_per_step_blocks"], d["spunup"]["ms_per_step"])
print({k: round(v,3) for k,v in d["stages_ms"].items()})
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --blocks 1 --spunup-steps 0 --no-dyncore-compare > /dev/null 2>$GRAFT_REPO_ROOT/$O/kt.err
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv,glob
f=glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in rows[:32]:
    print(f"{r['Name'].split('(')[0][:48]:50s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f}")
PY
