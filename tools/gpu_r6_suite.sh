#!/bin/bash
# the GPU suite + smoke + the default bench line on one box (what the driver runs at round end)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
tools/probes/copy_rate > $O/copy_rate.txt 2>&1
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/gpu_suite_full.txt 2>&1
grep -aE "passed|failed" $O/gpu_suite_full.txt | tail -1 | tee $O/gpu_suite.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 - $O/bench.json <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l)
        print({k: d.get(k) for k in ("value", "ms_per_step", "ms_per_step_median", "ms_per_step_min", "ms_per_step_max")}, d["roofline"]["kernel"], d["roofline"]["frac"])
        print({k: round(v, 3) for k, v in d["stages_ms"].items()})
        print("spunup", (d.get("spunup") or {}).get("ms_per_step"), "dyncore", (d.get("dyncore_only") or {}).get("ms_per_step"), "cpu", (d.get("cpu_baseline") or {}).get("reference_only_ms"))
PY
tools/probes/copy_rate >> $O/copy_rate.txt 2>&1; cat $O/copy_rate.txt
