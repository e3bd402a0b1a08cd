#!/bin/bash
# usage (GPU box, repo root): tools/gpu_timeline.sh TAG [bench opts]  -- kernel trace of a short bench run, then the timeline of one step
# (start relative to the step's first kernel, duration, queue) by tools/step_timeline.py
R="${GRAFT_REPO_ROOT:?}"; cd "$R" || exit 1
O=$R/gpurun_out/$1; shift; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $O/kt -o run -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --blocks 1 --spunup-steps 0 --no-dyncore-compare "$@" > $O/kt.log 2>&1)
f=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python3 tools/step_timeline.py "$f" > $O/timeline.txt
gzip -c "$f" > $O/kernel_trace.csv.gz; rm -rf $O/kt
tail -n +1 $O/timeline.txt | head -150
