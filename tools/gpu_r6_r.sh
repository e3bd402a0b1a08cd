#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dyncore-compare --blocks 1 --spunup-steps 0 --spinup 1000 --opt overlap=0 > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/ktrace_tail.py $O/kt 0.02 > $O/kstats.txt 2>&1
head -50 $O/kstats.txt
