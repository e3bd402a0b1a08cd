#!/bin/bash
# usage (on the GPU box, from the repo root): [CONFIG=channel] tools/gpu_profile_round.sh <tag> ["bench opts"]
# the profiles of a round: kernel trace of the default bench run and the two HBM counter passes (counters only), summarised by
# tools/prof_summarize.py; everything lands in gpurun_out/profiles_<tag>/ (copy the three summaries into profiles/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; OPTS=$2; O=$R/gpurun_out/profiles_$TAG
mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dyncore-compare $OPTS > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-dyncore-compare $OPTS > $O/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-dyncore-compare $OPTS > $O/pw.log 2>&1
KS=$(ls $O/kt/*/*_kernel_trace.csv | head -1); PF=$(ls $O/pf/*/*_counter_collection.csv | head -1); PW=$(ls $O/pw/*/*_counter_collection.csv | head -1)
CONFIG=${CONFIG:-channel} python3 tools/prof_summarize.py $TAG $KS $PF $PW > $O/summary.log 2>&1
cp profiles/${TAG}_* $O/
tail -3 $O/kt.log | head -1 | cut -c1-400
head -40 $O/${TAG}_${CONFIG:-channel}_kernel_stats.txt
