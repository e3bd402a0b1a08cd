#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_pmc.sh <outdir> "<kernel substrings>" "<bench opts>" CTR1,CTR2 CTR3 ...
# one rocprofv3 --pmc pass of a short bench run per counter group (counters only: no trace domains), then the
# per-kernel averages of the kernels matching the substrings
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; PATS=$2; OPTS=$3; shift 3
mkdir -p $O; cd $R
n=0
for grp in "$@"; do
  n=$((n+1))
  rocprofv3 --pmc ${grp//,/ } --output-format csv -d $O/p$n -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dyncore-compare $OPTS > $O/p$n.log 2>&1
done
python3 tools/pmc_summary.py $O $PATS
