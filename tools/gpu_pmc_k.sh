#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_pmc_k.sh <outdir> <kernel regex> "<bench opts>" CTR1,CTR2 CTR3 ...
# as gpu_pmc.sh, but the counters are collected for the kernels matching the regex only (the other dispatches run unprofiled)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; RE=$2; OPTS=$3; shift 3
mkdir -p $O; cd $R
n=0
for grp in "$@"; do
  n=$((n+1))
  rocprofv3 --pmc ${grp//,/ } --kernel-include-regex "$RE" --output-format csv -d $O/p$n -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dyncore-compare $OPTS > $O/p$n.log 2>&1
done
python3 tools/pmc_summary.py $O ${RE//|/ }
