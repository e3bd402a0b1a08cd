#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_ksweep.sh <outdir> <pattern> "<opts of run 1>" "<opts of run 2>" ...
# one rocprofv3 kernel trace of a short bench run per option set; prints the kernels matching <pattern>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; PAT=$2; shift 2
mkdir -p $O; cd $R
n=0
for opts in "$@"; do
  n=$((n+1))
  echo "=== run $n: $opts"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$n -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-dyncore-compare $opts > $O/kt$n.log 2>&1
  python3 tools/kstats.py $O/kt$n $PAT
  grep -o '"state_crc": "[0-9a-f]*"' $O/kt$n.log
done
