#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_ksweep_pmc.sh <outdir> "<kernel substrings>" "<opts of run 1>" "<opts of run 2>" ...
# per option set: a kernel trace (durations) and FETCH_SIZE / WRITE_SIZE passes of a short bench run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; PATS=$2; shift 2
mkdir -p $O; cd $R
n=0
for opts in "$@"; do
  n=$((n+1))
  echo "=== run $n: $opts"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$n -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline $opts > $O/kt$n.log 2>&1
  for p in $PATS; do python3 tools/kstats.py $O/kt$n $p | grep -v "^total"; done
  grep -o '"state_crc": "[0-9a-f]*"' $O/kt$n.log
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf$n -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $opts > $O/pf$n.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw$n -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $opts > $O/pw$n.log 2>&1
  mkdir -p $O/pmc$n; mv $O/pf$n $O/pw$n $O/pmc$n/
  python3 tools/pmc_summary.py $O/pmc$n $PATS
done
