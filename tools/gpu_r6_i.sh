#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
trace() {
  n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dyncore-compare --blocks 1 --spunup-steps 0 "$@" > $O/kt_$n.log 2>&1
  python3 tools/kstats.py $O/kt_$n k_mom_ > $O/kstats_$n.txt 2>&1
}
trace aw0 --opt overlap=0
trace aw1 --opt overlap=0 --opt mom_force_aw=1
trace aw0b --opt overlap=0
trace aw1b --opt overlap=0 --opt mom_force_aw=1
grep -h "k_mom_visc" $O/kstats_*.txt
