#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
export BLOMGPU_LIB=$GRAFT_REPO_ROOT/tools/probes/libblomgpu_kprof.so
python3 tools/kprof_waves.py --sel 1 --steps 12 --spinup 1000 --save $O/pgf_spun.npy > $O/pgf_spun.txt 2>&1; tail -9 $O/pgf_spun.txt
python3 tools/kprof_waves.py --sel 1 --steps 12 > $O/pgf_rest.txt 2>&1; tail -9 $O/pgf_rest.txt
