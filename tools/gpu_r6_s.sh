#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
export BLOMGPU_LIB=$GRAFT_REPO_ROOT/tools/probes/libblomgpu_kprof.so
L="scan,detrain iteration,detrain rest,entrain walk,entrain rest,tail"
python3 tools/kprof_waves.py --sel 3 --nt 7 --labels "$L" --steps 12 > $O/mxl_rest.txt 2>&1; tail -12 $O/mxl_rest.txt
python3 tools/kprof_waves.py --sel 3 --nt 7 --labels "$L" --steps 12 --spinup 1000 --save $O/mxl_spun.npy > $O/mxl_spun.txt 2>&1; tail -12 $O/mxl_spun.txt
