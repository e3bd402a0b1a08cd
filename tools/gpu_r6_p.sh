#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_xcheck_difest.py tests/test_xcheck_fullstep.py tests/test_gpu_fortran_host.py -m gpu -x -q > $O/gpu_sub.txt 2>&1; grep -aE "passed|failed" $O/gpu_sub.txt | tail -1; grep -a "Error\|assert" $O/gpu_sub.txt | head -8
