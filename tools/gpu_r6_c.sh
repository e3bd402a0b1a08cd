#!/bin/bash
# per-wave phase timestamps of k_pgf_uv (old / ring variants), at the bench state and a spun-up one
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export BLOMGPU_LIB=$GRAFT_REPO_ROOT/tools/probes/libblomgpu_kprof.so
for sp in 0 600; do
  for v in 0 1 3 2; do
    python3 tools/kprof_waves.py --spinup $sp --opt pgf_uv_ring=$v --save $O/waves_sp${sp}_v$v.npy 2>/dev/null | tee -a $O/kprof.txt
  done
done
