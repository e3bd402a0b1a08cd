#!/bin/bash
# the GPU suite under kernel variants (BLOMGPU_OPTS sets an option on every context of the process)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
for v in "barotp_block=2" "convec_nsingle=-1" "barotp_block=0"; do
  BLOMGPU_OPTS="$v" timeout 1500 python3 -m pytest tests -m gpu -x -q -k "not four_barotropic and not round6_kernel_variants" > $O/suite_$v.txt 2>&1
  echo "$v: $(grep -aE 'passed|failed' $O/suite_$v.txt | tail -1)"; grep -a "Error\|assert" $O/suite_$v.txt | head -5
done
