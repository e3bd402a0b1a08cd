#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
run() { n=$1; shift; timeout 600 python3 bench.py --no-cpu-baseline --no-dyncore-compare --spunup-steps 0 "$@" 2>$O/bench_$n.err | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$n', round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), d['config']['state_crc'], round(d['stages_ms']['remap'],3), d['roofline'])"; }
run base
BLOMGPU_LIB=$GRAFT_REPO_ROOT/tools/probes/libblomgpu_rt16.so run rt16
run base2
BLOMGPU_LIB=$GRAFT_REPO_ROOT/tools/probes/libblomgpu_rt16.so run rt16b
