#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_golden.py tests/test_xcheck_mxlayr.py tests/test_xcheck_fullstep.py -m gpu -x -q > $O/gpu_sub.txt 2>&1; grep -aE "passed|failed" $O/gpu_sub.txt | tail -1; grep -a "Error\|assert" $O/gpu_sub.txt | head -8
L="scan,detrain iteration,detrain rest,entrain walk,entrain rest,tail"
BLOMGPU_LIB=$GRAFT_REPO_ROOT/tools/probes/libblomgpu_kprof.so python3 tools/kprof_waves.py --sel 3 --nt 7 --labels "$L" --steps 12 > $O/mxl_rest.txt 2>&1; tail -12 $O/mxl_rest.txt | head -3; tail -1 $O/mxl_rest.txt
BLOMGPU_LIB=$GRAFT_REPO_ROOT/tools/probes/libblomgpu_kprof.so python3 tools/kprof_waves.py --sel 3 --nt 7 --labels "$L" --steps 12 --spinup 1000 > $O/mxl_spun.txt 2>&1; tail -12 $O/mxl_spun.txt | head -3; tail -1 $O/mxl_spun.txt
python3 bench.py --no-cpu-baseline --no-dyncore-compare 2>$O/bench.err | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('bench', round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), d['config']['state_crc'], round(d['stages_ms']['mxlayr'],3), d['spunup']['ms_per_step'], d['spunup'].get('stages_ms',{}).get('mxlayr'))"
