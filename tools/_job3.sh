cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_job4_tests.txt 2>&1
grep -E "passed|failed" gpurun_out/r05_job4_tests.txt | tail -2
tools/gpu_ksweep.sh r05_ks_tr2 "remap" "--config tnx1v4s --tracers 24" "" "--config tnx1v4s --tracers 3" > gpurun_out/r05_ks_tr2.txt 2>&1
grep -E "===|remap|mxl|convec_col" gpurun_out/r05_ks_tr2.txt
for n in 24 3; do python3 bench.py --steps 10 --no-cpu-baseline --config tnx1v4s --tracers $n 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print($n, d['ms_per_step'], d['config']['state_crc'])"; done
python3 bench.py --steps 20 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('channel', d['ms_per_step'], d['config']['state_crc'], d.get('dyncore_only_ms_per_step'))"
