cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_suite_v3_full.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r05_gpu_suite_v3_full.txt | tail -5
