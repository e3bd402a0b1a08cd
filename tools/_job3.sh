cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -x -q -k "many_tracers or tnx1v4s or tracer_batches or mxlayr or convec" > gpurun_out/r05_job3_tests.txt 2>&1
grep -E "passed|failed" gpurun_out/r05_job3_tests.txt | tail -2
tools/gpu_ksweep.sh r05_ks_tr "remap|convec|mxl|tmsmt|pbc|diffus|diapfl" "--config tnx1v4s --tracers 24" "--config tnx1v4s --tracers 24 --opt remap_nfirst=4" "--config tnx1v4s --tracers 24 --opt remap_nfirst=2" > gpurun_out/r05_ks_tr.txt 2>&1
cat gpurun_out/r05_ks_tr.txt
for n in 24 3; do python3 bench.py --steps 10 --no-cpu-baseline --config tnx1v4s --tracers $n 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print($n, d['ms_per_step'])"; done
