cd $GRAFT_REPO_ROOT
tools/gpu_ksweep.sh r05_ks7 "k_diapfl_mom k_cmn_nslope" "--opt overlap=0" > gpurun_out/r05_ks7.txt 2>&1
grep -aE "===|k_|crc" gpurun_out/r05_ks7.txt | cut -c1-80
timeout 900 python3 -m pytest tests -m gpu -x -q -k "stage_parity or cmnfld or full_step or full_size" 2>&1 | grep -aE "passed|failed" | tail -1
