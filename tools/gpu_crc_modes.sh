#!/bin/bash
# usage (GPU box): tools/gpu_crc_modes.sh -- the state checksum of bench.py's other modes (frozen diffusivities, dynamical core, frozen slopes, 24 tracers at tnx1v4 size)
# with the stages side by side (phys_dag = 7), without (0) and with every stage on one stream (overlap = 0): the three must agree
cd "$GRAFT_REPO_ROOT"
for mode in "--frozen-diffusivities" "--physics dyncore" "--slopes frozen" "--config tnx1v4s --tracers 24 --steps 6"; do
  for o in phys_dag=7 phys_dag=0 overlap=0; do
    python3 bench.py $mode --steps 12 --warmup 3 --blocks 1 --no-cpu-baseline --spunup-steps 0 --no-dyncore-compare --opt $o 2>/dev/null | grep '^{' | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(sys.argv[1], '|', sys.argv[2], d['config'].get('state_crc'), round(d['ms_per_step'], 3))" "$mode" $o
  done
done
