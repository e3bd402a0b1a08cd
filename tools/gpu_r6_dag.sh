#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r6_dag.sh TAG  -- phys_dag on/off A/B on one box: ms per step (first block, median), state CRC, class times
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/$1; mkdir -p $O
for rep in 1 2 3; do
  for o in ${DAG_OPTS:-phys_dag=1 phys_dag=0}; do
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --spunup-steps 0 --no-dyncore-compare --opt ${o/,/ --opt } 2>$O/err.txt | grep '^{' | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print(sys.argv[1], 'first', round(d['ms_per_step'], 3), 'median', round(d.get('ms_per_step_median', 0), 3), 'min', round(d.get('ms_per_step_min', 0), 3), d['config'].get('state_crc'), {k: round(v, 3) for k, v in d.get('stages_ms', {}).items() if k in ('cmnfld', 'difest', 'eddtra')})" $o
  done
done 2>&1 | tee $O/ab.txt
for c in tnx2v1s; do
  for o in phys_dag=1 phys_dag=0; do
    python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --spunup-steps 0 --no-dyncore-compare --opt ${o/,/ --opt } 2>/dev/null | grep '^{' | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print(sys.argv[1], sys.argv[2], 'first', round(d['ms_per_step'], 3), 'median', round(d.get('ms_per_step_median', 0), 3), d['config'].get('state_crc'))" $c $o
  done
done 2>&1 | tee -a $O/ab.txt
