#!/bin/bash
# usage (on the GPU box): tools/gpu_opt_sweep.sh <stage key of stages_ms> "<opts 1>" "<opts 2>" ...   -- bench.py per option set
for opts in "$@"; do
  [ "$opts" = "$1" ] && continue
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline $opts 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('$opts', 'ms/step', round(d['ms_per_step'],3), '$1', round(d['stages_ms']['$1'],3), 'crc', d['config']['state_crc'])"
done
