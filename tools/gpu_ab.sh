#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_ab.sh "<bench opts>" OPT=VAL OPT=VAL ...
# the same short bench run once per option setting, alternating twice: ms per step, one line each
cd "$GRAFT_REPO_ROOT" || exit 1
OPTS=$1; shift
for rep in 1 2; do
  for o in "$@"; do
    python3 bench.py $OPTS --steps 20 --no-cpu-baseline --opt $o 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step'],3), {k: round(v,3) for k,v in d.get('stages_ms',{}).items() if k in ('ale_regrid_remap','ndiff')}, {k: round(v,3) for k,v in d.get('ndiff_kernels_ms',{}).items()})" $o
  done
done
