#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_round5_final.sh <tag>
# everything profiles/<tag>_* is made from, in one job on one box: the box's copy rate, the GPU suite, the kernel trace and HBM counters
# of the default bench run (overlap on) and of --opt overlap=0 (one kernel at a time: the bounds table), the SQ counters, the bench lines.
cd $GRAFT_REPO_ROOT
T=$1; O=gpurun_out/$T; mkdir -p $O
tools/probes/copy_rate > $O/copy_rate.txt 2>&1
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/gpu_suite_full.txt 2>&1
grep -aE "passed|failed" $O/gpu_suite_full.txt | tail -1 > $O/gpu_suite.txt
tools/gpu_profile_round.sh $T > $O/round.log 2>&1
CONFIG=chanovl0 tools/gpu_profile_round.sh ${T}o "--opt overlap=0" > $O/round_ovl0.log 2>&1
tools/gpu_pmc.sh ${T}_sq "k_" "--opt overlap=0" SQ_WAVES,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR,SQ_INSTS_LDS,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_VALU > $O/sq_counters.txt 2>&1
python3 bench.py --steps 20 > $O/bench.json 2> $O/bench.err
python3 bench.py --steps 20 --no-cpu-baseline --frozen-diffusivities > $O/bench_frozen.json 2>/dev/null
python3 bench.py --steps 20 --no-cpu-baseline --physics dyncore > $O/bench_dyncore.json 2>/dev/null
python3 bench.py --steps 20 --no-cpu-baseline --opt overlap=0 > $O/bench_ovl0.json 2>/dev/null
python3 bench.py --steps 10 --no-cpu-baseline --config tnx1v4s --tracers 3 > $O/bench_tnx1v4s_3tr.json 2>/dev/null
python3 bench.py --steps 10 --no-cpu-baseline --config tnx1v4s --tracers 24 > $O/bench_tnx1v4s_24tr.json 2>/dev/null
python3 bench.py --steps 10 --no-cpu-baseline --config tnx2v1s > $O/bench_tnx2v1s.json 2>/dev/null
python3 bench.py --steps 10 --no-cpu-baseline --config hybrid > $O/bench_hybrid.json 2>/dev/null
NTR=24 CONFIG=tnx1v4s tools/gpu_profile_round.sh ${T}_24tr "--config tnx1v4s --tracers 24" > $O/round_24tr.log 2>&1
tools/probes/copy_rate >> $O/copy_rate.txt 2>&1
cat $O/copy_rate.txt $O/gpu_suite.txt
for f in $O/bench*.json; do python3 -c "
import sys,json
for l in open('$f'):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('$f',d['value'],d['ms_per_step'],(d.get('dyncore_only') or {}).get('ms_per_step'), d['roofline']['frac'])"; done
