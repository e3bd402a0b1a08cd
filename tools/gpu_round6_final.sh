#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_round6_final.sh <tag>
# Everything profiles/<tag>_* is made from, in one job on one box: the box's copy rate (first and last), the headline bench line, the GPU suite, the kernel trace and
# HBM counters of the default bench run (two streams) and of --opt overlap=0 (one kernel at a time: the bounds table), the SQ counters, the
# bench lines of the other configurations.  A missing bench line or a failed step makes the script exit non-zero; stderr is kept.
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set}" || exit 1
T=${1:?usage: gpu_round6_final.sh <tag>}
O="$GRAFT_REPO_ROOT/gpurun_out/$T"; mkdir -p "$O" || exit 1
bad=0
tools/probes/copy_rate > "$O/copy_rate.txt" 2> "$O/copy_rate.err" || bad=1
# the headline line first, as the driver runs it (its own invocation on a box that has not been under load for minutes: the pool's boxes slow down by 3 - 5 %
# under sustained load -- the same library 6.37 -> 6.75 ms over an hour of A/B runs, back at 6.37 after a quarter of an hour's rest), then the suite
bench() { n=$1; shift; python3 bench.py "$@" > "$O/bench_$n.json" 2> "$O/bench_$n.err"; grep -q '^{' "$O/bench_$n.json" || { echo "bench $n produced no line" >&2; bad=1; }; }
bench default --steps 20
timeout 1500 python3 -m pytest tests -m gpu -x -q > "$O/gpu_suite_full.txt" 2>&1 || bad=1
grep -aE "passed|failed" "$O/gpu_suite_full.txt" | tail -1 > "$O/gpu_suite.txt"
tools/gpu_profile_round.sh "$T" "--blocks 1 --spunup-steps 0" > "$O/round.log" 2>&1 || bad=1
CONFIG=chanovl0 tools/gpu_profile_round.sh "${T}o" "--opt overlap=0 --blocks 1 --spunup-steps 0" > "$O/round_ovl0.log" 2>&1 || bad=1
tools/gpu_pmc.sh "${T}_sq" "k_" "--opt overlap=0 --blocks 1 --spunup-steps 0" SQ_WAVES,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR,SQ_INSTS_LDS,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_VALU > "$O/sq_counters.txt" 2> "$O/sq_counters.err" || bad=1
bench frozen --steps 20 --no-cpu-baseline --frozen-diffusivities
bench dyncore --steps 20 --no-cpu-baseline --physics dyncore
bench ovl0 --steps 20 --no-cpu-baseline --opt overlap=0
bench calm --steps 20 --no-cpu-baseline --forcing calm
bench spinup600 --steps 20 --spinup 600
bench tnx1v4s_3tr --steps 10 --no-cpu-baseline --config tnx1v4s --tracers 3
bench tnx1v4s_24tr --steps 10 --no-cpu-baseline --config tnx1v4s --tracers 24
bench tnx2v1s --steps 10 --no-cpu-baseline --config tnx2v1s
bench hybrid --steps 10 --no-cpu-baseline --config hybrid
NTR=24 CONFIG=tnx1v4s tools/gpu_profile_round.sh "${T}_24tr" "--config tnx1v4s --tracers 24 --blocks 1 --spunup-steps 0" > "$O/round_24tr.log" 2>&1 || bad=1
# the long run against the reference's (profiles/<tag>_longrun.txt is assembled from these): NorESM's defaults (rhsctp on), the options of
# round 5 (rhsctp off: the run whose extreme samples that round's review asked about) step by step around step 600, the calm forcing
python3 tools/longrun_full_physics.py --steps 1200 --every 200 --golden tests/golden/channel_tke_live_long_crc.json > "$O/longrun_default.txt" 2> "$O/longrun.err" || bad=1
python3 tools/longrun_full_physics.py --rhsctp 0 --steps 600 --every 1 --from 588 --budget-step 300 --golden tests/golden/channel_tke_live_long_rhsctp0_crc.json > "$O/longrun_rhsctp0.txt" 2>> "$O/longrun.err" || bad=1
python3 tools/longrun_full_physics.py --steps 1200 --every 200 --forcing calm > "$O/longrun_calm.txt" 2>> "$O/longrun.err" || bad=1
tools/probes/copy_rate >> "$O/copy_rate.txt" 2>> "$O/copy_rate.err" || bad=1
cat "$O/copy_rate.txt" "$O/gpu_suite.txt"
for f in "$O"/bench_*.json; do python3 - "$f" <<'PY' || bad=1
import sys, json
ok = False
for l in open(sys.argv[1]):
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); ok = True
        print(sys.argv[1].split('/')[-1], d['value'], round(d['ms_per_step'], 4), d.get('ms_per_step_median'), (d.get('spunup') or {}).get('ms_per_step'),
              (d.get('dyncore_only') or {}).get('ms_per_step'), d['roofline']['frac'], (d.get('cpu_baseline') or {}).get('reference_only_ms'))
sys.exit(0 if ok else 1)
PY
done
exit $bad
