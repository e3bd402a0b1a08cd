#!/bin/bash
# usage: tools/install_profiles.sh <tag> <round>      (e.g. r06c r06)
# Copies what tools/gpu_round6_final.sh <tag> brought back under gpurun_out/ into profiles/<round>_*: the kernel-trace summaries, the HBM
# counter summaries and the SQ counters of the three profiled runs, the bench lines, the copy rate of the box; then rebuilds the bounds
# table and the long-run text from them.
set -e
T=${1:?tag}; R=${2:?round}
cd "$(dirname "$0")/.."
G=gpurun_out
for f in kernel_stats.txt pmc_hbm_traffic.txt class_traffic.json; do
  cp $G/profiles_$T/${T}_channel_$f profiles/${R}_channel_$f
  cp $G/profiles_${T}o/${T}o_chanovl0_$f profiles/${R}_chanovl0_$f
  cp $G/profiles_${T}_24tr/${T}_24tr_tnx1v4s_$f profiles/${R}_24tr_tnx1v4s_$f
done
cp $G/$T/sq_counters.txt profiles/${R}_chanovl0_sq_counters.txt
line() { grep -a '^{' "$1" | tail -1; }
line $G/$T/bench_default.json > profiles/${R}_channel_bench.json
for v in calm dyncore frozen ovl0 spinup600; do line $G/$T/bench_$v.json > profiles/${R}_channel_bench_$v.json; done
line $G/$T/bench_hybrid.json > profiles/${R}_hybrid_bench.json
line $G/$T/bench_tnx1v4s_3tr.json > profiles/${R}_tnx1v4s_bench.json
line $G/$T/bench_tnx1v4s_24tr.json > profiles/${R}_24tr_tnx1v4s_bench.json
line $G/$T/bench_tnx2v1s.json > profiles/${R}_tnx2v1s_bench.json
{ echo "# tools/probes/copy_rate at the start and at the end of the job ($T), GPU suite of the same job"; cat $G/$T/copy_rate.txt $G/$T/gpu_suite.txt; } > profiles/${R}_box.txt
python3 tools/assemble_longrun.py $G/$T > profiles/${R}_longrun.txt
RATE=$(awk '/copy_rate_TBps/ {s += $2; n++} END {printf "%.2f", s / n}' $G/$T/copy_rate.txt)
ROUND=$(echo ${R#r} | sed "s/^0//") python3 tools/bounds_table.py profiles/${R}_chanovl0_kernel_stats.txt profiles/${R}_chanovl0_pmc_hbm_traffic.txt profiles/${R}_chanovl0_sq_counters.txt $RATE tools/bounds_notes_${R}.md > profiles/${R}_bounds.md
echo "installed profiles/${R}_* from $T (copy rate $RATE TB/s)"
