#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
B="--opt overlap=0 --blocks 1 --spunup-steps 0"
for v in 0 5 6; do
  tools/gpu_pmc.sh $T/pmc_v$v "k_pgf_uv" "$B --opt pgf_uv_ring=$v" FETCH_SIZE WRITE_SIZE TCC_HIT_sum,TCC_MISS_sum TCP_TCC_READ_REQ_sum,TCP_TOTAL_CACHE_ACCESSES_sum SQ_WAVES,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_VALU > $O/pmc_v$v.txt 2>&1
done
cat $O/pmc_v*.txt
trace() {
  n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dyncore-compare --blocks 1 --spunup-steps 0 "$@" > $O/kt_$n.log 2>&1
  python3 tools/kstats.py $O/kt_$n k_cmn_nslope > $O/kstats_$n.txt 2>&1
}
for v in 4 3 2; do trace nb$v --opt overlap=0 --opt cmn_nslope_nb=$v; trace sp_nb$v --spinup 600 --opt overlap=0 --opt cmn_nslope_nb=$v; done
grep -h "k_cmn_nslope" $O/kstats_*.txt
python3 tools/longrun_full_physics.py --steps 1200 --every 200 --golden tests/golden/channel_tke_live_long_crc.json > $O/longrun_default.txt 2> $O/longrun.err
python3 tools/longrun_full_physics.py --steps 1200 --every 200 --forcing calm > $O/longrun_calm.txt 2>> $O/longrun.err
cat $O/longrun_default.txt $O/longrun_calm.txt
python3 bench.py --forcing calm --no-cpu-baseline > $O/bench_calm.json 2> /dev/null
python3 bench.py --no-cpu-baseline --opt pgf_uv_ring=5 > $O/bench_next.json 2> /dev/null
python3 bench.py --no-cpu-baseline > $O/bench_old.json 2> /dev/null
for f in calm next old; do python3 - $O/bench_$f.json <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l)
        print(sys.argv[1].split('/')[-1], {k: d.get(k) for k in ("ms_per_step", "ms_per_step_median", "ms_per_step_min", "ms_per_step_max")}, "spunup", (d.get("spunup") or {}).get("ms_per_step"), {k: round(v, 3) for k, v in (d.get("spunup") or {}).get("stages_ms", {}).items()})
PY
done
