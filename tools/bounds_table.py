"""profiles/r05_bounds.md: for the longest kernels of the step, the time each of three bounds allows beside the time measured.

  bytes    counted HBM-side traffic of a launch (2 x FETCH_SIZE + WRITE_SIZE: profiles/*_pmc_hbm_traffic.txt) at the rate a plain copy
           reaches on the box the profile was taken on (tools/probes/copy_rate in the same job; the pool's boxes differ by ~15 %:
           5.2 - 6.2 TB/s, profiles/r02_fetch_calibration.txt has 5.7 - 6.2); `alg` the same for the kernel's algorithmic bytes at
           the 8 TB/s of the data sheet, where bench.py defines them
  issue    instructions a SIMD must issue: (SQ_INSTS_VALU + SQ_INSTS_SALU + SQ_INSTS_LDS + SQ_INSTS_VMEM) per wave x waves x 4 cycles
           / 1024 SIMDs / 2.4 GHz -- every instruction of a wave64 occupies its SIMD's issue port for 4 cycles (DESIGN.md 3b;
           fp64 division and sqrt sequences are longer, so this is a lower bound)
  wait     share of the wave cycles in which the wave was parked on a counter (SQ_WAIT_ANY / SQ_WAVE_CYCLES) -- for the column kernels
           (one thread per column, 1.6 - 3.4 waves per SIMD) the k-serial chain of dependent loads
  chain    mean lifetime of a wave (4 x SQ_WAVE_CYCLES / SQ_WAVES at 2.4 GHz; the counter ticks every 4 cycles): a column kernel's
           waves are all resident from the start (1.7 per SIMD), so the launch lasts as long as its LONGEST wave -- the column
           with the deepest mixed layer / the most iterations -- and the mean says how much of the launch is that tail

usage: bounds_table.py <kernel_stats.txt> <pmc_hbm_traffic.txt> <sq_counters.txt> [copy rate of the box, TB/s] [text to append] > profiles/r05_bounds.md"""
import re
import sys

ks, tr, sq = sys.argv[1:4]
RATE = float(sys.argv[4]) if len(sys.argv) > 4 else 6.0
import os
ROUND = os.environ.get("ROUND", "6")
F = 208 * 512 * 53 * 8.0
ALG = {"k_remap_tile": 20, "k_mom_cor_march": 24, "k_mom_visc_march": 8, "k_diapfl_column3": 20, "k_pgf_uv": 15, "k_pgf_uv_next": 15, "k_pbc_tile": 24,
       "k_diffus_tile": 21, "k_tmsmt2": 24, "k_remap_update": 18}


def key(n):
    return n.replace("void ", "").split("<")[0].split("(")[0].strip()


stats = []
for ln in open(ks):
    if ln.startswith("#") or not ln.strip():
        continue
    p = ln.split()
    try:
        stats.append((" ".join(p[:-4]), float(p[-4]), float(p[-3]), float(p[-2])))
    except ValueError:
        pass
traffic = {}
for ln in open(tr):
    if ln.startswith("#") or not ln.strip():
        continue
    p = ln.split()
    try:
        traffic[key(" ".join(p[:-4]))] = (float(p[-2]) + float(p[-1])) * 1e6
    except ValueError:
        pass
ctr, cur = {}, None
for ln in open(sq):
    if ln.strip() and not ln.startswith(" "):
        cur = key(ln.strip())
        ctr.setdefault(cur, {})
    else:
        m = re.match(r"\s+(\S+)\s+(\d+)", ln)
        if m and cur:
            ctr[cur][m.group(1)] = float(m.group(2))
print(f"# Bounds of the longest kernels (round {ROUND})\n")
print("Channel 208x512x53, ntr = 3, one MI355X; `python3 bench.py` (config 2's step with live diffusivities).  Columns: measured average launch")
print("(rocprofv3 kernel trace, overlap off), launches per step, and what each bound allows -- see tools/bounds_table.py for the definitions.")
print("A kernel sits at its floor when `measured` is close to the largest of the bounds; `wait` says how much of the rest is a k-serial chain.\n")
print(f"| kernel | per step | measured us | counted MB | bytes bound us ({RATE:.2f} TB/s) | alg. bytes us (8 TB/s) | instr / wave (VALU+SALU+LDS+VMEM) | waves | issue bound us | wait % | mean wave lifetime us | nearest bound / measured |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
tot = 0.0
floor = 0.0
floor_col = 0.0
meas_col = 0.0
for name, per, avg, ms in stats[:24]:
    k = key(name)
    t = traffic.get(k)
    c = ctr.get(k, {})
    w = c.get("SQ_WAVES")
    bb = t / (RATE * 1e12) * 1e6 if t else None
    alg = ALG.get(k)
    ab = alg * F / 8.0e12 * 1e6 if alg else None
    ib = wt = ipw = life = None
    if w:
        n = sum(c.get(x, 0.0) for x in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
        ipw = n / w
        ib = n * 4.0 / 1024.0 / 2.4e9 * 1e6
        wc = c.get("SQ_WAVE_CYCLES")
        wt = 100.0 * c.get("SQ_WAIT_ANY", 0.0) / wc if wc else None
        life = 4.0 * wc / w / 2.4e3 if wc else None
    best = max([x for x in (bb, ib) if x] or [0.0])
    f = lambda x, d=0: "-" if x is None else f"{x:.{d}f}"
    print(f"| `{name}` | {per:.1f} | {avg:.1f} | {f(t / 1e6 if t else None)} | {f(bb)} | {f(ab)} | {f(ipw)} | {f(w)} | {f(ib)} | {f(wt)} | {f(life, 1)} | {best / avg:.2f} |")
    tot += ms
    floor += per * best * 1e-3
    if w and w < 4000:
        meas_col += ms
        floor_col += per * best * 1e-3
print(f"\nThe {min(24, len(stats))} kernels above are {tot:.2f} ms of the step; the larger of each kernel's two bounds sums to {floor:.2f} ms.")
print(f"The kernels with fewer than 4 000 waves -- one thread per column or per strip row, at most 3.4 waves per SIMD -- are {meas_col:.2f} ms of that against bounds of {floor_col:.2f} ms;")
print(f"the tiled and per-level kernels (>= 90 000 waves) are {tot - meas_col:.2f} ms against {floor - floor_col:.2f} ms.")
if len(sys.argv) > 5:
    print()
    print(open(sys.argv[5]).read())
