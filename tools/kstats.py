"""Print per-kernel averages (and register/LDS use) from a rocprofv3 --kernel-trace --stats output directory.
usage: kstats.py <dir> [substring ...]"""
import csv, glob, sys
d, pats = sys.argv[1], sys.argv[2:]
stats = glob.glob(f"{d}/*/*_kernel_stats.csv")[0]
trace = glob.glob(f"{d}/*/*_kernel_trace.csv")[0]
res = {}
for r in csv.DictReader(open(trace)):
    res.setdefault(r["Kernel_Name"], r)
tot = 0.0
for r in csv.DictReader(open(stats)):
    n = r["Name"]
    if pats and not any(p in n for p in pats):
        continue
    t = res.get(n, {})
    tot += float(r["TotalDurationNs"])
    print(f"{n.split('(')[0][:46]:46s} calls {int(r['Calls']):5d} avg_us {float(r['AverageNs']) / 1e3:9.1f}  vgpr {t.get('VGPR_Count', '?'):>4s} agpr {t.get('Accum_VGPR_Count', '?'):>3s} "
          f"sgpr {t.get('SGPR_Count', '?'):>4s} lds {t.get('LDS_Block_Size', '?'):>7s} scratch {t.get('Scratch_Size', '?'):>5s} wg {t.get('Workgroup_Size', '?'):>5s} grid {t.get('Grid_Size', '?')}")
print(f"total of listed: {tot / 1e6:.3f} ms")
