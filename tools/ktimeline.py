"""Timeline of the kernels of ONE baroclinic step from a rocprofv3 --kernel-trace output directory: start (us, relative to the
step's first kernel), duration, queue, name -- shows which kernels of the two streams actually ran side by side.
usage: ktimeline.py <dir> [step_index_from_end=2]"""
import csv, glob, sys
d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
trace = glob.glob(f"{d}/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
# a step starts with k_init_fluxes
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_init_fluxes")]
a, b = starts[-back - 1], starts[-back]
t0 = int(rows[a]["Start_Timestamp"])
qs = {}
busy_end = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = qs.setdefault(r["Queue_Id"], len(qs))
    ov = "  ||" if s < busy_end else ""
    busy_end = max(busy_end, e)
    print(f"{s / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q} {r['Kernel_Name'].split('(')[0][:60]}{ov}")
print(f"step: {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
