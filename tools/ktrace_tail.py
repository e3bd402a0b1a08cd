"""Per-kernel average duration over the LAST part of a rocprofv3 --kernel-trace run (a long run whose state changes: bench.py --spinup N).
usage: ktrace_tail.py <dir with */*_kernel_trace.csv> [fraction of each kernel's calls, from the end: 0.02] [substring]"""
import csv, glob, sys
d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
pat = sys.argv[3] if len(sys.argv) > 3 else "k_"
trace = glob.glob(f"{d}/*/*_kernel_trace.csv")[0]
calls = {}
for r in csv.DictReader(open(trace)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if pat in n:
        calls.setdefault(n, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows = []
for n, c in calls.items():
    c.sort()
    k = max(1, int(len(c) * frac))
    head, tail = c[:k], c[-k:]
    rows.append((sum(x[1] for x in tail) / k / 1e3, sum(x[1] for x in head) / k / 1e3, len(c), n))
rows.sort(reverse=True)
print(f"# {'kernel':46s} calls   first {frac:.0%} avg_us   last {frac:.0%} avg_us")
for t, h, ncall, n in rows[:60]:
    print(f"{n[:48]:48s} {ncall:6d} {h:12.1f} {t:12.1f}")
