"""Per-kernel averages of rocprofv3 --pmc counters.  usage: pmc_summary.py <dir> <kernel substring> [...]"""
import csv, glob, sys, collections
d, pats = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0]
        if any(p in n for p in pats):
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n in acc:
    print(n)
    for cn, v in sorted(acc[n].items()):
        print(f"   {cn:28s} {sum(v) / len(v):16.0f}   (n={len(v)})")
