"""Summarise the FETCH_SIZE / WRITE_SIZE calibration (tools/probes/fetch_calib.hip run under rocprofv3 --pmc, one
counter per pass) against the probe's known byte counts.  usage: fetch_calib_summary.py <dir with calib_f calib_w calib_t>"""
import csv, glob, collections, sys
d = sys.argv[1]
def per(path, cname):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == cname:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return acc
f = per(glob.glob(f"{d}/calib_f/*/*_counter_collection.csv")[0], "FETCH_SIZE")
w = per(glob.glob(f"{d}/calib_w/*/*_counter_collection.csv")[0], "WRITE_SIZE")
t = {r["Name"].split("(")[0]: float(r["AverageNs"]) for r in csv.DictReader(open(glob.glob(f"{d}/calib_t/*/*_kernel_stats.csv")[0]))}
B = 216 * 520 * 1060 * 8
rd = {"k_copy8": B, "k_copy16": B, "k_read8": B, "k_stencil8": B, "k_column8": B}
wr = {"k_copy8": B, "k_copy16": B, "k_read8": B / 64, "k_stencil8": B, "k_column8": B}
print("# tools/probes/fetch_calib.hip on MI355X: rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE, --kernel-trace --stats (3 passes)")
print(f"# array = 216 x 520 x 1060 doubles = {B / 1e6:.1f} MB (>> 256 MiB Infinity Cache); counters in KB, averaged over 3 launches")
print(f"# {'kernel':12s} {'known read MB':>14s} {'FETCH_SIZE MB':>14s} {'FETCH/known':>12s} {'known write MB':>15s} {'WRITE_SIZE MB':>14s} {'WRITE/known':>12s} {'avg us':>8s} {'TB/s (known bytes)':>19s}")
for k in rd:
    fv = sum(f[k]) / len(f[k]) * 1024
    wv = sum(w[k]) / len(w[k]) * 1024
    print(f"  {k:12s} {rd[k] / 1e6:14.1f} {fv / 1e6:14.1f} {fv / rd[k]:12.3f} {wr[k] / 1e6:15.1f} {wv / 1e6:14.1f} {wv / wr[k]:12.3f} {t[k] / 1e3:8.1f} {(rd[k] + wr[k]) / t[k] / 1e3:19.2f}")
print("# => unit-stride 8 B/lane loads (the fp64 point and column kernels of this library) read FETCH_SIZE = 0.500 x bytes, exactly as the")
print("#    guide states for 16 B/lane: the correction is 2.0 x FETCH_SIZE (round 1 used 1.5, which under-counted reads); WRITE_SIZE is exact.")
print("#    k_stencil8 (natural block order, no XCD-aware numbering) really fetches 2 x 1.372 = 2.74 passes for 1 compulsory pass.")
