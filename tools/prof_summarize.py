"""Turn rocprofv3 outputs under gpurun_out/ into the per-round summaries committed under profiles/.

usage: [NTR=3] [CONFIG=channel] prof_summarize.py <tag> <kernel_trace.csv> [<fetch_counter.csv> <write_counter.csv>]

Only what runs INSIDE the baroclinic steps is counted: the window of a step opens with its k_init_fluxes launch (the
first kernel of blomgpu_step's sequence) and closes before the next step's k_init_fluxes -- tmsmt2's dp halo update, its
pressure scan and k_dpudpv, and cmnfld1 of the other vertical coordinates, belong to the step --; the last step of the trace
closes with the run of such kernels after its last k_tmsmt2* launch.  The initialisation (per-field memsets of blomgpu_create, uploads, the
stages init_state runs) and the bench's own checksum kernels lie outside every window and are listed separately.
Output names carry the configuration: profiles/<tag>_<config>_kernel_stats.txt (bench.py picks the newest one of ITS
configuration)."""
import collections
import csv
import json
import os
import sys

ntr = int(os.environ.get("NTR", "3"))
config = os.environ.get("CONFIG", "channel")
dims = os.environ.get("DIMS", {"channel": "208x512x53", "tnx2v1s": "180x193x53", "tnx1v4s": "360x385x53"}.get(config, ""))
tag, trace = sys.argv[1], sys.argv[2]
base = f"profiles/{tag}_{config}"


def short(name):
    n = name.split("(")[0]
    return n[5:] if n.startswith("void ") else n


# kernels the bench itself launches between and after the steps
BENCH_OWN = ("k_crc_", "__amd_rocclr_copyBuffer")
# what tmsmt2 (and, for the other vertical coordinates, cmnfld1) still launches after the k_tmsmt2* kernels: the last step's window
# closes at the last of these
STEP_TAIL = ("k_tmsmt2", "k_xctilr", "k_pscan", "k_dpudpv", "k_cmn_")


def inside_step(names, k0, knext, mark):
    """marks the launches of the step that opens at k0: up to the launch before the next step's k_init_fluxes (knext), the
    bench's own kernels excepted; the last step of the trace (knext None) ends with the contiguous run of step-tail kernels
    that follows its last k_tmsmt2* launch"""
    if knext is None:
        knext = len(names)
        last = max((k for k in range(k0, knext) if names[k].startswith("k_tmsmt2")), default=k0)
        while last + 1 < knext and names[last + 1].startswith(STEP_TAIL):
            last += 1
        knext = last + 1
    for k in range(k0, knext):
        if not names[k].startswith(BENCH_OWN):
            mark(k)


rows = [r for r in csv.DictReader(open(trace)) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [short(r["Kernel_Name"]) for r in rows]
first = [k for k, n in enumerate(names) if n.startswith("k_init_fluxes")]
if not first:
    raise SystemExit("no k_init_fluxes launch in the trace: not a trace of blomgpu_step")
inside = [False] * len(rows)
nst = len(first)
for s, k0 in enumerate(first):
    inside_step(names, k0, first[s + 1] if s + 1 < nst else None, lambda k: inside.__setitem__(k, True))
acc = collections.defaultdict(lambda: [0, 0.0])
out_acc = collections.defaultdict(lambda: [0, 0.0])
for r, n, ins in zip(rows, names, inside):
    a = (acc if ins else out_acc)[n]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in acc.values())
tot_out = sum(v[1] for v in out_acc.values())
with open(f"{base}_kernel_stats.txt", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline  ({config} {dims}, ntr = {ntr}, 1 GPU)\n")
    f.write(f"# {nst} baroclinic steps in the trace; kernel time per step {tot / 1e6 / nst:.3f} ms (launches inside the steps only;\n")
    f.write(f"# outside them -- initialisation, uploads, the bench's checksums -- {tot_out / 1e6:.3f} ms in total, not counted)\n")
    f.write(f"# {'kernel':42s} {'calls/step':>10s} {'avg_us':>10s} {'ms/step':>9s} {'%':>6s}\n")
    for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        if t / tot < 0.0005:
            continue
        f.write(f"{n[:44]:44s} {c / nst:10.1f} {t / c / 1e3:10.1f} {t / 1e6 / nst:9.3f} {100 * t / tot:6.1f}\n")
    f.write("# outside the steps (total ms, calls):\n")
    for n, (c, t) in sorted(out_acc.items(), key=lambda kv: -kv[1][1])[:8]:
        f.write(f"#   {n[:44]:44s} {t / 1e6:9.3f} {c:6d}\n")

if len(sys.argv) > 4:
    # the counter passes run the same command (other --steps): a step's launches are found the same way, by dispatch order
    def per_kernel(path, cname):
        recs = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == cname]
        recs.sort(key=lambda r: int(r["Dispatch_Id"]))
        nm = [short(r["Kernel_Name"]) for r in recs]
        fi = [k for k, n in enumerate(nm) if n.startswith("k_init_fluxes")]
        a = collections.defaultdict(lambda: [0, 0.0])
        def add(k):
            a[nm[k]][0] += 1
            a[nm[k]][1] += float(recs[k]["Counter_Value"])
        for s, k0 in enumerate(fi):
            inside_step(nm, k0, fi[s + 1] if s + 1 < len(fi) else None, add)
        return a, len(fi)
    (fe, nst_f), (wr, nst_w) = per_kernel(sys.argv[3], "FETCH_SIZE"), per_kernel(sys.argv[4], "WRITE_SIZE")
    with open(f"{base}_pmc_hbm_traffic.txt", "w") as f:
        f.write(f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 3 --warmup 2 ({config} {dims}, ntr = {ntr});\n")
        f.write(f"# launches inside the {nst_f} baroclinic steps only.\n")
        f.write("# per-launch averages in MB (counters are in KB).  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports\n")
        f.write("# half the bytes of a coalesced stream, for the 8 B/lane loads of these fp64 kernels as for 16 B/lane ones (calibrated on known\n")
        f.write("# byte counts: tools/probes/fetch_calib.hip, profiles/r02_fetch_calibration.txt): fetch_x2 is the corrected figure.\n")
        f.write(f"# {'kernel':30s} {'launches':>8s} {'fetch_MB':>10s} {'fetch_x2_MB':>12s} {'write_MB':>10s}\n")
        ks = sorted(fe, key=lambda k: -(fe[k][1] + wr.get(k, [0, 0])[1]))
        for k in ks[:44]:
            n, v = fe[k]
            w = wr.get(k, [1, 0.0])
            f.write(f"{k[:32]:32s} {n:8d} {v / n / 1024:10.2f} {2 * v / n / 1024:12.2f} {w[1] / max(1, w[0]) / 1024:10.2f}\n")
    # per bench class (bench.py `stages_ms` keys): HBM bytes per baroclinic step from the two PMC passes,
    # fetch corrected by the 2.0x calibrated for the 8 B/lane loads of these fp64 kernels
    # (profiles/r02_fetch_calibration.txt), writes as counted.  bench.py reports it as roofline.traffic.
    prefixes = [("k_cmn_", "cmnfld"), ("k_mom_", "momtum"), ("k_remap_", "remap"), ("k_adv_", "remap"), ("k_cppm_", "cppm"),
                ("k_diffus_", "diffus"), ("k_pgf_", "pgforc"), ("k_diapfl_", "diapfl"), ("k_convec_", "convec"),
                ("k_bt_", "barotp"), ("k_pbc_", "pbcor"), ("k_eddtra_", "eddtra"), ("k_mxl_", "mxlayr"), ("k_difest_", "difest"),
                ("k_thermf_", "thermf"), ("k_xcsum_", "thermf"), ("k_niw_", "difest"), ("k_dfi_", "difest")]
    cls = collections.defaultdict(float)
    allk = collections.defaultdict(float)
    for k in set(fe) | set(wr):
        b = 2.0 * fe.get(k, [0, 0.0])[1] * 1024.0 / max(1, nst_f) + wr.get(k, [0, 0.0])[1] * 1024.0 / max(1, nst_w)
        allk["all"] += b
        for pre, c in prefixes:
            if k.startswith(pre):
                cls[c] += b
                break
        else:
            cls["other"] += b
    if "pbcor" in cls:
        cls["pbcor1"] = cls["pbcor2"] = cls.pop("pbcor") / 2
    with open(f"{base}_class_traffic.json", "w") as f:
        json.dump({"source": f"{base}_pmc_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; launches inside the steps only)",
                   "correction": "2.0 x FETCH_SIZE (calibrated: profiles/r02_fetch_calibration.txt) + WRITE_SIZE, bytes per baroclinic step and class",
                   "ntr": ntr, "config": config,
                   "bytes_per_step_total": round(allk["all"]),
                   "bytes_per_step": {k: round(v) for k, v in sorted(cls.items())}}, f, indent=1)
print(open(f"{base}_kernel_stats.txt").read()[:4500])
