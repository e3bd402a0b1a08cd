"""Turn rocprofv3 outputs under gpurun_out/ into the per-round summaries committed under profiles/.
usage: [NTR=3] prof_summarize.py <tag> <kernel_stats.csv> <nsteps_total | 0: count them> [<fetch_counter.csv> <write_counter.csv>]"""
import csv, os, sys, collections
ntr = int(os.environ.get("NTR", "3"))
tag, stats, nst = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = list(csv.DictReader(open(stats)))
if nst <= 0:                                   # one k_init_fluxes launch per baroclinic step
    nst = next(int(r["Calls"]) for r in rows if r["Name"].startswith("k_init_fluxes"))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(f"profiles/{tag}_kernel_stats.txt", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline  (channel 208x512x53, ntr = {ntr}, 1 GPU)\n")
    f.write(f"# {nst} baroclinic steps in the trace; kernel time per step {tot / 1e6 / nst:.3f} ms\n")
    f.write(f"# {'kernel':42s} {'calls/step':>10s} {'avg_us':>10s} {'ms/step':>9s} {'%':>6s}\n")
    for r in rows:
        n, t = int(r["Calls"]), float(r["TotalDurationNs"])
        if t / tot < 0.0005:
            continue
        f.write(f"{r['Name'].split('(')[0][:44]:44s} {n / nst:10.1f} {t / n / 1e3:10.1f} {t / 1e6 / nst:9.3f} {100 * t / tot:6.1f}\n")
if len(sys.argv) > 5:
    def per_kernel(path, cname):
        acc = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != cname:
                continue
            a = acc[r["Kernel_Name"].split("(")[0]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        return acc
    fe, wr = per_kernel(sys.argv[4], "FETCH_SIZE"), per_kernel(sys.argv[5], "WRITE_SIZE")
    with open(f"profiles/{tag}_pmc_hbm_traffic.txt", "w") as f:
        f.write(f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 3 --warmup 2 (channel 208x512x53, ntr = {ntr})\n")
        f.write("# per-launch averages in MB (counters are in KB).  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports\n")
        f.write("# half the bytes of a coalesced stream, for the 8 B/lane loads of these fp64 kernels as for 16 B/lane ones (calibrated on known\n")
        f.write("# byte counts: tools/probes/fetch_calib.hip, profiles/r02_fetch_calibration.txt): fetch_x2 is the corrected figure.\n")
        f.write(f"# {'kernel':30s} {'launches':>8s} {'fetch_MB':>10s} {'fetch_x2_MB':>12s} {'write_MB':>10s}\n")
        ks = sorted(fe, key=lambda k: -(fe[k][1] + wr.get(k, [0, 0])[1]))
        for k in ks[:40]:
            n, v = fe[k]
            w = wr.get(k, [1, 0.0])
            f.write(f"{k[:32]:32s} {n:8d} {v / n / 1024:10.2f} {2 * v / n / 1024:12.2f} {w[1] / max(1, w[0]) / 1024:10.2f}\n")
    # per bench class (bench.py `stages_ms` keys): HBM bytes per baroclinic step from the two PMC passes,
    # fetch corrected by the 2.0x calibrated for the 8 B/lane loads of these fp64 kernels
    # (profiles/r02_fetch_calibration.txt), writes as counted.  bench.py reports it as roofline.traffic.
    import json
    prefixes = [("k_cmn_", "cmnfld"), ("k_mom_", "momtum"), ("void k_mom_", "momtum"), ("k_remap_", "remap"), ("k_adv_", "remap"), ("k_cppm_", "cppm"),
                ("k_diffus_", "diffus"), ("k_pgf_", "pgforc"), ("k_diapfl_", "diapfl"), ("k_convec_", "convec"),
                ("k_bt_", "barotp"), ("void k_bt_", "barotp"), ("k_pbc_", "pbcor"), ("k_eddtra_", "eddtra")]
    once = [k for k in fe if k.startswith("k_mom_column") or k.startswith("k_mom_update")]
    nsteps_pmc = max(fe[k][0] for k in once) if once else 1
    cls = collections.defaultdict(float)
    for k in fe:
        kk_ = k[5:] if k.startswith("void ") else k          # templated kernels are reported as "void name<..>"
        for pre, c in prefixes:
            if kk_.startswith(pre):
                cls[c] += (2.0 * fe[k][1] + wr.get(k, [0, 0.0])[1]) * 1024.0 / nsteps_pmc
                break
    if "pbcor" in cls:
        cls["pbcor1"] = cls["pbcor2"] = cls.pop("pbcor") / 2
    with open(f"profiles/{tag}_class_traffic.json", "w") as f:
        json.dump({"source": f"profiles/{tag}_pmc_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
                   "correction": "2.0 x FETCH_SIZE (calibrated: profiles/r02_fetch_calibration.txt) + WRITE_SIZE, bytes per baroclinic step and class",
                   "ntr": ntr,
                   "bytes_per_step": {k: round(v) for k, v in sorted(cls.items())}}, f, indent=1)
print(open(f"profiles/{tag}_kernel_stats.txt").read()[:3500])
