"""What the bench workload does over a long run, on the device (GPU box), and whether the reference does the same: config 2's step on
the channel 208x512x53, ntr = 3, the channel experiment's forcing, diffusivities estimated every step (hostinit.DIFEST_NORESM).

Every `--every` steps from step `--from` on: range of T with the cell (i, j, k; 1-based) it is taken in and the dp there, the mixed
layer depth (the two bulk layers) min / mean / max, the columns whose mixed layer iterations ended at their limit per step (named as the
reference's messages name them, phy/mod_mxlayr.F90:440, :950), max |u|, the drift of total mass, heat and salt relative to the start.
With `--golden FILE` (tests/golden/channel_tke_live_long_crc.json, written by tools/longrun_reference.py from the reference's own
modules) every sampled step's extremes, their cells and the three sums are compared with the reference's, and the xccrc of the state at
the steps the file has checksums for.

usage: python3 tools/longrun_full_physics.py [--steps 1200] [--every 200] [--from 0] [--frozen-diffusivities] [--forcing calm]
                                              [--golden tests/golden/channel_tke_live_long_crc.json] [--budget-step S]"""
import argparse
import json
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench
from blom_amd.gpu import BlomGpu
from blom_amd import hostinit
from longrun_reference import sample, CRC_FIELDS

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=1200)
ap.add_argument("--every", type=int, default=200)
ap.add_argument("--from", dest="start", type=int, default=0)
ap.add_argument("--frozen-diffusivities", action="store_true")
ap.add_argument("--forcing", default="default")
ap.add_argument("--golden", default=None)
ap.add_argument("--rhsctp", type=int, default=1, help="0: rhsctp off, the options of rounds 5 and before (golden: ..._rhsctp0_crc.json)")
ap.add_argument("--budget-step", type=int, default=0, help="after every stage of this step: the dp-weighted heat sums of both time levels")
args = ap.parse_args()
frozen = args.frozen_diffusivities
case, nreg, masks = bench.build_case("channel", "remap", "default", forcing=args.forcing)
gpu = bench.device_for_bench(case, nreg, masks, live=not frozen)
if not args.rhsctp:
    gpu.set("rhsctp", 0)
gold = json.load(open(args.golden)) if args.golden else None
gtrace = {t["step"]: t for t in gold["trace"]} if gold else {}
wet = masks["ip"][4:-4, 4:-4] > 0
scp2 = gpu.get("scp2")[0][4:-4, 4:-4]
kk = case.kdm
print(f"# channel {case.idm}x{case.jdm}x{kk}, ntr = {case.ntr}, full physics, diffusivities {'frozen' if frozen else 'live (NorESM defaults' + ('' if args.rhsctp else ', rhsctp off') + ')'}, forcing {args.forcing}; baclin = {case.params['baclin']} s")
print("# step  Tmin (i j k dp)  Tmax (i j k dp)  mld_min  mld_mean  mld_max [m]  maxitr_detrain/step  maxitr_entrain/step  |u|max  d(mass)/mass  d(heat)/heat  d(salt)/salt  difdia_max  difint_mean  finite  vs reference")
ns, base, last, bad = 0, None, 0, 0
gpu.get_real("mxlayr_maxitr_entrain"); gpu.get_real("mxlayr_maxitr_detrain")
while True:
    if ns >= args.start and (ns - args.start) % args.every == 0 or ns == 0:
        smp = sample(gpu, case, masks, ns, scp2)
        nn = hostinit.step_indices(ns, kk)[3] if ns else hostinit.step_indices(0, kk)[3]
        dp = gpu.get("dp")[nn:nn + kk][:, 4:-4, 4:-4]
        u = gpu.get("u")[nn:nn + kk][:, 4:-4, 4:-4]
        mld = (dp[0] + dp[1])[wet] / 9806.0
        if base is None:
            base = (smp["mass"], smp["heat"], smp["salt"])
        nsteps = max(1, ns - last)
        nd = gpu.get_real("mxlayr_maxitr_detrain") / nsteps
        ne = gpu.get_real("mxlayr_maxitr_entrain") / nsteps
        last = ns
        dd = gpu.get("difdia")[:, 4:-4, 4:-4]; di = gpu.get("difint")[:, 4:-4, 4:-4]
        w2 = np.broadcast_to(wet[None], dd.shape)
        uu = u[np.isfinite(u) & (np.abs(u) < 1e10)]
        verdict = ""
        if ns in gtrace:
            g = gtrace[ns]
            same = all(list(smp[k_]) == list(g[k_]) for k_ in ("tmin", "tmax")) and all(smp[k_] == g[k_] for k_ in ("mass", "heat", "salt"))
            verdict = "== reference (extremes, their cells, the three sums)" if same else f"DIFFERS from the reference: {g}"
            bad += not same
        if gold and str(ns) in gold["crc"]:
            from blom_amd.checksum import grid_of
            diff = [nm for nm in CRC_FIELDS if gpu.crc(nm, 1, gpu.field_info(nm)[0], grid_of(nm)) != gold["crc"][str(ns)][nm]]
            verdict += "; xccrc of " + ", ".join(CRC_FIELDS) + (" == reference" if not diff else f" DIFFERS in {diff}")
            bad += bool(diff)
        lo, hi = smp["tmin"], smp["tmax"]
        print(ns, f"{lo[0]:.4f} ({lo[1]} {lo[2]} {lo[3]} {lo[4]:.3e})  {hi[0]:.4f} ({hi[1]} {hi[2]} {hi[3]} {hi[4]:.3e})  {mld.min():.3f} {mld.mean():.3f} {mld.max():.3f}  {nd:.2f} {ne:.2f}  {np.abs(uu).max():.4f}  "
              f"{(smp['mass'] - base[0]) / base[0]:.3e} {(smp['heat'] - base[1]) / base[1]:.3e} {(smp['salt'] - base[2]) / base[2]:.3e}  {dd[w2].max():.4e} {di[w2].mean():.2f}  "
              f"{bool(np.isfinite(dp[np.broadcast_to(wet[None], dp.shape)]).all())}  {verdict}", flush=True)
    if ns >= args.steps:
        break
    nxt = args.steps
    if ns < args.start:
        nxt = min(nxt, args.start)
    else:
        nxt = min(nxt, ns + args.every)
    if args.budget_step and ns < args.budget_step <= nxt:
        if args.budget_step - 1 > ns:
            ns = gpu.step(ns, args.budget_step - 1 - ns)
        from blom_amd.stepper import dyncore_step, FULL_STAGES_LIVE, FULL_STAGES
        rows = []

        def hook(st, six):
            r = [st]
            for off in (six[2], six[3]):
                sl = (slice(off, off + kk), slice(4, -4), slice(4, -4))
                w = gpu.get("dp")[sl] * scp2[None] * wet[None]
                r.append(float((gpu.get("temp")[sl] * w).sum()))
            rows.append(r)
        ns = dyncore_step(gpu, ns, case.params["baclin"], stages=FULL_STAGES if frozen else FULL_STAGES_LIVE, hook=hook)
        hook("end", hostinit.step_indices(ns - 1, kk))
        print(f"# heat sums (dp-weighted, time levels m and n) BEFORE each stage of step {ns}:")
        for a, b in zip(rows[:-1], rows[1:]):
            print(f"#   {a[0]:22s} d(heat_m) {(b[1] - a[1]) / a[1]: .3e}  d(heat_n) {(b[2] - a[2]) / a[2]: .3e}")
        if gold and gold.get("budget", {}).get("step") == ns:
            gr = gold["budget"]["rows"]
            same = len(gr) == len(rows) and all(g["heat_m"] == r[1] and g["heat_n"] == r[2] for g, r in zip(gr, rows))
            print("#   the same sums in the reference's run:", "== bit for bit" if same else "DIFFER")
            bad += not same
        continue
    ns = gpu.step(ns, nxt - ns)
gpu.close()
if gold:
    print("# every comparison with the reference's run agreed" if not bad else f"# {bad} comparisons with the reference's run FAILED")
    raise SystemExit(1 if bad else 0)
