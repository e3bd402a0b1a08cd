"""TEST INFRASTRUCTURE (build container only: needs oracle/_ref/channel_tke_omp_xdf, i.e. /root/reference compiled by oracle/Makefile).

The reference's own stage routines (phy/mod_blom_step.F90:96-253 in the order of stepper.FULL_STAGES_LIVE) run for N steps from
the initial state bench.py times (bench.ref_full_init: channel 208x512x53, ntr = 3, NorESM's &DIFFUSION defaults, the channel
experiment's forcing), and what the long run of the device is compared with is written to tests/golden/channel_tke_live_long_crc.json:

  * every `--every` steps: xccrc (phy/mod_xc.F90:4164-4205) of dp, temp, saln, u, v, trc over both time levels, of difint / difdia,
    and the range of temp in the current time level;
  * every step: min / max of temp over the wet points of the current time level with their (i, j, k) [1-based, the reference's
    indices], the dp there, the dp-weighted sums of mass / heat / salt;
  * `--budget-step S`: after every stage of step S the dp-weighted heat sum of both time levels (which stage moves it: the
    reference's budget terms, phy/mod_budget.F90:95-196, are these sums taken after advect+diffus, diapfl, thermf ...).

usage: python3 tools/longrun_reference.py [--steps 600] [--every 100] [--forcing default|calm] [--out tests/golden/...json]"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

CRC_FIELDS = ["dp", "temp", "saln", "u", "v", "trc", "difint", "difdia"]


def extremes(t, dp, wet3):
    """min / max of t over wet points with mass: (value, i, j, k) 1-based, and dp there"""
    tm = np.where(wet3, t, np.inf)
    k, j, i = np.unravel_index(int(np.argmin(tm)), tm.shape)
    lo = (float(t[k, j, i]), int(i) + 1, int(j) + 1, int(k) + 1, float(dp[k, j, i]))
    tm = np.where(wet3, t, -np.inf)
    k, j, i = np.unravel_index(int(np.argmax(tm)), tm.shape)
    hi = (float(t[k, j, i]), int(i) + 1, int(j) + 1, int(k) + 1, float(dp[k, j, i]))
    return lo, hi


def sample(be, case, masks, ns, scp2):
    from blom_amd import hostinit
    kk = case.kdm
    nn = hostinit.step_indices(ns, kk)[3] if ns else hostinit.step_indices(0, kk)[3]
    sl = (slice(nn, nn + kk), slice(4, -4), slice(4, -4))
    t, s, dp = be.get("temp")[sl], be.get("saln")[sl], be.get("dp")[sl]
    wet = masks["ip"][4:-4, 4:-4] > 0
    wet3 = np.broadcast_to(wet[None], t.shape)
    lo, hi = extremes(t, dp, wet3)
    w = dp * scp2[None] * wet[None]
    return dict(step=ns, tmin=lo, tmax=hi, mass=float(w.sum()), heat=float((t * w).sum()), salt=float((s * w).sum()))


def run(args, res):
    import bench
    from blom_amd import hostinit
    from blom_amd.checksum import chksum
    from blom_amd.stepper import dyncore_step, FULL_STAGES_LIVE
    from oracle.refblom import get_ref_backend
    case, nreg, masks = bench.build_case("channel", "remap", "default", forcing=args.forcing)
    os.environ["OMP_NUM_THREADS"] = str(bench.usable_cores())
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ["OMP_STACKSIZE"] = "1G"
    be = get_ref_backend("channel_tke_omp_xdf", case.depth, ntr=case.ntr)
    be.ref.set("eitmth", "gm")
    be.has_stage = lambda name: True
    bench.ref_full_init(be, case, True, True)
    if not args.rhsctp:                  # (round 5's long run, whose extreme samples the review asked about, had rhsctp off)
        be.ref.set("rhsctp", 0)
    scp2 = np.array(be.get("scp2")[0][4:-4, 4:-4])
    kk = case.kdm
    out = dict(workload=f"channel {case.idm}x{case.jdm}x{kk}, ntr = {case.ntr}, stepper.FULL_STAGES_LIVE, hostinit.DIFEST_NORESM, forcing {args.forcing}",
               reference_build="oracle/_ref/channel_tke_omp_xdf (the reference's own modules; mod_difest behind the interface-only CVMix stand-in)",
               generator="tools/longrun_reference.py", rhsctp=int(args.rhsctp), crc_fields=CRC_FIELDS, every=args.every, crc={}, trace=[], budget={})
    out["trace"].append(sample(be, case, masks, 0, scp2))
    ns = 0
    t0 = time.time()
    while ns < args.steps:
        hook = None
        if ns + 1 == args.budget_step:
            rows = []

            def hook(st, six, rows=rows):
                wet = masks["ip"][4:-4, 4:-4] > 0
                r = {"before": st}
                for nm, off in (("m", six[2]), ("n", six[3])):
                    sl = (slice(off, off + kk), slice(4, -4), slice(4, -4))
                    w = be.get("dp")[sl] * scp2[None] * wet[None]
                    r["heat_" + nm] = float((be.get("temp")[sl] * w).sum())
                    r["mass_" + nm] = float(w.sum())
                rows.append(r)
            out["budget"] = dict(step=ns + 1, rows=rows)
        ns = dyncore_step(be, ns, case.params["baclin"], stages=FULL_STAGES_LIVE, hook=hook)
        if hook is not None:
            hook("end", hostinit.step_indices(ns - 1, kk))
        out["trace"].append(sample(be, case, masks, ns, scp2))
        if ns % args.every == 0:
            out["crc"][str(ns)] = {nm: chksum(nm, be.get(nm), masks, case.idm, case.jdm) for nm in CRC_FIELDS}
            tr = out["trace"][-1]
            print(f"step {ns}: {time.time() - t0:.0f} s  Tmin {tr['tmin']}  Tmax {tr['tmax']}  heat {tr['heat']:.9e}", flush=True)
            with open(args.out, "w") as f:
                json.dump(out, f)
    res["ok"] = True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--every", type=int, default=100)
    ap.add_argument("--budget-step", type=int, default=300)
    ap.add_argument("--forcing", default="default")
    ap.add_argument("--rhsctp", type=int, default=1, help="0: hostinit.DIFEST_NORESM with rhsctp = .false. (the bench's options of rounds 5 and before; "
                    "tests/golden/channel_tke_live_long_rhsctp0_crc.json)")
    ap.add_argument("--out", default="tests/golden/channel_tke_live_long_crc.json")
    args = ap.parse_args()
    res = {}
    threading.stack_size(2 << 30)
    th = threading.Thread(target=run, args=(args, res))
    th.start()
    th.join()
    raise SystemExit(0 if res.get("ok") else 1)


if __name__ == "__main__":
    main()
