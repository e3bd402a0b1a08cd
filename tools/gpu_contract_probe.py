#!/usr/bin/env python3
"""What does bit parity cost?  (GPU box; MEASUREMENT ONLY.)

The reference's release flags forbid contraction (meson.build:18-20: -ffp-contract=off) and the device library is built the
same way, which is what makes `==` against the reference possible.  This probe loads blom_amd/lib/libblomgpu_contract.so --
the same sources compiled with the compiler's default -ffp-contract=fast (make -C blom_amd/csrc contract) -- and reports
  * for every stage of the sequence, from IDENTICAL inputs (the reference's state before the stage), the largest relative
    difference to the reference in any field it writes (relative to the field's largest magnitude; integer fields must
    stay equal), over 4 steps on chan_s_tke, box_s_tke, tri_s_tke;
  * the channel's ms/step with either library (bench.py in a child process each).
usage: python tools/gpu_contract_probe.py [--no-bench]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = os.path.join(ROOT, "blom_amd", "lib", "libblomgpu_contract.so")


def stage_differences():
    os.environ["BLOMGPU_LIB"] = CONTRACT
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from blom_amd.cases import make_case
    from blom_amd import hostinit
    from blom_amd.gpu import BlomGpu
    from blom_amd.stepper import dyncore_step, STAGES_FROZEN_EDDY_FLUXES
    from oracle.refblom import get_ref_backend
    from parity import copy_state, STATE_FIELDS, INT_FIELDS
    scratch = {"uflux", "vflux", "uflux2", "vflux2", "uflux3", "vflux3", "utotm", "vtotm", "util1", "util2", "util3", "util4"}
    worst = {}
    for cfg in ("chan_s_tke", "box_s_tke", "tri_s_tke"):
        case = make_case(cfg)
        ref = get_ref_backend(cfg, case.depth)
        hostinit.init_state(ref, case)
        hostinit.frozen_eddy_fluxes(ref, case)
        gpu = BlomGpu(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
        for nm, v in case.params.items():
            if not nm.endswith("0"):
                gpu.set(nm, v)
        pending, nstep = {}, [0]

        def check():
            if "st" not in pending:
                return
            st = pending.pop("st")
            for nm in STATE_FIELDS + INT_FIELDS:
                if nm in scratch:
                    continue
                try:
                    a, b = np.asarray(ref.get(nm)), np.asarray(gpu.get(nm))
                except KeyError:
                    continue
                n = min(a.shape[0], b.shape[0])
                a, b = a[:n], b[:n]
                ne = ~((a == b) | (np.isnan(a) & np.isnan(b)))
                if not ne.any():
                    continue
                if nm in INT_FIELDS:
                    worst[(st, nm)] = float("inf")
                    continue
                fin = np.isfinite(a) & (np.abs(a) < 1e30)
                scale = float(np.abs(a[fin]).max()) if fin.any() else 1.0
                with np.errstate(invalid="ignore", over="ignore"):
                    d = float(np.nanmax(np.where(ne & fin, np.abs(a - b), 0.0))) / max(scale, 1e-300)
                worst[(st, nm)] = max(worst.get((st, nm), 0.0), d)

        def hook(st, six):
            check()
            copy_state(ref, gpu)
            gpu.set("nstep", nstep[0] + 1)
            gpu.set("delt1", ref.ref.get_real("delt1"))
            gpu.stage(st, *six)
            pending["st"] = st

        for _ in range(4):
            new = dyncore_step(ref, nstep[0], case.params["baclin"], hook=hook, stages=STAGES_FROZEN_EDDY_FLUXES)
            check()
            nstep[0] = new
        gpu.close()
    by_stage = {}
    for (st, nm), d in worst.items():
        if d > by_stage.get(st, (0.0, ""))[0]:
            by_stage[st] = (d, nm)
    print("stage: largest |device(contracted) - reference| / max|field| after ONE stage from identical inputs")
    for st, (d, nm) in by_stage.items():
        print(f"  {st:14s} {d:10.3e}  ({nm})")
    print("  stages not listed: bit-identical to the reference also with contraction")
    return by_stage


def bench(lib):
    env = dict(os.environ)
    if lib:
        env["BLOMGPU_LIB"] = lib
    else:
        env.pop("BLOMGPU_LIB", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"], env=env, capture_output=True, text=True).stdout
    line = [x for x in out.splitlines() if x.startswith('{"metric')][-1]
    d = json.loads(line)
    return d["ms_per_step"], d["stages_ms"]


if __name__ == "__main__":
    if "--no-bench" not in sys.argv:
        a, sa = bench(None)
        b, sb = bench(CONTRACT)
        a2, _ = bench(None)
        print(f"channel ms/step: -ffp-contract=off {a:.3f} / {a2:.3f}   -ffp-contract=fast {b:.3f}   gain {100 * (min(a, a2) - b) / min(a, a2):.1f} %")
        for st in sa:
            print(f"  {st:8s} {sa[st]:.3f} -> {sb[st]:.3f}")
    stage_differences()
