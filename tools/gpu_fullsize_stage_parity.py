"""Per-stage comparison of the device against the reference at BASELINE.json's channel size (needs a GPU and
oracle/_ref/channel_tke_omp).  Before every stage the device gets the reference's state; after it the fields are
compared.  usage: python tools/gpu_fullsize_stage_parity.py [nsteps]"""
import os, sys, threading
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, INT_FIELDS

# utotn, vtotn: per-layer work arrays that are thread-private in the OpenMP build of the reference (phy/mod_momtum.F90:342)
SCRATCH = {"uflux", "vflux", "uflux2", "vflux2", "uflux3", "vflux3", "utotm", "vtotm", "util1", "util2", "util3", "util4",
           "utotn", "vtotn"}
TOL = {}


def body(nsteps):
    from oracle.refblom import get_ref_backend
    from blom_amd.gpu import BlomGpu
    cfg = "channel_tke"
    case = make_case(cfg)
    ref = get_ref_backend(cfg + "_omp", case.depth)
    hostinit.init_state(ref, case)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            gpu.set(nm, v)
    pending, nstep = {}, [0]
    fields = [f for f in STATE_FIELDS + INT_FIELDS if f not in SCRATCH]

    def check():
        if "st" in pending:
            st = pending.pop("st")
            rtol, atol = TOL.get(st, (0.0, 0.0))
            bad = diff_report(ref, gpu, fields=fields, rtol=rtol, atol=atol)
            print(f"step {nstep[0] + 1} stage {st}: " + ("ok" if not bad else "\n" + fmt_report(bad)), flush=True)

    def hook(st, six):
        check()
        if not gpu.has_stage(st) if hasattr(gpu, "has_stage") else False:
            return
        copy_state(ref, gpu)
        gpu.set("nstep", nstep[0] + 1)
        gpu.set("delt1", ref.ref.get_real("delt1"))
        try:
            gpu.stage(st, *six)
        except Exception as e:
            print("stage", st, "not run on device:", e)
            return
        pending["st"] = st

    for _ in range(nsteps):
        new = dyncore_step(ref, nstep[0], case.params["baclin"], hook=hook)
        check()
        nstep[0] = new
    gpu.close()


os.environ["OMP_NUM_THREADS"] = str(min(16, os.cpu_count() or 1))
os.environ["OMP_STACKSIZE"] = "1G"
threading.stack_size(2 << 30)
th = threading.Thread(target=body, args=(int(sys.argv[1]) if len(sys.argv) > 1 else 3,))
th.start()
th.join()
