"""Free run at BASELINE.json's channel size: the device-resident loop against the reference's own Fortran (OpenMP build,
oracle/_ref/channel_tke_omp) for N steps from the same initial state, compared bit for bit every `every` steps.
usage: python tools/gpu_fullsize_freerun.py [nsteps=100] [every=25] [cfg=channel_tke | tnx2v1s_tke] [live]
live: the sequence bench.py times -- cmnfld2's slopes from the evolving state and eddtra on them -- against the cross-check
build of the reference (oracle/_ref/channel_tke_omp_xed: its real mod_cmnfld_routines and mod_eddtra, compiled against the
stand-ins of oracle/xcheck/; a cross-check, not a pin)."""
import os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step
from parity import copy_state, diff_report, fmt_report

FIELDS = ["u", "v", "dp", "temp", "saln", "sigma", "trc", "p", "pb", "ub", "vb", "dpu", "dpv", "uflx", "vflx", "pgfx", "pgfy",
          "ubflxs_p", "pb_p", "kfpla"]


def body(nsteps, every, cfg, live=False):
    from oracle.refblom import get_ref_backend
    from blom_amd.gpu import BlomGpu
    from blom_amd.stepper import DYNCORE_STAGES
    case = make_case(cfg, nslp0=0.0) if live else make_case(cfg)
    ref = get_ref_backend(cfg + ("_omp_xed" if live else "_omp"), case.depth)
    stages = DYNCORE_STAGES
    if live:
        ref.ref.set("eitmth", "gm")
        ref.has_stage = lambda name: True
        stages = tuple("cmnfld2" if s == "halo_cmnfld2" else s for s in DYNCORE_STAGES)
    hostinit.init_state(ref, case)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            gpu.set(nm, v)
    copy_state(ref, gpu)
    if live:
        copy_state(ref, gpu, fields=["nslpx", "nslpy", "nnslpx", "nnslpy", "bfsqi", "bfsql", "bfsqf"])
        gpu.set("live_slopes", 1)
        FIELDS.extend(["umfltd", "vmfltd", "nslpx", "nslpy"])
    gpu.set("delt1", case.params["baclin"])
    nr = ng = 0
    t0 = time.time()
    while nr < nsteps:
        for _ in range(min(every, nsteps - nr)):
            nr = dyncore_step(ref, nr, case.params["baclin"], stages=stages)
        ng = gpu.step(ng, nr - ng)
        gpu.sync()
        bad = diff_report(ref, gpu, fields=FIELDS)
        ua = np.abs(np.asarray(ref.get("u")))
        umax = float(ua[ua < 1e30].max())                       # land points hold the reference's fill value
        print(f"step {nr}: {'bit-identical in ' + str(len(FIELDS)) + ' fields' if not bad else 'DIFFERS'}; max|u| = {umax:.4f} m/s; "
              f"{time.time() - t0:.0f} s", flush=True)
        if bad:
            print(fmt_report(bad[:10]), flush=True)
            break
    gpu.close()


os.environ["OMP_NUM_THREADS"] = str(min(16, os.cpu_count() or 1))
os.environ["OMP_STACKSIZE"] = "1G"
threading.stack_size(2 << 30)
a = sys.argv[1:]
th = threading.Thread(target=body, args=(int(a[0]) if a else 100, int(a[1]) if len(a) > 1 else 25,
                                        a[2] if len(a) > 2 else "channel_tke", len(a) > 3 and a[3] == "live"))
th.start()
th.join()
