"""eddtra on the device against the C restatement (oracle/c/eddtra.c), bit for bit.

The restatement is UNPINNED -- the reference module cannot be compiled in this image
(mod_eddtra -> mod_difest -> CVMix) -- so what is established here is that the HIP kernels and the
restatement, both written from phy/mod_eddtra.F90:152-1000,1808-1857, agree on every bit, for the
Gent-McWilliams branch (weak, moderate and limiter-saturating frozen slopes) and for interface
diffusion, at every step of a short integration of the full stage sequence."""
import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, INT_FIELDS

pytestmark = pytest.mark.gpu
OUT = ["umfltd", "vmfltd", "utfltd", "vtfltd", "usfltd", "vsfltd"]


def _bigrid(case):
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm)
    return nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq)


@pytest.mark.parametrize("cfg,eitmth,nslp0", [("chan_s", "gm", 2e-4), ("chan_s", "gm", 5e-3), ("chan_s", "gm", 0.5),
                                              ("box_s", "gm", 5e-3), ("box_s", "gm", 0.5), ("fuk95", "gm", 1e-2),
                                              ("chan_s", "intdif", 0.0), ("box_s", "intdif", 0.0)])
def test_eddtra_matches_restatement(cfg, eitmth, nslp0):
    from oracle.coracle import COracle
    from blom_amd.gpu import BlomGpu
    case = make_case(cfg, nslp0=nslp0, eitmth=eitmth)
    nreg, masks = _bigrid(case)
    orc = COracle(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
    hostinit.init_state(orc, case)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            gpu.set(nm, v)
    fails, pend, nstep, nonzero = [], {}, [0], [0]

    def check():
        if pend.pop("st", None):
            bad = diff_report(orc, gpu, fields=OUT)
            nonzero[0] += int(np.count_nonzero(orc.get("umfltd"))) + int(np.count_nonzero(orc.get("vmfltd")))
            if bad:
                fails.append(f"step {nstep[0] + 1}:\n" + fmt_report(bad))

    def hook(st, six):
        check()
        if st != "eddtra":
            return
        copy_state(orc, gpu, fields=STATE_FIELDS + INT_FIELDS + ["nslpx", "nslpy"] +
                   ["scp2", "scuy", "scvx", "scu2", "scv2", "scuxi", "scvyi"])
        gpu.set("delt1", orc.get_real("delt1") if hasattr(orc, "get_real") else (case.params["baclin"] * (1 if nstep[0] == 0 else 2)))
        gpu.stage("eddtra", *six)
        pend["st"] = True

    for _ in range(4):
        new = dyncore_step(orc, nstep[0], case.params["baclin"], hook=hook)
        check()
        nstep[0] = new
    gpu.close()
    assert not fails, "\n".join(fails[:10])
    assert nonzero[0] > 0, "the case did not produce any eddy-induced transport"


@pytest.mark.parametrize("cfg", ["chan_s", "box_s"])
def test_freerun_with_eddtra(cfg):
    """Whole sequence incl. eddtra, device-resident loop vs C restatement (exp() tolerance as in
    test_gpu_stage_parity.test_freerun_device_resident)."""
    from oracle.coracle import COracle
    from blom_amd.gpu import BlomGpu
    case = make_case(cfg, nslp0=5e-3)
    nreg, masks = _bigrid(case)
    orc = COracle(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
    hostinit.init_state(orc, case)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
    hostinit.init_state(gpu, case)
    ns = 0
    for _ in range(4):
        ns = dyncore_step(orc, ns, case.params["baclin"])
    assert gpu.step(0, 4) == 4
    bad = diff_report(orc, gpu, fields=["u", "v", "dp", "temp", "saln", "umfltd", "vmfltd", "uflx", "vflx"], rtol=1e-7, atol=1e-7)
    gpu.close()
    assert not bad, fmt_report(bad)


def test_uniform_slope_gives_the_streamfunction_of_the_equations_on_the_device():
    """The analytic column of tests/test_oracle_eddtra.py (expectation written from the equations) on the device."""
    from blom_amd.gpu import BlomGpu
    from blom_amd.hostinit import step_indices
    from test_oracle_eddtra import uniform_slope_expectation, check_uniform_slope
    case = make_case("chan_s", nslp0=0.0)
    nreg, masks = _bigrid(case)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
    hostinit.init_state(gpu, case)
    m_, n_, mm, nn, k1m, k1n = step_indices(0, case.kdm)
    X = uniform_slope_expectation(gpu, case, 1.0e-3, 300.0)
    gpu.stage("eddtra", m_, n_, mm, nn, k1m, k1n)
    check_uniform_slope(gpu, case, X, masks, nn, mm)
    gpu.close()


def test_eddtra_of_the_bench_workload_at_full_size_matches_the_restatement():
    """BASELINE.json's channel at full size with the frozen slopes bench.py runs (nslp0 = 2e-4): over three steps of the
    device-resident sequence, eddtra's inputs are handed to the C restatement before the stage and its six outputs
    compared after it, all bits.  (The restatement is unpinned; this closes the gap between the reference-pinned
    full-size runs, which use zero slopes, and the bench workload.)"""
    from oracle.coracle import COracle
    from blom_amd.gpu import BlomGpu
    from blom_amd.hostinit import step_indices
    case = make_case("channel_tke", nslp0=2e-4)
    nreg, masks = _bigrid(case)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
    hostinit.init_state(gpu, case)
    orc = COracle(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            orc.set(nm, v)
    static = ["scp2", "scuy", "scvx", "scu2", "scv2", "scuxi", "scvyi", "difint", "nslpx", "nslpy"]
    dynamic = ["dp", "p", "temp", "saln", "kfpla", "pbu", "pbv", "dpu", "dpv"]
    copy_state(gpu, orc, fields=static)
    ns, bad, nz = 0, [], 0
    for step in range(3):
        six = step_indices(ns, case.kdm)
        gpu.set("nstep", ns + 1)
        from blom_amd.stepper import DYNCORE_STAGES
        for st in DYNCORE_STAGES:
            if st == "eddtra":
                copy_state(gpu, orc, fields=dynamic)
                orc.set("delt1", case.params["baclin"] * (1 if ns == 0 else 2))
                orc.stage("eddtra", *six)
                gpu.stage("eddtra", *six)
                rep = diff_report(orc, gpu, fields=OUT)
                nz += int(np.count_nonzero(orc.get("umfltd")))
                if rep:
                    bad.append(f"step {step + 1}:\n" + fmt_report(rep))
            else:
                gpu.stage(st, *six)
        gpu.set("delt1", 2 * case.params["baclin"])
        ns += 1
    gpu.close()
    assert not bad, "\n".join(bad)
    assert nz > 100000
