"""CROSS-CHECK (not a pin) of eddtra against the reference's REAL phy/mod_eddtra.F90.

mod_eddtra imports one array, OBLdepth, from mod_difest, which needs the CVMix library (absent here), so the oracle's
reference builds leave eddtra out and oracle/c/eddtra.c stays "parity unpinned".  The *_xed builds of oracle/Makefile
compile the reference's own mod_eddtra.F90 against oracle/xcheck/mod_difest_standin.F90 -- a stand-in module that holds
only that array, which the isopyc_bulkml branches (phy/mod_eddtra.F90:152-1000) never read.  Because a stand-in is
involved this does not lift the "unpinned" label; it does replace "restatement and kernels by the same
author agree with each other" by "both reproduce the reference's own compiled arithmetic", bit for bit:
  * the whole stage sequence INCLUDING eddtra, reference against the C restatement (CPU suite), Gent-McWilliams with
    weak, moderate and limiter-saturating slopes and interface diffusion;
  * eddtra on the device against the reference stage by stage (GPU suite)."""
import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, INT_FIELDS

OUT = ["umfltd", "vmfltd", "utfltd", "vtfltd", "usfltd", "vsfltd"]
CASES = [("chan_s", "gm", 2e-4), ("chan_s", "gm", 5e-3), ("chan_s", "gm", 0.5), ("box_s", "gm", 5e-3), ("box_s", "gm", 0.5),
         ("fuk95", "gm", 1e-2), ("chan_s", "intdif", 0.0), ("box_s", "intdif", 0.0)]


class _WithEddtra:
    """the reference backend of a *_xed build: its harness knows the stage eddtra"""
    def __init__(self, be):
        self._be = be

    def __getattr__(self, nm):
        return getattr(self._be, nm)

    def has_stage(self, name):
        return True


def _reference(cfg, eitmth, nslp0):
    from oracle.refblom import get_ref_backend, have_ref
    if not have_ref(cfg + "_xed"):
        pytest.skip(f"oracle/_ref/{cfg}_xed/libblomref.so not built")
    case = make_case(cfg, nslp0=nslp0, eitmth=eitmth)
    ref = _WithEddtra(get_ref_backend(cfg + "_xed", case.depth))
    ref.ref.set("eitmth", eitmth)
    hostinit.init_state(ref, case)
    return case, ref


@pytest.mark.parametrize("cfg,eitmth,nslp0", CASES)
def test_restatement_equals_the_real_mod_eddtra(cfg, eitmth, nslp0):
    from oracle.coracle import COracle, have_coracle
    if not have_coracle():
        pytest.skip("C oracle not built")
    case, ref = _reference(cfg, eitmth, nslp0)
    co = COracle(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            co.set(nm, v)
    copy_state(ref, co, fields=STATE_FIELDS + INT_FIELDS + ["nslpx", "nslpy"] +
               ["scqx", "scqy", "scpx", "scpy", "scux", "scuy", "scvx", "scvy", "scq2", "scp2", "scu2", "scv2", "scq2i", "scp2i",
                "scuxi", "scuyi", "scvxi", "scvyi", "corioq", "coriop"])
    co.set("delt1", case.params["baclin"])
    fields = [f for f in STATE_FIELDS + INT_FIELDS if f not in ("util1", "util2")]
    nr = nc = 0
    nonzero = 0
    for _ in range(6):
        nr = dyncore_step(ref, nr, case.params["baclin"])
        nc = dyncore_step(co, nc, case.params["baclin"])
        bad = diff_report(ref, co, fields=fields)
        assert not bad, f"step {nr}\n" + fmt_report(bad[:8])
        nonzero += int(np.count_nonzero(ref.get("umfltd")[:, 4:-4, 4:-4] * (np.abs(ref.get("umfltd")[:, 4:-4, 4:-4]) < 1e30)))
    assert nonzero > 0, "the case did not produce any eddy-induced transport"


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,eitmth,nslp0", CASES)
def test_device_eddtra_equals_the_real_mod_eddtra(cfg, eitmth, nslp0):
    from blom_amd.gpu import BlomGpu
    case, ref = _reference(cfg, eitmth, nslp0)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            gpu.set(nm, v)
    fails, pend, nstep = [], {}, [0]

    def check():
        if pend.pop("st", None):
            bad = diff_report(ref, gpu, fields=OUT)
            if bad:
                fails.append(f"step {nstep[0] + 1}:\n" + fmt_report(bad))

    def hook(st, six):
        check()
        if st != "eddtra":
            return
        copy_state(ref, gpu, fields=STATE_FIELDS + INT_FIELDS + ["nslpx", "nslpy"] +
                   ["scp2", "scuy", "scvx", "scu2", "scv2", "scuxi", "scvyi"])
        gpu.set("delt1", ref.ref.get_real("delt1"))
        gpu.stage("eddtra", *six)
        pend["st"] = True

    for _ in range(4):
        new = dyncore_step(ref, nstep[0], case.params["baclin"], hook=hook)
        check()
        nstep[0] = new
    gpu.close()
    assert not fails, "\n".join(fails[:10])


@pytest.mark.gpu
def test_full_size_sequence_with_eddtra_equals_the_reference_with_its_real_mod_eddtra():
    """BASELINE.json's channel at full size (208x512x53, ntr = 3) with the analytic slopes of bench.py --slopes frozen
    (nslp0 = 2e-4) driving Gent-McWilliams transports: the device-resident sequence, eddtra included, against the
    reference's own Fortran with its real mod_eddtra (channel_tke_omp_xed build: OpenMP, stand-in for mod_difest's one
    array) over the forward step and three leap-frog steps.  Bit for bit.  Cross-check, not a pin (see the module header)."""
    import os
    import threading
    from oracle.refblom import get_ref_backend, have_ref
    from blom_amd.gpu import BlomGpu
    if not have_ref("channel_tke_omp_xed"):
        pytest.skip("oracle/_ref/channel_tke_omp_xed/libblomref.so not built")
    nsteps, res = 4, {}

    def body():
        case = make_case("channel_tke", nslp0=2.0e-4)
        ref = _WithEddtra(get_ref_backend("channel_tke_omp_xed", case.depth))
        ref.ref.set("eitmth", "gm")
        hostinit.init_state(ref, case)
        gpu = BlomGpu(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
        for nm, v in case.params.items():
            if not nm.endswith("0"):
                gpu.set(nm, v)
        copy_state(ref, gpu, fields=STATE_FIELDS + INT_FIELDS + ["nslpx", "nslpy"] +
                   ["scqx", "scqy", "scpx", "scpy", "scux", "scuy", "scvx", "scvy", "scq2", "scp2", "scu2", "scv2", "scq2i", "scp2i",
                    "scuxi", "scuyi", "scvxi", "scvyi", "corioq", "coriop"])
        gpu.set("delt1", case.params["baclin"])
        ns = 0
        for _ in range(nsteps):
            ns = dyncore_step(ref, ns, case.params["baclin"])
        assert gpu.step(0, nsteps) == nsteps
        gpu.sync()
        res["bad"] = diff_report(ref, gpu, fields=["u", "v", "dp", "temp", "saln", "sigma", "pb", "ub", "vb", "trc", "uflx", "vflx",
                                                   "umfltd", "vmfltd", "utfltd", "usfltd", "dpu", "dpv"])
        um = ref.get("umfltd")[:, 4:-4, 4:-4]
        res["nonzero"] = int(np.count_nonzero(um * (np.abs(um) < 1e30)))
        gpu.close()

    os.environ["OMP_NUM_THREADS"] = str(min(16, os.cpu_count() or 1))
    os.environ["OMP_STACKSIZE"] = "1G"
    threading.stack_size(2 << 30)
    th = threading.Thread(target=body)
    th.start()
    th.join()
    threading.stack_size(0)
    assert "bad" in res, "the comparison did not complete"
    assert res["nonzero"] > 100000
    assert not res["bad"], fmt_report(res["bad"])
