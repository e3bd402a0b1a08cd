"""cmnfld2 on the device: bit for bit the C restatement (oracle/c/cmnfld.c; both unpinned, both written from
phy/mod_cmnfld_routines.F90:61-227, :423-652, :1158-1238), and the construction checks of tests/test_cmnfld.py on the
device itself.  With `live_slopes` the stage replaces the frozen analytic slopes in the step: eddtra then runs on the
slopes of the evolving state."""
import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step, DYNCORE_STAGES
from blom_amd.hostinit import step_indices
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, INT_FIELDS

pytestmark = pytest.mark.gpu
OUT = ["bfsqi", "bfsql", "bfsqf", "nslpx", "nslpy", "nnslpx", "nnslpy", "phi", "kfpla", "temp", "saln"]
LIVE = tuple("cmnfld2" if s == "halo_cmnfld2" else s for s in DYNCORE_STAGES)


@pytest.mark.parametrize("cfg", ["chan_s", "box_s", "tri_s", "chan_s_tke", "fuk95"])
def test_cmnfld2_matches_restatement_in_a_free_run(cfg):
    """C restatement stepped through the sequence with live slopes; before every cmnfld2 its state goes to the device,
    the stage runs there too, all outputs are compared; the slopes must become non-trivial."""
    from oracle.coracle import COracle
    from blom_amd.gpu import BlomGpu
    case = make_case(cfg, nslp0=0.0)
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm, arctic=case.nreg == 2)
    masks = dict(ip=ip, iu=iu, iv=iv, iq=iq)
    orc = COracle(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
    hostinit.init_state(orc, case)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            gpu.set(nm, v)
    fails, pend, ns, amp = [], {}, [0], [0.0]

    def check():
        if pend.pop("st", None):
            bad = diff_report(orc, gpu, fields=OUT)
            amp[0] = max(amp[0], float(np.abs(orc.get("nslpx")).max()))
            if bad:
                fails.append(f"step {ns[0] + 1}:\n" + fmt_report(bad))

    def hook(st, six):
        check()
        if st != "cmnfld2":
            return
        copy_state(orc, gpu)
        gpu.stage("cmnfld2", *six)
        pend["st"] = True

    for _ in range(4):
        new = dyncore_step(orc, ns[0], case.params["baclin"], stages=LIVE, hook=hook)
        check()
        ns[0] = new
    gpu.close()
    assert not fails, "\n".join(fails[:10])
    assert amp[0] > 0.0


def test_construction_checks_on_the_device():
    from blom_amd.gpu import BlomGpu
    from test_cmnfld import _setup, check_bfsq, check_slope_of_tilted_layers
    case, be, masks = _setup(backend=BlomGpu)
    check_bfsq(be, case, masks)
    be.close()
    case, be, masks = _setup(backend=BlomGpu)
    check_slope_of_tilted_layers(be, case, masks)
    be.close()
    from test_cmnfld import check_depths
    case, be, masks = _setup(backend=BlomGpu)
    check_depths(be, case, masks)
    be.close()


@pytest.mark.parametrize("cfg", ["chan_s", "box_s"])
def test_cmnfld1_matches_restatement(cfg):
    """cmnfld1 (z, dz) at the end of a step: the device bit for bit the C restatement, stage call and step option alike"""
    from blom_amd.gpu import BlomGpu
    from oracle.coracle import COracle
    case = make_case(cfg, nslp0=0.0)
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm)
    masks = dict(ip=ip, iu=iu, iv=iv, iq=iq)
    orc = COracle(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
    hostinit.init_state(orc, case)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
    hostinit.init_state(gpu, case)
    gpu.set("live_slopes", 1)
    gpu.set("cmnfld1", 1)
    ns = 0
    for _ in range(3):
        six = step_indices(ns, case.kdm)
        ns = dyncore_step(orc, ns, case.params["baclin"], stages=LIVE)
        orc.stage("cmnfld1", *six)
    assert gpu.step(0, 3) == 3
    bad = diff_report(orc, gpu, fields=["z", "dz", "dp", "temp", "phi"])
    J, I = slice(4, 4 + case.jdm), slice(4, 4 + case.idm)
    assert np.abs(gpu.get("dz")[:, J, I]).max() > 1.0
    gpu.close()
    assert not bad, fmt_report(bad)


def test_live_slopes_in_the_device_resident_step():
    """blomgpu_step with the option live_slopes = 1 equals the stage-by-stage sequence with cmnfld2 in place of its halo part"""
    from blom_amd.gpu import BlomGpu
    case = make_case("chan_s", nslp0=0.0)
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm)
    masks = dict(ip=ip, iu=iu, iv=iv, iq=iq)
    a = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
    hostinit.init_state(a, case)
    b = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
    hostinit.init_state(b, case)
    b.set("live_slopes", 1)
    ns = 0
    for _ in range(5):
        ns = dyncore_step(a, ns, case.params["baclin"], stages=LIVE)
    assert b.step(0, 5) == 5
    bad = diff_report(a, b, fields=["u", "v", "dp", "temp", "saln", "nslpx", "nslpy", "umfltd", "vmfltd"])
    assert np.abs(a.get("umfltd")).max() > 0.0, "the live slopes did not drive any eddy-induced transport"
    a.close()
    b.close()
    assert not bad, fmt_report(bad)
