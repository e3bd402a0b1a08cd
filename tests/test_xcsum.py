"""xcsum (phy/mod_xc.F90:4116-4161) and budget_sums (phy/mod_budget.F90:95-196).

xcsum is the reference's reproducible global sum -- strips of 2*nbdy+1 points per row, the strip sums added in
order, the row sums added serially -- so its result is defined to the last bit and is compared that way:
  * C restatement (oracle/c/simple.c: orc_xcsum, orc_budget_sums) against the reference's compiled routines -- CPU,
  * device (blom_amd/csrc/halo.hip: k_xcsum; stage_simple.hip: k_budget_columns) through the C ABI -- GPU.
The sums of mod_budget are private to the module; what budget_sums leaves in util1/util2 is public, and the
reference's xcsum of those is what its private sdp/tdp/trdp hold.
"""
import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step
from parity import copy_state

CFGS = ["chan_s", "box_s", "fuk95", "tri_s", "chan_s_tke", "tri_s_tke", "chan_s_tk2"]


def _setup(cfg, make_other, nsteps=2):
    from oracle.refblom import get_ref_backend, have_ref
    if not have_ref(cfg):
        pytest.skip(f"oracle/_ref/{cfg}/libblomref.so not built")
    case = make_case(cfg)
    ref = get_ref_backend(cfg, case.depth)
    hostinit.init_state(ref, case)
    nstep = 0
    for _ in range(nsteps):
        nstep = dyncore_step(ref, nstep, case.params["baclin"])
    other = make_other(case, ref)
    copy_state(ref, other)
    return case, ref, other, nstep


def _check_xcsum(case, ref, other):
    kk = case.kdm
    rng = np.random.default_rng(5)
    shape = np.asarray(ref.get("util3")).reshape(-1, case.jdm + 8, case.idm + 8).shape
    # a field with 30 orders of magnitude of dynamic range: any other order of summation shows
    wild = rng.standard_normal(shape) * 10.0 ** rng.integers(-15, 15, shape)
    ref.put("util3", wild)
    other.put("util3", wild)
    n = 0
    for name, lev, itype in [("temp", 1, 1), ("temp", kk + 3, 1), ("dp", 2, 1), ("u", 1, 3), ("v", kk + 1, 4),
                             ("pb", 1, 1), ("pbu", 2, 3), ("util3", 1, 1), ("util3", 1, 2), ("util3", 1, 3),
                             ("util3", 1, 4), ("scp2", 1, 1)]:
        a = np.asarray(ref.get(name)).reshape(-1, case.jdm + 8, case.idm + 8)[lev - 1]
        want = ref.xcsum(a, itype)
        got = other.xcsum(name, lev, itype)
        assert got == want, (name, lev, itype, got, want)
        n += want != 0.0
    assert n >= 10


def _check_budget(case, ref, other, nstep):
    kk = case.kdm
    n = nstep % 2 + 1
    nn = (n - 1) * kk
    u1 = np.array(ref.get("util1"))
    # off: nothing happens (phy/mod_budget.F90:105)
    ref.set("cnsvdi", 0)
    other.set("cnsvdi", 0)
    ref.budget_sums(2, n, nn)
    other.budget_sums(2, n, nn)
    assert np.array_equal(np.asarray(ref.get("util1")), u1)
    assert np.array_equal(np.asarray(other.get("util1")), u1)
    ref.set("cnsvdi", 1)
    other.set("cnsvdi", 1)
    ref.budget_sums(2, n, nn)
    other.budget_sums(2, n, nn)
    for nm in ("util1", "util2"):
        assert np.array_equal(np.asarray(ref.get(nm)), np.asarray(other.get(nm))), nm
    pl = lambda nm: np.asarray(ref.get(nm)).reshape(-1, case.jdm + 8, case.idm + 8)
    assert other.budget_get("tdp", 2, n) == ref.xcsum(pl("util2")[0], 1)
    if ref.ntr >= 1:
        assert other.budget_get("trdp", 2, n) == ref.xcsum(pl("util1")[0], 1)
    if case.params.get("itrtke", -1) >= 1:
        assert np.array_equal(np.asarray(ref.get("util3")), np.asarray(other.get("util3")))
        assert other.budget_get("tkedp", 2, n) == ref.xcsum(pl("util3")[0], 1) != 0.0
    # sdp: the salt columns were overwritten by the tracer's; rebuild them with the statement order of :130-132
    s = np.zeros_like(pl("util1")[0])
    for k in range(kk):
        s = s + pl("saln")[k + nn] * (pl("dp")[k + nn] * pl("scp2")[0])
    want = ref.xcsum(s, 1)
    assert other.budget_get("sdp", 2, n) == want
    assert want != 0.0 and other.budget_get("tdp", 2, n) != 0.0
    assert other.budget_get("sdp", 3, n) == 0.0          # other calls of the step untouched


@pytest.mark.parametrize("cfg", CFGS)
def test_c_restatement_xcsum_and_budget_sums_match_reference(cfg):
    from oracle.coracle import COracle, have_coracle
    if not have_coracle():
        pytest.skip("oracle/_ref/liboracle_c.so not built")

    def make(case, ref):
        co = COracle(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
        for nm, v in case.params.items():
            if not nm.endswith("0"):
                co.set(nm, v)
        return co
    case, ref, co, nstep = _setup(cfg, make)
    _check_xcsum(case, ref, co)
    _check_budget(case, ref, co, nstep)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", CFGS)
def test_device_xcsum_and_budget_sums_match_reference(cfg):
    from blom_amd.gpu import BlomGpu

    def make(case, ref):
        gpu = BlomGpu(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
        for nm, v in case.params.items():
            if not nm.endswith("0"):
                gpu.set(nm, v)
        return gpu
    case, ref, gpu, nstep = _setup(cfg, make)
    try:
        _check_xcsum(case, ref, gpu)
        _check_budget(case, ref, gpu, nstep)
    finally:
        gpu.close()
