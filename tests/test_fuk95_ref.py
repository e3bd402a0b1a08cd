"""The reference's own test case, restated from its generator routines (blom_amd/cases.py: fuk95_ref_case <-
fuk95/mod_fuk95.F90:117-229, :262-445 and the generic steps phy/mod_inicon.F90:985-1095), and the namelist reader
against the reference's test input tests/fuk95/limits (kept as a data fixture: tests/golden/fuk95_limits).  The generator
modules themselves cannot be built here (mod_fuk95 uses mod_mxlayr -> netCDF; mod_inicon netCDF + GSW), so the state is
checked against what the routines are meant to produce; what the reference's stages then make of it is pinned by
tests/golden/fuk95_ref_crc.json (tests/test_oracle_golden.py, tests/test_gpu_golden.py)."""
import os

import numpy as np

from blom_amd.cases import make_case, FUK95_LIMITS, _delphi, GRAV
from blom_amd.namelist import read_namelist, options_from_namelists

HERE = os.path.dirname(os.path.abspath(__file__))


def test_limits_file_of_the_reference_test_gives_the_case_options():
    g = read_namelist(os.path.join(HERE, "golden", "fuk95_limits"))
    assert set(g) == {"limits", "vcoord", "ale_regrid_remap", "diffusion", "diaphy"}
    o = options_from_namelists(g)
    assert g["limits"]["runid"] == "BLOM_fuk95" and g["limits"]["itest"] == 78 and g["limits"]["aptflx"] is False
    assert g["diaphy"]["glb_fnametag"] == ["hd", "hm"] and g["diaphy"]["glb_aveperio"] == [-8, -8]
    # the dynamical core's options: everything the file sets, except the three that select parts not built
    # (VCOORD_TYPE = 'cntiso_hybrid' with cppm and dynamic enthalpy: the ALE stack)
    for k, v in FUK95_LIMITS.items():
        assert o[k] == v, (k, o[k], v)
    assert (o["vcoord_type"], o["advmth"], o["pgfmth"]) == ("cntiso_hybrid", "cppm", "dynamic enthalpy")
    ref = "/root/reference/tests/fuk95/limits"
    if os.path.exists(ref):                              # the fixture is the reference's file
        assert open(ref).read() == open(os.path.join(HERE, "golden", "fuk95_limits")).read()


def test_state_is_what_the_generator_describes():
    case = make_case("fuk95_ref")
    J, I = slice(4, 4 + case.jdm), slice(5, 3 + case.idm)         # wet interior (i = 2 .. itdm-1)
    kk = case.kdm
    z, phi, dp = case.ic["z"][:, J, I], case.ic["phi"][:, J, I], case.ic["dp"][:, J, I]
    T, S, sg, sr = case.ic["temp"][:, J, I], case.ic["saln"][:, J, I], case.ic["sigma"][:, J, I], case.ic["sigmar"][:, J, I]
    assert case.depth[4:-4, 4].max() == 0.0 and case.depth[4:-4, 3 + case.idm].max() == 0.0      # walls at i = 1, itdm
    assert np.all(case.depth[J, I] == 200.0) and case.grid["scpx"][10, 10] == 20.8e3 / 32
    assert np.all(np.diff(z, axis=0) >= 0.0) and np.all(z[kk] == 200.0) and np.all(z[kk - 1] == 100.0)
    assert np.all(z[1] == 2.5) and np.all(z[2] == 5.0)                                          # mixed layer: mltmin
    # hydrostatic: the geopotential thickness of every layer, integrated with the equation of state over the
    # pressures found by getpl, is -g x its geometric thickness (getpl stops at a correction of 1e-5 Pa)
    p = np.concatenate([np.zeros((1,) + dp.shape[1:]), np.cumsum(dp, axis=0)])
    for k in range(kk):
        dphi, _ = _delphi(p[k], p[k + 1], T[k], S[k])
        assert np.allclose(dphi, phi[k + 1] - phi[k], rtol=0, atol=1e-7), k
    assert np.allclose(phi, -GRAV * z)
    # isopycnic layers carry their reference density; the jet is a front: isopycnals rise across the channel
    assert np.allclose(sg[2:], sr[2:], rtol=0, atol=1e-10)
    mid = z[kk // 2 + 2]
    assert mid[:, 5].mean() - mid[:, -5].mean() > 30.0                  # the front: tens of metres across the jet
    assert np.all(np.abs(np.diff(mid, axis=1)) < 20.0)
    # along-channel perturbation of the jet axis (x_nudge): one wave of amplitude 0.1 grid spacings
    col = np.argmin(np.abs(mid - 50.0), axis=1)
    assert col.max() - col.min() <= 2


def test_case_steps_on_the_restatement():
    from oracle.coracle import COracle, have_coracle
    from blom_amd import hostinit
    from blom_amd.stepper import dyncore_step
    import pytest
    if not have_coracle():
        pytest.skip("oracle/_ref/liboracle_c.so not built")
    case = make_case("fuk95_ref")
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm)
    co = COracle(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    hostinit.init_state(co, case)
    ns = 0
    for _ in range(10):
        ns = dyncore_step(co, ns, case.params["baclin"])
    v = co.get("v")[:, 4:-4, 4:-4]
    assert np.isfinite(v).all() and 1e-4 < np.abs(v).max() < 1.0      # the front spins the jet up: cm/s after half an hour
