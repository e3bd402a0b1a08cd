"""CROSS-CHECK (not a pin) of the device's ale_forcing against the reference's REAL phy/mod_ale_forcing.F90.

The module imports the two-band shortwave absorption of mod_swabs, which reads chlorophyll climatologies with netCDF; the
*_xale builds of oracle/Makefile compile the real mod_ale_forcing against oracle/xcheck/mod_swabs_standin.F90 (a module that
holds just the five imported variables) -- hence a cross-check.  Inputs are synthetic: absorption bands in the
range of the reference's Jerlov water types (phy/mod_swabs.F90:100-140), mixed layer depths from a few metres to deeper than
the column, surface fluxes of both signs; state: the isopycnic state after a few steps, with massless layers.  t_sw_nonloc,
s_br_nonloc and buoyfl must agree bit for bit (the exponentials through the library's libm-exact exp)."""
import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, GRID_FIELDS, INT_FIELDS

pytestmark = pytest.mark.gpu
OUT = ["t_sw_nonloc", "s_br_nonloc", "buoyfl"]


@pytest.mark.parametrize("cfg,nsteps,seed,frac", [("chan_s", 4, 1, 0.4), ("box_s", 3, 2, 0.0), ("fuk95", 3, 3, 1.0), ("tri_s", 3, 4, 0.25)])
def test_device_ale_forcing_equals_the_real_module(cfg, nsteps, seed, frac):
    from oracle.refblom import get_ref_backend, have_ref
    from blom_amd.gpu import BlomGpu
    lib = cfg + "_xale"
    if not have_ref(lib):
        pytest.skip(f"oracle/_ref/{lib}/libblomref.so not built")
    case = make_case(cfg)
    ref = get_ref_backend(lib, case.depth)
    if not ref.has_field("swfc1"):
        pytest.skip("reference library built before ale_forcing was added")
    kk, nj, ni = case.kdm, case.jdm + 8, case.idm + 8
    gpu = BlomGpu(case.idm, case.jdm, kk, ref.ntr, ref.nreg, ref.masks)
    hostinit.init_state(gpu, case)
    assert gpu.step(0, nsteps) == nsteps
    hostinit.init_state(ref, case)
    copy_state(gpu, ref, fields=STATE_FIELDS + GRID_FIELDS + INT_FIELDS)
    rng = np.random.default_rng(seed)
    depth_m = float(np.max(gpu.get("p")[kk][ref.masks["ip"] > 0])) / 9806.0
    f = {"swfc1": rng.uniform(0.3, 0.8, (1, nj, ni))}
    f["swfc2"] = 1.0 - f["swfc1"]
    f["swal1"] = rng.uniform(0.35, 2.0, (1, nj, ni))
    f["swal2"] = rng.uniform(8.0, 25.0, (1, nj, ni))
    f["mld"] = rng.uniform(0.01, 1.5, (1, nj, ni)) * depth_m
    f["sswflx"] = -rng.uniform(0.0, 250.0, (1, nj, ni))
    f["surflx"] = f["sswflx"] + rng.uniform(-300.0, 300.0, (1, nj, ni))
    f["brnflx"] = -rng.uniform(0.0, 1e-3, (1, nj, ni))
    f["salflx"] = f["brnflx"] + rng.uniform(-2e-2, 2e-2, (1, nj, ni))
    for be in (ref, gpu):
        for nm, a in f.items():
            be.put(nm, a)
    swamxd = 0.6 * depth_m
    ref.ref.set("swamxd", swamxd)
    ref.ref.set("brine_mlbase_frac", frac)
    gpu.set("swamxd", swamxd)
    gpu.set("brine_mlbase_frac", frac)
    ref.ref.set("vcoord_tag", 2)
    gpu.set("vcoord_type", "cntiso_hybrid")
    six = hostinit.step_indices(nsteps, kk)
    try:
        for nm in OUT:                       # the reference's initial pattern where the stage does not write
            gpu.put(nm, ref.get(nm))
        ref.ref.stage("ale_forcing", *six)
        gpu.stage("ale_forcing", *six)
        bad = diff_report(ref, gpu, fields=OUT)
        assert not bad, fmt_report(bad[:10])
        wet = ref.masks["ip"][4:-4, 4:-4] > 0
        t = gpu.get("t_sw_nonloc")[:, 4:-4, 4:-4][:, wet]
        assert (t[0] == 1.0).all() and (t[-1] == 0.0).all() and (np.diff(t, axis=0) <= 1e-15).all()
        assert np.abs(gpu.get("buoyfl")[:, 4:-4, 4:-4][:, wet]).max() > 0.0
    finally:
        ref.ref.set("vcoord_tag", 1)
        gpu.close()
