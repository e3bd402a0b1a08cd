"""CROSS-CHECK (not a pin) of eddtra for the hybrid vertical coordinate (phy/mod_eddtra.F90:1001-1739 eddtra_ale and the driver's
heat and salt components :1859-1901) on the device against the reference's REAL module.

The reference module compiles against oracle/xcheck/mod_difest_standin.F90, which holds the one array it imports from the
CVMix-bound mod_difest -- OBLdepth, the boundary layer depth, here an uploaded synthetic field -- hence a cross-check (the *_xale /
*_xaln builds of oracle/Makefile carry it).  State: the isopycnic state after a few steps; neutral slopes at the interfaces,
layer diffusivities, mixed layer depth and boundary layer depth are synthetic (their producers are cross-checked on their own):
slopes of both signs from weak to limiter-saturating, mixed layers from a few metres to deeper than the column.  The stage runs
three times in a row so that the running means (hbl_tf, hml_tf1, hml_tf) carry state from call to call, with mlrmth = 'fox08'
(the reference's default) and with mlrmth = 'bod23' (:1058-1081, :1127-1154: friction and convective velocity cubed -- ustar3, wstar3 of
mod_forcing, whose producer for this coordinate is the CVMix-bound difest_vertical_hybrid -- are uploaded synthetic fields; their
weighted sum to the power 2/3 is the host libm's pow on the reference's side, pow_libm.h on the device's; wpup_tf is one more
running mean).  Every array the stage writes must agree bit for bit."""
import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, GRID_FIELDS, INT_FIELDS

pytestmark = pytest.mark.gpu
OUT = ["umfltd", "vmfltd", "umflsm", "vmflsm", "utfltd", "vtfltd", "utflsm", "vtflsm", "usfltd", "vsfltd", "usflsm", "vsflsm",
       "hbl_tf", "hml_tf1", "hml_tf"]


def eddtra_ale_inputs(case, seed, slope):
    rng = np.random.default_rng(seed)
    kk, nj, ni = case.kdm, case.jdm + 8, case.idm + 8
    f = {}
    for nm in ("nslpx", "nslpy"):
        f[nm] = slope * rng.standard_normal((kk, nj, ni)) * 10.0 ** rng.uniform(-2.0, 0.0, (1, nj, ni))
    f["difint"] = rng.uniform(100.0, 2000.0, (kk, nj, ni))
    f["mld"] = 10.0 ** rng.uniform(0.5, 3.5, (1, nj, ni))
    f["OBLdepth"] = 10.0 ** rng.uniform(0.5, 2.5, (1, nj, ni))
    f["ustar3"] = (10.0 ** rng.uniform(-3.0, -1.3, (1, nj, ni))) ** 3        # friction velocities of 1 mm/s .. 5 cm/s
    f["wstar3"] = np.where(rng.random((1, nj, ni)) < 0.4, 0.0, 10.0 ** rng.uniform(-9.0, -4.0, (1, nj, ni)))
    return f


@pytest.mark.parametrize("cfg,method,nsteps,slope,seed,mlrmth", [
    ("fuk95", "nudge", 3, 1e-3, 1, "fox08"), ("chan_s", "direct", 4, 1e-4, 2, "fox08"), ("chan_s", "nudge", 4, 1e-1, 3, "fox08"),
    ("box_s", "direct", 4, 1e-2, 4, "fox08"), ("tri_s", "direct", 3, 1e-2, 5, "fox08"), ("box_s", "nudge", 2, 1.0, 6, "fox08"),
    ("chan_s", "direct", 4, 1e-2, 7, "bod23"), ("box_s", "nudge", 3, 1e-1, 8, "bod23"), ("tri_s", "direct", 3, 1e-3, 9, "bod23"),
    ("fuk95", "direct", 3, 1e-2, 10, "bod23")])
def test_device_eddtra_ale_equals_the_real_module(cfg, method, nsteps, slope, seed, mlrmth):
    from oracle.refblom import get_ref_backend, have_ref
    from blom_amd.gpu import BlomGpu
    lib = cfg + ("_xaln" if method == "nudge" else "_xale")
    if not have_ref(lib):
        pytest.skip(f"oracle/_ref/{lib}/libblomref.so not built")
    case = make_case(cfg)
    ref = get_ref_backend(lib, case.depth)
    if not ref.has_field("OBLdepth") or (mlrmth == "bod23" and not ref.has_field("wpup_tf")):
        pytest.skip("reference library built before eddtra_ale's fields were added to the harness")
    kk = case.kdm
    gpu = BlomGpu(case.idm, case.jdm, kk, ref.ntr, ref.nreg, ref.masks)
    hostinit.init_state(gpu, case)
    assert gpu.step(0, nsteps) == nsteps
    hostinit.init_state(ref, case)
    copy_state(gpu, ref, fields=STATE_FIELDS + GRID_FIELDS + INT_FIELDS)
    six = hostinit.step_indices(nsteps, kk)
    ref.ref.set("vcoord_tag", 2)
    ref.ref.stage("eddtra_init_" + mlrmth, *six)            # inivar_eddtra + init_eddtra: zero running means, mlrmth resolved
    f = eddtra_ale_inputs(case, seed, slope)
    for nm, a in f.items():
        ref.put(nm, a)
        gpu.put(nm, a)
    state = ("hbl_tf", "hml_tf1", "hml_tf") + (("wpup_tf",) if mlrmth == "bod23" else ())
    for nm in state + ("util1",):                           # the reference's initial patterns (spval on land)
        gpu.put(nm, ref.get(nm))
    delt1 = 2.0 * case.params["baclin"]
    ref.ref.set("delt1", delt1)
    gpu.set("delt1", delt1)
    ref.ref.set("eitmth", "gm")
    gpu.set("eitmth", "gm")
    gpu.set("vcoord_type", "cntiso_hybrid")
    gpu.set("mlrmth", mlrmth)
    try:
        for it in range(3):
            ref.ref.stage("eddtra", *six)
            gpu.stage("eddtra", *six)
            bad = diff_report(ref, gpu, fields=OUT + ["util1"] + (["wpup_tf"] if mlrmth == "bod23" else []))
            assert not bad, f"call {it + 1}\n" + fmt_report(bad[:10])
            if it == 0:                                         # deeper mixed layers on the second call, shallower on the third
                for be in (ref, gpu):
                    be.put("mld", 3.0 * f["mld"])
                    be.put("ustar3", 0.3 * f["ustar3"])       # ... and a decaying vertical momentum flux
            else:
                for be in (ref, gpu):
                    be.put("mld", 0.2 * f["mld"])
                    be.put("ustar3", 5.0 * f["ustar3"])
        wu = np.broadcast_to((ref.masks["iu"][4:-4, 4:-4] > 0)[None], (kk, case.jdm, case.idm))
        mm = six[2]
        gm = gpu.get("umfltd")[mm:mm + kk, 4:-4, 4:-4][wu]
        sm = gpu.get("umflsm")[mm:mm + kk, 4:-4, 4:-4][wu]
        assert np.isfinite(gm).all() and np.isfinite(sm).all()
        assert np.abs(gm).max() > 0.0, "no Gent-McWilliams transport"
        assert np.abs(sm).max() > 0.0, "no submesoscale transport"
    finally:
        ref.ref.set("vcoord_tag", 1)
        if mlrmth != "fox08":
            ref.ref.stage("eddtra_init_fox08", *six)        # (the cached backend goes back to the default)
        gpu.close()


def test_bod23_is_refused_with_the_isopycnic_coordinate():
    """init_eddtra stops with mlrmth = 'bod23' and vcoord = 'isopyc_bulkml' (phy/mod_eddtra.F90:1787-1795); so does the stage."""
    from blom_amd.gpu import BlomGpu, BlomGpuError
    case = make_case("chan_s")
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    hostinit.init_state(gpu, case)
    gpu.set("mlrmth", "bod23")
    with pytest.raises(BlomGpuError, match="bod23 is unsupported with vcoord = 'isopyc_bulkml'"):
        gpu.stage("eddtra", *hostinit.step_indices(0, case.kdm))
    gpu.close()
