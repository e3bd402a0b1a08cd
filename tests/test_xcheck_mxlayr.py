"""CROSS-CHECK (not a pin) of the device's mxlayr against the reference's REAL phy/mod_mxlayr.F90.

mod_mxlayr has a bare `use mod_nctools` (netCDF-bound; it is where the module gets ii, jj, kk from) and imports three variables
of the netCDF-bound mod_swabs, so it is compiled against oracle/xcheck/mod_nctools_standin.F90 and mod_swabs_standin.F90 in
the *_xml builds of oracle/Makefile (with the reference's real mod_eddtra -- ce, lfmin, tau_mlr -- and mod_niw): hence a
cross-check, not a pin.  What is compared: after a few steps of the isopycnic sequence on the device (uneven
layers, massless ones, non-zero velocities) the state goes to the reference, both get the same synthetic surface fluxes --
heating and cooling, fresh water and brine, shortwave, relaxation, strong and weak winds, so that columns with a turbulent
kinetic energy deficit (detrainment into the isopycnic layers, the fossil mixed layer's placement rules) and with a surplus
(entrainment through several layers) both occur, with and without brine plumes -- and run mxlayr twice; every array the stage
writes must agree bit for bit: dp, T, S, sigma, tracers, u, v, dpu, dpv, p, pu, pv, kfpla, the six TKE tendencies, pbrnda,
buoyfl, salt_corr, trc_corr."""
import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, GRID_FIELDS, INT_FIELDS

pytestmark = pytest.mark.gpu

OUT = ["dp", "temp", "saln", "sigma", "trc", "u", "v", "dpu", "dpv", "p", "pu", "pv", "kfpla", "mtkeus", "mtkeni", "mtkebf", "mtkers",
       "mtkepe", "mtkeke", "pbrnda", "buoyfl", "salt_corr", "trc_corr"]
FORCING = ["surflx", "sswflx", "surrlx", "salflx", "brnflx", "salrlx", "swfc2", "swal2", "ustar", "ustar3", "idkedt", "trflx"]


def forcing_fields(case, ntr, seed, strength):
    """synthetic surface forcing: smooth large-scale patterns of either sign times a random factor"""
    rng = np.random.default_rng(seed)
    nj, ni = case.jdm + 8, case.idm + 8
    y = np.linspace(0.0, 1.0, nj)[:, None] + 0.0 * np.arange(ni)[None, :]
    x = np.linspace(0.0, 1.0, ni)[None, :] + 0.0 * y
    wave = np.sin(2 * np.pi * (x + 0.3 * y)) * np.cos(np.pi * y)
    f = {}
    f["sswflx"] = -rng.uniform(0.0, 250.0, (1, nj, ni)) * strength * (wave > -0.2)
    f["surflx"] = f["sswflx"] + strength * 400.0 * wave * rng.uniform(0.2, 1.0, (1, nj, ni))     # cooling where positive
    f["surrlx"] = rng.uniform(-30.0, 30.0, (1, nj, ni)) * strength
    f["brnflx"] = -rng.uniform(0.0, 2e-3, (1, nj, ni)) * (rng.uniform(size=(1, nj, ni)) < 0.4)  # brine rejection in 40 % of the columns
    f["salflx"] = f["brnflx"] + rng.uniform(-1e-2, 1e-2, (1, nj, ni)) * strength
    f["salrlx"] = rng.uniform(-3e-3, 3e-3, (1, nj, ni))
    f["swfc2"] = rng.uniform(0.3, 0.6, (1, nj, ni))
    f["swal2"] = rng.uniform(10.0, 25.0, (1, nj, ni))
    ust = 10.0 ** rng.uniform(-3.5, -1.3, (1, nj, ni))                      # calm to stormy
    f["ustar"] = ust
    f["ustar3"] = ust * ust * ust
    f["idkedt"] = 10.0 ** rng.uniform(-9.0, -6.0, (1, nj, ni))
    if ntr:
        f["trflx"] = rng.uniform(-1e-4, 1e-4, (ntr, nj, ni))
        f["trflx"][:, rng.uniform(size=(nj, ni)) < 0.02] = 30.0             # drives a tracer below zero in the top layer
    strong = rng.uniform(size=(1, nj, ni)) < 0.02
    f["salflx"] = np.where(strong, 2.0e3, f["salflx"])                      # drives the top layer's salinity below zero
    return f


VARIANTS = {
    # name: (options, strength of the heat fluxes)
    "default": (dict(rm0=1.2, rm5=0.0, niwgf=0.0, niwbf=0.35, mlrttp="constant"), 1.0),
    "momentum_entrainment+niw": (dict(rm0=1.2, rm5=4.0, niwgf=0.4, niwbf=0.35, mlrttp="variable"), 1.0),
    "limited_timescale_strong_forcing": (dict(rm0=2.0, rm5=0.0, niwgf=0.0, niwbf=0.35, mlrttp="limited"), 6.0),
}


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("cfg,nsteps", [("chan_s_tke", 3), ("box_s", 4), ("tri_s_tke", 3)])
def test_device_mxlayr_equals_the_real_module(cfg, nsteps, variant):
    _mxlayr_check(cfg, nsteps, variant)


@pytest.mark.parametrize("cfg,nsteps,variant,ntr", [("chan_s_tke", 3, "default", 9), ("tri_s_tke", 3, "momentum_entrainment+niw", 6),
                                                    ("chan_s_tke", 3, list(VARIANTS)[1], 13)])
def test_device_mxlayr_with_many_tracers(cfg, nsteps, variant, ntr):
    """more tracers than the column kernel loads at a time (four): the reference carries that many itself (ref_set_ntr); the extra
    ones are passive tracers with their own surface fluxes, a few values negative so that the copy-back's clamp books into trc_corr"""
    _mxlayr_check(cfg, nsteps, variant, ntr=ntr)


def test_full_size_channel_mxlayr_equals_the_real_module():
    """the same at BASELINE.json's channel size (208x512x53, ntr = 3; oracle/_ref/channel_tke_omp_xml, OpenMP)"""
    from test_xcheck_ale import run_with_big_stack
    run_with_big_stack(_mxlayr_check, "channel_tke", 2, "default")


def _mxlayr_check(cfg, nsteps, variant, ntr=None):
    from oracle.refblom import get_ref_backend, have_ref
    from blom_amd.gpu import BlomGpu
    lib = "channel_tke_omp_xml" if cfg == "channel_tke" else cfg.replace("_tke", "") + "_xml"
    if not have_ref(lib):
        pytest.skip(f"oracle/_ref/{lib}/libblomref.so not built")
    opts, strength = VARIANTS[variant]
    case = make_case(cfg, ntr=ntr)
    ref = get_ref_backend(lib, case.depth, ntr=ntr)
    assert ref.ntr == case.ntr
    kk = case.kdm
    gpu = BlomGpu(case.idm, case.jdm, kk, ref.ntr, ref.nreg, ref.masks)
    hostinit.init_state(gpu, case)
    assert gpu.step(0, nsteps) == nsteps
    six = hostinit.step_indices(nsteps, kk)
    # thin and massless layers under the mixed layer in part of the columns, so that entrainment runs through several layers
    # and the first physical layer differs from column to column (the cases' third layers are thick): every third column
    # loses its layer 3 (first physical layer 4), every third keeps 1 % of layers 3 and 4; the mass goes to the bottom layer
    nn, n = six[3], six[1]
    dp, kf = gpu.get("dp"), gpu.get("kfpla")
    jj_, ii_ = np.meshgrid(np.arange(dp.shape[1]), np.arange(dp.shape[2]), indexing="ij")
    pat = (jj_ + 2 * ii_) % 3
    a = pat == 0
    dp[nn + kk - 1][a] += dp[nn + 2][a]
    dp[nn + 2][a] = 0.0
    kf[n - 1][a & (kf[n - 1] == 3)] = 4
    b = pat == 1
    for k in (2, 3):
        dp[nn + kk - 1][b] += 0.99 * dp[nn + k][b]
        dp[nn + k][b] = dp[nn + k][b] - 0.99 * dp[nn + k][b]
    gpu.put("dp", dp)
    gpu.put("kfpla", kf)
    gpu.stage("mxlayr_tail", *six)                       # p, dpu, dpv of the edited thicknesses
    hostinit.init_state(ref, case)
    copy_state(gpu, ref, fields=STATE_FIELDS + GRID_FIELDS + INT_FIELDS)
    ref.ref.stage("mxlayr_init", *six)
    for nm in ("mtkeus", "mtkeni", "mtkebf", "mtkers", "mtkepe", "mtkeke", "pbrnda", "buoyfl"):     # the reference's initial patterns
        gpu.put(nm, ref.get(nm))
    zero2 = np.zeros((1, case.jdm + 8, case.idm + 8))
    for be in (ref, gpu):
        be.put("salt_corr", zero2)
        if ref.ntr:
            be.put("trc_corr", np.zeros((ref.ntr, case.jdm + 8, case.idm + 8)))
    common = dict(ce=0.06, tau_mlr=86400.0, lfmin=5.0e3, swamxd=200.0)
    for nm, v in {**common, **opts}.items():
        ref.ref.set(nm, v)
        gpu.set(nm, v)
    delt1 = 2.0 * case.params["baclin"]
    ref.ref.set("delt1", delt1)
    gpu.set("delt1", delt1)
    try:
        kf0 = gpu.get("kfpla").copy()
        assert len(np.unique(kf0[n - 1][4:-4, 4:-4][ref.masks["ip"][4:-4, 4:-4] > 0])) > 1
        for rep in range(2):
            f = forcing_fields(case, ref.ntr, 17 + rep, strength * (1.0 if rep == 0 else -0.7))
            for be in (ref, gpu):
                for nm in FORCING:
                    if nm in f:
                        be.put(nm, f[nm])
            dp0 = gpu.get("dp").copy()
            ref.ref.stage("mxlayr", *six)
            gpu.stage("mxlayr", *six)
            bad = diff_report(ref, gpu, fields=OUT)
            assert not bad, f"call {rep + 1}\n" + fmt_report(bad[:12])
        # the stage did something in both directions: the first physical layer moved up and down, layers changed, mass of every column kept
        wet = ref.masks["ip"][4:-4, 4:-4] > 0
        kf1 = gpu.get("kfpla")[n - 1][4:-4, 4:-4][wet]
        kfo = kf0[n - 1][4:-4, 4:-4][wet]
        assert (kf1 != kfo).any(), "kfpla did not change anywhere"
        d1 = gpu.get("dp")[nn:nn + kk][:, 4:-4, 4:-4]
        d0 = dp0[nn:nn + kk][:, 4:-4, 4:-4]
        assert np.abs(d1 - d0).max() > 0.0
        np.testing.assert_allclose(d1.sum(axis=0)[wet], d0.sum(axis=0)[wet], rtol=1e-12)
        pe = gpu.get("mtkepe")[0, 4:-4, 4:-4][wet]
        assert (pe != 0.0).any(), "no column entrained (mtkepe is zero everywhere)"
        assert (pe == 0.0).any(), "no column detrained"
        assert (gpu.get("pbrnda")[0, 4:-4, 4:-4][wet] > 0.0).any(), "no brine plume"
        assert (gpu.get("salt_corr")[0, 4:-4, 4:-4][wet] > 0.0).any(), "no column had its salinity clamped"
    finally:
        for nm, v in dict(rm0=1.2, rm5=0.0, niwgf=0.0, niwbf=0.35, mlrttp="constant").items():
            ref.ref.set(nm, v)
        gpu.close()


def test_mxlayr_refuses_what_the_reference_refuses():
    from blom_amd.gpu import BlomGpu, BlomGpuError
    case = make_case("chan_s")
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    hostinit.init_state(gpu, case)
    six = hostinit.step_indices(0, case.kdm)
    gpu.set("mlrttp", "sometimes")
    with pytest.raises(BlomGpuError, match="mlrttp = sometimes is unsupported!"):        # phy/mod_mxlayr.F90:203-212
        gpu.stage("mxlayr", *six)
    gpu.set("mlrttp", "constant")
    gpu.set("vcoord_type", "cntiso_hybrid")
    with pytest.raises(BlomGpuError, match="isopyc_bulkml"):
        gpu.stage("mxlayr", *six)
    gpu.close()


@pytest.mark.parametrize("cfg,nsteps", [("chan_s_tke", 3), ("box_s", 4), ("tri_s_tke", 3)])
def test_device_niw_ke_tendency_equals_the_real_module(cfg, nsteps):
    """The built part of difest_isobml (phy/mod_difest.F90:778-790): near-inertial wave energy input idkedt with its running-mean
    reservoirs, against the reference's real phy/mod_niw.F90 (no stand-in inside mod_niw; the library is an *_xml one), called
    three times in a row with alternating time levels as blom_step does."""
    from oracle.refblom import get_ref_backend, have_ref
    from blom_amd.gpu import BlomGpu
    lib = cfg.replace("_tke", "") + "_xml"
    if not have_ref(lib):
        pytest.skip(f"oracle/_ref/{lib}/libblomref.so not built")
    case = make_case(cfg)
    ref = get_ref_backend(lib, case.depth)
    kk = case.kdm
    gpu = BlomGpu(case.idm, case.jdm, kk, ref.ntr, ref.nreg, ref.masks)
    hostinit.init_state(gpu, case)
    assert gpu.step(0, nsteps) == nsteps
    hostinit.init_state(ref, case)
    copy_state(gpu, ref, fields=STATE_FIELDS + GRID_FIELDS + INT_FIELDS)
    ref.ref.stage("mxlayr_init", *hostinit.step_indices(nsteps, kk))
    fields = ["uml", "vml", "umlres", "vmlres", "idkedt"]
    for nm in fields:
        gpu.put(nm, ref.get(nm))
    delt1 = 2.0 * case.params["baclin"]
    ref.ref.set("delt1", delt1)
    gpu.set("delt1", delt1)
    amax = 0.0
    for it in range(3):
        six = hostinit.step_indices(nsteps + it, kk)
        ref.ref.stage("niw_ke_tendency", *six)
        gpu.stage("niw_ke_tendency", *six)
        bad = diff_report(ref, gpu, fields=fields + ["util1", "util2"])
        assert not bad, f"call {it + 1}\n" + fmt_report(bad[:10])
        a = gpu.get("idkedt")[0, 4:-4, 4:-4][ref.masks["ip"][4:-4, 4:-4] > 0]
        assert np.isfinite(a).all()
        amax = max(amax, float(a.max()))
    assert amax > 0.0
    gpu.close()
