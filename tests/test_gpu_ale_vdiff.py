"""ale_vdifft and ale_vdiffm (phy/mod_ale_vdiff.F90:50-374) on the device against the reference's own
module -- PINNED: mod_ale_vdiff builds from the reference's sources on top of the plain module set, no stand-in involved
(oracle/Makefile, libraries *_vdf).

The two stages take the vertical diffusivities, the non-local transport fractions and the surface fluxes as inputs (their
producers, difest_vertical_hybrid and thermf, need CVMix / forcing files and are not built): here they are synthetic --
diffusivities between 1e-5 and 1e-1 m2 s-1 with a boundary-layer bulge, fractions that fall from one at the surface to zero
at depth, fluxes of both signs, in a few columns strong enough to drive the salinity and a tracer negative so that the
clamping and the correction accumulators (salt_corr, trc_corr) act.  State: the isopycnic state after a few steps of the
dynamical core.  Every array the stages write must agree bit for bit: temp, saln, sigma, trc, salt_corr, trc_corr, u, v,
and the halo of kvisc_m."""
import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, GRID_FIELDS, INT_FIELDS

pytestmark = pytest.mark.gpu

INPUTS_3D = ["kvisc_m", "kdiff_t", "kdiff_s", "t_ns_nonloc", "s_nb_nonloc", "t_sw_nonloc", "t_rs_nonloc", "s_br_nonloc",
             "s_rs_nonloc"]
INPUTS_2D = ["surflx", "sswflx", "surrlx", "salflx", "brnflx", "salrlx"]
OUT = ["temp", "saln", "sigma", "trc", "salt_corr", "trc_corr", "u", "v", "kvisc_m", "dp", "dpu", "dpv"]


def _inputs(case, ntr, seed):
    rng = np.random.default_rng(seed)
    kk, nj, ni = case.kdm, case.jdm + 8, case.idm + 8
    z = np.arange(kk + 1)[:, None, None] / kk
    f = {}
    for nm in ("kvisc_m", "kdiff_t", "kdiff_s"):
        bulge = 10.0 ** rng.uniform(-3.0, -1.0, (1, nj, ni)) * np.exp(-((z - 0.15) / 0.12) ** 2)
        f[nm] = 10.0 ** rng.uniform(-5.0, -3.5, (kk + 1, nj, ni)) + bulge
    for nm in ("t_ns_nonloc", "s_nb_nonloc", "t_sw_nonloc", "t_rs_nonloc", "s_br_nonloc", "s_rs_nonloc"):
        depth = rng.uniform(0.1, 0.9, (1, nj, ni))
        a = np.clip(1.0 - z / depth, 0.0, 1.0) ** rng.uniform(1.0, 3.0, (1, nj, ni))
        a[0] = 1.0
        f[nm] = a
    f["sswflx"] = -rng.uniform(0.0, 250.0, (1, nj, ni))
    f["surflx"] = f["sswflx"] + rng.uniform(-300.0, 300.0, (1, nj, ni))
    f["surrlx"] = rng.uniform(-40.0, 40.0, (1, nj, ni))
    f["brnflx"] = -rng.uniform(0.0, 1e-3, (1, nj, ni))
    f["salflx"] = f["brnflx"] + rng.uniform(-2e-2, 2e-2, (1, nj, ni))
    f["salrlx"] = rng.uniform(-5e-3, 5e-3, (1, nj, ni))
    strong = rng.uniform(size=(1, nj, ni)) < 0.03
    f["salflx"] = np.where(strong, 5.0e3, f["salflx"])           # drives the top layer's salinity below zero
    trflx = rng.uniform(-1e-3, 1e-3, (max(ntr, 1), nj, ni))
    trflx[:, strong[0]] = 50.0
    f["trflx"] = trflx[:ntr] if ntr else trflx[:0]
    return f


@pytest.mark.parametrize("cfg,nsteps,seed,ntr", [("chan_s_tke", 4, 1, None), ("box_s", 4, 2, None), ("fuk95", 3, 3, None), ("tri_s_tke", 3, 4, None),
                                                 ("chan_s_tke", 1, 5, None),
                                                 # more tracers than one pass of the fused kernel carries (4), and counts that
                                                 # leave 1, 2 tracers to the last pass: the reference carries them itself
                                                 ("chan_s_tke", 3, 6, 9), ("tri_s_tke", 3, 7, 6), ("box_s", 3, 8, 4), ("chan_s_tke", 3, 9, 5)])
def test_ale_vdiff_equals_the_reference(cfg, nsteps, seed, ntr):
    _ale_vdiff_check(cfg, nsteps, seed, ntr)


def test_full_size_channel_ale_vdiff_equals_the_reference():
    """PINNED at BASELINE.json's channel size (208x512x53, ntr = 3): oracle/_ref/channel_tke_omp_vdf is the reference's own
    phy/mod_ale_vdiff.F90 on the plain build, no stand-in."""
    from test_xcheck_ale import run_with_big_stack
    run_with_big_stack(_ale_vdiff_check, "channel_tke", 2, 21, None)


def _ale_vdiff_check(cfg, nsteps, seed, ntr):
    from oracle.refblom import get_ref_backend, have_ref
    from blom_amd.gpu import BlomGpu
    lib = "channel_tke_omp_vdf" if cfg == "channel_tke" else cfg.replace("_tke", "") + "_vdf"
    if not have_ref(lib):
        pytest.skip(f"oracle/_ref/{lib}/libblomref.so not built")
    case = make_case(cfg, ntr=ntr) if ntr else make_case(cfg)
    ref = get_ref_backend(lib, case.depth, ntr=ntr)
    assert ref.ntr == case.ntr
    kk = case.kdm
    gpu = BlomGpu(case.idm, case.jdm, kk, ref.ntr, ref.nreg, ref.masks)
    hostinit.init_state(gpu, case)
    assert gpu.step(0, nsteps) == nsteps
    hostinit.init_state(ref, case)
    copy_state(gpu, ref, fields=STATE_FIELDS + GRID_FIELDS + INT_FIELDS)
    f = _inputs(case, ref.ntr, seed)
    zero2 = np.zeros((1, case.jdm + 8, case.idm + 8))
    for be in (ref, gpu):
        for nm, a in f.items():
            if a.shape[0]:
                be.put(nm, a)
        be.put("salt_corr", zero2)
        if ref.ntr:
            be.put("trc_corr", np.zeros((ref.ntr, case.jdm + 8, case.idm + 8)))
    delt1 = 2.0 * case.params["baclin"]
    ref.ref.set("delt1", delt1)
    gpu.set("delt1", delt1)
    ref.ref.set("vcoord_tag", 2)
    gpu.set("vcoord_type", "cntiso_hybrid")
    six = hostinit.step_indices(nsteps, kk)
    try:
        t0, s0 = gpu.get("temp").copy(), gpu.get("saln").copy()
        for st in ("ale_vdifft", "ale_vdiffm"):
            ref.ref.stage(st, *six)
            gpu.stage(st, *six)
            bad = diff_report(ref, gpu, fields=OUT)
            assert not bad, st + "\n" + fmt_report(bad[:10])
        # the stages did something, and the clamps acted
        assert np.abs(gpu.get("temp") - t0)[:, 4:-4, 4:-4].max() > 1e-6
        assert np.abs(gpu.get("saln") - s0)[:, 4:-4, 4:-4].max() > 1e-6
        wet = ref.masks["ip"][4:-4, 4:-4] > 0
        assert (gpu.get("salt_corr")[0, 4:-4, 4:-4][wet] > 0).any(), "no column had its salinity clamped"
        assert np.isfinite(gpu.get("u")[:, 4:-4, 4:-4]).all()
    finally:
        ref.ref.set("vcoord_tag", 1)
        gpu.close()


def test_ale_vdiff_is_refused_for_the_isopycnic_coordinate():
    from blom_amd.gpu import BlomGpu, BlomGpuError
    case = make_case("chan_s")
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    hostinit.init_state(gpu, case)
    six = hostinit.step_indices(0, case.kdm)
    for st in ("ale_vdifft", "ale_vdiffm"):
        with pytest.raises(BlomGpuError, match="isopyc_bulkml"):
            gpu.stage(st, *six)
    gpu.close()


@pytest.mark.parametrize("cfg,nsteps,seed", [("chan_s_tke", 3, 11), ("box_s", 3, 12), ("fuk95", 2, 13), ("tri_s_tke", 3, 14)])
def test_momtum_with_the_wind_stress_of_the_other_coordinates(cfg, nsteps, seed):
    """momtum for vcoord_type /= 'isopyc_bulkml' (phy/mod_momtum.F90:937-946, :1100-1109): the wind stress enters every layer through
    the non-local fractions mu_nonloc, mv_nonloc instead of the top layer alone.  PINNED: mod_momtum is part of the plain
    reference build; the fractions (produced by difest_vertical_hybrid, not built) are synthetic."""
    from oracle.refblom import get_ref_backend, have_ref
    from blom_amd.gpu import BlomGpu
    lib = cfg.replace("_tke", "") + "_vdf"
    if not have_ref(lib):
        pytest.skip(f"oracle/_ref/{lib}/libblomref.so not built")
    case = make_case(cfg)
    ref = get_ref_backend(lib, case.depth)
    if not ref.has_field("mu_nonloc"):
        pytest.skip("reference library built before mu_nonloc was added to the harness")
    kk, nj, ni = case.kdm, case.jdm + 8, case.idm + 8
    gpu = BlomGpu(case.idm, case.jdm, kk, ref.ntr, ref.nreg, ref.masks)
    hostinit.init_state(gpu, case)
    assert gpu.step(0, nsteps) == nsteps
    hostinit.init_state(ref, case)
    copy_state(gpu, ref, fields=STATE_FIELDS + GRID_FIELDS + INT_FIELDS)
    rng = np.random.default_rng(seed)
    z = np.arange(kk + 1)[:, None, None] / kk
    f = {}
    for nm in ("mu_nonloc", "mv_nonloc"):
        a = np.clip(1.0 - z / rng.uniform(0.1, 0.9, (1, nj, ni)), 0.0, 1.0) ** rng.uniform(1.0, 3.0, (1, nj, ni))
        a[0] = 1.0
        f[nm] = a
    f["taux"] = rng.uniform(-0.3, 0.3, (1, nj, ni))
    f["tauy"] = rng.uniform(-0.3, 0.3, (1, nj, ni))
    for be in (ref, gpu):
        for nm, a in f.items():
            be.put(nm, a)
    ref.ref.set("vcoord_tag", 2)
    gpu.set("vcoord_type", "cntiso_hybrid")
    delt1 = 2.0 * case.params["baclin"]
    ref.ref.set("delt1", delt1)
    gpu.set("delt1", delt1)
    six = hostinit.step_indices(nsteps, kk)
    try:
        u0 = gpu.get("u").copy()
        ref.ref.stage("momtum", *six)
        gpu.stage("momtum", *six)
        bad = diff_report(ref, gpu, fields=["u", "v", "p", "pu", "pv", "utotn", "vtotn", "ubflxs_p", "vbflxs_p"])
        assert not bad, fmt_report(bad[:10])
        assert np.abs(gpu.get("u") - u0)[:, 4:-4, 4:-4].max() > 0.0
    finally:
        ref.ref.set("vcoord_tag", 1)
        gpu.close()
