"""CROSS-CHECK (not a pin): the step of the hybrid vertical coordinate, as far as it is built, device-resident against the
reference's own modules stage by stage (config 1 of BASELINE.json is the reference's tests/fuk95/limits:
cntiso_hybrid + cppm).

Sequence (blom_amd/stepper.py HYBRID_STAGES = the order of phy/mod_blom_step.F90:126-233): init_fluxes, tmsmt1,
ale_regrid_remap, cmnfld2, eddtra (eddtra_ale), advect, pbcor1, diffus, pgforc, momtum, cmnfld_bfsqi_ale, ale_forcing, ale_vdifft,
ale_vdiffm, updtrc, barotp, pbcor2, tmsmt2, cmnfld1.  Left out on BOTH sides because their modules need CVMix or forcing files:
difest_lateral_hybrid, difest_vertical_hybrid (diffusivities, non-local fractions and the boundary layer depth stay as uploaded),
thermf (surface fluxes as uploaded).  The reference side runs its real modules -- ale_regrid_remap, ale_forcing and cmnfld
against the stand-ins of oracle/xcheck (hence a cross-check), everything else from the plain build.  The device runs
blomgpu_step.  After every step all state arrays must agree bit for bit."""
import os

import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step, HYBRID_STAGES
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, GRID_FIELDS, INT_FIELDS
from test_xcheck_ale import OPTIONS, ale_init_once, set_device_ale_options, run_with_big_stack

pytestmark = pytest.mark.gpu
ALE_FIELDS = ["hbl_tf", "hml_tf1", "hml_tf", "OBLdepth", "kvisc_m", "kdiff_t", "kdiff_s", "t_ns_nonloc", "s_nb_nonloc", "t_sw_nonloc", "t_rs_nonloc", "s_br_nonloc",
              "s_rs_nonloc", "mu_nonloc", "mv_nonloc", "surflx", "sswflx", "surrlx", "salflx", "brnflx", "salrlx", "salt_corr",
              "trc_corr", "trflx", "swfc1", "swfc2", "swal1", "swal2", "mld", "mldl82", "dpml", "buoyfl", "sigint", "bfsqi",
              "bfsql", "bfsqf", "nslpx", "nslpy", "nnslpx", "nnslpy", "z", "dz", "told", "sold", "trcold"]
CHECK = STATE_FIELDS + ["hbl_tf", "hml_tf1", "hml_tf", "umflsm", "vmflsm", "utflsm", "vtflsm", "usflsm", "vsflsm", "sigint", "t_sw_nonloc", "s_br_nonloc", "buoyfl", "salt_corr", "trc_corr", "mld", "dpml", "bfsqi", "bfsqf",
                        "nslpx", "nslpy", "z", "dz"]


# work arrays of mod_utility that stages use as scratch (cppm leaves parts of them as they were)
SCRATCH = ("util1", "util2", "util3", "util4", "utotm", "vtotm", "uflux", "vflux", "uflux2", "vflux2", "uflux3", "vflux3")


@pytest.mark.parametrize("cfg,advmth,method,vcoord,nsteps", [
    # fuk95's layers are not isopycnic (a jet in uniformly thick layers): its densities lie outside the layers' targets and the
    # density-following coordinate would collapse the column into one layer -- it runs with pressure levels
    ("fuk95", "cppm", "nudge", "plevel", 4), ("fuk95", "cppm", "nudge", "cntiso_hybrid", 3), ("chan_s", "remap", "nudge", "cntiso_hybrid", 5),
    ("box_s", "cppm", "direct", "cntiso_hybrid", 4), ("tri_s", "remap", "direct", "cntiso_hybrid", 4),
    ("chan_s", "cppm", "direct", "plevel", 3)])
def test_hybrid_step_equals_the_reference_stage_sequence(cfg, advmth, method, vcoord, nsteps, tmp_path):
    _hybrid_step_check(cfg, advmth, method, vcoord, nsteps, tmp_path)


@pytest.mark.parametrize("cfg,advmth,method,vcoord,nsteps", [
    ("chan_s", "remap", "nudge", "cntiso_hybrid", 6), ("fuk95", "cppm", "nudge", "plevel", 4), ("box_s", "cppm", "direct", "cntiso_hybrid", 4),
    ("tri_s", "remap", "nudge", "cntiso_hybrid", 4)])
def test_hybrid_step_with_neutral_diffusion_equals_the_reference_stage_sequence(cfg, advmth, method, vcoord, nsteps, tmp_path):
    """ltedtp = 'neutral', the reference's default for the hybrid coordinate (cime_config/namelist_definition_blom.xml:1851-1862):
    ale_regrid_remap carries the neutral diffusion (phy/mod_ndiff.F90), cmnfld2 takes its slopes (cmnfld_nnslope_ale) for
    eddtra_ale, diffus is halo updates only"""
    _hybrid_step_check(cfg, advmth, method, vcoord, nsteps, tmp_path, neutral=True)


@pytest.mark.parametrize("cfg,advmth,method,nsteps,neutral", [("chan_s", "remap", "nudge", 5, True), ("tri_s", "remap", "direct", 4, False),
                                                              ("box_s", "cppm", "direct", 4, True)])
def test_hybrid_step_with_bod23_equals_the_reference_stage_sequence(cfg, advmth, method, nsteps, neutral, tmp_path):
    """mlrmth = 'bod23' (phy/mod_eddtra.F90:1058-1081, :1127-1154) inside the whole hybrid step: the running means hbl_tf, wpup_tf, hml_tf
    carry from step to step, the submesoscale transport enters advect"""
    _hybrid_step_check(cfg, advmth, method, "cntiso_hybrid", nsteps, tmp_path, neutral=neutral, mlrmth="bod23")


@pytest.mark.parametrize("advmth,method", [("remap", "nudge"), ("cppm", "direct")])
def test_full_size_channel_hybrid_step_equals_the_reference_stage_sequence(advmth, method, tmp_path):
    """Two whole hybrid steps at BASELINE.json's channel size (208x512x53, ntr = 3) with the &ALE_REGRID_REMAP group of the
    reference's tests/fuk95/limits (ppm 6/4): what `bench.py --config hybrid` times, against the reference's modules built with
    OpenMP (oracle/_ref/channel_tke_omp_xaln / _xale)."""
    run_with_big_stack(_hybrid_step_check, "channel_tke", advmth, method, "cntiso_hybrid", 2, tmp_path)


def test_full_size_channel_hybrid_step_with_neutral_diffusion_equals_the_reference_stage_sequence(tmp_path):
    """what `bench.py --config hybrid` times by default (ltedtp = 'neutral'), two steps at the channel's size"""
    run_with_big_stack(_hybrid_step_check, "channel_tke", "remap", "nudge", "cntiso_hybrid", 2, tmp_path, True)


def _hybrid_step_check(cfg, advmth, method, vcoord, nsteps, tmp_path, neutral=False, mlrmth="fox08"):
    import ctypes as C
    from oracle.refblom import get_ref_backend, have_ref
    from blom_amd.gpu import BlomGpu
    lib = (cfg + "_omp" if cfg.startswith("channel") else cfg) + ("_xaln" if method == "nudge" else "_xale")
    if not have_ref(lib):
        pytest.skip(f"oracle/_ref/{lib}/libblomref.so not built")
    case = make_case(cfg, advmth=advmth)
    ref = get_ref_backend(lib, case.depth)
    if not ref.has_field("mu_nonloc"):
        pytest.skip("reference library built before the hybrid step's fields were added to the harness")
    kk, nj, ni = case.kdm, case.jdm + 8, case.idm + 8
    gpu = BlomGpu(case.idm, case.jdm, kk, ref.ntr, ref.nreg, ref.masks)
    hostinit.init_state(ref, case)
    hostinit.init_state(gpu, case)
    copy_state(ref, gpu, fields=STATE_FIELDS + GRID_FIELDS + INT_FIELDS)      # the reference's patterns where nothing writes
    # what the parts of the step that are not built would produce: smooth synthetic fields, the same on both sides
    rng = np.random.default_rng(7)
    z = np.arange(kk + 1)[:, None, None] / kk
    f = {}
    for nm in ("kvisc_m", "kdiff_t", "kdiff_s"):
        f[nm] = 1e-5 + 10.0 ** rng.uniform(-3.5, -2.0, (1, nj, ni)) * np.exp(-((z - 0.1) / 0.15) ** 2)
    for nm in ("t_ns_nonloc", "s_nb_nonloc", "t_rs_nonloc", "s_rs_nonloc", "mu_nonloc", "mv_nonloc"):
        a = np.clip(1.0 - z / rng.uniform(0.1, 0.6, (1, nj, ni)), 0.0, 1.0) ** 2
        a[0] = 1.0
        f[nm] = a
    f["sswflx"] = -rng.uniform(0.0, 150.0, (1, nj, ni))
    f["surflx"] = f["sswflx"] + rng.uniform(-100.0, 100.0, (1, nj, ni))
    f["surrlx"] = rng.uniform(-10.0, 10.0, (1, nj, ni))
    f["brnflx"] = -rng.uniform(0.0, 1e-4, (1, nj, ni))
    f["salflx"] = f["brnflx"] + rng.uniform(-2e-3, 2e-3, (1, nj, ni))
    f["salrlx"] = rng.uniform(-5e-4, 5e-4, (1, nj, ni))
    f["swfc1"] = rng.uniform(0.4, 0.7, (1, nj, ni))
    f["swfc2"] = 1.0 - f["swfc1"]
    f["swal1"] = rng.uniform(0.5, 1.5, (1, nj, ni))
    f["swal2"] = rng.uniform(10.0, 20.0, (1, nj, ni))
    if ref.ntr:
        f["trflx"] = rng.uniform(-1e-6, 1e-6, (ref.ntr, nj, ni))
        f["trc_corr"] = np.zeros((ref.ntr, nj, ni))
    f["salt_corr"] = np.zeros((1, nj, ni))
    f["OBLdepth"] = 10.0 ** rng.uniform(0.8, 2.2, (1, nj, ni))      # boundary layer depth (difest_vertical_hybrid's, CVMix)
    if mlrmth == "bod23":                                           # ... and its friction / convective velocities cubed
        if not ref.has_field("wpup_tf"):
            pytest.skip("reference library built before bod23's fields were added to the harness")
        f["ustar3"] = (10.0 ** rng.uniform(-3.0, -1.5, (1, nj, ni))) ** 3
        f["wstar3"] = np.where(rng.random((1, nj, ni)) < 0.3, 0.0, 10.0 ** rng.uniform(-9.0, -5.0, (1, nj, ni)))
    pbot = float(np.max(ref.get("p")[kk][4:-4, 4:-4][ref.masks["ip"][4:-4, 4:-4] > 0]))
    plevel = 0.3 * pbot * (np.arange(kk) / kk) ** 1.3
    for nm, a in f.items():
        ref.put(nm, a)
    for nm in ("ustar3", "wstar3"):
        if nm in f:
            gpu.put(nm, f[nm])
    for nm in ALE_FIELDS:                     # the reference's initial patterns (spval) where nothing writes
        if ref.has_field(nm) and gpu.has_field(nm):
            gpu.put(nm, ref.get(nm))
    ierr = C.c_int(0)
    v = np.ascontiguousarray(plevel, dtype=np.float64)
    ref.ref.lib.ref_set_vec(b"plevel", v.ctypes.data_as(C.c_void_p), C.c_int(kk), C.byref(ierr))
    assert ierr.value == 0
    tag = 2 if vcoord == "cntiso_hybrid" else 3
    ref.ref.set("vcoord_tag", 2)                # the reader resolves regrid_method only for 'cntiso_hybrid'
    ref.ref.set("swamxd", 200.0)
    ref.ref.set("brine_mlbase_frac", 0.4)
    o = dict(OPTIONS[cfg], regrid_method=method)       # one option set per reference library and process (tests/test_xcheck_ale.py)
    gpu.set("vcoord_type", vcoord)
    set_device_ale_options(gpu, o)
    gpu.set_vector("plevel", plevel)
    gpu.set("swamxd", 200.0)
    gpu.set("brine_mlbase_frac", 0.4)
    six0 = hostinit.step_indices(0, kk)
    ale_init_once(ref, lib, o, six0, tmp_path)
    ref.ref.stage("eddtra_init_" + mlrmth, *six0)                 # inivar_eddtra + init_eddtra (mlrmth = 'fox08' is the default)
    for nm in ("hbl_tf", "hml_tf1", "hml_tf") + (("wpup_tf",) if mlrmth == "bod23" else ()):
        gpu.put(nm, ref.get(nm))
    ref.ref.set("eitmth", "gm")
    gpu.set("eitmth", "gm")
    gpu.set("mlrmth", mlrmth)
    try:
        ref.ref.set("vcoord_tag", tag)
        for be in (ref.ref, gpu):
            be.set("ltedtp_opt", 2 if neutral else 1)
            be.set("ndiff_surface_align", 1)
        # blom_init's cmnfld1 (phy/mod_blom_init.F90): the mixed layer depth the first ale_forcing reads
        ref.ref.stage("cmnfld1", *hostinit.init_indices(0, kk))
        gpu.stage("cmnfld1", *hostinit.init_indices(0, kk))
        nr = ng = 0
        for _ in range(nsteps):
            if os.environ.get("BLOM_STAGEWISE") == "1":             # development aid: both sides stage by stage, compared after each
                six = hostinit.step_indices(nr, kk)
                for be in (ref, gpu):
                    be.set("nstep", nr + 1)
                for st in HYBRID_STAGES:
                    ref.stage(st, *six)
                    gpu.stage(st, *six)
                    bad = diff_report(ref, gpu, fields=[nm for nm in CHECK if nm not in SCRATCH])
                    assert not bad, f"step {nr + 1} after {st}\n" + fmt_report(bad[:12])
                for be in (ref, gpu):
                    be.set("delt1", 2 * case.params["baclin"])
                nr += 1
                ng += 1
                continue
            nr = dyncore_step(ref, nr, case.params["baclin"], stages=HYBRID_STAGES)
            ng = gpu.step(ng, 1)
            # the OpenMP build of the reference (channel size) makes utotn, vtotn firstprivate in momtum's layer loop
            # (phy/mod_momtum.F90:342-350), so the module arrays keep whatever they held outside the interior -- in the serial
            # build the last layer's values, which is what the device leaves there.  Interior only for that build.
            omp = cfg.startswith("channel")
            chk = CHECK + (["nnslpx", "nnslpy", "utflld", "usflld", "vtflld", "vsflld"] if neutral else []) + (["wpup_tf"] if mlrmth == "bod23" else [])
            bad = diff_report(ref, gpu, fields=[nm for nm in chk if nm not in SCRATCH and not (omp and nm in ("utotn", "vtotn"))])
            assert not bad, f"step {nr}\n" + fmt_report(bad[:12])
            if omp:
                for nm in ("utotn", "vtotn"):
                    assert np.array_equal(ref.get(nm)[:, 4:-4, 4:-4], gpu.get(nm)[:, 4:-4, 4:-4]), nm
        wu = (ref.masks["iu"][4:-4, 4:-4] > 0)[None]
        uu = gpu.get("u")[:, 4:-4, 4:-4]
        assert np.isfinite(uu[np.broadcast_to(wu, uu.shape)]).all() and np.abs(uu[np.broadcast_to(wu, uu.shape)]).max() > 0.0
        # eddtra_ale took part: Gent-McWilliams mass fluxes are there (submesoscale ones only where the mixed layer spans more than
        # one layer: tests/test_xcheck_eddtra_ale.py has those)
        for nm in ("umfltd", "umflsm"):
            a = gpu.get(nm)[:, 4:-4, 4:-4]
            a = a[np.broadcast_to(wu, a.shape)]
            assert np.isfinite(a).all() and (nm == "umflsm" or np.abs(a).max() > 0.0), nm
        if neutral:
            mq = hostinit.step_indices(ng - 1, kk)[2]
            assert np.abs(gpu.get("utflld")[mq:mq + kk]).max() > 0.0 and np.abs(gpu.get("nnslpx")).max() > 0.0
    finally:
        ref.ref.set("vcoord_tag", 1)
        ref.ref.set("ltedtp_opt", 1)
        if mlrmth != "fox08":
            ref.ref.stage("eddtra_init_fox08", *six0)           # (the cached backend goes back to the default)
        gpu.close()
