"""CROSS-CHECK (not a pin) of the device's ale_regrid_remap against the reference's REAL phy/mod_ale_regrid_remap.F90.

SURVEY.md 8 row f3, first piece: the regrid + remap step of the vertical coordinates other than isopyc_bulkml, built on the
device hor3map (blom_amd/csrc/stage_ale.hip).  The reference module imports the z-level diagnostic flags and arrays of the
netCDF-bound mod_dia, so it is compiled against oracle/xcheck/mod_dia_standin.F90 (all flags zero: no diagnostic requested)
in the *_xale builds of oracle/Makefile -- hence a cross-check, not a pin.  What is compared: after a few steps
of the isopycnic sequence on the device (so that the layers are uneven, some massless, and the velocities non-zero) the
state goes to the reference, both run ale_regrid_remap with vcoord_type = 'plevel' or 'cntiso_hybrid' (regrid_method =
'direct') and the options of the namelist group
&ALE_REGRID_REMAP -- the reference reads them from a file `limits`, the device takes them through blomgpu_set_str /
set_int -- and every array the stage writes must agree bit for bit: dp, T, S, sigma, tracers, u, v, dpu, dpv, dpuold, dpvold,
p, pu, pv, and the diagnostic interface densities sigint."""
import os

import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, GRID_FIELDS, INT_FIELDS

pytestmark = pytest.mark.gpu

# one reference library per configuration, and its ALE structures can be initialised once per process: one option set each
OPTIONS = {
    # the group of the reference's own tests/fuk95/limits
    "fuk95": dict(reconstruction_method="ppm", upper_bndr_ord=6, lower_bndr_ord=4, tracer_limiting="non_oscillatory",
                  velocity_limiting="non_oscillatory", tracer_pc_upper_bndr=True, tracer_pc_lower_bndr=False,
                  velocity_pc_upper_bndr=True, velocity_pc_lower_bndr=False),
    "chan_s": dict(reconstruction_method="pqm", upper_bndr_ord=4, lower_bndr_ord=4, tracer_limiting="monotonic",
                   velocity_limiting="monotonic", tracer_pc_upper_bndr=False, tracer_pc_lower_bndr=True,
                   velocity_pc_upper_bndr=False, velocity_pc_lower_bndr=False),
    "box_s": dict(reconstruction_method="plm", upper_bndr_ord=2, lower_bndr_ord=2, tracer_limiting="non_oscillatory",
                  velocity_limiting="monotonic", tracer_pc_upper_bndr=True, tracer_pc_lower_bndr=True,
                  velocity_pc_upper_bndr=True, velocity_pc_lower_bndr=True),
    "tri_s": dict(reconstruction_method="ppm", upper_bndr_ord=3, lower_bndr_ord=2, tracer_limiting="non_oscillatory",
                  velocity_limiting="non_oscillatory", tracer_pc_upper_bndr=True, tracer_pc_lower_bndr=False,
                  velocity_pc_upper_bndr=True, velocity_pc_lower_bndr=False),
}
# BASELINE.json's channel at full size (208x512x53, ntr = 3): the group of tests/fuk95/limits (ppm, 6/4)
OPTIONS["channel_tke"] = dict(OPTIONS["fuk95"])
OUT = ["dp", "temp", "saln", "sigma", "trc", "u", "v", "dpu", "dpv", "dpuold", "dpvold", "p", "pu", "pv"]


def ale_init_once(ref, lib, o, six, tmp_path):
    """readnml_ale_regrid_remap + init_ale_regrid_remap of a reference library, once per process (the flag lives on the shared
    backend object: tests/test_xcheck_hybrid_step.py uses the same libraries with the same option sets)"""
    if getattr(ref, "_ale_options", None) is not None:
        assert ref._ale_options == o, "the reference library's ALE structures were initialised with other options"
        return
    ref.ref.set("vcoord_tag", 2)               # the reader resolves regrid_method only for 'cntiso_hybrid' (:1323)
    ref.ref.set("ltedtp_opt", 2)               # the structures get the index range neutral diffusion needs (:1384-1390): a superset
    (tmp_path / "limits").write_text(_limits_text(o))
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        ref.ref.stage("ale_init", *six)
    finally:
        os.chdir(cwd)
        ref.ref.set("ltedtp_opt", 1)
    ref._ale_options = dict(o)


def set_device_ale_options(gpu, o):
    gpu.set("ale_regrid_method", o["regrid_method"])
    gpu.set("ale_reconstruction_method", o["reconstruction_method"])
    gpu.set("ale_tracer_limiting", o["tracer_limiting"])
    gpu.set("ale_velocity_limiting", o["velocity_limiting"])
    for nm in ("upper_bndr_ord", "lower_bndr_ord"):
        gpu.set("ale_" + nm, int(o[nm]))
    for nm in ("tracer_pc_upper_bndr", "tracer_pc_lower_bndr", "velocity_pc_upper_bndr", "velocity_pc_lower_bndr"):
        gpu.set("ale_" + nm, 1 if o[nm] else 0)


def _limits_text(o):
    f = lambda b: ".true." if b else ".false."
    return (" &ALE_REGRID_REMAP\n"
            f"  RECONSTRUCTION_METHOD  = '{o['reconstruction_method']}'\n  UPPER_BNDR_ORD = {o['upper_bndr_ord']}\n"
            f"  LOWER_BNDR_ORD = {o['lower_bndr_ord']}\n  DENSITY_LIMITING = 'monotonic'\n"
            f"  TRACER_LIMITING = '{o['tracer_limiting']}'\n  VELOCITY_LIMITING = '{o['velocity_limiting']}'\n"
            f"  TRACER_PC_UPPER_BNDR = {f(o['tracer_pc_upper_bndr'])}\n  TRACER_PC_LOWER_BNDR = {f(o['tracer_pc_lower_bndr'])}\n"
            f"  VELOCITY_PC_UPPER_BNDR = {f(o['velocity_pc_upper_bndr'])}\n  VELOCITY_PC_LOWER_BNDR = {f(o['velocity_pc_lower_bndr'])}\n"
            f"  REGRID_METHOD = '{o.get('regrid_method', 'direct')}'\n /\n")


@pytest.mark.parametrize("cfg,nsteps,spread,vcoord", [
    ("fuk95", 3, 1.0, "plevel"), ("chan_s", 4, 1.0, "plevel"), ("chan_s", 2, 0.35, "plevel"), ("box_s", 4, 1.3, "plevel"),
    ("tri_s", 3, 1.0, "plevel"),
    # regrid_method = 'direct': the interfaces follow the layers' target densities
    ("fuk95", 3, 0.6, "cntiso_hybrid"), ("chan_s", 4, 0.5, "cntiso_hybrid"), ("chan_s", 2, 0.1, "cntiso_hybrid"),
    ("box_s", 4, 0.8, "cntiso_hybrid"), ("tri_s", 3, 0.3, "cntiso_hybrid"),
    # regrid_method = 'nudge' (the reference's default; libraries *_xaln), with the lateral smoothing of the interfaces
    ("fuk95", 3, 0.6, "nudge"), ("chan_s", 4, 0.5, "nudge"), ("chan_s", 2, 0.1, "nudge"), ("box_s", 4, 0.8, "nudge"),
    ("tri_s", 3, 0.3, "nudge"), ("fuk95", 3, 1.0, "nudge+plevel")])
def test_device_ale_regrid_remap_equals_the_real_module(cfg, nsteps, spread, vcoord, tmp_path):
    _ale_regrid_remap_check(cfg, nsteps, spread, vcoord, tmp_path)


def run_with_big_stack(fn, *args):
    """the channel-sized reference keeps its stage-local work arrays on the stack (BLOM runs with ulimit -s unlimited) and is
    built with its OpenMP directives on"""
    import threading
    res = {}

    def body():
        try:
            fn(*args)
            res["ok"] = True
        except BaseException as e:          # noqa: BLE001 -- handed to the caller's thread
            res["err"] = e
    os.environ["OMP_NUM_THREADS"] = str(min(16, os.cpu_count() or 1))
    os.environ["OMP_STACKSIZE"] = "1G"
    threading.stack_size(2 << 30)
    th = threading.Thread(target=body)
    th.start()
    th.join()
    threading.stack_size(0)
    if "err" in res:
        raise res["err"]
    assert res.get("ok"), "the comparison did not complete"


@pytest.mark.parametrize("vcoord", ["nudge", "cntiso_hybrid"])
def test_full_size_channel_ale_regrid_remap_equals_the_real_module(vcoord, tmp_path):
    """The same comparison at BASELINE.json's channel size (208x512x53, ntr = 3) with the options of the reference's
    tests/fuk95/limits (ppm, boundary orders 6/4, non-oscillatory limiting; regrid_method 'nudge' -- the default -- and
    'direct'): the loads-ahead and active-column logic of stage_ale.hip / hor3map.hip at the column count and depth they are
    timed on.  Reference: oracle/_ref/channel_tke_omp_xaln / _xale (real phy/mod_ale_regrid_remap.F90, OpenMP)."""
    run_with_big_stack(_ale_regrid_remap_check, "channel_tke", 3, 0.5, vcoord, tmp_path)


NDIFF_OUT = ["utflld", "usflld", "vtflld", "vsflld", "utflx", "usflx", "vtflx", "vsflx", "nslpx", "nslpy"]


@pytest.mark.parametrize("cfg,nsteps,spread,vcoord,align", [
    ("chan_s", 4, 0.5, "nudge", 1), ("chan_s", 2, 0.1, "nudge", 0), ("fuk95", 3, 0.6, "nudge", 1), ("box_s", 4, 0.8, "nudge", 0),
    ("tri_s", 3, 0.3, "nudge", 1), ("fuk95", 3, 1.0, "nudge+plevel", 1), ("chan_s", 4, 0.5, "cntiso_hybrid", 1),
    ("box_s", 4, 0.8, "cntiso_hybrid", 1), ("tri_s", 3, 1.0, "plevel", 0)])
def test_device_neutral_diffusion_equals_the_real_module(cfg, nsteps, spread, vcoord, align, tmp_path):
    """ltedtp = 'neutral' (the reference's default with vcoord_type = 'cntiso_hybrid'): ale_regrid_remap with phy/mod_ndiff.F90's
    neutral diffusion between the regridding and the remapping -- the fluxes per face and layer, the neutral slopes, and the
    tracers with the flux convergence applied, bit for bit; with and without the alignment with the surface in the mixed layer"""
    _ale_regrid_remap_check(cfg, nsteps, spread, vcoord, tmp_path, ndiff=align)


def test_full_size_channel_neutral_diffusion_equals_the_real_module(tmp_path):
    """neutral diffusion at BASELINE.json's channel size (208x512x53, ntr = 3, ppm 6/4, nudge): the cell-centred flux kernel's
    scratch planes and face ranges at the size it is timed on (oracle/_ref/channel_tke_omp_xaln)"""
    run_with_big_stack(_ale_regrid_remap_check, "channel_tke", 3, 0.5, "nudge", tmp_path, 1)


def _ale_regrid_remap_check(cfg, nsteps, spread, vcoord, tmp_path, ndiff=None):
    import ctypes as C
    from oracle.refblom import get_ref_backend, have_ref
    from blom_amd.gpu import BlomGpu
    method = "nudge" if vcoord.startswith("nudge") else "direct"
    lib = (cfg + "_omp" if cfg.startswith("channel") else cfg) + ("_xaln" if method == "nudge" else "_xale")
    vcoord = "plevel" if vcoord.endswith("plevel") else "cntiso_hybrid"
    if not have_ref(lib):
        pytest.skip(f"oracle/_ref/{lib}/libblomref.so not built")
    o = dict(OPTIONS[cfg], regrid_method=method)
    case = make_case(cfg)
    ref = get_ref_backend(lib, case.depth)
    kk = case.kdm
    # device: the isopycnic sequence for a few steps
    gpu = BlomGpu(case.idm, case.jdm, kk, ref.ntr, ref.nreg, ref.masks)
    hostinit.init_state(gpu, case)
    assert gpu.step(0, nsteps) == nsteps
    # pressure levels: from the surface to `spread` times the deepest bottom pressure (levels below the bottom collapse there)
    pbot = float(np.nanmax(gpu.get("p")[kk][4:-4, 4:-4] * (ref.masks["ip"][4:-4, 4:-4] > 0)))
    plevel = spread * pbot * (np.arange(kk) / kk) ** 1.3
    # reference: same state, vcoord_type = 'plevel', its structures from the namelist group
    hostinit.init_state(ref, case)
    copy_state(gpu, ref, fields=STATE_FIELDS + GRID_FIELDS + INT_FIELDS)
    tag = 3 if vcoord == "plevel" else 2
    ierr = C.c_int(0)
    v = np.ascontiguousarray(plevel, dtype=np.float64)
    ref.ref.lib.ref_set_vec(b"plevel", v.ctypes.data_as(C.c_void_p), C.c_int(kk), C.byref(ierr))
    assert ierr.value == 0
    six = hostinit.step_indices(nsteps, kk)
    ale_init_once(ref, lib, o, six, tmp_path)
    try:
        ref.ref.set("vcoord_tag", tag)
        delt1 = 2.0 * case.params["baclin"]            # the nudging factor and the smoothing's diffusion read it
        ref.ref.set("delt1", delt1)
        gpu.set("delt1", delt1)
        pre_sigint = ref.get("sigint").copy()
        if ndiff is not None:
            nj, ni = case.jdm + 8, case.idm + 8
            yy, xx = np.meshgrid(np.linspace(0.0, 1.0, nj), np.linspace(0.0, 2 * np.pi, ni), indexing="ij")
            difiso = 800.0 * (1.0 + 0.5 * np.sin(xx) * yy)[None] * np.linspace(1.0, 0.3, kk)[:, None, None]
            dpml = 9806.0 * (20.0 + 60.0 * yy * (1.0 + 0.5 * np.cos(2 * xx)))[None]
            for be in (ref, gpu):
                be.put("difiso", difiso)
                be.put("dpml", dpml)
            ref.ref.set("ltedtp_opt", 2)
            ref.ref.set("ndiff_surface_align", int(ndiff))
            gpu.set("ltedtp_opt", 2)
            gpu.set("ndiff_surface_align", int(ndiff))
            copy_state(gpu, ref, fields=NDIFF_OUT)
        ref.ref.stage("ale_regrid_remap", *six)
        # device
        gpu.set("vcoord_type", vcoord)
        set_device_ale_options(gpu, o)
        gpu.set_vector("plevel", plevel)
        if tag == 2:
            gpu.put("sigint", pre_sigint)          # the reference's initial pattern (spval) where the stage does not write
        before = gpu.get("dp").copy()
        gpu.stage("ale_regrid_remap", *six)
        bad = diff_report(ref, gpu, fields=OUT + (["sigint"] if tag == 2 else []) + (NDIFF_OUT if ndiff is not None else []))
        assert not bad, fmt_report(bad[:10])
        if ndiff is not None:
            mmq = six[2]
            assert np.abs(gpu.get("utflld")[mmq:mmq + kk]).max() > 0.0 and np.abs(gpu.get("nslpx")).max() > 0.0
        # the stage did something: layers moved, and mass, heat and salt of every column are what they were
        after = gpu.get("dp")
        nn = six[3]
        assert np.abs(after[nn:nn + kk] - before[nn:nn + kk])[:, 4:-4, 4:-4].max() > 0.0
        wet = ref.masks["ip"][4:-4, 4:-4] > 0
        col0 = before[nn:nn + kk][:, 4:-4, 4:-4].sum(axis=0)[wet]
        col1 = after[nn:nn + kk][:, 4:-4, 4:-4].sum(axis=0)[wet]
        np.testing.assert_allclose(col1, col0, rtol=1e-12)
    finally:
        ref.ref.set("vcoord_tag", 1)
        ref.ref.set("ltedtp_opt", 1)
        gpu.close()


def test_other_coordinates_fail_loudly():
    from blom_amd.gpu import BlomGpu, BlomGpuError
    case = make_case("chan_s")
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    hostinit.init_state(gpu, case)
    six = hostinit.step_indices(0, case.kdm)
    with pytest.raises(BlomGpuError, match="isopyc_bulkml"):
        gpu.stage("ale_regrid_remap", *six)
    gpu.set("vcoord_type", "plevel")
    with pytest.raises(BlomGpuError, match="plevel"):
        gpu.stage("ale_regrid_remap", *six)                 # no pressure levels given
    with pytest.raises(BlomGpuError, match="unsupported"):
        gpu.set("ale_reconstruction_method", "spline")
    gpu.close()


@pytest.mark.parametrize("cfg,method,ntr,neutral", [("chan_s", "nudge", 9, False), ("tri_s", "direct", 7, False), ("box_s", "nudge", 12, False),
                                                    ("chan_s", "nudge", 9, True), ("tri_s", "direct", 11, True)])
def test_more_tracers_than_one_engine_batch(cfg, method, ntr, neutral):
    """ale_regrid_remap with more than the eight fields (T, S + tracers) one batch of the engine's *_many calls carries: the
    further batches reconstruct and remap on their own.  Pinned by reduction: tracers that are copies of the case's one tracer
    must each come out as that tracer does in the run with one tracer (which the cases above check against the reference),
    and nothing else may change.  neutral: with the neutral diffusion, which keeps the coefficients of every batch and carries
    one flux per field in its records."""
    from blom_amd.gpu import BlomGpu
    nsteps = 3
    case1, caseN = make_case(cfg), make_case(cfg, ntr=ntr)
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case1.depth, case1.idm, case1.jdm, arctic=case1.nreg == 2)
    masks = dict(ip=ip, iu=iu, iv=iv, iq=iq)
    kk = case1.kdm
    out = {}
    for tag, case in (("one", case1), ("many", caseN)):
        gpu = BlomGpu(case.idm, case.jdm, kk, case.ntr, nreg, masks)
        hostinit.init_state(gpu, case)
        assert gpu.step(0, nsteps) == nsteps
        if tag == "many":                                    # every tracer a copy of the first (the ideal age, which updtrc treats on its own)
            t = gpu.get("trc")
            for nt in range(1, case.ntr):
                t[nt * 2 * kk:(nt + 1) * 2 * kk] = t[:2 * kk]
            gpu.put("trc", t)
        pbot = float(np.max(gpu.get("p")[kk][4:-4, 4:-4][ip[4:-4, 4:-4] > 0]))
        gpu.set("vcoord_type", "cntiso_hybrid")
        gpu.set("ale_regrid_method", method)
        gpu.set_vector("plevel", 0.4 * pbot * (np.arange(kk) / kk) ** 1.3)
        gpu.set("delt1", 2.0 * case.params["baclin"])
        if neutral:
            gpu.set("ltedtp_opt", 2)
            gpu.set("ndiff_surface_align", 1)
            gpu.put("dpml", 9806.0 * 40.0 * np.ones((1, case.jdm + 8, case.idm + 8)))
        gpu.stage("ale_regrid_remap", *hostinit.step_indices(nsteps, kk))
        out[tag] = {nm: gpu.get(nm) for nm in OUT + (NDIFF_OUT if neutral else [])}
        gpu.close()
    wet = ip > 0
    for nm in OUT + (NDIFF_OUT if neutral else []):
        if nm == "trc":
            continue
        a, b = out["one"][nm], out["many"][nm]
        assert np.array_equal(a[:, wet], b[:, wet]), nm
    t1, tn = out["one"]["trc"], out["many"]["trc"]
    for nt in range(ntr):
        assert np.array_equal(tn[nt * 2 * kk:(nt + 1) * 2 * kk][:, wet], t1[:2 * kk][:, wet]), f"tracer {nt + 1}"
