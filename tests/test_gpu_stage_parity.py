"""Stage-by-stage parity of the HIP kernels (through the C-ABI) against the reference's own
compiled Fortran (oracle/_ref/<cfg>/libblomref.so), on identical inputs.

The reference is stepped through the dyncore stage sequence; before every stage its complete
state is uploaded to the device, the same stage is run there, and every array is compared.
Bar: bit-exact (==) for everything -- these stages contain only + - * / sqrt min max, compiled
without contraction on both sides.  Stages containing exp() (barotp's uglue, diapfl) use the
tolerance stated next to them.
"""
import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step, DYNCORE_STAGES, STAGES_FROZEN_EDDY_FLUXES
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, INT_FIELDS

pytestmark = pytest.mark.gpu

# stage -> (rtol, atol); default exact.  Empty: every stage is compared bit for bit, exp() included
# (blom_amd/csrc/exp_libm.h evaluates it the way the libm under the reference does; tests/test_exp_libm.py)
TOL = {}
GPU_STAGES = ["init_fluxes", "tmsmt1", "halo_cmnfld2", "halo_difest", "advect", "pbcor1", "diffus", "pgforc",
              "momtum", "convec", "diapfl", "mxlayr_tail", "updtrc", "barotp", "pbcor2", "tmsmt2"]
# 2-D scratch arrays of mod_utility that the reference leaves holding the last layer's
# temporaries (phy/mod_momtum.F90:398-423, phy/mod_pbcor.F90:172-236); the device keeps such
# temporaries in its work space instead.
SCRATCH = {"uflux", "vflux", "uflux2", "vflux2", "uflux3", "vflux3", "utotm", "vtotm"}


def _run(cfg, nsteps, stages, eddy=False, ntr=None, **overrides):
    from oracle.refblom import get_ref_backend, have_ref
    from blom_amd.gpu import BlomGpu
    if not have_ref(cfg):
        pytest.skip(f"oracle/_ref/{cfg}/libblomref.so not built")
    case = make_case(cfg, ntr=ntr, **overrides)
    ref = get_ref_backend(cfg, case.depth, ntr=ntr)
    hostinit.init_state(ref, case)
    if eddy:                                    # non-zero umfltd, vmfltd, umflsm, vmflsm in front of advect
        hostinit.frozen_eddy_fluxes(ref, case)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            gpu.set(nm, v)
    failures = []
    pending = {}
    nstep = [0]

    def check():
        if "st" in pending:
            st = pending.pop("st")
            rtol, atol = TOL.get(st, (0.0, 0.0))
            fields = [f for f in STATE_FIELDS + INT_FIELDS if f not in SCRATCH]
            bad = diff_report(ref, gpu, fields=fields, rtol=rtol, atol=atol)
            if bad:
                failures.append(f"step {nstep[0] + 1} stage {st}:\n" + fmt_report(bad))

    def hook(st, six):
        check()
        if st not in stages:
            return
        copy_state(ref, gpu)
        gpu.set("nstep", nstep[0] + 1)
        gpu.set("delt1", ref.ref.get_real("delt1"))
        gpu.stage(st, *six)
        pending["st"] = st

    for _ in range(nsteps):
        new = dyncore_step(ref, nstep[0], case.params["baclin"], hook=hook,
                           stages=STAGES_FROZEN_EDDY_FLUXES if eddy else DYNCORE_STAGES)
        check()
        nstep[0] = new
    gpu.close()
    assert not failures, "\n".join(failures[:40])


@pytest.mark.parametrize("cfg", ["chan_s", "box_s", "tri_s", "chan_s_tke", "box_s_tke", "chan_s_tk2", "chan_s_tk0"])
def test_stage_parity_small(cfg):
    _run(cfg, 4, GPU_STAGES)


@pytest.mark.parametrize("cfg", ["chan_s", "box_s", "tri_s", "chan_s_tke", "tri_s_tke"])
def test_stage_parity_with_eddy_fluxes(cfg):
    """The bench workload has eddtra's non-zero mass fluxes in front of advect; the reference build has no mod_eddtra, so
    a synthetic field of them (smooth, sign-changing, every 97th point beyond the CFL clamp) is written into its module
    arrays: advect's cau/cav (phy/mod_advect.F90:72-94) and everything downstream of them against the compiled reference."""
    _run(cfg, 4, GPU_STAGES, eddy=True)


@pytest.mark.parametrize("cfg,ntr", [("chan_s_tke", 11), ("tri_s_tke", 6), ("box_s_tke", 9), ("chan_s_tk0", 7)])
def test_stage_parity_with_many_tracers(cfg, ntr):
    """More tracers than one batch (4) of the tile kernels holds -- BASELINE.json's config 5 advects iHAMOCC's through these
    stages -- against the reference's own stages carrying that many (its tracer count is a run-time quantity,
    trc/mod_tracers.F90:116-126: ntr = ... + ntrbgc; the harness re-allocates its tracer arrays, ref_set_ntr).  The extra
    tracers are plain passive ones.  With the eddy-induced fluxes on, so that remap works on all of them."""
    _run(cfg, 4, GPU_STAGES, eddy=True, ntr=ntr)


def test_stage_parity_fuk95():
    _run("fuk95", 2, GPU_STAGES)


# namelist-selected variants of the stages (phy/mod_momtum.F90:723-820, phy/mod_pgforc.F90:524-534,
# phy/mod_pbcor.F90:99-105), each against the reference run with the same option
@pytest.mark.parametrize("opts", [dict(mommth="enecon"), dict(mommth="enedis"), dict(pgfmth="dynamic enthalpy"),
                                  dict(bmcmth="dluc"), dict(mommth="enedis", pgfmth="dynamic enthalpy", bmcmth="dluc")],
                         ids=lambda o: "+".join(f"{k}={v}" for k, v in o.items()).replace(" ", "_"))
@pytest.mark.parametrize("cfg", ["chan_s", "box_s"])
def test_stage_parity_option_variants(cfg, opts):
    try:
        _run(cfg, 3, GPU_STAGES, **opts)
    finally:                        # the reference instance is shared between tests: back to the defaults
        from oracle.refblom import get_ref_backend
        case = make_case(cfg)
        hostinit.init_state(get_ref_backend(cfg, case.depth), case)


# ---------------------------------------------------------------------------------------------
# Free-running: the device-resident stepping loop (blomgpu_step) against the reference stepped
# stage by stage, from the same initial state.  Bit for bit: exp() -- barotp's coastal damping
# coefficient, diapfl's bottom boundary layer term (phy/mod_barotp.F90:183,205,
# phy/mod_diapfl.F90:204) -- is evaluated on the device with the algorithm and the fused operations
# of the libm the reference calls (blom_amd/csrc/exp_libm.h), so no tolerance is needed; with the
# device math library's exp (1 ulp) runs used to decorrelate through limiter decisions.
# ---------------------------------------------------------------------------------------------
FREERUN_FIELDS = ["u", "v", "dp", "temp", "saln", "sigma", "pb", "ub", "vb", "ubflxs_p", "pb_p", "trc",
                  "uflx", "vflx", "pgfx", "pgfy", "dpu", "dpv"]


@pytest.mark.parametrize("cfg,nsteps,rtol", [("chan_s", 40, 0.0), ("box_s", 40, 0.0), ("fuk95", 12, 0.0),
                                             ("chan_s_tke", 40, 0.0), ("tri_s", 24, 0.0), ("tri_s_tke", 24, 0.0), ("chan_s_tk2", 40, 0.0), ("chan_s_tk0", 40, 0.0)])
def test_freerun_device_resident(cfg, nsteps, rtol):
    _freerun_device(cfg, nsteps, rtol, False)


@pytest.mark.parametrize("cfg,nsteps", [("chan_s", 24), ("box_s", 24), ("chan_s_tke", 24), ("tri_s", 16), ("tri_s_tke", 16)])
def test_freerun_device_resident_with_eddy_fluxes(cfg, nsteps):
    """blomgpu_step with the option eddtra_frozen against the reference stepped stage by stage, both holding the same frozen
    field of non-zero eddy-induced mass fluxes."""
    _freerun_device(cfg, nsteps, 0.0, True)


@pytest.mark.parametrize("cfg,nsteps,ntr", [("chan_s_tke", 24, 11), ("tri_s_tke", 16, 6), ("box_s_tke", 16, 24)])
def test_freerun_device_resident_with_many_tracers(cfg, nsteps, ntr):
    _freerun_device(cfg, nsteps, 0.0, True, ntr=ntr)


def _freerun_device(cfg, nsteps, rtol, eddy, ntr=None):
    from oracle.refblom import get_ref_backend, have_ref
    from blom_amd.gpu import BlomGpu
    if not have_ref(cfg):
        pytest.skip(f"oracle/_ref/{cfg}/libblomref.so not built")
    case = make_case(cfg, ntr=ntr)
    ref = get_ref_backend(cfg, case.depth, ntr=ntr)
    hostinit.init_state(ref, case)
    if eddy:
        hostinit.frozen_eddy_fluxes(ref, case)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            gpu.set(nm, v)
    gpu.set("eddtra_frozen", int(eddy))
    copy_state(ref, gpu)
    gpu.set("delt1", case.params["baclin"])
    ns = 0
    for _ in range(nsteps):
        ns = dyncore_step(ref, ns, case.params["baclin"], stages=STAGES_FROZEN_EDDY_FLUXES if eddy else DYNCORE_STAGES)
    assert gpu.step(0, nsteps) == nsteps
    gpu.sync()
    exact = diff_report(ref, gpu, fields=FREERUN_FIELDS)
    print(f"{cfg}: after {nsteps} steps, fields not bit-identical: {[b[0] for b in exact]}")
    bad = diff_report(ref, gpu, fields=FREERUN_FIELDS, rtol=rtol, atol=rtol)
    gpu.close()
    assert not bad, fmt_report(bad)


def test_full_size_with_24_tracers_matches_reference():
    """BASELINE.json's channel at full size carrying 24 tracers (config 5 advects iHAMOCC's through these stages), with the
    eddy-induced fluxes on: the reference's own Fortran carrying them (ref_set_ntr) against the device-resident sequence."""
    test_full_size_matches_reference("channel_tke", True, ntr=24, nsteps=3)


@pytest.mark.parametrize("ntr", [3, 24])
def test_full_size_tnx1v4s_matches_reference(ntr):
    """BASELINE.json's config 5 at its size: the tnx1v4 grid's dimensions (360x385x53, bld/tnx1v4/patch.input.32:2, arctic
    patch, synthetic bathymetry) with the default tracer set and with 24 tracers (the extra ones standing in for iHAMOCC's,
    advected on the device), eddy-induced fluxes on: three device-resident steps against the reference's own Fortran
    carrying that many tracers (oracle/_ref/tnx1v4s_tke_omp).  Bit for bit."""
    test_full_size_matches_reference("tnx1v4s_tke", True, ntr=ntr, nsteps=3)


@pytest.mark.parametrize("eddy", [False, True], ids=["zero_eddy_fluxes", "eddy_fluxes"])
@pytest.mark.parametrize("cfg", ["channel_tke", "tnx2v1s_tke"])
def test_full_size_matches_reference(cfg, eddy, ntr=None, nsteps=4):
    """BASELINE.json's channel (208x512x53) with the reference's default tracer set (ntr = 3), the bench workload, and
    the tnx2v1 grid's dimensions (180x193x53, arctic patch, synthetic bathymetry): the device-resident sequence against
    the reference's own Fortran (built with its OpenMP directives, oracle/_ref/<cfg>_omp) over the forward step and
    three leap-frog steps.  Bit for bit."""
    import os
    import threading
    from oracle.refblom import get_ref_backend, have_ref
    from blom_amd.gpu import BlomGpu
    if not have_ref(cfg + "_omp"):
        pytest.skip(f"oracle/_ref/{cfg}_omp/libblomref.so not built")
    res = {}

    def body():
        case = make_case(cfg, ntr=ntr)
        ref = get_ref_backend(cfg + "_omp", case.depth, ntr=ntr)
        hostinit.init_state(ref, case)
        if eddy:
            hostinit.frozen_eddy_fluxes(ref, case)
        gpu = BlomGpu(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
        for nm, v in case.params.items():
            if not nm.endswith("0"):
                gpu.set(nm, v)
        gpu.set("eddtra_frozen", int(eddy))
        copy_state(ref, gpu)
        gpu.set("delt1", case.params["baclin"])
        ns = 0
        for _ in range(nsteps):
            ns = dyncore_step(ref, ns, case.params["baclin"], stages=STAGES_FROZEN_EDDY_FLUXES if eddy else DYNCORE_STAGES)
        assert gpu.step(0, nsteps) == nsteps
        gpu.sync()
        res["bad"] = diff_report(ref, gpu, fields=FREERUN_FIELDS)
        res["ntr"] = ref.ntr
        gpu.close()

    # the reference keeps its stage-local 2-D work arrays on the stack (BLOM runs with ulimit -s unlimited)
    os.environ["OMP_NUM_THREADS"] = str(min(16, os.cpu_count() or 1))
    os.environ["OMP_STACKSIZE"] = "1G"
    threading.stack_size(2 << 30)
    th = threading.Thread(target=lambda: res.update(err=None) or body())
    th.start()
    th.join()
    threading.stack_size(0)
    assert "bad" in res, "the comparison did not complete"
    assert res["ntr"] == (ntr or 3)
    assert not res["bad"], fmt_report(res["bad"])


@pytest.mark.parametrize("itype", [1, 2, 3, 4, 11, 12, 13, 14])
def test_xctilr_arctic_patch_matches_reference(itype):
    """nreg = 2: the device halo update for every grid/field type (phy/mod_xc.F90:107-110, :4262-4372)
    against the reference built with ARCTIC, through the C-ABI."""
    from oracle.refblom import get_ref_backend, have_ref
    from blom_amd.gpu import BlomGpu
    if not have_ref("tri_s"):
        pytest.skip("oracle/_ref/tri_s/libblomref.so not built")
    case = make_case("tri_s")
    ref = get_ref_backend("tri_s", case.depth)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
    rng = np.random.default_rng(itype)
    kk = case.kdm
    for mh, nh in ((0, 0), (1, 1), (2, 3), (4, 4), (3, 0), (0, 2)):
        a = rng.standard_normal((2 * kk, case.jdm + 8, case.idm + 8))
        b = a.copy()
        gpu.put("u", a)
        gpu.xctilr("u", 1, 1, 2 * kk, mh, nh, itype)
        ref.ref.xctilr(b, 1, 2 * kk, mh, nh, itype)
        assert np.array_equal(gpu.get("u"), b), (itype, mh, nh)
    gpu.close()


def test_sfcstr_follows_the_reference_switch():
    """phy/mod_sfcstr.F90:33-62: empty for the idealised experiments, an error for unknown ones"""
    from blom_amd.gpu import BlomGpu, BlomGpuError
    case = make_case("chan_s")
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    six = hostinit.step_indices(0, case.kdm)
    for e in ("channel", "fuk95", "noforcing"):
        gpu.set("expcnf", e)
        gpu.stage("sfcstr", *six)
    gpu.set("expcnf", "nonsense")
    with pytest.raises(BlomGpuError, match="sfcstr: expcnf = nonsense is unsupported!"):
        gpu.stage("sfcstr", *six)
    gpu.set("expcnf", "cesm")
    with pytest.raises(BlomGpuError, match="not built on the device"):
        gpu.stage("sfcstr", *six)
    gpu.close()
