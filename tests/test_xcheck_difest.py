"""CROSS-CHECK (not a pin): the diffusivity estimates of difest_isobml -- difest_common_iso, difest_vertical_iso, difest_lateral_iso
(phy/mod_difest.F90:353-586, :2629-3084, :2040-2627) -- on the device against the reference's REAL mod_difest.F90, compiled against
interface-only stand-ins for the CVMix modules it imports at module level and does not call on this path (oracle/xcheck/
cvmix_standin.F90, mod_tidaldissip_standin.F90; builds *_xdf of oracle/Makefile).

  * the TKE closure's derived constants (initke) as this library evaluates them equal the reference's, bit for bit;
  * config 2's whole step with LIVE diffusivities (blom_amd/stepper.py FULL_STAGES_LIVE; blomgpu_step with full_physics + difest_live),
    device-resident against the reference's modules stepped stage by stage: after every step every state array, difint, difiso,
    difdia, difwgt and the closure's arrays agree bit for bit -- with the option sets of the reference's own tests/fuk95/limits
    (large scale Eady growth rate, eddy suppression), of the shear form with anisotropy, of two-dimensional diffusivities with a 3-D
    Redi profile and lateral smoothing, with the TKE closure (chan_s, tri_s) and with the Richardson number form (box_s: a build
    without -DTKE), latitude dependent background mixing, tidal mixing across the critical latitude, surface TKE penetration."""
import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step, FULL_STAGES_LIVE
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, GRID_FIELDS, INT_FIELDS
from test_xcheck_eddtra import _WithEddtra
from test_xcheck_fullstep import CMN, ML, CHECK, _forcing

pytestmark = pytest.mark.gpu
DFE_GRID = ["plat", "betafp", "betatp", "cosang", "sinang", "hangle", "twedon", "ficem"]
DFE_OUT = ["difint", "difiso", "difdia", "difwgt", "Prod", "Buoy", "Shear2", "L_scale"]
TKE_CONSTS = ["sqrt2", "cmu_fac1", "cmu_fac2", "cmu_fac3", "tke_exp1", "gls_exp1", "gls_fac6", "gls_s0", "gls_s1", "gls_s2", "gls_s4", "gls_s5",
              "gls_s6", "gls_b0", "gls_b1", "gls_b2", "gls_b3", "gls_b4", "gls_b5"]
# option sets: (reals, ints)
OPT_FUK95 = (dict(egc=0.85, eggam=200.0, eglsmn=4000.0, egmndf=50.0, egmxdf=1500.0, egidfq=1.0, ri0=1.2, tkepf=0.0),
             dict(eddf2d=0, edsprs=1, edanis=0, redi3d=0, edfsmo=0, edritp_opt=2, edwmth_opt=1, bdmtyp=2, iwdflg=1, bdmldp=0))
OPT_SHEAR = (dict(egc=0.85, eggam=200.0, eglsmn=400.0, egmndf=10.0, egmxdf=2500.0, egidfq=1.25, ri0=0.7, tkepf=0.006),
             dict(eddf2d=0, edsprs=0, edanis=1, redi3d=0, edfsmo=0, edritp_opt=1, edwmth_opt=2, bdmtyp=1, iwdflg=1, bdmldp=1))
OPT_2D = (dict(egc=0.85, eggam=200.0, eglsmn=4000.0, egmndf=50.0, egmxdf=1500.0, egidfq=1.25, ri0=1.2, tkepf=0.006),
          dict(eddf2d=1, edsprs=1, edanis=0, redi3d=1, edfsmo=1, edritp_opt=2, edwmth_opt=1, bdmtyp=2, iwdflg=0, bdmldp=1))


def _setup(cfg, opts, ntr=None):
    from oracle.refblom import get_ref_backend, have_ref
    from blom_amd.gpu import BlomGpu
    big = cfg.split("_")[0] in ("channel",)
    lib = cfg + "_omp_xdf" if big else cfg.replace("_tke", "") + "_xdf"
    if not have_ref(lib):
        pytest.skip(f"oracle/_ref/{lib}/libblomref.so not built")
    case = make_case(cfg, nslp0=0.0, ntr=ntr)
    ref = _WithEddtra(get_ref_backend(lib, case.depth, ntr=ntr))
    ref.ref.set("eitmth", "gm")
    six0 = hostinit.step_indices(0, case.kdm)
    ref.ref.stage("difest_init", *six0)                 # (before the state: initke resets the TKE tracers, difdia, ustarb)
    hostinit.init_state(ref, case)
    ref.ref.stage("mxlayr_init", *six0)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            gpu.set(nm, v)
    hostinit.init_forcing(ref, case)
    _forcing(case, ref, 1.0)
    hostinit.init_difest(ref, case)
    ref.put("buoyfl", 1.0e-8 * np.sin(np.arange(case.idm + 8))[None, None, :] * np.ones((case.kdm + 1, case.jdm + 8, 1)))   # both signs under the surface layer's stability function
    base = dict(rm0=1.2, rm5=0.0, niwgf=0.4, niwbf=0.35, niwlf=0.5, ce=0.06, tau_mlr=86400.0, lfmin=5.0e3, swamxd=200.0, sref=34.65, xmi=0.25,
                trxday=0.0, srxday=0.0, trxdpt=1.0, srxdpt=1.0, trxlim=1.5, srxlim=0.5, bdmc1=5.0e-8, bdmc2=1.0e-5, iwdfac=0.06, nubmin=1.0e-6)
    base.update(opts[0])
    for nm, v in base.items():
        ref.ref.set(nm, float(v))
        gpu.set(nm, float(v))
    ref.ref.set("mlrttp", "constant")
    gpu.set("mlrttp", "constant")
    ints = dict(l1mi=12, l2mi=1, l3mi=2, l4mi=3, l5mi=4, aptflx=0, apsflx=0, ditflx=0, disflx=0, srxbal=0, rhsctp=0)
    ints.update(opts[1])
    for nm, v in ints.items():
        ref.ref.set(nm, int(v))
        gpu.set(nm, int(v))
    for nm, v in dict(nstep_in_day=96, nday_of_year=20, nday_in_year=365).items():
        ref.ref.set(nm, int(v))
    copy_state(ref, gpu, fields=STATE_FIELDS + INT_FIELDS + GRID_FIELDS + CMN + ML + DFE_GRID + DFE_OUT)
    hostinit.init_forcing(gpu, case)
    _forcing(case, gpu, 1.0)
    hostinit.init_difest(gpu, case, device=True)
    gpu.put("buoyfl", ref.get("buoyfl"))
    gpu.set("delt1", case.params["baclin"])
    gpu.set("full_physics", 1)
    gpu.set("difest_live", 1)
    return case, ref, gpu, big


def test_the_tke_closures_constants_equal_the_reference():
    from oracle.refblom import get_ref_backend, have_ref
    from blom_amd.gpu import BlomGpu
    if not have_ref("chan_s_xdf"):
        pytest.skip("oracle/_ref/chan_s_xdf/libblomref.so not built")
    case = make_case("chan_s_tke", nslp0=0.0)
    ref = get_ref_backend("chan_s_xdf", case.depth)
    ref.ref.stage("difest_init", *hostinit.step_indices(0, case.kdm))
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
    try:
        bad = [(nm, ref.ref.get_real(nm).hex(), gpu.tke_const(nm).hex()) for nm in TKE_CONSTS if ref.ref.get_real(nm) != gpu.tke_const(nm)]
        assert not bad, bad
    finally:
        gpu.close()


def _live_step_check(cfg, nsteps, opts, stagewise=False, ntr=None):
    case, ref, gpu, big = _setup(cfg, opts, ntr=ntr)
    check = [f for f in CHECK if not (big and f in ("utotn", "vtotn"))] + [f for f in DFE_OUT if f not in CHECK]
    try:
        nr = ng = 0
        for _ in range(nsteps):
            if stagewise:                     # the stage on its own, from the reference's state in front of it
                def hook(st, six):
                    if st == "difest_isobml":
                        # (the small cases start at rest with level isopycnals: cmnfld2's large scale slopes are zero for many steps.
                        # The stage on its own is given synthetic ones, smooth and sign-changing, of the size a front produces)
                        for nm, ph in (("nnslpx", 0.0), ("nnslpy", 1.1)):
                            a = ref.get(nm)
                            kk_, nj_, ni_ = a.shape
                            kg, jg, ig = np.meshgrid(np.arange(kk_), np.arange(nj_), np.arange(ni_), indexing="ij")
                            a[...] = 3.0e-6 * np.sin(2 * np.pi * ig / 9.0 + ph + 0.3 * kg) * np.cos(2 * np.pi * jg / 7.0 - ph) * (1.0 + 0.1 * nr)
                        copy_state(ref, gpu, fields=STATE_FIELDS + INT_FIELDS + CMN + ML + DFE_OUT)
                        gpu.set("delt1", case.params["baclin"] * (1.0 if nr == 0 else 2.0))      # phy/mod_blom_step.F90:300
                        gpu.stage("difest_isobml", *six)
                        hook.six = six
                    elif getattr(hook, "six", None) is not None:
                        bad = diff_report(ref, gpu, fields=DFE_OUT + ["trc", "p", "ustar3", "idkedt"])
                        assert not bad, f"step {nr + 1}, difest_isobml on its own\n" + fmt_report(bad[:12])
                        hook.six = None
                nr = dyncore_step(ref, nr, case.params["baclin"], stages=FULL_STAGES_LIVE, hook=hook)
                continue
            nr = dyncore_step(ref, nr, case.params["baclin"], stages=FULL_STAGES_LIVE)
            assert gpu.step(ng, 1) == ng + 1
            ng += 1
            bad = diff_report(ref, gpu, fields=check)
            assert not bad, f"step {nr}\n" + fmt_report(bad[:12])
        wet = ref.masks["ip"][4:-4, 4:-4] > 0
        dd = gpu.get("difdia")[:, 4:-4, 4:-4][:, wet]
        di = gpu.get("difint")[:, 4:-4, 4:-4][:, wet]
        assert np.isfinite(dd).all() and np.isfinite(di).all()
        assert dd.max() > 2.0e-5, "the estimates did not move the diapycnal diffusivity"
        if stagewise and cfg != "tri_s_tke":              # (on tri_s every column sits on one of the bounds)
            assert np.ptp(di) > 0.0, "the estimates did not move the lateral diffusivities"
    finally:
        gpu.close()


@pytest.mark.parametrize("cfg,opts", [("chan_s_tke", OPT_FUK95), ("chan_s_tke", OPT_SHEAR), ("box_s", OPT_SHEAR), ("tri_s_tke", OPT_2D),
                                      ("box_s", OPT_2D), ("chan_s_tke", hostinit.DIFEST_NORESM), ("tri_s_tke", hostinit.DIFEST_NORESM),
                                      ("chan_s_tk2", OPT_FUK95), ("chan_s_tk2", OPT_SHEAR)])
def test_difest_isobml_on_its_own_equals_the_real_module(cfg, opts):
    """(chan_s_tk2: the build with -DGLS, turbclo = twoeq -- the length-scale variable a prognostic tracer stepped by difest_vertical_iso,
    phy/mod_difest.F90:2788-2814, :2858-2863, :2921-2927, with its surface flux from thermf_channel, channel/mod_thermf_channel.F90:167-175)"""
    _live_step_check(cfg, 4, opts, stagewise=True)


@pytest.mark.parametrize("cfg,nsteps,opts", [("chan_s_tke", 8, OPT_FUK95), ("chan_s_tke", 6, OPT_2D), ("box_s", 6, OPT_FUK95), ("box_s", 6, OPT_SHEAR),
                                             ("tri_s_tke", 6, OPT_FUK95), ("tri_s_tke", 6, OPT_SHEAR), ("chan_s_tke", 6, hostinit.DIFEST_NORESM),
                                             ("box_s", 6, hostinit.DIFEST_NORESM), ("chan_s_tk2", 8, hostinit.DIFEST_NORESM), ("chan_s_tk2", 6, OPT_2D)])
def test_full_step_with_live_diffusivities_equals_the_reference_stage_sequence(cfg, nsteps, opts):
    _live_step_check(cfg, nsteps, opts)


@pytest.mark.parametrize("cfg,nsteps,ntr", [("chan_s_tke", 5, 9), ("tri_s_tke", 4, 14), ("box_s", 4, 7)])
def test_full_step_with_live_diffusivities_and_many_tracers(cfg, nsteps, ntr):
    """config 5's regime on config 2's step: more tracers than the kernels take at a time (remap's batches of four, convec's and mxlayr's
    groups of four, diapfl's) through the WHOLE step with live diffusivities, the reference carrying them itself (ref_set_ntr)"""
    _live_step_check(cfg, nsteps, hostinit.DIFEST_NORESM, ntr=ntr)


def test_full_size_channel_step_with_24_tracers_equals_the_reference_stage_sequence():
    """two steps at the channel's size with 24 tracers (the default three + 21 passive ones): what bench.py --tracers 24 times, against the
    reference's modules carrying them"""
    from test_xcheck_ale import run_with_big_stack
    run_with_big_stack(_live_step_check, "channel_tke", 2, hostinit.DIFEST_NORESM, False, 24)


def test_full_size_channel_step_with_live_diffusivities_equals_the_reference_stage_sequence():
    """three steps at BASELINE.json's channel size (208x512x53, ntr = 3) with the options bench.py times by default (NorESM's &DIFFUSION
    defaults for isopyc_bulkml, rhsctp = .true. included: hostinit.DIFEST_NORESM)"""
    from test_xcheck_ale import run_with_big_stack
    run_with_big_stack(_live_step_check, "channel_tke", 3, hostinit.DIFEST_NORESM)


def _spunup_check(nspin, nsteps):
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    from oracle.refblom import get_ref_backend, have_ref
    if not have_ref("channel_tke_omp_xdf"):
        pytest.skip("oracle/_ref/channel_tke_omp_xdf/libblomref.so not built")
    case, nreg, masks = bench.build_case("channel", "remap", "default")
    gpu = bench.device_for_bench(case, nreg, masks, live=True)
    try:
        be = get_ref_backend("channel_tke_omp_xdf", case.depth, ntr=case.ntr)
        be.ref.set("eitmth", "gm")
        be.has_stage = lambda name: True
        bench.ref_full_init(be, case, True, True)
        ns = gpu.step(0, nspin)
        assert ns == nspin
        nr = bench.continue_from_device(be, gpu, ns, case)
        check = [f for f in CHECK if f not in ("utotn", "vtotn")] + [f for f in DFE_OUT if f not in CHECK]
        for _ in range(nsteps):
            nr = dyncore_step(be, nr, case.params["baclin"], stages=FULL_STAGES_LIVE)
            ns = gpu.step(ns, 1)
            bad = diff_report(be, gpu, fields=check)
            assert not bad, f"step {nr}\n" + fmt_report(bad[:12])
    finally:
        gpu.close()


def test_full_size_channel_step_from_a_spun_up_state_equals_the_reference_stage_sequence():
    """A state a model run is in, not the transient from rest: the bench workload (208x512x53, ntr = 3, NorESM's defaults) after 300
    device-resident steps -- deep mixed layers, an eddying flow of 2 m/s, massless layers throughout -- handed to the reference's
    modules (bench.continue_from_device: what `bench.py --spinup N` starts its CPU baseline from); two more steps on both sides,
    every array of the step equal bit for bit"""
    from test_xcheck_ale import run_with_big_stack
    run_with_big_stack(_spunup_check, 300, 2)


@pytest.mark.parametrize("cfg", ["chan_s_tke", "box_s"])
def test_rhsctp_acts_on_the_layer_interface_diffusivities(cfg):
    """rhsctp = .true. (NorESM's default; the topographic Rhines scale with sin / atan2 of the flow direction, phy/mod_difest.F90:2281-2340,
    :2393-2398) is not a no-op on the synthetic slopes and topographic beta of these cases: with the option on, difint differs from the run
    with it off in a good part of the cells -- and equals the reference's real mod_difest (the NorESM parameter sets of the tests above)"""
    case, ref, gpu, big = _setup(cfg, hostinit.DIFEST_NORESM)
    try:
        nr = dyncore_step(ref, 0, case.params["baclin"], stages=FULL_STAGES_LIVE)        # a state in motion
        res = {}
        six = hostinit.step_indices(nr, case.kdm)
        for nm, ph in (("nnslpx", 0.0), ("nnslpy", 1.1)):
            a = ref.get(nm)
            kk_, nj_, ni_ = a.shape
            kg, jg, ig = np.meshgrid(np.arange(kk_), np.arange(nj_), np.arange(ni_), indexing="ij")
            a[...] = 3.0e-6 * np.sin(2 * np.pi * ig / 9.0 + ph + 0.3 * kg) * np.cos(2 * np.pi * jg / 7.0 - ph)
        for rh in (0, 1):
            copy_state(ref, gpu, fields=STATE_FIELDS + INT_FIELDS + CMN + ML + DFE_OUT)
            gpu.set("delt1", 2.0 * case.params["baclin"])
            gpu.set("rhsctp", rh)
            gpu.stage("difest_isobml", *six)
            res[rh] = gpu.get("difint").copy()
        wet = ref.masks["ip"][4:-4, 4:-4] > 0
        d = (res[0] != res[1])[:, 4:-4, 4:-4][:, wet]
        assert d.mean() > 0.05, f"rhsctp changed {d.sum()} of {d.size} values"
        assert np.isfinite(res[1]).all()
    finally:
        gpu.close()
