"""CROSS-CHECK (not a pin): config 2's "full blom_step" as far as it is built -- blom_amd/stepper.py FULL_STAGES = the order of
phy/mod_blom_step.F90:96-253 for isopyc_bulkml: init_fluxes, tmsmt1, cmnfld2, difest_isobml (its part in front of the
diffusivity estimates), eddtra, advect, pbcor1, diffus, pgforc, momtum, convec, diapfl, thermf, mxlayr, updtrc, barotp, pbcor2,
tmsmt2, cmnfld1 -- device-resident (blomgpu_step with the option full_physics) against the reference's own modules stepped stage
by stage (builds *_xml: real mod_cmnfld_routines, mod_eddtra, mod_niw, mod_thermf_channel, mod_mxlayr behind the stand-ins of
oracle/xcheck).  A surface heat flux that changes sign across the domain and a fresh water flux are switched on (the channel
experiment's own forcing is zero), so that thermf's fluxes are non-trivial and mxlayr entrains and detrains while the run goes
on.  After every step all state arrays must agree bit for bit."""
import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step, FULL_STAGES
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, GRID_FIELDS, INT_FIELDS
from test_xcheck_eddtra import _WithEddtra

pytestmark = pytest.mark.gpu
CMN = ["bfsqi", "bfsql", "bfsqf", "nslpx", "nslpy", "nnslpx", "nnslpy"]
ML = ["ustar", "ustar3", "idkedt", "uml", "vml", "umlres", "vmlres", "surflx", "sswflx", "surrlx", "salflx", "brnflx", "salrlx", "salt_corr",
      "trc_corr", "trflx", "mtkeus", "mtkeni", "mtkebf", "mtkers", "mtkepe", "mtkeke", "pbrnda", "buoyfl", "fmltfz", "sfl", "hmltfz"]
CHECK = [f for f in STATE_FIELDS + INT_FIELDS if f not in ("util1", "util2", "util3", "util4", "uflux", "vflux", "uflux2", "vflux2", "uflux3",
                                                            "vflux3", "utotm", "vtotm")] + CMN + ML


def _forcing(case, be, amp):
    nj, ni = case.jdm + 8, case.idm + 8
    y = np.linspace(-1.0, 1.0, nj)[:, None] + 0.0 * np.arange(ni)[None, :]
    x = np.linspace(0.0, 2 * np.pi, ni)[None, :] + 0.0 * y
    be.put("nsf", (amp * 300.0 * (y + 0.3 * np.sin(x)))[None])          # cooling in one half of the domain, heating in the other
    be.put("swa", (amp * 120.0 * (1.0 + np.cos(x)) * (y > -0.5))[None])
    be.put("eva", (-2e-5 * (1.0 + 0.5 * np.sin(2 * x)))[None])
    be.put("lip", (3e-5 * (y > 0.0))[None])


def _full_step_check(cfg, nsteps, relax):
    from oracle.refblom import get_ref_backend, have_ref
    from blom_amd.gpu import BlomGpu
    big = cfg.split("_")[0] in ("channel", "tnx2v1s", "tnx1v4s")
    lib = cfg + "_omp_xml" if big else cfg.replace("_tke", "") + "_xml"
    if not have_ref(lib):
        pytest.skip(f"oracle/_ref/{lib}/libblomref.so not built")
    case = make_case(cfg, nslp0=0.0)
    ref = _WithEddtra(get_ref_backend(lib, case.depth))
    ref.ref.set("eitmth", "gm")
    hostinit.init_state(ref, case)
    six0 = hostinit.step_indices(0, case.kdm)
    ref.ref.stage("mxlayr_init", *six0)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            gpu.set(nm, v)
    hostinit.init_forcing(ref, case)
    _forcing(case, ref, 1.0)
    opts = dict(rm0=1.2, rm5=0.0, niwgf=0.4, niwbf=0.35, ce=0.06, tau_mlr=86400.0, lfmin=5.0e3, swamxd=200.0, sref=34.65, xmi=0.25,
                trxday=20.0 if relax else 0.0, srxday=40.0 if relax else 0.0, trxdpt=1.0, srxdpt=1.0, trxlim=1.5, srxlim=0.5)
    for nm, v in opts.items():
        ref.ref.set(nm, float(v))
        gpu.set(nm, float(v))
    ref.ref.set("mlrttp", "constant")
    gpu.set("mlrttp", "constant")
    for nm, v in dict(l1mi=12, l2mi=1, l3mi=2, l4mi=3, l5mi=4, aptflx=0, apsflx=0, ditflx=0, disflx=0, srxbal=0).items():
        ref.ref.set(nm, int(v))
        gpu.set(nm, int(v))
    for nm, v in dict(nstep_in_day=96, nday_of_year=20, nday_in_year=365).items():
        ref.ref.set(nm, int(v))
    copy_state(ref, gpu, fields=STATE_FIELDS + INT_FIELDS + GRID_FIELDS + CMN + ML)
    hostinit.init_forcing(gpu, case)                     # (after the grid is there: it sums the ocean area)
    _forcing(case, gpu, 1.0)
    gpu.set("delt1", case.params["baclin"])
    gpu.set("full_physics", 1)
    try:
        nr = ng = 0
        kf0 = gpu.get("kfpla").copy()
        for _ in range(nsteps):
            nr = dyncore_step(ref, nr, case.params["baclin"], stages=FULL_STAGES)
            assert gpu.step(ng, 1) == ng + 1
            ng += 1
            # (the OpenMP build of the reference at channel size leaves utotn, vtotn untouched outside the interior: they are
            # firstprivate in momtum's layer loop, phy/mod_momtum.F90:342-350 -- interior only there)
            omp = big
            bad = diff_report(ref, gpu, fields=[f for f in CHECK if not (omp and f in ("utotn", "vtotn"))])
            assert not bad, f"step {nr}\n" + fmt_report(bad[:12])
            if omp:
                for nm in ("utotn", "vtotn"):
                    assert np.array_equal(ref.get(nm)[:, 4:-4, 4:-4], gpu.get(nm)[:, 4:-4, 4:-4]), nm
        wet = ref.masks["ip"][4:-4, 4:-4] > 0
        pe = gpu.get("mtkepe")[0, 4:-4, 4:-4][wet]
        assert (pe != 0.0).any(), "mxlayr entrained nowhere"       # (columns of both kinds: tests/test_xcheck_mxlayr.py)
        assert np.abs(gpu.get("surflx")[0, 4:-4, 4:-4][wet]).max() > 0.0
        assert np.isfinite(gpu.get("u")[:, 4:-4, 4:-4]).all()
        return kf0
    finally:
        for nm in ("trxday", "srxday"):
            ref.ref.set(nm, 0.0)
        gpu.close()


@pytest.mark.parametrize("cfg,nsteps,relax", [("chan_s_tke", 10, False), ("box_s", 8, True), ("tri_s_tke", 8, False)])
def test_full_physics_step_equals_the_reference_stage_sequence(cfg, nsteps, relax):
    _full_step_check(cfg, nsteps, relax)


def test_full_size_channel_full_physics_step_equals_the_reference_stage_sequence():
    """three steps at BASELINE.json's channel size (208x512x53, ntr = 3): what `bench.py --opt full_physics=1` times"""
    from test_xcheck_ale import run_with_big_stack
    run_with_big_stack(_full_step_check, "channel_tke", 3, False)


def test_full_size_tnx2v1s_full_physics_step_equals_the_reference_stage_sequence():
    """two steps at the tnx2v1 grid's size (180x193x53, arctic patch): thermf's sums leave the seam row out, mxlayr and the
    velocity halos cross the patch"""
    from test_xcheck_ale import run_with_big_stack
    run_with_big_stack(_full_step_check, "tnx2v1s_tke", 2, True)
