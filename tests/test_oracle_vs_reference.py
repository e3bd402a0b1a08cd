"""Pins the C restatement against the reference's own compiled Fortran (only where
oracle/_ref/<cfg>/libblomref.so exists): both start from the same state and are stepped
independently through the dyncore sequence; every field must stay bit-identical."""
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step, DYNCORE_STAGES, STAGES_FROZEN_EDDY_FLUXES
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, INT_FIELDS


@pytest.mark.parametrize("cfg,nsteps", [("chan_s", 12), ("box_s", 12), ("fuk95", 3), ("tri_s", 8),
                                        ("chan_s_tke", 12), ("box_s_tke", 8), ("tri_s_tke", 8), ("chan_s_tk2", 12), ("chan_s_tk0", 12)])
def test_freerun_bit_identical(cfg, nsteps):
    _freerun(cfg, nsteps, False)


# the same with non-zero eddy-induced mass fluxes in front of advect (hostinit.frozen_eddy_fluxes: a frozen synthetic field,
# written into the reference's module arrays; its build has no mod_eddtra): cau/cav of phy/mod_advect.F90:72-94 with the
# umfltd/umflsm terms acting and the clamp reached, remap on the flux areas they cause
@pytest.mark.parametrize("cfg,nsteps", [("chan_s", 12), ("box_s", 12), ("tri_s", 8), ("chan_s_tke", 12), ("tri_s_tke", 8)])
def test_freerun_bit_identical_with_eddy_fluxes(cfg, nsteps):
    _freerun(cfg, nsteps, True)


# the reference's stages carrying MORE tracers than its build options give it: its tracer count is a run-time quantity
# (ntr = ntrocn + ntrtke + ntrgls + ntriag + ntrbgc, trc/mod_tracers.F90:116-126; every stage loops do nt = 1,ntr), the
# harness re-allocates its tracer arrays (ref_set_ntr).  The extra tracers are plain passive ones, what iHAMOCC's are to
# the dynamical core.  ntr = 11, 6: beyond the 4 tracers one batch of the device's tile kernels holds.
@pytest.mark.parametrize("cfg,nsteps,ntr", [("chan_s_tke", 8, 11), ("tri_s_tke", 6, 6), ("box_s_tke", 6, 9)])
def test_freerun_bit_identical_with_many_tracers(cfg, nsteps, ntr):
    _freerun(cfg, nsteps, True, ntr=ntr)


def _freerun(cfg, nsteps, eddy, ntr=None):
    from oracle.refblom import get_ref_backend, have_ref
    from oracle.coracle import COracle, have_coracle
    if not (have_ref(cfg) and have_coracle()):
        pytest.skip("reference / C oracle libraries not built")
    case = make_case(cfg, ntr=ntr)
    ref = get_ref_backend(cfg, case.depth, ntr=ntr)
    assert ref.ntr == case.ntr
    hostinit.init_state(ref, case)
    if eddy:
        hostinit.frozen_eddy_fluxes(ref, case)
    stages = STAGES_FROZEN_EDDY_FLUXES if eddy else DYNCORE_STAGES
    co = COracle(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            co.set(nm, v)
    copy_state(ref, co)
    co.set("delt1", case.params["baclin"])
    fields = [f for f in STATE_FIELDS + INT_FIELDS if f not in ("util1", "util2")]
    nr = nc = 0
    for _ in range(nsteps):
        nr = dyncore_step(ref, nr, case.params["baclin"], stages=stages)
        nc = dyncore_step(co, nc, case.params["baclin"], stages=stages)
        bad = diff_report(ref, co, fields=fields)
        assert not bad, f"step {nr}\n" + fmt_report(bad[:8])
