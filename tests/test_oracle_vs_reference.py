"""Pins the C restatement against the reference's own compiled Fortran (only where
oracle/_ref/<cfg>/libblomref.so exists): both start from the same state and are stepped
independently through the dyncore sequence; every field must stay bit-identical."""
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, INT_FIELDS


@pytest.mark.parametrize("cfg,nsteps", [("chan_s", 12), ("box_s", 12), ("fuk95", 3), ("tri_s", 8),
                                        ("chan_s_tke", 12), ("box_s_tke", 8), ("tri_s_tke", 8), ("chan_s_tk2", 12), ("chan_s_tk0", 12)])
def test_freerun_bit_identical(cfg, nsteps):
    from oracle.refblom import get_ref_backend, have_ref
    from oracle.coracle import COracle, have_coracle
    if not (have_ref(cfg) and have_coracle()):
        pytest.skip("reference / C oracle libraries not built")
    case = make_case(cfg)
    ref = get_ref_backend(cfg, case.depth)
    hostinit.init_state(ref, case)
    co = COracle(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            co.set(nm, v)
    copy_state(ref, co)
    co.set("delt1", case.params["baclin"])
    fields = [f for f in STATE_FIELDS + INT_FIELDS if f not in ("util1", "util2")]
    nr = nc = 0
    for _ in range(nsteps):
        nr = dyncore_step(ref, nr, case.params["baclin"])
        nc = dyncore_step(co, nc, case.params["baclin"])
        bad = diff_report(ref, co, fields=fields)
        assert not bad, f"step {nr}\n" + fmt_report(bad[:8])
