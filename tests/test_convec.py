"""convec (phy/mod_convec.F90:43): the test cases of the suite are statically stable, so the
free-running comparisons only exercise its velocity remap.  Here the state handed to the stage is
made unstable first (tests/parity.py: destabilise_for_convec) and the stage is compared, bit for bit,
  * C restatement (oracle/c/convec.c) against the reference's compiled mod_convec  -- CPU suite,
  * device (blom_amd/csrc/stage_convec.hip, through the C ABI) against the reference -- GPU suite.
"""
import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step
from parity import (copy_state, diff_report, fmt_report, destabilise_for_convec, STATE_FIELDS, INT_FIELDS)

SCRATCH = {"uflux", "vflux", "uflux2", "vflux2", "uflux3", "vflux3", "utotm", "vtotm", "util1", "util2", "util3",
           "util4"}


def _drive(cfg, nsteps, make_other, ntr=None):
    from oracle.refblom import get_ref_backend, have_ref
    if not have_ref(cfg):
        pytest.skip(f"oracle/_ref/{cfg}/libblomref.so not built")
    case = make_case(cfg, ntr=ntr)
    ref = get_ref_backend(cfg, case.depth, ntr=ntr)
    hostinit.init_state(ref, case)
    other = make_other(case, ref)
    failures, acted = [], []
    pending = {}
    nstep = [0]
    fields = [f for f in STATE_FIELDS + INT_FIELDS if f not in SCRATCH]

    def check():
        if pending.pop("st", None):
            before = pending.pop("before")
            changed = {nm: int((np.asarray(ref.get(nm)) != before[nm]).sum()) for nm in before}
            acted.append(changed)
            bad = diff_report(ref, other, fields=fields)
            if bad:
                failures.append(f"step {nstep[0] + 1}:\n" + fmt_report(bad))

    def hook(st, six):
        check()
        if st != "convec":
            return
        destabilise_for_convec(ref, case, six[1], seed=nstep[0])
        copy_state(ref, other)
        other.set("nstep", nstep[0] + 1)
        other.stage(st, *six)
        pending["st"] = st
        pending["before"] = {nm: np.array(ref.get(nm)) for nm in ("dp", "temp", "saln", "kfpla", "u") + (("trc",) if ntr else ())}

    for _ in range(nsteps):
        # the edited state is not meant to be integrated further: only convec itself is compared,
        # and every step starts again from the reference's own (restored) trajectory
        new = dyncore_step(ref, nstep[0], case.params["baclin"], hook=hook,
                           stages=("init_fluxes", "tmsmt1", "convec"))
        check()
        nstep[0] = new
    if hasattr(other, "close"):
        other.close()
    assert not failures, "\n".join(failures[:10])
    # the stage did real work: layers were merged, the mixed-layer base moved, velocities were remapped
    for ch in acted:
        assert ch["dp"] > 0 and ch["temp"] > 0 and ch["kfpla"] > 0 and ch["u"] > 0, ch
        assert not ntr or ch["trc"] > 0, ch


@pytest.mark.parametrize("cfg", ["chan_s", "box_s", "fuk95", "tri_s", "chan_s_tke"])
def test_c_restatement_matches_reference_on_unstable_columns(cfg):
    from oracle.coracle import COracle, have_coracle
    if not have_coracle():
        pytest.skip("oracle/_ref/liboracle_c.so not built")

    def make(case, ref):
        co = COracle(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
        for nm, v in case.params.items():
            if not nm.endswith("0"):
                co.set(nm, v)
        return co
    _drive(cfg, 3, make)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,ntr", [("chan_s", None), ("box_s", None), ("fuk95", None), ("tri_s", None), ("chan_s_tke", None),
                                     ("chan_s_tke", 10), ("box_s_tke", 7), ("tri_s_tke", 13)])
def test_device_matches_reference_on_unstable_columns(cfg, ntr):
    """(ntr: the reference carries that many tracers itself, ref_set_ntr -- the tracers' share of a mixing event is evaluated four
    tracers at a time once the event's extent is known, stage_convec.hip)"""
    from blom_amd.gpu import BlomGpu

    def make(case, ref):
        gpu = BlomGpu(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
        for nm, v in case.params.items():
            if not nm.endswith("0"):
                gpu.set(nm, v)
        return gpu
    _drive(cfg, 3, make, ntr=ntr)
