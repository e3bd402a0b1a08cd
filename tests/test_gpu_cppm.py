"""advect with advmth = 'cppm' (phy/mod_cppm.F90; all four variants: cppm_compatibility
full/partial x cppm_limiting non_oscillatory/monotonic) on the device against the reference's own compiled code, bit for bit.

The reference library runs the whole stage sequence with cppm; before every advect its state is
uploaded to the device, advect is run there, and every array is compared (==).  The coefficient
tables of init_cppm are private to mod_cppm, so they are validated through the results: the device
computes its own tables (blomgpu_init_cppm) from ip, scpx, scpy."""
import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, INT_FIELDS

pytestmark = pytest.mark.gpu
SCRATCH = {"uflux", "vflux", "uflux2", "vflux2", "uflux3", "vflux3", "utotm", "vtotm", "util1"}


VARIANTS = [("full", "non_oscillatory"), ("full", "monotonic"), ("partial", "non_oscillatory"), ("partial", "monotonic")]


def _setup(cfg, compat="full", limiting="non_oscillatory", ntr=None):
    from oracle.refblom import get_ref_backend, have_ref
    from blom_amd.gpu import BlomGpu
    if not have_ref(cfg):
        pytest.skip(f"oracle/_ref/{cfg}/libblomref.so not built")
    case = make_case(cfg, ntr=ntr, advmth="cppm", cppm_compatibility=compat, cppm_limiting=limiting)
    ref = get_ref_backend(cfg, case.depth, ntr=ntr)
    hostinit.init_state(ref, case)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            gpu.set(nm, v)
    return case, ref, gpu


def test_advect_cppm_with_many_tracers():
    """cppm carrying 9 tracers, against the reference's own cppm carrying them (oracle/harness: ref_set_ntr)"""
    test_advect_cppm_stage_parity("chan_s_tke", 4, "full", "non_oscillatory", ntr=9)
    test_advect_cppm_stage_parity("tri_s_tke", 4, "partial", "monotonic", ntr=6)


@pytest.mark.parametrize("compat,limiting", VARIANTS)
@pytest.mark.parametrize("cfg,nsteps", [("chan_s", 6), ("box_s", 6), ("fuk95", 4), ("tri_s", 6), ("chan_s_tke", 6)])
def test_advect_cppm_stage_parity(cfg, nsteps, compat, limiting, ntr=None):
    case, ref, gpu = _setup(cfg, compat, limiting, ntr)
    failures, pending, nstep, ready = [], {}, [0], [False]

    def check():
        if pending.pop("st", None):
            fields = [f for f in STATE_FIELDS + INT_FIELDS if f not in SCRATCH]
            bad = diff_report(ref, gpu, fields=fields)
            if bad:
                failures.append(f"step {nstep[0] + 1} advect(cppm):\n" + fmt_report(bad))

    def hook(st, six):
        check()
        if st != "advect":
            return
        copy_state(ref, gpu)
        if not ready[0]:
            gpu.stage("init_cppm", *six)
            ready[0] = True
        gpu.set("nstep", nstep[0] + 1)
        gpu.set("delt1", ref.ref.get_real("delt1"))
        gpu.stage("advect", *six)
        pending["st"] = True

    for _ in range(nsteps):
        new = dyncore_step(ref, nstep[0], case.params["baclin"], hook=hook)
        check()
        nstep[0] = new
    gpu.close()
    from oracle.refblom import get_ref_backend           # leave the shared reference instance on 'remap', with its own tracers
    dflt = make_case(cfg)
    hostinit.init_state(get_ref_backend(cfg, dflt.depth), dflt)   #
    assert not failures, "\n".join(failures[:20])


@pytest.mark.parametrize("cfg,nsteps,rtol", [("chan_s", 8, 0.0), ("fuk95", 4, 0.0)])
def test_freerun_cppm(cfg, nsteps, rtol):
    case, ref, gpu = _setup(cfg)
    copy_state(ref, gpu)
    gpu.stage("init_cppm", 2, 1, case.kdm, 0, case.kdm + 1, 1)
    gpu.set("delt1", case.params["baclin"])
    ns = 0
    for _ in range(nsteps):
        ns = dyncore_step(ref, ns, case.params["baclin"])
    assert gpu.step(0, nsteps) == nsteps
    bad = diff_report(ref, gpu, fields=["u", "v", "dp", "temp", "saln", "trc", "uflx", "vflx", "utflx", "vsflx"], rtol=rtol, atol=rtol)
    gpu.close()
    hostinit.init_state(ref, make_case(cfg))
    assert not bad, fmt_report(bad)
