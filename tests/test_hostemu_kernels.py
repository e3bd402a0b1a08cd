"""The device kernels on the CPU: the device library compiled for the host (tests/hostemu: every kernel runs thread by
thread, one fiber per thread, workgroup barriers and LDS as on the device, dynamic LDS poisoned with NaN) stepped through the
stage sequence -- the LDS-tiled remap, pbcor and diffus kernels, the pipelined diapfl column pass, the fused momtum marches,
the barotropic pair kernel against one kernel per equation -- and against the C restatement, also with more tracers than one
batch of the tile kernels holds.  The GPU suite makes the comparisons on the device; this one runs where there is no GPU."""
import os

import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step
from parity import STATE_FIELDS

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "hostemu", "libblomgpu_hostemu.so")
pytestmark = pytest.mark.skipif(not os.path.exists(EMU), reason="tests/hostemu/libblomgpu_hostemu.so not built")

OLD = dict(barotp_fused=0, barotp_persist=0)
SKIP = {"util1", "util2", "util3", "util4"}


@pytest.fixture()
def emu_lib():
    import blom_amd.gpu as g
    old = g.LIB_PATH
    g.LIB_PATH = EMU
    yield
    g.LIB_PATH = old


def _run(cfg, nsteps, ntr=None, **opts):
    from blom_amd.gpu import BlomGpu
    case = make_case(cfg, ntr=ntr)
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm, arctic=case.nreg == 2)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    hostinit.init_state(gpu, case)
    for k, v in opts.items():
        gpu.set(k, v)
    if "barotp_block" in opts and opts["barotp_block"]:
        assert gpu.get_real("barotp_block_mode") == opts["barotp_block"], "the blocked barotp kernel is not usable on this case"
    assert gpu.step(0, nsteps) == nsteps
    out = {nm: gpu.get(nm) for nm in STATE_FIELDS if gpu.has_field(nm)}
    gpu.close()
    return case, out


@pytest.mark.parametrize("cfg,nsteps", [("chan_s", 4), ("box_s", 3), ("tri_s", 3)])
def test_pair_kernel_equals_one_kernel_per_equation(emu_lib, cfg, nsteps):
    _, new = _run(cfg, nsteps)
    _, old = _run(cfg, nsteps, **OLD)
    bad = [nm for nm in new if nm not in SKIP and not np.array_equal(new[nm], old[nm], equal_nan=True)]
    assert not bad, bad
    assert np.isfinite(new["u"]).all() and np.abs(new["u"]).max() > 0.0


@pytest.mark.parametrize("cfg,nsteps", [("chan_s", 4), ("per_s", 3), ("fuk95", 2), ("chan_b", 3)])
def test_four_substeps_per_tile_launch_equal_one_kernel_per_equation(emu_lib, cfg, nsteps):
    """k_bt_steps4 (temporal blocking: rim of 4 / 6 cells, every point identified with its home point) with one launch per
    four substeps -- the form the host emulation can run; the device's default is the same kernel walking a whole phase"""
    _, new = _run(cfg, nsteps, barotp_block=2)
    _, old = _run(cfg, nsteps, **OLD)
    bad = [nm for nm in new if nm not in SKIP and not np.array_equal(new[nm], old[nm], equal_nan=True)]
    assert not bad, bad
    assert np.isfinite(new["u"]).all() and np.abs(new["u"]).max() > 0.0


@pytest.mark.parametrize("cfg,ntr", [("chan_s", None), ("box_s", None), ("tri_s", None), ("chan_s_tke", 11), ("tri_s_tke", 6)])
def test_current_kernels_equal_the_c_restatement(emu_lib, cfg, ntr):
    """the same sequence on the C restatement (pinned on the compiled reference, tests/test_oracle_vs_reference.py, also with
    the reference carrying 11 tracers); ntr = 11, 6: more than the 4 tracers one batch of the tile kernels holds"""
    from oracle.coracle import COracle
    nsteps = 3
    case, new = _run(cfg, nsteps, ntr=ntr)
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm, arctic=case.nreg == 2)
    orc = COracle(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    hostinit.init_state(orc, case)
    ns = 0
    for _ in range(nsteps):
        ns = dyncore_step(orc, ns, case.params["baclin"])
    J, I = slice(4, 4 + case.jdm), slice(4, 4 + case.idm)
    for nm in ("dp", "temp", "saln", "u", "v", "pb", "ub", "vb", "uflx", "vflx", "trc"):
        a, b = orc.get(nm)[..., J, I], new[nm][..., J, I]
        assert np.array_equal(a, b, equal_nan=True), nm
