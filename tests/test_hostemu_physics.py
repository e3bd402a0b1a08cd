"""The physics added in round 4, on the CPU: the device library compiled for the host (tests/hostemu) against the reference's REAL
modules (oracle/_ref/*_xaln, *_xml: cross-check builds, see oracle/xcheck) -- neutral diffusion inside ale_regrid_remap, the hybrid
step with it, and config 2's step with thermf, mxlayr and the front of difest_isobml.  The same comparisons as the GPU suite's
tests/test_xcheck_ale.py, test_xcheck_hybrid_step.py and test_xcheck_fullstep.py (whose functions run here unchanged), made where
there is no GPU: bit for bit."""
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "hostemu", "libblomgpu_hostemu.so")
pytestmark = pytest.mark.skipif(not os.path.exists(EMU), reason="tests/hostemu/libblomgpu_hostemu.so not built")


@pytest.fixture()
def emu_lib():
    import blom_amd.gpu as g
    import blom_amd.hor3map as h
    old, oldh = g.LIB_PATH, h._LIB
    g.LIB_PATH = EMU
    h._LIB = EMU
    yield
    g.LIB_PATH = old
    h._LIB = oldh


@pytest.mark.parametrize("cfg,nsteps,spread,vcoord,align", [("chan_s", 4, 0.5, "nudge", 1), ("tri_s", 3, 0.3, "nudge", 1), ("box_s", 4, 0.8, "cntiso_hybrid", 0)])
def test_neutral_diffusion_on_the_host_emulation(emu_lib, cfg, nsteps, spread, vcoord, align, tmp_path):
    from test_xcheck_ale import _ale_regrid_remap_check
    _ale_regrid_remap_check(cfg, nsteps, spread, vcoord, tmp_path, ndiff=align)


def test_hybrid_step_with_neutral_diffusion_on_the_host_emulation(emu_lib, tmp_path):
    from test_xcheck_hybrid_step import _hybrid_step_check
    _hybrid_step_check("chan_s", "remap", "nudge", "cntiso_hybrid", 4, tmp_path, neutral=True)


def test_full_physics_step_on_the_host_emulation(emu_lib):
    from test_xcheck_fullstep import _full_step_check
    _full_step_check("chan_s_tke", 5, False)


def test_difest_isobml_diffusivity_estimates_on_the_host_emulation(emu_lib):
    """the cross-checks of tests/test_xcheck_difest.py (GPU suite) on the emulation: the stage on its own and config 2's step with live diffusivities"""
    from test_xcheck_difest import _live_step_check, OPT_FUK95, OPT_SHEAR, OPT_2D
    _live_step_check("chan_s_tke", 3, OPT_FUK95, stagewise=True)
    _live_step_check("box_s", 3, OPT_SHEAR, stagewise=True)
    _live_step_check("tri_s_tke", 3, OPT_2D, stagewise=True)
    _live_step_check("chan_s_tke", 4, OPT_FUK95)
    # NorESM's defaults, rhsctp = .true. included (round 6: the topographic Rhines scale through sin_libm.h / atan2_libm.h)
    from blom_amd import hostinit
    _live_step_check("chan_s_tke", 3, hostinit.DIFEST_NORESM, stagewise=True)
    _live_step_check("box_s", 4, hostinit.DIFEST_NORESM)
    from test_xcheck_difest import test_rhsctp_acts_on_the_layer_interface_diffusivities as rhs_effect
    rhs_effect("chan_s_tke")
    # the two-equation closure (use_GLS; round 6): the stage on its own and the whole step against the -DGLS build
    _live_step_check("chan_s_tk2", 3, OPT_SHEAR, stagewise=True)
    _live_step_check("chan_s_tk2", 4, hostinit.DIFEST_NORESM)
