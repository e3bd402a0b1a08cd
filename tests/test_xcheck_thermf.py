"""CROSS-CHECK (not a pin) of the device's thermf against the reference's REAL channel/mod_thermf_channel.F90.

The module imports two integers of the netCDF-bound mod_ben02, so it is compiled against oracle/xcheck/mod_ben02_standin.F90 in
the *_xml builds of oracle/Makefile: a cross-check, not a pin.  The channel experiment's own forcing is zero except for the
friction velocity (channel/mod_channel.F90:365-388); here the forcing fields, the climatologies and the relaxation options are
synthetic and non-trivial on both sides, so that every term of the routine acts: fresh water and virtual salt flux with its
global correction (two xcsum), heat flux, tracer fluxes, SST and SSS relaxation with the cubic interpolation in time."""
import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, GRID_FIELDS, INT_FIELDS

pytestmark = pytest.mark.gpu
OUT = ["surflx", "sswflx", "surrlx", "salflx", "brnflx", "salrlx", "trflx", "ustar", "fmltfz", "sfl", "hmltfz", "util1", "util2", "util3", "util4"]
IN2 = ["swa", "nsf", "eva", "lip", "sop", "rnf", "rfi", "ustarw"]


@pytest.mark.parametrize("relax", [True, False], ids=["relaxation", "no_relaxation"])
@pytest.mark.parametrize("cfg,nsteps", [("chan_s_tke", 3), ("box_s", 2), ("tri_s_tke", 3)])
def test_device_thermf_equals_the_real_module(cfg, nsteps, relax):
    from oracle.refblom import get_ref_backend, have_ref
    from blom_amd.gpu import BlomGpu
    lib = cfg.replace("_tke", "") + "_xml"
    if not have_ref(lib):
        pytest.skip(f"oracle/_ref/{lib}/libblomref.so not built")
    case = make_case(cfg)
    ref = get_ref_backend(lib, case.depth)
    kk, nj, ni = case.kdm, case.jdm + 8, case.idm + 8
    gpu = BlomGpu(case.idm, case.jdm, kk, ref.ntr, ref.nreg, ref.masks)
    hostinit.init_state(gpu, case)
    assert gpu.step(0, nsteps) == nsteps
    hostinit.init_state(ref, case)
    copy_state(gpu, ref, fields=STATE_FIELDS + GRID_FIELDS + INT_FIELDS)
    rng = np.random.default_rng(5 + nsteps)
    f = {"swa": rng.uniform(0.0, 300.0, (1, nj, ni)), "nsf": rng.uniform(-250.0, 50.0, (1, nj, ni)),
         "eva": -rng.uniform(0.0, 8e-5, (1, nj, ni)), "lip": rng.uniform(0.0, 1e-4, (1, nj, ni)), "sop": rng.uniform(0.0, 2e-5, (1, nj, ni)),
         "rnf": rng.uniform(0.0, 3e-5, (1, nj, ni)), "rfi": rng.uniform(0.0, 1e-5, (1, nj, ni)), "ustarw": rng.uniform(1e-3, 2e-2, (1, nj, ni)),
         "sstclm": rng.uniform(-1.0, 25.0, (12, nj, ni)), "ricclm": np.clip(rng.uniform(-0.5, 0.7, (12, nj, ni)), 0.0, 1.0),
         "sssclm": rng.uniform(32.0, 37.0, (12, nj, ni))}
    for be in (ref, gpu):
        for nm, a in f.items():
            be.put(nm, a)
        for nm in OUT:                                       # the outputs start from one pattern on both sides
            if nm != "trflx":
                be.put(nm, np.full((1, nj, ni), 7.0))
        if ref.ntr:
            be.put("trflx", np.full((ref.ntr, nj, ni), 7.0))
    wet = ref.masks["ip"][4:-4, 4:-4] > 0
    area = float(ref.get("scp2")[0, 4:-4, 4:-4][wet].sum())
    opts = dict(trxday=30.0 if relax else 0.0, srxday=60.0 if relax else 0.0, trxdpt=1.0, srxdpt=0.7, trxlim=1.5, srxlim=0.5, sref=34.65,
                area=area, xmi=0.37)
    ints = dict(l1mi=3, l2mi=4, l3mi=5, l4mi=6, l5mi=7, aptflx=0, apsflx=0, ditflx=0, disflx=0, srxbal=0)
    for nm, v in opts.items():
        ref.ref.set(nm, float(v))
        gpu.set(nm, float(v))
    for nm, v in ints.items():
        ref.ref.set(nm, int(v))
        gpu.set(nm, int(v))
    for nm, v in dict(nstep=nsteps + 1, nstep_in_day=96, nday_of_year=40, nday_in_year=365).items():   # the reference's position in the year (for its unused weights)
        ref.ref.set(nm, int(v))
    six = hostinit.step_indices(nsteps, kk)
    try:
        ref.ref.stage("thermf", *six)
        gpu.stage("thermf", *six)
        bad = diff_report(ref, gpu, fields=OUT)
        assert not bad, fmt_report(bad[:12])
        s = gpu.get("salflx")[0, 4:-4, 4:-4][wet]
        assert np.isfinite(s).all() and np.abs(s).max() > 0.0
        if relax:
            assert np.abs(gpu.get("surrlx")[0, 4:-4, 4:-4][wet]).max() > 0.0 and np.abs(gpu.get("salrlx")[0, 4:-4, 4:-4][wet]).max() > 0.0
    finally:
        for nm in ("trxday", "srxday"):
            ref.ref.set(nm, 0.0)
        gpu.close()


def test_thermf_follows_the_reference_switch():
    """phy/mod_thermf.F90:43-63: nothing for the experiments without forcing, an error for unknown ones; what is not built fails loudly"""
    from blom_amd.gpu import BlomGpu, BlomGpuError
    case = make_case("chan_s")
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    hostinit.init_state(gpu, case)
    six = hostinit.step_indices(0, case.kdm)
    for e in ("fuk95", "noforcing", "isomip1"):
        gpu.set("expcnf", e)
        gpu.stage("thermf", *six)
    gpu.set("expcnf", "nonsense")
    with pytest.raises(BlomGpuError, match="thermf: expcnf = nonsense is unsupported!"):
        gpu.stage("thermf", *six)
    gpu.set("expcnf", "cesm")
    with pytest.raises(BlomGpuError, match="not built on the device"):
        gpu.stage("thermf", *six)
    gpu.set("expcnf", "channel")
    with pytest.raises(BlomGpuError, match="area"):
        gpu.stage("thermf", *six)
    gpu.set("area", 1.0e12)
    gpu.set("ditflx", 1)
    with pytest.raises(BlomGpuError, match="not built on the device"):
        gpu.stage("thermf", *six)
    gpu.close()
