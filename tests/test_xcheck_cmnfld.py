"""CROSS-CHECK (not a pin) of cmnfld1 / cmnfld2 against the reference's REAL phy/mod_cmnfld_routines.F90.

mod_cmnfld_routines imports diagnostic-request flags from mod_dia, which needs netCDF (absent here), so the oracle's
reference builds leave cmnfld out and oracle/c/cmnfld.c stays "parity unpinned".  The *_xed builds of oracle/Makefile compile
the reference's own mod_cmnfld_routines.F90 against oracle/xcheck/mod_dia_standin.F90 -- a stand-in module holding those
flags, all zero (no diagnostics requested).  Because a stand-in is involved this does not lift the "unpinned" label
(DESIGN.md); it replaces "restatement and kernels by the same author agree" by "both reproduce the reference's own compiled
arithmetic", bit for bit: buoyancy frequency (interface, layer, filtered), neutral slopes, geopotential, interface depths,
and -- with the real mod_eddtra of the same builds -- the whole stage sequence with cmnfld2 AND eddtra live, which is the
sequence bench.py times."""
import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step, DYNCORE_STAGES
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, INT_FIELDS, GRID_FIELDS
from test_xcheck_eddtra import _WithEddtra

LIVE = tuple("cmnfld2" if s == "halo_cmnfld2" else s for s in DYNCORE_STAGES)
CMN = ["bfsqi", "bfsql", "bfsqf", "nslpx", "nslpy", "nnslpx", "nnslpy"]


def _reference(cfg):
    from oracle.refblom import get_ref_backend, have_ref
    if not have_ref(cfg + "_xed"):
        pytest.skip(f"oracle/_ref/{cfg}_xed/libblomref.so not built")
    case = make_case(cfg, nslp0=0.0, eitmth="gm")
    ref = _WithEddtra(get_ref_backend(cfg + "_xed", case.depth))
    ref.ref.set("eitmth", "gm")
    hostinit.init_state(ref, case)
    return case, ref


@pytest.mark.parametrize("cfg", ["chan_s", "box_s", "fuk95"])
def test_restatement_equals_the_real_cmnfld_routines_with_live_slopes(cfg):
    """reference (real cmnfld2 + real eddtra) against the C restatement, both stepping the whole sequence on the slopes of
    the evolving state; every field after every step, cmnfld's own fields included; then cmnfld1's z, dz"""
    from oracle.coracle import COracle, have_coracle
    if not have_coracle():
        pytest.skip("C oracle not built")
    case, ref = _reference(cfg)
    co = COracle(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            co.set(nm, v)
    copy_state(ref, co, fields=STATE_FIELDS + INT_FIELDS + GRID_FIELDS + CMN)
    co.set("delt1", case.params["baclin"])
    fields = [f for f in STATE_FIELDS + INT_FIELDS if f not in ("util1", "util2")] + CMN
    nr = nc = 0
    amp = 0.0
    for _ in range(6):
        nr = dyncore_step(ref, nr, case.params["baclin"], stages=LIVE)
        nc = dyncore_step(co, nc, case.params["baclin"], stages=LIVE)
        bad = diff_report(ref, co, fields=fields)
        assert not bad, f"step {nr}\n" + fmt_report(bad[:8])
        a = ref.get("nslpx")[:, 4:-4, 4:-4]
        amp = max(amp, float(np.abs(a[np.abs(a) < 1e30]).max()))
    assert amp > 0.0, "the slopes stayed zero"
    six = hostinit.step_indices(nr, case.kdm)
    copy_state(ref, co, fields=["z", "dz"])               # the points cmnfld1 does not touch keep the inivar pattern
    ref.stage("cmnfld1", *six)
    co.stage("cmnfld1", *six)
    bad = diff_report(ref, co, fields=["z", "dz"])
    assert not bad, fmt_report(bad)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["chan_s", "box_s", "fuk95"])
def test_device_sequence_with_live_slopes_equals_the_reference(cfg):
    """blomgpu_step with live_slopes (cmnfld2 and eddtra inside the device-resident sequence, what bench.py times) against
    the reference stepping the same sequence with its real mod_cmnfld_routines and mod_eddtra"""
    from blom_amd.gpu import BlomGpu
    case, ref = _reference(cfg)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            gpu.set(nm, v)
    copy_state(ref, gpu, fields=STATE_FIELDS + INT_FIELDS + GRID_FIELDS + CMN)
    gpu.set("delt1", case.params["baclin"])
    gpu.set("live_slopes", 1)
    nsteps, ns = 12, 0
    for _ in range(nsteps):
        ns = dyncore_step(ref, ns, case.params["baclin"], stages=LIVE)
    assert gpu.step(0, nsteps) == nsteps
    gpu.sync()
    bad = diff_report(ref, gpu, fields=["u", "v", "dp", "temp", "saln", "sigma", "pb", "ub", "vb", "trc", "uflx", "vflx",
                                        "umfltd", "vmfltd", "nslpx", "nslpy", "bfsqf", "dpu", "dpv"])
    gpu.close()
    assert not bad, fmt_report(bad)


@pytest.mark.gpu
def test_the_bench_sequence_at_full_size_equals_the_reference_with_its_real_cmnfld_and_eddtra():
    """BASELINE.json's channel at full size (208x512x53, ntr = 3), the sequence bench.py times -- cmnfld2's slopes from the
    evolving state, eddtra on them, advect/remap on eddtra's fluxes and everything downstream -- device-resident against
    the reference's own Fortran with its real mod_cmnfld_routines and mod_eddtra (channel_tke_omp_xed: OpenMP; stand-ins
    for mod_dia's flags and mod_difest's one array), four steps.  Bit for bit.  Cross-check, not a pin."""
    import os
    import threading
    from oracle.refblom import get_ref_backend, have_ref
    from blom_amd.gpu import BlomGpu
    if not have_ref("channel_tke_omp_xed"):
        pytest.skip("oracle/_ref/channel_tke_omp_xed/libblomref.so not built")
    nsteps, res = 4, {}

    def body():
        case = make_case("channel_tke", nslp0=0.0)
        ref = _WithEddtra(get_ref_backend("channel_tke_omp_xed", case.depth))
        ref.ref.set("eitmth", "gm")
        hostinit.init_state(ref, case)
        gpu = BlomGpu(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
        for nm, v in case.params.items():
            if not nm.endswith("0"):
                gpu.set(nm, v)
        copy_state(ref, gpu, fields=STATE_FIELDS + INT_FIELDS + GRID_FIELDS + CMN)
        gpu.set("delt1", case.params["baclin"])
        gpu.set("live_slopes", 1)
        ns = 0
        for _ in range(nsteps):
            ns = dyncore_step(ref, ns, case.params["baclin"], stages=LIVE)
        assert gpu.step(0, nsteps) == nsteps
        gpu.sync()
        res["bad"] = diff_report(ref, gpu, fields=["u", "v", "dp", "temp", "saln", "sigma", "pb", "ub", "vb", "trc", "uflx", "vflx",
                                                   "umfltd", "vmfltd", "nslpx", "nslpy", "bfsqf", "dpu", "dpv"])
        um = ref.get("umfltd")[:, 4:-4, 4:-4]
        res["nonzero"] = int(np.count_nonzero(um * (np.abs(um) < 1e30)))
        gpu.close()

    os.environ["OMP_NUM_THREADS"] = str(min(16, os.cpu_count() or 1))
    os.environ["OMP_STACKSIZE"] = "1G"
    threading.stack_size(2 << 30)
    th = threading.Thread(target=body)
    th.start()
    th.join()
    threading.stack_size(0)
    assert "bad" in res, "the comparison did not complete"
    assert res["nonzero"] > 100000
    assert not res["bad"], fmt_report(res["bad"])
