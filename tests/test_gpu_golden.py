"""The device against the COMMITTED golden fixtures alone (no reference library needed on the GPU box):
tests/golden/<cfg>_init.npz is the input state, <cfg>_crc.json the reference's own checksum of every field
after every stage, <cfg>_final.npz the reference's fields after three steps.

  * after every stage of every recorded step the device's fields must have the reference's CRCs;
  * at the end the device's fields must equal the reference's final fields.
Both bit for bit (exp() included: blom_amd/csrc/exp_libm.h)."""
import json
import os

import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd.checksum import chksum
from blom_amd.stepper import dyncore_step, DYNCORE_STAGES, STAGES_FROZEN_EDDY_FLUXES
from parity import load_golden_init, put_fields

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
EDDTRA_OUT = {"umfltd", "vmfltd", "utfltd", "vtfltd", "usfltd", "vsfltd"}     # see tests/test_oracle_golden.py


@pytest.mark.parametrize("cfg", ["chan_s", "box_s"])
def test_device_reproduces_golden_fixtures(cfg):
    from blom_amd.gpu import BlomGpu
    case = make_case(cfg)
    masks, fields = load_golden_init(cfg)
    gold = json.load(open(os.path.join(HERE, "golden", f"{cfg}_crc.json")))
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, case.nreg, masks)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            gpu.set(nm, v)
    put_fields(gpu, fields)
    gpu.set("delt1", case.params["baclin"])
    bad, state = [], {"step": 1, "checked": 0}

    def check(st):
        exp = gold["crc"].get(str(state["step"]), {}).get(st)
        if exp is None:
            return
        for nm, want in exp.items():
            if nm in EDDTRA_OUT or not gpu.has_field(nm):
                continue
            got = chksum(nm, gpu.get(nm), masks, case.idm, case.jdm)
            state["checked"] += 1
            if got != want:
                bad.append(f"step {state['step']} {st} {nm}: crc 0x{got:08x} != 0x{want:08x}")

    pending = []

    def hook(st, six):
        if pending:
            check(pending.pop())
        pending.append(st)
    ns = dyncore_step(gpu, 0, case.params["baclin"], hook=hook)
    check(pending.pop())
    assert not bad, "\\n".join(bad[:20])
    assert state["checked"] > 500
    for _ in range(gold["nsteps"] - 1):
        ns = dyncore_step(gpu, ns, case.params["baclin"])
    z = np.load(os.path.join(HERE, "golden", f"{cfg}_final.npz"))
    J, I = slice(4, 4 + case.jdm), slice(4, 4 + case.idm)
    for nm in z.files:
        a = z[nm][:, J, I]
        b = gpu.get(nm)[:a.shape[0], J, I]
        assert np.array_equal(a, b, equal_nan=True), (nm, float(np.abs(a - b).max()))
    gpu.close()


@pytest.mark.parametrize("cfg", ["fuk95", "fuk95_ref", "tri_s", "chan_s_tke", "channel_tke", "tnx2v1s_tke",
                                 "tri_s+edf", "chan_s_tke+edf", "channel_tke+edf", "tnx2v1s_tke+edf",
                                 "tnx1v4s_tke+edf", "tnx1v4s_tke@24+edf"])
def test_device_reproduces_reference_checksums_from_analytic_init(cfg):
    """Fixtures that hold only the reference's per-stage checksums; the inputs are the analytic host initialisation.
    channel_tke is BASELINE.json's channel at full size (208x512x53, ntr = 3, the bench workload): the device must
    produce the checksums the reference's own Fortran produced for every recorded field after every stage of three
    steps; tnx2v1s_tke has the dimensions of the tnx2v1 production grid (180x193x53) with the arctic patch.  The checksums are taken on the device (blomgpu_crc = xccrc, phy/mod_xc.F90:4164)."""
    from blom_amd.gpu import BlomGpu
    from blom_amd import hostinit
    from blom_amd.checksum import grid_of
    # <cfg>+edf: the same run with the frozen synthetic eddy-induced mass fluxes of hostinit.frozen_eddy_fluxes in front
    # of advect (the bench workload has eddtra's there; the fixture was made by writing them into the reference's arrays)
    # <cfg>@N: with N tracers (tnx1v4s_tke@24: BASELINE.json's config 5 at its size, the extra tracers standing in for iHAMOCC's)
    eddy = cfg.endswith("+edf")
    cfg = cfg[:-4] if eddy else cfg
    ntr = None
    if "@" in cfg:
        cfg, n = cfg.split("@")
        ntr = int(n)
    case = make_case(cfg, ntr=ntr)
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm, arctic=case.nreg == 2)
    gold = json.load(open(os.path.join(HERE, "golden", cfg + (f"_n{ntr}" if ntr else "") + ("_edf" if eddy else "") + "_crc.json")))
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    hostinit.init_state(gpu, case)
    if eddy:
        hostinit.frozen_eddy_fluxes(gpu, case)
    stages = STAGES_FROZEN_EDDY_FLUXES if eddy else DYNCORE_STAGES
    bad, state = [], {"checked": 0}

    def check(st):
        exp = gold["crc"][str(state["step"])].get(st)
        if exp is None:
            return
        if st == "pgforc":
            state["old_set"] = True
        for nm, want in exp.items():
            # eddtra: see tests/test_oracle_golden.py; *_o: first written by pgforc (phy/mod_pgforc.F90:487-522)
            if (nm in EDDTRA_OUT and not eddy) or not gpu.has_field(nm) or (nm.endswith("_o") and not state.get("old_set")):
                continue
            got = gpu.crc(nm, 1, gpu.field_info(nm)[0], grid_of(nm))
            state["checked"] += 1
            if got != want:
                bad.append(f"step {state['step']} {st} {nm}: crc 0x{got:08x} != 0x{want:08x}")

    ns = 0
    for _ in range(gold["nsteps"]):
        state["step"] = ns + 1
        pending = []

        def hook(st, six):
            if pending:
                check(pending.pop())
            pending.append(st)
        ns = dyncore_step(gpu, ns, case.params["baclin"], hook=hook, stages=stages)
        check(pending.pop())
    gpu.close()
    assert not bad, "\n".join(bad[:20])
    assert state["checked"] > 600


@pytest.mark.parametrize("golden,rhsctp", [("channel_tke_live_long_crc.json", 1), ("channel_tke_live_long_rhsctp0_crc.json", 0)])
def test_600_steps_of_the_bench_workload_equal_the_reference_long_run(golden, rhsctp):
    """The long-run question of round 5's review: config 2's step with live diffusivities (what bench.py times) for 600 steps from the
    bench's initial state, device-resident (blomgpu_step), against the reference's own modules run for the same 600 steps in the build
    container (tools/longrun_reference.py -> tests/golden/channel_tke_live_long_crc.json): xccrc of dp, temp, saln, u, v, the tracers,
    difint and difdia over both time levels at steps 100, 200, ..., 600, and the extremes of temp with the cells they are taken in.
    Twice: with NorESM's defaults as bench.py runs them now (rhsctp = .true.), and with rhsctp off -- the options of round 5, whose
    -111 degC sample at step 600 this run settled (profiles/r06_longrun.txt)."""
    import sys
    sys.path.insert(0, os.path.join(HERE, ".."))
    sys.path.insert(0, os.path.join(HERE, "..", "tools"))
    import bench
    from blom_amd.checksum import grid_of
    from longrun_reference import sample
    path = os.path.join(HERE, "golden", golden)
    gold = json.load(open(path))
    trace = {t["step"]: t for t in gold["trace"]}
    case, nreg, masks = bench.build_case("channel", "remap", "default")
    gpu = bench.device_for_bench(case, nreg, masks, live=True)
    gpu.set("rhsctp", rhsctp)
    scp2 = gpu.get("scp2")[0][4:-4, 4:-4]
    bad, ns = [], 0
    try:
        for step in sorted(int(s) for s in gold["crc"]):
            ns = gpu.step(ns, step - ns)
            assert ns == step
            for nm, want in gold["crc"][str(step)].items():
                got = gpu.crc(nm, 1, gpu.field_info(nm)[0], grid_of(nm))
                if got != want:
                    bad.append(f"step {step} {nm}: crc 0x{got:08x} != 0x{want:08x}")
            smp, g = sample(gpu, case, masks, ns, scp2), trace[step]
            for k_ in ("tmin", "tmax"):
                if list(smp[k_]) != list(g[k_]):
                    bad.append(f"step {step} {k_}: {smp[k_]} != {g[k_]}")
            for k_ in ("mass", "heat", "salt"):
                if smp[k_] != g[k_]:
                    bad.append(f"step {step} {k_}: {smp[k_]!r} != {g[k_]!r}")
            if bad:
                break
    finally:
        gpu.close()
    assert not bad, "\n".join(bad[:20])
    assert ns >= 600
