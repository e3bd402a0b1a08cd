"""The Fortran host (blom_amd/fortran: ISO_C_BINDING shim + `program blom_dyncore`) drives the
same C-ABI: its end-of-run `chksum:` lines (cf. drivers/nocoupler/blom.F:56-57) must equal the
checksums of the Python-driven device run on the same inputs."""
import os
import re
import subprocess

import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.statefile import write_state
from parity import STATE_FIELDS, GRID_FIELDS, INT_FIELDS, load_golden_init, put_fields

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fortran_driver_matches_python_driver(tmp_path):
    from blom_amd.gpu import BlomGpu
    exe = os.path.join(ROOT, "blom_amd", "lib", "blom_dyncore")
    if not os.path.exists(exe):
        pytest.skip("Fortran driver not built")
    cfg, nsteps = "chan_s", 5
    case = make_case(cfg)
    masks, fields = load_golden_init(cfg)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, case.nreg, masks)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            gpu.set(nm, v)
    put_fields(gpu, fields)
    names = [n for n in STATE_FIELDS + GRID_FIELDS + INT_FIELDS if gpu.has_field(n) and n in fields]
    state = str(tmp_path / "blom_state.bin")
    write_state(state, gpu, case, nsteps, names)
    gpu.set("delt1", case.params["baclin"])
    assert gpu.step(0, nsteps) == nsteps
    want = {"dp": gpu.crc("dp", 1, 2 * case.kdm, 1), "temp": gpu.crc("temp", 1, 2 * case.kdm, 1),
            "u": gpu.crc("u", 1, 2 * case.kdm, 13)}
    gpu.close()
    out = subprocess.run([exe, state], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    got = {m.group(1): int(m.group(2), 16) for m in re.finditer(r"chksum: (\w+): 0x([0-9A-Fa-f]+)", out.stdout)}
    assert got == want, (got, want, out.stdout)
    assert open(tmp_path / "run.status").read().strip() == "success"


def test_fortran_driver_runs_the_full_physics_sequence(tmp_path):
    """config 2's step as far as built (thermf, mxlayr, the front of difest_isobml, cmnfld2, cmnfld1), sequenced
    stage by stage by the Fortran host (option record full_physics = 1 of the state file) against blomgpu_step with the same
    option: equal checksums; see tests/test_fortran_namelist_host.py"""
    from blom_amd.gpu import BlomGpu
    from test_fortran_namelist_host import run_full_physics
    exe = os.path.join(ROOT, "blom_amd", "lib", "blom_dyncore")
    if not os.path.exists(exe):
        pytest.skip("Fortran driver not built")
    run_full_physics(tmp_path, exe, BlomGpu)


def test_fortran_driver_estimates_the_diffusivities_when_difest_live_is_set(tmp_path):
    """gpu_set('difest_live', 1) in the Fortran host makes the shim's difest_isobml the whole routine (mod_blomgpu: difest_estimates,
    public since round 6): the host-sequenced step reproduces blomgpu_step's checksums with live diffusivities, difint included"""
    from blom_amd.gpu import BlomGpu
    from test_fortran_namelist_host import run_full_physics
    exe = os.path.join(ROOT, "blom_amd", "lib", "blom_dyncore")
    if not os.path.exists(exe):
        pytest.skip("Fortran driver not built")
    run_full_physics(tmp_path, exe, BlomGpu, live=True)


def test_fortran_hor3map_shim(tmp_path):
    """blom_amd/fortran/mod_hor3map_gpu.F90 (the reference's mod_hor3map names over the C ABI),
    driven by h3m_demo, against the Python binding on the same analytic columns"""
    import numpy as np
    from blom_amd import hor3map as h3
    exe = os.path.join(ROOT, "blom_amd", "lib", "h3m_demo")
    if not os.path.exists(exe):
        pytest.skip("Fortran hor3map demo not built")
    out = subprocess.run([exe], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    got = {m.group(1): float(m.group(2)) for m in re.finditer(r"(sum_\w+):\s+(\S+)", out.stdout)}
    assert "errstat 3: Source grid edges do not monotonically increase or decrease!" in out.stdout
    ncol, ns, nd = 96, 20, 15
    i = np.arange(1, ncol + 1)[:, None]
    k = np.arange(1, ns + 1)[None, :]
    h = np.where((k + i) % 5 == 0, 0.0, 9806.0 * (1.0 + 0.5 * np.sin(0.37 * k + 0.11 * i)))
    p_src = np.concatenate([np.zeros((ncol, 1)), np.cumsum(h, 1)], 1)
    trc = 1.0 + np.sin(0.5 * k + 0.05 * i)
    p_dst = p_src[:, -1:] * (np.arange(nd + 1)[None, :] / nd) ** 2
    p_dst[:, -1] = p_src[:, -1]
    g = h3.ReconGrid(ncol, ns, h3.PPM, 6, 4)
    s = h3.ReconSrc(g, h3.NON_OSCILLATORY_POSDEF, True, False)
    r = h3.Remap(g, nd)
    g.prepare_reconstruction(np.ascontiguousarray(p_src))
    s.reconstruct(np.ascontiguousarray(trc))
    pc = s.extract_polycoeff()
    r.prepare_remapping(np.ascontiguousarray(p_dst))
    ud = r.remap(s)
    g.free()
    # inputs are recomputed with another libm (sin): equal to rounding, not bitwise
    assert abs(got["sum_polycoeff"] - pc.sum()) <= 1e-9 * abs(pc).sum()
    assert abs(got["sum_remapped"] - ud.sum()) <= 1e-9 * abs(ud).sum()


def test_fortran_driver_reproduces_reference_checksums_at_full_size(tmp_path):
    """The north-star configuration end to end: a Fortran host program (blom_amd/fortran/blom_dyncore.F90, the shape
    of the reference's drivers/nocoupler/blom.F) drives the device through the ISO_C_BINDING shim on BASELINE.json's
    channel (208x512x53, default tracer set) for three steps and prints, with the reference's chksum line format, the
    checksums the REFERENCE's own Fortran produced (tests/golden/channel_tke_crc.json, after tmsmt2 of step 3)."""
    import json
    import shutil
    import numpy as np
    from blom_amd.gpu import BlomGpu
    from blom_amd import hostinit
    exe = os.path.join(ROOT, "blom_amd", "lib", "blom_dyncore")
    if not os.path.exists(exe):
        pytest.skip("Fortran driver not built")
    if shutil.disk_usage(str(tmp_path)).free < 16 << 30:
        pytest.skip("less than 16 GiB free for the state file")
    cfg = "channel_tke"
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", f"{cfg}_crc.json")))
    nsteps = gold["nsteps"]
    case = make_case(cfg)
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    hostinit.init_state(gpu, case)
    # the device zero-initialises its arrays: only fields the initialisation filled need to travel
    names = [n for n in STATE_FIELDS + GRID_FIELDS + INT_FIELDS if gpu.has_field(n) and np.any(gpu.get(n))]
    state = str(tmp_path / "blom_state.bin")
    write_state(state, gpu, case, nsteps, names)
    gpu.close()
    out = subprocess.run([exe, state], cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    os.remove(state)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    got = {m.group(1): int(m.group(2), 16) for m in re.finditer(r"chksum: (\w+): 0x([0-9A-Fa-f]+)", out.stdout)}
    want = {nm: gold["crc"][str(nsteps)]["tmsmt2"][nm] for nm in ("dp", "temp", "u")}
    assert got == want, (got, want)
    assert open(tmp_path / "run.status").read().strip() == "success"


def test_namelist_file_drives_the_fortran_host_for_a_model_day(tmp_path):
    """the reference's tests/fuk95/limits (isopyc_bulkml variant) as the only source of options: one model day, 480 steps,
    of the reference's own test case restated from its generator (fuk95_ref); see tests/test_fortran_namelist_host.py"""
    from blom_amd.gpu import BlomGpu
    from test_fortran_namelist_host import run_case
    exe = os.path.join(ROOT, "blom_amd", "lib", "blom_dyncore")
    if not os.path.exists(exe):
        pytest.skip("Fortran driver not built")
    run_case(tmp_path, exe, BlomGpu, 480)


@pytest.mark.parametrize("neutral", [False, True])
def test_reference_limits_file_drives_the_hybrid_step_on_the_device(tmp_path, neutral):
    """the reference's tests/fuk95/limits with its own vertical coordinate (cntiso_hybrid, cppm, &ALE_REGRID_REMAP / nudge): the
    Fortran host takes the hybrid branch of blom_step; see tests/test_fortran_namelist_host.py.  neutral: with LTEDTP = 'neutral'
    and NDIFF_SURFACE_ALIGN = .true. in &DIFFUSION (neutral diffusion inside ale_regrid_remap)"""
    from blom_amd.gpu import BlomGpu
    from test_fortran_namelist_host import run_case_hybrid
    exe = os.path.join(ROOT, "blom_amd", "lib", "blom_dyncore")
    if not os.path.exists(exe):
        pytest.skip("Fortran driver not built")
    run_case_hybrid(tmp_path, exe, BlomGpu, 20, neutral)
