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
