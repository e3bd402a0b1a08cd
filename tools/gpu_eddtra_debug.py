import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step
from parity import copy_state, STATE_FIELDS, INT_FIELDS
from oracle.coracle import COracle
from blom_amd.gpu import BlomGpu
case = make_case("chan_s", nslp0=2e-4)
nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm)
masks = dict(ip=ip, iu=iu, iv=iv, iq=iq)
orc = COracle(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
hostinit.init_state(orc, case)
gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
for nm, v in case.params.items():
    if not nm.endswith("0"):
        gpu.set(nm, v)
ns = dyncore_step(orc, 0, case.params["baclin"])
six = hostinit.step_indices(ns, case.kdm)
# advance orc to just before eddtra of step 2
for st in ("init_fluxes", "tmsmt1", "halo_cmnfld2", "halo_difest"):
    orc.set("nstep", 2); orc.stage(st, *six)
copy_state(orc, gpu, fields=STATE_FIELDS + INT_FIELDS + ["nslpx", "nslpy", "scp2", "scuy", "scvx", "scu2", "scv2", "scuxi", "scvyi"])
gpu.set("delt1", 2 * case.params["baclin"])
gpu.stage("eddtra", *six); orc.stage("eddtra", *six)
a, b = orc.get("umfltd"), gpu.get("umfltd")
w = np.argwhere(a != b)
print("six", six, "n diff", len(w), w[:6])
k, j, i = w[0]
print("orc col", a[:, j, i]); print("gpu col", b[:, j, i])
for nm in ("dpu", "dp", "p", "pbu", "kfpla", "difint", "nslpx", "temp", "saln", "scp2", "scu2", "scuy"):
    x, y = orc.get(nm), gpu.get(nm)
    print(nm, "equal@col", np.array_equal(x[:, j, i - 1:i + 1], y[:, j, i - 1:i + 1], equal_nan=True), x[:, j, i - 1].tolist()[:14], x[:, j, i].tolist()[:14])
