import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step
from oracle.refblom import get_ref_backend
from blom_amd.gpu import BlomGpu
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, INT_FIELDS
case = make_case("tri_s")
ref = get_ref_backend("tri_s", case.depth)
hostinit.init_state(ref, case)
for fused in (0, 1):
    hostinit.init_state(ref, case)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            gpu.set(nm, v)
    gpu.set("barotp_fused", fused)
    res = []
    def hook(st, six):
        if st != "barotp": return
        copy_state(ref, gpu)
        gpu.set("delt1", ref.ref.get_real("delt1"))
        gpu.stage(st, *six)
        res.append(six)
    ns = dyncore_step(ref, 0, case.params["baclin"], hook=hook)
    bad = diff_report(ref, gpu, fields=[f for f in STATE_FIELDS if f not in ("uflux","vflux","uflux2","vflux2","uflux3","vflux3","utotm","vtotm")], rtol=1e-12, atol=1e-9)
    for nm in ("ubflxs", "vbflxs"):
        a, b = ref.get(nm), gpu.get(nm)
        w = np.argwhere(~((np.abs(a - b) <= 1e-9 + 1e-12 * np.abs(a)) | (a == b)))
        if len(w): print(nm, "diff at levels", sorted(set(w[:,0].tolist())), "j", sorted(set((w[:,1]-3).tolist())), "i", sorted(set((w[:,2]-3).tolist())), "six", res[-1])
    print("fused", fused, "mismatch:" if bad else "OK", fmt_report(bad[:6]))
    if bad:
        a, b = ref.get("pb"), gpu.get("pb")
        w = np.argwhere(np.abs(a - b) > 1e-9 + 1e-12 * np.abs(a))
        print("pb rows (j) with diffs:", sorted(set((w[:, 1] - 3).tolist())))
    gpu.close()
