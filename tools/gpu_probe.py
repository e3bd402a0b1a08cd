"""Dev helper (GPU box): per-stage parity over many steps + where a free run first diverges."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step, DYNCORE_STAGES
from blom_amd.gpu import BlomGpu
from oracle.refblom import get_ref_backend
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, INT_FIELDS
cfg = sys.argv[1]; nsteps = int(sys.argv[2])
SCRATCH = {"uflux", "vflux", "uflux2", "vflux2", "uflux3", "vflux3", "utotm", "vtotm", "util1", "util2"}
fields = [f for f in STATE_FIELDS + INT_FIELDS if f not in SCRATCH]
case = make_case(cfg)
ref = get_ref_backend(cfg, case.depth)
hostinit.init_state(ref, case)
gpu = BlomGpu(case.idm, case.jdm, case.kdm, ref.ntr, ref.nreg, ref.masks)
for nm, v in case.params.items():
    if not nm.endswith("0"): gpu.set(nm, v)
# free run, stage by stage on both, no re-sync: first stage where they differ
copy_state(ref, gpu); gpu.set("delt1", case.params["baclin"])
ns = 0
found = False
for it in range(nsteps):
    six = hostinit.step_indices(ns, case.kdm)
    ref.set("nstep", ns + 1); gpu.set("nstep", ns + 1)
    for st in DYNCORE_STAGES:
        ref.stage(st, *six); gpu.stage(st, *six)
        bad = diff_report(ref, gpu, fields=fields)
        if bad and not found:
            print(f"first divergence: step {ns+1} stage {st}\n" + fmt_report(bad[:10])); found = True
            break
    if found: break
    ref.set("delt1", 2 * case.params["baclin"]); gpu.set("delt1", 2 * case.params["baclin"])
    ns += 1
if not found: print(f"{cfg}: {nsteps} steps stage-by-stage free run bit-identical")
