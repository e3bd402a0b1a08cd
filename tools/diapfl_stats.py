"""Trip-count statistics of diapfl's data-dependent loops on a case (C oracle, CPU)."""
import sys, ctypes as C
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from bench import build_case
from oracle.coracle import COracle
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step
cfg = sys.argv[1]; nsteps = int(sys.argv[2])
case, nreg, masks = build_case(cfg)
be = COracle(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
hostinit.init_state(be, case)
lib = be.lib
ns = 0
for s in range(nsteps):
    for nm in ("orc_diapfl_stat_lim", "orc_diapfl_stat_niter"):
        C.memset(C.addressof((C.c_long * 101).in_dll(lib, nm)), 0, 101 * 8)
    C.c_long.in_dll(lib, "orc_diapfl_stat_span").value = 0
    ns = dyncore_step(be, ns, case.params["baclin"])
    lim = np.array((C.c_long * 101).in_dll(lib, "orc_diapfl_stat_lim")[:])
    nit = np.array((C.c_long * 101).in_dll(lib, "orc_diapfl_stat_niter")[:])
    span = C.c_long.in_dll(lib, "orc_diapfl_stat_span").value
    print("step", s, "cols", nit.sum(), "mean span", span / max(1, nit.sum()),
          "lim hist", {i: int(v) for i, v in enumerate(lim) if v}, "niter hist", {i: int(v) for i, v in enumerate(nit) if v}, flush=True)
