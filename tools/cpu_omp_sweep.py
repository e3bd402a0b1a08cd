import os, sys, time, threading
os.environ.setdefault("OMP_NUM_THREADS", sys.argv[1] if len(sys.argv) > 1 else "8")
os.environ["OMP_STACKSIZE"] = "1G"
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from bench import build_case
from blom_amd import hostinit
from blom_amd.stepper import dyncore_step
from oracle.refblom import get_ref_backend
case, nreg, masks = build_case("channel")
def run():
    be = get_ref_backend("channel_omp", case.depth)
    hostinit.init_state(be, case)
    ns = dyncore_step(be, 0, case.params["baclin"])
    t0 = time.time(); n = 0
    while n < 3:
        ns = dyncore_step(be, ns, case.params["baclin"]); n += 1
    print("threads", os.environ["OMP_NUM_THREADS"], "ms/step", (time.time() - t0) / n * 1e3, flush=True)
    import numpy as np
    print("finite", np.isfinite(be.get("u")[:, 4:-4, 4:-4]).all())
threading.stack_size(2 << 30)
t = threading.Thread(target=run); t.start(); t.join()
