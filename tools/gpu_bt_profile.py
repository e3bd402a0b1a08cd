import sys, ctypes as C
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from bench import build_case
from blom_amd.gpu import BlomGpu
from blom_amd import hostinit
case, nreg, masks = build_case("channel")
gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
hostinit.init_state(gpu, case)
ns = gpu.step(0, 3)
nb = C.c_int(0)
gpu.lib.blomgpu_dbg_bt_profile(gpu.ctx, None, 0, C.byref(nb))
ns = gpu.step(ns, 1); gpu.sync()
buf = (C.c_longlong * (16 * nb.value))()
rc = gpu.lib.blomgpu_dbg_bt_profile(gpu.ctx, buf, nb.value, C.byref(nb))
a = np.array(buf[:]).reshape(nb.value, 16)
print("rc", rc, "blocks", nb.value)
t0 = a[:, 0].min()
rel = (a[:, :10] - t0) / 100.0      # wall_clock64 is 100 MHz -> microseconds
print("start spread us: min %.2f max %.2f" % (rel[:, 0].min(), rel[:, 0].max()))
names = ["start", "loads issued", "sync1(load done)", "cont1", "odd done", "cont2", "even done", "end"]
for k in range(8):
    print("%-18s mean %.2f  min %.2f  max %.2f  | dt from prev mean %.2f" % (names[k], rel[:, k].mean(), rel[:, k].min(), rel[:, k].max(), (rel[:, k] - rel[:, k - 1]).mean() if k else 0))
