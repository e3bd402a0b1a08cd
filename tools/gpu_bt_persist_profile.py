"""Where an iteration of the persistent barotropic kernel spends its time (library built with -DBT_PROFILE)."""
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
nbx, nby = (case.idm + 25) // 26, (case.jdm + 15) // 16      # the persistent form: 26 x 16 tiles
nw = nbx * nby * 16 * 8
gpu.lib.blomgpu_dbg_bt_prof.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert gpu.lib.blomgpu_dbg_bt_prof(gpu.ctx, None, nw) == 0
ns = gpu.step(ns, 1); gpu.sync()
buf = (C.c_longlong * nw)()
assert gpu.lib.blomgpu_dbg_bt_prof(gpu.ctx, buf, nw) == 0
a = np.array(buf[:], dtype=np.int64).reshape(nbx * nby, 16, 8)       # the LAST phase's launch overwrote the earlier ones
names = ["top", "counts seen", "rim re-read", "sweeps done", "stores drained", "count published"]
it = slice(1, 11)
t = a[:, it, :6].astype(float) / 100.0                                 # us
ok = (a[:, it, 5] > 0).all(axis=1)
t = t[ok]
print("tiles", t.shape[0], "iterations per tile", t.shape[1])
for k in range(1, 6):
    d = t[:, :, k] - t[:, :, k - 1]
    print("%-16s -> %-16s mean %.2f us  p10 %.2f  p90 %.2f" % (names[k - 1], names[k], d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
t8 = a[:, it, :8].astype(float)[ok] / 100.0
print("rim re-read -> odd substep done %.2f us; -> even continuity done %.2f us; -> sweeps done %.2f us" % (
    (t8[:, :, 6] - t8[:, :, 2]).mean(), (t8[:, :, 7] - t8[:, :, 6]).mean(), (t8[:, :, 3] - t8[:, :, 7]).mean()))
d = t[:, 1:, 0] - t[:, :-1, 5]
print("%-16s -> %-16s mean %.2f us" % ("count published", "next top", d.mean()))
per = t[:, 1:, 0] - t[:, :-1, 0]
print("iteration: mean %.2f us" % per.mean())
