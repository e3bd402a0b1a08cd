"""channel-size run: drift of the global integrals over N steps (to set the tolerances of the property test)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.gpu import BlomGpu
cfg = sys.argv[1] if len(sys.argv) > 1 else "channel"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
case = make_case(cfg, nslp0=2e-4)
nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm, arctic=case.nreg == 2)
gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
hostinit.init_state(gpu, case)
kk, J, I = case.kdm, slice(4, 4 + case.jdm), slice(4, 4 + case.idm)
area = gpu.get("scp2")[0][J, I] * ip[J, I]
def integrals(n):
    nn = (n - 1) * kk
    dp = gpu.get("dp")[nn:nn + kk, J, I]
    out = {"mass_pb": float((gpu.get("pb")[n - 1][J, I] * area).sum()), "mass_dp": float((dp.sum(0) * area).sum())}
    for nm in ("temp", "saln"):
        out[nm] = float(((gpu.get(nm)[nn:nn + kk, J, I] * dp).sum(0) * area).sum())
    t = gpu.get("temp")[nn:nn + kk, J, I]
    out["tmin"], out["tmax"] = float(t[:, ip[J, I] > 0].min()), float(t[:, ip[J, I] > 0].max())
    return out
ns = gpu.step(0, 1)
ref = None
for it in range(nsteps):
    ns = gpu.step(ns, 1)
    n = (ns + 1) % 2 + 1 if False else ((ns - 1 + 1) % 2 + 1)
    # level n of the step just taken: step index ns-1 -> n = (ns-1+1) % 2 + 1
    d = integrals(n)
    if ref is None:
        ref = d
    if it in (0, 1, 2, 5, 10, nsteps - 1):
        print(it, {k: (v if k.startswith("t") and k != "temp" else (v - ref[k]) / abs(ref[k])) for k, v in d.items()})
gpu.close()
