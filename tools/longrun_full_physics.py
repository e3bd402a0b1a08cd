import sys, numpy as np
sys.path.insert(0, '.')
import bench
from blom_amd.gpu import BlomGpu
from blom_amd import hostinit
case, nreg, masks = bench.build_case("channel", "remap", "default")
gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
hostinit.init_state(gpu, case)
gpu.set("live_slopes", 1)
hostinit.init_forcing(gpu, case)
gpu.set("full_physics", 1)
wet = masks["ip"][4:-4, 4:-4] > 0
ns = 0
for blk in range(6):
    ns = gpu.step(ns, 200)
    six = hostinit.step_indices(ns, case.kdm); nn = six[3]; kk = case.kdm
    t = gpu.get("temp")[nn:nn+kk][:, 4:-4, 4:-4]; dp = gpu.get("dp")[nn:nn+kk][:, 4:-4, 4:-4]; u = gpu.get("u")[nn:nn+kk][:, 4:-4, 4:-4]
    w = np.broadcast_to(wet[None], t.shape)
    print(ns, "T", float(t[w].min()), float(t[w].max()), "dp1 mean [m]", float(dp[0][wet].mean()/9806), "|u|max", float(np.abs(u[np.isfinite(u)&(np.abs(u)<1e10)]).max()), "finite", bool(np.isfinite(t[w]).all() and np.isfinite(dp[w]).all()), flush=True)
gpu.close()
