"""What the bench workload does physically over a long run (GPU box): config 2's step on the channel 208x512x53, ntr = 3, the channel
experiment's own forcing, diffusivities estimated every step (hostinit.DIFEST_NORESM) -- 1200 baroclinic steps = 12.5 model days.
Every 200 steps: range of T, the mixed layer depth (the two bulk layers, (p(3) - p(1)) / onem) min / mean / max, the columns whose
mixed layer iteration ended at its limit per step (mxlayr's maxitr, which the reference prints), max |u|, and the drift of total
mass, heat and salt relative to the start.  usage: python3 tools/longrun_full_physics.py [--frozen-diffusivities] > profiles/r05_longrun.txt"""
import sys
import numpy as np
sys.path.insert(0, '.')
import bench
from blom_amd.gpu import BlomGpu
from blom_amd import hostinit

frozen = "--frozen-diffusivities" in sys.argv
case, nreg, masks = bench.build_case("channel", "remap", "default")
gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
hostinit.init_state(gpu, case)
gpu.set("live_slopes", 1)
hostinit.init_forcing(gpu, case)
gpu.set("full_physics", 1)
if not frozen:
    hostinit.init_difest(gpu, case, device=True)
    for d_ in hostinit.DIFEST_NORESM:
        for nm, v in d_.items():
            gpu.set(nm, v)
    gpu.set("difest_live", 1)
wet = masks["ip"][4:-4, 4:-4] > 0
scp2 = gpu.get("scp2")[0][4:-4, 4:-4]
kk = case.kdm
print(f"# channel {case.idm}x{case.jdm}x{kk}, ntr = {case.ntr}, full physics, diffusivities {'frozen' if frozen else 'live (NorESM defaults)'}; baclin = {case.params['baclin']} s")
print("# step  Tmin  Tmax  mld_min  mld_mean  mld_max [m]  maxitr_entrain/step  maxitr_detrain/step  |u|max  d(mass)/mass  d(heat)/heat  d(salt)/salt  difdia_max  difint_mean  finite")
ns, base = 0, None
gpu.get_real("mxlayr_maxitr_entrain"); gpu.get_real("mxlayr_maxitr_detrain")
for blk in range(7):
    if blk:
        ns = gpu.step(ns, 200)
    nn = hostinit.step_indices(ns, kk)[3] if ns else hostinit.step_indices(0, kk)[3]
    t = gpu.get("temp")[nn:nn + kk][:, 4:-4, 4:-4]
    s = gpu.get("saln")[nn:nn + kk][:, 4:-4, 4:-4]
    dp = gpu.get("dp")[nn:nn + kk][:, 4:-4, 4:-4]
    u = gpu.get("u")[nn:nn + kk][:, 4:-4, 4:-4]
    w = np.broadcast_to(wet[None], t.shape)
    mld = (dp[0] + dp[1])[wet] / 9806.0
    mass = float(np.sum((dp * scp2[None])[w])); heat = float(np.sum((t * dp * scp2[None])[w])); salt = float(np.sum((s * dp * scp2[None])[w]))
    if base is None:
        base = (mass, heat, salt)
    ne = gpu.get_real("mxlayr_maxitr_entrain") / max(1, 200 if blk else 1)
    nd = gpu.get_real("mxlayr_maxitr_detrain") / max(1, 200 if blk else 1)
    dd = gpu.get("difdia")[:, 4:-4, 4:-4]; di = gpu.get("difint")[:, 4:-4, 4:-4]
    w2 = np.broadcast_to(wet[None], dd.shape)
    uu = u[np.isfinite(u) & (np.abs(u) < 1e10)]
    print(ns, f"{t[w].min():.4f} {t[w].max():.4f}  {mld.min():.3f} {mld.mean():.3f} {mld.max():.3f}  {ne:.2f} {nd:.2f}  {np.abs(uu).max():.4f}  "
          f"{(mass - base[0]) / base[0]:.3e} {(heat - base[1]) / base[1]:.3e} {(salt - base[2]) / base[2]:.3e}  {dd[w2].max():.4e} {di[w2].mean():.2f}  "
          f"{bool(np.isfinite(t[w]).all() and np.isfinite(dp[w]).all())}", flush=True)
gpu.close()
