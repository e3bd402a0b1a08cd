"""Where do two kernel variants differ on a case?  usage: gpu_variant_diff.py cfg nsteps opt=a opt=b"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.gpu import BlomGpu
cfg, nsteps = sys.argv[1], int(sys.argv[2])
opts = [dict((o.split("=")[0], int(o.split("=")[1])) for o in a.split(",")) for a in sys.argv[3:5]]
case = make_case(cfg, nslp0=2e-4)
nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm)
res = []
for o in opts + opts[-1:]:
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    hostinit.init_state(gpu, case)
    for k, v in o.items():
        gpu.set(k, v)
    gpu.step(0, nsteps)
    res.append({nm: gpu.get(nm) for nm in ("u", "v", "absvor", "dpvor", "utotn", "vtotn", "dp")})
    gpu.close()
for label, (a, b) in (("A vs B", (res[0], res[1])), ("B vs B again", (res[1], res[2]))):
    for nm in a:
        ne = ~((a[nm] == b[nm]) | (np.isnan(a[nm]) & np.isnan(b[nm])))
        if ne.any():
            w = np.argwhere(ne)
            print(label, nm, len(w), "first", w[:5].tolist(), "k range", w[:, 0].min(), w[:, 0].max(), "j range", w[:, 1].min() - 3, w[:, 1].max() - 3,
                  "i range", w[:, 2].min() - 3, w[:, 2].max() - 3)
        else:
            print(label, nm, "identical")
