"""Times the batched hor3map kernels on a full channel slab (106 080 columns x 53 layers) with
device-resident inputs: per-kernel HIP-event time, algorithmic bytes and the HBM roofline fraction.
Also times the reference's mod_hor3map on the host (when oracle/_ref/hor3map travelled) on a sample."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import h3m_cases as hc                    # noqa: E402
from blom_amd import hor3map as h3        # noqa: E402

ncol, n = int(os.environ.get("H3M_NCOL", "106080")), 53
reps = int(os.environ.get("H3M_REPS", "20"))
cfgs = {"ppm_tracer": (hc.PPM, 6, 4, hc.NON_OSCILLATORY_POSDEF, True, False),
        "ppm_density": (hc.PPM, 6, 4, hc.MONOTONIC, False, False),
        "pqm_tracer": (hc.PQM, 6, 4, hc.NON_OSCILLATORY_POSDEF, True, False),
        "plm": (hc.PLM, 0, 0, hc.MONOTONIC, True, False)}
kind = os.environ.get("H3M_KIND", "slab")
x, u, xd, ug = hc.make_slab(11, ncol, n, n, n + 1) if kind == "slab" else hc.make_columns(11, ncol, n, n, n + 1, kind)
if os.environ.get("H3M_UNIFORM"):
    x = np.ascontiguousarray(np.tile(np.arange(n + 1.0) * 9806.0, (ncol, 1))); xd = x.copy()
dev = torch.device("cuda:0")
tx, tu, txd, tug = (torch.from_numpy(a).to(dev) for a in (x, u, xd, ug))
out = {}
for name, cfg in cfgs.items():
    g = h3.ReconGrid(ncol, n, cfg[0], cfg[1], cfg[2])
    g.set_io(device_pointers=True, check_errors=False)
    s = h3.ReconSrc(g, cfg[3], cfg[4], cfg[5])
    r = h3.Remap(g, n)
    np_ = h3.P_ORD[cfg[0]] + 1
    tpc = torch.empty((ncol, n, np_), dtype=torch.float64, device=dev)
    tud = torch.empty((ncol, n), dtype=torch.float64, device=dev)
    txg = torch.empty((ncol, n + 1), dtype=torch.float64, device=dev)
    calls = {
        "prepare_reconstruction": lambda: g.prepare_reconstruction(tx.data_ptr()),
        "reconstruct": lambda: s.reconstruct(tu.data_ptr()),
        "extract_polycoeff": lambda: s.extract_polycoeff(out=tpc.data_ptr()),
        "regrid": lambda: s.regrid(tug.data_ptr(), -1e33, h3.REGRID_METHOD_2, out=txg.data_ptr(), n_grd=n + 1),
        "prepare_remapping": lambda: r.prepare_remapping(txd.data_ptr()),
        "remap": lambda: r.remap(s, out=tud.data_ptr()),
    }
    F = ncol * n * 8.0
    # algorithmic bytes: caller arrays in + out of each entry (state kept by the library not counted)
    alg = {"prepare_reconstruction": F, "reconstruct": F, "extract_polycoeff": np_ * F, "regrid": 2 * F,
           "prepare_remapping": F, "remap": F}
    res = {}
    for cn, f in calls.items():
        f()
        g.sync()
        ms, wall = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            f()
            g.sync()
            wall.append((time.perf_counter() - t0) * 1e3)
            ms.append(g.last_kernel_ms())
        k = float(np.median(ms))
        res[cn] = dict(kernel_ms=round(k, 4), call_ms=round(float(np.median(wall)), 4),
                       alg_GBps=round(alg[cn] / k / 1e6, 1), frac_hbm=round(alg[cn] / k / 1e6 / 8000.0, 4))
    g.free()
    out[name] = res
    print(name, json.dumps(res), flush=True)

if hc.have_ref():
    m = 4000
    for name, cfg in cfgs.items():
        t0 = time.perf_counter()
        hc.run_ref(*cfg, x[:m].copy(), u[:m].copy(), xd[:m].copy(), ug[:m].copy(), hc.METHOD_2)
        dt = time.perf_counter() - t0
        print(f"reference (1 core) {name}: {dt / m * ncol * 1e3:.1f} ms per slab for the 6-call sequence "
              f"(sample {m} columns)", flush=True)

# several fields per launch (the tracer loop): kernel time vs number of fields
cfg = cfgs["ppm_tracer"]
g = h3.ReconGrid(ncol, n, cfg[0], cfg[1], cfg[2])
g.set_io(device_pointers=True, check_errors=False)
g.prepare_reconstruction(tx.data_ptr())
r = h3.Remap(g, n)
r.prepare_remapping(txd.data_ptr())
srcs = [h3.ReconSrc(g, cfg[3], cfg[4], cfg[5]) for _ in range(8)]
ins = [tu.clone() for _ in range(8)]
outs = [torch.empty((ncol, n), dtype=torch.float64, device=dev) for _ in range(8)]
for nf in (1, 2, 4, 8):
    t_rec, t_rem = [], []
    for _ in range(reps):
        h3.reconstruct_many(g, srcs[:nf], [t.data_ptr() for t in ins[:nf]])
        g.sync()
        t_rec.append(g.last_kernel_ms())
        h3.remap_many(srcs[:nf], r, [t.data_ptr() for t in outs[:nf]])
        g.sync()
        t_rem.append(g.last_kernel_ms())
    a, b = float(np.median(t_rec)), float(np.median(t_rem))
    F = ncol * n * 8.0
    print(f"many nf={nf}: reconstruct {a:.3f} ms ({a / nf:.3f} per field, {nf * F / a / 1e6:.0f} GB/s alg), "
          f"remap {b:.3f} ms ({b / nf:.3f} per field, {nf * F / b / 1e6:.0f} GB/s alg)", flush=True)
g.free()
