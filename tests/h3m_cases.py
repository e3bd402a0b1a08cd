"""Column generators and the two CPU-side runners used by the hor3map tests:
   run_ref       -- the reference's own mod_hor3map (oracle/_ref/hor3map/libhor3mapref.so)
   run_hostcheck -- the device column routines compiled for the host (tests/hostcheck)
Both take/return the caller-layout arrays a(level, column) as numpy arrays of shape (ncol, nlev)."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "hor3map", "libhor3mapref.so")
HOST_LIB = os.path.join(HERE, "hostcheck", "libh3m_hostcheck.so")

PCM, PLM, PPM, PQM = 100, 101, 102, 103
NO_LIMITING, MONOTONIC, NON_OSCILLATORY, NON_OSCILLATORY_POSDEF = 200, 201, 203, 204
METHOD_1, METHOD_2 = 301, 302
P_ORD = {PCM: 0, PLM: 1, PPM: 2, PQM: 4}
MISSING = -1.0e33


def have_ref():
    return os.path.exists(REF_LIB)


def have_hostcheck():
    return os.path.exists(HOST_LIB)


def make_slab(seed, ncol, n_src, n_dst, n_grd):
    """Columns as a model slab holds them (bench input): neighbouring columns resemble each other -- the
    water depth varies smoothly along the slab, layers below the bottom are massless (one contiguous run at
    the end of the column), interior thicknesses are a smooth profile with a few percent of noise, the
    destination grid is the source grid relaxed towards fixed target depths.  (make_columns below draws
    every layer of every column independently, empty and thin layers anywhere: the worst case for
    wavefront divergence, which is what the parity tests want.)"""
    rng = np.random.default_rng(seed)
    s = np.arange(ncol) / 512.0
    depth = 2500.0 + 1500.0 * np.sin(0.7 * s) * np.cos(0.23 * s) + 300.0 * np.sin(5.1 * s)        # m
    depth = np.maximum(depth, 150.0) * 9806.0
    k = np.arange(n_src)
    prof = 10.0 * 1.09 ** k * 9806.0                                                                  # stretched
    h = prof[None, :] * (1.0 + 0.03 * rng.standard_normal((ncol, n_src)))
    top = np.concatenate([np.zeros((ncol, 1)), np.cumsum(h, 1)], 1)
    x_src = np.minimum(top, depth[:, None])                       # layers below the bottom become massless
    x_src[:, -1] = depth
    zc = 0.5 * (x_src[:, 1:] + x_src[:, :-1]) / depth[:, None]
    u = 24.0 + 6.0 * zc ** 0.7 + 0.02 * np.sin(9.0 * zc + s[:, None]) + 0.002 * rng.standard_normal((ncol, n_src))
    kd = np.arange(n_dst)
    tgt = np.concatenate([[0.0], np.cumsum(12.0 * 1.085 ** kd)]) * 9806.0
    x_dst = np.minimum(0.7 * tgt[None, :] + 0.3 * np.interp(np.linspace(0, n_src, n_dst + 1), np.arange(n_src + 1), top[0])[None, :],
                       depth[:, None])
    x_dst[:, 0] = 0.0
    x_dst[:, -1] = depth
    x_dst = np.maximum.accumulate(x_dst, axis=1)
    lo, hi = u.min(1, keepdims=True), u.max(1, keepdims=True)
    u_grd = lo + (hi - lo) * (np.arange(n_grd)[None, :] + 0.5) / n_grd
    return (np.ascontiguousarray(x_src), np.ascontiguousarray(u), np.ascontiguousarray(x_dst),
            np.ascontiguousarray(u_grd))


def make_columns(seed, ncol, n_src, n_dst, n_grd, kind="ocean", decreasing=False):
    """Seeded synthetic columns in the shape BLOM feeds hor3map (mod_ale_regrid_remap.F90:224-247):
    interface pressures with empty and very thin layers, a stratified field with noise, a
    destination grid over the same range, target interface values for regridding."""
    rng = np.random.default_rng(seed)
    h = rng.uniform(0.2, 3.0, (ncol, n_src)) * 9806.0
    style = rng.integers(0, 6, (ncol, n_src))
    h[style == 0] = 0.0                                   # massless layers
    h[style == 1] *= 10.0 ** rng.uniform(-9, -3, (ncol, n_src))[style == 1]   # thin layers (merge paths)
    if kind == "few":                                     # almost everything empty: method fall-back
        keep = rng.integers(0, n_src, (ncol, 3))
        m = np.zeros((ncol, n_src), bool)
        for c in range(ncol):
            m[c, keep[c, : rng.integers(1, 4)]] = True
        h[~m] = 0.0
    h[:, 0] = np.where(h.sum(1) == 0.0, 9806.0, h[:, 0])
    x_src = np.concatenate([np.zeros((ncol, 1)), np.cumsum(h, 1)], 1)
    zc = 0.5 * (x_src[:, 1:] + x_src[:, :-1]) / np.maximum(x_src[:, -1:], 1.0)
    if kind == "tracer":                                  # non-monotonic, partly negative (posdef paths)
        u = np.sin(7.0 * zc + rng.uniform(0, 6, (ncol, 1))) + 0.3 * rng.standard_normal((ncol, n_src)) + 0.4
    else:                                                 # density-like, mostly increasing downward
        u = 24.0 + 6.0 * zc ** 0.7 + 0.05 * rng.standard_normal((ncol, n_src))
        flat = rng.integers(0, 8, (ncol, n_src)) == 0
        u[:, 1:][flat[:, 1:]] = u[:, :-1][flat[:, 1:]]
    w = rng.uniform(0.0, 1.0, (ncol, n_dst))
    w[rng.integers(0, 5, (ncol, n_dst)) == 0] = 0.0
    w[:, 0] = np.where(w.sum(1) == 0.0, 1.0, w[:, 0])
    cw = np.cumsum(w, 1) / w.sum(1, keepdims=True)
    x_dst = np.concatenate([np.zeros((ncol, 1)), cw], 1) * x_src[:, -1:]
    x_dst = np.minimum(x_dst, x_src[:, -1:])
    x_dst[:, -1] = x_src[:, -1]
    if kind == "bad":                                     # error paths: errstat 3, 8 and 9
        # (not in column 0: there the reference's structures would still be un-initialised, a state
        # the batched library, which allocates at creation, does not have)
        x_src[3::7, 3] = x_src[3::7, 2] - 1.0
        x_dst[4::7, -1] += 5.0
        x_dst[5::7, 2] = x_dst[5::7, 1] - 1.0e-3
    lo, hi = u.min(1, keepdims=True), u.max(1, keepdims=True)
    u_grd = np.sort(lo - 0.05 + (hi - lo + 0.1) * rng.uniform(0, 1, (ncol, n_grd)), 1)
    if decreasing:
        x_src, x_dst = -x_src, -x_dst
    return (np.ascontiguousarray(x_src), np.ascontiguousarray(u), np.ascontiguousarray(x_dst),
            np.ascontiguousarray(u_grd))


def _run(lib, fname, method, lb, rb, limiting, pc_l, pc_r, x_src, u_src, x_dst, u_grd, regrid_method):
    ncol, n_src = u_src.shape
    n_dst, n_grd = x_dst.shape[1] - 1, u_grd.shape[1]
    np_ = P_ORD[method] + 1
    out = dict(polycoeff=np.zeros((ncol, n_src, np_)), u_dst=np.zeros((ncol, n_dst)),
               x_grd=np.zeros((ncol, n_grd)), errs=np.zeros((ncol, 6), np.int32),
               n_act=np.zeros(ncol, np.int32), m_act=np.zeros(ncol, np.int32))
    f = getattr(lib, fname)
    f.restype = None
    dp = ctypes.POINTER(ctypes.c_double)
    ip = ctypes.POINTER(ctypes.c_int)
    ci = ctypes.c_int
    f.argtypes = [ci] * 11 + [dp] * 4 + [ctypes.c_double] + [dp] * 3 + [ip] * 3
    P = lambda a: a.ctypes.data_as(dp)
    I = lambda a: a.ctypes.data_as(ip)
    f(method, lb, rb, limiting, int(pc_l), int(pc_r), ncol, n_src, n_dst, n_grd, regrid_method,
      P(x_src), P(u_src), P(x_dst), P(u_grd), MISSING, P(out["polycoeff"]), P(out["u_dst"]), P(out["x_grd"]),
      I(out["errs"]), I(out["n_act"]), I(out["m_act"]))
    return out


_libs = {}


def _lib(path):
    if path not in _libs:
        _libs[path] = ctypes.CDLL(path)
    return _libs[path]


def run_ref(*a, **k):
    return _run(_lib(REF_LIB), "ref_h3m_run", *a, **k)


def run_hostcheck(*a, **k):
    return _run(_lib(HOST_LIB), "h3m_hostcheck_run", *a, **k)


def compare(a, b, what=""):
    """bitwise comparison of two result dicts; returns a list of mismatch descriptions"""
    bad = []
    if not np.array_equal(a["errs"], b["errs"]):
        c = np.argwhere(a["errs"] != b["errs"])[0]
        bad.append(f"{what} errstat differs at column {c[0]} call {c[1]}: {a['errs'][c[0]]} vs {b['errs'][c[0]]}")
        return bad
    ok0 = a["errs"][:, 0] == 0
    for nm in ("n_act", "m_act"):
        if not np.array_equal(a[nm][ok0], b[nm][ok0]):
            c = np.argwhere((a[nm] != b[nm]) & ok0)[0][0]
            bad.append(f"{what} {nm} differs at column {c}: {a[nm][c]} vs {b[nm][c]}")
    for nm, call in (("polycoeff", 2), ("x_grd", 3), ("u_dst", 5)):
        ok = a["errs"][:, call] == 0
        x, y = a[nm][ok], b[nm][ok]
        if not np.array_equal(x.view(np.int64), y.view(np.int64)):
            d = np.argwhere(x.view(np.int64) != y.view(np.int64))
            c = d[0]
            col = np.flatnonzero(ok)[c[0]]
            bad.append(f"{what} {nm}: {len(d)} values differ; first at column {col} index {tuple(c[1:])}: "
                       f"{x[tuple(c)]!r} vs {y[tuple(c)]!r} (n_act {a['n_act'][col]}, m_act {a['m_act'][col]})")
    return bad


# the configurations every runner is put through: (method, left/right_bndr_ord, limiting, pc_left, pc_right)
CONFIGS = [
    (PCM, 0, 0, MONOTONIC, True, True),
    (PLM, 0, 0, NO_LIMITING, False, False),
    (PLM, 0, 0, MONOTONIC, True, False),
    (PPM, 6, 4, NO_LIMITING, False, False),
    (PPM, 6, 4, MONOTONIC, False, False),            # BLOM density (tests/fuk95/limits:229-236)
    (PPM, 6, 4, NON_OSCILLATORY, True, False),        # BLOM velocity
    (PPM, 6, 4, NON_OSCILLATORY_POSDEF, True, False), # BLOM tracers (mod_ale_regrid_remap.F90:1414-1416)
    (PPM, 2, 3, MONOTONIC, True, True),
    (PQM, 6, 4, NO_LIMITING, False, False),
    (PQM, 6, 4, MONOTONIC, False, False),
    (PQM, 6, 4, NON_OSCILLATORY, True, False),
    (PQM, 0, 0, NON_OSCILLATORY_POSDEF, True, False),
    (PQM, 3, 5, MONOTONIC, False, True),
]
KINDS = ["ocean", "tracer", "few", "bad"]


def run_gpu(method, lb, rb, limiting, pc_l, pc_r, x_src, u_src, x_dst, u_grd, regrid_method):
    """the same sequence through the C ABI on the GPU (blom_amd.hor3map)"""
    from blom_amd import hor3map as h3
    ncol, n_src = u_src.shape
    n_dst = x_dst.shape[1] - 1
    g = h3.ReconGrid(ncol, n_src, method, lb, rb)
    g.raise_on_error = False
    s = h3.ReconSrc(g, limiting, pc_l, pc_r)
    r = h3.Remap(g, n_dst)
    errs = np.zeros((ncol, 6), np.int32)
    rcs = []
    rcs.append(g.prepare_reconstruction(x_src)); errs[:, 0] = g.errstat()
    n_act, m_act = g.info()
    rcs.append(s.reconstruct(u_src)); errs[:, 1] = g.errstat()
    pc, rc = s.extract_polycoeff(); rcs.append(rc); errs[:, 2] = g.errstat()
    xg, rc = s.regrid(u_grd, MISSING, regrid_method); rcs.append(rc); errs[:, 3] = g.errstat()
    rcs.append(r.prepare_remapping(x_dst)); errs[:, 4] = g.errstat()
    ud, rc = r.remap(s); rcs.append(rc); errs[:, 5] = g.errstat()
    g.free()
    # the call's return value is the errstat of the first failing column
    for k in range(6):
        nz = np.flatnonzero(errs[:, k])
        assert rcs[k] == (errs[nz[0], k] if len(nz) else 0), (k, rcs[k], errs[nz[:3], k])
    return dict(polycoeff=pc, u_dst=ud, x_grd=xg, errs=errs, n_act=n_act, m_act=m_act)
