"""Shared helpers for the parity tests: field lists, state transfer between backends and
exact comparison with readable diagnostics."""
import numpy as np

# every array that is an input or output of some stage on the path
STATE_FIELDS = [
    "u", "v", "dp", "dpu", "dpv", "temp", "saln", "sigma", "uflx", "vflx", "utflx", "vtflx",
    "usflx", "vsflx", "p", "pu", "pv", "phi", "cau", "cav", "ubflxs", "vbflxs", "ub", "vb", "pb",
    "pbu", "pbv", "ubflxs_p", "vbflxs_p", "pb_p", "pbu_p", "pbv_p", "ubcors_p", "vbcors_p", "sealv",
    "pgfx", "pgfy", "pgfx_o", "pgfy_o", "pgfxm", "pgfym", "xixp", "xixm", "xiyp", "xiym",
    "pgfxm_o", "pgfym_o", "xixp_o", "xixm_o", "xiyp_o", "xiym_o",
    "ubflx", "vbflx", "pb_mn", "ubflx_mn", "vbflx_mn", "pvtrop",
    "dpold", "dpuold", "dpvold", "sigmar", "temmin", "difint", "difiso", "difdia", "difmxp", "difmxq", "difwgt",
    "umfltd", "vmfltd", "umflsm", "vmflsm", "utfltd", "vtfltd", "utflsm", "vtflsm", "utflld", "vtflld",
    "usfltd", "vsfltd", "usflsm", "vsflsm", "usflld", "vsflld",
    "utotm", "vtotm", "utotn", "vtotn", "uflux", "vflux", "uflux2", "vflux2", "uflux3", "vflux3",
    "umax", "vmax", "util1", "util2", "taux", "tauy", "ustarb", "trc",
]
GRID_FIELDS = ["scqx", "scqy", "scpx", "scpy", "scux", "scuy", "scvx", "scvy", "scq2", "scp2", "scu2",
               "scv2", "scq2i", "scp2i", "scuxi", "scuyi", "scvxi", "scvyi", "corioq", "coriop"]
INT_FIELDS = ["kfpla"]


def copy_state(src, dst, fields=None):
    """Copy every field both backends know from src to dst."""
    for nm in (fields or (STATE_FIELDS + GRID_FIELDS + INT_FIELDS)):
        try:
            a = src.get(nm)
        except KeyError:
            continue
        if hasattr(dst, "has_field") and not dst.has_field(nm):
            continue
        dst.put(nm, a)


def diff_report(ref, cand, fields=None, rtol=0.0, atol=0.0):
    """Returns a list of (field, nbad, maxabs, first_index) for fields that differ."""
    bad = []
    for nm in (fields or (STATE_FIELDS + INT_FIELDS)):
        try:
            a = np.asarray(ref.get(nm))
            b = np.asarray(cand.get(nm))
        except KeyError:
            continue
        n = min(a.shape[0], b.shape[0])
        a, b = a[:n], b[:n]
        if rtol == 0.0 and atol == 0.0:
            ne = ~((a == b) | (np.isnan(a) & np.isnan(b)))
        else:
            with np.errstate(invalid="ignore"):
                ne = ~((np.abs(a - b) <= atol + rtol * np.abs(a)) | (np.isnan(a) & np.isnan(b)) | (a == b))
        if ne.any():
            k, j, i = np.argwhere(ne)[0]
            with np.errstate(invalid="ignore", over="ignore"):
                d = np.abs(a.astype(np.float64) - b.astype(np.float64))
                mx = float(np.nanmax(np.where(ne, d, 0.0)))
            bad.append((nm, int(ne.sum()), mx, (int(k) + 1, int(j) - 3, int(i) - 3),
                        float(a[k, j, i]), float(b[k, j, i])))
    return bad


def fmt_report(bad):
    return "\n".join(f"  {nm}: {n} differ, max|d|={mx:.3e}, first at (k,j,i)={idx} ref={ra!r} got={rb!r}"
                     for nm, n, mx, idx, ra, rb in bad)


def load_golden_init(cfg):
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    z = np.load(os.path.join(here, "golden", f"{cfg}_init.npz"))
    masks = {m: z["mask_" + m] for m in ("ip", "iu", "iv", "iq")}
    fields = {k: z[k] for k in z.files if not k.startswith("mask_")}
    return masks, fields


def put_fields(be, fields):
    for nm, a in fields.items():
        if hasattr(be, "has_field") and not be.has_field(nm):
            continue
        be.put(nm, a)


def destabilise_for_convec(be, case, n, seed=0):
    """Edits backend `be` (level n = 1|2 of the two-time-level arrays) so that convec has real work:
    in about a third of the columns the mixed layer (layers 1, 2) is made colder and saltier than
    several interior layers, some interior layers are emptied, and kfpla is moved both below and
    beyond the first non-empty interior layer (the `kfpl < kfplo <= kk` and `kfplo > kk` branches,
    phy/mod_convec.F90:110-191).  Returns the number of edited columns."""
    kk = case.kdm
    nn = (n - 1) * kk
    rng = np.random.default_rng(seed)
    temp, saln, dp, kf = (np.array(be.get(nm)) for nm in ("temp", "saln", "dp", "kfpla"))
    nj, ni = temp.shape[1:]
    pick = rng.random((nj, ni)) < 0.35
    dT = rng.uniform(2.0, 14.0, (nj, ni))
    dS = rng.uniform(0.0, 1.5, (nj, ni))
    for k in (0, 1):
        temp[nn + k][pick] -= dT[pick]
        saln[nn + k][pick] += dS[pick]
    empty = (rng.random((kk, nj, ni)) < 0.15) & pick[None]
    empty[:2] = False
    dpn = dp[nn:nn + kk]
    dpn[empty] = 0.0
    kf[n - 1][pick] = rng.integers(3, kk + 2, (nj, ni))[pick]
    be.put("temp", temp)
    be.put("saln", saln)
    be.put("dp", dp)
    be.put("kfpla", kf)
    for nm in ("u", "v"):                      # a sheared flow for the velocity remap to redistribute
        a = np.array(be.get(nm))
        a[nn:nn + kk] = 0.1 * rng.standard_normal((kk, nj, ni))
        be.put(nm, a)
    return int(pick.sum())
