"""Host-side initialisation of the dynamical-core state (numpy).

This is the netCDF-free equivalent of what blom_init_phase1/2 and inicon do between
"grid + layer structure known" and "first blom_step" (phy/mod_blom_init.F90:203-444,
phy/mod_inicon.F90:932-1459), plus the integer mask construction of bigrid
(phy/mod_bigrid.F90:44-317) and the single-tile halo rules of xctilr
(phy/mod_xc.F90:4222-4428).  It drives any backend that offers

    get(name) -> ndarray(nlev, nj, ni)      put(name, ndarray)
    set(name, scalar)                       stage(name, m, n, mm, nn, k1m, k1n)
    masks: dict(ip, iu, iv, iq)             kdm, idm, jdm, ntr

so the very same input state can be handed to the device library and (in tests) to the
reference.  Fortran a(i,j,k) == arr[k-1, j+3, i+3].
"""
import numpy as np

NBDY = 4
EPSILP = 1.0e-12        # phy/mod_constants.F90:43
SPVAL = 1.0e33
GRAV = 9.806            # phy/mod_constants.F90:32


def sl(lo, hi):
    """numpy slice for the Fortran index range lo..hi (inclusive) of a halo-4 axis."""
    return slice(lo + NBDY - 1, hi + NBDY)


# ----------------------------------------------------------------------------------
# xctilr, single tile (phy/mod_xc.F90:4374-4419; arctic form not needed for nreg != 2)
# ----------------------------------------------------------------------------------
def xctilr_np(a, l1, ld, mh, nh, nreg, ii, jj, vland=0.0, itype=1):
    """In-place halo update of levels l1..ld (1-based) of a (nlev, nj, ni) array.
    nreg = 2 (arctic patch, single tile): phy/mod_xc.F90:4262-4372 -- closed in the south, the
    northern rows mirrored across the arctic seam according to the grid (itype % 10: 1 p, 2 q,
    3 u, 4 v) with a sign change for vector fields (itype > 10), periodic in i."""
    mhl = max(0, min(mh, NBDY))
    nhl = max(0, min(nh, NBDY))
    v = a[l1 - 1:ld]
    o = NBDY - 1            # index of Fortran 0
    if nreg == 2:
        sgn = -1.0 if itype > 10 else 1.0
        g = itype % 10
        i = np.arange(1, ii + 1)
        if nhl > 0:
            v[:, o + 1 - nhl:o + 1, sl(1, ii)] = vland
        if g in (1, 4):
            io = ii - (i - 1) % ii                      # p-, v-grid
        else:
            io = (ii - (i - 1)) % ii + 1                # q-, u-grid
        if g in (1, 3):                                 # p-, u-grid: rows jj..jj+nhl <- rows jj-1..jj-1-nhl
            for j in range(0, nhl + 1):
                v[:, o + jj + j, o + i] = sgn * v[:, o + jj - 1 - j, o + io]
        else:                                           # q-, v-grid: second half of row jj, rows jj+1..
            h = i > ii // 2
            v[:, o + jj, o + i[h]] = sgn * v[:, o + jj, o + io[h]]
            for j in range(1, nhl + 1):
                v[:, o + jj + j, o + i] = sgn * v[:, o + jj - j, o + io]
        if mhl > 0:
            # the reference's E/W loop runs over k = 1..ld (not l1..ld); same data movement for the
            # levels a caller passes, which is all this helper is given
            js = sl(1 - nhl, jj + nhl)
            for q in range(1, mhl + 1):
                v[:, js, o + 1 - q] = v[:, js, o + ii + 1 - q]
                v[:, js, o + ii + q] = v[:, js, o + q]
        return
    assert nreg in (0, 1, 3, 4)
    if nhl > 0:
        if nreg <= 2:       # closed in latitude
            v[:, o + 1 - nhl:o + 1, sl(1, ii)] = vland
            v[:, o + jj + 1:o + jj + 1 + nhl, sl(1, ii)] = vland
        else:
            for j in range(1, nhl + 1):
                v[:, o + 1 - j, sl(1, ii)] = v[:, o + jj + 1 - j, sl(1, ii)]
                v[:, o + jj + j, sl(1, ii)] = v[:, o + j, sl(1, ii)]
    if mhl > 0:
        js = sl(1 - nhl, jj + nhl)
        if nreg in (0, 4):  # closed in longitude
            v[:, js, o + 1 - mhl:o + 1] = vland
            v[:, js, o + ii + 1:o + ii + 1 + mhl] = vland
        else:
            for i in range(1, mhl + 1):
                v[:, js, o + 1 - i] = v[:, js, o + ii + 1 - i]
                v[:, js, o + ii + i] = v[:, js, o + i]


# ----------------------------------------------------------------------------------
# bigrid: integer masks (bit-exact integer work)
# ----------------------------------------------------------------------------------
def bigrid_np(depth_in, idm, jdm, arctic=False):
    """Returns (nreg, depth_with_halo, ip, iu, iv, iq) following phy/mod_bigrid.F90:59-302
    for a single tile (i0=j0=0, ii=idm, jj=jdm).  Water in the last row means periodicity in j, or
    -- with `arctic`, the reference's nreg = 2 from patch.input -- the arctic patch (:77-78)."""
    ii, jj = idm, jdm
    depth = depth_in.copy()
    o = NBDY - 1
    lperiodi = depth[sl(1, jj), o + ii].max() > 0.0
    top = depth[o + jj, sl(1, ii)].max() > 0.0
    larctic = top and arctic
    lperiodj = top and not arctic
    if larctic:
        assert lperiodi, "the arctic patch needs a domain that is periodic in i"
        nreg = 2
    elif not lperiodi and not lperiodj:
        nreg = 0
    elif lperiodi and not lperiodj:
        nreg = 1
    elif lperiodi and lperiodj:
        nreg = 3
    else:
        nreg = 4
    d3 = depth[None]
    xctilr_np(d3, 1, 1, NBDY, NBDY, nreg, ii, jj, itype=1)
    if not lperiodj:
        depth[:o + 1, :] = 0.0
        if not larctic:
            depth[o + jj + 1:, :] = 0.0
    if not lperiodi:
        depth[:, :o + 1] = 0.0
        depth[:, o + ii + 1:] = 0.0
    # single-width inlets and 1-point seas are refused, as the reference does (:164-195: "Must correct bathymetry
    # before running BLOM", xcstop)
    J, I = sl(1, jj), sl(1, ii)
    wet = depth[J, I] > 0.0
    nzero = ((depth[J, sl(0, ii - 1)] <= 0.0).astype(int) + (depth[J, sl(2, ii + 1)] <= 0.0)
             + (depth[sl(0, jj - 1), I] <= 0.0) + (depth[sl(2, jj + 1), I] <= 0.0))
    bad = np.argwhere(wet & (nzero >= 3))
    if len(bad):
        raise ValueError("bigrid: must correct bathymetry before running BLOM: " +
                         ", ".join(f"dh({i + 1},{j + 1}) has {nzero[j, i]} land neighbours" for j, i in bad[:8]))
    nj, ni = depth.shape
    ip = (depth > 0.0).astype(np.int32)
    iu = np.zeros_like(ip)
    iv = np.zeros_like(ip)
    iq = np.zeros_like(ip)
    J, I = sl(1, jj), sl(1, ii)
    Jm, Im = sl(0, jj - 1), sl(0, ii - 1)
    iu[J, I] = ((ip[J, Im] > 0) & (ip[J, I] > 0)).astype(np.int32)
    iv[J, I] = ((ip[Jm, I] > 0) & (ip[J, I] > 0)).astype(np.int32)
    allfour = np.minimum(np.minimum(ip[J, I], ip[J, Im]), np.minimum(ip[Jm, I], ip[Jm, Im])) > 0
    diag = ((ip[J, I] > 0) & (ip[Jm, Im] > 0)) | ((ip[J, Im] > 0) & (ip[Jm, I] > 0))
    iq[J, I] = (allfour | diag).astype(np.int32)
    for msk, it in ((iu, 3), (iv, 4), (iq, 2)):               # halo_us, halo_vs, halo_qs, :249-251
        f = msk.astype(np.float64)[None]
        xctilr_np(f, 1, 1, NBDY, NBDY, nreg, ii, jj, itype=it)
        msk[:, :] = np.rint(f[0]).astype(np.int32)
        if not lperiodj:
            msk[:o + 1, :] = 0
            if not larctic:
                msk[o + jj + 1:, :] = 0
        if not lperiodi:
            msk[:, :o + 1] = 0
            msk[:, o + ii + 1:] = 0
    return nreg, depth, ip, iu, iv, iq


# ----------------------------------------------------------------------------------
# numerical_bounds (phy/mod_blom_init.F90:446-555)
# ----------------------------------------------------------------------------------
def numerical_bounds_np(grid, masks, baclin, nreg, ii, jj):
    scpx, scpy, scqx, scqy = grid["scpx"], grid["scpy"], grid["scqx"], grid["scqy"]
    dx2, dy2 = scpx * scpx, scpy * scpy
    difmxp = .9 * .5 * dx2 * dy2 / np.maximum(1.0, (dx2 + dy2) * (baclin + baclin))
    dx2, dy2 = scqx * scqx, scqy * scqy
    difmxq = .9 * .5 * dx2 * dy2 / np.maximum(1.0, (dx2 + dy2) * (baclin + baclin))
    scp2, scuy, scvx = grid["scp2"], grid["scuy"], grid["scvx"]
    umax = np.zeros_like(scp2)
    vmax = np.zeros_like(scp2)
    J, I = sl(1, jj), sl(1, ii)
    Jm, Im = sl(0, jj - 1), sl(0, ii - 1)
    um = .9 * .125 * np.minimum(scp2[J, Im], scp2[J, I]) / (scuy[J, I] * baclin)
    vm = .9 * .125 * np.minimum(scp2[Jm, I], scp2[J, I]) / (scvx[J, I] * baclin)
    umax[J, I] = np.where(masks["iu"][J, I] > 0, um, 0.0)
    vmax[J, I] = np.where(masks["iv"][J, I] > 0, vm, 0.0)
    for a, it in ((umax, 3), (vmax, 4)):                 # halo_us, halo_vs
        xctilr_np(a[None], 1, 1, NBDY, NBDY, nreg, ii, jj, itype=it)
    return difmxp, difmxq, umax, vmax


# ----------------------------------------------------------------------------------
# the initialisation sequence proper
# ----------------------------------------------------------------------------------
def _halo(be, name, nlev, mh, nh, nreg, ii, jj, l1=1, itype=1):
    a = be.get(name)
    xctilr_np(a, l1, nlev, mh, nh, nreg, ii, jj, itype=itype)
    be.put(name, a)


def init_state(be, case):
    """Bring backend `be` (already set up for case.depth: masks known, arrays holding
    their inivar_* patterns) to the state blom_step expects at nstep = 0."""
    kk, ii, jj, nreg = case.kdm, case.idm, case.jdm, case.nreg
    P = case.params
    ip, iu, iv, iq = (be.masks[k] for k in ("ip", "iu", "iv", "iq"))
    J, I = sl(1, jj), sl(1, ii)
    wet = ip[J, I] > 0

    for nm, v in P.items():
        if nm.endswith("0"):
            continue            # generator-only amplitudes (difiso0, taux0, ...)
        be.set(nm, v)
    be.set("nstep", 0)

    for nm, arr in case.grid.items():
        be.put(nm, arr[None])
    difmxp, difmxq, umax, vmax = numerical_bounds_np(case.grid, be.masks, P["baclin"], nreg, ii, jj)
    be.put("difmxp", difmxp[None])
    be.put("difmxq", difmxq[None])
    be.put("umax", umax[None])
    be.put("vmax", vmax[None])

    # -- layer structure at both time levels (mod_inicon.F90:1163-1186) ---------------
    for nm in ("dp", "temp", "saln", "sigma"):
        a = be.get(nm)
        src = case.ic[nm]
        for lev in (0, kk):
            a[lev:lev + kk, J, I] = np.where(wet, src[:, J, I], a[lev:lev + kk, J, I])
        be.put(nm, a)
    a = be.get("sigmar")
    a[:, J, I] = np.where(wet, case.ic["sigmar"][:, J, I], a[:, J, I])
    be.put("sigmar", a)
    if be.ntr > 0:
        a = be.get("trc")                      # (ntr*2*kk, nj, ni), Fortran trc(i,j,k,nt)
        for nt in range(be.ntr):
            for lev in (0, kk):
                b = nt * 2 * kk + lev
                a[b:b + kk, J, I] = np.where(wet, case.ic["trc"][nt][:, J, I], a[b:b + kk, J, I])
        be.put("trc", a)

    # -- first physical interior layer kfpla (mod_inicon.F90:1394-1420), bulkml only ---
    dp = be.get("dp")
    kf = be.get("kfpla")
    d1 = dp[:kk, J, I].copy()                       # vectorised over columns
    # k = first layer index >= 3 with dp >= epsilp (kk+1 if none); thin layers above it are
    # emptied and their mass (summed top-down, as the reference does) moved to layer k, or to
    # layer 2 when the column holds no interior mass
    thick = d1[2:] >= EPSILP                        # layers 3..kk
    anyt = thick.any(axis=0)
    first = np.where(anyt, thick.argmax(axis=0) + 3, kk + 1)
    dps = np.zeros_like(d1[0])
    for k in range(3, kk + 1):
        sel = wet & (k < first)
        dps = np.where(sel, dps + d1[k - 1], dps)
        d1[k - 1] = np.where(sel, 0.0, d1[k - 1])
    d1[1] = np.where(wet & (first > kk), d1[1] + dps, d1[1])
    for k in range(3, kk + 1):
        d1[k - 1] = np.where(wet & (first == k), d1[k - 1] + dps, d1[k - 1])
    dp[:kk, J, I] = np.where(wet, d1, dp[:kk, J, I])
    for lev in (0, 1):
        kf[lev, J, I] = np.where(wet, first, kf[lev, J, I])
    dp[kk:2 * kk, J, I] = np.where(wet, dp[:kk, J, I], dp[kk:2 * kk, J, I])
    be.put("dp", dp)
    be.put("kfpla", kf)

    for nm, nl in (("dp", 2 * kk), ("temp", 2 * kk), ("saln", 2 * kk), ("sigma", 2 * kk), ("sigmar", kk)):
        _halo(be, nm, nl, NBDY, NBDY, nreg, ii, jj)
    if be.ntr > 0:
        _halo(be, "trc", 2 * kk * be.ntr, NBDY, NBDY, nreg, ii, jj)

    # -- interface pressures and velocity-point thicknesses (mod_blom_init.F90:269-310) -
    dp = be.get("dp")
    p = be.get("p")
    dpu, dpv = be.get("dpu"), be.get("dpv")
    pu, pv = be.get("pu"), be.get("pv")
    Jp, Ip = sl(-2, jj + 2), sl(-2, ii + 2)
    wp = ip[Jp, Ip] > 0
    for mmt in (0, kk):                        # n = 1 first, then m = 2
        for k in range(kk):
            p[k + 1, Jp, Ip] = np.where(wp, p[k, Jp, Ip] + dp[k + mmt, Jp, Ip], p[k + 1, Jp, Ip])
        Ju, Iu = sl(-1, jj + 2), sl(-1, ii + 2)
        Jum, Ium = sl(-2, jj + 1), sl(-2, ii + 1)
        mu = iu[Ju, Iu] > 0
        mv = iv[Ju, Iu] > 0
        qu = np.minimum(p[kk, Ju, Iu], p[kk, Ju, Ium])
        qv = np.minimum(p[kk, Ju, Iu], p[kk, Jum, Iu])
        for k in range(kk):
            du = .5 * ((np.minimum(qu, p[k + 1, Ju, Ium]) - np.minimum(qu, p[k, Ju, Ium]))
                       + (np.minimum(qu, p[k + 1, Ju, Iu]) - np.minimum(qu, p[k, Ju, Iu])))
            dv = .5 * ((np.minimum(qv, p[k + 1, Jum, Iu]) - np.minimum(qv, p[k, Jum, Iu]))
                       + (np.minimum(qv, p[k + 1, Ju, Iu]) - np.minimum(qv, p[k, Ju, Iu])))
            dpu[k + mmt, Ju, Iu] = np.where(mu, du, dpu[k + mmt, Ju, Iu])
            dpv[k + mmt, Ju, Iu] = np.where(mv, dv, dpv[k + mmt, Ju, Iu])
            if mmt == 0:                       # mod_inicon.F90:1138,1148
                pu[k + 1, Ju, Iu] = np.where(mu, pu[k, Ju, Iu] + du, pu[k + 1, Ju, Iu])
                pv[k + 1, Ju, Iu] = np.where(mv, pv[k, Ju, Iu] + dv, pv[k + 1, Ju, Iu])
    for nm, a in (("p", p), ("dpu", dpu), ("dpv", dpv), ("pu", pu), ("pv", pv)):
        be.put(nm, a)

    # -- bottom pressures (mod_inicon.F90:1095-1125) ------------------------------------
    J0, I0 = sl(0, jj + 1), sl(0, ii + 1)
    w0 = ip[J0, I0] > 0
    pb, pb_mn, pb_p = be.get("pb"), be.get("pb_mn"), be.get("pb_p")
    for a in (pb, pb_mn):
        for lev in (0, 1):
            a[lev, J0, I0] = np.where(w0, p[kk, J0, I0], a[lev, J0, I0])
    pb_p[0, J0, I0] = np.where(w0, p[kk, J0, I0], pb_p[0, J0, I0])
    be.put("pb", pb)
    be.put("pb_mn", pb_mn)
    be.put("pb_p", pb_p)
    Jm, Im = sl(0, jj - 1), sl(0, ii - 1)
    pbu, pbv, pbu_p, pbv_p = be.get("pbu"), be.get("pbv"), be.get("pbu_p"), be.get("pbv_p")
    mu, mv = iu[J, I] > 0, iv[J, I] > 0
    bu = np.minimum(pb[0, J, I], pb[0, J, Im])
    bv = np.minimum(pb[0, J, I], pb[0, Jm, I])
    for lev in (0, 1):
        pbu[lev, J, I] = np.where(mu, bu, pbu[lev, J, I])
        pbv[lev, J, I] = np.where(mv, bv, pbv[lev, J, I])
    pbu_p[0, J, I] = np.where(mu, bu, pbu_p[0, J, I])
    pbv_p[0, J, I] = np.where(mv, bv, pbv_p[0, J, I])
    for nm, a in (("pbu", pbu), ("pbv", pbv), ("pbu_p", pbu_p), ("pbv_p", pbv_p)):
        be.put(nm, a)

    # -- barotropic potential vorticity (mod_inicon.F90:1192-1232) ----------------------
    corioq = case.grid["corioq"]
    pv_ = be.get("pvtrop")
    # three sweeps, later ones overriding earlier ones, and within a sweep the later loop index
    # overriding the earlier one -- applied in that order with masked assignments
    def q2(a, b):
        with np.errstate(divide="ignore"):          # land points are masked out below
            return 2. / (a + b)
    Ju0, Iu0 = sl(0, jj), sl(1, ii)                       # u-points j=0..jj, i=1..ii
    mu0 = iu[Ju0, Iu0] > 0
    qu = q2(pb_p[0, Ju0, Iu0], pb_p[0, Ju0, sl(0, ii - 1)])
    # writer (i,j') -> q(i,j'+1) first (it is overridden by writer (i,j'+1) -> q(i,j'+1) if valid)
    tgt = pv_[:, sl(1, jj + 1), Iu0]
    tgt[:] = np.where(mu0[None], (corioq[sl(1, jj + 1), Iu0] * qu)[None], tgt)
    tgt = pv_[:, Ju0, Iu0]
    tgt[:] = np.where(mu0[None], (corioq[Ju0, Iu0] * qu)[None], tgt)
    Jv0, Iv0 = sl(1, jj), sl(0, ii)                       # v-points j=1..jj, i=0..ii
    mv0 = iv[Jv0, Iv0] > 0
    qv = q2(pb_p[0, Jv0, Iv0], pb_p[0, sl(0, jj - 1), Iv0])
    tgt = pv_[:, Jv0, sl(1, ii + 1)]
    tgt[:] = np.where(mv0[None], (corioq[Jv0, sl(1, ii + 1)] * qv)[None], tgt)
    tgt = pv_[:, Jv0, Iv0]
    tgt[:] = np.where(mv0[None], (corioq[Jv0, Iv0] * qv)[None], tgt)
    mq = iq[J, I] > 0
    with np.errstate(divide="ignore", invalid="ignore"):
        qq = corioq[J, I] * 4. / (pb_p[0, J, I] + pb_p[0, J, Im] + pb_p[0, Jm, I] + pb_p[0, Jm, Im])
    tgt = pv_[:, J, I]
    tgt[:] = np.where(mq[None], qq[None], tgt)
    be.put("pvtrop", pv_)

    # -- frozen diffusivities and forcing (difest_* needs CVMix: out of scope) ----------
    wfull = ip > 0
    for nm, v in (("difiso", P["difiso0"]), ("difint", P["difint0"]), ("difdia", P["difdia0"])):
        a = be.get(nm)
        a[:] = np.where(wfull[None], np.minimum(v, difmxp[None]), a)
        be.put(nm, a)
    a = be.get("difwgt")
    a[:] = np.where(wfull[None], P["difwgt0"], a)
    be.put("difwgt", a)
    # minimum physical layer temperature (phy/mod_temmin.F90:57-66 uses -3 for non-bulkml
    # configurations; settemmin itself is an initialisation routine outside the path)
    a = be.get("temmin")
    a[:] = np.where(wfull[None], -3.0, a)
    be.put("temmin", a)
    jy = (np.arange(1, jj + 1)[:, None] - 0.5) / jj
    taux = be.get("taux")
    taux[0, J, I] = np.where(iu[J, I] > 0, P["taux0"] * np.sin(np.pi * jy) ** 2 * np.ones((jj, ii)),
                             taux[0, J, I])
    be.put("taux", taux)
    tauy = be.get("tauy")
    tauy[0, J, I] = np.where(iv[J, I] > 0, 0.0, tauy[0, J, I])
    be.put("tauy", tauy)
    for nm in ("umfltd", "vmfltd", "umflsm", "vmflsm", "utfltd", "vtfltd", "utflsm", "vtflsm",
               "usfltd", "vsfltd", "usflsm", "vsflsm"):
        a = be.get(nm)
        msk = (iu if nm[0] == "u" else iv) > 0
        a[:] = np.where(msk[None], 0.0, a)
        be.put(nm, a)

    # -- frozen isopycnal slopes read by eddtra (cmnfld2, phy/mod_cmnfld_routines.F90:1090, is out
    #    of scope): an analytic pattern of amplitude nslp0, growing with depth ------------
    if be.has_field("nslpx"):
        ji, ii_ = np.meshgrid(np.arange(1, jj + 1), np.arange(1, ii + 1), indexing="ij")
        prof = (np.arange(1, kk + 1) / kk)[:, None, None]
        for nm, msk, pat in (("nslpx", iu, np.sin(2 * np.pi * ii_ / ii) * np.sin(np.pi * ji / jj)),
                             ("nslpy", iv, np.cos(2 * np.pi * ii_ / ii) * np.sin(2 * np.pi * ji / jj))):
            a = be.get(nm)
            a[:, J, I] = np.where(msk[J, I] > 0, P.get("nslp0", 0.0) * prof * pat[None], 0.0)
            be.put(nm, a)

    # -- geopotential of the sea floor (cf. channel/mod_channel.F90:311-320) ------------
    _, depth_h, _, _, _, _ = bigrid_np(case.depth, ii, jj, arctic=nreg == 2)
    phi = be.get("phi")
    phi[kk] = np.where(wfull, -GRAV * depth_h, phi[kk])
    be.put("phi", phi)

    # -- pressure gradient force fields via the stage itself (mod_inicon.F90:1351) ------
    be.stage("pgforc", 2, 1, kk, 0, kk + 1, 1)
    for nm, msk in (("pgfx", mu), ("pgfy", mv)):
        a = be.get(nm)
        a[kk:2 * kk, J, I] = np.where(msk, a[:kk, J, I], a[kk:2 * kk, J, I])
        be.put(nm, a)
    for nm, msk in (("pgfxm", mu), ("xixp", mu), ("xixm", mu), ("pgfym", mv), ("xiyp", mv), ("xiym", mv)):
        a = be.get(nm)
        a[1, J, I] = np.where(msk, a[0, J, I], a[1, J, I])
        be.put(nm, a)

    # -- cppm coefficient tables (init_cppm, phy/mod_cppm.F90:2504, called from blom_init) ---
    if P.get("advmth") == "cppm":
        be.stage("init_cppm", 2, 1, kk, 0, kk + 1, 1)

    # -- old-level copies for the time filters (mod_inicon.F90:1426-1437, initms) --------
    dp = be.get("dp")
    dpold = be.get("dpold")
    for lev in (0, kk):
        dpold[lev:lev + kk, J, I] = np.where(wet, dp[:kk, J, I], dpold[lev:lev + kk, J, I])
    be.put("dpold", dpold)
    be.stage("initms", 2, 1, kk, 0, kk + 1, 1)

    # -- halo updates of blom_init_phase2 (mod_blom_init.F90:360-378), with the grid/field types that
    #    matter across the arctic seam (halo_ps 1, halo_qs 2, halo_us 3, halo_vs 4, halo_uv 13, halo_vv 14)
    for nm, nl, mh, nh, it in (("sigmar", kk, 2, 2, 1), ("uflx", 2 * kk, 1, 1, 13), ("vflx", 2 * kk, 1, 1, 14),
                               ("pvtrop", 2, 1, 3, 2), ("pgfxm", 2, 1, 2, 13), ("xixp", 2, 1, 2, 3),
                               ("xixm", 2, 1, 2, 3), ("pgfym", 2, 1, 2, 14), ("xiyp", 2, 1, 2, 4),
                               ("xiym", 2, 1, 2, 4), ("difiso", kk, 1, 1, 1), ("taux", 1, 1, 1, 13),
                               ("tauy", 1, 1, 1, 14)):
        _halo(be, nm, nl, mh, nh, nreg, ii, jj, itype=it)
    a = be.get("phi")
    xctilr_np(a, kk + 1, kk + 1, 2, 2, nreg, ii, jj)
    be.put("phi", a)
    if nreg == 2:
        # with the arctic patch xixp <-> xixm and xiyp <-> xiym change roles in the halo next to the
        # grid intersection (mod_blom_init.F90:380-400)
        xp, xm = be.get("xixp"), be.get("xixm")
        Js, Is = sl(jj, jj + 2), sl(0, ii + 1)
        for lev in (0, 1):
            t = xp[lev, Js, Is].copy()
            xp[lev, Js, Is] = xm[lev, Js, Is]
            xm[lev, Js, Is] = t
        be.put("xixp", xp)
        be.put("xixm", xm)
        yp, ym = be.get("xiyp"), be.get("xiym")
        o = NBDY - 1
        Iy = sl(max(0, ii // 2 + 1), ii + 1)
        J2, I2 = sl(jj + 1, jj + 2), sl(0, ii + 1)
        for lev in (0, 1):
            t = yp[lev, o + jj, Iy].copy()
            yp[lev, o + jj, Iy] = ym[lev, o + jj, Iy]
            ym[lev, o + jj, Iy] = t
            t = yp[lev, J2, I2].copy()
            yp[lev, J2, I2] = ym[lev, J2, I2]
            ym[lev, J2, I2] = t
        be.put("xiyp", yp)
        be.put("xiym", ym)


def step_indices(nstep, kk):
    """Time-level sextuple of blom_step (phy/mod_blom_step.F90:89-94) for the value of
    nstep BEFORE step_time increments it."""
    m = nstep % 2 + 1
    n = (nstep + 1) % 2 + 1
    mm = (m - 1) * kk
    nn = (n - 1) * kk
    return m, n, mm, nn, 1 + mm, 1 + nn


def init_forcing(be, case):
    """The forcing thermf_channel and mxlayr read, for the idealised channel (channel/mod_channel.F90:365-394: everything zero but
    the open-water friction velocity 0.005 m/s, the climatologies set to constants that the default relaxation time scales of 0
    days never touch), the shortwave absorption of Jerlov water type 3 (phy/mod_swabs.F90:104-107, :262-267: the defaults of
    isopyc_bulkml, cime_config/namelist_definition_blom.xml swamth / jwtype), zero friction velocity and zeroed reservoirs at the
    start (phy/mod_forcing.F90:310, phy/mod_niw.F90:85-112), the ocean area (mod_grid: area, the xcsum of scp2 over ips).
    `case.params["ustarw0"]` overrides the friction velocity: thermf_channel multiplies it by 1e2 (channel/mod_thermf_channel.F90:259),
    so the reference's 0.005 acts as 0.5 m/s on this SI state; bench.py --forcing calm sets 5e-5 (0.005 m/s after the factor)."""
    nj, ni = case.jdm + 2 * NBDY, case.idm + 2 * NBDY
    one = np.ones((1, nj, ni))
    vals = dict(ustarw=float(case.params.get("ustarw0", 0.005)), swa=0.0, nsf=0.0, hmltfz=0.0, lip=0.0, sop=0.0, eva=0.0, rnf=0.0, rfi=0.0, fmltfz=0.0, sfl=0.0,
                swfc1=0.67, swfc2=1.0 - 0.67, swal1=1.0, swal2=17.0, ustar=0.0, ustar3=0.0, idkedt=0.0,
                surflx=0.0, sswflx=0.0, surrlx=0.0, salflx=0.0, brnflx=0.0, salrlx=0.0, salt_corr=0.0)
    has = getattr(be, "has_field", lambda nm: True)
    for nm, v in vals.items():
        if has(nm):
            be.put(nm, v * one)
    for nm, v, nl in (("sstclm", 10.0, 12), ("ricclm", 0.0, 12), ("sssclm", 35.0, 12), ("uml", 0.0, 4), ("vml", 0.0, 4),
                      ("umlres", 0.0, 2), ("vmlres", 0.0, 2)):
        if has(nm):
            be.put(nm, np.full((nl, nj, ni), v))
    if case.ntr:
        for nm in ("trflx", "trc_corr"):
            if has(nm):
                be.put(nm, np.zeros((case.ntr, nj, ni)))
    area = ocean_area(be, case)
    if area is not None:
        be.set("area", area)


def ocean_area(be, case):
    """mod_grid's area: the sum of scp2 over ips (ip without the seam row of an arctic patch) of the WHOLE domain `be` holds"""
    ip = be.masks["ip"] if hasattr(be, "masks") else None
    if ip is None:
        return None
    scp2 = np.asarray(be.get("scp2"))[0]
    w = (ip[NBDY:-NBDY, NBDY:-NBDY] > 0)
    if case.nreg == 2:
        w = w.copy()
        w[-1, :] = False                          # ips: without the seam row of the arctic patch
    return float(np.sum(scp2[NBDY:-NBDY, NBDY:-NBDY][w]))


def init_indices(nstep1, kk):
    """Time-level sextuple of blom_init (phy/mod_blom_init.F90:256-261: m = mod(nstep1+1,2)+1, n = mod(nstep1,2)+1): what its
    start-up cmnfld1 is called with -- the other way round from the first step's."""
    m = (nstep1 + 1) % 2 + 1
    n = nstep1 % 2 + 1
    mm = (m - 1) * kk
    nn = (n - 1) * kk
    return m, n, mm, nn, 1 + mm, 1 + nn


def frozen_eddy_fluxes(be, case, amp=0.15, spike=3.0):
    """A synthetic, frozen field of eddy-induced mass fluxes umfltd, vmfltd (thickness diffusion) and umflsm, vmflsm
    (submesoscale) at both time levels: what eddtra would hand to advect (phy/mod_advect.F90:72-94,
    cau = ... + (umfltd+umflsm)/max(onemm,dpu)), as an analytic function of the grid and of the state the backend
    holds.  Smooth, sign-changing, a fraction `amp` of the flux area the CFL clamp allows times the layer thickness at
    the velocity point; every 97th wet velocity point carries `spike` times the clamp (either sign), so that both
    branches of the clamp act.  Land points keep what the backend has there (the reference's inivar pattern).
    The reference build used as oracle has no mod_eddtra (CVMix), so its advect otherwise only ever sees zeros here;
    tests write this field into the reference's module arrays and into the device (option eddtra_frozen)."""
    kk, ii, jj = case.kdm, case.idm, case.jdm
    baclin = case.params["baclin"]
    J, I = sl(1, jj), sl(1, ii)
    jg, ig = np.meshgrid(np.arange(1, jj + 1, dtype=np.float64), np.arange(1, ii + 1, dtype=np.float64), indexing="ij")
    cnt = (np.arange(jj * ii).reshape(jj, ii) % 97) == 13
    for comp, (flds, mask, vmx, sc, dpn) in enumerate(((("umfltd", "umflsm"), "iu", "umax", "scuy", "dpu"),
                                                      (("vmfltd", "vmflsm"), "iv", "vmax", "scvx", "dpv"))):
        wet = be.masks[mask][J, I] > 0
        clamp = be.get(vmx)[0][J, I] * (2.0 * baclin) * be.get(sc)[0][J, I]          # umax*delt1*scuy: an area
        dpv = be.get(dpn)
        for nf, nm in enumerate(flds):
            a = be.get(nm)
            for lev in range(2 * kk):
                k = lev % kk
                ph = 0.37 * k + 1.3 * (lev // kk) + 0.9 * comp + 2.1 * nf
                pat = np.sin(2.0 * np.pi * ig / 11.0 + ph) * np.cos(2.0 * np.pi * jg / 7.0 - 0.61 * ph)
                pat = (amp if nf == 0 else 0.4 * amp) * pat
                if nf == 0:
                    sgn = np.where(((ig + jg + k) % 2) == 0, 1.0, -1.0)
                    pat = np.where(cnt & ((k % 5) == 2), spike * sgn, pat)
                val = pat * clamp * dpv[k + (1 - lev // kk) * kk][J, I]
                a[lev][J, I] = np.where(wet, val, a[lev][J, I])
            be.put(nm, a)


# constants of phy/mod_difest.F90:200-203, :191, :193 read by the two host-evaluated planes below
_TDMLS0, _TDMLS1, _TDCLAT, _TDDLAT = 500.0 * 9806.0, 100.0 * 9806.0, 74.5, 3.0
_CORI30, _BVF0 = 7.2722e-5, 5.24e-3


def difest_host_planes(plat, coriop):
    """The two inputs of difest_vertical_iso that depend on the grid only and go through the libm: the tidally driven mixing length
    scale `q = .5*(tanh(4.*(abs(plat)-tdclat)/tddlat-2.)+1.); q = (1.-q)*tdmls0+q*tdmls1` (phy/mod_difest.F90:2926-2927) and, for
    the latitude dependent background mixing (bdmldp, :2747-2750), log(2 bvf0 / max(1e-9, |coriop|)) with the scalar
    log(2 bvf0 / cori30).  math.tanh / math.log are the C library's, i.e. the functions the reference's compiled Fortran calls on
    this machine; every other operation is IEEE double arithmetic in the reference's order."""
    import math
    plat = np.asarray(plat, dtype=np.float64)
    cor = np.asarray(coriop, dtype=np.float64)
    tdmls = np.empty_like(plat)
    bdmlq = np.empty_like(plat)
    fp, fc, ft, fb = plat.ravel(), cor.ravel(), tdmls.ravel(), bdmlq.ravel()
    for x in range(fp.size):
        q = .5 * (math.tanh(4. * (abs(float(fp[x])) - _TDCLAT) / _TDDLAT - 2.) + 1.)
        ft[x] = (1. - q) * _TDMLS0 + q * _TDMLS1
        qq = max(1.e-9, abs(float(fc[x]))) if np.isfinite(fc[x]) else 1.e-9
        fb[x] = math.log(2. * _BVF0 / qq)
    return tdmls, bdmlq, math.log(2. * _BVF0 / _CORI30)


def init_difest(be, case, twedon0=2.0e-3, ficem0=0.3, device=False):
    """What difest_isobml's diffusivity estimates read beside the model state, for the idealised cases: latitude, the angle of the
    grid, the topographic beta and the angle of the topography (read with rhsctp), a smooth tidal dissipation field and a sea ice concentration (synthetic:
    the reference reads them from files), zero surface buoyancy flux.  device: also the two host-evaluated planes (the reference
    evaluates them itself)."""
    nj, ni = case.jdm + 2 * NBDY, case.idm + 2 * NBDY
    y = np.linspace(-1.0, 1.0, nj)[:, None] + 0.0 * np.arange(ni)[None, :]
    x = np.linspace(0.0, 2.0 * np.pi, ni)[None, :] + 0.0 * y
    plat = (55.0 + 24.0 * y)[None]                           # crosses the critical latitude of the M2 tide (74.5 N)
    ang = 0.2 * np.sin(x) * y
    # topographic beta and the angle of the topography (the reference reads them from `tbfile`): smooth synthetic fields of the size
    # that makes the topographic Rhines scale the active one in places (5 egr / betatp ~ 1 - 30 km against a Rossby radius of ~10 km)
    betatp = (1.0e-9 * (1.0 + 0.8 * np.sin(x) * np.cos(np.pi * y)))[None]
    hangle = (0.3 + 0.7 * np.sin(2.0 * x) * y)[None]
    vals = dict(plat=plat, cosang=np.cos(ang)[None], sinang=np.sin(ang)[None], betatp=betatp, hangle=hangle,
                twedon=(twedon0 * (1.0 + 0.5 * np.cos(2.0 * x)) * (1.0 - 0.3 * y))[None], ficem=(ficem0 * (y > 0.2) * (1.0 + np.sin(x)) * 0.5)[None])
    has = getattr(be, "has_field", lambda nm: True)
    for nm, v in vals.items():
        if has(nm):
            be.put(nm, v)
    if device:
        tdmls, bdmlq, logc = difest_host_planes(plat[0], np.asarray(be.get("coriop"))[0])
        be.put("tdmls", tdmls[None])
        be.put("bdmlq", bdmlq[None])
        be.set("bdml_logc", logc)


# &DIFFUSION as far as difest_isobml reads it: NorESM's defaults for vcoord_type = 'isopyc_bulkml' (cime_config/namelist_definition_blom.xml),
# rhsctp = .true. and rhiscf = 5 included since round 6 (:1677-1715; the topographic beta is a synthetic field here, init_difest); what
# bench.py runs config 2's step with
DIFEST_NORESM = (dict(egc=2.5, eggam=200.0, eglsmn=4000.0, egmndf=50.0, egmxdf=2500.0, egidfq=1.25, rhiscf=5.0, ri0=1.2, tkepf=0.006,
                      bdmc1=5.0e-8, bdmc2=1.0e-5, iwdfac=0.06, nubmin=2.0e-6, niwgf=0.0, niwbf=0.35, niwlf=0.5),
                 dict(eddf2d=1, edsprs=0, edanis=1, redi3d=0, edfsmo=0, edritp_opt=2, edwmth_opt=1, bdmtyp=2, iwdflg=1, bdmldp=1, rhsctp=1))
