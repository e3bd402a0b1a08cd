"""cmnfld2 (isopyc_bulkml, eitmth = 'gm'): the C restatement oracle/c/cmnfld.c checked BY CONSTRUCTION -- the reference
module (phy/mod_cmnfld_routines.F90) uses mod_dia/netCDF and cannot be built here, so parity is unpinned.  Each
expectation is written from the definition of the quantity, not from the code:
  * N^2 at an interior interface = g^2 x (difference of the density of the two adjacent layers, both taken at the
    interface pressure) / (pressure distance of the layer centres);
  * the vertical filter is an implicit diffusion (an M-matrix with unit row sums): the filtered profile stays
    between the smallest and the largest unfiltered value of its column;
  * the interface geopotential is hydrostatic (dphi = -alpha dp), and where temperature and salinity are horizontally
    uniform the neutral slope is the geometric slope of the interface, (phi(i) - phi(i-1)) / g / dx."""
import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.hostinit import step_indices

GRAV, ONEM, BFSQMN = 9.806, 9806., 1.e-7


def rho(p, th, s):          # phy/mod_eos.F90:157-172 with the coefficients of :50-75
    a11, a12, a13, a14, a15, a16 = 9.9985372432159340e+02, 1.0380621928183473e+01, 1.7073577195684715e+00, -3.6570490496333680e-02, -7.3677944503527477e-03, -3.5529175999643348e-03
    b11, b12, b13 = 1.7083494994335439e-06, 7.1567921402953455e-09, 1.2821026080049485e-09
    a21, a22, a23, a24, a25, a26 = 1.0, 1.0316374535350838e-02, 8.9521792365142522e-04, -2.8438341552142710e-05, -1.1887778959461776e-05, -4.0163964812921489e-06
    b21, b22, b23 = 1.1995545126831476e-09, 5.5234008384648383e-12, 8.4310335919950873e-13
    return ((a11 + (a12 + a14 * th + a15 * s) * th + (a13 + a16 * s) * s + (b11 + b12 * th + b13 * s) * p) /
            (a21 + (a22 + a24 * th + a25 * s) * th + (a23 + a26 * s) * s + (b21 + b22 * th + b23 * s) * p))


def _setup(cfg="chan_s", backend=None):
    from oracle.coracle import COracle
    case = make_case(cfg, nslp0=0.0)
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm, arctic=case.nreg == 2)
    masks = dict(ip=ip, iu=iu, iv=iv, iq=iq)
    be = (backend or COracle)(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
    hostinit.init_state(be, case)
    return case, be, masks


def check_bfsq(be, case, masks):
    kk = case.kdm
    m, n, mm, nn, k1m, k1n = step_indices(0, kk)
    be.stage("cmnfld2", m, n, mm, nn, k1m, k1n)
    J, I = slice(4, 4 + case.jdm), slice(4, 4 + case.idm)
    p, T, S = be.get("p")[:, J, I], be.get("temp")[nn:nn + kk, J, I], be.get("saln")[nn:nn + kk, J, I]
    bi, bf = be.get("bfsqi")[:, J, I], be.get("bfsqf")[:, J, I]
    kf = be.get("kfpla")[n - 1, J, I]
    wet = (masks["ip"][J, I] > 0) & (kf == 3)
    assert wet.sum() > 100
    checked = 0
    raw_max = np.full(wet.shape, BFSQMN)
    for k in range(3, kk):                      # interface k+1 (1-based) between layers k and k+1, both with mass
        thick = (p[k + 1] - p[k] > 2 * ONEM) & (p[k] - p[k - 1] > 2 * ONEM) & wet & (k >= 3)
        pc_lo, pc_up = .5 * (p[k] + p[k + 1]), .5 * (p[k - 1] + p[k])
        if k == 3:
            continue                            # the layer above is the mixed layer: its "centre" is the ML base (:120-127)
        with np.errstate(divide="ignore", invalid="ignore"):
            want = GRAV * GRAV * (rho(p[k], T[k], S[k]) - rho(p[k], T[k - 1], S[k - 1])) / (pc_lo - pc_up)
        full = thick & (p[kk] - p[k + 1] > 1e-12)
        raw_max = np.where(full, np.maximum(raw_max, want), raw_max)
        got = bi[k]
        assert np.allclose(got[full], want[full], rtol=1e-11, atol=0), k
        checked += int(full.sum())
    assert checked > 100
    # maximum principle of the implicit filter (columns whose every interior interface was verified above)
    allthick = wet & np.all(p[4:kk + 1] - p[3:kk] > 2 * ONEM, axis=0)
    if allthick.sum() > 10:
        for k in range(2, kk + 1):
            assert np.all(bf[k][allthick] >= BFSQMN * (1 - 1e-12))
            assert np.all(bf[k][allthick] <= np.maximum(raw_max, bi.max(axis=0))[allthick] * (1 + 1e-12))


def tilt_state(be, case):
    """Temperature and salinity horizontally uniform per layer, layer thicknesses varying linearly with x."""
    kk = case.kdm
    dp, T, S = be.get("dp"), be.get("temp"), be.get("saln")
    jj0, ii0 = 4 + case.jdm // 2, 4 + case.idm // 2
    x = (np.arange(case.idm + 8) - 4.0) / case.idm
    for k in range(2 * kk):
        T[k] = T[k, jj0, ii0]
        S[k] = S[k, jj0, ii0]
        dp[k] = dp[k, jj0, ii0] * (1.0 + 0.05 * np.sin(2 * np.pi * x)[None, :] * (1 if (k % kk) % 2 else -1))
    be.put("dp", dp)
    be.put("temp", T)
    be.put("saln", S)
    p = be.get("p")
    for k in range(kk):
        p[k + 1] = p[k] + dp[k]
    be.put("p", p)
    kf = be.get("kfpla")
    kf[:] = 3
    be.put("kfpla", kf)
    phi = be.get("phi")                   # a flat sea floor: nothing but the layer thicknesses varies, and only with x
    phi[kk] = phi[kk, jj0, ii0]
    be.put("phi", phi)


def check_slope_of_tilted_layers(be, case, masks):
    kk = case.kdm
    tilt_state(be, case)
    m, n, mm, nn, k1m, k1n = step_indices(1, kk)           # n = 1: the levels tilt_state filled first
    be.stage("cmnfld2", m, n, mm, nn, k1m, k1n)
    J, I = slice(4, 4 + case.jdm), slice(5, 4 + case.idm)
    Iw = slice(4, 3 + case.idm)
    phi, ns = be.get("phi"), be.get("nslpx")
    scuxi = be.get("scuxi")[0]
    wet = masks["iu"][J, I] > 0
    for k in range(4, kk + 1):                               # interior interfaces (1-based k)
        want = ((phi[k - 1][J, I] - phi[k - 1][J, Iw]) / GRAV) * scuxi[J, I]
        got = ns[k - 1][J, I]
        sel = wet & (got != 0.0)
        assert sel.sum() > 50
        assert np.array_equal(got[sel], want[sel]), k       # rho_x = 0 exactly: the slope is the interface's tilt
    assert np.abs(ns).max() > 1e-7
    assert not np.any(be.get("nslpy")[:, 4:4 + case.jdm, 4:4 + case.idm] != 0.0)     # nothing varies with y
    # the geopotential is hydrostatic: a midpoint-rule integration with alpha = 1/rho agrees to the rule's accuracy
    p, T, S = be.get("p"), be.get("temp")[nn:nn + kk], be.get("saln")[nn:nn + kk]
    ph = phi[kk].copy()
    for k in range(kk - 1, 1, -1):
        ph = ph + (p[k + 1] - p[k]) / rho(.5 * (p[k] + p[k + 1]), T[k], S[k])
        sel = masks["ip"] > 0
        assert np.allclose(ph[4:-4, 4:-4][sel[4:-4, 4:-4]], phi[k][4:-4, 4:-4][sel[4:-4, 4:-4]], rtol=0, atol=2e-6 * np.abs(phi[kk]).max())


def test_bfsq_from_its_definition():
    case, be, masks = _setup()
    check_bfsq(be, case, masks)


def test_slope_of_tilted_layers():
    case, be, masks = _setup()
    check_slope_of_tilted_layers(be, case, masks)


def check_depths(be, case, masks):
    """cmnfld1 (cmnfld_z, phy/mod_cmnfld_routines.F90:885-921) from the definitions: the sea floor lies at -phi/g; an
    interface lies the hydrostatic thickness of the layer above the one below it -- alpha dp / g with the specific volume
    at the layer's mean pressure, to the accuracy of that mid-point rule; layers without mass have no thickness; dz is the
    difference of the interface depths."""
    kk = case.kdm
    m, n, mm, nn, k1m, k1n = step_indices(0, kk)
    be.stage("cmnfld2", m, n, mm, nn, k1m, k1n)                 # phi of the columns (hydrostatic, checked above)
    phi = be.get("phi")
    dpm = be.get("dp")
    dpm[mm:mm + kk] = dpm[nn:nn + kk]                           # cmnfld1 works on time level m: give it the same state
    be.put("dp", dpm)
    for nm in ("temp", "saln"):
        a = be.get(nm)
        a[mm:mm + kk] = a[nn:nn + kk]
        be.put(nm, a)
    be.stage("cmnfld1", m, n, mm, nn, k1m, k1n)
    J, I = slice(4, 4 + case.jdm), slice(4, 4 + case.idm)
    wet = masks["ip"][J, I] > 0
    z, dz = be.get("z")[:, J, I], be.get("dz")[:, J, I]
    p, dp = be.get("p")[:, J, I], be.get("dp")[mm:mm + kk, J, I]
    T, S = be.get("temp")[mm:mm + kk, J, I], be.get("saln")[mm:mm + kk, J, I]
    assert np.array_equal(z[kk][wet], (-phi[kk, J, I] / GRAV)[wet])
    for k in range(kk):
        assert np.array_equal(dz[k][wet], (z[k + 1] - z[k])[wet])
        empty = wet & (dp[k] < 1e-12)
        assert not np.any(dz[k][empty] != 0.0)
        full = wet & (dp[k] > ONEM)
        if not full.any():
            continue
        want = (p[k + 1] - p[k]) / rho(.5 * (p[k] + p[k + 1]), T[k], S[k]) / GRAV
        assert np.all(np.abs(dz[k][full] - want[full]) <= 1e-5 * want[full]), k      # mid-point rule over layers up to 400 m thick
    # the surface ends up where the geopotential of the surface puts it (same hydrostatic integral, other routine)
    assert np.all(np.abs(z[0][wet] + phi[0, J, I][wet] / GRAV) <= 1e-9 * np.abs(z[kk][wet]) + 1e-9)


def test_cmnfld1_depths_from_their_definition():
    case, be, masks = _setup()
    check_depths(be, case, masks)
