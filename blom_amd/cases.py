"""Synthetic, fully analytic input cases for the BLOM dynamical-core hot path.

Host-side (numpy) input generation only: grid metrics, bathymetry, initial layer
structure and the namelist-type options each stage reads.  The shapes follow
BASELINE.json's configs:

  fuk95    156x32x12  closed in i / periodic in j (nreg=4)  bld/fuk95/patch.input.1
  channel  208x512x53 periodic in i / closed in j (nreg=1)  bld/channel/patch.input.1
  chan_s   20x24x6    small channel used by the parity tests
  chan_m   80x40x8    channel spanning several device tiles (kernel-variant tests)
  tnx2v1s  180x193x53 synthetic stand-in for the tnx2v1 tripolar production grid (nreg=2)
  tnx1v4s  360x385x53 synthetic stand-in for the tnx1v4 grid (nreg=2); with ntr > 3 the extra tracers stand in for iHAMOCC's
  tri_s    24x20x6    small ocean with the arctic patch of the tripolar grids (nreg=2): closed in the south,
                      folded onto itself across the last row, periodic in i
  box_s    24x20x8    small closed basin with an island and a promontory coast
  per_s    24x24x6    doubly periodic (nreg = 3): halo corner and N/S-exchange tests

The idealised definitions mirror the spirit of the reference's test cases
(fuk95/mod_fuk95.F90:122-447 flat-bottom front; channel/mod_channel.F90:61-323 tanh
shelves + tanh stratification) but are restated in SI units and kept netCDF-free.
The channel namelists &IDLGEO/&IDLINI/&IDLFOR have no defaults anywhere in the
reference, so the values below ARE this project's published choice.

Array convention everywhere on the host side: numpy C-order (nlev, jdm+8, idm+8);
Fortran a(i,j,k) is arr[k-1, j+3, i+3]  (nbdy = 4, phy/mod_xc.F90:45).
"""
from dataclasses import dataclass, field
import numpy as np

NBDY = 4
ONEM = 9806.0          # phy/mod_constants.F90:49
GRAV = 9.806
SPVAL = 1.0e33

# Coefficients of the functional fit of in situ density, phy/mod_eos.F90:36-54
_A11, _A12, _A13, _A14, _A15, _A16 = (9.9985372432159340e+02, 1.0380621928183473e+01,
                                       1.7073577195684715e+00, -3.6570490496333680e-02,
                                       -7.3677944503527477e-03, -3.5529175999643348e-03)
_B11, _B12, _B13 = 1.7083494994335439e-06, 7.1567921402953455e-09, 1.2821026080049485e-09
_A21, _A22, _A23, _A24, _A25, _A26 = (1.0, 1.0316374535350838e-02, 8.9521792365142522e-04,
                                       -2.8438341552142710e-05, -1.1887778959461776e-05,
                                       -4.0163964812921489e-06)
_B21, _B22, _B23 = 1.1995545126831476e-09, 5.5234008384648383e-12, 8.4310335919950873e-13
_ALPHA0 = 1.0e-3


def eos_pref_coeffs(pref):
    """ap11..ap26 of inieos (phy/mod_eos.F90:105-116)."""
    ap21 = _A21 + _B21 * pref
    ap22 = _A22 + _B22 * pref
    ap23 = _A23 + _B23 * pref
    ap24, ap25, ap26 = _A24, _A25, _A26
    ap11 = _A11 + _B11 * pref - ap21 / _ALPHA0
    ap12 = _A12 + _B12 * pref - ap22 / _ALPHA0
    ap13 = _A13 + _B13 * pref - ap23 / _ALPHA0
    ap14 = _A14 - ap24 / _ALPHA0
    ap15 = _A15 - ap25 / _ALPHA0
    ap16 = _A16 - ap26 / _ALPHA0
    return (ap11, ap12, ap13, ap14, ap15, ap16, ap21, ap22, ap23, ap24, ap25, ap26)


def sig(th, s, pref):
    """Potential density in sigma units (phy/mod_eos.F90:191-203)."""
    ap11, ap12, ap13, ap14, ap15, ap16, ap21, ap22, ap23, ap24, ap25, ap26 = eos_pref_coeffs(pref)
    return ((ap11 + (ap12 + ap14 * th + ap15 * s) * th + (ap13 + ap16 * s) * s)
            / (ap21 + (ap22 + ap24 * th + ap25 * s) * th + (ap23 + ap26 * s) * s))


def tofsig(sg, s, pref):
    """Potential temperature from potential density and salinity (mod_eos.F90:346-364)."""
    ap11, ap12, ap13, ap14, ap15, ap16, ap21, ap22, ap23, ap24, ap25, ap26 = eos_pref_coeffs(pref)
    a = ap14 - ap24 * sg
    b = ap12 - ap22 * sg + (ap15 - ap25 * sg) * s
    c = ap11 - ap21 * sg + (ap13 - ap23 * sg + (ap16 - ap26 * sg) * s) * s
    return (-b - np.sqrt(b * b - 4.0 * a * c)) / (2.0 * a)


@dataclass
class Case:
    name: str
    idm: int
    jdm: int
    kdm: int
    nreg: int                       # phy/mod_bigrid.F90:81-95 region type
    params: dict                    # namelist-type options (reals, ints, strings)
    depth: np.ndarray               # (nj, ni) incl. halo, 0 = land, interior only significant
    grid: dict = field(default_factory=dict)   # metric arrays (nj, ni)
    ic: dict = field(default_factory=dict)     # initial wet-point state, see hostinit.py
    ntr: int = 1

    @property
    def ni(self):
        return self.idm + 2 * NBDY

    @property
    def nj(self):
        return self.jdm + 2 * NBDY


_DIMS = {
    # name: (idm, jdm, kdm, nreg, dx[m], baclin, batrop)
    "chan_s": (20, 24, 6, 1, 10.0e3, 900.0, 18.0),
    "chan_m": (80, 40, 8, 1, 10.0e3, 900.0, 18.0),     # several 32x8 device tiles, periodic in i
    # the channel over 4 x 3 of barotp's 26 x 16 tiles, the last column 6 wide, the last row 8 high (k_bt_steps4: ragged tiles, periodic seam)
    "chan_b": (84, 40, 6, 1, 10.0e3, 900.0, 18.0),
    "tri_s": (24, 20, 6, 2, 10.0e3, 900.0, 18.0),      # periodic in i, arctic patch along the last row (nreg = 2)
    # the same topology over several 26 x 16 tiles of barotp's persistent kernel (3 x 3, last column 12 wide, last row 8 high)
    "tri_m": (64, 40, 6, 2, 10.0e3, 900.0, 18.0),
    # synthetic stand-in for the tnx2v1 production grid: its dimensions, region type
    # and time steps (baclin 4800 s, batrop 96 s => lstep 50); analytic continents instead of grid.nc
    "tnx2v1s": (180, 193, 53, 2, 100.0e3, 4800.0, 96.0),
    "box_s": (24, 20, 8, 0, 10.0e3, 900.0, 18.0),
    "per_s": (24, 24, 6, 3, 10.0e3, 900.0, 18.0),      # doubly periodic f-plane with an island and a seamount
    "fuk95": (156, 32, 12, 4, 650.0, 180.0, 6.0),
    # the reference's own test case restated from its generator routines, see fuk95_ref_case below
    "fuk95_ref": (156, 32, 12, 4, 650.0, 180.0, 6.0),
    "channel": (208, 512, 53, 1, 10.0e3, 900.0, 18.0),
    # synthetic stand-in for the tnx1v4 production grid (BASELINE.json config 5; bld/tnx1v4/patch.input.32: 360 x 385, kdm 53
    # for isopyc_bulkml, nreg = 2; baclin 3200 s, batrop 64 s): tnx2v1s' analytic continents at these dimensions
    "tnx1v4s": (360, 385, 53, 2, 50.0e3, 3200.0, 64.0),
    # a channel of the size of ONE tile of the 2 x 4 decomposition of `channel` (bench.py --gpus 8): what a rank of that run
    # computes per step, for timing on one GPU (tools/); not a configuration of the reference
    "chan_t8": (104, 128, 53, 1, 10.0e3, 900.0, 18.0),
}


def default_params(baclin, batrop):
    lstep = 2 * int(np.ceil(0.5 * baclin / batrop))       # phy/mod_time.F90:139
    return dict(
        expcnf="channel",
        baclin=baclin, batrop=batrop, lstep=lstep, dlt=baclin / lstep,  # mod_time.F90:142
        delt1=baclin,                                      # forward first step, mod_blom_init.F90:231
        pref=2000.0e4,
        mdv2hi=0.02, mdv2lo=0.004, mdv4hi=0.005, mdv4lo=0.005,
        mdc2hi=5000.0, mdc2lo=300.0,
        vsc2hi=0.5, vsc2lo=0.5, vsc4hi=0.06, vsc4lo=0.06,
        cbar=0.05, cb=0.002, cwbdts=5.0e-5, cwbdls=25.0,
        mommth="enscon", pgfmth="geopotential", bmcmth="uc", advmth="remap", eitmth="gm",
        vcoord_tag=1,        # vcoord_isopyc_bulkml, phy/mod_vcoord.F90
        ltedtp_opt=1,        # ltedtp_layer, phy/mod_diffusion.F90
        bdmtyp=2, bdmc1=5.0e-8, bdmc2=1.0e-5, iwdflg=1, iwdfac=0.06, nubmin=1.0e-6,
        bdmldp=0,
        # frozen diffusivities (difest needs CVMix, absent: SURVEY.md 8c)
        difiso0=300.0, difint0=300.0, difdia0=1.0e-5, difwgt0=1.0,
        nday_in_year=365,    # mod_time (calendar): scales the ideal age increment, idlage/mod_idlage.F90:81
        itriag=1,            # index of the ideal age tracer (trc/mod_tracers.F90:100 with -DTRC -DIDLAGE)
        itrtke=-1, itrgls=-1, tkeadv=1, tkeidf=0, gls=0,   # phy/mod_ifdefs.F90:16-35 without -DTKE
        taux0=0.1,           # zonal wind stress amplitude [N m-2]
        nslp0=0.0,           # amplitude of the frozen isopycnal slopes nslpx/nslpy [] (cmnfld2 is out of scope)
    )


def _depth_for(name, idm, jdm, dx):
    """Bathymetry [m] on the interior (jdm, idm); 0 = land."""
    ii = np.arange(1, idm + 1)[None, :]
    jj = np.arange(1, jdm + 1)[:, None]
    if name in ("chan_s", "chan_m", "chan_b", "channel", "chan_t8"):
        # tanh shelves on both walls (cf. channel/mod_channel.F90:168-207), southern and
        # northern-most rows land
        sf, sl = (200.0, 800.0) if name in ("chan_s", "chan_m", "chan_b") else (200.0, 3800.0)
        width = 0.18 * jdm * dx
        ys = (jj - 0.5) * dx
        yn = (jdm - jj + 0.5) * dx
        y = np.minimum(ys, yn)
        d = sf + 0.5 * sl * (1.0 + np.tanh(np.pi * (y - 1.2 * width) / width))
        # gentle along-channel corrugation so that fields vary in i as well
        d = d * (1.0 + 0.05 * np.sin(2.0 * np.pi * ii / idm) * np.exp(-((y - 1.2 * width) / width) ** 2))
        d = np.broadcast_to(d, (jdm, idm)).copy()
        d[0, :] = 0.0
        d[-1, :] = 0.0
        return d
    if name in ("tnx2v1s", "tnx1v4s", "tri_m"):
        x = (ii - 0.5) / idm
        y = (jj - 0.5) / jdm
        d = 3000.0 + 1500.0 * np.sin(2.0 * np.pi * x) * np.sin(np.pi * y) + 0.0 * (ii + jj)
        if name == "tri_m":
            d = d * 0.2
        d[0:4, :] = 0.0                                           # antarctic coast
        # two meridional continents with shelves, one reaching the arctic seam
        for x0, w, j0, j1 in ((0.20, 0.055, 0.25, 0.80), (0.62, 0.07, 0.30, 1.01)):
            land = (np.abs(x - x0) < w * (0.6 + 0.4 * np.sin(np.pi * (y - j0) / (j1 - j0)))) & (y > j0) & (y < j1)
            shelf = (np.abs(x - x0) < 1.6 * w) & (y > j0 - 0.03) & (y < j1 + 0.03)
            d = np.where(shelf, np.minimum(d, 400.0 if name != "tri_m" else 150.0), d)
            d = np.where(land, 0.0, d)
        return np.broadcast_to(d, (jdm, idm)).copy()
    if name == "tri_s":
        # open ocean up to the arctic seam, a southern coast, one continent reaching the seam; the last
        # row is overwritten with the mirror image of the row below it by the arctic halo rule
        x = (ii - 0.5) / idm
        y = (jj - 0.5) / jdm
        d = 400.0 + 400.0 * (0.5 + 0.5 * np.sin(2.0 * np.pi * x)) * np.sin(0.5 * np.pi * y) + 0.0 * (ii + jj)
        d[0, :] = 0.0
        d[1:3, 3:9] = 0.0
        d[jdm - 6:, 15:19] = 0.0           # land touching the seam
        d[8:11, 5:8] = 0.0                 # island
        return d
    if name == "per_s":
        x = (ii - 0.5) / idm
        y = (jj - 0.5) / jdm
        d = 700.0 - 300.0 * np.exp(-((x - 0.7) ** 2 + (y - 0.3) ** 2) / 0.02) + 0.0 * (ii + jj)
        d[10:13, 6:10] = 0.0               # island
        return d
    if name == "fuk95":
        d = np.full((jdm, idm), 200.0)     # fuk95/mod_fuk95.F90:126-134 flat, walls in i
        d[:, 0] = 0.0
        d[:, -1] = 0.0
        return d
    if name == "box_s":
        d = np.full((jdm, idm), 900.0)
        x = (ii - 0.5) / idm
        y = (jj - 0.5) / jdm
        d = 300.0 + 600.0 * np.sin(np.pi * x) * np.sin(np.pi * y) + 0.0 * d
        d[0, :] = 0.0
        d[-1, :] = 0.0
        d[:, 0] = 0.0
        d[:, -1] = 0.0
        # irregular coast: a promontory, a bay, an island (no 1-point inlets,
        # phy/mod_bigrid.F90:165-193)
        d[1:4, 1:6] = 0.0
        d[1:3, 14:19] = 0.0
        d[jdm - 5:jdm - 1, 8:12] = 0.0
        d[9:12, 11:14] = 0.0               # island
        d[6:8, 20:23] = 0.0
        return d
    raise KeyError(name)


# ---- the reference's fuk95 test case, from its own generator routines --------------------------------------------
_RHO0_REF, _PI_REF, _RADIAN_REF, _REARTH_REF = 1.e3, 3.1415926536, 57.295779513, 6.37122e6   # phy/mod_constants.F90:30-44


def _rho(p, th, s):
    """in situ density, phy/mod_eos.F90:157-172"""
    return ((_A11 + (_A12 + _A14 * th + _A15 * s) * th + (_A13 + _A16 * s) * s + (_B11 + _B12 * th + _B13 * s) * p)
            / (_A21 + (_A22 + _A24 * th + _A25 * s) * th + (_A23 + _A26 * s) * s + (_B21 + _B22 * th + _B23 * s) * p))


def _delphi(p1, p2, th, s):
    """delphi(p1,p2,th,s) -> (dphi, alp2), phy/mod_eos.F90:478-529"""
    a1 = _A11 + (_A12 + _A14 * th + _A15 * s) * th + (_A13 + _A16 * s) * s
    a2 = _A21 + (_A22 + _A24 * th + _A25 * s) * th + (_A23 + _A26 * s) * s
    b1 = _B11 + _B12 * th + _B13 * s
    b2 = _B21 + _B22 * th + _B23 * s
    pm = .5 * (p2 + p1)
    r = .5 * (p2 - p1) / (a1 + b1 * pm)
    q = b1 * r
    qq = q * q
    dphi = -2. * r * (a2 + b2 * pm + (a2 - a1 * b2 / b1) * qq * (1. / 3. + qq * (1. / 5. + qq * (1. / 7. + qq * (1. / 9.)))))
    return dphi, (a2 + b2 * p2) / (a1 + b1 * p2)


def _getpl(th, s, phiu, phil, pup):
    """pressure at the lower interface of a layer from the geopotentials of its interfaces, phy/mod_inicon.F90:105-137
    (vectorised: every column iterates until its own correction is below the reference's 1e-5)"""
    plo = pup - _rho(pup, th, s) * (phil - phiu)
    q = np.ones_like(plo)
    for _ in range(50):
        act = np.abs(q) > 1.e-5
        if not act.any():
            break
        dphi, alpl = _delphi(pup, plo, th, s)
        qn = (phil - phiu - dphi) / alpl
        q = np.where(act, qn, q)
        plo = np.where(act, plo - qn, plo)
    return plo


def sofsig(sg, th, pref):
    """salinity from potential density and temperature, phy/mod_eos.F90:366-384"""
    ap11, ap12, ap13, ap14, ap15, ap16, ap21, ap22, ap23, ap24, ap25, ap26 = eos_pref_coeffs(pref)
    a = ap16 - ap26 * sg
    b = ap13 - ap23 * sg + (ap15 - ap25 * sg) * th
    c = ap11 - ap21 * sg + (ap12 - ap22 * sg + (ap14 - ap24 * sg) * th) * th
    return (-b + np.sqrt(b * b - 4. * a * c)) / (2. * a)


# tests/fuk95/limits of the reference: the options of its `run fuk95` test that the dynamical core reads (the vertical
# coordinate is taken as isopyc_bulkml with remap/geopotential, SURVEY.md 8d config 1, second variant; the file's own
# cntiso_hybrid + cppm + dynamic enthalpy needs the ALE stack)
FUK95_LIMITS = dict(pref=0., baclin=180., batrop=6., mdv2hi=0., mdv2lo=0., mdv4hi=0., mdv4lo=0., mdc2hi=0., mdc2lo=0.,
                    vsc2hi=.2, vsc2lo=.2, vsc4hi=0., vsc4lo=0., cbar=.05, cb=.002, cwbdts=0., cwbdls=25.,
                    mommth="enscon", bmcmth="uc", eitmth="gm", bdmtyp=2, bdmc1=5.e-8, bdmc2=1.e-5, bdmldp=0, iwdflg=1,
                    iwdfac=.06, nubmin=1.e-6, expcnf="fuk95")


def fuk95_ref_case(ntr=1, **overrides):
    """The reference's fuk95 case from its generator: geoenv_fuk95 (fuk95/mod_fuk95.F90:117-229: flat channel of depth
    h0 closed at i = 1 and i = itdm, grid spacing lambda/jtdm, f-plane) and inicon_fuk95 for isopyc_bulkml (:262-338,
    :412-445: reference densities, the front of Fukamachi et al. (1995) in the interface depths, mixed layer of
    thickness mltmin, zero velocity), followed by the steps of the generic inicon that turn them into a model state
    (phy/mod_inicon.F90:985-1095: freezing-point floor, salinity of the isopycnic layers from sigmar, interface
    pressures by getpl, layer thickness).  Options: tests/fuk95/limits.  Single tile (i0 = j0 = 0)."""
    idm, jdm, kdm, nreg, dx, baclin, batrop = _DIMS["fuk95_ref"]
    kk, itdm, jtdm = kdm, idm, jdm
    ni, nj = idm + 2 * NBDY, jdm + 2 * NBDY
    u0, h1, h0, l0, drho, rhoc, rhob, f = .3, 1.e2, 2.e2, 2.e4, 0.19, 1025.9, 1027.0, 1.e-4     # :44-56
    lat0, lam, mindz, saln0, mltmin = 45., 20.8e3, 1., 35., 5.                                  # mltmin: phy/mod_mxlayr.F90:73
    rho0, pi, grav = _RHO0_REF, _PI_REF, GRAV
    p = default_params(baclin, batrop)
    p.update(FUK95_LIMITS)
    p.update(taux0=0.0, nslp0=0.0)
    p.update(overrides)
    pref = p["pref"]
    depth = np.zeros((nj, ni))
    d = np.full((jdm, idm), h0)
    d[:, 0] = 0.0
    d[:, -1] = 0.0
    depth[NBDY:NBDY + jdm, NBDY:NBDY + idm] = d
    gs = lam / jtdm
    g, one = {}, np.ones((nj, ni))
    for nm in ("scqx", "scqy", "scpx", "scpy", "scux", "scuy", "scvx", "scvy"):
        g[nm] = gs * one
    for nm in ("scq2", "scp2", "scu2", "scv2"):
        g[nm] = (gs * gs) * one
    for nm, src in (("scq2i", "scq2"), ("scp2i", "scp2"), ("scuxi", "scux"), ("scuyi", "scuy"), ("scvxi", "scvx"), ("scvyi", "scvy")):
        g[nm] = 1.0 / g[src]
    g["corioq"] = f * one
    g["coriop"] = f * one
    g["betafp"] = (f / (np.tan(lat0 / _RADIAN_REF) * _REARTH_REF)) * one

    def x_nudge(ri, rj):                                                               # :66-76
        return (ri - itdm // 2 - .5 + .1 * np.sin(2. * (rj - 1) * pi / jtdm)) * lam / jtdm

    def x_psi(x):                                                                      # :94-108
        return np.where(x <= -l0, -.5 * l0, np.where(x >= l0, .5 * l0, .5 * (x + l0 / pi * np.sin(pi * x / l0))))

    drhojet = rhoc * f * u0 * l0 / (grav * h1)                                         # :283-290
    dsig = (drho + drhojet) / (kk - 4)
    sigref = np.zeros(kk)
    sigref[kk - 1] = rhob - rho0
    sigref[kk - 2] = rhoc + .5 * (drho + drhojet) - rho0
    for k in range(kk - 2, 0, -1):
        sigref[k - 1] = sigref[k] - dsig
    jj_, ii_ = np.meshgrid(np.arange(1., jdm + 1), np.arange(1., idm + 1), indexing="ij")
    x = x_nudge(ii_, jj_)
    sigmar = np.broadcast_to(sigref[:, None, None], (kk, jdm, idm)).copy()
    sigma = sigmar.copy()
    saln = np.full((kk, jdm, idm), saln0)
    temp = tofsig(sigma, saln, pref)
    z = np.zeros((kk + 1, jdm, idm))
    z[1] = .5 * mltmin
    z[2] = mltmin
    z[kk - 1] = h1
    z[kk] = h0
    sigm = rhoc * (1. + f * u0 * x_psi(x) / (grav * h1)) - rho0
    sigma[0] = sigm + .5 * drho * (z[1] + z[0] - h1) / h1
    sigma[1] = sigm + .5 * drho * (z[2] + z[1] - h1) / h1
    temp[0] = tofsig(sigma[0], saln[0], pref)
    temp[1] = tofsig(sigma[1], saln[1], pref)
    for k in range(4, kk):                                                              # :320-332
        sigi = .5 * (sigref[k - 2] + sigref[k - 1])
        zk = ((sigi - sigm) / drho + .5) * h1
        z[k - 1] = np.minimum(z[kk - 1] - mindz * (kk - k), np.maximum(z[2], zk))
    phi = -grav * z                                                                     # :433-443
    # generic part, phy/mod_inicon.F90:985-1095 (atf = -0.0547, btf = ctf = 0 for expcnf = 'fuk95', phy/mod_eos.F90:137-141)
    tfrz = -0.0547 * saln
    temp = np.maximum(tfrz, temp)
    saln[2:] = sofsig(sigmar[2:], temp[2:], pref)
    sigma = sig(temp, saln, pref)
    pint = np.zeros((kk + 1, jdm, idm))
    pint[0] = _getpl(temp[0], saln[0], 0., phi[0], 0.)
    for k in range(kk):
        pint[k + 1] = _getpl(temp[k], saln[k], phi[k], phi[k + 1], pint[k])
    dp = np.diff(pint, axis=0)
    wet = d > 0.0
    dp[:, ~wet] = 0.0
    kgrid = np.arange(1, kk + 1)
    trc = (1.0 + 0.5 * np.sin(2 * np.pi * (ii_ - .5) / idm) * np.cos(2 * np.pi * (jj_ - .5) / jdm))[None] * (1.0 + 0.1 * kgrid[:, None, None])

    def pad3(a):
        out = np.zeros((a.shape[0], nj, ni))
        out[:, NBDY:NBDY + jdm, NBDY:NBDY + idm] = a
        return out
    ic = dict(dp=pad3(dp), temp=pad3(temp), saln=pad3(saln), sigma=pad3(sigma), sigmar=pad3(sigmar),
              trc=pad3(trc)[None].repeat(max(ntr, 1), axis=0)[:ntr], z=pad3(z), phi=pad3(phi))
    return Case(name="fuk95_ref", idm=idm, jdm=jdm, kdm=kdm, nreg=nreg, params=p, depth=depth, grid=g, ic=ic, ntr=ntr)


def make_case(name, ntr=None, carve=None, **overrides):
    """`<grid>_tke`: the grid with the reference's default tracer set (meson_options.txt:17-21: turbclo = oneeq +
    advection, iage => -DTKE -DTKEADV -DIDLAGE): ntr = 3 = TKE, the generic-length-scale slot, ideal age
    (trc/mod_tracers.F90:85-127).  Without the suffix: the -DTRC -DIDLAGE build, ntr = 1."""
    if name == "fuk95_ref":
        return fuk95_ref_case(ntr=1 if ntr is None else ntr, **overrides)
    full_name = name
    tk2 = name.endswith("_tk2")          # turbclo = twoeq, advection, isodif: -DTKE -DGLS -DTKEADV -DTKEIDF
    tk0 = name.endswith("_tk0")          # turbclo = oneeq: -DTKE alone, TKE tracers not advected
    tke = name.endswith("_tke") or tk2 or tk0
    if tke:
        name = name[:-4]
    if ntr is None:
        ntr = 3 if tke else 1
    idm, jdm, kdm, nreg, dx, baclin, batrop = _DIMS[name]
    ni, nj = idm + 2 * NBDY, jdm + 2 * NBDY
    p = default_params(baclin, batrop)
    if tke:
        p.update(itrtke=1, itrgls=2, itriag=3, tkeadv=0 if tk0 else 1, tkeidf=1 if tk2 else 0, gls=1 if tk2 else 0)
    if name == "fuk95":
        p.update(expcnf="fuk95", taux0=0.0, cwbdts=0.0)
    p.update(overrides)

    depth = np.zeros((nj, ni))
    depth[NBDY:NBDY + jdm, NBDY:NBDY + idm] = _depth_for(name, idm, jdm, dx)
    if carve is not None:          # test hook: edit the interior bathymetry (0 = land) before the state is derived from it
        carve(depth[NBDY:NBDY + jdm, NBDY:NBDY + idm])

    # ---- grid metrics: uniform Cartesian f-plane (channel/mod_channel.F90:140-161) ----
    g = {}
    one = np.ones((nj, ni))
    for nm in ("scqx", "scqy", "scpx", "scpy", "scux", "scuy", "scvx", "scvy"):
        g[nm] = dx * one
    for nm in ("scq2", "scp2", "scu2", "scv2"):
        g[nm] = (dx * dx) * one
    for nm, src in (("scq2i", "scq2"), ("scp2i", "scp2"), ("scuxi", "scux"), ("scuyi", "scuy"),
                    ("scvxi", "scvx"), ("scvyi", "scvy")):
        g[nm] = 1.0 / g[src]
    f0 = 1.0e-4
    g["corioq"] = f0 * one
    g["coriop"] = f0 * one
    g["betafp"] = 0.0 * one

    # ---- initial layer structure on the interior ------------------------------------
    pref = p["pref"]
    S0 = 35.0
    k = np.arange(1, kdm + 1)
    # reference potential densities [kg m-3, sigma units], tanh profile
    # (cf. channel/mod_channel.F90:245-253)
    sigmr0 = 33.0 + 4.2 * (k - 1) / max(1, kdm - 1) + 0.6 * np.tanh(3.0 * (k - 1) / kdm)
    sigmr0[0] = sigmr0[1] = sigmr0[2] - 0.4 if kdm > 2 else sigmr0[0]
    maxdep = depth.max()
    # nominal interior layer thickness [m]: thin on top, thicker below, sum > max depth
    dz0 = np.empty(kdm)
    dz0[0] = dz0[1] = 10.0
    w = np.tanh(2.5 * (k[2:] - 1) / kdm)
    dz0[2:] = w / w.sum() * (1.15 * maxdep - 20.0)

    jj_, ii_ = np.meshgrid(np.arange(1, jdm + 1), np.arange(1, idm + 1), indexing="ij")
    x = (ii_ - 0.5) / idm
    y = (jj_ - 0.5) / jdm
    dep = depth[NBDY:NBDY + jdm, NBDY:NBDY + idm]
    wet = dep > 0.0

    z = np.zeros((kdm + 1, jdm, idm))
    for kk_ in range(kdm):
        # smooth interface undulations (a front across the domain + a wave) that decay
        # with depth; drives pressure gradients, geostrophic adjustment and advection
        amp = 0.35 * dz0[kk_] if kk_ >= 2 else 0.0
        if name == "fuk95":
            pert = amp * (np.tanh((x - 0.5) * 12.0) + 0.3 * np.sin(2 * np.pi * y) * np.exp(-((x - 0.5) * 6) ** 2))
        else:
            pert = amp * (np.tanh((y - 0.5) * 8.0) + 0.3 * np.sin(2 * np.pi * x * 2) * np.exp(-((y - 0.5) * 5) ** 2))
        z[kk_ + 1] = np.minimum(dep, z[kk_] + np.maximum(0.0, dz0[kk_] + pert))
    # (cf. channel/mod_channel.F90:296-306) collapse thin slivers above the bottom
    for kk_ in range(2, kdm):
        thin = (dep - z[kk_]) < 1.0e-3
        z[kk_][thin] = dep[thin]
    z[kdm] = dep
    dz = np.diff(z, axis=0)
    dz[:, ~wet] = 0.0

    dp = ONEM * dz                                   # [Pa], 1 m = 9806 Pa
    sigmar = np.broadcast_to(sigmr0[:, None, None], (kdm, jdm, idm)).copy()
    saln = np.full((kdm, jdm, idm), S0)
    temp = tofsig(sigmar, saln, pref)
    # mixed layer (layers 1-2) slightly lighter and laterally varying
    dsg = 0.25 * (1.0 + np.cos(2 * np.pi * (y if name != "fuk95" else x)))
    for kk_ in range(min(2, kdm)):
        temp[kk_] = tofsig(sigmr0[2] - 0.4 - dsg, saln[kk_], pref)
    sigma = sig(temp, saln, pref)
    # a passive tracer with structure (ideal-age-like, trc/mod_tracers.F90:96-102)
    trc = (1.0 + 0.5 * np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y))[None] * (1.0 + 0.1 * k[:, None, None])

    def pad3(a):
        out = np.zeros((a.shape[0], nj, ni))
        out[:, NBDY:NBDY + jdm, NBDY:NBDY + idm] = a
        return out

    trcs = pad3(trc)[None].repeat(max(ntr, 1), axis=0)[:ntr]
    if tke:
        # TKE [m2 s-2] decaying from the surface, in places below tke_min = 7.6e-8 (phy/mod_tke.F90:61) so that
        # diapfl's lower bound acts; the generic-length-scale slot is carried as a plain tracer (no -DGLS)
        tk = 2.0e-5 * np.exp(-1.2 * (k[:, None, None] - 1)) * (1.0 + 0.8 * np.sin(2 * np.pi * (x + y)))[None]
        trcs[0] = pad3(tk)
        # in the _tk2 build it is the length-scale variable, in places below gls_psi_min = 1e-14 (phy/mod_tke.F90:62)
        trcs[1] = pad3((1.0e-15 if tk2 else 1.0e-9) * (1.0 + 0.5 * np.cos(2 * np.pi * x) * np.sin(4 * np.pi * y))[None] * (1.0 + 0.2 * k[:, None, None]))
        trcs[2] = pad3(trc)
        # further tracers (ntr > 3): plain passive tracers, as the biogeochemical ones are to the dynamical core
        # (trc/mod_tracers.F90:116-126: ntr = ... + ntrbgc), each with its own structure
        for nt in range(3, ntr):
            trcs[nt] = pad3((1.0 + 0.3 * np.sin(2 * np.pi * (x * (1 + nt % 3) + 0.1 * nt)) * np.cos(2 * np.pi * y * (1 + nt % 2)))[None]
                            * (1.0 + 0.05 * nt + 0.1 * k[:, None, None]))
    ic = dict(dp=pad3(dp), temp=pad3(temp), saln=pad3(saln), sigma=pad3(sigma),
              sigmar=pad3(sigmar), trc=trcs)
    return Case(name=full_name, idm=idm, jdm=jdm, kdm=kdm, nreg=nreg, params=p, depth=depth,
                grid=g, ic=ic, ntr=ntr)
