// difest_isobml's diffusivity estimates -- phy/mod_difest.F90:735-809 (difest_isobml), :353-586 (difest_common_iso), :2629-3084
// (difest_vertical_iso), :2040-2627 (difest_lateral_iso): what makes difint, difiso, difdia, difwgt -- the coefficients of eddtra,
// diffus, diapfl and momtum -- follow the state every step.
//
// Every routine walks columns: the Richardson number needs running values down a column (tup), the Eady growth rate and the
// vertical averages are sums over a column's layers in the reference's order, the TKE closure is pointwise in k but sits inside
// those loops.  One thread per column (u-, v- or p-point), the planes of 64 neighbouring columns contiguous; the reference's
// per-row 1-D temporaries (cr, bcrrd, dps, egrs ..) are registers, its (i,k) temporaries (du2, dv2, bvfsq, bvf, egr, anisok) planes of
// the work space.
//   k_dfi_kmax_kfil   kmax, kfil of the columns 0..ii+1 x 0..jj+1                                   (:366-390; kfil's halo :392-411)
//   k_dfi_uv2         squared vertical velocity differences at u- and v-columns, msku, mskv         (:413-511)
//   k_dfi_common      drhol, du2l, rig                                                              (:513-560)
//   k_dfi_vert_a/b/c  Brunt-Vaisala frequency, the one-equation TKE closure's source step (or the Richardson number
//                     parameterisation), background / tidal / weak-stability / near-inertial mixing -> difdia, trc(tke, gls);
//                     the per-level part with one thread per point AND level
//   k_dfi_lateral     Rossby radius, difwgt, Eady growth rate (shear or large scale), Eden-Greatbatch diffusivities with the
//                     suppression options -> difint, difiso; then the halo updates (and the optional smoothing) of :2577-2614
// Real powers and exponentials are the host libm's bits (pow_libm.h, exp_libm.h); tanh of the latitude (tidal mixing length
// scale, :2926-2927) and log of the Coriolis parameter (bdmldp, :2747-2750) depend on the grid only: the host evaluates them once
// with its own libm into the planes tdmls and bdmlq.  rhsctp's sin and atan2 of the flow direction are
// evaluated on the device with the bits of the host libm's (sin_libm.h, atan2_libm.h; round 6).  The two-equation closure (use_GLS:
// the length-scale variable a prognostic tracer, :2788-2814, :2858-2863, :2921-2927, :2970-2973) with the option gls = 1 (round 6).
// Parity: cross-checked against the reference's REAL mod_difest.F90 compiled against interface-only stand-ins for the CVMix
// modules it imports but does not call on this path (oracle/xcheck/cvmix_standin.F90; builds *_xdf) -- a cross-check, not a pin.
// Roofline: HBM, ~60 F of column traffic; the kernels are bound by their k-serial chains like the other column kernels.
#include "blomgpu_internal.h"
#include "../../include/blomgpu.h"
#include "eos.h"
#include "pow_libm.h"
#include "sin_libm.h"
#include "atan2_libm.h"
#include <cmath>

#define PLANE_IJ(V)                                                        \
  unsigned bx_, by_;                                                       \
  xcd_block(bx_, by_);                                                     \
  const int t_ = bx_ * blockDim.x + threadIdx.x;                           \
  (void)by_;                                                               \
  if (t_ >= (V).nplane) return;                                            \
  const int i = t_ % (V).ni - (NBDY - 1), j = t_ / (V).ni - (NBDY - 1);    \
  const size_t c = t_;                                                     \
  (void)i; (void)j; (void)c

#define GRAV 9.806
#define ALPHA0 1.e-3
#define PI_BLOM 3.1415926536      // phy/mod_constants.F90:40
#define BU 8                      // levels whose loads the column loops keep in flight
#define EPSILP 1.e-12
#define ONEM 9806.
#define DPBMIN 98.06              // onecm, :174
#define DPTMIN 9806.              // `integer, parameter :: dptmin = onem`, :173
#define DRHOMN 6.e-3
#define NU0 1.e-5
#define NUS0 5.e-3
#define NUG0 2.5e-1
#define DRHO0 6.e-3
#define NULS0 5.e-2
#define DMXEFF .2
#define TDMQ (1. / 3.)
#define TKEPLS (20. * ONEM)
#define NIWLS (300. * ONEM)
#define CORI30 7.2722e-5
#define DPGC (300. * ONEM)
#define DPGRAV (100. * ONEM)
#define DPDIAV (100. * ONEM)
#define DPDDAV (10. * ONEM)
#define DPNBAV (250. * ONEM)
#define USTMIN .001
#define KAPPA .4
#define BFEPS 1.e-16
#define SLEPS .1
#define ZETAS (-1.)
#define CPSEMIN (-0.2)
#define URMSEMIN 0.05
#define AS_ (-28.86)
#define CS_ 98.96
// mod_tke's parameters, phy/mod_tke.F90:36-63
#define GLS_CMU0 .527
#define PR_T 1.
#define GLS_C1 1.44
#define GLS_C2 1.92
#define GLS_C3PLUS 1.
#define GLS_C3MINUS (-.63)
#define GLS_GH0 .0329
#define GLS_GHMIN (-.28)
#define GLS_GHCRI .03
#define LS_UNLMT_MIN 1.e-8

// work-space slots (fields of kk levels)
enum { W_DU2 = 0, W_DV2, W_BVFSQ, W_BVF, W_EGR, W_ANISOK, W_SM1, W_SM2, W_NUB, W_NSLOT };

struct TkeC {            // initke's derived constants, phy/mod_tke.F90:133-160
  double sqrt2, cmu_fac1, cmu_fac2, cmu_fac3, tke_exp1, gls_exp1, gls_fac6, s0, s1, s2, s4, s5, s6, b0, b1, b2, b3, b4, b5, cmu0p3;
  double gls_qfac, gls_bbc, gls_cmu0p;     // the two-equation closure's constant factors (use_GLS), see tke_consts
};
struct DfePar {
  double egc, eggam, eglsmn, egmndf, egmxdf, egidfq, rhiscf, ri0, tkepf, niwgf, niwbf, niwlf, bdml_logc;
  int eddf2d, edsprs, edanis, redi3d, edritp, edwmth, use_tke, use_gls, itke, igls, rhsctp;
  TkeC T;
};

static TkeC tke_consts() {
  // the reference's statements evaluated as its compiler does: real powers through the libm's pow (the operands are PARAMETERs, so
  // flang folds them at compile time -- with the pow of the machine it runs on), integer powers by repeated squaring
  const double L1 = .107, L2 = .0032, L3 = .0864, L4 = .12, L5 = 11.9, L6 = .4, L7 = .0, L8 = .48;
  const double gls_p = 3., gls_m = 1.5, gls_n = -1.;
  TkeC t;
  t.sqrt2 = std::sqrt(2.);
  t.cmu_fac1 = std::pow(GLS_CMU0, -gls_p / gls_n);
  t.cmu_fac2 = std::pow(GLS_CMU0, 3. + gls_p / gls_n);
  t.cmu_fac3 = t.sqrt2;
  t.tke_exp1 = gls_m / gls_n;
  t.gls_exp1 = 1. / gls_n;
  const double c2 = GLS_CMU0 * GLS_CMU0;
  t.gls_fac6 = 8. / (c2 * (c2 * c2));
  t.cmu0p3 = GLS_CMU0 * c2;
  // use_GLS: .56**(.5*gls_n)*gls_cmu0**gls_p (:2806: two folded real powers and their product), (gls_cmu0**(gls_p-2.*gls_m)) ... *(kappa)**gls_n
  // (:2859-2862: the first factor is cmu0**0. = 1, the last .4**(-1.)), gls_cmu0**gls_p (:2926)
  t.gls_qfac = std::pow(.56, .5 * gls_n) * std::pow(GLS_CMU0, gls_p);
  t.gls_bbc = std::pow(.4, gls_n);
  t.gls_cmu0p = std::pow(GLS_CMU0, gls_p);
  t.s0 = 1.5 * L1 * (L5 * L5);
  t.s1 = -L4 * (L6 + L7) + 2. * L4 * L5 * (L1 - 1. / 3. * L2 - L3) + 1.5 * L1 * L5 * L8;
  t.s2 = -3. / 8. * L1 * (L6 * L6 - L7 * L7);
  t.s4 = 2. * L5;
  t.s5 = 2. * L4;
  t.s6 = 2. / 3. * L5 * (3. * (L3 * L3) - L2 * L2) - .5 * L5 * L1 * (3. * L3 - L2) + .75 * L1 * (L6 - L7);
  t.b0 = 3. * (L5 * L5);
  t.b1 = L5 * (7. * L4 + 3. * L8);
  t.b2 = (L5 * L5) * (3. * (L3 * L3) - L2 * L2) - .75 * (L6 * L6 - L7 * L7);
  t.b3 = L4 * (4. * L4 + 3. * L8);
  t.b4 = L4 * (L2 * L6 - 3. * L3 * L7 - L5 * (L2 * L2 - L3 * L3)) + L5 * L8 * (3. * (L3 * L3) - L2 * L2);
  t.b5 = .25 * (L2 * L2 - 3. * (L3 * L3)) * (L6 * L6 - L7 * L7);
  return t;
}

// ---- :366-390 ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_dfi_kmax_kfil(const DevView *__restrict__ Vp, int n, int nn) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < 0 || j > V.jj + 1 || i < 0 || i > V.ii + 1) return;
  gi_t kmax = V.m[I_dfe_kmax], kfil = V.m[I_dfe_kfil];
  if (!V.m[I_ip][c]) { kmax[c] = 0; return; }
  const size_t np = V.nplane;
  const int kk = V.kk;
  gcd_t dp = V.f[F_dp] + c + (size_t)nn * np;
  int km = 1;
  for (int k0 = 3; k0 <= kk; k0 += COLUMN_U) {
    double a[COLUMN_U];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) a[u] = dp[(size_t)((k0 + u <= kk ? k0 + u : kk) - 1) * np];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++)
      if (k0 + u <= kk && a[u] > DPBMIN) km = k0 + u;
  }
  kmax[c] = km;
  const int kf = V.m[I_kfpla][c + (size_t)(n - 1) * np];
  int r;
  if (kf >= km) r = kf + 1;
  else {
    gcd_t sr = V.f[F_sigmar] + c;
    if (V.f[F_sigma][c + (size_t)(kf - 1 + nn) * np] < .5 * (sr[(size_t)(kf - 1) * np] + sr[(size_t)kf * np])) r = kf + 1;
    else r = kf + 2;
  }
  kfil[c] = r;
}

// :392-411: kfil of the interior through util1 and its halo update back into kfil on 0..ii+1
__global__ void k_dfi_kfil_util(const DevView *__restrict__ Vp, int back) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (!V.m[I_ip][c]) return;
  gi_t kfil = V.m[I_dfe_kfil];
  if (!back) {
    if (j >= 1 && j <= V.jj && i >= 1 && i <= V.ii) V.f[F_util1][c] = (double)kfil[c];
  } else if (j >= 0 && j <= V.jj + 1 && i >= 0 && i <= V.ii + 1)
    kfil[c] = (int)lround(V.f[F_util1][c]);
}

// ---- :413-511: blockIdx.y = 0 the u-columns (j = 1..jj, i = 1..ii+1), 1 the v-columns (j = 1..jj+1, i = 1..ii) -------------
__global__ __launch_bounds__(64) void k_dfi_uv2(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  const bool isv = by_ == 1;
  if (isv ? (j < 1 || j > V.jj + 1 || i < 1 || i > V.ii) : (j < 1 || j > V.jj || i < 1 || i > V.ii + 1)) return;
  const size_t np = V.nplane;
  const int kk = V.kk;
  gd_t d2 = WK(V, isv ? W_DV2 : W_DU2) + c;
  gi_t msk = V.m[isv ? I_mskv : I_msku] + c;
  if (!V.m[isv ? I_iv : I_iu][c]) {
    for (int k = 0; k < kk; k++) { d2[(size_t)k * np] = 0.; msk[(size_t)k * np] = 0; }
    return;
  }
  gcd_t dpz = V.f[isv ? F_dpv : F_dpu] + c + (size_t)nn * np, vel = V.f[isv ? F_v : F_u] + c + (size_t)nn * np;
  gci_t kfil = V.m[I_dfe_kfil];
  const int ka = kfil[isv ? c - V.ni : c - 1], kb = kfil[c];
  const int kf = isv ? (ka > kb ? ka : kb) : (ka < kb ? ka : kb);       // max for v (:436), min for u (:482)
  int klpl = 1, kfpl = kk + 1;
  for (int k0 = 3; k0 <= kk; k0 += COLUMN_U) {
    double a[COLUMN_U];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) a[u] = dpz[(size_t)((k0 + u <= kk ? k0 + u : kk) - 1) * np];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++)
      if (k0 + u <= kk) {
        const int k = k0 + u;
        if (a[u] > DPBMIN) klpl = k;
        if (k >= 4 && k >= kf && a[u] > DPTMIN && kfpl == kk + 1) kfpl = k;      // the smallest such k (the reference walks k = kk..4)
      }
  }
  double tup = 0., vk = vel[0];
  for (int k0 = 1; k0 <= kk; k0 += BU) {
    double a[BU];
#pragma unroll
    for (int uu = 0; uu < BU; uu++) a[uu] = vel[(size_t)(k0 + uu < kk ? k0 + uu : kk - 1) * np];
#pragma unroll
    for (int uu = 0; uu < BU; uu++) {
      const int k = k0 + uu;
      if (k > kk) break;
      const double vn = k < kk ? a[uu] : 0.;
      double d = 0.;
      int m = 0;
      if (k >= kfpl && k <= klpl && klpl - kfpl >= 1) {
        if (k == kfpl) {
          double q = vn - vk;
          q = q * q;
          d = q;
          tup = q;
        } else if (k < klpl) {
          double q = vn - vk;
          q = q * q;
          d = .5 * (tup + q);
          tup = q;
        } else
          d = tup;
        m = 1;
      }
      d2[(size_t)(k - 1) * np] = d;
      msk[(size_t)(k - 1) * np] = m;
      vk = vn;
    }
  }
}

// ---- :513-560 ------------------------------------------------------------------------------------------------------------
// One thread per point AND level: the routine's one recurrence in k is the harmonic mean of a level's density jump q(k) with that
// of the level above, and q(k) is a function of the levels k and k+1 alone -- a thread evaluates both jumps it needs (the upper one
// a second time, bit for bit what the thread above evaluates) instead of waiting for it.  (As a column kernel: 97 us for 192 MB,
// a third of what its bytes allow.)
__global__ void k_dfi_common(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const int kf = V.m[I_dfe_kfil][c], km = V.m[I_dfe_kmax][c];
  if (km - kf < 1) return;
  const int kk = V.kk, ni = V.ni;
  const int k = by_ + 1;
  const int k1 = kf > 4 ? kf : 4, k2 = km < kk ? km : kk;
  if (k < k1 || k > k2) return;
  const size_t np = V.nplane, o = (size_t)(k - 1) * np + c;
  gcd_t p = V.f[F_p], temp = V.f[F_temp] + (size_t)nn * np, saln = V.f[F_saln] + (size_t)nn * np;
  // the density jump across the interface under level kq (< km), :520-531: both densities at the interface's pressure
  auto jump = [&](int kq) {
    const size_t oq = (size_t)(kq - 1) * np + c, o1 = (size_t)(kq < kk ? kq : kk - 1) * np + c;
    const double pk1 = p[oq + np];
    return fmax2(0., eos::rho(pk1, temp[o1], saln[o1]) - eos::rho(pk1, temp[oq], saln[oq]));
  };
  const double tup = k - 1 >= k1 ? jump(k - 1) : 0.;          // (the sweep starts at level max(kfil, 4) with zero)
  double dr;
  if (k < km) {
    const double q = jump(k);
    if (k == kf) dr = q;
    else dr = 2. * tup * q / fmax2(1.e-11, tup + q);
  } else
    dr = tup;
  V.f[F_drhol][o] = dr;
  const int mu0 = V.m[I_msku][o], mu1 = V.m[I_msku][o + 1], mv0 = V.m[I_mskv][o], mv1 = V.m[I_mskv][o + ni];
  gcd_t du2 = WK(V, W_DU2), dv2 = WK(V, W_DV2);
  const double d2 = ((double)mu0 * du2[o] + (double)mu1 * du2[o + 1]) / (double)(mu0 + mu1 > 1 ? mu0 + mu1 : 1) +
                    ((double)mv0 * dv2[o] + (double)mv1 * dv2[o + ni]) / (double)(mv0 + mv1 > 1 ? mv0 + mv1 : 1);
  V.f[F_du2l][o] = d2;
  V.f[F_rig][o] = ALPHA0 * ALPHA0 * fmax2(DRHOMN, dr) * V.f[F_dp][o + (size_t)nn * np] / fmax2(1.e-13, d2);
}

// ---- difest_vertical_iso, :2629-3084 ----------------------------------------------------------------------------------------
// Three kernels.  What the routine does per level inside the range kfil..kmax -- background / shear or TKE-closure / tidal / weak-stability
// mixing -- reads nothing of the other levels but two column quantities (the bottom-weighted buoyancy frequency bvfbot and
// exp(p_bottom / q)), so it runs with one thread per point AND level (k_dfi_vert_b: 5.6 M threads on the channel instead of 106 k
// k-serial ones; the first, all-in-one column kernel took 0.68 ms of a 7 ms step).  The column sums and everything that is a
// recurrence in k stay column kernels in the reference's order: k_dfi_vert_a in front (bvfsq, bvf, the closure's Buoy / Shear2 / Prod
// from the old difdia, bvfbot, and exp(p(k)/q) of every interface once -- a level needs its own and the next one's), k_dfi_vert_c
// behind (values copied down through the levels outside the range, the vertical averages dfddsu / dfddsl, the fill above kfil, the
// near-inertial term, the surface interface).
#define S2_BVFBOT 10
#define S2_EXPB 11
__global__ __launch_bounds__(64) void k_dfi_vert_a(const DevView *__restrict__ Vp, DfePar D, int nn) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const int kf = V.m[I_dfe_kfil][c], km = V.m[I_dfe_kmax][c];
  const size_t np = V.nplane;
  const int kk = V.kk;
  gcd_t p = V.f[F_p] + c, dp = V.f[F_dp] + c + (size_t)nn * np;
  gcd_t drhol = V.f[F_drhol] + c, du2l = V.f[F_du2l] + c, difdia = V.f[F_difdia] + c;
  gd_t bvfsq = WK(V, W_BVFSQ) + c, bvf = WK(V, W_BVF) + c, ex = V.f[F_wkp1] + c;
  gd_t Buoy = V.f[F_Buoy] + c, Shear2 = V.f[F_Shear2] + c, Prod = V.f[F_Prod] + c;
  const double pbot = p[(size_t)kk * np], q = V.f[F_tdmls][c];
  double bvfbot = 0., dps = 0.;
  if (km - kf >= 1) {
    const int k1 = kf > 4 ? kf : 4;
    double pk = p[(size_t)(k1 - 1) * np];
    ex[(size_t)(k1 - 1) * np] = exp_libm(pk / q);
    const int k2 = km < kk ? km : kk;
    for (int k0 = k1; k0 <= k2; k0 += BU) {                                        // :2653-2701
      double a0[BU], a1[BU], a2[BU], a3[BU], a4[BU];
#pragma unroll
      for (int uu = 0; uu < BU; uu++) {
        const size_t o = (size_t)((k0 + uu <= k2 ? k0 + uu : k2) - 1) * np;
        a0[uu] = dp[o]; a1[uu] = p[o + np]; a2[uu] = drhol[o]; a3[uu] = D.use_tke ? du2l[o] : 0.; a4[uu] = D.use_tke ? difdia[o] : 0.;
      }
#pragma unroll
      for (int uu = 0; uu < BU; uu++) {
        const int k = k0 + uu;
        if (k > k2) break;
        const size_t o = (size_t)(k - 1) * np;
        const double dpk = a0[uu], pk1 = a1[uu];
        ex[o + np] = exp_libm(pk1 / q);
        const double b2 = GRAV * GRAV * fmax2(DRHOMN, a2[uu]) / fmax2(EPSILP, dpk);
        const double b = sqrt(b2);
        bvfsq[o] = b2;
        bvf[o] = b;
        if (D.use_tke) {
          if (dpk > DPBMIN) {
            const double dd = a4[uu];
            Buoy[o] = -dd * b2;
            const double h = fmax2(ONEM, dpk) * ALPHA0 / GRAV;
            const double s2 = fmax2(1.e-13, a3[uu]) / (h * h);
            Shear2[o] = s2;
            Prod[o] = dd * PR_T * s2;
          } else {
            Buoy[o] = 0.;
            Shear2[o] = 1.e-9;
            Prod[o] = 0.;
          }
        }
        const double w = fmax2(0., pk1 - fmax2(pbot - DPNBAV, pk));
        if (w > 0.) {
          bvfbot = bvfbot + b * w;
          dps = dps + w;
        }
        pk = pk1;
      }
    }
  }
  if (dps > 0.) bvfbot = bvfbot / dps;
  WK2(V, S2_BVFBOT)[c] = bvfbot;
  WK2(V, S2_EXPB)[c] = exp_libm(pbot / q);
}

// one thread per point and level: the level's diffusivity, :2729-2949
__global__ void k_dfi_vert_b(const DevView *__restrict__ Vp, DfePar D, int nn) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const int k = by_ + 1;
  const int kf = V.m[I_dfe_kfil][c], km = V.m[I_dfe_kmax][c];
  if (!(km - kf >= 1 && k >= kf && k <= km && k >= 2)) return;
  const size_t np = V.nplane, o = (size_t)(k - 1) * np + c;
  const int kk = V.kk;
  const Params &P = V.P;
  const TkeC &T = D.T;
  gcd_t p = V.f[F_p];
  const double pk = p[o], pk1 = p[o + np], pbot = p[c + (size_t)kk * np];
  const double dpk = V.f[F_dp][o + (size_t)nn * np], b2 = WK(V, W_BVFSQ)[o];
  double nub;
  if (P.bdmtyp == 1) nub = P.bdmc1 / WK(V, W_BVF)[o];
  else if (P.bdmtyp == 2) nub = P.bdmc2;
  else nub = 0.;
  if (P.iwdflg == 1) nub = nub * (1. + (P.iwdfac - 1.) * V.f[F_ficem][c]);
  if (P.bdmldp) {
    const double q = fmax2(1.e-9, fabs(V.f[F_coriop][c]));
    nub = nub * q / CORI30 * V.f[F_bdmlq][c] / D.bdml_logc;
  }
  nub = fmax2(P.nubmin, nub);
  double nus;
  if (!D.use_tke) {
    const double ri = V.f[F_rig][o];
    if (ri < D.ri0) {                                                             // :2756-2775
      double q = (pbot - pk + .5 * dpk) / fmin2(DPGC, .5 * pbot);
      q = fmax2(0., 1. - q * q);
      q = q * q * q;
      nus = q * NUG0 + (1. - q) * NUS0;
      q = ri / D.ri0;
      q = fmax2(0., 1. - q * q);
      nus = nus * q * q * q;
    } else
      nus = 0.;
  } else {                                                                        // the one-equation closure, :2776-2921
    gd_t tke = V.f[F_trc] + o + ((size_t)nn + (size_t)(D.itke - 1) * 2 * kk) * np;
    gd_t gls = V.f[F_trc] + o + ((size_t)nn + (size_t)(D.igls - 1) * 2 * kk) * np;
    const double delt1 = P.delt1;
    const double gls_c3 = b2 > 0. ? GLS_C3MINUS : GLS_C3PLUS;
    const double prod = V.f[F_Prod][o], buoy = V.f[F_Buoy][o];
    // one-equation closure: the length-scale slot is diagnosed from production and buoyancy (:2778-2782); two-equation closure
    // (use_GLS, gls_p = 3, gls_m = 1.5, gls_n = -1: k-epsilon): it is a prognostic tracer, stepped here (:2788-2814)
    double tk = *tke, gl = D.use_gls ? *gls : fmax2((GLS_C1 * prod + gls_c3 * buoy) / GLS_C2, GLS_PSI_MIN);
    // (the real powers trc**(1.5+gls_m/gls_n), trc**(-1./gls_n) have the exponents 0 and 1: the compiler folds them)
    double tke_epsilon = T.cmu_fac2 * gl;
    if (D.use_gls) {
      const double r = gl / tk;
      const double gprod = r * GLS_C1 * prod, gbuoy = r * gls_c3 * buoy, gdiss = r * GLS_C2 * tke_epsilon;
      const double gq = gdiss / gl;
      if (gprod + gbuoy >= 0.) gl = (gl + delt1 * (gprod + gbuoy)) / (1. + delt1 * gq);
      else gl = (gl + delt1 * gprod) / (1. + delt1 * (gq - (gbuoy / gl)));
      gl = fmax2(gl, GLS_PSI_MIN);
      // q = .56**(.5 gls_n) cmu0**gls_p * tke**(gls_m + .5 gls_n) * bvf**(-gls_n): the last two exponents are 1; gls_n < 0: the larger one
      gl = fmax2(gl, T.gls_qfac * tk * WK(V, W_BVF)[o]);
      tke_epsilon = T.cmu_fac2 * gl;
    }
    const double tke_q = tke_epsilon / tk;
    if (prod + buoy >= 0.) tk = (tk + delt1 * (prod + buoy)) / (1. + delt1 * tke_q);
    else {
      tk = (tk + delt1 * prod) / (1. + delt1 * (tke_q - (buoy / tk)));
      tk = fmax2(tk, TKE_MIN);
    }
    if (D.tkepf > 0.) {                                                            // :2840-2850
      double q;
      if (dpk < EPSILP) q = exp_libm(-pk / TKEPLS);
      else q = TKEPLS * (exp_libm(-pk / TKEPLS) - exp_libm(-pk1 / TKEPLS)) / dpk;
      const double ustar = V.f[F_ustar][c];
      tk = tk + 67.83 * D.tkepf * q * (ustar * ustar);
    }
    if (dpk < EPSILP) { tk = TKE_MIN; gl = GLS_PSI_MIN; }
    if (k == km) {                                                                 // bottom boundary condition, :2863-2872
      const double ust = fmax2(V.f[F_ustarb][c], USTMIN);
      const double r = ust / GLS_CMU0;
      tk = fmax2(TKE_MIN, r * r);
      if (D.use_gls) gl = fmax2(GLS_PSI_MIN, pow_libm(ust, 3.) * T.gls_bbc);          // cmu0**(p - 2m) ust**(2m) kappa**n, :2858-2863
    }
    *tke = tk;
    if (!D.use_gls) *gls = gl;
    // trc(tke)**(-tke_exp1) and, below, trc(tke)**(-gls_m/gls_n): the same base and the same exponent, 1.5 -- one evaluation
    const double tk15 = pow_libm(tk, -T.tke_exp1);
    const double ls_unlmt = fmax2(LS_UNLMT_MIN, T.cmu_fac1 * pow_libm(gl, T.gls_exp1) * tk15);
    double ls;
    if (b2 > 0.) ls = fmin2(ls_unlmt, (T.tke_exp1 == -1.5 ? tk15 : pow_libm(tk, 1.5)) * (1. / gl));
    else ls = ls_unlmt;
    double gh = fmin2(GLS_GH0, -(b2 * ls * ls) / (2. * tk));                        // Canuto-A stability functions, :2892-2908
    const double ghc = gh - GLS_GHCRI;
    gh = fmin2(gh, (gh - ghc * ghc) / (gh + GLS_GH0 - 2. * GLS_GHCRI));
    gh = fmax2(gh, GLS_GHMIN);
    gh = fmin2(gh, GLS_GH0);
    const double f6 = T.gls_fac6, f62 = f6 * f6;
    double gm = (T.b0 / f6 - T.b1 * gh + T.b3 * f6 * (gh * gh)) / (T.b2 - T.b4 * f6 * gh);
    gm = fmin2(gm, V.f[F_Shear2][o] * ls * ls / (2. * tk));
    const double cff = T.b0 - T.b1 * f6 * gh + T.b2 * f6 * gm + T.b3 * f62 * (gh * gh) - T.b4 * f62 * gh * gm + T.b5 * f62 * gm * gm;
    double sh = (T.s4 - T.s5 * f6 * gh + T.s6 * f6 * gm) / cff;
    sh = fmax2(sh, 0.);
    sh = sh * T.cmu_fac3 / T.cmu0p3;
    const double ql = T.sqrt2 * ls * sqrt(tk);
    nus = fmin2(sh * ql, 4.05 * NUG0);
    const double lsc = fmax2(ls, LS_UNLMT_MIN);
    V.f[F_L_scale][o] = lsc;
    // use_GLS: the length-scale variable recomputed from the limited length scale, cmu0**p tke**m L**n (:2921-2927)
    if (D.use_gls) *gls = fmax2(T.gls_cmu0p * (T.tke_exp1 == -1.5 ? tk15 : pow_libm(tk, 1.5)) * (1. / lsc), GLS_PSI_MIN);
  }
  double nut;                                                                      // tidally driven mixing, :2924-2937
  {
    const double q = V.f[F_tdmls][c];
    gcd_t ex = V.f[F_wkp1];
    const double eb = WK2(V, S2_EXPB)[c];
    double vsf;
    if (dpk < EPSILP) vsf = ex[o] / (q * (eb - 1.));
    else vsf = (ex[o + np] - ex[o]) / (dpk * (eb - 1.));
    nut = GRAV * TDMQ * DMXEFF * V.f[F_twedon][c] * WK2(V, S2_BVFBOT)[c] * vsf / b2;
  }
  double nuls;                                                                     // weak local stability, :2940-2946
  const double dr = V.f[F_drhol][o];
  if (dr < DRHO0) {
    double q = dr / DRHO0;
    q = fmax2(0., 1. - q * q);
    nuls = NULS0 * q * q * q;
  } else
    nuls = 0.;
  V.f[F_difdia][o] = nub + nus + nut + nuls;
  WK(V, W_NUB)[o] = nub;
}

__global__ __launch_bounds__(64) void k_dfi_vert_c(const DevView *__restrict__ Vp, DfePar D, int nn) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const int kf = V.m[I_dfe_kfil][c], km = V.m[I_dfe_kmax][c];
  const bool any = km - kf >= 1;
  const size_t np = V.nplane;
  const int kk = V.kk;
  gcd_t p = V.f[F_p] + c, dp = V.f[F_dp] + c + (size_t)nn * np;
  gd_t difdia = V.f[F_difdia] + c, Lsc = V.f[F_L_scale] + c;
  gcd_t bvfsq = WK(V, W_BVFSQ) + c, nubw = WK(V, W_NUB) + c;
  double *tke = nullptr, *gls = nullptr;
  if (D.use_tke) {
    tke = V.f[F_trc] + c + ((size_t)nn + (size_t)(D.itke - 1) * 2 * kk) * np;
    gls = V.f[F_trc] + c + ((size_t)nn + (size_t)(D.igls - 1) * 2 * kk) * np;
  }
#define PL(k) p[(size_t)((k) - 1) * np]
  const double pbot = PL(kk + 1);
  difdia[0] = NU0;                                                                 // :2715
  double dfddsu = 0., dfddsl = 0., dps = 0.;
  const double pkf = kf <= kk + 1 ? PL(kf < 1 ? 1 : kf) : 0.;
  {
    // levels above the range take the value of the level above them (:2958-2970): difdia = nu0, the TKE and the length scale of layer 1
    const int ka = any ? (kf > 2 ? kf : 2) : kk + 1, kb = any ? (km < kk ? km : kk) : kk;
    double ld = NU0, lt = D.use_tke ? tke[0] : 0., ll = D.use_tke ? Lsc[0] : 0., lg = D.use_gls ? gls[0] : 0.;
    for (int k = 2; k < ka; k++) {
      const size_t o = (size_t)(k - 1) * np;
      difdia[o] = ld;
      if (D.use_tke) { tke[o] = lt; Lsc[o] = ll; }
      if (D.use_gls) gls[o] = lg;                                                  // :2970-2973
    }
    if (any) {
      if (D.use_tke) {                        // what the first level inside the range does to the two mixed layer layers, :2857-2860
        tke[0] = TKE_MIN; tke[np] = TKE_MIN;
        gls[0] = GLS_PSI_MIN; gls[np] = GLS_PSI_MIN;
      }
      const double pq = pkf + DPDDAV;
      double pk = PL(ka);
      for (int k0 = ka; k0 <= kb; k0 += BU) {                                      // :2951-2956
        double a0[BU], a1[BU], a2[BU];
#pragma unroll
        for (int uu = 0; uu < BU; uu++) {
          const size_t o = (size_t)((k0 + uu <= kb ? k0 + uu : kb) - 1) * np;
          a0[uu] = p[o + np]; a1[uu] = nubw[o]; a2[uu] = difdia[o];
        }
#pragma unroll
        for (int uu = 0; uu < BU; uu++) {
          if (k0 + uu > kb) break;
          const double q = fmax2(0., fmin2(pq, a0[uu]) - pk);
          dps = dps + q;
          dfddsu = dfddsu + a1[uu] * q;
          dfddsl = dfddsl + a2[uu] * q;
          pk = a0[uu];
          ld = a2[uu];
        }
      }
      if (D.use_tke) { lt = tke[(size_t)(kb - 1) * np]; ll = Lsc[(size_t)(kb - 1) * np]; }
      if (D.use_gls) lg = gls[(size_t)(kb - 1) * np];
      for (int k = kb + 1; k <= kk; k++) {
        const size_t o = (size_t)(k - 1) * np;
        difdia[o] = ld;
        if (D.use_tke) { tke[o] = lt; Lsc[o] = ll; }
        if (D.use_gls) gls[o] = lg;
      }
    }
  }
  if (dps > 0.) { dfddsu = dfddsu / dps; dfddsl = dfddsl / dps; }
  else { dfddsu = NU0; dfddsl = NU0; }
  const double p3 = PL(3);
  {                                                                                // :2991-3007
    const int ke = kf - 1 < kk - 1 ? kf - 1 : kk - 1;         // the levels above the range
    const bool lin = kf <= kk && PL(kf < kk ? kf : kk) - p3 > EPSILP;
    if (ke >= 2) difdia[np] = dfddsu;
    if (!lin)
      for (int k = 3; k <= ke; k++) difdia[(size_t)(k - 1) * np] = dfddsu;
    else
      for (int k0 = 3; k0 <= ke; k0 += BU) {
        double a[BU + 1];
#pragma unroll
        for (int uu = 0; uu <= BU; uu++) a[uu] = PL(k0 + uu <= ke + 1 ? k0 + uu : ke + 1);
#pragma unroll
        for (int uu = 0; uu < BU; uu++) {
          if (k0 + uu > ke) break;
          const double q = .5 * (a[uu + 1] + a[uu]);
          difdia[(size_t)(k0 + uu - 1) * np] = ((q - p3) * dfddsl + (pkf - q) * dfddsu) / (pkf - p3);
        }
      }
  }
  // (niwgf = 0: the term is zero times a finite number, and a diffusivity plus zero is the diffusivity -- nothing to do)
  if (any && D.niwgf != 0.) {                                                      // near-inertial waves, :3011-3032
    const double idk = V.f[F_idkedt][c];
    const double q = NIWLS;
    const double den = 1. - exp_libm((p3 - pbot) / q);
    double e_lo = exp_libm((p3 - PL(3)) / q);                 // exp((p3 - p(k+1))/q) of k = 2; every level's lower value is the next one's upper
    const int kb = km < kk - 1 ? km : kk - 1;
    for (int k0 = 2; k0 <= kb; k0 += BU) {
      double a0[BU], a1[BU], a2[BU], a3[BU];
#pragma unroll
      for (int uu = 0; uu < BU; uu++) {
        const int kq = k0 + uu <= kb ? k0 + uu : kb;
        const size_t o = (size_t)(kq - 1) * np;
        a0[uu] = dp[o]; a1[uu] = p[o + np]; a2[uu] = bvfsq[(size_t)((kq > kf ? kq : kf) - 1) * np]; a3[uu] = difdia[o];
      }
#pragma unroll
      for (int uu = 0; uu < BU; uu++) {
        const int k = k0 + uu;
        if (k > kb) break;
        const double dpk = a0[uu];
        const double e_up = k == 2 ? 0. : e_lo;                                    // exp((p3 - p(k))/q)
        if (k > 2) e_lo = exp_libm((p3 - a1[uu]) / q);
        double vsf;
        if (k == 2 || dpk < EPSILP) vsf = e_lo / (q * den);
        else vsf = (e_up - e_lo) / (dpk * den);
        const double nusm = GRAV * D.niwgf * (1. - D.niwbf) * D.niwlf * DMXEFF * idk * vsf / (ALPHA0 * a2[uu]);
        difdia[(size_t)(k - 1) * np] = a3[uu] + nusm;
      }
    }
  }
  {                                                                                // the lower interface of the top layer, :3035-3066
    const double ust = fmax2(USTMIN, V.f[F_ustar][c]);
    const double bf = V.f[F_buoyfl][c];
    const double mols = ust * ust * ust / (KAPPA * copysign(fmax2(fabs(bf), BFEPS), -bf));
    const double p1 = PL(1);
    const double h = (p3 - p1) / ONEM;
    const double sg = (PL(2) - p1) / (p3 - p1);
    double phis;
    if (mols < 0.) {
      const double zeta = fmin2(SLEPS, sg) * h / mols;
      if (zeta > ZETAS) phis = pow_libm(1. - 16. * zeta, -1. / 2.);
      else phis = pow_libm(AS_ - CS_ * zeta, -(1. / 3.));
    } else {
      const double zeta = sg * h / mols;
      phis = 1. + 5. * zeta;
    }
    const double ws = KAPPA * ust / phis;
    difdia[0] = h * ws * sg * ((1. - sg) * (1. - sg));
  }
#undef PL
}

// falign of :2331 -- 1 / max(sin(atan2(y, x) - hangle)**10, 1e-10) -- as a function of its own: inlined four times into the unrolled
// level loop, sin and atan2 took k_dfi_lateral from 219 to 260 VGPRs with spills and one wave per SIMD (1 755 waves: two rounds)
__device__ __noinline__ double dfi_falign(double y, double x, double hangle) {
  const double sa_ = sin_libm(atan2_libm(y, x) - hangle);
  const double s2 = sa_ * sa_, s4 = s2 * s2, s8 = s4 * s4;      // x**10 as the reference's compiler expands it: x2 = x x, x4 = x2 x2, x8 = x4 x4, x8 x2
  return 1. / fmax2(s8 * s2, 1.e-10);
}

// falign of every point and level, one thread each (rhsctp): the baroclinic velocities at the p-point (:2311-2316), the barotropic ones
// (:2286-2295), sin / atan2.  In the column kernel the call cost 97 us a launch (117 -> 215 us: ~300 dependent instructions per level and
// column); here the same arithmetic runs on 93 000 wavefronts -- the levels of a column's range only: evaluated at every level the kernel took
// 115 us --, and the column kernel reads one more plane (work slot W_EGR).
__global__ void k_dfi_falign(const DevView *__restrict__ Vp, int n, int nn) {
  const DevView &V = *Vp;
  unsigned bx_, by_;
  xcd_block(bx_, by_);
  const int t_ = bx_ * blockDim.x + threadIdx.x;
  if (t_ >= V.nplane) return;
  const int i = t_ % V.ni - (NBDY - 1), j = t_ / V.ni - (NBDY - 1);
  const size_t c = t_;
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const int k = by_ + 1, ni = V.ni;
  // only the levels the column kernel evaluates a diffusivity at read their value: kfil .. kmax of a column with kmax - kfil >= 1 (:2303-2305)
  const int kf = V.m[I_dfe_kfil][c], km = V.m[I_dfe_kmax][c];
  if (km - kf < 1 || k < (kf > 2 ? kf : 2) || k > km) return;
  const size_t np = V.nplane, on = (size_t)(n - 1) * np, o = c + (size_t)(k - 1 + nn) * np;
  const Params &P = V.P;
  const double tsfac = P.dlt / P.delt1;
  gcd_t ubf = V.f[F_ubflxs_p] + c + on, vbf = V.f[F_vbflxs_p] + c + on, pbu = V.f[F_pbu] + c + on, pbv = V.f[F_pbv] + c + on;
  gcd_t scuyi = V.f[F_scuyi] + c, scvxi = V.f[F_scvxi] + c;
  const double ubt = (ubf[0] * scuyi[0] + ubf[1] * scuyi[1]) * tsfac / fmax2(EPSILP, pbu[0] + pbu[1]);
  const double vbt = (vbf[0] * scvxi[0] + vbf[ni] * scvxi[ni]) * tsfac / fmax2(EPSILP, pbv[0] + pbv[ni]);
  gcd_t u = V.f[F_u], v = V.f[F_v], dpu = V.f[F_dpu], dpv = V.f[F_dpv];
  const double ubc = (u[o] * dpu[o] + u[o + 1] * dpu[o + 1]) / fmax2(EPSILP, dpu[o] + dpu[o + 1]);
  const double vbc = (v[o] * dpv[o] + v[o + ni] * dpv[o + ni]) / fmax2(EPSILP, dpv[o] + dpv[o + ni]);
  WK(V, W_EGR)[c + (size_t)(k - 1) * np] = dfi_falign(vbc + vbt, ubc + ubt, V.f[F_hangle][c]);
}

// ---- difest_lateral_iso, :2040-2575 -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_dfi_lateral(const DevView *__restrict__ Vp, DfePar D, int n, int nn) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane, on = (size_t)(n - 1) * np;
  const int kk = V.kk, ni = V.ni;
  const Params &P = V.P;
  gci_t kmaxa = V.m[I_dfe_kmax];
  const int kf = V.m[I_dfe_kfil][c], km = kmaxa[c];
  const bool any = km - kf >= 1;
  gcd_t p = V.f[F_p] + c, temp = V.f[F_temp] + c + (size_t)nn * np, saln = V.f[F_saln] + c + (size_t)nn * np;
  gd_t difint = V.f[F_difint] + c, difiso = V.f[F_difiso] + c;
  gd_t anisok = WK(V, W_ANISOK) + c;
  gcd_t rig = V.f[F_rig] + c;
  gcd_t u = V.f[F_u] + c + (size_t)nn * np, v = V.f[F_v] + c + (size_t)nn * np;
  gcd_t dpu = V.f[F_dpu] + c + (size_t)nn * np, dpv = V.f[F_dpv] + c + (size_t)nn * np;
  gci_t msku = V.m[I_msku] + c, mskv = V.m[I_mskv] + c;
#define PL(k) p[(size_t)((k) - 1) * np]
  const double pbot = PL(kk + 1);
  // first baroclinic Rossby radius (WKB, Chelton et al. 1998), :2065-2107
  const int kfp = V.m[I_kfpla][c + on];
  double pup = .5 * (3. * PL(3) - PL((kk < kfp ? kk : kfp) + 1));
  double tup = temp[np], sup = saln[np], cr = 0.;
  {
    // (BU levels' loads issued before the first is used -- blomgpu_internal.h: COLUMN_U -- here and in every loop below: a level's
    // inputs do not depend on what the loop computes, the sums travel in registers in the reference's order)
    const int k3 = kfp > 3 ? kfp : 3;
    double pk = k3 <= kk ? PL(k3) : 0.;
    for (int k0 = k3; k0 <= kk; k0 += BU) {
      double a[BU], b[BU], d[BU];
#pragma unroll
      for (int uu = 0; uu < BU; uu++) {
        const size_t o = (size_t)((k0 + uu <= kk ? k0 + uu : kk) - 1) * np;
        a[uu] = temp[o]; b[uu] = saln[o]; d[uu] = p[o + np];
      }
#pragma unroll
      for (int uu = 0; uu < BU; uu++) {
        if (k0 + uu > kk) break;
        const double pk1 = d[uu];
        double plo;
        if (pbot - pk1 < EPSILP) plo = pbot;
        else plo = .5 * (pk + pk1);
        const double tlo = a[uu], slo = b[uu];
        cr = cr + sqrt(fmax2(0., (eos::rho(pk, tlo, slo) - eos::rho(pk, tup, sup)) * (plo - pup)));
        pup = plo;
        tup = tlo;
        sup = slo;
        pk = pk1;
      }
    }
  }
  const double coriop = V.f[F_coriop][c], betafp = V.f[F_betafp][c];
  cr = ALPHA0 * cr / PI_BLOM;
  const double bcrrd = sqrt(cr * cr / fmax2(coriop * coriop + 2. * betafp * cr, 1.e-24));
  const double afeql = fmax2(fabs(coriop), sqrt(2. * betafp * cr));
  double difwgt;
  {                                                                                // :2111-2131
    const double sx = V.f[F_scpx][c], sy = V.f[F_scpy][c];
    const double q = bcrrd / sqrt(.5 * (sx * sx + sy * sy));
    if (D.edwmth == 1) {
      difwgt = 1. / (1. + .25 * (q * q * q * q));        // q**4 as the reference's compiler expands it at run time: ((q q) q) q
    } else
      difwgt = q <= 2. ? 1. : 0.;
    V.f[F_difwgt][c] = difwgt;
  }
  // Eady growth rate, :2160-2257, and the layer interface diffusivities, :2283-2375, in ONE sweep
  // over the levels of the range: the reference's second loop reads nothing of the first but egr(k) of the same level (its vertical mean
  // egrs enters only after both), so a level's diffusivity follows its growth rate while the loads of the next levels are in flight, and
  // egr never goes to memory.  Each of the two sets of sums keeps its own order.  Outside the range a level takes the value of the one
  // above it: egmndf above the range, the last level's below it.
  const bool sa = D.edsprs || D.edanis;
  const bool vel = D.edanis;                                          // the baroclinic velocities at the p-point are needed for the anisotropy (:2308)
  // rhsctp (round 6): the topographic beta; falign of the level comes from k_dfi_falign's plane
  const double betatp = D.rhsctp ? V.f[F_betatp][c] : 0.;
  gcd_t fal = WK(V, W_EGR) + c;
  double egrs = 0., dps_e = 0.;
  const int ka = kf > 2 ? kf : 2, kb = km < kk ? km : kk;            // the levels inside the range (when there are any)
  const double pkf_g = any ? PL(kf) + DPGRAV : 0.;
  const double pkf_d = any ? PL(kf) + DPDIAV : 0.;
  const bool keep_anisok = D.edanis && !(D.eddf2d && !D.redi3d);     // read again by the last loop of the routine
  double dfints = 0., anisos = 0., dps = 0., last = D.egmndf;
  difint[0] = D.egmndf;
  // a level's diffusivity from its growth rate e; d = p(k+1), pk = p(k)
  auto diff_level = [&](int k, double e, double d, double pk, double u0, double u1, double du0, double du1, double v0, double v1, double dv0,
                        double dv1, double falign) {
    const size_t o = (size_t)(k - 1) * np;
    double rhisc = e / fmax2(1.e-22, betafp);
    double speed = 0.;
    if (vel) {
      const double ubc = (u0 * du0 + u1 * du1) / fmax2(EPSILP, du0 + du1);
      const double vbc = (v0 * dv0 + v1 * dv1) / fmax2(EPSILP, dv0 + dv1);
      speed = fmax2(1.e-22, sqrt(ubc * ubc + vbc * vbc));
    }
    if (D.rhsctp) {
      // topographic Rhines scale, masked where the flow is not along the topography, :2320-2337 (falign: k_dfi_falign)
      const double rhisct = e / fmax2(1.e-22, betatp);
      rhisc = fmin2(rhisc, falign * D.rhiscf * rhisct);
    }
    const double els = fmax2(D.eglsmn, fmin2(bcrrd, rhisc));
    const double di = D.egc * e * els * els;
    difint[o] = di;
    last = di;
    double q;
    if (D.eddf2d) q = fmax2(0., d - pk);
    else q = fmax2(0., fmin2(pkf_d, d) - pk);
    dps = dps + q;
    dfints = dfints + di * q;
    if (D.edanis) {
      const double r = speed / fmax2(1.e-22, e * els);
      const double a = 1. / (1. + r * r);
      if (keep_anisok) anisok[o] = a;
      anisos = anisos + a * q;
    }
  };
#define BL 4      /* levels in flight in the fused sweep: 13 loads a level */
  if (any) {
    for (int k = 2; k < ka; k++) difint[(size_t)(k - 1) * np] = D.egmndf;
    if (D.edritp == 1) {
      double pk = PL(ka);
      for (int k0 = ka; k0 <= kb; k0 += BL) {
        double a[BL], d[BL], u0[BL], u1[BL], du0[BL], du1[BL], v0[BL], v1[BL], dv0[BL], dv1[BL], fa[BL];
#pragma unroll
        for (int uu = 0; uu < BL; uu++) {
          const size_t o = (size_t)((k0 + uu <= kb ? k0 + uu : kb) - 1) * np;
          a[uu] = rig[o]; d[uu] = p[o + np];
          fa[uu] = D.rhsctp ? fal[o] : 0.;
          if (vel) {
            u0[uu] = u[o]; u1[uu] = u[o + 1]; du0[uu] = dpu[o]; du1[uu] = dpu[o + 1];
            v0[uu] = v[o]; v1[uu] = v[o + ni]; dv0[uu] = dpv[o]; dv1[uu] = dpv[o + ni];
          }
        }
#pragma unroll
        for (int uu = 0; uu < BL; uu++) {
          const int k = k0 + uu;
          if (k > kb) break;
          const double e = afeql / sqrt(a[uu] + D.eggam);
          if (sa) {
            double q;
            if (D.eddf2d) q = fmax2(0., d[uu] - pk);
            else q = fmax2(0., fmin2(pkf_g, d[uu]) - pk);
            dps_e = dps_e + q;
            egrs = egrs + e * q;
          }
          diff_level(k, e, d[uu], pk, u0[uu], u1[uu], du0[uu], du1[uu], v0[uu], v1[uu], dv0[uu], dv1[uu], fa[uu]);
          pk = d[uu];
        }
      }
    } else {
      gcd_t nx = V.f[F_nnslpx] + c, ny = V.f[F_nnslpy] + c;
      const int kmw = kmaxa[c - 1], kme = kmaxa[c + 1], kms = kmaxa[c - ni], kmn = kmaxa[c + ni];
      // the squared large scale slope x buoyancy frequency at interface kq from the four values around the point, :2190-2205 / :2212-2227
      auto slope2 = [&](int kq, double x0, double x1, double y0, double y1) {
        const bool w = kmw >= kq, e = kme >= kq, s_ = kms >= kq, n_ = kmn >= kq;
        double q;
        if (w && e) { const double t = x0 + x1; q = .25 * (t * t); }
        else if (w) q = x0 * x0;
        else if (e) q = x1 * x1;
        else q = 0.;
        if (s_ && n_) { const double t = y0 + y1; q = q + .25 * (t * t); }
        else if (s_) q = q + y0 * y0;
        else if (n_) q = q + y1 * y1;
        return q;
      };
      double egrup, egr_prev = 0.;
      {
        const size_t o = (size_t)(kf - 1) * np;
        egrup = sqrt(slope2(kf, nx[o], nx[o + 1], ny[o], ny[o + ni]));
      }
      double pk = PL(ka);
      for (int k0 = ka; k0 <= kb; k0 += BL) {
        double x0[BL], x1[BL], y0[BL], y1[BL], d[BL], u0[BL], u1[BL], du0[BL], du1[BL], v0[BL], v1[BL], dv0[BL], dv1[BL], fa[BL];
#pragma unroll
        for (int uu = 0; uu < BL; uu++) {
          const int kq = k0 + uu <= kb ? k0 + uu : kb;
          const size_t oi = (size_t)(kq < kk ? kq : kk - 1) * np;            // the interface below the level (not read for the last level)
          x0[uu] = nx[oi]; x1[uu] = nx[oi + 1]; y0[uu] = ny[oi]; y1[uu] = ny[oi + ni]; d[uu] = p[(size_t)kq * np];
          fa[uu] = D.rhsctp ? fal[(size_t)(kq - 1) * np] : 0.;
          if (vel) {
            const size_t o = (size_t)(kq - 1) * np;
            u0[uu] = u[o]; u1[uu] = u[o + 1]; du0[uu] = dpu[o]; du1[uu] = dpu[o + 1];
            v0[uu] = v[o]; v1[uu] = v[o + ni]; dv0[uu] = dpv[o]; dv1[uu] = dpv[o + ni];
          }
        }
#pragma unroll
        for (int uu = 0; uu < BL; uu++) {
          const int k = k0 + uu;
          if (k > kb) break;
          double e;
          if (k < km) {
            const double egrlo = sqrt(slope2(k + 1, x0[uu], x1[uu], y0[uu], y1[uu]));
            e = .5 * (egrup + egrlo);
            egr_prev = e;
            egrup = egrlo;
            if (sa) {
              double q;
              if (D.eddf2d) q = fmax2(0., d[uu] - pk);
              else q = fmax2(0., fmin2(pkf_g, d[uu]) - pk);
              dps_e = dps_e + q;
              egrs = egrs + e * q;
            }
          } else
            e = egr_prev;
          diff_level(k, e, d[uu], pk, u0[uu], u1[uu], du0[uu], du1[uu], v0[uu], v1[uu], dv0[uu], dv1[uu], fa[uu]);
          pk = d[uu];
        }
      }
    }
    for (int k = kb + 1; k <= kk; k++) difint[(size_t)(k - 1) * np] = last;
  } else
    for (int k = 2; k <= kk; k++) difint[(size_t)(k - 1) * np] = D.egmndf;
  if (sa) {
    if (dps_e > 0.) egrs = egrs / dps_e;
    else egrs = 0.;
  }
  // the surface non-isopycnic layers, :2383-2513
  double urmse = 0., cpse = 0., els_s = 0.;
  if (sa) {
    double rhisc = egrs / fmax2(1.e-22, betafp);
    if (D.rhsctp) rhisc = fmin2(rhisc, D.rhiscf * (egrs / fmax2(1.e-22, betatp)));       // :2393-2398
    els_s = fmax2(D.eglsmn, fmin2(bcrrd, rhisc));
    if (D.edsprs) {
      urmse = 2.86 * D.egc * egrs * els_s;
      cpse = fmax2(CPSEMIN, -betafp * (bcrrd * bcrrd));
    }
  }
  const double difmxp = V.f[F_difmxp][c];
  const double cosang = V.f[F_cosang][c], sinang = V.f[F_sinang][c];
  if (dps > 0.) {
    dfints = dfints / dps;
    double esfac;
    if (sa) {
      gci_t ip = V.m[I_ip];
      const size_t o2 = np;
      auto mlvel = [&](const double *w, const double *dw, size_t off) {   // thickness weighted velocity of the two mixed layer layers
        return (w[off] * dw[off] + w[off + o2] * dw[off + o2]) / (dw[off] + dw[off + o2]);
      };
      double ubc, vbc;
      if (ip[c - 1] + ip[c + 1] == 2) ubc = .5 * (mlvel(u, dpu, 0) + mlvel(u, dpu, 1));
      else if (ip[c - 1] == 1) ubc = mlvel(u, dpu, 0);
      else if (ip[c + 1] == 1) ubc = mlvel(u, dpu, 1);
      else ubc = 0.;
      if (ip[c - ni] + ip[c + ni] == 2) vbc = .5 * (mlvel(v, dpv, 0) + mlvel(v, dpv, ni));
      else if (ip[c - ni] == 1) vbc = mlvel(v, dpv, 0);
      else if (ip[c + ni] == 1) vbc = mlvel(v, dpv, ni);
      else vbc = 0.;
      if (D.edanis) {
        anisos = anisos / dps;
        const double speed = fmax2(1.e-22, sqrt(ubc * ubc + vbc * vbc));
        const double r = speed / fmax2(1.e-22, egrs * els_s);
        esfac = 1. / (1. + r * r);
      } else {
        double umnsc = ubc * cosang;
        umnsc = umnsc - vbc * sinang - cpse;
        const double r = umnsc / fmax2(URMSEMIN, fabs(urmse));
        esfac = 1. / (1. + 4. * (r * r));
      }
    } else
      esfac = 1.;
    if (D.eddf2d) {
      double d1;
      if (D.edanis) d1 = anisos * dfints * difwgt;
      else d1 = dfints * difwgt;
      if (D.redi3d) difiso[0] = fmin3(difmxp, D.egmxdf, fmax2(D.egmndf, esfac * dfints * D.egidfq * difwgt));
      else difiso[0] = fmin3(difmxp, D.egmxdf, fmax2(D.egmndf, d1 * D.egidfq));
      difint[0] = fmin3(difmxp, D.egmxdf, fmax2(D.egmndf, d1));
    } else {
      const double d1 = fmin3(difmxp, D.egmxdf, fmax2(D.egmndf, dfints * difwgt * esfac));
      difint[0] = d1;
      difiso[0] = d1 * D.egidfq;
    }
  } else {
    dfints = D.egmndf;
    difiso[0] = difint[0] * D.egidfq;
  }
  // the isopycnic layers, :2516-2572; outside the range again the value of the level above
  {
    double li = difint[0], ls_ = difiso[0];
    const double di1 = li, ds1 = ls_;
    const int k_end = any ? ka - 1 : kk;
    for (int k = 2; k <= k_end; k++) { difint[(size_t)(k - 1) * np] = li; difiso[(size_t)(k - 1) * np] = ls_; }
    if (any && D.eddf2d && !D.redi3d) {
      // (one value for the column: nothing of the levels is read)
      for (int k = ka; k <= kk; k++) { difint[(size_t)(k - 1) * np] = di1; difiso[(size_t)(k - 1) * np] = ds1; }
    } else if (any) {
      for (int k0 = ka; k0 <= kb; k0 += BU) {
        double dv[BU], av[BU], u0[BU], u1[BU], v0[BU], v1[BU];
        int mu0[BU], mu1[BU], mv0[BU], mv1[BU];
#pragma unroll
        for (int uu = 0; uu < BU; uu++) {
          const size_t o = (size_t)((k0 + uu <= kb ? k0 + uu : kb) - 1) * np;
          dv[uu] = difint[o];
          av[uu] = D.edanis ? anisok[o] : 1.;
          if (D.edsprs) {
            mu0[uu] = msku[o]; mu1[uu] = msku[o + 1]; mv0[uu] = mskv[o]; mv1[uu] = mskv[o + ni];
            u0[uu] = u[o]; u1[uu] = u[o + 1]; v0[uu] = v[o]; v1[uu] = v[o + ni];
          }
        }
#pragma unroll
        for (int uu = 0; uu < BU; uu++) {
          const int k = k0 + uu;
          if (k > kb) break;
          const size_t o = (size_t)(k - 1) * np;
          double esfac;
          if (D.edsprs) {
            const double umnsc = ((double)mu0[uu] * u0[uu] + (double)mu1[uu] * u1[uu]) / (double)(mu0[uu] + mu1[uu] > 1 ? mu0[uu] + mu1[uu] : 1) * cosang -
                                 ((double)mv0[uu] * v0[uu] + (double)mv1[uu] * v1[uu]) / (double)(mv0[uu] + mv1[uu] > 1 ? mv0[uu] + mv1[uu] : 1) * sinang - cpse;
            const double r = umnsc / fmax2(URMSEMIN, fabs(urmse));
            esfac = 1. / (1. + 4. * (r * r));
          } else if (D.edanis)
            esfac = av[uu];
          else
            esfac = 1.;
          if (D.eddf2d) {
            li = di1;
            if (D.redi3d) ls_ = fmin3(difmxp, D.egmxdf, fmax2(D.egmndf, esfac * dfints * D.egidfq * difwgt));
            else ls_ = ds1;
          } else {
            li = fmin3(difmxp, D.egmxdf, fmax2(D.egmndf, dv[uu] * difwgt * esfac));
            ls_ = li * D.egidfq;
          }
          difint[o] = li;
          difiso[o] = ls_;
        }
      }
      for (int k = kb + 1; k <= kk; k++) { difint[(size_t)(k - 1) * np] = li; difiso[(size_t)(k - 1) * np] = ls_; }
    }
  }
#undef PL
  (void)P;
}

// ---- :2582-2607: lateral smoothing of difint, difiso (edfsmo); the unsmoothed fields lie in two work-space slots -------------------
__global__ void k_dfi_smooth(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  const int mrg = 2;
  if (j < 1 - mrg || j > V.jj + mrg || i < 1 - mrg || i > V.ii + mrg || !V.m[I_ip][c]) return;
  const int k = by_, ni = V.ni;
  const size_t np = V.nplane, o = (size_t)k * np, okn = (size_t)(k + nn) * np;
  gci_t ip = V.m[I_ip];
  gcd_t dp = V.f[F_dp] + okn, u1 = WK(V, W_SM1) + o, u2 = WK(V, W_SM2) + o;
  const double ws = .125 * (double)ip[c - ni] * fmin2(ONEM, dp[c - ni]) / ONEM;
  const double ww = .125 * (double)ip[c - 1] * fmin2(ONEM, dp[c - 1]) / ONEM;
  const double we = .125 * (double)ip[c + 1] * fmin2(ONEM, dp[c + 1]) / ONEM;
  const double wn = .125 * (double)ip[c + ni] * fmin2(ONEM, dp[c + ni]) / ONEM;
  const double wc = -((ws + ww) + (we + wn)) + 1.;
  V.f[F_difint][c + o] = (ws * u1[c - ni] + ww * u1[c - 1]) + (we * u1[c + 1] + wn * u1[c + ni]) + wc * u1[c];
  V.f[F_difiso][c + o] = (ws * u2[c - ni] + ww * u2[c - 1]) + (we * u2[c + 1] + wn * u2[c + ni]) + wc * u2[c];
}

// the TKE closure's derived constants as this library evaluates them (tests compare them with the reference's)
extern "C" int blomgpu_tke_const(const char *name, double *v) {
  const TkeC t = tke_consts();
  const std::string s(name);
#define G(nm, f) if (s == #nm) { *v = t.f; return 0; }
  G(sqrt2, sqrt2) G(cmu_fac1, cmu_fac1) G(cmu_fac2, cmu_fac2) G(cmu_fac3, cmu_fac3) G(tke_exp1, tke_exp1) G(gls_exp1, gls_exp1)
  G(gls_fac6, gls_fac6) G(gls_s0, s0) G(gls_s1, s1) G(gls_s2, s2) G(gls_s4, s4) G(gls_s5, s5) G(gls_s6, s6)
  G(gls_b0, b0) G(gls_b1, b1) G(gls_b2, b2) G(gls_b3, b3) G(gls_b4, b4) G(gls_b5, b5)
#undef G
  return 1;
}

int st_difest_isobml(blomgpu_ctx *c, int m, int n, int mm, int nn) {
  const DevView &h = c->h;
  if (h.P.vcoord_tag != 1) return ctx_fail(c, "difest_isobml is the isopycnic coordinate's (phy/mod_blom_step.F90:140)");
  if (h.P.itrtke >= 1 && (h.P.itrtke > h.ntr || h.P.itrgls < 1 || h.P.itrgls > h.ntr)) return ctx_fail(c, "difest: itrtke / itrgls outside 1..ntr");
  if (h.P.bdmldp && c->bdml_logc == 0.) return ctx_fail(c, " difest_vertical_iso: bdmldp needs the plane bdmlq and the option bdml_logc (the host's log)");
  if (W_NSLOT > h.nwk) return ctx_fail(c, "difest: work space too small");
  if (int rc = st_difest_isobml_pre(c, m, n, mm, nn)) return rc;                  // :750-790
  DfePar D;
  D.egc = c->egc; D.eggam = c->eggam; D.eglsmn = c->eglsmn; D.egmndf = c->egmndf; D.egmxdf = c->egmxdf; D.egidfq = c->egidfq;
  D.rhiscf = c->rhiscf; D.ri0 = c->ri0; D.tkepf = c->tkepf; D.niwgf = c->niwgf; D.niwbf = c->niwbf; D.niwlf = c->niwlf;
  D.bdml_logc = c->bdml_logc;
  D.rhsctp = c->rhsctp; D.eddf2d = c->eddf2d; D.edsprs = c->edsprs; D.edanis = c->edanis; D.redi3d = c->redi3d; D.edritp = c->edritp_opt; D.edwmth = c->edwmth_opt;
  D.use_tke = h.P.itrtke >= 1; D.use_gls = D.use_tke && h.P.gls; D.itke = h.P.itrtke; D.igls = h.P.itrgls;
  D.T = tke_consts();
  const dim3 g1 = plane_grid(h, 1, 64), g2 = plane_grid(h, 2, 64), b64(64);
  {
    TimeScope ts(c, "difest");
    hipLaunchKernelGGL(k_dfi_kmax_kfil, g1, b64, 0, c->stream, c->d, n, nn);
    hipLaunchKernelGGL(k_dfi_kfil_util, plane_grid(h), dim3(256), 0, c->stream, c->d, 0);
  }
  if (int rc = st_xctilr(c, h.f[F_util1], 1, 1, 1, 1, 1)) return rc;
  {
    TimeScope ts(c, "difest");
    hipLaunchKernelGGL(k_dfi_kfil_util, plane_grid(h), dim3(256), 0, c->stream, c->d, 1);
    hipLaunchKernelGGL(k_dfi_uv2, g2, b64, 0, c->stream, c->d, nn);
    hipLaunchKernelGGL(k_dfi_common, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, nn);
    // difest_vertical_iso (vert_a/b/c: reads the common part's drhol, du2l, rig; writes difdia, the closure's tracers and fields, work
    // slots W_BVFSQ, W_BVF, W_NUB, wkp1, two 2-D slots) and difest_lateral_iso (falign, lateral: reads rig, msku/v, the state, cmnfld2's
    // nnslpx/y; writes difint, difiso, difwgt, W_EGR, W_ANISOK, W_SM1/2) share no array one of them writes: inside blomgpu_step (phys_dag)
    // the lateral part runs on the second stream -- behind cmnfld2's kernels if they are there -- beside the vertical chain
    const bool aside = ctx_overlap_on(c) && (c->phys_dag & 1) && !c->tiling.multi();
    hipStream_t sl = c->stream;
    if (aside) {
      if (int rc = ctx_side_fork(c, 5)) return rc;
      sl = c->side;
    }
    hipLaunchKernelGGL(k_dfi_vert_a, g1, b64, 0, c->stream, c->d, D, nn);
    hipLaunchKernelGGL(k_dfi_vert_b, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, D, nn);
    hipLaunchKernelGGL(k_dfi_vert_c, g1, b64, 0, c->stream, c->d, D, nn);
    if (D.rhsctp) hipLaunchKernelGGL(k_dfi_falign, plane_grid(h, h.kk), dim3(256), 0, sl, c->d, n, nn);
    hipLaunchKernelGGL(k_dfi_lateral, g1, b64, 0, sl, c->d, D, n, nn);
    if (aside) {
      c->cmn_on_side = false;
      if (int rc = ctx_side_done(c, 4)) return rc;
      if (int rc = ctx_side_join(c, 4)) return rc;
    }
  }
  HIPCHK(c, hipGetLastError());
  const int mrgint = 1, mrgiso = 2;
  if (c->edfsmo) {
    if (int rc = st_xctilr(c, h.f[F_difint], 1, h.kk, mrgint + 1, mrgint + 1, 1)) return rc;
    if (int rc = st_xctilr(c, h.f[F_difiso], 1, h.kk, mrgiso + 1, mrgiso + 1, 1)) return rc;
    const size_t bytes = sizeof(double) * (size_t)h.kk * h.nplane;
    HIPCHK(c, hipMemcpyAsync(WK(h, W_SM1), h.f[F_difint], bytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(WK(h, W_SM2), h.f[F_difiso], bytes, hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(k_dfi_smooth, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, nn);
    HIPCHK(c, hipGetLastError());
  } else {
    if (int rc = st_xctilr(c, h.f[F_difint], 1, h.kk, mrgint, mrgint, 1)) return rc;
    if (int rc = st_xctilr(c, h.f[F_difiso], 1, h.kk, mrgiso, mrgiso, 1)) return rc;
  }
  return 0;
}
