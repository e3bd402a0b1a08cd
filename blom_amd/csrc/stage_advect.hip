// advect -- flux areas + incremental remapping of layer thickness, T, S and tracers.
// phy/mod_advect.F90:59-189 (advmth='remap') and phy/mod_remap.F90:53-1522.
//
// The reference calls remap() once per layer on 2-D slices; layers are independent, so each
// kernel here covers all layers at once (blockIdx.y = layer) with one thread per point of the
// padded plane (unit-stride in i across the wavefront):
//   k_adv_flux_area  cau,cav = clamp(u*dt*scuy + ...)                        (mod_advect:71-94)
//   k_adv_pbmin      9-point wet-aware minimum of bottom pressure            (mod_advect:100-121)
//   k_remap_grad     limited gradients + centre-of-mass offsets at p-points   (mod_remap:358-584)
//   k_remap_flux     corner velocities, then the u-face and v-face flux polygon integrals
//                    (<=2 triangles + 1 pentagon each), accumulation into uflx.. (mod_remap:588-1462)
//   k_remap_update   flux-divergence update of dp,T,S,trc                     (mod_remap:1468-1520)
// "dp = max(0,dp)+dpeps; pup = plo-dp" (mod_remap:297-303) is applied on the fly wherever a
// cell is read (every donor/neighbour cell is wet and inside the reference's -2..+3 range), and
// committed for the outer ring by k_remap_update, which reproduces the reference's in-place
// side effect on halo cells.
// The four directional variants of each face in the Fortran differ only by the donor cell and
// the sign of the half-cell offset; they are folded into one body with sh = +/-0.5 (exact).
//
// Algorithmic bytes: (25 + 2*ntr) F.  This first version keeps the gradient and
// flux fields in HBM work space between the three remap kernels (extra ~(26+10*ntr) F);
// roofline: HBM.
#include "remap_common.h"

#define THREAD_IJ(V)                                                       \
  unsigned bx_, by_;                                                       \
  xcd_block(bx_, by_);                                                     \
  const int t_ = bx_ * blockDim.x + threadIdx.x;                           \
  (void)by_;                                                               \
  if (t_ >= (V).nplane) return;                                            \
  const int i = t_ % (V).ni - (NBDY - 1), j = t_ / (V).ni - (NBDY - 1);    \
  const size_t c = t_

// ---- mod_advect.F90:71-94 --------------------------------------------------------------------
__global__ void k_adv_flux_area(const DevView *__restrict__ Vp, int m, int mm, int nn) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const int k = by_;
  const size_t np = V.nplane, okm = c + (size_t)(k + mm) * np, okn = c + (size_t)(k + nn) * np;
  const size_t om = c + (size_t)(m - 1) * np;
  const double delt1 = V.P.delt1, dlt = V.P.dlt;
  if (V.m[I_iu][c]) {
    const double dtdl = delt1 * V.f[F_scuy][c];
    const double ca_tmp = V.f[F_u][okm] * dtdl + V.f[F_ubflxs_p][om] * dlt / V.f[F_pbu][om] +
                          (V.f[F_umfltd][okm] + V.f[F_umflsm][okm]) / fmax2(ONEMM, V.f[F_dpu][okn]);
    const double um = V.f[F_umax][c];
    V.f[F_cau][c + (size_t)k * np] = fmax2(-um * dtdl, fmin2(um * dtdl, ca_tmp));
  }
  if (V.m[I_iv][c]) {
    const double dtdl = delt1 * V.f[F_scvx][c];
    const double ca_tmp = V.f[F_v][okm] * dtdl + V.f[F_vbflxs_p][om] * dlt / V.f[F_pbv][om] +
                          (V.f[F_vmfltd][okm] + V.f[F_vmflsm][okm]) / fmax2(ONEMM, V.f[F_dpv][okn]);
    const double vm = V.f[F_vmax][c];
    V.f[F_cav][c + (size_t)k * np] = fmax2(-vm * dtdl, fmin2(vm * dtdl, ca_tmp));
  }
}

// wet-restricted neighbour plane offsets, mod_remap.F90:365-376 == mod_advect.F90:103-114
struct Nbr {
  size_t w, e, s, n, sw, se, nw, ne;
  int dxw, dyw;   // ie-iw, jn-js
};
__device__ inline Nbr wet_nbr(const DevView &V, size_t c) {
  const int *ip = V.m[I_ip], *iu = V.m[I_iu], *iv = V.m[I_iv];
  const int ni = V.ni;
  const int a = iu[c], b = iu[c + 1], d = iv[c], e = iv[c + ni];
  Nbr r;
  r.w = c - a;
  r.e = c + b;
  r.s = c - (size_t)d * ni;
  r.n = c + (size_t)e * ni;
  r.dxw = a + b;
  r.dyw = d + e;
  // corner (iw,js): if wet use it, else fall back to the centre cell
  const size_t sw = c - a - (size_t)d * ni, se = c + b - (size_t)d * ni;
  const size_t nw = c - a + (size_t)e * ni, ne = c + b + (size_t)e * ni;
  r.sw = ip[sw] ? sw : c;
  r.se = ip[se] ? se : c;
  r.nw = ip[nw] ? nw : c;
  r.ne = ip[ne] ? ne : c;
  return r;
}

// ---- mod_advect.F90:100-121 ---------------------------------------------------------------------
__global__ void k_adv_pbmin(const DevView *__restrict__ Vp) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < -1 || j > V.jj + 2 || i < -1 || i > V.ii + 2 || !V.m[I_ip][c]) return;
  const double *pb = V.f[F_p] + (size_t)V.kk * V.nplane;
  const Nbr b = wet_nbr(V, c);
  double r = fmin2(pb[b.sw], pb[b.s]);
  r = fmin2(r, pb[b.se]);
  r = fmin2(r, pb[b.w]);
  r = fmin2(r, pb[c]);
  r = fmin2(r, pb[b.e]);
  r = fmin2(r, pb[b.nw]);
  r = fmin2(r, pb[b.n]);
  r = fmin2(r, pb[b.ne]);
  WK2(V, 0)[c] = r;
}

// limited gradient of one scalar, mod_remap.F90:412-439 (same code for T, S and each tracer)
__device__ inline void limited_gradient(const double *__restrict__ f, const Nbr &b, size_t c, double dxi,
                                        double dyi, double xd, double yd, double &gx, double &gy, double &gd) {
  const double fc = f[c], fw = f[b.w], fe = f[b.e], fs = f[b.s], fn = f[b.n];
  double tx = (fe - fw) * dxi;
  double ty = (fn - fs) * dyi;
  const double q1 = tx * (-.5 - xd), q2 = tx * (.5 - xd), q3 = ty * (-.5 - yd), q4 = ty * (.5 - yd);
  const double tgmx = fmax2(q1, q2) + fmax2(q3, q4);
  const double tgmn = fmin2(q1, q2) + fmin2(q3, q4);
  const double fsw = f[b.sw], fse = f[b.se], fnw = f[b.nw], fne = f[b.ne];
  const double tfmx = fmax2(0., max8(fsw, fs, fse, fw, fe, fnw, fn, fne) - fc);
  const double tfmn = fmin2(0., min8(fsw, fs, fse, fw, fe, fnw, fn, fne) - fc);
  if (tfmx > 0. && tfmn < 0.) {
    const double q = fmin2(tfmx / fmax2(tfmx, tgmx), tfmn / fmin2(tfmn, tgmn));
    tx = tx * q;
    ty = ty * q;
    gd = fc - tx * xd - ty * yd;
  } else {
    tx = 0.;
    ty = 0.;
    gd = fc;
  }
  gx = tx;
  gy = ty;
}

// ---- mod_remap.F90:358-584 ---------------------------------------------------------------------
__global__ void k_remap_grad(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < -1 || j > V.jj + 2 || i < -1 || i > V.ii + 2 || !V.m[I_ip][c]) return;
  const int k = by_;
  const size_t np = V.nplane, ok = (size_t)k * np, okn = (size_t)(k + nn) * np;
  const double *dp = V.f[F_dp] + okn, *plo = V.f[F_p] + (size_t)(k + 1) * np;
  const Nbr b = wet_nbr(V, c);
  const double dxi = 1. / (b.dxw > 1 ? b.dxw : 1);
  const double dyi = 1. / (b.dyw > 1 ? b.dyw : 1);
  const double pm = WK2(V, 0)[c];
  // dp' = max(0,dp)+dpeps ; pup = plo - dp' ; lim = max(dpeps, min(pbmin - pup, dp'))
#define LIM(x) ({ const double d_ = fmax2(0., dp[x]) + DPEPS; fmax2(DPEPS, fmin2(pm - (plo[x] - d_), d_)); })
  const double dpsw = LIM(b.sw), dps = LIM(b.s), dpse = LIM(b.se), dpw = LIM(b.w), dpc = LIM(c);
  const double dpe = LIM(b.e), dpnw = LIM(b.nw), dpn = LIM(b.n), dpne = LIM(b.ne);
#undef LIM
  double dx = (dpe - dpw) * dxi, dy = (dpn - dps) * dyi;
  const double dgmx = .5 * (fabs(dx) + fabs(dy));
  const double dfmx = fmax2(0., max8(dpsw, dps, dpse, dpw, dpe, dpnw, dpn, dpne) - dpc);
  const double dfmn = fmin2(0., min8(dpsw, dps, dpse, dpw, dpe, dpnw, dpn, dpne) - dpc);
  double xd, yd;
  if (dfmx > 0. && dfmn < 0.) {
    const double q = fmin2(dfmx / fmax2(dfmx, dgmx), dfmn / fmin2(dfmn, -dgmx));
    const double dpt = fmax2(0., dp[c]) + DPEPS;
    dx = dx * q;
    dy = dy * q;
    xd = dx / (12. * dpt);
    yd = dy / (12. * dpt);
  } else {
    dx = 0.; dy = 0.; xd = 0.; yd = 0.;
  }
  WK(V, G_DX)[c + ok] = dx;
  WK(V, G_DY)[c + ok] = dy;
  double gx, gy, gd;
  limited_gradient(V.f[F_temp] + okn, b, c, dxi, dyi, xd, yd, gx, gy, gd);
  WK(V, G_TX)[c + ok] = gx; WK(V, G_TY)[c + ok] = gy; WK(V, G_TD)[c + ok] = gd;
  limited_gradient(V.f[F_saln] + okn, b, c, dxi, dyi, xd, yd, gx, gy, gd);
  WK(V, G_SX)[c + ok] = gx; WK(V, G_SY)[c + ok] = gy; WK(V, G_SD)[c + ok] = gd;
  for (int nt = 0; nt < V.ntr; nt++) {
    limited_gradient(V.f[F_trc] + okn + (size_t)nt * 2 * V.kk * np, b, c, dxi, dyi, xd, yd, gx, gy, gd);
    WK(V, G_TRX(nt))[c + ok] = gx; WK(V, G_TRY(nt))[c + ok] = gy; WK(V, G_TRD(nt))[c + ok] = gd;
  }
}

// non-dimensional face velocities (mod_remap.F90:588-610); zero where no u/v point exists
// (the reference zero-initialises cu,cv under use_TRC, :333-334)
__device__ inline double cu_at(const DevView &V, const double *cau, size_t x) {
  if (!V.m[I_iu][x]) return 0.;
  const double ca = cau[x];
  return ca > 0. ? ca * V.f[F_scp2i][x - 1] : ca * V.f[F_scp2i][x];
}
__device__ inline double cv_at(const DevView &V, const double *cav, size_t x) {
  if (!V.m[I_iv][x]) return 0.;
  const double ca = cav[x];
  return ca > 0. ? ca * V.f[F_scp2i][x - V.ni] : ca * V.f[F_scp2i][x];
}

// corner velocities at corner x (common corner of cells x-1-ni, x-ni, x-1, x), mod_remap.F90:623-659
__device__ inline void corner(const DevView &V, const double *cau, const double *cav, size_t x, double &cuc,
                              double &cvc) {
  const int *ip = V.m[I_ip];
  const int ni = V.ni;
  const int psw = ip[x - 1 - ni], pse = ip[x - ni], pnw = ip[x - 1], pne = ip[x];
  const int nw = psw + pse + pnw + pne;
  if (nw == 4) {
    const double cus = cu_at(V, cau, x - ni), cun = cu_at(V, cau, x);
    const double cvw = cv_at(V, cav, x - 1), cve = cv_at(V, cav, x);
    cuc = (cus * cun <= 0.) ? 0. : 2. * cus * cun / (cus + cun);
    cvc = (cvw * cve <= 0.) ? 0. : 2. * cvw * cve / (cvw + cve);
  } else if (nw == 2) {
    if (psw + pse == 2) { cuc = cu_at(V, cau, x - ni); cvc = 0.; }
    else if (pnw + pne == 2) { cuc = cu_at(V, cau, x); cvc = 0.; }
    else if (psw + pnw == 2) { cuc = 0.; cvc = cv_at(V, cav, x - 1); }
    else if (pse + pne == 2) { cuc = 0.; cvc = cv_at(V, cav, x); }
    else { cuc = 0.; cvc = 0.; }
  } else {
    cuc = 0.; cvc = 0.;
  }
}

// one polygon's contribution from donor cell x (mod_remap.F90:700-742 and its siblings)
__device__ inline void add_contrib(const DevView &V, size_t ok, const double *dp, const double *plo, size_t x,
                                   double pbface, double a, double ax, double ay, double axx, double ayy,
                                   double axy, Acc &A) {
  const double dpt = fmax2(0., dp[x]) + DPEPS;
  const double pup = plo[x] - dpt;
  const double dl = fmin2(dpt, fmax2(0., pbface - pup));
  const double dx = WK(V, G_DX)[x + ok], dy = WK(V, G_DY)[x + ok];
  const double fd = a * dl + ax * dx + ay * dy;
  A.fd = A.fd + fd;
  const double qx = ax * dl + axx * dx + axy * dy;
  const double qy = ay * dl + axy * dx + ayy * dy;
  A.ft = A.ft + fd * WK(V, G_TD)[x + ok] + qx * WK(V, G_TX)[x + ok] + qy * WK(V, G_TY)[x + ok];
  A.fs = A.fs + fd * WK(V, G_SD)[x + ok] + qx * WK(V, G_SX)[x + ok] + qy * WK(V, G_SY)[x + ok];
#pragma unroll
  for (int nt = 0; nt < MAXTR; nt++)
    if (nt < V.ntr)
      A.ftr[nt] = A.ftr[nt] + fd * WK(V, G_TRD(nt))[x + ok] + qx * WK(V, G_TRX(nt))[x + ok] +
                  qy * WK(V, G_TRY(nt))[x + ok];
}

// ---- mod_remap.F90:588-1462 ----------------------------------------------------------------------
__global__ void k_remap_flux(const DevView *__restrict__ Vp, int n, int mm, int nn) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  const int k = by_;
  const int ni = V.ni, ntr = V.ntr;
  const size_t np = V.nplane, ok = (size_t)k * np, okn = (size_t)(k + nn) * np, okm = (size_t)(k + mm) * np;
  const double *dp = V.f[F_dp] + okn, *plo = V.f[F_p] + (size_t)(k + 1) * np;
  const double *cau = V.f[F_cau] + ok, *cav = V.f[F_cav] + ok;
  const double *scp2 = V.f[F_scp2], *scp2i = V.f[F_scp2i];
  const bool in_u = j >= 0 && j <= V.jj + 1 && i >= 0 && i <= V.ii + 2;
  const bool in_v = j >= 0 && j <= V.jj + 2 && i >= 0 && i <= V.ii + 1;
  if (!in_u && !in_v) return;
  const bool do_u = in_u && V.m[I_iu][c], do_v = in_v && V.m[I_iv][c];
  double cuc0 = 0., cvc0 = 0.;
  if (do_u || do_v) corner(V, cau, cav, c, cuc0, cvc0);

  if (in_u) {
    Acc A;
    A.fd = 0.; A.ft = 0.; A.fs = 0.;
#pragma unroll
    for (int nt = 0; nt < MAXTR; nt++) A.ftr[nt] = 0.;
    if (do_u) {
      double cuc1, cvc1;
      corner(V, cau, cav, c + ni, cuc1, cvc1);
      const double cu = cu_at(V, cau, c);
      const double ym = -.5 * (cvc0 + cvc1);
      const double xm = ((ym + .5) * cuc0 - (ym - .5) * cuc1 - 2. * cu) / (1. + cvc0 - cvc1);
      const size_t ic = cu > 0. ? c - 1 : c;          // donor column
      const double sh = cu > 0. ? .5 : -.5;
      const double pb = V.f[F_pbu][c + (size_t)(n - 1) * np];
      double a, ax, ay, axx, ayy, axy, x2, y2, x4, y4;
      if (cvc0 > 0.) {
        const double xc0 = (xm * cvc0 - cuc0 * (ym + .5)) / (cvc0 + ym + .5);
        const double xc1 = xc0 * scp2[ic] * scp2i[ic - ni];
        x4 = xc0 + sh;
        y4 = -.5;
        triint(scp2[ic - ni], xc1 + sh, .5, -cuc0 + sh, -cvc0 + .5, sh, .5, a, ax, ay, axx, ayy, axy);
        add_contrib(V, ok, dp, plo, ic - ni, pb, a, ax, ay, axx, ayy, axy, A);
      } else {
        x4 = -cuc0 + sh;
        y4 = -cvc0 - .5;
      }
      if (cvc1 < 0.) {
        const double xc0 = (xm * cvc1 - cuc1 * (ym - .5)) / (cvc1 + ym - .5);
        const double xc1 = xc0 * scp2[ic] * scp2i[ic + ni];
        x2 = xc0 + sh;
        y2 = .5;
        triint(scp2[ic + ni], xc1 + sh, -.5, sh, -.5, -cuc1 + sh, -cvc1 - .5, a, ax, ay, axx, ayy, axy);
        add_contrib(V, ok, dp, plo, ic + ni, pb, a, ax, ay, axx, ayy, axy, A);
      } else {
        x2 = -cuc1 + sh;
        y2 = -cvc1 + .5;
      }
      penint(scp2[ic], sh, .5, x2, y2, xm + sh, ym, x4, y4, sh, -.5, a, ax, ay, axx, ayy, axy);
      add_contrib(V, ok, dp, plo, ic, pb, a, ax, ay, axx, ayy, axy, A);
      // mod_remap.F90:1054-1056
      V.f[F_uflx][c + okm] = V.f[F_uflx][c + okm] + A.fd;
      V.f[F_utflx][c + okm] = V.f[F_utflx][c + okm] + A.ft;
      V.f[F_usflx][c + okm] = V.f[F_usflx][c + okm] + A.fs;
    }
    WK(V, W_FDU(ntr))[c + ok] = A.fd;
    WK(V, W_FTU(ntr))[c + ok] = A.ft;
    WK(V, W_FSU(ntr))[c + ok] = A.fs;
#pragma unroll
    for (int nt = 0; nt < MAXTR; nt++)
      if (nt < ntr) WK(V, W_FTRU(ntr, nt))[c + ok] = A.ftr[nt];
  }

  if (in_v) {
    Acc A;
    A.fd = 0.; A.ft = 0.; A.fs = 0.;
#pragma unroll
    for (int nt = 0; nt < MAXTR; nt++) A.ftr[nt] = 0.;
    if (do_v) {
      double cuc1, cvc1;
      corner(V, cau, cav, c + 1, cuc1, cvc1);
      const double cv = cv_at(V, cav, c);
      const double xm = -.5 * (cuc0 + cuc1);
      const double ym = ((xm + .5) * cvc0 - (xm - .5) * cvc1 - 2. * cv) / (1. + cuc0 - cuc1);
      const size_t jc = cv > 0 ? c - ni : c;           // donor row
      const double sh = cv > 0 ? .5 : -.5;
      const double pb = V.f[F_pbv][c + (size_t)(n - 1) * np];
      double a, ax, ay, axx, ayy, axy, x2, y2, x4, y4;
      if (cuc0 > 0.) {
        const double yc0 = (ym * cuc0 - cvc0 * (xm + .5)) / (cuc0 + xm + .5);
        const double yc1 = yc0 * scp2[jc] * scp2i[jc - 1];
        x2 = -.5;
        y2 = yc0 + sh;
        triint(scp2[jc - 1], .5, yc1 + sh, .5, sh, -cuc0 + .5, -cvc0 + sh, a, ax, ay, axx, ayy, axy);
        add_contrib(V, ok, dp, plo, jc - 1, pb, a, ax, ay, axx, ayy, axy, A);
      } else {
        x2 = -cuc0 - .5;
        y2 = -cvc0 + sh;
      }
      if (cuc1 < 0.) {
        const double yc0 = (ym * cuc1 - cvc1 * (xm - .5)) / (cuc1 + xm - .5);
        const double yc1 = yc0 * scp2[jc] * scp2i[jc + 1];
        x4 = .5;
        y4 = yc0 + sh;
        triint(scp2[jc + 1], -.5, yc1 + sh, -cuc1 - .5, -cvc1 + sh, -.5, sh, a, ax, ay, axx, ayy, axy);
        add_contrib(V, ok, dp, plo, jc + 1, pb, a, ax, ay, axx, ayy, axy, A);
      } else {
        x4 = -cuc1 + .5;
        y4 = -cvc1 + sh;
      }
      penint(scp2[jc], -.5, sh, x2, y2, xm, ym + sh, x4, y4, .5, sh, a, ax, ay, axx, ayy, axy);
      add_contrib(V, ok, dp, plo, jc, pb, a, ax, ay, axx, ayy, axy, A);
      // mod_remap.F90:1455-1457: assignment (not accumulation) for the v-components
      V.f[F_vflx][c + okm] = A.fd;
      V.f[F_vtflx][c + okm] = A.ft;
      V.f[F_vsflx][c + okm] = A.fs;
    }
    WK(V, W_FDV(ntr))[c + ok] = A.fd;
    WK(V, W_FTV(ntr))[c + ok] = A.ft;
    WK(V, W_FSV(ntr))[c + ok] = A.fs;
#pragma unroll
    for (int nt = 0; nt < MAXTR; nt++)
      if (nt < ntr) WK(V, W_FTRV(ntr, nt))[c + ok] = A.ftr[nt];
  }
}

// ---- mod_remap.F90:1468-1520 (+ the in-place dp side effect of :297-303 on the outer ring) --------
__global__ void k_remap_update(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < -2 || j > V.jj + 3 || i < -2 || i > V.ii + 3 || !V.m[I_ip][c]) return;
  const int k = by_, ntr = V.ntr;
  const size_t np = V.nplane, ok = (size_t)k * np, okn = (size_t)(k + nn) * np;
  double *dp = V.f[F_dp] + okn;
  const double q = fmax2(0., dp[c]) + DPEPS;
  if (j < 0 || j > V.jj + 1 || i < 0 || i > V.ii + 1) {
    dp[c] = q;
    return;
  }
  const size_t e = c + 1, nb = c + V.ni;
  const double s2i = V.f[F_scp2i][c];
  const double *fdu = WK(V, W_FDU(ntr)) + ok, *fdv = WK(V, W_FDV(ntr)) + ok;
  const double *ftu = WK(V, W_FTU(ntr)) + ok, *ftv = WK(V, W_FTV(ntr)) + ok;
  const double *fsu = WK(V, W_FSU(ntr)) + ok, *fsv = WK(V, W_FSV(ntr)) + ok;
  const double dpn = q - (fdu[e] - fdu[c] + fdv[nb] - fdv[c]) * s2i;
  double *temp = V.f[F_temp] + okn, *saln = V.f[F_saln] + okn;
  temp[c] = (q * temp[c] - (ftu[e] - ftu[c] + ftv[nb] - ftv[c]) * s2i) / dpn;
  saln[c] = (q * saln[c] - (fsu[e] - fsu[c] + fsv[nb] - fsv[c]) * s2i) / dpn;
  for (int nt = 0; nt < ntr; nt++) {
    if (trc_skip_adv(V.P, nt + 1)) continue;                   // phy/mod_remap.F90:1497-1499
    double *tr = V.f[F_trc] + okn + (size_t)nt * 2 * V.kk * np;
    const double *fu = WK(V, W_FTRU(ntr, nt)) + ok, *fv = WK(V, W_FTRV(ntr, nt)) + ok;
    tr[c] = (q * tr[c] - (fu[e] - fu[c] + fv[nb] - fv[c]) * s2i) / dpn;
  }
  dp[c] = fmax2(0., dpn - DPEPS);
}

int st_advect(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  const DevView &h = c->h;
  const size_t np = h.nplane;
  if (h.ntr > MAXTR) return ctx_fail(c, "advect: more tracers than MAXTR");
  hipLaunchKernelGGL(k_adv_flux_area, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, m, mm, nn);
  if (h.P.advmth == 1) {                                                              // mod_advect:155-164
    {
      TimeScope ts(c, "cppm");
      if (int rc = st_cppm(c, m, n, mm, nn, k1m, k1n)) return rc;
    }
    if (int rc = st_xctilr(c, h.f[F_dp] + (size_t)(k1n - 1) * np, 1, h.kk, 1, 1, 1)) return rc;
    if (int rc = st_xctilr(c, h.f[F_temp] + (size_t)(k1n - 1) * np, 1, h.kk, 1, 1, 1)) return rc;
    if (int rc = st_xctilr(c, h.f[F_saln] + (size_t)(k1n - 1) * np, 1, h.kk, 1, 1, 1)) return rc;
    for (int nt = 0; nt < h.ntr; nt++)
      if (int rc = st_xctilr(c, h.f[F_trc] + ((size_t)(k1n - 1) + (size_t)nt * 2 * h.kk) * np, 1, h.kk, 1, 1, 1)) return rc;
    return 0;
  }
  hipLaunchKernelGGL(k_adv_pbmin, plane_grid(h), dim3(256), 0, c->stream, c->d);
  double *ptrs[2 + MAXTR] = {h.f[F_cau], h.f[F_cav]};                                 // mod_advect:124-131
  int nl[2 + MAXTR] = {h.kk, h.kk}, it[2 + MAXTR] = {13, 14};
  int nf = 2;
  for (int nt = 0; nt < h.ntr; nt++) {
    if (trc_skip_adv(h.P, nt + 1)) continue;                                          // :127-129
    ptrs[nf] = h.f[F_trc] + ((size_t)(k1n - 1) + (size_t)nt * 2 * h.kk) * np;
    nl[nf] = h.kk;
    it[nf++] = 1;
  }
  // RCCL tiles: the exchange (pack, send/recv, unpack) goes to the second stream and the tiles of k_remap_tile whose
  // rim lies inside the tile -- they read no halo point of cau, cav or the tracers -- run meanwhile; the tiles along the
  // edge follow when the halos have landed.  The exchange reads interior strips and writes halo points only.
  const bool ovl = c->tiling.rccl && c->xstream && c->halo_overlap && c->remap_v == 2 && h.nreg != 2 && !c->timing;
  if (ovl) {
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->xstream, c->ev_fork, 0));
    c->halo_stream = c->xstream;
    const int rc = st_xctilr_multi(c, nf, ptrs, nl, 3, 3, it);
    c->halo_stream = nullptr;
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev_join, c->xstream));
    if (int rc2 = remap_tile_launch(c, n, mm, nn, 1)) return rc2;
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    if (int rc2 = remap_tile_launch(c, n, mm, nn, 2)) return rc2;
    hipLaunchKernelGGL(k_remap_update, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, nn);
  } else {
    if (int rc = st_xctilr_multi(c, nf, ptrs, nl, 3, 3, it)) return rc;
    TimeScope ts(c, "remap");
    if (c->remap_v == 2) {
      if (int rc = remap_tile_launch(c, n, mm, nn, 0)) return rc;
    } else {
      hipLaunchKernelGGL(k_remap_grad, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, nn);
      hipLaunchKernelGGL(k_remap_flux, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, n, mm, nn);
    }
    hipLaunchKernelGGL(k_remap_update, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, nn);
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}
