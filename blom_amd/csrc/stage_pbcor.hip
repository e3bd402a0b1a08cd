// pbcor1 / pbcor2 -- baroclinic mass-flux correction so that the vertical sum of layer
// thicknesses matches the bottom pressure of the barotropic solution.
// phy/mod_pbcor.F90:66-412 (pbcor1) and :416-743 (pbcor2); bmcmth 'uc' and 'dluc'.
//
// Kernels (thread per point of the padded plane; layers on blockIdx.y where independent):
//   k_pbc_pscan    [pbcor2: dp = max(0,dp)+epsilp;] p(k+1) = p(k)+dp, j,i = 0..+1   (column)
//   k_pbc_total    utot = dlt*ubflxs - sum_k uflx(k)   (u: i 1..ii+1 ; v: j 1..jj+1)  (column)
//   k_pbc_flux     upstream-column fluxes of mass, salt, heat, tracers; accumulate uflx.. (i,j,k)
//   k_pbc_update   divergence update of dp, S, T, trc [, sigma]                       (i,j,k)
//   k_pbc_rescale  p scan, pbfac = pb/p(kk+1), dp *= pbfac                            (column)
// The per-layer 2-D scratch arrays uflux,uflux2,uflux3,uflxtr of the reference become work-space
// fields over all layers (layers are independent given utot).  Roofline: HBM.
// Measured alternative, not kept: one kernel evaluating the four face fluxes of a cell and updating it (new
// state to work planes, moved into place by the rescale pass) -- 0.57/0.63 ms against 0.52/0.56 ms for
// pbcor1/2: the 4 divisions per cell and the column-wise copy cost more than the 8 flux planes saved.
#include "blomgpu_internal.h"
#include "eos.h"

#define EPSILP 1.e-12
#define DPEPS1 1.e-5   // phy/mod_pbcor.F90:58
#define DPEPS2 1.e-7   // phy/mod_pbcor.F90:59

#define THREAD_IJ(V)                                                       \
  unsigned bx_, by_;                                                       \
  xcd_block(bx_, by_);                                                     \
  const int t_ = bx_ * blockDim.x + threadIdx.x;                           \
  (void)by_;                                                               \
  if (t_ >= (V).nplane) return;                                            \
  const int i = t_ % (V).ni - (NBDY - 1), j = t_ / (V).ni - (NBDY - 1);    \
  const size_t c = t_

// work-space slots
#define S_UF 0
#define S_UF2 1
#define S_UF3 2
#define S_VF 3
#define S_VF2 4
#define S_VF3 5
#define S_UTR(nt) (6 + 2 * (nt))
#define S_VTR(nt) (7 + 2 * (nt))
#define S2_PBUT 1
#define S2_PBVT 2

__global__ void k_pbc_pscan(const DevView *__restrict__ Vp, int which, int offc) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 0 || j > V.jj + 1 || i < 0 || i > V.ii + 1 || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  double *dp = V.f[F_dp] + (size_t)offc * np, *p = V.f[F_p];
  double acc = p[c];
  int k = 0;
  for (; k + COLUMN_U <= V.kk; k += COLUMN_U) {              // COLUMN_U levels' loads in flight (blomgpu_internal.h)
    double v[COLUMN_U];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) v[u] = dp[c + (size_t)(k + u) * np];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) {
      double d = v[u];
      if (which == 2) { d = fmax2(0., d) + EPSILP; dp[c + (size_t)(k + u) * np] = d; }   // :448
      acc = acc + d;
      p[c + (size_t)(k + u + 1) * np] = acc;
    }
  }
  for (; k < V.kk; k++) {
    double d = dp[c + (size_t)k * np];
    if (which == 2) { d = fmax2(0., d) + EPSILP; dp[c + (size_t)k * np] = d; }   // :448
    acc = acc + d;
    p[c + (size_t)(k + 1) * np] = acc;
  }
}

__global__ void k_pbc_total(const DevView *__restrict__ Vp, int which, int m, int n, int offf) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  const size_t np = V.nplane;
  const double dlt = V.P.dlt;
  const double *pbot = V.f[F_p] + (size_t)V.kk * np;
  if (j >= 1 && j <= V.jj && i >= 1 && i <= V.ii + 1 && V.m[I_iu][c]) {
    const double *bfx = which == 1 ? V.f[F_ubflxs_p] + (size_t)(m - 1) * np : V.f[F_ubflxs] + (size_t)(n - 1) * np;
    double t = dlt * bfx[c];
    if (V.P.bmcmth == 1) WK2(V, S2_PBUT)[c] = fmin2(pbot[c], pbot[c - 1]);
    const double *uflx = V.f[F_uflx] + (size_t)offf * np;
    for (int k0 = 0; k0 < V.kk; k0 += COLUMN_U) {                // COLUMN_U levels' loads in flight (blomgpu_internal.h)
      double a0[COLUMN_U];
#pragma unroll
      for (int u = 0; u < COLUMN_U; u++) a0[u] = uflx[c + (size_t)(k0 + u < V.kk ? k0 + u : V.kk - 1) * np];
#pragma unroll
      for (int u = 0; u < COLUMN_U; u++)
        if (k0 + u < V.kk) t = t - a0[u];
    }
    (which == 1 ? V.f[F_utotm] : V.f[F_utotn])[c] = t;
  }
  if (j >= 1 && j <= V.jj + 1 && i >= 1 && i <= V.ii && V.m[I_iv][c]) {
    const double *bfx = which == 1 ? V.f[F_vbflxs_p] + (size_t)(m - 1) * np : V.f[F_vbflxs] + (size_t)(n - 1) * np;
    double t = dlt * bfx[c];
    if (V.P.bmcmth == 1) WK2(V, S2_PBVT)[c] = fmin2(pbot[c], pbot[c - V.ni]);
    const double *vflx = V.f[F_vflx] + (size_t)offf * np;
    for (int k0 = 0; k0 < V.kk; k0 += COLUMN_U) {
      double a0[COLUMN_U];
#pragma unroll
      for (int u = 0; u < COLUMN_U; u++) a0[u] = vflx[c + (size_t)(k0 + u < V.kk ? k0 + u : V.kk - 1) * np];
#pragma unroll
      for (int u = 0; u < COLUMN_U; u++)
        if (k0 + u < V.kk) t = t - a0[u];
    }
    (which == 1 ? V.f[F_vtotm] : V.f[F_vtotn])[c] = t;
  }
}

__global__ void k_pbc_flux(const DevView *__restrict__ Vp, int which, int offc, int offf) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  const int k = by_, ntr = V.ntr;
  const size_t np = V.nplane, ok = (size_t)k * np, okc = (size_t)(k + offc) * np, okf = (size_t)(k + offf) * np;
  const double *dp = V.f[F_dp] + okc, *saln = V.f[F_saln] + okc, *temp = V.f[F_temp] + okc;
  const double *p = V.f[F_p];
  const double *pbot = p + (size_t)V.kk * np;
  if (j >= 1 && j <= V.jj && i >= 1 && i <= V.ii + 1) {
    double f = 0., f2 = 0., f3 = 0.;
    const bool w = V.m[I_iu][c] != 0;
    size_t up = c;
    if (w) {
      const double tot = (which == 1 ? V.f[F_utotm] : V.f[F_utotn])[c];
      up = tot > 0. ? c - 1 : c;
      if (V.P.bmcmth == 0) f = tot * dp[up] / pbot[up];
      else {
        const double pbt = WK2(V, S2_PBUT)[c];
        f = tot * fmax2(0., fmin2(pbt, p[up + (size_t)(k + 1) * np]) - p[up + ok]) / pbt;
      }
      f2 = f * saln[up];
      f3 = f * temp[up];
      V.f[F_uflx][c + okf] = V.f[F_uflx][c + okf] + f;
      V.f[F_usflx][c + okf] = V.f[F_usflx][c + okf] + f2;
      V.f[F_utflx][c + okf] = V.f[F_utflx][c + okf] + f3;
    }
    WK(V, S_UF)[c + ok] = f;
    WK(V, S_UF2)[c + ok] = f2;
    WK(V, S_UF3)[c + ok] = f3;
    for (int nt = 0; nt < ntr; nt++)
      WK(V, S_UTR(nt))[c + ok] = w ? f * V.f[F_trc][up + okc + (size_t)nt * 2 * V.kk * np] : 0.;
  }
  if (j >= 1 && j <= V.jj + 1 && i >= 1 && i <= V.ii) {
    double f = 0., f2 = 0., f3 = 0.;
    const bool w = V.m[I_iv][c] != 0;
    size_t up = c;
    if (w) {
      const double tot = (which == 1 ? V.f[F_vtotm] : V.f[F_vtotn])[c];
      up = tot > 0. ? c - V.ni : c;
      if (V.P.bmcmth == 0) f = tot * dp[up] / pbot[up];
      else {
        const double pbt = WK2(V, S2_PBVT)[c];
        f = tot * fmax2(0., fmin2(pbt, p[up + (size_t)(k + 1) * np]) - p[up + ok]) / pbt;
      }
      f2 = f * saln[up];
      f3 = f * temp[up];
      V.f[F_vflx][c + okf] = V.f[F_vflx][c + okf] + f;
      V.f[F_vsflx][c + okf] = V.f[F_vsflx][c + okf] + f2;
      V.f[F_vtflx][c + okf] = V.f[F_vtflx][c + okf] + f3;
    }
    WK(V, S_VF)[c + ok] = f;
    WK(V, S_VF2)[c + ok] = f2;
    WK(V, S_VF3)[c + ok] = f3;
    for (int nt = 0; nt < ntr; nt++)
      WK(V, S_VTR(nt))[c + ok] = w ? f * V.f[F_trc][up + okc + (size_t)nt * 2 * V.kk * np] : 0.;
  }
}

__global__ void k_pbc_update(const DevView *__restrict__ Vp, int which, int offc) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const int k = by_, ntr = V.ntr;
  const size_t np = V.nplane, ok = (size_t)k * np, okc = (size_t)(k + offc) * np, e = c + 1, nb = c + V.ni;
  double *dp = V.f[F_dp] + okc, *saln = V.f[F_saln] + okc, *temp = V.f[F_temp] + okc;
  const double *uf = WK(V, S_UF) + ok, *uf2 = WK(V, S_UF2) + ok, *uf3 = WK(V, S_UF3) + ok;
  const double *vf = WK(V, S_VF) + ok, *vf2 = WK(V, S_VF2) + ok, *vf3 = WK(V, S_VF3) + ok;
  const double dv = uf[e] - uf[c] + vf[nb] - vf[c];
  const double dv2 = uf2[e] - uf2[c] + vf2[nb] - vf2[c];
  const double dv3 = uf3[e] - uf3[c] + vf3[nb] - vf3[c];
  const double s2i = V.f[F_scp2i][c];
  double dpo = dp[c];
  if (which == 1) {                                   // :339-361
    const double dpn = fmax2(0., dpo - dv * s2i);
    dpo = dpo + DPEPS1;
    const double dpni = 1. / (dpn + DPEPS1);
    saln[c] = (dpo * saln[c] - dv2 * s2i) * dpni;
    temp[c] = (dpo * temp[c] - dv3 * s2i) * dpni;
    for (int nt = 0; nt < ntr; nt++) {
      if (trc_skip_adv(V.P, nt + 1)) continue;          // phy/mod_pbcor.F90:353-355 (pbcor2, :684, has no such test)
      double *tr = V.f[F_trc] + okc + (size_t)nt * 2 * V.kk * np;
      const double *fu = WK(V, S_UTR(nt)) + ok, *fv = WK(V, S_VTR(nt)) + ok;
      tr[c] = (dpo * tr[c] - (fu[e] - fu[c] + fv[nb] - fv[c]) * s2i) * dpni;
    }
    dp[c] = dpn < DPEPS2 ? 0. : dpn;
  } else {                                            // :671-692
    double dpn = dpo - s2i * dv;
    const double dpni = 1. / dpn;
    const double sn = (dpo * saln[c] - s2i * dv2) * dpni;
    const double tn = (dpo * temp[c] - s2i * dv3) * dpni;
    saln[c] = sn;
    temp[c] = tn;
    for (int nt = 0; nt < ntr; nt++) {
      double *tr = V.f[F_trc] + okc + (size_t)nt * 2 * V.kk * np;
      const double *fu = WK(V, S_UTR(nt)) + ok, *fv = WK(V, S_VTR(nt)) + ok;
      tr[c] = (dpo * tr[c] - (fu[e] - fu[c] + fv[nb] - fv[c]) * s2i) * dpni;
    }
    V.f[F_sigma][c + okc] = eos::sig(V.P, tn, sn);
    dpn = dpn - EPSILP;
    dp[c] = dpn < DPEPS2 ? 0. : dpn;
  }
}

__global__ void k_pbc_rescale(const DevView *__restrict__ Vp, int which, int m, int offc) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  double *dp = V.f[F_dp] + (size_t)offc * np, *p = V.f[F_p];
  double acc = p[c];
  for (int k = 0; k < V.kk; k++) {
    acc = acc + dp[c + (size_t)k * np];
    p[c + (size_t)(k + 1) * np] = acc;
  }
  const double pbfac = (which == 1 ? V.f[F_pb_p][c] : V.f[F_pb][c + (size_t)(m - 1) * np]) / acc;
  acc = p[c];
  for (int k = 0; k < V.kk; k++) {
    const double d = dp[c + (size_t)k * np] * pbfac;
    dp[c + (size_t)k * np] = d;
    if (which == 2) { acc = acc + d; p[c + (size_t)(k + 1) * np] = acc; }
  }
}

static int pbcor(blomgpu_ctx *c, int which, int m, int n, int mm, int nn, int k1m) {
  const DevView &h = c->h;
  const size_t np = h.nplane;
  const int offc = which == 1 ? nn : mm, offf = which == 1 ? mm : nn;
  if (which == 2) {                                                             // :434-440
    if (int rc = st_xctilr(c, h.f[F_ubflxs] + (size_t)(n - 1) * np, 1, 1, 1, 1, 13)) return rc;
    if (int rc = st_xctilr(c, h.f[F_vbflxs] + (size_t)(n - 1) * np, 1, 1, 1, 1, 14)) return rc;
    for (int nt = 0; nt < h.ntr; nt++)
      if (int rc = st_xctilr(c, h.f[F_trc] + ((size_t)(k1m - 1) + (size_t)nt * 2 * h.kk) * np, 1, h.kk, 1, 1, 1))
        return rc;
  }
  TimeScope ts(c, which == 1 ? "pbcor1" : "pbcor2");
  hipLaunchKernelGGL(k_pbc_pscan, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, which, offc);
  hipLaunchKernelGGL(k_pbc_total, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, which, m, n, offf);
  if (c->pbcor_v == 2) {
    if (int rc = pbcor_tile_launch(c, which, m, offc, offf)) return rc;
  } else {
    hipLaunchKernelGGL(k_pbc_flux, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, which, offc, offf);
    hipLaunchKernelGGL(k_pbc_update, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, which, offc);
    hipLaunchKernelGGL(k_pbc_rescale, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, which, m, offc);
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}

int st_pbcor1(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) { (void)k1n; return pbcor(c, 1, m, n, mm, nn, k1m); }
int st_pbcor2(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) { (void)k1n; return pbcor(c, 2, m, n, mm, nn, k1m); }
