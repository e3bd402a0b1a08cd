// pbcor1 / pbcor2 -- baroclinic mass-flux correction so that the vertical sum of layer
// thicknesses matches the bottom pressure of the barotropic solution.
// phy/mod_pbcor.F90:66-412 (pbcor1) and :416-743 (pbcor2); bmcmth 'uc' and 'dluc'.
//
// Kernels (thread per point of the padded plane; layers on blockIdx.y where independent):
//   k_pbc_pscan    [pbcor2: dp = max(0,dp)+epsilp;] p(k+1) = p(k)+dp, j,i = 0..+1   (column)
//   k_pbc_total    utot = dlt*ubflxs - sum_k uflx(k)   (u: i 1..ii+1 ; v: j 1..jj+1)  (column)
//   k_pbc_tile     upstream-column fluxes of mass, salt, heat, tracers, accumulation into uflx.., divergence update of
//                  dp, S, T, trc [, sigma] of a layer in one LDS-tiled kernel (stage_pbcor_tile.hip)   (i,j,k)
//   k_pbc_rescale_from  p scan, pbfac = pb/p(kk+1), dp *= pbfac; new state into place   (column; stage_pbcor_tile.hip)
// Roofline: HBM.
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

#define S2_PBUT 1
#define S2_PBVT 2

// from_remap: remap left the new dp in the work space (stage_remap_tile.hip, FOLD)
__global__ void k_pbc_pscan(const DevView *__restrict__ Vp, int which, int offc, int from_remap) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 0 || j > V.jj + 1 || i < 0 || i > V.ii + 1 || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  gd_t dp = from_remap ? WK(V, R_DP(V.ntr)) : V.f[F_dp] + (size_t)offc * np, p = V.f[F_p];
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
  gcd_t pbot = V.f[F_p] + (size_t)V.kk * np;
  if (j >= 1 && j <= V.jj && i >= 1 && i <= V.ii + 1 && V.m[I_iu][c]) {
    gcd_t bfx = which == 1 ? V.f[F_ubflxs_p] + (size_t)(m - 1) * np : V.f[F_ubflxs] + (size_t)(n - 1) * np;
    double t = dlt * bfx[c];
    if (V.P.bmcmth == 1) WK2(V, S2_PBUT)[c] = fmin2(pbot[c], pbot[c - 1]);
    gcd_t uflx = V.f[F_uflx] + (size_t)offf * np;
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
    gcd_t bfx = which == 1 ? V.f[F_vbflxs_p] + (size_t)(m - 1) * np : V.f[F_vbflxs] + (size_t)(n - 1) * np;
    double t = dlt * bfx[c];
    if (V.P.bmcmth == 1) WK2(V, S2_PBVT)[c] = fmin2(pbot[c], pbot[c - V.ni]);
    gcd_t vflx = V.f[F_vflx] + (size_t)offf * np;
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

static int pbcor(blomgpu_ctx *c, int which, int m, int n, int mm, int nn, int k1m) {
  const DevView &h = c->h;
  const size_t np = h.nplane;
  const int offc = which == 1 ? nn : mm, offf = which == 1 ? mm : nn;
  if (which == 2) {                                                             // :434-440
    // (one gather launch for the lot: up to 16 plane stacks per launch)
    std::vector<double *> ptrs = {h.f[F_ubflxs] + (size_t)(n - 1) * np, h.f[F_vbflxs] + (size_t)(n - 1) * np};
    std::vector<int> nl = {1, 1}, it = {13, 14};
    for (int nt = 0; nt < h.ntr; nt++) {
      ptrs.push_back(h.f[F_trc] + ((size_t)(k1m - 1) + (size_t)nt * 2 * h.kk) * np);
      nl.push_back(h.kk);
      it.push_back(1);
    }
    for (int f = 0; f < (int)ptrs.size(); f += 16) {
      const int g = (int)ptrs.size() - f < 16 ? (int)ptrs.size() - f : 16;
      if (int rc = st_xctilr_multi(c, g, ptrs.data() + f, nl.data() + f, 1, 1, it.data() + f)) return rc;
    }
  }
  TimeScope ts(c, which == 1 ? "pbcor1" : "pbcor2");
  const int from_remap = which == 1 && c->in_sequence && c->remap_handed_over ? 1 : 0;
  if (which == 1) c->remap_handed_over = false;
  hipLaunchKernelGGL(k_pbc_pscan, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, which, offc, from_remap);
  hipLaunchKernelGGL(k_pbc_total, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, which, m, n, offf);
  if (int rc = pbcor_tile_launch(c, which, m, offc, offf, from_remap)) return rc;
  HIPCHK(c, hipGetLastError());
  return 0;
}

int st_pbcor1(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) { (void)k1n; return pbcor(c, 1, m, n, mm, nn, k1m); }
int st_pbcor2(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) { (void)k1n; return pbcor(c, 2, m, n, mm, nn, k1m); }
