// diffus -- layer-wise lateral diffusion of T, S and tracers, phy/mod_diffus.F90:41-185.
//
// Two streaming kernels over (i,j,k) (k on blockIdx.y), exactly the reference's two sweeps:
//   k_diffus_flux   : face fluxes q*(T_l - T_r) at u- and v-points, accumulation into
//                     utflx/usflx/vtflx/vsflx               (:89-134)
//   k_diffus_update : flux divergence update of T,S,trc and sigma = sig(T,S)   (:137-160)
// The update cannot be fused into the flux kernel without double-buffering T,S (a
// neighbour's flux reads the old value), so the fluxes make one round trip through HBM.
// Algorithmic bytes: (19 + 2*ntr) F.  Roofline: HBM.
#include "blomgpu_internal.h"
#include "eos.h"

#define DPEPS 1.e-5   // phy/mod_diffus.F90:55-56

// SHFL: the west neighbour's dp, S, T, difiso come from the adjacent lane (DPP row shift / __shfl_up) instead of a
// second, cached load; lane 0 of a wavefront still loads.  A/B option diffus_shfl; measured 3 % slower than the cached load.
template <bool SHFL>
__global__ void k_diffus_flux(const DevView *__restrict__ Vp, int mm, int nn) {
  const DevView &V = *Vp;
  unsigned bx_, by_;
  xcd_block(bx_, by_);
  const int t = bx_ * blockDim.x + threadIdx.x;
  if (t >= V.nplane) return;
  const int i = t % V.ni - (NBDY - 1), j = t / V.ni - (NBDY - 1);
  const int k = by_;
  const size_t np = V.nplane, c = t, w = c - 1, s = c - V.ni;
  const size_t okn = (size_t)(k + nn) * np, okm = (size_t)(k + mm) * np, ok = (size_t)k * np;
  const double *dp = V.f[F_dp] + okn, *temp = V.f[F_temp] + okn, *saln = V.f[F_saln] + okn;
  const double *difiso = V.f[F_difiso] + ok;
  const double delt1 = V.P.delt1;
  double dpw, snw, tmw, dfw;
  if (SHFL) {
    const double dc = dp[c], sc = saln[c], tc = temp[c], fc = difiso[c];
    const bool lane0 = (threadIdx.x & 63) == 0;
    dpw = __shfl_up(dc, 1); snw = __shfl_up(sc, 1); tmw = __shfl_up(tc, 1); dfw = __shfl_up(fc, 1);
    if (lane0 && c > 0) { dpw = dp[w]; snw = saln[w]; tmw = temp[w]; dfw = difiso[w]; }
  }
  if (V.m[I_iu][c] && j >= 0 && j <= V.jj + 1 && i >= 0 && i <= V.ii + 2) {
    if (!SHFL) { dpw = dp[w]; snw = saln[w]; tmw = temp[w]; dfw = difiso[w]; }
    const double q = delt1 * .5 * (dfw + difiso[c]) * V.f[F_scuy][c] * V.f[F_scuxi][c] *
                     fmax2(fmin2(dpw, dp[c]), DPEPS);
    const double fs = q * (snw - saln[c]);
    const double ft = q * (tmw - temp[c]);
    V.f[F_usflld][c + okm] = fs;
    V.f[F_utflld][c + okm] = ft;
    for (int nt = 0; nt < V.ntr; nt++) {
      if (trc_skip_dif(V.P, nt + 1)) continue;
      const double *tr = V.f[F_trc] + okn + (size_t)nt * 2 * V.kk * np;
      WK(V, 2 * nt)[c + ok] = q * (tr[w] - tr[c]);
    }
    V.f[F_usflx][c + okm] = V.f[F_usflx][c + okm] + fs;
    V.f[F_utflx][c + okm] = V.f[F_utflx][c + okm] + ft;
  } else if (j >= 0 && j <= V.jj + 1 && i >= 0 && i <= V.ii + 2) {
    // uflxtr is 0 where no u-point exists (trc/mod_tracers.F90:166-186); the work space is shared
    for (int nt = 0; nt < V.ntr; nt++) WK(V, 2 * nt)[c + ok] = 0.;
  }
  if (V.m[I_iv][c] && j >= 0 && j <= V.jj + 2 && i >= 0 && i <= V.ii + 1) {
    const double q = delt1 * .5 * (difiso[s] + difiso[c]) * V.f[F_scvx][c] * V.f[F_scvyi][c] *
                     fmax2(fmin2(dp[s], dp[c]), DPEPS);
    const double fs = q * (saln[s] - saln[c]);
    const double ft = q * (temp[s] - temp[c]);
    V.f[F_vsflld][c + okm] = fs;
    V.f[F_vtflld][c + okm] = ft;
    for (int nt = 0; nt < V.ntr; nt++) {
      if (trc_skip_dif(V.P, nt + 1)) continue;
      const double *tr = V.f[F_trc] + okn + (size_t)nt * 2 * V.kk * np;
      WK(V, 2 * nt + 1)[c + ok] = q * (tr[s] - tr[c]);
    }
    V.f[F_vsflx][c + okm] = V.f[F_vsflx][c + okm] + fs;
    V.f[F_vtflx][c + okm] = V.f[F_vtflx][c + okm] + ft;
  } else if (j >= 0 && j <= V.jj + 2 && i >= 0 && i <= V.ii + 1) {
    for (int nt = 0; nt < V.ntr; nt++) WK(V, 2 * nt + 1)[c + ok] = 0.;
  }
}

__global__ void k_diffus_update(const DevView *__restrict__ Vp, int mm, int nn) {
  const DevView &V = *Vp;
  unsigned bx_, by_;
  xcd_block(bx_, by_);
  const int t = bx_ * blockDim.x + threadIdx.x;
  if (t >= V.nplane) return;
  const int i = t % V.ni - (NBDY - 1), j = t / V.ni - (NBDY - 1);
  if (j < 0 || j > V.jj + 1 || i < 0 || i > V.ii + 1 || !V.m[I_ip][t]) return;
  const int k = by_;
  const size_t np = V.nplane, c = t, e = c + 1, nb = c + V.ni;
  const size_t okn = (size_t)(k + nn) * np, okm = (size_t)(k + mm) * np, ok = (size_t)k * np;
  const double q = 1. / (V.f[F_scp2][c] * fmax2(V.f[F_dp][c + okn], DPEPS));
  const double *usflld = V.f[F_usflld] + okm, *vsflld = V.f[F_vsflld] + okm;
  const double *utflld = V.f[F_utflld] + okm, *vtflld = V.f[F_vtflld] + okm;
  const double sn = V.f[F_saln][c + okn] - q * (usflld[e] - usflld[c] + vsflld[nb] - vsflld[c]);
  const double tn = V.f[F_temp][c + okn] - q * (utflld[e] - utflld[c] + vtflld[nb] - vtflld[c]);
  V.f[F_saln][c + okn] = sn;
  V.f[F_temp][c + okn] = tn;
  for (int nt = 0; nt < V.ntr; nt++) {
    if (trc_skip_dif(V.P, nt + 1)) continue;
    double *tr = V.f[F_trc] + okn + (size_t)nt * 2 * V.kk * np;
    const double *fu = WK(V, 2 * nt) + ok, *fv = WK(V, 2 * nt + 1) + ok;
    tr[c] = tr[c] - q * (fu[e] - fu[c] + fv[nb] - fv[c]);
  }
  V.f[F_sigma][c + okn] = eos::sig(V.P, tn, sn);
}

int st_diffus(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)n; (void)k1m;
  const DevView &h = c->h;
  const size_t np = h.nplane;
  if (h.P.ltedtp_opt != 1) return ctx_fail(c, "diffus: ltedtp = 'neutral' (hybrid coordinate) is not built yet");
  if (int rc = st_xctilr(c, h.f[F_dp] + (size_t)(k1n - 1) * np, 1, h.kk, 3, 3, 1)) return rc;        // :58
  {                                                                                                 // :72-80
    double *ptrs[10] = {h.f[F_temp] + (size_t)(k1n - 1) * np, h.f[F_saln] + (size_t)(k1n - 1) * np};
    int nl[10] = {h.kk, h.kk}, it[10] = {1, 1};
    const int ntr = h.ntr < 8 ? h.ntr : 8;
    int nf = 2;
    for (int nt = 0; nt < ntr; nt++) {
      if (trc_skip_dif(h.P, nt + 1)) continue;               // :76-78: no halo either
      ptrs[nf] = h.f[F_trc] + ((size_t)(k1n - 1) + (size_t)nt * 2 * h.kk) * np;
      nl[nf] = h.kk;
      it[nf++] = 1;
    }
    if (int rc = st_xctilr_multi(c, nf, ptrs, nl, 2, 2, it)) return rc;
    for (int nt = ntr; nt < h.ntr; nt++)
      if (int rc = trc_skip_dif(h.P, nt + 1) ? 0 : st_xctilr(c, h.f[F_trc] + ((size_t)(k1n - 1) + (size_t)nt * 2 * h.kk) * np, 1, h.kk, 2, 2, 1)) return rc;
  }
  // The reference's per-layer flux work arrays uflxtr/vflxtr keep stale values where no
  // u/v point exists; ours are per-layer planes of wk0.. that must start from the same
  // stale content (0 after inivar_tracers, trc/mod_tracers.F90:166-209).
  {
    TimeScope ts(c, "diffus");
    if (c->diffus_shfl) hipLaunchKernelGGL(k_diffus_flux<true>, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, mm, nn);
    else hipLaunchKernelGGL(k_diffus_flux<false>, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, mm, nn);
    hipLaunchKernelGGL(k_diffus_update, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, mm, nn);
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}
