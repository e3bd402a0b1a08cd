// momtum -- baroclinic momentum equation, phy/mod_momtum.F90:215-1282: the column kernels in front of the layer loop.
//   k_mom_pupv, k_mom_pscan    p,pu,pv from dp,dpu,dpv at the mid level               (:245-254,:322-338)
//   k_mom_drag                 bottom drag coefficient and friction velocity (column) (:260-292)
// The layer loop itself (:342-1144) and the vertical pass (:1153-1267) are the row-marching fused kernels of
// stage_momtum_fused.hip.
// "first/last point of a wet segment" logic (ifu/ilu/jfu/jlu lists, phy/mod_bigrid.F90:320-429)
// is expressed with the masks: i is a segment start iff iu(i,j)=1 and iu(i-1,j)=0, etc.; where
// several sweeps of the reference write the same q-point the last writer in its order wins.
// Roofline: HBM.
#include "blomgpu_internal.h"

#include "momtum_common.h"

#define THREAD_IJ(V)                                                       \
  unsigned bx_, by_;                                                       \
  xcd_block(bx_, by_);                                                     \
  const int t_ = bx_ * blockDim.x + threadIdx.x;                           \
  (void)by_;                                                               \
  if (t_ >= (V).nplane) return;                                            \
  const int i = t_ % (V).ni - (NBDY - 1), j = t_ / (V).ni - (NBDY - 1);    \
  const size_t c = t_

#define S2_DRAG 3      // 2-D work plane
#define S2_QUM 4       // four consecutive 2-D work planes: the barotropic part of utotm, vtotm, utotn, vtotn


// pu(k+1) = pu(k) + dpu(k+off), same for pv; range lo..+hi (:322-338 with lo=-1,hi=2; :1252-1267 interior)
__global__ void k_mom_pupv(const DevView *__restrict__ Vp, int off, int lo, int hi) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < lo || j > V.jj + hi || i < lo || i > V.ii + hi) return;
  const size_t np = V.nplane;
  if (V.m[I_iu][c]) column_scan(V.f[F_pu][c], V.f[F_dpu] + (size_t)off * np + c, V.f[F_pu] + c, np, V.kk);
  if (V.m[I_iv][c]) column_scan(V.f[F_pv][c], V.f[F_dpv] + (size_t)off * np + c, V.f[F_pv] + c, np, V.kk);
}

// p(k+1) = p(k) + dp(k+off) for j,i = lo..+hi
__global__ void k_mom_pscan(const DevView *__restrict__ Vp, int off, int lo, int hi) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < lo || j > V.jj + hi || i < lo || i > V.ii + hi || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  column_scan(V.f[F_p][c], V.f[F_dp] + (size_t)off * np + c, V.f[F_p] + c, np, V.kk);
}

// ---- :260-292 bottom drag ------------------------------------------------------------------------
__global__ void k_mom_drag(const DevView *__restrict__ Vp, int n, int nn) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 0 || j > V.jj || i < 0 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane, on = (size_t)(n - 1) * np, e = c + 1, nb = c + V.ni;
  const double thkbop = THKBOT * ONEM, tsfac = V.P.dlt / V.P.delt1;
  const double *p = V.f[F_p];
  const double pbot = p[c + (size_t)V.kk * np];
  double u1 = 0., u2 = 0.;
  for (int k = 0; k < V.kk; k++) {
    const size_t okn = (size_t)(k + nn) * np;
    const double pbotl = fmax2(p[c + (size_t)(k + 1) * np], pbot - thkbop);
    const double ptopl = fmax2(p[c + (size_t)k * np], pbot - thkbop);
    u1 = u1 + (V.f[F_u][c + okn] + V.f[F_u][e + okn]) * (pbotl - ptopl);
    u2 = u2 + (V.f[F_v][c + okn] + V.f[F_v][nb + okn]) * (pbotl - ptopl);
  }
  V.f[F_util1][c] = u1;
  V.f[F_util2][c] = u2;
  const double *ubf = V.f[F_ubflxs_p] + on, *vbf = V.f[F_vbflxs_p] + on, *pbu = V.f[F_pbu] + on, *pbv = V.f[F_pbv] + on;
  const double ubot = (ubf[c] / fmax2(EPSILPL, pbu[c] * V.f[F_scuy][c]) + ubf[e] / fmax2(EPSILPL, pbu[e] * V.f[F_scuy][e])) * tsfac +
                      u1 / thkbop;
  const double vbot = (vbf[c] / fmax2(EPSILPL, pbv[c] * V.f[F_scvx][c]) + vbf[nb] / fmax2(EPSILPL, pbv[nb] * V.f[F_scvx][nb])) * tsfac +
                      u2 / thkbop;
  const double ubbl = .5 * sqrt(ubot * ubot + vbot * vbot);
  const double q = V.P.cb * (ubbl + V.P.cbar);
  WK2(V, S2_DRAG)[c] = q * GRAV / (ALPHA0 * thkbop);
  V.f[F_ustarb][c] = sqrt(q * ubbl);
}


// ---- the barotropic part of the total velocities, utotm - u(km) and utotn - u(kn) (:360-431):
// ubflxs_p*tsfac/(pbu*scuy), the same for every layer -- evaluated once per step into four 2-D work planes
// (u, v at time levels m, n) instead of once per layer and sweep inside the marches (a division each)
__global__ void k_mom_qplanes(const DevView *__restrict__ Vp, int m, int n) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  (void)i; (void)j;
  const size_t np = V.nplane, om = (size_t)(m - 1) * np, on = (size_t)(n - 1) * np;
  const double tsfac = V.P.dlt / V.P.delt1;
  double *q = WK2(V, S2_QUM);
  if (V.m[I_iu][c]) {
    const double sy = V.f[F_scuy][c];
    q[c] = V.f[F_ubflxs_p][c + om] * tsfac / (V.f[F_pbu][c + om] * sy);
    q[c + 2 * np] = V.f[F_ubflxs_p][c + on] * tsfac / (V.f[F_pbu][c + on] * sy);
  }
  if (V.m[I_iv][c]) {
    const double sx = V.f[F_scvx][c];
    q[c + np] = V.f[F_vbflxs_p][c + om] * tsfac / (V.f[F_pbv][c + om] * sx);
    q[c + 3 * np] = V.f[F_vbflxs_p][c + on] * tsfac / (V.f[F_pbv][c + on] * sx);
  }
}

// pu, pv of the range lo..+hi from dpu, dpv at level offset off (mxlayr's 'old' interface pressures, phy/mod_mxlayr.F90:1246-1262)
int st_mom_pupv(blomgpu_ctx *c, int off, int lo, int hi) {
  hipLaunchKernelGGL(k_mom_pupv, plane_grid(c->h, 1, 64), dim3(64), 0, c->stream, c->d, off, lo, hi);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int st_momtum(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)k1m; (void)k1n;
  const DevView &h = c->h;
  const dim3 gcol = plane_grid(h, 1, 64), b64(64);
  TimeScope ts(c, "momtum");
  hipLaunchKernelGGL(k_mom_pscan, gcol, b64, 0, c->stream, c->d, mm, -1, 2);
  hipLaunchKernelGGL(k_mom_drag, gcol, b64, 0, c->stream, c->d, n, nn);
  hipLaunchKernelGGL(k_mom_pupv, gcol, b64, 0, c->stream, c->d, mm, -1, 2);
  hipLaunchKernelGGL(k_mom_qplanes, plane_grid(h, 1, 256), dim3(256), 0, c->stream, c->d, m, n);
  if (int rc = st_xctilr(c, h.f[F_difwgt], 1, 1, 2, 2, 1)) return rc;                       // :340
  return st_momtum_fused_layers(c, m, n, mm, nn);                                           // stage_momtum_fused.hip
}
