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
// ring: only the points of the range that lie outside the tile's interior
__global__ void k_mom_pupv(const DevView *__restrict__ Vp, int off, int lo, int hi, int ring) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < lo || j > V.jj + hi || i < lo || i > V.ii + hi) return;
  if (ring && j >= 1 && j <= V.jj && i >= 1 && i <= V.ii) return;
  const size_t np = V.nplane;
  if (V.m[I_iu][c]) column_scan(V.f[F_pu][c], V.f[F_dpu] + (size_t)off * np + c, V.f[F_pu] + c, np, V.kk);
  if (V.m[I_iv][c]) column_scan(V.f[F_pv][c], V.f[F_dpv] + (size_t)off * np + c, V.f[F_pv] + c, np, V.kk);
}

// p(k+1) = p(k) + dp(k+off) for j,i = lo..+hi
__global__ void k_mom_pscan(const DevView *__restrict__ Vp, int off, int lo, int hi, int ring) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < lo || j > V.jj + hi || i < lo || i > V.ii + hi || !V.m[I_ip][c]) return;
  if (ring && j >= 1 && j <= V.jj && i >= 1 && i <= V.ii) return;
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
  gcd_t p = V.f[F_p];
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
  gcd_t ubf = V.f[F_ubflxs_p] + on, vbf = V.f[F_vbflxs_p] + on, pbu = V.f[F_pbu] + on, pbv = V.f[F_pbv] + on;
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
  gd_t q = WK2(V, S2_QUM);
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
  hipLaunchKernelGGL(k_mom_pupv, plane_grid(c->h, 1, 64), dim3(64), 0, c->stream, c->d, off, lo, hi, 0);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// The stage's kernels see momtum's own copies of p, pu, pv and of the work space (the alternative views of
// blomgpu_internal.h), so that inside blomgpu_step the viscous chain can run ahead of the stage, beside the stages that
// precede it, and convec's column kernel beside the Coriolis march (blomgpu_ctx::overlap).  Only the vertical pass
// writes pu, pv into the module arrays: they are what the reference leaves there (:1252-1267); p is left as pgforc
// wrote it, the reference's momtum does not change it either (its p is recomputed by every stage that reads it).
static int momtum_visc_chain(blomgpu_ctx *c, int m, int n, int mm, int nn) {
  const DevView &h = c->h;
  hipLaunchKernelGGL(k_mom_pupv, plane_grid(h, 1, 64), dim3(64), 0, c->stream, ctx_view(c, VIEW_MOM_A), mm, -1, 2, 0);
  hipLaunchKernelGGL(k_mom_qplanes, plane_grid(h, 1, 256), dim3(256), 0, c->stream, ctx_view(c, VIEW_MOM_A), m, n);
  return st_momtum_fused_layers(c, m, n, mm, nn, 1);
}

// in sequence, right after difest's halo updates of u, v, ubflxs_p, pbu (phy/mod_difest.F90:750-760): the viscous chain
// reads u, v (n), dpu, dpv (m), pbu, pbv, ubflxs_p, difwgt and the grid -- nothing that eddtra, advect, pbcor1, diffus or
// pgforc write -- and writes momtum's own planes and utotn, vtotn (read by barotp only)
int st_momtum_early(blomgpu_ctx *c, int m, int n, int mm, int nn) {
  c->mom_early_done = false;
  if (!ctx_overlap_on(c) || c->tiling.multi()) return 0;
  if (int rc = st_xctilr(c, c->h.f[F_difwgt], 1, 1, 2, 2, 1)) return rc;                    // :340 (difwgt is not written in between)
  if (int rc = ctx_side_fork(c, 0)) return rc;
  hipStream_t main = c->stream;
  c->stream = c->side;
  const int rc = momtum_visc_chain(c, m, n, mm, nn);
  c->stream = main;
  if (rc) return rc;
  if (int rc2 = ctx_side_done(c, 1)) return rc2;
  c->mom_early_done = true;
  return 0;
}

int st_momtum(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)k1m; (void)k1n;
  const DevView &h = c->h;
  const dim3 gcol = plane_grid(h, 1, 64), b64(64);
  // convec's column kernel (phy/mod_convec.F90:95-302) reads and writes dp, T, S, sigma, tracers of level n, kfpla and p;
  // momtum reads none of them but p, of which it has its own copy: the kernel runs beside momtum on the second stream
  if (ctx_overlap_on(c) && !c->tiling.multi())
    if (int rc = st_convec_column_ahead(c, n, nn)) return rc;
  TimeScope ts(c, "momtum");
  hipLaunchKernelGGL(k_mom_pscan, gcol, b64, 0, c->stream, ctx_view(c, VIEW_MOM_B), mm, -1, 2, 0);
  // the module arrays p, pu, pv as the reference's momtum leaves them: p (:245-254) and, outside the interior, pu, pv (:322-338; inside
  // it the vertical pass rewrites them, :1252-1267).  Inside blomgpu_step's isopycnic sequence only the ring of p outside the interior:
  // convec's column kernel, the next writer of the interior's p, may already be running on the second stream
  hipLaunchKernelGGL(k_mom_pscan, gcol, b64, 0, c->stream, c->d, mm, -1, 2, c->in_sequence && h.P.vcoord_tag == 1 ? 1 : 0);
  hipLaunchKernelGGL(k_mom_pupv, gcol, b64, 0, c->stream, c->d, mm, -1, 2, 1);
  hipLaunchKernelGGL(k_mom_drag, gcol, b64, 0, c->stream, ctx_view(c, VIEW_MOM_B), n, nn);
  if (c->mom_early_done) {
    c->mom_early_done = false;
    if (int rc = ctx_side_join(c, 1)) return rc;
  } else {
    if (int rc = st_xctilr(c, h.f[F_difwgt], 1, 1, 2, 2, 1)) return rc;                     // :340
    if (int rc = momtum_visc_chain(c, m, n, mm, nn)) return rc;
  }
  return st_momtum_fused_layers(c, m, n, mm, nn, 2);                                        // stage_momtum_fused.hip
}
