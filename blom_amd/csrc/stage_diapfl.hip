// diapfl -- diapycnal mixing (isopycnic bulk-mixed-layer branch), phy/mod_diapfl.F90:49-1046.
//
// A per-water-column algorithm: mixed-layer/interior mass fluxes, density-restoring flux
// corrections with an iterative limiter (:296-330), an implicit nonlinear diffusion solve by
// alternating downward/upward tridiagonal sweeps (:359-533), implicit T/S/tracer mixing
// (:546-576), then per u-/v-column implicit momentum mixing (:740-966) and dpu/dpv (:971-1000).
// Mapping: one thread per column, columns of a wavefront adjacent in i.  The p-column pass is k_diapfl_column3
// (stage_diapfl_col3.hip); here: kming, the momentum mixing of the u-/v-columns, dpu/dpv and the stage driver.
// The trip counts of the two iteration loops are data dependent; lanes that
// converged early idle until their wavefront neighbours finish (hard part 7).
// Algorithmic bytes: (23 + 2*ntr) F; roofline: HBM.
#include "blomgpu_internal.h"
#include "eos.h"

#define GRAV 9.806
#define ALPHA0 1.e-3
#define EPSILP 1.e-12
#define ONEM 9806.

#define THREAD_IJ(V)                                                       \
  unsigned bx_, by_;                                                       \
  xcd_block(bx_, by_);                                                     \
  const int t_ = bx_ * blockDim.x + threadIdx.x;                           \
  (void)by_;                                                               \
  if (t_ >= (V).nplane) return;                                            \
  const int i = t_ % (V).ni - (NBDY - 1), j = t_ / (V).ni - (NBDY - 1);    \
  const size_t c = t_

// 1-based layer index k, as in the Fortran
#define AR(slot, k) WK(V, slot)[c + (size_t)((k)-1) * np]

#include "diapfl_common.h"

__global__ void k_diapfl_kming(const DevView *__restrict__ Vp) {                                // :725-733
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 0 || j > V.jj + 1 || i < 0 || i > V.ii + 1 || !V.m[I_ip][c]) return;
  V.m[I_kming][c] = (int)rint(V.f[F_util1][c]);
}

// momentum mixing of one component (blockIdx.y = 0: u, 1: v), :740-852 / :856-966.
// As in the column pass, the reference's work arrays uc, delp (the mixed layer moved to positions
// kmin, kmin+1) are the velocity and thickness arrays themselves under the position -> layer map
// (kmin -> 1, kmin+1 -> 2), the interface fluxes fpu/fpl are evaluated where the tridiagonal sweep
// consumes them, and the forward-eliminated velocity is kept in place; only gtd needs a work plane.
enum { U_GTD = 0, U_NSLOT };

// fluxes through the upper interface of position k (fpu(k)) and, same interface seen from above,
// fpl(k-1): the average of the two neighbouring p-columns, clipped at the velocity-point bottom (:779-817)
// the same from the six values it reads: p, fpug of level k and fplg of level k-1 at the two p-columns (m: mns, c)
__device__ inline void diapfl_mom_flux_v(double pm, double fum, double flm, double pc, double fuc, double flc, double pzb,
                                         double &fpu_k, double &fpl_km1) {
  double fpum, fplm, fpup, fplp;
  double pnew = pm, fu = fum, fl = flm;
  double pold = pnew - fl + fu;
  if (pold <= pzb) {
    if (pnew <= pzb) { fpum = fu; fplm = fl; } else { fpum = fu; fplm = fl - pnew + pzb; }
  } else {
    if (pnew <= pzb) { fpum = fu - pold + pzb; fplm = fl; } else { fpum = .5 * (fu + fl); fplm = fpum; }
  }
  pnew = pc; fu = fuc; fl = flc;
  pold = pnew - fl + fu;
  if (pold <= pzb) {
    if (pnew <= pzb) { fpup = fu; fplp = fl; } else { fpup = fu; fplp = fl - pnew + pzb; }
  } else {
    if (pnew <= pzb) { fpup = fu - pold + pzb; fplp = fl; } else { fpup = .5 * (fu + fl); fplp = fpup; }
  }
  fpu_k = .5 * (fpum + fpup);
  fpl_km1 = .5 * (fplm + fplp);
}
__device__ inline void diapfl_mom_flux(const double *p, const double *fpug, const double *fplg, size_t mns, size_t c, size_t np,
                                       int k, double pzb, double &fpu_k, double &fpl_km1) {
  double fpum, fplm, fpup, fplp;
  const size_t o = (size_t)(k - 1) * np, om = (size_t)(k - 2) * np;
  double pnew = p[mns + o], fu = fpug[mns + o], fl = fplg[mns + om];
  double pold = pnew - fl + fu;
  if (pold <= pzb) {
    if (pnew <= pzb) { fpum = fu; fplm = fl; } else { fpum = fu; fplm = fl - pnew + pzb; }
  } else {
    if (pnew <= pzb) { fpum = fu - pold + pzb; fplm = fl; } else { fpum = .5 * (fu + fl); fplm = fpum; }
  }
  pnew = p[c + o]; fu = fpug[c + o]; fl = fplg[c + om];
  pold = pnew - fl + fu;
  if (pold <= pzb) {
    if (pnew <= pzb) { fpup = fu; fplp = fl; } else { fpup = fu; fplp = fl - pnew + pzb; }
  } else {
    if (pnew <= pzb) { fpup = fu - pold + pzb; fplp = fl; } else { fpup = .5 * (fu + fl); fplp = fpup; }
  }
  fpu_k = .5 * (fpum + fpup);
  fpl_km1 = .5 * (fplm + fplp);
}

__global__ void k_diapfl_momentum(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const bool isv = by_ == 1;
  if (!(isv ? V.m[I_iv][c] : V.m[I_iu][c])) return;
  const size_t np = V.nplane, mns = isv ? c - V.ni : c - 1;
  const int kk = V.kk, sb = isv ? U_NSLOT : 0;
  gd_t vel = (isv ? V.f[F_v] : V.f[F_u]) + (size_t)nn * np;
  gcd_t dpz = (isv ? V.f[F_dpv] : V.f[F_dpu]) + (size_t)nn * np;
  gcd_t p = V.f[F_p], fpug = V.f[F_fpug], fplg = V.f[F_fplg];
#define LV(a, k) (a)[c + (size_t)((k)-1) * np]
  const int kmin = min(V.m[I_kming][mns], V.m[I_kming][c]);
  int kmax = 1;
#define MU_ 4       /* levels whose loads a sweep keeps in flight */
  for (int k0 = 2; k0 <= kk; k0 += 2 * MU_) {
    double a0[2 * MU_];
#pragma unroll
    for (int u = 0; u < 2 * MU_; u++) a0[u] = LV(dpz, k0 + u <= kk ? k0 + u : kk);
#pragma unroll
    for (int u = 0; u < 2 * MU_; u++)
      if (k0 + u <= kk && a0[u] > 0.) kmax = k0 + u;
  }
  if (!(kmin < kmax)) return;
  const double pzb = (isv ? V.f[F_pv] : V.f[F_pu])[c + (size_t)kk * np];
  // forward elimination over positions kmin..kmax, :822-836; position -> layer: kmin -> 1, kmin+1 -> 2
  double ctd = 0., bitd = 1., g = 0., uprev = 0.;
  double fu = 0., fl_prev_unused = 0.;                        // fpu(kmin) = 0
  (void)fl_prev_unused;
  double fu_next = 0., fl;
  for (int k0 = kmin; k0 <= kmax; k0 += MU_) {
   double b0[MU_], b1[MU_], b2[MU_], b3[MU_], b4[MU_], b5[MU_], b6[MU_], b7[MU_];
#pragma unroll
   for (int u = 0; u < MU_; u++) {
     const int kq = k0 + u <= kmax ? k0 + u : kmax, kn = kq + 1 <= kk ? kq + 1 : kk;
     const int lq = kq == kmin ? 1 : (kq == kmin + 1 ? 2 : kq);
     const size_t o = (size_t)(kn - 1) * np, om = (size_t)(kn - 2) * np;
     b0[u] = p[mns + o]; b1[u] = fpug[mns + o]; b2[u] = fplg[mns + om];
     b3[u] = p[c + o]; b4[u] = fpug[c + o]; b5[u] = fplg[c + om];
     b6[u] = LV(dpz, lq); b7[u] = LV(vel, lq);
   }
#pragma unroll
   for (int u = 0; u < MU_; u++) {
    const int k = k0 + u;
    if (k > kmax) break;
    const int lay = k == kmin ? 1 : (k == kmin + 1 ? 2 : k);
    if (k < kmax) diapfl_mom_flux_v(b0[u], b1[u], b2[u], b3[u], b4[u], b5[u], pzb, fu_next, fl);    // fpu(k+1), fpl(k)
    else fl = 0.;                                                                         // fpl(kmax) = 0
    g = ctd * bitd;
    AR(sb + U_GTD, k) = g;
    const double dk = b6[u];
    const double q = 1. / (dk + fu + fl);
    const double atd = -fu * q;
    ctd = -fl * q;
    const double dtd = dk * q;
    bitd = 1. / (1. - atd * g);
    const double uk = b7[u];
    // km1 = max(k-1, kmin): at k = kmin the reference reads uc(kmin) itself, i.e. the unmodified value
    uprev = (dtd * uk - atd * (k == kmin ? uk : uprev)) * bitd;
    LV(vel, lay) = uprev;
    fu = fu_next;
   }
  }
  // back substitution, :838
  double unext = uprev, gnext = g;
  for (int k0 = kmax - 1; k0 >= kmin; k0 -= 2 * MU_) {
    double b0[2 * MU_], b1[2 * MU_];
#pragma unroll
    for (int u = 0; u < 2 * MU_; u++) {
      const int kq = k0 - u >= kmin ? k0 - u : kmin;
      const int lq = kq == kmin ? 1 : (kq == kmin + 1 ? 2 : kq);
      b0[u] = LV(vel, lq); b1[u] = AR(sb + U_GTD, kq);
    }
#pragma unroll
    for (int u = 0; u < 2 * MU_; u++) {
      const int k = k0 - u;
      if (k < kmin) break;
      const int lay = k == kmin ? 1 : (k == kmin + 1 ? 2 : k);
      unext = b0[u] - gnext * unext;
      LV(vel, lay) = unext;
      gnext = b1[u];
    }
  }
  const double ub = uprev;                                    // uc(kmax): untouched by the back substitution
  for (int k0 = kmax + 1; k0 <= kk; k0 += 2 * MU_) {           // (the massless layers under the column: their loads in flight too)
    double b0[2 * MU_], b1[2 * MU_];
#pragma unroll
    for (int u = 0; u < 2 * MU_; u++) {
      const int kq = k0 + u <= kk ? k0 + u : kk;
      b0[u] = p[mns + (size_t)(kq - 1) * np]; b1[u] = LV(p, kq);
    }
#pragma unroll
    for (int u = 0; u < 2 * MU_; u++)
      if (k0 + u <= kk && fmin2(b0[u], b1[u]) < pzb) LV(vel, k0 + u) = ub;
  }
}

// dpu/dpv at the new level from the updated p, :971-1000 (u: i 1..ii+1, j 1..jj ; v: i 1..ii, j 1..jj+1)
__global__ void k_diapfl_dpudpv(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  const int k = by_;
  const size_t np = V.nplane, o0 = (size_t)k * np, o1 = (size_t)(k + 1) * np, ob = (size_t)V.kk * np;
  gcd_t p = V.f[F_p];
  if (V.m[I_iu][c] && j >= 1 && j <= V.jj && i >= 1 && i <= V.ii + 1) {
    const size_t w = c - 1;
    const double q = fmin2(p[c + ob], p[w + ob]);
    V.f[F_dpu][c + (size_t)(k + nn) * np] = .5 * ((fmin2(q, p[w + o1]) - fmin2(q, p[w + o0])) + (fmin2(q, p[c + o1]) - fmin2(q, p[c + o0])));
  }
  if (V.m[I_iv][c] && j >= 1 && j <= V.jj + 1 && i >= 1 && i <= V.ii) {
    const size_t s = c - V.ni;
    const double q = fmin2(p[c + ob], p[s + ob]);
    V.f[F_dpv][c + (size_t)(k + nn) * np] = .5 * ((fmin2(q, p[s + o1]) - fmin2(q, p[s + o0])) + (fmin2(q, p[c + o1]) - fmin2(q, p[c + o0])));
  }
}

int st_diapfl(blomgpu_ctx *c, int n, int nn, int k1n) {
  (void)k1n;
  const DevView &h = c->h;
  if (h.P.vcoord_tag != 1) return ctx_fail(c, "diapfl is only called for isopyc_bulkml (phy/mod_blom_step.F90:172-186)");
  if (int rc = ctx_err_words(c)) return rc;
  int *errflag = c->err_dev + 0;
  {
    TimeScope ts(c, "diapfl");
    if (int rc = diapfl_column3_launch(c, n, nn, errflag)) return rc;                     // stage_diapfl_col3.hip
    {                                                                                     // :711-713, :723
      double *ptrs[4] = {h.f[F_p], h.f[F_fpug], h.f[F_fplg], h.f[F_util1]};
      const int nl[4] = {h.kk + 1, h.kk, h.kk, 1}, it[4] = {1, 1, 1, 1};
      if (int rc = st_xctilr_multi(c, 4, ptrs, nl, 1, 1, it)) return rc;
    }
    hipLaunchKernelGGL(k_diapfl_kming, plane_grid(h), dim3(256), 0, c->stream, c->d);
    // Inside blomgpu_step's full step (phys_dag) the momentum mixing and the new dpu, dpv run on the second stream: they read p, pu, pv,
    // dpu, dpv, the fluxes fpug / fplg and kming, and write u, v, dpu, dpv of level n -- thermf_channel (T, S, dp, forcing fields, util1-4: after
    // k_diapfl_kming, which reads util1) and mxlayr's first kernels (util1-3) touch none of them and are a dozen launches of ~6 us each.
    // st_mxlayr waits in front of its halo updates of u, v.
    const bool aside = ctx_overlap_on(c) && (c->phys_dag & 2) && !c->tiling.multi() && c->full_physics;
    hipStream_t s = c->stream;
    if (aside) {
      if (int rc = ctx_side_fork(c, 6)) return rc;
      s = c->side;
    }
    hipLaunchKernelGGL(k_diapfl_momentum, plane_grid(h, 2, 64), dim3(64), 0, s, c->d, nn);
    // inside blomgpu_step the stage that follows (mxlayr's tail, phy/mod_mxlayr.F90:1266-1310) recomputes dpu, dpv of this
    // level from the same p with the same expression over a larger range before anything reads them: skipped there
    // (with full_physics the real mxlayr follows, which reads them first)
    if (!c->in_sequence || c->full_physics) hipLaunchKernelGGL(k_diapfl_dpudpv, plane_grid(h, h.kk), dim3(256), 0, s, c->d, nn);
    if (aside) {
      if (int rc = ctx_side_done(c, 7)) return rc;
      c->diapfl_mom_on_side = true;
    }
  }
  HIPCHK(c, hipGetLastError());
  // the reference aborts (xchalt) when the implicit solve does not converge, :520-530
  if (!c->defer_checks) return ctx_check_errors(c);
  return 0;
}
