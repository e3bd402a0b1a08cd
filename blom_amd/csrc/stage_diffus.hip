// diffus -- layer-wise lateral diffusion of T, S and tracers, phy/mod_diffus.F90:41-185.
//
// One LDS-tiled kernel over (i,j,k) (k on blockIdx.y).  The reference's three sweeps per layer -- u-face fluxes
// q*(X(i-1)-X(i)), v-face fluxes, divergence update of T, S, tracers and sigma = sig(T,S) (:89-160) -- cannot be fused in
// place (a neighbour's flux reads the old value), so the stage READS its scalars from one storage and WRITES them to
// another:
//   source       S, T and the diffused tracers at the new time level in the kk-level work space (slots N_S, N_T, N_TR(nt));
//   destination  the fields saln, temp, trc, sigma at level kn.
// pbcor1, the stage in front of diffus (phy/mod_blom_step.F90:126-131), ends with exactly that: its LDS-tiled kernel
// leaves the new dp, S, T, tracers in those work-space slots (stage_pbcor_tile.hip).  Inside blomgpu_step it therefore only
// rescales dp into its field and hands S, T and the tracers over where they are; the halo updates of :58-80 then act on
// the work-space planes.  Called on its own (blomgpu_diffus), the stage first copies the fields into the slots.
// A workgroup owns a tile of DT_TW x DT_TH cells of one layer:
//   phase 0  dp, difiso, S, T of the tile with a one-cell rim -> LDS, every load independent and issued up front;
//   phase 1  each thread evaluates the diffusive weights q of the four faces of its cell (a face is evaluated by the two
//            cells it separates, with the same operands), stores / accumulates the S and T fluxes of its west and south
//            face (utflld.., utflx..: the reference's loop ranges), updates its cell;
//   phase 2  the diffused tracers in batches of DT_TB through the same LDS slots, weights kept in registers.
// Besides: cells of the second halo ring take the halo-updated value (the reference's field holds it after :72-80),
// tracers left out of diffusion (:64-66) that pbcor1 left in the work space move into their field.
// Algorithmic bytes: (19 + 2*ntr_dif) F.  Roofline: HBM.
#include "blomgpu_internal.h"
#include "eos.h"

#define DPEPS 1.e-5   // phy/mod_diffus.F90:55-56

#define DT_TW 32
#define DT_TH 8
#define DT_LW (DT_TW + 2)
#define DT_LN (DT_LW * (DT_TH + 2))
#define DT_NT (DT_TW * DT_TH)
#define DT_TB 4                      // tracers per batch
// work-space slots shared with pbcor's tile kernel (stage_pbcor_tile.hip)
#define N_S 1
#define N_T 2
#define N_TR(nt) (3 + (nt))

#define MP(m) ((m) & 1)
#define MU(m) (((m) >> 1) & 1)
#define MV(m) (((m) >> 2) & 1)

// move_mask: tracers (bit nt) that are NOT diffused but wait in the work space (pbcor1 left them there): interior copy
__global__ void __launch_bounds__(DT_NT) k_diffus_tile(const DevView *__restrict__ Vp, int mm, int nn, int ntx, unsigned long long move_mask) {
  const DevView &V = *Vp;
  __shared__ double sc[(4 + DT_TB) * DT_LN];                  // dp, difiso, S, T, tracer batch
  unsigned bx_, by_;
  xcd_block(bx_, by_);
  const int k = by_, ni = V.ni, nj = V.nj, ntr = V.ntr, t = threadIdx.x;
  const int x0 = (bx_ % ntx) * DT_TW, y0 = (bx_ / ntx) * DT_TH;
  const size_t np = V.nplane, ok = (size_t)k * np, okn = (size_t)(k + nn) * np, okm = (size_t)(k + mm) * np;
  gcd_t f_dp = V.f[F_dp] + okn, f_di = V.f[F_difiso] + ok, f_s = WK(V, N_S) + ok, f_t = WK(V, N_T) + ok;

  // ---- phase 0 ---------------------------------------------------------------------------------------------------
  double sv[2][4];
  size_t csr[2];
  bool inr[2];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int m = t + r * DT_NT;
    const int xs = x0 - 1 + m % DT_LW, ys = y0 - 1 + m / DT_LW;
    inr[r] = m < DT_LN && xs >= 0 && xs < ni && ys >= 0 && ys < nj;
    csr[r] = inr[r] ? (size_t)ys * ni + xs : 0;
    sv[r][0] = f_dp[csr[r]]; sv[r][1] = f_di[csr[r]]; sv[r][2] = f_s[csr[r]]; sv[r][3] = f_t[csr[r]];
  }
  const int lx = t % DT_TW, ly = t / DT_TW;
  const int x = x0 + lx, y = y0 + ly;
  const int i = x - (NBDY - 1), j = y - (NBDY - 1);
  // cells with a face or an update: -1 <= i <= ii+2, -1 <= j <= jj+2 (ring 2 only takes the halo value)
  const bool live = x < ni && y < nj && j >= -1 && j <= V.jj + 2 && i >= -1 && i <= V.ii + 2;
  const size_t c = live ? (size_t)y * ni + x : (size_t)ni + 1, e = c + 1, nb = c + ni;
  const int q = (ly + 1) * DT_LW + lx + 1;
  gci_t mpk = V.m[I_mpack];
  const int mp_c = live ? mpk[c] : 0, mp_e = live ? mpk[e] : 0, mp_n = live ? mpk[nb] : 0;
  // the reference's loop ranges: u-faces j = 0..jj+1, i = 0..ii+2; v-faces j = 0..jj+2, i = 0..ii+1; cells 0..ii+1, 0..jj+1
  const bool inner = live && i >= 0 && j >= 0;
  const bool uw = inner && j <= V.jj + 1 && MU(mp_c), vs = inner && i <= V.ii + 1 && MV(mp_c);
  const bool upd = inner && i <= V.ii + 1 && j <= V.jj + 1 && MP(mp_c);
  const bool ue = upd && MU(mp_e), vn = upd && MV(mp_n);
  const bool ring2 = live && !(inner && i <= V.ii + 1 && j <= V.jj + 1) && MP(mp_c);
  const bool interior = upd && i >= 1 && i <= V.ii && j >= 1 && j <= V.jj;
  const double delt1 = V.P.delt1;
  const double yc = V.f[F_scuy][c], xic = V.f[F_scuxi][c], ye = V.f[F_scuy][e], xie = V.f[F_scuxi][e];
  const double vxc = V.f[F_scvx][c], vyic = V.f[F_scvyi][c], vxn = V.f[F_scvx][nb], vyin = V.f[F_scvyi][nb];
  const double s2 = V.f[F_scp2][c];
  gd_t o_us = V.f[F_usflx] + c + okm, o_ut = V.f[F_utflx] + c + okm, o_vs = V.f[F_vsflx] + c + okm, o_vt = V.f[F_vtflx] + c + okm;
  const double us_o = *o_us, ut_o = *o_ut, vs_o = *o_vs, vt_o = *o_vt;
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int m = t + r * DT_NT;
    if (m < DT_LN) {
#pragma unroll
      for (int s = 0; s < 4; s++) sc[s * DT_LN + m] = sv[r][s];
    }
  }
  __syncthreads();

  // ---- phase 1 ---------------------------------------------------------------------------------------------------
  const double *l_dp = sc, *l_di = sc + DT_LN, *l_s = sc + 2 * DT_LN, *l_t = sc + 3 * DT_LN;
  // :92-94, :114-116 -- the weight of a face from the two cells it separates
  const double qw = uw ? delt1 * .5 * (l_di[q - 1] + l_di[q]) * yc * xic * fmax2(fmin2(l_dp[q - 1], l_dp[q]), DPEPS) : 0.;
  const double qs = vs ? delt1 * .5 * (l_di[q - DT_LW] + l_di[q]) * vxc * vyic * fmax2(fmin2(l_dp[q - DT_LW], l_dp[q]), DPEPS) : 0.;
  const double qe = ue ? delt1 * .5 * (l_di[q] + l_di[q + 1]) * ye * xie * fmax2(fmin2(l_dp[q], l_dp[q + 1]), DPEPS) : 0.;
  const double qn = vn ? delt1 * .5 * (l_di[q] + l_di[q + DT_LW]) * vxn * vyin * fmax2(fmin2(l_dp[q], l_dp[q + DT_LW]), DPEPS) : 0.;
  const double rq = upd ? 1. / (s2 * fmax2(l_dp[q], DPEPS)) : 0.;                       // :139
  {
    // faces without a u-/v-point keep the 0 the flux arrays hold there (phy/mod_diffusion.F90:452-500)
    const double fsw = uw ? qw * (l_s[q - 1] - l_s[q]) : 0., ftw = uw ? qw * (l_t[q - 1] - l_t[q]) : 0.;
    const double fss = vs ? qs * (l_s[q - DT_LW] - l_s[q]) : 0., fts = vs ? qs * (l_t[q - DT_LW] - l_t[q]) : 0.;
    if (uw) {                                                // :95-96, :105-106
      V.f[F_usflld][c + okm] = fsw;
      V.f[F_utflld][c + okm] = ftw;
      *o_us = us_o + fsw;
      *o_ut = ut_o + ftw;
    }
    if (vs) {                                                // :117-118, :127-128
      V.f[F_vsflld][c + okm] = fss;
      V.f[F_vtflld][c + okm] = fts;
      *o_vs = vs_o + fss;
      *o_vt = vt_o + fts;
    }
    if (upd) {                                               // :140-145, :157
      const double fse = ue ? qe * (l_s[q] - l_s[q + 1]) : 0., fte = ue ? qe * (l_t[q] - l_t[q + 1]) : 0.;
      const double fsn = vn ? qn * (l_s[q] - l_s[q + DT_LW]) : 0., ftn = vn ? qn * (l_t[q] - l_t[q + DT_LW]) : 0.;
      const double sn = l_s[q] - rq * (fse - fsw + fsn - fss);
      const double tn = l_t[q] - rq * (fte - ftw + ftn - fts);
      V.f[F_saln][c + okn] = sn;
      V.f[F_temp][c + okn] = tn;
      V.f[F_sigma][c + okn] = eos::sig(V.P, tn, sn);
    } else if (ring2) {
      V.f[F_saln][c + okn] = l_s[q];
      V.f[F_temp][c + okn] = l_t[q];
    }
  }

  // ---- phase 2: tracers ----------------------------------------------------------------------------------------------
  // which tracers of a batch are diffused (uniform over the workgroup); a batch's values are loaded while the batch before it
  // is worked on (tv is free again once it has gone to LDS)
  bool dif[DT_TB], any = false;
  double tv[2][DT_TB];
  auto look = [&](int nt0, bool *d) {
    bool a = false;
#pragma unroll
    for (int b = 0; b < DT_TB; b++) { d[b] = nt0 + b < ntr && !trc_skip_dif(V.P, nt0 + b + 1); a = a || d[b]; }
    return a;
  };
  auto fetch = [&](int nt0, const bool *d) {
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int b = 0; b < DT_TB; b++) tv[r][b] = d[b] ? WK(V, N_TR(nt0 + b))[csr[r] + ok] : 0.;
  };
  any = look(0, dif);
  if (any) fetch(0, dif);
  for (int nt0 = 0; nt0 < ntr; nt0 += DT_TB) {
    if (any) {
      __syncthreads();                                       // the previous batch has been read
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const int m = t + r * DT_NT;
        if (m < DT_LN) {
#pragma unroll
          for (int b = 0; b < DT_TB; b++) sc[(4 + b) * DT_LN + m] = tv[r][b];
        }
      }
    }
    bool dif_n[DT_TB];
    const bool any_n = nt0 + DT_TB < ntr && look(nt0 + DT_TB, dif_n);
    if (any_n) fetch(nt0 + DT_TB, dif_n);
    if (any) __syncthreads();
#pragma unroll
    for (int b = 0; b < DT_TB; b++) {
      const int nt = nt0 + b;
      if (nt >= ntr) break;
      gd_t tr = V.f[F_trc] + okn + (size_t)nt * 2 * V.kk * np;
      if (dif[b]) {
        const double *l_x = sc + (4 + b) * DT_LN;
        if (upd) {                                           // :99-104, :121-126, :147-155
          const double fw = uw ? qw * (l_x[q - 1] - l_x[q]) : 0., fs = vs ? qs * (l_x[q - DT_LW] - l_x[q]) : 0.;
          const double fe = ue ? qe * (l_x[q] - l_x[q + 1]) : 0., fn = vn ? qn * (l_x[q] - l_x[q + DT_LW]) : 0.;
          tr[c] = l_x[q] - rq * (fe - fw + fn - fs);
        } else if (ring2) tr[c] = l_x[q];
      } else if (interior && ((move_mask >> nt) & 1ull)) tr[c] = WK(V, N_TR(nt))[c + ok];
    }
    any = any_n;
#pragma unroll
    for (int b = 0; b < DT_TB; b++) dif[b] = any_n && dif_n[b];
  }
}

// blomgpu_diffus on its own: S, T and the diffused tracers of level kn, whole padded planes, into the work-space slots
__global__ void k_diffus_stage_in(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  unsigned bx_, by_;
  xcd_block(bx_, by_);
  const int t = bx_ * blockDim.x + threadIdx.x;
  if (t >= V.nplane) return;
  const int k = by_;
  const size_t np = V.nplane, c = t, ok = (size_t)k * np, okn = (size_t)(k + nn) * np;
  WK(V, N_S)[c + ok] = V.f[F_saln][c + okn];
  WK(V, N_T)[c + ok] = V.f[F_temp][c + okn];
  for (int nt = 0; nt < V.ntr; nt++)
    if (!trc_skip_dif(V.P, nt + 1)) WK(V, N_TR(nt))[c + ok] = V.f[F_trc][c + okn + (size_t)nt * 2 * V.kk * np];
}
// a diffused tracer that pbcor1 did not touch (left out of advection, phy/mod_pbcor.F90:353-355) joins the others
__global__ void k_diffus_stage_in_trc(const DevView *__restrict__ Vp, int nn, int nt) {
  const DevView &V = *Vp;
  unsigned bx_, by_;
  xcd_block(bx_, by_);
  const int t = bx_ * blockDim.x + threadIdx.x;
  if (t >= V.nplane) return;
  const size_t np = V.nplane, c = t, ok = (size_t)by_ * np, okn = (size_t)(by_ + nn) * np;
  WK(V, N_TR(nt))[c + ok] = V.f[F_trc][c + okn + (size_t)nt * 2 * V.kk * np];
}

int st_diffus(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)n; (void)k1m;
  const DevView &h = c->h;
  const size_t np = h.nplane;
  if (3 + h.ntr > h.nwk || h.ntr > 64) return ctx_fail(c, "diffus: work space too small for this many tracers");
  const bool handed = c->pbcor1_handed_over;                 // pbcor1 left S, T and its tracers in the work space
  c->pbcor1_handed_over = false;
  if (int rc = st_xctilr(c, h.f[F_dp] + (size_t)(k1n - 1) * np, 1, h.kk, 3, 3, 1)) return rc;        // :58
  if (h.P.ltedtp_opt == 2) {
    // ltedtp = 'neutral', :59-70: the lateral diffusion is ale_regrid_remap's (stage_ndiff.hip); only the halo updates remain
    if (handed) return ctx_fail(c, "diffus: ltedtp = 'neutral' after a pbcor1 that handed its fields over");
    if (int rc = st_xctilr(c, h.f[F_temp] + (size_t)(k1n - 1) * np, 1, h.kk, 1, 1, 1)) return rc;
    if (int rc = st_xctilr(c, h.f[F_saln] + (size_t)(k1n - 1) * np, 1, h.kk, 1, 1, 1)) return rc;
    for (int nt = 0; nt < h.ntr; nt++) {
      if (h.P.itrtke > 0 && !h.P.tkeidf && (nt + 1 == h.P.itrtke || nt + 1 == h.P.itrgls)) continue;
      if (int rc = st_xctilr(c, h.f[F_trc] + ((size_t)(k1n - 1) + (size_t)nt * 2 * h.kk) * np, 1, h.kk, 1, 1, 1)) return rc;
    }
    return 0;
  }
  unsigned long long move_mask = 0;
  if (handed) {
    for (int nt = 0; nt < h.ntr; nt++) {
      const bool in_wk = !trc_skip_adv(h.P, nt + 1);          // pbcor1's rule (stage_pbcor_tile.hip)
      if (trc_skip_dif(h.P, nt + 1)) { if (in_wk) move_mask |= 1ull << nt; }
      else if (!in_wk) hipLaunchKernelGGL(k_diffus_stage_in_trc, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, nn, nt);
    }
  } else {
    hipLaunchKernelGGL(k_diffus_stage_in, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, nn);
  }
  {                                                                                                 // :72-80, on the work-space planes
    double *wk = h.wk;
    std::vector<double *> ptrs = {wk + (size_t)N_T * h.kk * np, wk + (size_t)N_S * h.kk * np};
    for (int nt = 0; nt < h.ntr; nt++)
      if (!trc_skip_dif(h.P, nt + 1)) ptrs.push_back(wk + (size_t)N_TR(nt) * h.kk * np);           // :76-78: no halo for the others
    std::vector<int> nl(ptrs.size(), h.kk), it(ptrs.size(), 1);
    for (size_t f = 0; f < ptrs.size(); f += 8) {
      const int g = (int)(ptrs.size() - f < 8 ? ptrs.size() - f : 8);
      if (int rc = st_xctilr_multi(c, g, ptrs.data() + f, nl.data() + f, 2, 2, it.data() + f)) return rc;
    }
  }
  {
    TimeScope ts(c, "diffus");
    const int ntx = (h.ni + DT_TW - 1) / DT_TW, nty = (h.nj + DT_TH - 1) / DT_TH;
    hipLaunchKernelGGL(k_diffus_tile, dim3(ntx * nty, h.kk), dim3(DT_NT), 0, c->stream, c->d, mm, nn, ntx, move_mask);
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}
