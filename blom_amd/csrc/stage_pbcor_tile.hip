// pbcor1 / pbcor2: the upstream fluxes and the divergence update of a layer in one LDS-tiled kernel --
// phy/mod_pbcor.F90:246-361 (pbcor1) and :592-692 (pbcor2).
//
// The one-kernel-per-sweep form (stage_pbcor.hip) writes the layer fluxes of mass, salt, heat and tracers at u- and
// v-points to 6 + 2 ntr work-space planes and its update kernel reads each of them at two points: 3.0 GB per call for
// ~1.1 GB of algorithmic bytes.  Here a workgroup owns a tile of PT_TW x PT_TH cells of one layer:
//   phase 0  dp, S, T, tracers of the tile with a one-cell rim go to LDS (the upstream cell of a face is the cell on
//            either side of it); everything else a thread needs -- transports, masks, bottom pressures, the old fluxes
//            of the faces it owns -- is loaded at the same time, all loads independent;
//   phase 1  each thread evaluates the fluxes through the four faces of its cell from the LDS values (every face is
//            evaluated by the two cells it separates, with the same operands), accumulates those of its west and its
//            south face into uflx.. / vflx.. (the reference's loop ranges), and updates its cell.
// The update cannot be written in place: the neighbouring tiles read the old values in their rims.  The new state goes
// to 3 + ntr work-space planes; k_pbc_rescale_from, the column pass that rescales dp to the bottom pressure anyway
// (:365-395, :696-726), reads them instead of the fields and moves S, T and the tracers into place.
// Expressions are those of k_pbc_flux / k_pbc_update / k_pbc_rescale (stage_pbcor.hip), operator for operator.
// Roofline: HBM; (25 + 6 ntr) F per call.
#include "blomgpu_internal.h"
#include "eos.h"

#define EPSILP 1.e-12
#define DPEPS1 1.e-5   // phy/mod_pbcor.F90:58
#define DPEPS2 1.e-7   // phy/mod_pbcor.F90:59

#define PT_TW 32
#define PT_TH 8
#define PT_LW (PT_TW + 2)
#define PT_LN (PT_LW * (PT_TH + 2))
#define PT_NT (PT_TW * PT_TH)

// work-space slots of the new state
#define N_DP 0
#define N_S 1
#define N_T 2
#define N_TR(nt) (3 + (nt))
// 2-D work planes written by k_pbc_total (stage_pbcor.hip)
#define S2_PBUT 1
#define S2_PBVT 2

#define MP(m) ((m) & 1)
#define MU(m) (((m) >> 1) & 1)
#define MV(m) (((m) >> 2) & 1)

#define PT_TB 4                    // tracers per batch: any number of tracers passes through the same PT_TB LDS slots

struct FaceFlux {
  double f, f2, f3;
  int up;                          // tile index of the upstream cell
};

// flux through a face with transport tot from the upstream cell at index up of the tile / plane index cup;
// k_pbc_flux, mod_pbcor.F90:246-275 resp. :592-621
__device__ inline void face_flux(const DevView &V, const double *sc, bool wet, double tot, int up, size_t cup, double pbot_up,
                                 double pbt, int k, FaceFlux &F) {
  F.f = 0.; F.f2 = 0.; F.f3 = 0.; F.up = up;
  if (!wet) return;
  if (V.P.bmcmth == 0) F.f = tot * sc[up] / pbot_up;
  else {
    const size_t np = V.nplane;
    F.f = tot * fmax2(0., fmin2(pbt, V.f[F_p][cup + (size_t)(k + 1) * np]) - V.f[F_p][cup + (size_t)k * np]) / pbt;
  }
  F.f2 = F.f * sc[PT_LN + up];
  F.f3 = F.f * sc[2 * PT_LN + up];
}

// from_remap (pbcor1 inside blomgpu_step): dp, T, S and the advected tracers of the level come from the work-space planes in
// which remap left them (stage_remap_tile.hip, FOLD), not from their fields
__global__ void __launch_bounds__(PT_NT) k_pbc_tile(const DevView *__restrict__ Vp, int which, int offc, int offf, int ntx, int from_remap KPROF_ARGS) {
  const DevView &V = *Vp;
  __shared__ double sc[(3 + PT_TB) * PT_LN];                 // dp, saln, temp, one batch of tracers
  unsigned bx_, by_;
  xcd_block(bx_, by_);
  const int k = by_, ni = V.ni, nj = V.nj, ntr = V.ntr, t = threadIdx.x;
  const int x0 = (bx_ % ntx) * PT_TW, y0 = (bx_ / ntx) * PT_TH;
  const size_t np = V.nplane, ok = (size_t)k * np, okc = (size_t)(k + offc) * np, okf = (size_t)(k + offf) * np;
  // KPROF words of workgroup bx_ of level 10 (thread 0): 0 start, 1 the tile is in LDS, 2 fluxes and S, T done, 3 end
  [[maybe_unused]] const bool kp_on = t == 0 && k == 10;
  if (kp_on) KPROF_MARK(bx_, 0);
  gcd_t f_dp = from_remap ? WK(V, R_DP(ntr)) + ok : V.f[F_dp] + okc;
  gcd_t f_s = from_remap ? WK(V, R_S(ntr)) + ok : V.f[F_saln] + okc;
  gcd_t f_t = from_remap ? WK(V, R_T(ntr)) + ok : V.f[F_temp] + okc;
  gcd_t f_tr = V.f[F_trc] + okc;
  // tracer nt of the level: remap advected it (then it lies in the work space) or left it alone (mod_remap.F90:314-316)
  auto trc_at = [&](int nt, size_t cs) {
    return from_remap && !trc_skip_adv(V.P, nt + 1) ? WK(V, R_TR(ntr, nt))[cs + ok] : f_tr[cs + (size_t)nt * 2 * V.kk * np];
  };

  // ---- phase 0 ---------------------------------------------------------------------------------------------------
  double sv[2][3 + PT_TB];
  size_t csr[2];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int m = t + r * PT_NT;
    const int xs = x0 - 1 + m % PT_LW, ys = y0 - 1 + m / PT_LW;
    const bool in = m < PT_LN && xs >= 0 && xs < ni && ys >= 0 && ys < nj;
    const size_t cs = in ? (size_t)ys * ni + xs : 0;
    csr[r] = cs;
    sv[r][0] = f_dp[cs]; sv[r][1] = f_s[cs]; sv[r][2] = f_t[cs];
#pragma unroll
    for (int nt = 0; nt < PT_TB; nt++) sv[r][3 + nt] = nt < ntr ? trc_at(nt, cs) : 0.;
  }
  const int lx = t % PT_TW, ly = t / PT_TW;
  const int x = x0 + lx, y = y0 + ly;
  const int i = x - (NBDY - 1), j = y - (NBDY - 1);
  // the cells with a face or an update lie at 1 <= i <= ii+1, 1 <= j <= jj+1: their neighbours exist
  const bool live = j >= 1 && j <= V.jj + 1 && i >= 1 && i <= V.ii + 1;
  const size_t c = live ? (size_t)y * ni + x : (size_t)ni + 1, e = c + 1, nb = c + ni;
  const int q = (ly + 1) * PT_LW + lx + 1;
  gci_t mpk = V.m[I_mpack];
  const int mp_c = live ? mpk[c] : 0, mp_e = live ? mpk[e] : 0, mp_n = live ? mpk[nb] : 0;
  gcd_t utot = which == 1 ? V.f[F_utotm] : V.f[F_utotn], vtot = which == 1 ? V.f[F_vtotm] : V.f[F_vtotn];
  gcd_t pbot = V.f[F_p] + (size_t)V.kk * np;
  const double ut_c = utot[c], ut_e = utot[e], vt_c = vtot[c], vt_n = vtot[nb];
  const double pb_c = pbot[c], pb_w = pbot[c - 1], pb_e = pbot[e], pb_s = pbot[c - ni], pb_n = pbot[nb];
  const bool dluc = V.P.bmcmth != 0;
  const double pbut_c = dluc ? WK2(V, S2_PBUT)[c] : 0., pbut_e = dluc ? WK2(V, S2_PBUT)[e] : 0.;
  const double pbvt_c = dluc ? WK2(V, S2_PBVT)[c] : 0., pbvt_n = dluc ? WK2(V, S2_PBVT)[nb] : 0.;
  const double s2i = V.f[F_scp2i][c];
  gd_t o_uf = V.f[F_uflx] + c + okf, o_us = V.f[F_usflx] + c + okf, o_ut = V.f[F_utflx] + c + okf;
  gd_t o_vf = V.f[F_vflx] + c + okf, o_vs = V.f[F_vsflx] + c + okf, o_vt = V.f[F_vtflx] + c + okf;
  const double uf_o = *o_uf, us_o = *o_us, ut_o = *o_ut, vf_o = *o_vf, vs_o = *o_vs, vt_o = *o_vt;
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int m = t + r * PT_NT;
    if (m < PT_LN) {
#pragma unroll
      for (int s = 0; s < 3 + PT_TB; s++) sc[s * PT_LN + m] = sv[r][s];
    }
  }
  __syncthreads();
  if (kp_on) KPROF_MARK(bx_, 1);

  // ---- phase 1 ---------------------------------------------------------------------------------------------------
  // faces of the reference's loops: u-faces at j = 1..jj, i = 1..ii+1; v-faces at j = 1..jj+1, i = 1..ii
  const bool uw = live && j <= V.jj && MU(mp_c), vs = live && i <= V.ii && MV(mp_c);
  const bool upd = live && j <= V.jj && i <= V.ii && MP(mp_c);
  FaceFlux W, S, E, N;
  face_flux(V, sc, uw, ut_c, ut_c > 0. ? q - 1 : q, ut_c > 0. ? c - 1 : c, ut_c > 0. ? pb_w : pb_c, pbut_c, k, W);
  face_flux(V, sc, vs, vt_c, vt_c > 0. ? q - PT_LW : q, vt_c > 0. ? c - ni : c, vt_c > 0. ? pb_s : pb_c, pbvt_c, k, S);
  if (uw) {                                                  // :262-264, :608-610
    *o_uf = uf_o + W.f;
    *o_us = us_o + W.f2;
    *o_ut = ut_o + W.f3;
  }
  if (vs) {
    *o_vf = vf_o + S.f;
    *o_vs = vs_o + S.f2;
    *o_vt = vt_o + S.f3;
  }
  face_flux(V, sc, upd && MU(mp_e), ut_e, ut_e > 0. ? q : q + 1, ut_e > 0. ? c : e, ut_e > 0. ? pb_c : pb_e, pbut_e, k, E);
  face_flux(V, sc, upd && MV(mp_n), vt_n, vt_n > 0. ? q : q + PT_LW, vt_n > 0. ? c : nb, vt_n > 0. ? pb_c : pb_n, pbvt_n, k, N);
  // k_pbc_update, :339-361 resp. :671-692
  double dpo = sc[q], dpni = 0.;
  if (upd) {
    const double dv = E.f - W.f + N.f - S.f;
    const double dv2 = E.f2 - W.f2 + N.f2 - S.f2;
    const double dv3 = E.f3 - W.f3 + N.f3 - S.f3;
    const double so = sc[PT_LN + q], to = sc[2 * PT_LN + q];
    if (which == 1) {
      const double dpn = fmax2(0., dpo - dv * s2i);
      dpo = dpo + DPEPS1;
      dpni = 1. / (dpn + DPEPS1);
      WK(V, N_S)[c + ok] = (dpo * so - dv2 * s2i) * dpni;
      WK(V, N_T)[c + ok] = (dpo * to - dv3 * s2i) * dpni;
      WK(V, N_DP)[c + ok] = dpn < DPEPS2 ? 0. : dpn;
    } else {
      double dpn = dpo - s2i * dv;
      dpni = 1. / dpn;
      const double sn = (dpo * so - s2i * dv2) * dpni;
      const double tn = (dpo * to - s2i * dv3) * dpni;
      WK(V, N_S)[c + ok] = sn;
      WK(V, N_T)[c + ok] = tn;
      V.f[F_sigma][c + (size_t)(k + offc) * np] = eos::sig(V.P, tn, sn);
      dpn = dpn - EPSILP;
      WK(V, N_DP)[c + ok] = dpn < DPEPS2 ? 0. : dpn;
    }
  }

  if (kp_on) KPROF_MARK(bx_, 2);
  // ---- phase 2: the tracers, PT_TB at a time through the same LDS slots (the first batch came in with phase 0) ----------
  // a batch's values are loaded while the batch before it is worked on: tv is free again once it has gone to LDS
  double tv[2][PT_TB];
#pragma unroll
  for (int r = 0; r < 2; r++)
#pragma unroll
    for (int b = 0; b < PT_TB; b++) tv[r][b] = PT_TB + b < ntr ? trc_at(PT_TB + b, csr[r]) : 0.;
  for (int nt0 = 0; nt0 < ntr; nt0 += PT_TB) {
    if (nt0 > 0) {
      __syncthreads();                                       // the previous batch has been read
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const int m = t + r * PT_NT;
        if (m < PT_LN) {
#pragma unroll
          for (int b = 0; b < PT_TB; b++) sc[(3 + b) * PT_LN + m] = tv[r][b];
        }
      }
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int b = 0; b < PT_TB; b++)
          tv[r][b] = nt0 + PT_TB + b < ntr ? trc_at(nt0 + PT_TB + b, csr[r]) : 0.;
      __syncthreads();
    }
    if (upd) {
#pragma unroll
      for (int b = 0; b < PT_TB; b++) {
        const int nt = nt0 + b;
        if (nt >= ntr) break;
        if (which == 1 && trc_skip_adv(V.P, nt + 1)) continue;            // :353-355 (pbcor2, :684, has no such test)
        const double *l_x = sc + (3 + b) * PT_LN;
        // a face without a u-/v-point carries no flux (F.f = 0 there, but 0 * tracer could be -0 or NaN: keep the exact 0)
        const double fw = uw ? W.f * l_x[W.up] : 0., fs = vs ? S.f * l_x[S.up] : 0.;
        const double fe = MU(mp_e) ? E.f * l_x[E.up] : 0., fn = MV(mp_n) ? N.f * l_x[N.up] : 0.;
        if (which == 1) WK(V, N_TR(nt))[c + ok] = (dpo * l_x[q] - (fe - fw + fn - fs) * s2i) * dpni;
        else WK(V, N_TR(nt))[c + ok] = (dpo * l_x[q] - (fe - fw + fn - fs) * s2i) * dpni;
      }
    }
  }
  if (kp_on) KPROF_MARK(bx_, 3);
}

// k_pbc_rescale (stage_pbcor.hip) reading the new state from the work space: p scan, pbfac = pb / p(kk+1), dp *= pbfac;
// S, T and the updated tracers move into their fields.  mod_pbcor.F90:365-395, :696-726.
#define RS_U 4
// move = 0: only dp is rescaled into its field; S, T and the tracers stay in the work space for the next stage (pbcor1 inside
// blomgpu_step: diffus takes them from there, stage_diffus.hip)
__global__ void __launch_bounds__(64) k_pbc_rescale_from(const DevView *__restrict__ Vp, int which, int m, int offc, int move) {
  const DevView &V = *Vp;
  const int t_ = blockIdx.x * blockDim.x + threadIdx.x;
  if (t_ >= V.nplane) return;
  const int i = t_ % V.ni - (NBDY - 1), j = t_ / V.ni - (NBDY - 1);
  const size_t c = t_;
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  const int kk = V.kk, ntr = V.ntr;
  gd_t dp = V.f[F_dp] + (size_t)offc * np + c, p = V.f[F_p] + c;
  gd_t saln = V.f[F_saln] + (size_t)offc * np + c, temp = V.f[F_temp] + (size_t)offc * np + c;
  gd_t trc = V.f[F_trc] + (size_t)offc * np + c;
  gcd_t ndp = WK(V, N_DP) + c, ns = WK(V, N_S) + c, nt_ = WK(V, N_T) + c;
  const double ptop = p[0];
  const double psum = column_scan(ptop, ndp, p, np, kk);
  const double pbfac = (which == 1 ? V.f[F_pb_p][c] : V.f[F_pb][c + (size_t)(m - 1) * np]) / psum;
  double acc = ptop;
  for (int k0 = 0; k0 < kk; k0 += RS_U) {
    double a0[RS_U], a1[RS_U], a2[RS_U];
#pragma unroll
    for (int u = 0; u < RS_U; u++) {
      const size_t o = (size_t)(k0 + u < kk ? k0 + u : kk - 1) * np;
      a0[u] = ndp[o];
      a1[u] = move ? ns[o] : 0.;
      a2[u] = move ? nt_[o] : 0.;
    }
#pragma unroll
    for (int u = 0; u < RS_U; u++) {
      const int k = k0 + u;
      if (k >= kk) break;
      const size_t o = (size_t)k * np;
      const double d = a0[u] * pbfac;
      dp[o] = d;
      if (which == 2) { acc = acc + d; p[o + np] = acc; }
      if (move) { saln[o] = a1[u]; temp[o] = a2[u]; }
    }
  }
  if (!move) return;
  for (int nt = 0; nt < ntr; nt++) {
    if (which == 1 && trc_skip_adv(V.P, nt + 1)) continue;
    gcd_t src = WK(V, N_TR(nt)) + c;
    double *dst = trc + (size_t)nt * 2 * kk * np;
    for (int k0 = 0; k0 < kk; k0 += 2 * RS_U) {
      double a[2 * RS_U];
#pragma unroll
      for (int u = 0; u < 2 * RS_U; u++) a[u] = src[(size_t)(k0 + u < kk ? k0 + u : kk - 1) * np];
#pragma unroll
      for (int u = 0; u < 2 * RS_U; u++)
        if (k0 + u < kk) dst[(size_t)(k0 + u) * np] = a[u];
    }
  }
}

int pbcor_tile_launch(blomgpu_ctx *c, int which, int m, int offc, int offf, int from_remap) {
  const DevView &h = c->h;
  if (3 + h.ntr > h.nwk) return ctx_fail(c, "pbcor: work space too small for this many tracers");
  const int ntx = (h.ni + PT_TW - 1) / PT_TW, nty = (h.nj + PT_TH - 1) / PT_TH;
  {
    TimeScope tk(c, "k_pbc_tile");
    hipLaunchKernelGGL(k_pbc_tile, dim3(ntx * nty, h.kk), dim3(PT_NT), 0, c->stream, c->d, which, offc, offf, ntx, from_remap KPROF_PASS(10));
  }
  // inside blomgpu_step pbcor1 hands S, T and the tracers to diffus through the work space, pbcor2 to tmsmt2
  // (with ltedtp = 'neutral' diffus is halo updates only: nothing to hand to it)
  const int move = !c->in_sequence || (which == 1 && h.P.ltedtp_opt == 2);
  (which == 1 ? c->pbcor1_handed_over : c->pbcor2_handed_over) = !move;
  // pbcor2 inside blomgpu_step: tmsmt2, the next stage, rewrites dp of level m pointwise from what this pass would store and
  // recomputes p: it takes the unscaled thicknesses from the work space and forms the factor itself (stage_simple.hip: k_tmsmt2_fac)
  if (which == 2 && !move && c->tmsmt_fold) { c->pbcor2_dp_in_wk = true; return 0; }
  hipLaunchKernelGGL(k_pbc_rescale_from, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, which, m, offc, move);
  return 0;
}
