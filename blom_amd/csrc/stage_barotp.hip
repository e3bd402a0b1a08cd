// barotp -- split-explicit barotropic subcycling, phy/mod_barotp.F90:148-1003.
//
// 2.5*lstep forward-backward substeps of 2-D continuity + u/v momentum (5 phases of lstep/2
// substeps).  The three equations of a substep depend on each other through nearest
// neighbours (u needs the new pb of the west cell, v the new u of four corners), hence one
// kernel per equation, in the reference's order: odd substep = halo update, continuity, u, v;
// even substep = continuity, v, u on ranges shrunk by the width the odd substep consumed
// (:401-413, :420-457, :520-557, :626-636, :646-682, :745-781).
// All fields are 2-D (G = ni*nj*8 B; the ~40 G working set lives in L2/Infinity Cache), so this
// stage is launch/latency bound, not HBM bound: first version = one launch
// per equation; the per-substep time weights wo,wm,wn are computed on the host in the
// reference's order of operations.
#include "blomgpu_internal.h"

#define ONEM 9806.

#define THREAD_IJ(V)                                                       \
  const int t_ = blockIdx.x * blockDim.x + threadIdx.x;                    \
  if (t_ >= (V).nplane) return;                                            \
  const int i = t_ % (V).ni - (NBDY - 1), j = t_ / (V).ni - (NBDY - 1);    \
  const size_t c = t_

// ---- :177-224 velocity bounds and coastal wave breaking coefficients (column max/min of u,v) ----
__global__ void k_bt_bounds(const DevView *__restrict__ Vp, int m, int nn) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const size_t np = V.nplane, om = c + (size_t)(m - 1) * np;
  if (V.m[I_iu][c]) {
    const double pbu = V.f[F_pbu][om];
    V.f[F_uglue][c] = V.P.cwbdts * exp_libm(1. - pbu / (V.P.cwbdls * ONEM));
    double mx = 0., mn = 0.;
    for (int k0 = 0; k0 < V.kk; k0 += COLUMN_U) {              // COLUMN_U levels' loads in flight (blomgpu_internal.h)
      double a0[COLUMN_U];
#pragma unroll
      for (int q = 0; q < COLUMN_U; q++) a0[q] = V.f[F_u][c + (size_t)((k0 + q < V.kk ? k0 + q : V.kk - 1) + nn) * np];
#pragma unroll
      for (int q = 0; q < COLUMN_U; q++)
        if (k0 + q < V.kk) { mx = fmax2(mx, a0[q]); mn = fmin2(mn, a0[q]); }
    }
    V.f[F_umaxb][c] = (V.f[F_umax][c] - mx) * pbu * V.f[F_scuy][c];
    V.f[F_uminb][c] = (V.f[F_umax][c] + mn) * pbu * V.f[F_scuy][c];
  }
  if (V.m[I_iv][c]) {
    const double pbv = V.f[F_pbv][om];
    V.f[F_vglue][c] = V.P.cwbdts * exp_libm(1. - pbv / (V.P.cwbdls * ONEM));
    double mx = 0., mn = 0.;
    for (int k0 = 0; k0 < V.kk; k0 += COLUMN_U) {              // COLUMN_U levels' loads in flight (blomgpu_internal.h)
      double a0[COLUMN_U];
#pragma unroll
      for (int q = 0; q < COLUMN_U; q++) a0[q] = V.f[F_v][c + (size_t)((k0 + q < V.kk ? k0 + q : V.kk - 1) + nn) * np];
#pragma unroll
      for (int q = 0; q < COLUMN_U; q++)
        if (k0 + q < V.kk) { mx = fmax2(mx, a0[q]); mn = fmin2(mn, a0[q]); }
    }
    V.f[F_vmaxb][c] = (V.f[F_vmax][c] - mx) * pbv * V.f[F_scvx][c];
    V.f[F_vminb][c] = (V.f[F_vmax][c] + mn) * pbv * V.f[F_scvx][c];
  }
}

// ---- :230-268 barotropic potential vorticity.  The reference writes pvtrop(:,:,n) in three
// sweeps (u-point pairs, v-point pairs, interior q-points), later sweeps overriding earlier
// ones; per q-point the surviving value is that of the LAST writer in that order. ------------
__global__ void k_bt_pvtrop_old(const DevView *__restrict__ Vp, int n) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < -2 || j > V.jj + 3 || i < 0 || i > V.ii + 1) return;
  V.f[F_pvtrop_o][c] = V.f[F_pvtrop][c + (size_t)(n - 1) * V.nplane];
}

__global__ void k_bt_pvtrop(const DevView *__restrict__ Vp, int n) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  const int ii = V.ii, jj = V.jj, ni = V.ni;
  if (j < 0 || j > jj + 1 || i < 0 || i > ii + 1) return;
  gcd_t pb_p = V.f[F_pb_p];
  gci_t iu = V.m[I_iu], iv = V.m[I_iv];
  const double cq = V.f[F_corioq][c];
  bool have = false;
  double val = 0.;
  // sweep 1, u-points (i',j') with j' = 0..jj, i' = 1..ii write q(i',j') and q(i',j'+1); a later
  // j' overrides an earlier one, so for q(i,j) the writer j'=j (if valid) wins over j'=j-1.
  if (i >= 1 && i <= ii) {
    if (j - 1 >= 0 && j - 1 <= jj && iu[c - ni]) { val = cq * (2. / (pb_p[c - ni] + pb_p[c - ni - 1])); have = true; }
    if (j >= 0 && j <= jj && iu[c]) { val = cq * (2. / (pb_p[c] + pb_p[c - 1])); have = true; }
  }
  // sweep 2, v-points (i',j') with j' = 1..jj, i' = 0..ii write q(i',j') and q(i'+1,j')
  if (j >= 1 && j <= jj) {
    if (i - 1 >= 0 && i - 1 <= ii && iv[c - 1]) { val = cq * (2. / (pb_p[c - 1] + pb_p[c - 1 - ni])); have = true; }
    if (i >= 0 && i <= ii && iv[c]) { val = cq * (2. / (pb_p[c] + pb_p[c - ni])); have = true; }
  }
  // sweep 3, interior q-points
  if (j >= 1 && j <= jj && i >= 1 && i <= ii && V.m[I_iq][c]) {
    val = cq * 4. / (pb_p[c] + pb_p[c - 1] + pb_p[c - ni] + pb_p[c - 1 - ni]);
    have = true;
  }
  if (have) V.f[F_pvtrop][c + (size_t)(n - 1) * V.nplane] = val;
}

// ---- nb == 1 reload of the subcycling state, :339-348 --------------------------------------------
__global__ void k_bt_load(const DevView *__restrict__ Vp) {   // always into buffer set 0 (*_t)
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  for (int l = 0; l < 2; l++) {
    const size_t o = c + (size_t)l * V.nplane;
    V.f[F_pb_t][o] = V.f[F_pb_mn][o];
    V.f[F_ubflx_t][o] = V.f[F_ubflx_mn][o];
    V.f[F_vbflx_t][o] = V.f[F_vbflx_mn][o];
  }
}

__global__ void k_bt_zero_sums(const DevView *__restrict__ Vp) {       // :361-379
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j >= -1 && j <= V.jj + 2 && i >= 0 && i <= V.ii + 1 && V.m[I_iu][c]) { V.f[F_ubflxs_t][c] = 0.; V.f[F_ubcors_t][c] = 0.; }
  if (j >= 0 && j <= V.jj + 2 && i >= 0 && i <= V.ii && V.m[I_iv][c]) { V.f[F_vbflxs_t][c] = 0.; V.f[F_vbcors_t][c] = 0.; }
}

struct BtArgs {
  int m, n, ml, nl;       // 1-based level indices
  double wo, wm, wn;
  int j0, j1, i0, i1;     // Fortran index range of this sweep
  int lv;                 // level of the "other" flux component used in the Coriolis term
};

// ---- continuity, :401-411 / :626-636 ---------------------------------------------------------------
__global__ void k_bt_cont(const DevView *__restrict__ Vp, BtArgs a) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < a.j0 || j > a.j1 || i < a.i0 || i > a.i1 || !V.m[I_ip][c]) return;
  const size_t np = V.nplane, oml = (size_t)(a.ml - 1) * np, onl = (size_t)(a.nl - 1) * np;
  const double wbaro = V.P.wbaro, dlt = V.P.dlt;
  gcd_t ub = V.f[F_ubflx_t] + oml, vb = V.f[F_vbflx_t] + oml;
  gd_t pb = V.f[F_pb_t];
  pb[c + onl] = (1. - wbaro) * pb[c + oml] + wbaro * pb[c + onl] -
                (1. + wbaro) * dlt * (ub[c + 1] - ub[c] + vb[c + V.ni] - vb[c]) * V.f[F_scp2i][c];
}

// ---- u momentum, :420-457 / :745-781 (enscon) and :464-502 / :788-826 (enecon, enedis) ------------
__global__ void k_bt_umom(const DevView *__restrict__ Vp, BtArgs a) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < a.j0 || j > a.j1 || i < a.i0 || i > a.i1 || !V.m[I_iu][c]) return;
  const size_t np = V.nplane, oml = (size_t)(a.ml - 1) * np, onl = (size_t)(a.nl - 1) * np;
  const size_t om = (size_t)(a.m - 1) * np, on = (size_t)(a.n - 1) * np;
  const int ni = V.ni;
  const double wbaro = V.P.wbaro, dlt = V.P.dlt, wo = a.wo, wm = a.wm, wn = a.wn;
  gd_t ub = V.f[F_ubflx_t];
  gcd_t vb = V.f[F_vbflx_t] + (size_t)(a.lv - 1) * np, pb = V.f[F_pb_t] + onl;
  gcd_t scvxi = V.f[F_scvxi], pvo = V.f[F_pvtrop_o], pvm = V.f[F_pvtrop] + om, pvn = V.f[F_pvtrop] + on;
  const double ubml = ub[c + oml], ubnl = ub[c + onl];
  V.f[F_ubflxs_t][c] = V.f[F_ubflxs_t][c] - wbaro * ubnl + (1. + wbaro) * ubml;
  double q;
  const size_t w = c - 1, nb = c + ni, nw = c - 1 + ni;
  if (V.P.mommth == 0)
    q = (vb[c] * scvxi[c] + vb[nb] * scvxi[nb] + vb[w] * scvxi[w] + vb[nw] * scvxi[nw]) *
        (wo * (pvo[c] + pvo[nb]) + wm * (pvm[c] + pvm[nb]) + wn * (pvn[c] + pvn[nb])) * .125;
  else
    q = .25 * ((vb[c] * scvxi[c] + vb[w] * scvxi[w]) * (wo * pvo[c] + wm * pvm[c] + wn * pvn[c]) +
               (vb[nb] * scvxi[nb] + vb[nw] * scvxi[nw]) * (wo * pvo[nb] + wm * pvm[nb] + wn * pvn[nb]));
  V.f[F_ubcors_t][c] = V.f[F_ubcors_t][c] + q;
  const double pbc = pb[c], pbw = pb[w];
  const double utndcy =
      q + (wo * (V.f[F_pgfxm_o][c] - (V.f[F_xixp_o][c] * pbc - V.f[F_xixm_o][c] * pbw)) +
           wm * (V.f[F_pgfxm][c + om] - (V.f[F_xixp][c + om] * pbc - V.f[F_xixm][c + om] * pbw)) +
           wn * (V.f[F_pgfxm][c + on] - (V.f[F_xixp][c + on] * pbc - V.f[F_xixm][c + on] * pbw))) *
              V.f[F_scuxi][c];
  const double x = (1. - wbaro) * ubml + wbaro * ubnl +
                   (1. + wbaro) * dlt *
                       ((utndcy + V.f[F_utotn][c]) * V.f[F_scuy][c] * fmin2(pbw, pbc) - V.f[F_uglue][c] * ubml);
  ub[c + onl] = fmax2(-V.f[F_uminb][c], fmin2(V.f[F_umaxb][c], x));
}

// ---- v momentum, :520-557 / :646-682 (enscon) and :564-602 / :690-728 ------------------------------
__global__ void k_bt_vmom(const DevView *__restrict__ Vp, BtArgs a) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < a.j0 || j > a.j1 || i < a.i0 || i > a.i1 || !V.m[I_iv][c]) return;
  const size_t np = V.nplane, oml = (size_t)(a.ml - 1) * np, onl = (size_t)(a.nl - 1) * np;
  const size_t om = (size_t)(a.m - 1) * np, on = (size_t)(a.n - 1) * np;
  const int ni = V.ni;
  const double wbaro = V.P.wbaro, dlt = V.P.dlt, wo = a.wo, wm = a.wm, wn = a.wn;
  gd_t vb = V.f[F_vbflx_t];
  gcd_t ub = V.f[F_ubflx_t] + (size_t)(a.lv - 1) * np, pb = V.f[F_pb_t] + onl;
  gcd_t scuyi = V.f[F_scuyi], pvo = V.f[F_pvtrop_o], pvm = V.f[F_pvtrop] + om, pvn = V.f[F_pvtrop] + on;
  const double vbml = vb[c + oml], vbnl = vb[c + onl];
  V.f[F_vbflxs_t][c] = V.f[F_vbflxs_t][c] - wbaro * vbnl + (1. + wbaro) * vbml;
  double q;
  const size_t e = c + 1, s = c - ni, se = c + 1 - ni;
  if (V.P.mommth == 0)
    q = -(ub[c] * scuyi[c] + ub[e] * scuyi[e] + ub[s] * scuyi[s] + ub[se] * scuyi[se]) *
        (wo * (pvo[c] + pvo[e]) + wm * (pvm[c] + pvm[e]) + wn * (pvn[c] + pvn[e])) * .125;
  else
    q = -.25 * ((ub[c] * scuyi[c] + ub[s] * scuyi[s]) * (wo * pvo[c] + wm * pvm[c] + wn * pvn[c]) +
                (ub[e] * scuyi[e] + ub[se] * scuyi[se]) * (wo * pvo[e] + wm * pvm[e] + wn * pvn[e]));
  V.f[F_vbcors_t][c] = V.f[F_vbcors_t][c] + q;
  const double pbc = pb[c], pbs = pb[s];
  const double vtndcy =
      q + (wo * (V.f[F_pgfym_o][c] - (V.f[F_xiyp_o][c] * pbc - V.f[F_xiym_o][c] * pbs)) +
           wm * (V.f[F_pgfym][c + om] - (V.f[F_xiyp][c + om] * pbc - V.f[F_xiym][c + om] * pbs)) +
           wn * (V.f[F_pgfym][c + on] - (V.f[F_xiyp][c + on] * pbc - V.f[F_xiym][c + on] * pbs))) *
              V.f[F_scvyi][c];
  const double x = (1. - wbaro) * vbml + wbaro * vbnl +
                   (1. + wbaro) * dlt *
                       ((vtndcy + V.f[F_vtotn][c]) * V.f[F_scvx][c] * fmin2(pbs, pbc) - V.f[F_vglue][c] * vbml);
  vb[c + onl] = fmax2(-V.f[F_vminb][c], fmin2(V.f[F_vmaxb][c], x));
}

// ---- phase epilogues, :847-977 ------------------------------------------------------------------------
__global__ void k_bt_epilogue(const DevView *__restrict__ Vp, int nb, int m, int n, int ml, int nl, int set) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const size_t np = V.nplane, om = (size_t)(m - 1) * np, on = (size_t)(n - 1) * np;
  const size_t oml = (size_t)(ml - 1) * np, onl = (size_t)(nl - 1) * np, o3 = 2 * np;
  const bool wp = V.m[I_ip][c], wu = V.m[I_iu][c], wv = V.m[I_iv][c];
  gcd_t pbt = set ? V.f[F_pb_t2] : V.f[F_pb_t], ubt = set ? V.f[F_ubflx_t2] : V.f[F_ubflx_t];
  gcd_t vbt = set ? V.f[F_vbflx_t2] : V.f[F_vbflx_t];
  const double us = V.f[F_ubflxs_t][c], vs = V.f[F_vbflxs_t][c];
  if (nb == 1 || nb == 3) {
    const size_t ol = nb == 1 ? om : on;
    if (wp) V.f[F_pb][c + ol] = pbt[c + oml];
    if (wu) {
      const double pbu = fmin2(pbt[c + oml], pbt[c - 1 + oml]);
      V.f[F_pbu][c + ol] = pbu;
      const double f = ubt[c + oml];
      V.f[F_ubflx][c + ol] = f;
      V.f[F_ub][c + ol] = f / (pbu * V.f[F_scuy][c]);
      if (nb == 1) {
        V.f[F_ubflxs][c + on] = V.f[F_ubflxs][c + on] + us;
        V.f[F_ubflxs][c + om] = V.f[F_ubflxs][c + o3] + us;
      } else {
        V.f[F_ubflxs_p][c + om] = V.f[F_ubflxs][c + om] + us;
        V.f[F_ubflxs_p][c + on] = V.f[F_ubflxs_p][c + on] + us;
        V.f[F_ubcors_p][c] = V.f[F_ubcors_p][c] + V.f[F_ubcors_t][c];
      }
    }
    if (wv) {
      const double pbv = fmin2(pbt[c + oml], pbt[c - V.ni + oml]);
      V.f[F_pbv][c + ol] = pbv;
      const double f = vbt[c + oml];
      V.f[F_vbflx][c + ol] = f;
      V.f[F_vb][c + ol] = f / (pbv * V.f[F_scvx][c]);
      if (nb == 1) {
        V.f[F_vbflxs][c + on] = V.f[F_vbflxs][c + on] + vs;
        V.f[F_vbflxs][c + om] = V.f[F_vbflxs][c + o3] + vs;
      } else {
        V.f[F_vbflxs_p][c + om] = V.f[F_vbflxs][c + om] + vs;
        V.f[F_vbflxs_p][c + on] = V.f[F_vbflxs_p][c + on] + vs;
        V.f[F_vbcors_p][c] = V.f[F_vbcors_p][c] + V.f[F_vbcors_t][c];
      }
    }
  } else if (nb == 2) {
    if (wp) { V.f[F_pb_mn][c + oml] = pbt[c + oml]; V.f[F_pb_mn][c + onl] = pbt[c + onl]; }
    if (wu) {
      V.f[F_ubflx_mn][c + oml] = ubt[c + oml];
      V.f[F_ubflx_mn][c + onl] = ubt[c + onl];
      V.f[F_ubflxs][c + om] = V.f[F_ubflxs][c + om] + us;
      V.f[F_ubflxs][c + o3] = us;
      V.f[F_ubflxs_p][c + on] = us;
      V.f[F_ubcors_p][c] = V.f[F_ubcors_t][c];
    }
    if (wv) {
      V.f[F_vbflx_mn][c + oml] = vbt[c + oml];
      V.f[F_vbflx_mn][c + onl] = vbt[c + onl];
      V.f[F_vbflxs][c + om] = V.f[F_vbflxs][c + om] + vs;
      V.f[F_vbflxs][c + o3] = vs;
      V.f[F_vbflxs_p][c + on] = vs;
      V.f[F_vbcors_p][c] = V.f[F_vbcors_t][c];
    }
  } else {
    if (nb == 5) {
      if (wp) V.f[F_pb_p][c] = pbt[c + oml];
      if (wu) V.f[F_pbu_p][c] = fmin2(pbt[c + oml], pbt[c - 1 + oml]);
      if (wv) V.f[F_pbv_p][c] = fmin2(pbt[c + oml], pbt[c - V.ni + oml]);
    }
    if (wu) {
      V.f[F_ubflxs_p][c + on] = V.f[F_ubflxs_p][c + on] + us;
      V.f[F_ubcors_p][c] = V.f[F_ubcors_p][c] + V.f[F_ubcors_t][c];
    }
    if (wv) {
      V.f[F_vbflxs_p][c + on] = V.f[F_vbflxs_p][c + on] + vs;
      V.f[F_vbcors_p][c] = V.f[F_vbcors_p][c] + V.f[F_vbcors_t][c];
    }
  }
}

// with the arctic patch umaxb/uminb, vmaxb/vminb, xixp/xixm, xiyp/xiym change roles in the halo next to the
// grid intersection, phy/mod_barotp.F90:290-325
__global__ void k_bt_arctic_swap(const DevView *__restrict__ Vp, int n) {
  const DevView &V = *Vp;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= V.nplane) return;
  if (V.j0 + V.jj != V.jtdm) return;                       // nproc == jpr: only the last tile row holds the seam
  const int i = t % V.ni - (NBDY - 1), j = t / V.ni - (NBDY - 1);
  if (i < 0 || i > V.ii + 1 || j < V.jj || j > V.jj + 2) return;
  const size_t c = t, on = (size_t)(n - 1) * V.nplane;
  double q = V.f[F_umaxb][c]; V.f[F_umaxb][c] = V.f[F_uminb][c]; V.f[F_uminb][c] = q;
  q = V.f[F_xixp][c + on]; V.f[F_xixp][c + on] = V.f[F_xixm][c + on]; V.f[F_xixm][c + on] = q;
  const int ilo = V.itdm / 2 - V.i0 + 1;                   // do i = max(0,itdm/2-i0+1),ii+1   (:303)
  if (j > V.jj || i >= (ilo > 0 ? ilo : 0)) {
    q = V.f[F_vmaxb][c]; V.f[F_vmaxb][c] = V.f[F_vminb][c]; V.f[F_vminb][c] = q;
    q = V.f[F_xiyp][c + on]; V.f[F_xiyp][c + on] = V.f[F_xiym][c + on]; V.f[F_xiym][c + on] = q;
  }
}

int bt_pair_halo(blomgpu_ctx *c, int set);
bool bt_phase_usable(blomgpu_ctx *c);
int bt_phase_launch(blomgpu_ctx *c, int m, int n, int ml, int nl, double woa, double wob, double wna, double wnb, int lll0,
                    int last, int src, int *src_out, int *ml_out, int *nl_out);
int bt_phase_check(blomgpu_ctx *c);
int bt_overlap_usable(blomgpu_ctx *c);
int bt_block_mode(blomgpu_ctx *c);
int bt_block_launch(blomgpu_ctx *c, int mode, int m, int n, int ml, int nl, const int *ph_last, const double (*ph_w)[4], int src,
                    int *src_out, int *ml_out, int *nl_out);
int bt_pair_launch(blomgpu_ctx *c, int m, int n, int ml, int nl, const double *wo, const double *wm, const double *wn,
                   int do_odd, int do_even, int src, int tsel, RcclLanded *rim);
int bt_pair_halo_landed(blomgpu_ctx *c, int set, RcclLanded *landed);

int st_barotp_bounds(blomgpu_ctx *c, int m, int nn) {
  hipLaunchKernelGGL(k_bt_bounds, plane_grid(c->h, 1, 64), dim3(64), 0, c->stream, c->d, m, nn);
  return 0;
}
int st_barotp_on(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n, bool with_bounds);
int st_barotp(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  TimeScope ts(c, "barotp");
  // RCCL tiles with a global barotropic context attached: gather, solve the whole 2-D domain here, take the window
  if (c->tiling.rccl && c->bt_global) return rccl_barotp_replicated(c, m, n, mm, nn, k1m, k1n);
  return st_barotp_on(c, m, n, mm, nn, k1m, k1n, true);
}
// with_bounds = false: the velocity bounds and wave-breaking coefficients (:177-224, the part that reads 3-D fields) are
// already in umaxb.., uglue.. (the global barotropic context of RCCL tiles holds no 3-D state)
int st_barotp_on(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n, bool with_bounds) {
  (void)mm; (void)k1m; (void)k1n;
  const DevView &h = c->h;
  const size_t np = h.nplane;
  const int ii = h.ii, jj = h.jj, lstep = h.P.lstep;
  if (lstep < 2 || lstep % 2) return ctx_fail(c, "barotp: lstep must be even (phy/mod_time.F90:137-139)");
  const dim3 g = plane_grid(h), b(256);
  if (with_bounds) hipLaunchKernelGGL(k_bt_bounds, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, m, nn);
  hipLaunchKernelGGL(k_bt_pvtrop_old, g, b, 0, c->stream, c->d, n);
  hipLaunchKernelGGL(k_bt_pvtrop, g, b, 0, c->stream, c->d, n);
  // :271-285
  struct { int f, lev, nh, it; } hl[] = {
      {F_uglue, 0, 2, 3}, {F_utotn, 0, 2, 13}, {F_umaxb, 0, 2, 3}, {F_uminb, 0, 2, 3},
      {F_vglue, 0, 2, 4}, {F_vtotn, 0, 2, 14}, {F_vmaxb, 0, 2, 4}, {F_vminb, 0, 2, 4},
      {F_pvtrop, n - 1, 3, 2}, {F_pgfxm, n - 1, 2, 13}, {F_xixp, n - 1, 2, 3}, {F_xixm, n - 1, 2, 3},
      {F_pgfym, n - 1, 2, 14}, {F_xiyp, n - 1, 2, 4}, {F_xiym, n - 1, 2, 4}};
  for (int nhw = 2; nhw <= 3; nhw++) {        // one launch per halo width
    double *ptrs[16];
    int nl[16], it[16], nf = 0;
    for (auto &x : hl)
      if (x.nh == nhw) { ptrs[nf] = h.f[x.f] + (size_t)x.lev * np; nl[nf] = 1; it[nf] = x.it; nf++; }
    if (int rc = st_xctilr_multi(c, nf, ptrs, nl, 1, nhw, it)) return rc;
  }

  if (h.nreg == 2) hipLaunchKernelGGL(k_bt_arctic_swap, g, b, 0, c->stream, c->d, n);       // :290-325
  if (int rc = ctx_err_words(c)) return rc;
  c->bt_restart = true;
  int lll0 = 1, ml = 1, nl = 2, set = 0;     // set: which buffer set (*_t / *_t2) holds the current state
  double woa = 0., wob = 0., wna = 0., wnb = 0.;
  if (c->barotp_fused && h.nreg != 2 && bt_block_mode(c)) {
    // k_bt_steps4: ONE launch walks the substeps of all five phases, four per hand-off, zeroing each phase's flux sums and doing its
    // epilogue itself (the time weights of the phases, :352-360 and below, as a table)
    int ph_last[5];
    double ph_w[5][4];
    for (int nb = 1; nb <= 5; nb++) {
      if (nb == 1) { lll0 = 1; woa = -1. / lstep; wob = .5 + (lll0 - .5) / lstep; wna = 0.; wnb = 0.; }
      else if (nb == 2) { woa = 0.; wob = 0.; wna = 1. / lstep; wnb = -(lll0 - .5) / lstep; }
      else if (nb == 4) { wna = 0.; wnb = 1.; }
      ph_w[nb - 1][0] = woa; ph_w[nb - 1][1] = wob; ph_w[nb - 1][2] = wna; ph_w[nb - 1][3] = wnb;
      ph_last[nb - 1] = lll0 + lstep / 2 - 1;
      lll0 = lll0 + lstep / 2;
    }
    hipLaunchKernelGGL(k_bt_load, g, b, 0, c->stream, c->d);
    int so, mo, no;
    if (int rc = bt_block_launch(c, bt_block_mode(c), m, n, 1, 2, ph_last, ph_w, 0, &so, &mo, &no)) return rc;
    HIPCHK(c, hipGetLastError());
    if (c->barotp_persist && bt_phase_usable(c)) return bt_phase_check(c);
    return 0;
  }
  for (int nb = 1; nb <= 5; nb++) {
    if (nb == 1) {
      lll0 = 1; ml = 1; nl = 2;
      woa = -1. / lstep;
      wob = .5 + (lll0 - .5) / lstep;
      wna = 0.; wnb = 0.;
      hipLaunchKernelGGL(k_bt_load, g, b, 0, c->stream, c->d);
    } else if (nb == 2) {
      woa = 0.; wob = 0.;
      wna = 1. / lstep;
      wnb = -(lll0 - .5) / lstep;
    } else if (nb == 4) {
      wna = 0.; wnb = 1.;
    }
    const int last = lll0 + lstep / 2 - 1;
    // With the arctic patch the halo update also rewrites the seam row jj, an interior row, so it must
    // happen exactly where the reference has it: before odd substeps only.  The fused kernels then publish
    // the margins they computed (PairArgs::write_margin), which is what the reference's arrays hold before
    // a lone even substep and before the epilogue; single tile only.
    const bool arctic1 = h.nreg == 2 && (!c->tiling.multi() || c->barotp_arctic_fused);
    const bool fused = c->barotp_fused && (h.nreg != 2 || arctic1);
    bool halo_done = false;
    hipLaunchKernelGGL(k_bt_zero_sums, g, b, 0, c->stream, c->d);
    if (fused && arctic1 && !c->tiling.multi() && c->barotp_persist && bt_phase_usable(c)) {
      // arctic patch, single tile: the halo update -- which also rewrites the seam row -- belongs in front of odd substeps
      // only, so the persistent launch takes the whole odd+even pairs of the phase (its tiles re-read rim and seam at the
      // top of every pair), and a lone even substep at the start or a lone odd one at the end of a phase is its own launch
      int lll = lll0;
      double wo[2], wm[2], wn[2];
      auto weights = [&](int l) {
        wo[0] = wo[1] = woa * l + wob;
        wn[0] = wn[1] = wna * l + wnb;
        wm[0] = wm[1] = 1. - wo[0] - wn[0];
      };
      if (lll % 2 == 0) {                                                  // lone even substep: reads the published margins
        weights(lll);
        bt_pair_launch(c, m, n, ml, nl, wo, wm, wn, 0, 1, set, 0, nullptr);
        set ^= 1;
        { const int ll = ml; ml = nl; nl = ll; }
        lll++;
      }
      const int npair = (last - lll + 1) / 2;
      if (npair > 0) {
        int so, mo, no;
        if (int rc = bt_phase_launch(c, m, n, ml, nl, woa, wob, wna, wnb, lll, lll + 2 * npair - 1, set, &so, &mo, &no)) return rc;
        set = so; ml = mo; nl = no;
        lll += 2 * npair;
      }
      if (lll <= last) {                                                   // lone odd substep: folds while loading
        weights(lll);
        bt_pair_launch(c, m, n, ml, nl, wo, wm, wn, 1, 0, set, 0, nullptr);
        set ^= 1;
        { const int ll = ml; ml = nl; nl = ll; }
      }
    } else if (fused && !arctic1 && c->barotp_persist && bt_phase_usable(c)) {
      // the whole phase in one launch (k_bt_steps<true>): coefficients stay on chip, tiles hand each other
      // their edge values through memory
      int so, mo, no;
      if (int rc = bt_phase_launch(c, m, n, ml, nl, woa, wob, wna, wnb, lll0, last, set, &so, &mo, &no)) return rc;
      set = so; ml = mo; nl = no;
    } else if (fused) {
      // fused odd+even substep pairs per LDS tile (stage_barotp_pair.hip); single substeps only
      // where a pair would straddle a phase boundary (epilogue + sum reset sit in between)
      int lll = lll0;
      const bool ovl = bt_overlap_usable(c) != 0;
      halo_done = false;
      RcclLanded landed;          // .prepacked carries over from launch to launch within the phase
      while (lll <= last) {
        const bool odd = lll % 2 == 1;
        const bool both = odd && lll + 1 <= last;
        double wo[2], wm[2], wn[2];
        for (int x = 0; x < 2; x++) {
          const int l = both ? lll + x : lll;
          wo[x] = woa * l + wob;
          wn[x] = wna * l + wnb;
          wm[x] = 1. - wo[x] - wn[x];
        }
        if (!both && !odd) { wo[1] = wo[0]; wm[1] = wm[0]; wn[1] = wn[0]; }
        // single tile: the pair kernel applies the halo rule while loading; otherwise exchange first
        // RCCL tiles along i: the received strips stay in the transport's buffers and the kernel loads its
        // E/W rim from there (c->barotp_rimbuf), saving the unpack launch of every exchange
        const bool rimbuf = c->tiling.rccl && c->barotp_rimbuf && c->tiling.npy == 1 && !ovl && h.nreg != 2;
        // (single tile with the arctic patch: the pair kernel applies xctilr's seam rule while loading, before odd substeps)
        if ((arctic1 ? odd && c->tiling.multi() : c->tiling.multi()) && !halo_done) {
          if (rimbuf) { if (int rc = bt_pair_halo_landed(c, set, &landed)) return rc; }
          else if (int rc = bt_pair_halo(c, set)) return rc;
        }
        if (ovl) {
          // fork: outer tile columns + exchange of the new state on xstream, inner columns on stream; join
          HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
          HIPCHK(c, hipStreamWaitEvent(c->xstream, c->ev_fork, 0));
          bt_pair_launch(c, m, n, ml, nl, wo, wm, wn, odd ? 1 : 0, (both || !odd) ? 1 : 0, set, 1, nullptr);
          set ^= 1;
          c->halo_stream = c->xstream;
          const int rc = bt_pair_halo(c, set);
          c->halo_stream = nullptr;
          if (rc) return rc;
          HIPCHK(c, hipEventRecord(c->ev_join, c->xstream));
          HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
          halo_done = true;
        } else {
          bt_pair_launch(c, m, n, ml, nl, wo, wm, wn, odd ? 1 : 0, (both || !odd) ? 1 : 0, set, 0, rimbuf ? &landed : nullptr);
          set ^= 1;
        }
        if (!both) { const int ll = ml; ml = nl; nl = ll; }
        lll += both ? 2 : 1;
      }
    } else
    for (int lll = lll0; lll <= lll0 + lstep / 2 - 1; lll++) {
      BtArgs a;
      a.m = m; a.n = n; a.ml = ml; a.nl = nl;
      a.wo = woa * lll + wob;
      a.wn = wna * lll + wnb;
      a.wm = 1. - a.wo - a.wn;
      if (lll % 2 == 1) {
        {                                                                          // :395-397, one launch
          double *ptrs[3] = {h.f[F_pb_t], h.f[F_ubflx_t], h.f[F_vbflx_t]};
          const int nl3[3] = {2, 2, 2}, mh3[3] = {2, 2, 2}, nh3[3] = {2, 2, 3}, it3[3] = {1, 13, 14};
          if (int rc = st_xctilr_arctic_multi(c, 3, ptrs, nl3, mh3, nh3, it3)) return rc;
        }
        a.j0 = -1; a.j1 = jj + 2; a.i0 = -1; a.i1 = ii + 1; a.lv = 0;
        hipLaunchKernelGGL(k_bt_cont, g, b, 0, c->stream, c->d, a);
        a.j0 = -1; a.j1 = jj + 2; a.i0 = 0; a.i1 = ii + 1; a.lv = ml;
        hipLaunchKernelGGL(k_bt_umom, g, b, 0, c->stream, c->d, a);
        a.j0 = 0; a.j1 = jj + 2; a.i0 = 0; a.i1 = ii; a.lv = nl;
        hipLaunchKernelGGL(k_bt_vmom, g, b, 0, c->stream, c->d, a);
      } else {
        a.j0 = 0; a.j1 = jj + 1; a.i0 = 0; a.i1 = ii; a.lv = 0;
        hipLaunchKernelGGL(k_bt_cont, g, b, 0, c->stream, c->d, a);
        a.j0 = 1; a.j1 = jj + 1; a.i0 = 0; a.i1 = ii; a.lv = ml;
        hipLaunchKernelGGL(k_bt_vmom, g, b, 0, c->stream, c->d, a);
        a.j0 = 1; a.j1 = jj; a.i0 = 1; a.i1 = ii; a.lv = nl;
        hipLaunchKernelGGL(k_bt_umom, g, b, 0, c->stream, c->d, a);
      }
      const int ll = ml; ml = nl; nl = ll;
    }
    lll0 = lll0 + lstep / 2;
    // the epilogue reads pb_t(i-1,j), pb_t(i,j-1); the fused kernels write tile interiors only
    if (fused && !halo_done && !arctic1)
      if (int rc = bt_pair_halo(c, set)) return rc;
    halo_done = false;
    hipLaunchKernelGGL(k_bt_epilogue, g, b, 0, c->stream, c->d, nb, m, n, ml, nl, set);
  }
  HIPCHK(c, hipGetLastError());
  if (c->barotp_fused && !c->tiling.multi() && c->barotp_persist && bt_phase_usable(c)) return bt_phase_check(c);
  return 0;
}
