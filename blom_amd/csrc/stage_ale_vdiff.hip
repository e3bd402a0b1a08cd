// ale_vdifft, ale_vdiffm -- phy/mod_ale_vdiff.F90:50-243, :245-374: implicit vertical diffusion of T, S and the
// tracers with the surface fluxes and their non-local transport applied, and of the baroclinic velocities, for the vertical
// coordinates other than isopyc_bulkml.  The diffusivities Kdiff_t, Kdiff_s, Kvisc_m and the non-local fractions come from
// difest_vertical_hybrid (CVMix: not built), the fluxes from thermf (not built): they are inputs here, uploaded by name.
// One thread per column; a column is three (T, S, tracers) resp. one tridiagonal solve by forward elimination and back
// substitution, done in place on the fields with the elimination factors gam in a work plane.  The arithmetic is the
// reference's statement by statement (fpbase is recomputed from the same expression where the reference stores it).
// Parity: PINNED -- the module builds from the reference's own sources without stand-ins (oracle/Makefile *_vdf).
#include "blomgpu_internal.h"
#include "eos.h"

#define GRAV 9.806
#define SPCIFH 3990.
#define ALPHA0 1.e-3
#define DPMIN_VDIFF (0.1 * 9806.)

#define COL(V)                                                             \
  const int t_ = blockIdx.x * blockDim.x + threadIdx.x;                    \
  if (t_ >= (V).nplane) return;                                            \
  const int i = t_ % (V).ni - (NBDY - 1), j = t_ / (V).ni - (NBDY - 1);    \
  const size_t c = t_
#define L(a, k) (a)[c + (size_t)((k)-1) * np]

// T, S and up to NTB tracers of a column in ONE forward and ONE backward sweep.  The reference solves field after field (:112-203);
// the systems of T and of the tracers have the same matrix (nutrc = Kdiff_t), and no solve reads what another wrote, so the
// statements of the separate solves are carried out side by side, each exactly as it stands: fp(k) = nu(k) * fpbase(k) is formed
// once per matrix and level and carried to the next level (the reference forms it as fp(k+1) at level k and again as fp(k) at
// level k+1: the same product), bei and gam once per matrix.  What the sweeps read at the next VU levels is loaded before the
// recurrence of the current ones runs (the loads do not depend on it).  Tracers beyond NTB: further passes with the same code.
#define VU 4
template <int NTB, bool TS>
__device__ __forceinline__ void vdifft_pass(const DevView &V, size_t c, int nn, int nt0) {
  const size_t np = V.nplane;
  const int kk = V.kk;
  const double cpi = 1. / SPCIFH, dtg = V.P.delt1 * GRAV, cc = GRAV * GRAV * V.P.delt1 / (ALPHA0 * ALPHA0);
  gcd_t __restrict__ dp = V.f[F_dp] + (size_t)nn * np;
  gd_t __restrict__ temp = V.f[F_temp] + (size_t)nn * np, saln = V.f[F_saln] + (size_t)nn * np;
  gd_t __restrict__ gamT = WK(V, 0), gamS = WK(V, 1);
  gcd_t __restrict__ nut = V.f[F_kdiff_t], nus = V.f[F_kdiff_s];
  const double hfsw = V.f[F_sswflx][c], hfns = V.f[F_surflx][c] - hfsw, hfrs = V.f[F_surrlx][c];       // :86-93
  const double sfbr = V.f[F_brnflx][c], sfnb = V.f[F_salflx][c] - sfbr, sfrs = V.f[F_salrlx][c];
  gcd_t __restrict__ tns = V.f[F_t_ns_nonloc], tsw = V.f[F_t_sw_nonloc], trs = V.f[F_t_rs_nonloc];
  gcd_t __restrict__ snb = V.f[F_s_nb_nonloc], sbr = V.f[F_s_br_nonloc], srs = V.f[F_s_rs_nonloc];
  double *xtr[NTB > 0 ? NTB : 1];
  double tf[NTB > 0 ? NTB : 1];
#pragma unroll
  for (int q = 0; q < NTB; q++) {
    xtr[q] = V.f[F_trc] + ((size_t)nn + (size_t)(nt0 + q) * 2 * kk) * np;
    tf[q] = V.f[F_trflx][c + (size_t)(nt0 + q) * np];
  }
  // ---- level 1 (:112-120 and twins) -------------------------------------------------------------------------------------------
  double dpk = L(dp, 1), dpk1 = L(dp, 2);
  double fpb = cc / fmax2(DPMIN_VDIFF, .5 * (dpk + dpk1));                                               // fpbase(2), :108-110
  double fT = L(nut, 2) * fpb, fS = L(nus, 2) * fpb;                                                     // fp(2) of the two matrices
  double beiT = 1. / (dpk + fT), beiS = 1. / (dpk + fS);
  double n_tns = L(tns, 2), n_tsw = L(tsw, 2), n_trs = L(trs, 2), n_snb = L(snb, 2), n_sbr = L(sbr, 2), n_srs = L(srs, 2);   // level k+1's
  double xT = 0., xS = 0., xq[NTB > 0 ? NTB : 1];
  if (TS) {
    xT = (dpk * L(temp, 1) - ((1. - n_tns) * hfns + (1. - n_tsw) * hfsw + (1. - n_trs) * hfrs) * dtg * cpi) * beiT;
    L(temp, 1) = xT;
    xS = (dpk * L(saln, 1) - ((1. - n_snb) * sfnb + (1. - n_sbr) * sfbr + (1. - n_srs) * sfrs) * dtg) * beiS;
    L(saln, 1) = xS;
  }
#pragma unroll
  for (int q = 0; q < NTB; q++) {
    xq[q] = (dpk * L(xtr[q], 1) - (1. - n_snb) * tf[q] * dtg) * beiT;
    L(xtr[q], 1) = xq[q];
  }
  // ---- levels 2 .. kk ---------------------------------------------------------------------------------------------------------
  dpk = dpk1;
  for (int k0 = 2; k0 <= kk; k0 += VU) {
    double a_dp[VU], a_nt[VU], a_ns[VU], a_tns[VU], a_tsw[VU], a_trs[VU], a_snb[VU], a_sbr[VU], a_srs[VU], a_t[VU], a_s[VU];
    double a_x[NTB > 0 ? NTB : 1][VU];
#pragma unroll
    for (int u = 0; u < VU; u++) {
      const int k = k0 + u <= kk ? k0 + u : kk, k1 = k + 1 <= kk ? k + 1 : kk;       // level k+1 does not exist for k = kk: not used there
      a_dp[u] = L(dp, k1); a_nt[u] = L(nut, k1); a_ns[u] = L(nus, k1);
      a_snb[u] = L(snb, k + 1);
      if (TS) {
        a_tns[u] = L(tns, k + 1); a_tsw[u] = L(tsw, k + 1); a_trs[u] = L(trs, k + 1); a_sbr[u] = L(sbr, k + 1); a_srs[u] = L(srs, k + 1);
        a_t[u] = L(temp, k); a_s[u] = L(saln, k);
      }
#pragma unroll
      for (int q = 0; q < NTB; q++) a_x[q][u] = L(xtr[q], k);
    }
#pragma unroll
    for (int u = 0; u < VU; u++) {
      const int k = k0 + u;
      if (k > kk) break;
      const double c_tns = n_tns, c_tsw = n_tsw, c_trs = n_trs, c_snb = n_snb, c_sbr = n_sbr, c_srs = n_srs;   // level k's
      n_snb = a_snb[u];
      if (TS) { n_tns = a_tns[u]; n_tsw = a_tsw[u]; n_trs = a_trs[u]; n_sbr = a_sbr[u]; n_srs = a_srs[u]; }
      const double gT = -fT * beiT, gS = -fS * beiS;
      L(gamT, k) = gT;
      if (TS) L(gamS, k) = gS;
      double fT1 = 0., fS1 = 0.;
      if (k < kk) {
        dpk1 = a_dp[u];
        fpb = cc / fmax2(DPMIN_VDIFF, .5 * (dpk + dpk1));
        fT1 = a_nt[u] * fpb; fS1 = a_ns[u] * fpb;
        beiT = 1. / (dpk + fT * (1. + gT) + fT1);
        beiS = 1. / (dpk + fS * (1. + gS) + fS1);
      } else {
        beiT = 1. / (dpk + fT * (1. + gT));
        beiS = 1. / (dpk + fS * (1. + gS));
      }
      if (TS) {
        const double rT = dpk * a_t[u] - ((c_tns - n_tns) * hfns + (c_tsw - n_tsw) * hfsw + (c_trs - n_trs) * hfrs) * dtg * cpi;
        xT = (rT + fT * xT) * beiT;
        L(temp, k) = xT;
        const double rS = dpk * a_s[u] - ((c_snb - n_snb) * sfnb + (c_sbr - n_sbr) * sfbr + (c_srs - n_srs) * sfrs) * dtg;
        xS = (rS + fS * xS) * beiS;
        L(saln, k) = xS;
      }
#pragma unroll
      for (int q = 0; q < NTB; q++) {
        const double r = dpk * a_x[q][u] - (c_snb - n_snb) * tf[q] * dtg;
        xq[q] = (r + fT * xq[q]) * beiT;
        L(xtr[q], k) = xq[q];
      }
      fT = fT1; fS = fS1; dpk = dpk1;
    }
  }
  // ---- back substitution, k = kk-1 .. 1 ---------------------------------------------------------------------------------------
  for (int k0 = kk - 1; k0 >= 1; k0 -= VU) {
    double b_gT[VU], b_gS[VU], b_t[VU], b_s[VU], b_x[NTB > 0 ? NTB : 1][VU];
#pragma unroll
    for (int u = 0; u < VU; u++) {
      const int k = k0 - u >= 1 ? k0 - u : 1;
      b_gT[u] = L(gamT, k + 1);
      if (TS) { b_gS[u] = L(gamS, k + 1); b_t[u] = L(temp, k); b_s[u] = L(saln, k); }
#pragma unroll
      for (int q = 0; q < NTB; q++) b_x[q][u] = L(xtr[q], k);
    }
#pragma unroll
    for (int u = 0; u < VU; u++) {
      const int k = k0 - u;
      if (k < 1) break;
      if (TS) {
        xT = b_t[u] - b_gT[u] * xT; L(temp, k) = xT;
        xS = b_s[u] - b_gS[u] * xS; L(saln, k) = xS;
      }
#pragma unroll
      for (int q = 0; q < NTB; q++) { xq[q] = b_x[q][u] - b_gT[u] * xq[q]; L(xtr[q], k) = xq[q]; }
    }
  }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k_ale_vdifft(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  COL(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  const int kk = V.kk, ntr = V.ntr;
  gcd_t dp = V.f[F_dp] + (size_t)nn * np;
  gd_t temp = V.f[F_temp] + (size_t)nn * np, saln = V.f[F_saln] + (size_t)nn * np, sigma = V.f[F_sigma] + (size_t)nn * np;
  // T and S with the first tracers, then the remaining tracers four at a time
  const int n0 = ntr < 4 ? ntr : 4;
  switch (n0) {
    case 0: vdifft_pass<0, true>(V, c, nn, 0); break;
    case 1: vdifft_pass<1, true>(V, c, nn, 0); break;
    case 2: vdifft_pass<2, true>(V, c, nn, 0); break;
    case 3: vdifft_pass<3, true>(V, c, nn, 0); break;
    default: vdifft_pass<4, true>(V, c, nn, 0); break;
  }
  for (int nt0 = 4; nt0 < ntr; nt0 += 4) {
    const int nb = ntr - nt0 < 4 ? ntr - nt0 : 4;
    switch (nb) {
      case 1: vdifft_pass<1, false>(V, c, nn, nt0); break;
      case 2: vdifft_pass<2, false>(V, c, nn, nt0); break;
      case 3: vdifft_pass<3, false>(V, c, nn, nt0); break;
      default: vdifft_pass<4, false>(V, c, nn, nt0); break;
    }
  }
  double sc = V.f[F_salt_corr][c];                                                                      // :205-222
  for (int k = 1; k <= kk; k++) {
    const double s1 = L(saln, k);
    sc = sc - fmin2(0., s1) * L(dp, k) / GRAV;
    const double sn = fmax2(0., s1);
    L(saln, k) = sn;
    L(sigma, k) = eos::sig(V.P, L(temp, k), sn);
  }
  V.f[F_salt_corr][c] = sc;
  for (int nt = 0; nt < ntr; nt++) {
    gd_t x = V.f[F_trc] + ((size_t)nn + (size_t)nt * 2 * kk) * np;
    double tc = V.f[F_trc_corr][c + (size_t)nt * np];
    for (int k = 1; k <= kk; k++) {
      const double x1 = L(x, k);
      tc = tc - fmin2(0., x1) * L(dp, k) / GRAV;
      L(x, k) = fmax2(0., x1);
    }
    V.f[F_trc_corr][c + (size_t)nt * np] = tc;
  }
}

// blockIdx.y = 0: u-columns, 1: v-columns, :258-356
__global__ __launch_bounds__(64) void k_ale_vdiffm(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  COL(V);
  const bool isv = blockIdx.y == 1;
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !(isv ? V.m[I_iv][c] : V.m[I_iu][c])) return;
  const size_t np = V.nplane, mns = isv ? c - V.ni : c - 1;
  const int kk = V.kk;
  const double cc = GRAV * GRAV * V.P.delt1 / (ALPHA0 * ALPHA0);
  gcd_t dp = (isv ? V.f[F_dpv] : V.f[F_dpu]) + (size_t)nn * np, kv = V.f[F_kvisc_m];
  gd_t x = (isv ? V.f[F_v] : V.f[F_u]) + (size_t)nn * np;
  gd_t gam = WK(V, isv ? 1 : 0);
  // fp(k) = nuv(k) * fpbase(k) is formed once per level and carried to the next one (the reference forms it as fp(k+1) at level k
  // and again as fp(k) at level k+1: the same product); the next four levels' inputs are loaded ahead
  auto fpv = [&](double kva, double kvb, double dpa, double dpb) {
    const double nuv = .5 * (kva + kvb);
    return nuv * (cc / fmax2(DPMIN_VDIFF, .5 * (dpa + dpb)));
  };
  double dpk = L(dp, 1), dpk1 = L(dp, 2);
  double f = fpv(kv[mns + np], L(kv, 2), dpk, dpk1);             // fp(2)
  double bei = 1. / (dpk + f);
  double xp = dpk * L(x, 1) * bei;
  L(x, 1) = xp;
  dpk = dpk1;
  for (int k0 = 2; k0 <= kk; k0 += 4) {
    double a_dp[4], a_ka[4], a_kb[4], a_x[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int k = k0 + u <= kk ? k0 + u : kk, k1 = k + 1 <= kk ? k + 1 : kk;
      a_dp[u] = L(dp, k1); a_ka[u] = kv[mns + (size_t)(k1 - 1) * np]; a_kb[u] = L(kv, k1); a_x[u] = L(x, k);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int k = k0 + u;
      if (k > kk) break;
      const double g = -f * bei;
      L(gam, k) = g;
      double f1 = 0.;
      if (k < kk) {
        dpk1 = a_dp[u];
        f1 = fpv(a_ka[u], a_kb[u], dpk, dpk1);
        bei = 1. / (dpk + f * (1. + g) + f1);
      } else {
        bei = 1. / (dpk + f * (1. + g));
      }
      xp = (dpk * a_x[u] + f * xp) * bei;
      L(x, k) = xp;
      f = f1; dpk = dpk1;
    }
  }
  for (int k0 = kk - 1; k0 >= 1; k0 -= 4) {
    double b_g[4], b_x[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int k = k0 - u >= 1 ? k0 - u : 1; b_g[u] = L(gam, k + 1); b_x[u] = L(x, k); }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int k = k0 - u;
      if (k < 1) break;
      xp = b_x[u] - b_g[u] * xp;
      L(x, k) = xp;
    }
  }
}

int st_ale_vdifft(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  TimeScope ts(c, "ale_vdiff");
  (void)m; (void)n; (void)mm; (void)k1m; (void)k1n;
  const DevView &h = c->h;
  if (h.P.vcoord_tag == 1) return ctx_fail(c, "ale_vdifft: vcoord_type = 'isopyc_bulkml' has no ALE step (phy/mod_blom_step.F90:196-212)");
  hipLaunchKernelGGL(k_ale_vdifft, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, nn);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int st_ale_vdiffm(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  TimeScope ts(c, "ale_vdiff");
  (void)m; (void)n; (void)mm; (void)k1m; (void)k1n;
  const DevView &h = c->h;
  if (h.P.vcoord_tag == 1) return ctx_fail(c, "ale_vdiffm: vcoord_type = 'isopyc_bulkml' has no ALE step (phy/mod_blom_step.F90:196-212)");
  if (int rc = st_xctilr(c, h.f[F_kvisc_m], 1, h.kk, 1, 1, 1)) return rc;                               // :256
  hipLaunchKernelGGL(k_ale_vdiffm, plane_grid(h, 2, 64), dim3(64), 0, c->stream, c->d, nn);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ---- ale_forcing, phy/mod_ale_forcing.F90:45-221: the fractions of the shortwave and of the brine flux that pass the layer
//      interfaces, and the buoyancy flux at the interfaces.  Inputs by name: swfc1, swfc2, swal1, swal2 (the two-band absorption
//      of mod_swabs -- netCDF-bound, hence uploaded), mld (mod_cmnfld), the surface fluxes; scalars swamxd, brine_mlbase_frac.
//      Parity: cross-checked against the real module built against a stand-in for mod_swabs (oracle/xcheck) -- not a pin.
#include "exp_libm.h"
#define ONEM 9806.
#define ONEMU .009806
__global__ __launch_bounds__(64) void k_ale_forcing(const DevView *__restrict__ Vp, int nn, double swamxd, double brine_mlbase_frac) {
  const DevView &V = *Vp;
  COL(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  const int kk = V.kk;
  // cbra1 = 2**(1/3), cbra2 = cbra1*cbra1/12 as the reference's compiler folds them (:56-57)
  const double cbra1 = __longlong_as_double(0x3FF428A2F98D728BLL), cbra2 = __longlong_as_double(0x3FC0EEA9C37E497ELL);
  const double cpi = 1. / SPCIFH, gaa = GRAV * ALPHA0 * ALPHA0;
  gcd_t dp = V.f[F_dp] + (size_t)nn * np, p = V.f[F_p];
  gd_t tsw = V.f[F_t_sw_nonloc], sbr = V.f[F_s_br_nonloc], buoyfl = V.f[F_buoyfl];
  {                                                                                    // :66-107 shortwave
    const double pmax = swamxd * ONEM;
    const double lei1 = 1. / (V.f[F_swal1][c] * ONEM), lei2 = 1. / (V.f[F_swal2][c] * ONEM);
    const double fc1 = V.f[F_swfc1][c], fc2 = V.f[F_swfc2][c];
    int kmax = 1;
    L(tsw, 1) = 1.;
    for (int k = 1; k <= kk; k++) {
      if (L(dp, k) > ONEMU) {
        L(tsw, k + 1) = fc1 * exp_libm(-lei1 * fmin2(pmax, L(p, k + 1))) + fc2 * exp_libm(-lei2 * fmin2(pmax, L(p, k + 1)));
        kmax = k;
      } else L(tsw, k + 1) = L(tsw, k);
      if (L(p, k + 1) > pmax) break;
    }
    const double pmaxi = 1. / fmin2(pmax, L(p, kmax + 1)), nlbot = L(tsw, kmax + 1);
    for (int k = kmax + 1; k <= kk + 1; k++) L(tsw, k) = 0.;
    for (int k = kmax; k >= 2; k--) {
      if (L(dp, k) > ONEMU) L(tsw, k) = L(tsw, k) - nlbot * L(p, k) * pmaxi;
      else L(tsw, k) = L(tsw, k + 1);
    }
  }
  {                                                                                    // :113-161 brine
    const double mld = V.f[F_mld][c];
    const double lei = 1. / (mld * ONEM), pmax = cbra1 * mld * ONEM;
    int kmax = 1;
    L(sbr, 1) = 1.;
    for (int k = 1; k <= kk; k++) {
      if (L(dp, k) > ONEMU) {
        const double q = fmin2(cbra1, lei * L(p, k + 1));
        const double q_c = q / cbra1, q3 = q * q * q, q_c3 = q_c * q_c * q_c;
        L(sbr, k + 1) = brine_mlbase_frac * (1. - cbra2 * q * q3 * (7. - 2. * q3)) +
                        (1. - brine_mlbase_frac) * (1. - q + q_c3 * q_c3 * (6. * cbra1 - 7. - (5. * cbra1 - 6.) * q_c));
        kmax = k;
      } else L(sbr, k + 1) = L(sbr, k);
      if (L(p, k + 1) > pmax) break;
    }
    const double pmaxi = 1. / fmin2(pmax, L(p, kmax + 1)), nlbot = L(sbr, kmax + 1);
    for (int k = kmax + 1; k <= kk + 1; k++) L(sbr, k) = 0.;
    for (int k = kmax; k >= 2; k--) {
      if (L(dp, k) > ONEMU) L(sbr, k) = L(sbr, k) - nlbot * L(p, k) * pmaxi;
      else L(sbr, k) = L(sbr, k + 1);
    }
  }
  {                                                                                    // :167-199 buoyancy flux
    using namespace eos;
    // the coefficients of potential density referenced at the surface, phy/mod_eos.F90:118-129
    const double ap210 = a21, ap220 = a22, ap230 = a23, ap240 = a24, ap250 = a25, ap260 = a26;
    const double ap110 = a11 - ap210 / ALPHA0, ap120 = a12 - ap220 / ALPHA0, ap130 = a13 - ap230 / ALPHA0;
    const double ap140 = a14 - ap240 / ALPHA0, ap150 = a15 - ap250 / ALPHA0, ap160 = a16 - ap260 / ALPHA0;
    const double th = V.f[F_temp][c + (size_t)nn * np], s = V.f[F_saln][c + (size_t)nn * np];
    const double r1 = ap110 + (ap120 + ap140 * th + ap150 * s) * th + (ap130 + ap160 * s) * s;           // dsigdt0, dsigds0 :263-343
    const double r2i = 1. / (ap210 + (ap220 + ap240 * th + ap250 * s) * th + (ap230 + ap260 * s) * s);
    const double dsgdt = (ap120 + 2. * ap140 * th + ap150 * s - (ap220 + 2. * ap240 * th + ap250 * s) * r1 * r2i) * r2i;
    const double dsgds = (ap130 + ap150 * th + 2. * ap160 * s - (ap230 + ap250 * th + 2. * ap260 * s) * r1 * r2i) * r2i;
    const double hf = V.f[F_surflx][c], hfsw = V.f[F_sswflx][c], sf = V.f[F_salflx][c], sfbr = V.f[F_brnflx][c];
    L(buoyfl, 1) = -(dsgdt * hf * cpi + dsgds * sf) * gaa;
    for (int k = 2; k <= kk + 1; k++) L(buoyfl, k) = -(dsgdt * L(tsw, k) * hfsw * cpi + dsgds * L(sbr, k) * sfbr) * gaa;
  }
}

int st_ale_forcing(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  TimeScope ts(c, "ale_forcing");
  (void)m; (void)n; (void)mm; (void)k1m; (void)k1n;
  const DevView &h = c->h;
  if (h.P.vcoord_tag == 1) return ctx_fail(c, "ale_forcing: vcoord_type = 'isopyc_bulkml' has no ALE step (phy/mod_blom_step.F90:196-212)");
  hipLaunchKernelGGL(k_ale_forcing, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, nn, c->swamxd, c->brine_mlbase_frac);
  HIPCHK(c, hipGetLastError());
  return 0;
}
