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

__global__ __launch_bounds__(64) void k_ale_vdifft(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  COL(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  const int kk = V.kk, ntr = V.ntr;
  const double cpi = 1. / SPCIFH, dtg = V.P.delt1 * GRAV, cc = GRAV * GRAV * V.P.delt1 / (ALPHA0 * ALPHA0);
  const double *dp = V.f[F_dp] + (size_t)nn * np;
  double *temp = V.f[F_temp] + (size_t)nn * np, *saln = V.f[F_saln] + (size_t)nn * np, *sigma = V.f[F_sigma] + (size_t)nn * np;
  double *gam = WK(V, 0);
  const double *nut = V.f[F_kdiff_t], *nus = V.f[F_kdiff_s];
  const double hfsw = V.f[F_sswflx][c], hfns = V.f[F_surflx][c] - hfsw, hfrs = V.f[F_surrlx][c];       // :86-93
  const double sfbr = V.f[F_brnflx][c], sfnb = V.f[F_salflx][c] - sfbr, sfrs = V.f[F_salrlx][c];
  auto fpbase = [&](int k) { return cc / fmax2(DPMIN_VDIFF, .5 * (L(dp, k - 1) + L(dp, k))); };         // :108-110
  // one tridiagonal solve: x in place, fp(k) = nu(k) * fpbase(k), flux(k) the surface-flux term of level k
  auto solve = [&](double *x, const double *nu, auto &&flux) {
    double bei = 1. / (L(dp, 1) + L(nu, 2) * fpbase(2));
    L(x, 1) = (L(dp, 1) * L(x, 1) - flux(1)) * bei;
    for (int k = 2; k <= kk - 1; k++) {
      const double fpk = L(nu, k) * fpbase(k), fpk1 = L(nu, k + 1) * fpbase(k + 1);
      const double g = -fpk * bei;
      L(gam, k) = g;
      bei = 1. / (L(dp, k) + fpk * (1. + g) + fpk1);
      const double rhs = L(dp, k) * L(x, k) - flux(k);
      L(x, k) = (rhs + fpk * L(x, k - 1)) * bei;
    }
    {
      const double fpk = L(nu, kk) * fpbase(kk);
      const double g = -fpk * bei;
      L(gam, kk) = g;
      bei = 1. / (L(dp, kk) + fpk * (1. + g));
      const double rhs = L(dp, kk) * L(x, kk) - flux(kk);
      L(x, kk) = (rhs + fpk * L(x, kk - 1)) * bei;
    }
    for (int k = kk - 1; k >= 1; k--) L(x, k) = L(x, k) - L(gam, k + 1) * L(x, k + 1);
  };
  const double *tns = V.f[F_t_ns_nonloc], *tsw = V.f[F_t_sw_nonloc], *trs = V.f[F_t_rs_nonloc];
  const double *snb = V.f[F_s_nb_nonloc], *sbr = V.f[F_s_br_nonloc], *srs = V.f[F_s_rs_nonloc];
  solve(temp, nut, [&](int k) {                                                                          // :112-139
    if (k == 1) return ((1. - L(tns, 2)) * hfns + (1. - L(tsw, 2)) * hfsw + (1. - L(trs, 2)) * hfrs) * dtg * cpi;
    return ((L(tns, k) - L(tns, k + 1)) * hfns + (L(tsw, k) - L(tsw, k + 1)) * hfsw + (L(trs, k) - L(trs, k + 1)) * hfrs) * dtg * cpi;
  });
  solve(saln, nus, [&](int k) {                                                                          // :141-168
    if (k == 1) return ((1. - L(snb, 2)) * sfnb + (1. - L(sbr, 2)) * sfbr + (1. - L(srs, 2)) * sfrs) * dtg;
    return ((L(snb, k) - L(snb, k + 1)) * sfnb + (L(sbr, k) - L(sbr, k + 1)) * sfbr + (L(srs, k) - L(srs, k + 1)) * sfrs) * dtg;
  });
  for (int nt = 0; nt < ntr; nt++) {                                                                     // :170-203 (nutrc = Kdiff_t)
    double *x = V.f[F_trc] + ((size_t)nn + (size_t)nt * 2 * kk) * np;
    const double tf = V.f[F_trflx][c + (size_t)nt * np];
    solve(x, nut, [&](int k) {
      if (k == 1) return (1. - L(snb, 2)) * tf * dtg;
      return (L(snb, k) - L(snb, k + 1)) * tf * dtg;
    });
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
    double *x = V.f[F_trc] + ((size_t)nn + (size_t)nt * 2 * kk) * np;
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
  const double *dp = (isv ? V.f[F_dpv] : V.f[F_dpu]) + (size_t)nn * np, *kv = V.f[F_kvisc_m];
  double *x = (isv ? V.f[F_v] : V.f[F_u]) + (size_t)nn * np;
  double *gam = WK(V, isv ? 1 : 0);
  auto fp = [&](int k) {
    const double nuv = .5 * (kv[mns + (size_t)(k - 1) * np] + L(kv, k));
    return nuv * (cc / fmax2(DPMIN_VDIFF, .5 * (L(dp, k - 1) + L(dp, k))));
  };
  double bei = 1. / (L(dp, 1) + fp(2));
  L(x, 1) = L(dp, 1) * L(x, 1) * bei;
  for (int k = 2; k <= kk - 1; k++) {
    const double fpk = fp(k), fpk1 = fp(k + 1);
    const double g = -fpk * bei;
    L(gam, k) = g;
    bei = 1. / (L(dp, k) + fpk * (1. + g) + fpk1);
    L(x, k) = (L(dp, k) * L(x, k) + fpk * L(x, k - 1)) * bei;
  }
  {
    const double fpk = fp(kk);
    const double g = -fpk * bei;
    L(gam, kk) = g;
    bei = 1. / (L(dp, kk) + fpk * (1. + g));
    L(x, kk) = (L(dp, kk) * L(x, kk) + fpk * L(x, kk - 1)) * bei;
  }
  for (int k = kk - 1; k >= 1; k--) L(x, k) = L(x, k) - L(gam, k + 1) * L(x, k + 1);
}

int st_ale_vdifft(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)n; (void)mm; (void)k1m; (void)k1n;
  const DevView &h = c->h;
  if (h.P.vcoord_tag == 1) return ctx_fail(c, "ale_vdifft: vcoord_type = 'isopyc_bulkml' has no ALE step (phy/mod_blom_step.F90:196-212)");
  hipLaunchKernelGGL(k_ale_vdifft, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, nn);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int st_ale_vdiffm(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
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
  const double *dp = V.f[F_dp] + (size_t)nn * np, *p = V.f[F_p];
  double *tsw = V.f[F_t_sw_nonloc], *sbr = V.f[F_s_br_nonloc], *buoyfl = V.f[F_buoyfl];
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
  (void)m; (void)n; (void)mm; (void)k1m; (void)k1n;
  const DevView &h = c->h;
  if (h.P.vcoord_tag == 1) return ctx_fail(c, "ale_forcing: vcoord_type = 'isopyc_bulkml' has no ALE step (phy/mod_blom_step.F90:196-212)");
  hipLaunchKernelGGL(k_ale_forcing, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, nn, c->swamxd, c->brine_mlbase_frac);
  HIPCHK(c, hipGetLastError());
  return 0;
}
