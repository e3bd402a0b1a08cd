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
