// thermf -- surface fluxes of heat, salt and tracers; for the idealised channel: channel/mod_thermf_channel.F90:56-374
// (phy/mod_thermf.F90:35-65 selects by expcnf; called between diapfl and mxlayr, phy/mod_blom_step.F90:185).
//
//   k_thermf_channel_flux   :98-263 per p-point: fresh water, virtual salt and heat fluxes from the forcing fields (swa, nsf, eva,
//                           lip, sop, rnf, rfi), tracer fluxes, relaxation towards the sea surface temperature / salinity
//                           climatologies (cubic interpolation in time, phy/mod_intp1d.F90), friction velocity
//   xcsum x 2               :272-273 the global sums of the virtual salt flux and the fresh water flux, in the reference's
//                           summation order (halo.hip: k_xcsum), left ON THE DEVICE: the sequence stays capturable
//   k_thermf_channel_corr   :275-343 the correction that makes the virtual salt flux globally consistent with the reference
//                           salinity, unit conversion
// Not built (fail loudly): the diagnosed relaxation fluxes (aptflx, apsflx, ditflx, disflx: four 48-level arrays), the balanced
// salinity relaxation (srxbal: needs the world ocean mask).  The surface flux of the generic length scale (use_GLS, :167-175) takes its
// one real power of a field value, trc(tke)**gls_m, from pow_libm.h; the others are powers of PARAMETERs, folded by the host's pow.
// On decomposed domains the two sums gather the plane (halo.hip: xcsum_group, comm_rccl.hip: rccl_xcsum_dev).
// Roofline: HBM, ~20 two-dimensional planes.
#include "blomgpu_internal.h"
#include "pow_libm.h"
#include <cmath>

#define PLANE_IJ(V)                                                        \
  unsigned bx_, by_;                                                       \
  xcd_block(bx_, by_);                                                     \
  const int t_ = bx_ * blockDim.x + threadIdx.x;                           \
  (void)by_;                                                               \
  if (t_ >= (V).nplane) return;                                            \
  const int i = t_ % (V).ni - (NBDY - 1), j = t_ / (V).ni - (NBDY - 1);    \
  const size_t c = t_;                                                     \
  (void)i; (void)j; (void)c

#define SPCIFH 3990.
#define T0DEG 273.15
#define EPSILT 1.e-11
#define ONEM 9806.
#define S2_VRTSFL 8      // 2-D work plane: the reference's local array vrtsfl

struct ThermfPar {
  double trxday, srxday, trxdpt, srxdpt, trxlim, srxlim, xmi, sref, area;
  int l1mi, l2mi, l3mi, l4mi, l5mi;
  int use_gls;                   // use_GLS: gls_cmu0**gls_p, vonKar**gls_n, Zos**(gls_n-1.) (phy/mod_tke.F90:36-58: .527, 3., .4, -1., .0002)
  double gls_c, gls_vk, gls_z;
};

// phy/mod_intp1d.F90:29-52
__device__ inline double intp1d(double d1, double d2, double d3, double d4, double d5, double x) {
  const double a1 = -3. / 7., a2 = -15. / 7., a3 = 3. / 2., b1 = 4. / 7., b2 = -16. / 7., b3 = 15. / 7., b4 = -5. / 7., b5 = 2. / 7.,
               c1 = -1. / 7., c2 = 9. / 14;
  const double a = a1 * (d1 + d5) + a2 * d3 + a3 * (d2 + d4);
  const double b = b1 * d1 + b2 * d2 + b3 * d3 + b4 * d4 + b5 * d5;
  const double cc = c1 * (d1 + d4) + c2 * (d2 + d3);
  return (a * x + b) * x + cc;
}

__global__ void k_thermf_channel_flux(const DevView *__restrict__ Vp, ThermfPar T, int nn, int k1n) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane, o1 = c + (size_t)nn * np, o2 = o1 + np, ok1 = c + (size_t)(k1n - 1) * np;
  const int kk = V.kk, ntr = V.ntr;
  gcd_t dp = V.f[F_dp], temp = V.f[F_temp], saln = V.f[F_saln];
  // the ocean top layer and the mixed layer, :103-116
  const double sotl = saln[ok1];
  const double dpmxl = dp[o1] + dp[o2];
  const double hmxl = dpmxl / ONEM;
  const double tmxl = (temp[o1] * dp[o1] + temp[o2] * dp[o2]) / dpmxl + T0DEG;
  const double smxl = (saln[o1] * dp[o1] + saln[o2] * dp[o2]) / dpmxl;
  // fresh water, salt and heat fluxes, :129-159
  V.f[F_fmltfz][c] = 0.;
  const double fwflx = V.f[F_eva][c] + V.f[F_lip][c] + V.f[F_sop][c] + V.f[F_rnf][c] + V.f[F_rfi][c] + 0.;
  V.f[F_sfl][c] = 0.;
  V.f[F_brnflx][c] = 0.;
  const double vrtsfl = -sotl * fwflx * 1.e-3;
  WK2(V, S2_VRTSFL)[c] = vrtsfl;
  const double scp2 = V.f[F_scp2][c];
  V.f[F_util1][c] = vrtsfl * scp2;
  V.f[F_util2][c] = fwflx * scp2;
  V.f[F_hmltfz][c] = 0.;
  V.f[F_surflx][c] = -(V.f[F_swa][c] + V.f[F_nsf][c] + 0.) * 1.e-4;
  V.f[F_sswflx][c] = 0.;
  // tracer fluxes (positive downwards), :165-197
  for (int nt = 0; nt < ntr; nt++) {
    gd_t trflx = V.f[F_trflx] + c + (size_t)nt * np;
    if (V.P.itrtke >= 1 && T.use_gls && nt + 1 == V.P.itrgls) {                  // :167-175 (-gls_n = 1)
      *trflx = V.f[F_difdia][c] * T.gls_c * pow_libm(V.f[F_trc][ok1 + (size_t)(V.P.itrtke - 1) * 2 * kk * np], 1.5) * T.gls_vk * T.gls_z;
      continue;
    }
    if (V.P.itrtke >= 1 && (nt + 1 == V.P.itrtke || nt + 1 == V.P.itrgls)) { *trflx = 0.; continue; }
    *trflx = -V.f[F_trc][ok1 + (size_t)nt * 2 * kk * np] * fwflx * 1.e-3;
  }
  // relaxation fluxes, :203-255
  double surrlx = 0.;
  if (T.trxday > EPSILT) {
    gcd_t sstclm = V.f[F_sstclm] + c, ricclm = V.f[F_ricclm] + c;
    double sstc = intp1d(sstclm[(size_t)(T.l1mi - 1) * np], sstclm[(size_t)(T.l2mi - 1) * np], sstclm[(size_t)(T.l3mi - 1) * np],
                         sstclm[(size_t)(T.l4mi - 1) * np], sstclm[(size_t)(T.l5mi - 1) * np], T.xmi);
    const double rice = intp1d(ricclm[(size_t)(T.l1mi - 1) * np], ricclm[(size_t)(T.l2mi - 1) * np], ricclm[(size_t)(T.l3mi - 1) * np],
                               ricclm[(size_t)(T.l4mi - 1) * np], ricclm[(size_t)(T.l5mi - 1) * np], T.xmi);
    sstc = (1. - rice) * sstc;
    const double trxflx = SPCIFH * 100. * fmin2(hmxl, T.trxdpt) / (T.trxday * 86400.) * fmin2(T.trxlim, fmax2(-T.trxlim, sstc - tmxl));
    surrlx = -trxflx;
  }
  V.f[F_surrlx][c] = surrlx;
  double salrlx = 0.;
  if (T.srxday > EPSILT) {
    gcd_t sssclm = V.f[F_sssclm] + c;
    const double sssc = intp1d(sssclm[(size_t)(T.l1mi - 1) * np], sssclm[(size_t)(T.l2mi - 1) * np], sssclm[(size_t)(T.l3mi - 1) * np],
                               sssclm[(size_t)(T.l4mi - 1) * np], sssclm[(size_t)(T.l5mi - 1) * np], T.xmi);
    const double srxflx = 100. * fmin2(hmxl, T.srxdpt) / (T.srxday * 86400.) * fmin2(T.srxlim, fmax2(-T.srxlim, sssc - smxl));
    salrlx = -srxflx;
    V.f[F_util3][c] = fmax2(0., salrlx) * scp2;
    V.f[F_util4][c] = fmin2(0., salrlx) * scp2;
  }
  V.f[F_salrlx][c] = salrlx;
  // friction velocity, :261
  V.f[F_ustar][c] = V.f[F_ustarw][c] * 1.e2;
}

// :275-343; sums[0] = totsfl, sums[1] = totwfl
__global__ void k_thermf_channel_corr(const DevView *__restrict__ Vp, ThermfPar T, const double *__restrict__ sums) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  const double sflxc = (-T.sref * sums[1] * 1.e-3 - sums[0]) / T.area;
  V.f[F_salflx][c] = -(WK2(V, S2_VRTSFL)[c] + sflxc + V.f[F_sfl][c]) * 1.e2;
  V.f[F_brnflx][c] = -V.f[F_brnflx][c] * 1.e2;
  const double trflxc = 0.;
  for (int nt = 0; nt < V.ntr; nt++) {
    if (V.P.itrtke >= 1 && (nt + 1 == V.P.itrtke || nt + 1 == V.P.itrgls)) continue;
    gd_t trflx = V.f[F_trflx] + c + (size_t)nt * np;
    *trflx = -(*trflx + trflxc) * 1.e2;
  }
}

int st_thermf(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)n; (void)mm; (void)k1m;
  const DevView &h = c->h;
  // phy/mod_thermf.F90:43-63
  if (c->expcnf == "fuk95" || c->expcnf == "noforcing" || c->expcnf == "isomip1" || c->expcnf == "isomip2") return 0;   // empty cases
  if (c->expcnf != "channel") {
    if (c->expcnf == "cesm" || c->expcnf == "ben02clim" || c->expcnf == "ben02syn" || c->expcnf == "single_column")
      return ctx_fail(c, "thermf: expcnf = " + c->expcnf + " is not built on the device");
    return ctx_fail(c, " thermf: expcnf = " + c->expcnf + " is unsupported!");
  }
  if (c->aptflx || c->apsflx || c->ditflx || c->disflx)
    return ctx_fail(c, "thermf_channel: applying / diagnosing relaxation fluxes (aptflx, apsflx, ditflx, disflx) is not built on the device");
  if (c->srxday > EPSILT && c->srxbal) return ctx_fail(c, "thermf_channel: the balanced salinity relaxation (srxbal) is not built on the device");
  ThermfPar T;
  T.trxday = c->trxday; T.srxday = c->srxday; T.trxdpt = c->trxdpt; T.srxdpt = c->srxdpt; T.trxlim = c->trxlim; T.srxlim = c->srxlim;
  T.xmi = c->xmi; T.sref = c->sref; T.area = c->area;
  T.use_gls = h.P.itrtke >= 1 && h.P.gls ? 1 : 0;
  T.gls_c = std::pow(.527, 3.); T.gls_vk = std::pow(.4, -1.); T.gls_z = std::pow(.0002, -1. - 1.);
  T.l1mi = c->lmi[0]; T.l2mi = c->lmi[1]; T.l3mi = c->lmi[2]; T.l4mi = c->lmi[3]; T.l5mi = c->lmi[4];
  if (c->area <= 0.) return ctx_fail(c, "thermf_channel: the ocean area (mod_grid: area) has not been set");
  TimeScope ts(c, "thermf");
  hipLaunchKernelGGL(k_thermf_channel_flux, plane_grid(h), dim3(256), 0, c->stream, c->d, T, nn, k1n);
  double *sums = nullptr;
  if (int rc = st_xcsum_dev(c, h.f[F_util1], 1, 0, &sums)) return rc;
  if (int rc = st_xcsum_dev(c, h.f[F_util2], 1, 1, &sums)) return rc;
  hipLaunchKernelGGL(k_thermf_channel_corr, plane_grid(h), dim3(256), 0, c->stream, c->d, T, sums);
  HIPCHK(c, hipGetLastError());
  c->ntda++;                                                                                    // :350
  return 0;
}
