// eddtra for the vertical coordinates other than isopyc_bulkml -- phy/mod_eddtra.F90:1001-1739 (eddtra_ale) and the
// driver's heat and salt components (:1859-1901).  SURVEY.md 8 row f3.
//
// Gent-McWilliams mass fluxes at the layer interfaces below the mixed layer base (kappa * neutral slope), tapered linearly
// to zero at the surface inside it, plus the submesoscale restratification of Fox-Kemper et al. (2008) with a prescribed
// vertical structure inside the mixed layer; the sum is limited by the alternating-sweep procedure of the isopycnic form
// so that no cell loses more than ffac of its mass, and split back into its two parts.
//   k_eda_mixed_layer  per p-point: the two-time-scale running means of boundary and mixed layer depth (rmeanfilt, :121-151;
//                      state hbl_tf, hml_tf1, hml_tf), the bounded mixed layer depth hml_tfbnd and the mixed layer's mean
//                      potential density (sig0) into util1, as the reference
//   k_eda_column       one thread per velocity-point column (blockIdx.y = 0: u, 1: v -- mirror images); the column's private
//                      arrays puv, mflgm, mflsm, mfl, dlm, dlp lie wavefront-major in the work space (level k of the 64 columns
//                      of a wavefront in six consecutive rows of 64 doubles); upsilon is evaluated in the column
//   k_eda_ts           utfltd, utflsm, usfltd, usflsm and the v twins
// mlrmth = 'bod23' needs ustar3 / wstar3 of the CVMix-bound mod_difest and is refused.  Non-convergence of the limiter and a
// violated final bound are the reference's xchalt exits and come back as errors.  Roofline: HBM.
// Parity: cross-checked against the reference's REAL mod_eddtra compiled against a stand-in for mod_difest that holds only
// OBLdepth (oracle/Makefile *_xale, tests/test_xcheck_eddtra_ale.py) -- a cross-check, not a pin.
#include "blomgpu_internal.h"
#include "eos.h"

#define GRAV 9.806
#define RHO0 1.e3
#define ALPHA0 1.e-3
#define ONEM 9806.
#define EPSILP 1.e-12
#define DBCL82 .0003          /* phy/mod_cmnfld.F90:48 */

#define THREAD_IJ(V)                                                       \
  unsigned bx_, by_;                                                       \
  xcd_block(bx_, by_);                                                     \
  const int t_ = bx_ * blockDim.x + threadIdx.x;                           \
  (void)by_;                                                               \
  if (t_ >= (V).nplane) return;                                            \
  const int i = t_ % (V).ni - (NBDY - 1), j = t_ / (V).ni - (NBDY - 1);    \
  const size_t c = t_

struct EdaPar {
  int mlrmth;
  double ce, tau_mlr, wf_growing_hbl, wf_decaying_hbl, wf_growing_hml, wf_decaying_hml, lfmin, mlbl_max_ratio;
};

__device__ inline void rmeanfilt(double &filtered, double signal, double wg, double wd) {   // :121-151
  const double wf = signal >= filtered ? wg : wd;
  filtered = wf * filtered + (1. - wf) * signal;
}

// :1050-1123 (mlrmth = 'fox08')
__global__ void k_eda_mixed_layer(const DevView *__restrict__ Vp, int nn, EdaPar Q) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  const int kk = V.kk;
  const double hbl = V.f[F_OBLdepth][c];
  double hbl_tf = V.f[F_hbl_tf][c], hml_tf1 = V.f[F_hml_tf1][c], hml_tf = V.f[F_hml_tf][c];
  rmeanfilt(hbl_tf, hbl, Q.wf_growing_hbl, Q.wf_decaying_hbl);
  rmeanfilt(hml_tf1, V.f[F_mld][c], Q.wf_growing_hbl, Q.wf_decaying_hbl);
  rmeanfilt(hml_tf, hml_tf1, Q.wf_growing_hml, Q.wf_decaying_hml);
  const double hml_tfbnd = fmin2(hml_tf, Q.mlbl_max_ratio * hbl_tf);
  V.f[F_hbl_tf][c] = hbl_tf;
  V.f[F_hml_tf1][c] = hml_tf1;
  V.f[F_hml_tf][c] = hml_tf;
  V.f[F_hml_tfbnd][c] = hml_tfbnd;
  // vertically averaged mixed layer density, :1103-1123
  const double *p = V.f[F_p], *dp = V.f[F_dp] + (size_t)nn * np, *temp = V.f[F_temp] + (size_t)nn * np, *saln = V.f[F_saln] + (size_t)nn * np;
  const double p1 = p[c];
  const double pml = fmin2(p1 + hml_tfbnd * ONEM, p[c + (size_t)kk * np]);
  const double dpmli = 1. / (pml - p1);
  double tmldp = 0., smldp = 0.;
  double pk = p1;
  for (int k = 0; k < kk; k++) {
    const double pk1 = p[c + (size_t)(k + 1) * np];
    const double t = temp[c + (size_t)k * np], s = saln[c + (size_t)k * np];
    if (pk1 < pml) {
      const double d = dp[c + (size_t)k * np];
      tmldp = tmldp + t * d;
      smldp = smldp + s * d;
    } else {
      tmldp = tmldp + t * (pml - pk);
      smldp = smldp + s * (pml - pk);
      break;
    }
    pk = pk1;
  }
  V.f[F_util1][c] = eos::sig0(tmldp * dpmli, smldp * dpmli);
}

enum { E_PUV = 0, E_GM, E_SM, E_MFL, E_DLM, E_DLP, E_NARR };

// :1197-1738
__global__ void k_eda_column(const DevView *__restrict__ Vp, int n, int mm, int nn, EdaPar Q, int *__restrict__ errflag) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const bool isv = by_ == 1;
  if (!(isv ? V.m[I_iv][c] : V.m[I_iu][c])) return;
  const size_t np = V.nplane, xb = c, xa = isv ? c - V.ni : c - 1;
  const int kk = V.kk;
  const double ffac = .0625, fface = .99 * ffac, eps = 1.e-14, c5_21 = 5. / 21., delt1 = V.P.delt1;
  double *mfgm = (isv ? V.f[F_vmfltd] : V.f[F_umfltd]) + (size_t)mm * np, *mfsm = (isv ? V.f[F_vmflsm] : V.f[F_umflsm]) + (size_t)mm * np;
  const double *nslp = isv ? V.f[F_nslpy] : V.f[F_nslpx], *dpz = (isv ? V.f[F_dpv] : V.f[F_dpu]) + (size_t)nn * np;
  const double *p = V.f[F_p], *dp = V.f[F_dp] + (size_t)nn * np, *difint = V.f[F_difint];
  const double *hmlb = V.f[F_hml_tfbnd], *scp2 = V.f[F_scp2];
#define AT(a, x, k) (a)[(x) + (size_t)((k)-1) * np]
  double *const wb = V.wk + ((size_t)(by_ * gridDim.x + bx_) * (kk + 2) * E_NARR) * 64 + threadIdx.x;
#define W(a, k) wb[((size_t)((k)-1) * E_NARR + (a)) * 64]
  for (int k = 1; k <= kk; k++) { AT(mfgm, xb, k) = 0.; AT(mfsm, xb, k) = 0.; }        // :1209-1213
  const double mfleps = eps * EPSILP * (isv ? V.f[F_scv2] : V.f[F_scu2])[c];           // :1216
  const double et2mf = -GRAV * RHO0 * delt1 * (isv ? V.f[F_scvx] : V.f[F_scuy])[c];    // :1219
  const double pt = fmax2(AT(p, xa, 1), AT(p, xb, 1));                                 // ptu / ptv, :1182-1194
  // interface pressures and the last layer with mass on either side, :1224-1230
  int kmax = 1;
  {
    double a = pt;
    W(E_PUV, 1) = a;
    for (int k = 1; k <= kk; k++) {
      a = a + AT(dpz, xb, k);
      W(E_PUV, k + 1) = a;
      if (AT(dp, xa, k) > EPSILP || AT(dp, xb, k) > EPSILP) kmax = k;
    }
  }
  const double hml = .5 * (hmlb[xa] + hmlb[xb]);                                       // :1233
  const double puv1 = pt;
  const double pml = fmin2(puv1 + hml * ONEM, W(E_PUV, kmax + 1));                     // :1236
  const double dpmli = 1. / (pml - puv1);                                              // :1240
  int kml = kmax + 1;                                                                  // :1244-1251
  for (int k = kmax; k >= 2; k--) {
    if (W(E_PUV, k) > pml) kml = k;
    else break;
  }
  // the submesoscale transport of the column (upsilon), :1125-1176 (fox08) or zero (:1030-1044)
  double upssm = 0.;
  if (Q.mlrmth == 1) {
    const double rtau = 1. / Q.tau_mlr, csm = GRAV * ALPHA0 * Q.ce;
    const double f = .5 * (V.f[F_coriop][xa] + V.f[F_coriop][xb]);
    const double absfi = 1. / sqrt(f * f + rtau * rtau);
    const double lfi = 1. / fmax2(sqrt(DBCL82 * hml) * absfi, Q.lfmin);
    const double drho = V.f[F_util1][xb] - V.f[F_util1][xa];
    upssm = csm * hml * hml * drho * lfi * absfi;
  }
  // GM mass flux below the mixed layer base, :1255-1260
  for (int k = kml; k <= kmax; k++) {
    const double kappa = .25 * (AT(difint, xa, k - 1) + AT(difint, xb, k - 1) + AT(difint, xa, k) + AT(difint, xb, k));
    W(E_GM, k) = -kappa * AT(nslp, xb, k) * et2mf;
  }
  W(E_GM, kmax + 1) = 0.;
  // linear in interface pressure inside the mixed layer, :1265-1268
  W(E_GM, 1) = 0.;
  {
    const double gml = W(E_GM, kml);
    for (int k = 2; k <= kml - 1; k++) W(E_GM, k) = gml * (W(E_PUV, k) - puv1) * dpmli;
  }
  // submesoscale mass flux inside the mixed layer, :1273-1280
  W(E_SM, 1) = 0.;
  for (int k = 2; k <= kml - 1; k++) {
    const double r = 2. * (puv1 - W(E_PUV, k)) * dpmli + 1.;
    const double q = r * r;
    W(E_SM, k) = -upssm * (1. - q) * (1. + c5_21 * q) * et2mf;
  }
  for (int k = kml; k <= kmax + 1; k++) W(E_SM, k) = 0.;
  for (int k = 1; k <= kmax + 1; k++) W(E_MFL, k) = W(E_GM, k) + W(E_SM, k);           // :1288-1290
  // thicknesses available to the fluxes, :1296-1301
  const double pb = (isv ? V.f[F_pbv] : V.f[F_pbu])[c + (size_t)(n - 1) * np];
  const double scp2a = scp2[xa], scp2b = scp2[xb];
  {
    double pa_k = AT(p, xa, 1), pb_k = AT(p, xb, 1);
    for (int k = 1; k <= kmax; k++) {
      const double pa1 = AT(p, xa, k + 1), pb1 = AT(p, xb, k + 1);
      W(E_DLM, k) = fmax2(0., fmin2(pa1, pb) - fmax2(pa_k, pt));
      W(E_DLP, k) = fmax2(0., fmin2(pb1, pb) - fmax2(pb_k, pt));
      pa_k = pa1; pb_k = pb1;
    }
  }
  // iterative limiter by alternating sweeps, :1306-1394
  bool changed = true;
  int niter = 0, kdir = 1;
  while (changed) {
    niter = niter + 1;
    if (niter == 1000) { atomicOr(errflag, 1); return; }
    changed = false;
    kdir = -kdir;
    const int kb = (1 + kdir + (1 - kdir) * kmax) / 2, ke = (1 - kdir + (1 + kdir) * kmax) / 2;
    for (int k = kb; kdir > 0 ? k <= ke : k >= ke; k += kdir) {
      double lo = W(E_MFL, k), hi = W(E_MFL, k + 1);
      if (fabs(hi - lo) > fmax2(mfleps, eps * fabs(hi + lo))) {
        const double dm = W(E_DLM, k), dq = W(E_DLP, k);
        if (hi - lo > ffac * fmax2(EPSILP, dm) * scp2a) {
          const double q = fface * dm * scp2a;
          if (hi > -lo) {
            if (lo > -.5 * q) hi = lo + q;
            else { hi = .5 * q; lo = -hi; }
          } else {
            if (hi < .5 * q) lo = hi - q;
            else { lo = -.5 * q; hi = -lo; }
          }
          W(E_MFL, k) = lo; W(E_MFL, k + 1) = hi;
          changed = true;
        } else if (hi - lo < -ffac * fmax2(EPSILP, dq) * scp2b) {
          const double q = fface * dq * scp2b;
          if (hi < -lo) {
            if (lo < .5 * q) hi = lo - q;
            else { hi = -.5 * q; lo = -hi; }
          } else {
            if (hi > -.5 * q) lo = hi + q;
            else { lo = .5 * q; hi = -lo; }
          }
          W(E_MFL, k) = lo; W(E_MFL, k + 1) = hi;
          changed = true;
        }
      }
    }
  }
  // the two parts follow the limited sum, :1398-1435
  for (int k = 1; k <= kmax + 1; k++) {
    const double mfl = W(E_MFL, k);
    double gm = W(E_GM, k), sm = W(E_SM, k);
    if (fabs(mfl) < mfleps) {
      W(E_MFL, k) = 0.;
      gm = 0.; sm = 0.;
    } else if (mfl > 0.) {
      if (gm > sm) {
        if (mfl > 2. * sm) gm = mfl - sm;
        else { gm = .5 * mfl; sm = gm; }
      } else {
        if (mfl > 2. * gm) sm = mfl - gm;
        else { sm = .5 * mfl; gm = sm; }
      }
    } else {
      if (gm < sm) {
        if (mfl < 2. * sm) gm = mfl - sm;
        else { gm = .5 * mfl; sm = gm; }
      } else {
        if (mfl < 2. * gm) sm = mfl - gm;
        else { sm = .5 * mfl; gm = sm; }
      }
    }
    W(E_GM, k) = gm; W(E_SM, k) = sm;
  }
  // final mass fluxes and the reference's bound checks, :1441-1466
  {
    double lo = W(E_MFL, 1), glo = W(E_GM, 1), slo = W(E_SM, 1);
    for (int k = 1; k <= kmax; k++) {
      const double hi = W(E_MFL, k + 1), ghi = W(E_GM, k + 1), shi = W(E_SM, k + 1);
      double fg = 0., fs = 0.;
      if (fabs(hi - lo) > fmax2(mfleps, eps * fabs(hi + lo))) { fg = ghi - glo; fs = shi - slo; }
      AT(mfgm, xb, k) = fg;
      AT(mfsm, xb, k) = fs;
      if (fg + fs > ffac * fmax2(EPSILP, W(E_DLM, k)) * scp2a || fg + fs < -ffac * fmax2(EPSILP, W(E_DLP, k)) * scp2b) atomicOr(errflag, 2);
      lo = hi; glo = ghi; slo = shi;
    }
  }
#undef W
#undef AT
}

// heat and salt components, :1874-1901
__global__ void k_eda_ts(const DevView *__restrict__ Vp, int mm) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const size_t np = V.nplane, o = c + (size_t)(by_ + mm) * np;
  const double *temp = V.f[F_temp], *saln = V.f[F_saln];
  if (V.m[I_iu][c]) {
    const double fd = V.f[F_umfltd][o], fs = V.f[F_umflsm][o];
    double q = .5 * (temp[o - 1] + temp[o]);
    V.f[F_utfltd][o] = fd * q;
    V.f[F_utflsm][o] = fs * q;
    q = .5 * (saln[o - 1] + saln[o]);
    V.f[F_usfltd][o] = fd * q;
    V.f[F_usflsm][o] = fs * q;
  }
  if (V.m[I_iv][c]) {
    const double fd = V.f[F_vmfltd][o], fs = V.f[F_vmflsm][o];
    double q = .5 * (temp[o - V.ni] + temp[o]);
    V.f[F_vtfltd][o] = fd * q;
    V.f[F_vtflsm][o] = fs * q;
    q = .5 * (saln[o - V.ni] + saln[o]);
    V.f[F_vsfltd][o] = fd * q;
    V.f[F_vsflsm][o] = fs * q;
  }
}

int st_eddtra_ale(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)k1m; (void)k1n;
  const DevView &h = c->h;
  if (h.P.eitmth != 2) return ctx_fail(c, " eitmth_opt is unsupported for vcoord = 'cntiso_hybrid'!");      // :1865-1872
  if (h.nwk < 2 * E_NARR + 3) return ctx_fail(c, "eddtra: device work space too small");
  if (int rc = ctx_err_words(c)) return rc;
  int *errflag = c->err_dev + 1;
  EdaPar Q;
  Q.mlrmth = c->mlrmth; Q.ce = c->eddtra_ce; Q.tau_mlr = c->tau_mlr; Q.lfmin = c->lfmin; Q.mlbl_max_ratio = c->mlbl_max_ratio;
  const double delt1 = h.P.delt1;
  Q.wf_growing_hbl = c->tau_growing_hbl / (c->tau_growing_hbl + delt1);                // :1054-1057
  Q.wf_decaying_hbl = c->tau_decaying_hbl / (c->tau_decaying_hbl + delt1);
  Q.wf_growing_hml = c->tau_growing_hml / (c->tau_growing_hml + delt1);
  Q.wf_decaying_hml = c->tau_decaying_hml / (c->tau_decaying_hml + delt1);
  if (c->mlrmth != 0) {
    {
      TimeScope ts(c, "eddtra");
      hipLaunchKernelGGL(k_eda_mixed_layer, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, nn, Q);
    }
    if (int rc = st_xctilr(c, h.f[F_hml_tfbnd], 1, 1, 1, 1, 1)) return rc;              // :1100, halo_ps
    if (int rc = st_xctilr(c, h.f[F_util1], 1, 1, 1, 1, 1)) return rc;                  // :1123
  }
  {
    TimeScope ts(c, "eddtra");
    hipLaunchKernelGGL(k_eda_column, plane_grid(h, 2, 64), dim3(64), 0, c->stream, c->d, n, mm, nn, Q, errflag);
    hipLaunchKernelGGL(k_eda_ts, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, mm);
  }
  HIPCHK(c, hipGetLastError());
  if (!c->defer_checks) return ctx_check_errors(c);
  return 0;
}
