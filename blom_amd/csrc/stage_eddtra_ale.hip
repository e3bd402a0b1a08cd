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
// mlrmth = 'bod23' (Bodner et al. 2023, :1058-1081, :1127-1154) takes ustar3 / wstar3 as they stand in mod_forcing -- their producer
// for this coordinate is the CVMix-bound difest_vertical_hybrid, out of scope: uploaded fields -- and raises their weighted sum to the
// power 2/3 with the host libm's bits (pow_libm.h); its running mean wpup_tf is state like hbl_tf.  Non-convergence of the limiter and a
// violated final bound are the reference's xchalt exits and come back as errors.  Roofline: HBM.
// Parity: cross-checked against the reference's REAL mod_eddtra compiled against a stand-in for mod_difest that holds only
// OBLdepth (oracle/Makefile *_xale, tests/test_xcheck_eddtra_ale.py) -- a cross-check, not a pin.
#include "blomgpu_internal.h"
#include "eos.h"
#include "pow_libm.h"

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
  double ce, tau_mlr, wf_growing_hbl, wf_decaying_hbl, wf_growing_hml, wf_decaying_hml, lfmin, mlbl_max_ratio, cl, mstar, nstar, wpup_min;
};

__device__ inline void rmeanfilt(double &filtered, double signal, double wg, double wd) {   // :121-151
  const double wf = signal >= filtered ? wg : wd;
  filtered = wf * filtered + (1. - wf) * signal;
}

// :1050-1123 (mlrmth = 'fox08' or 'bod23')
__global__ void k_eda_mixed_layer(const DevView *__restrict__ Vp, int nn, EdaPar Q) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  const int kk = V.kk;
  const double hbl = V.f[F_OBLdepth][c];
  double hbl_tf = V.f[F_hbl_tf][c], hml_tf1 = V.f[F_hml_tf1][c], hml_tf = V.f[F_hml_tf][c];
  rmeanfilt(hbl_tf, hbl, Q.wf_growing_hbl, Q.wf_decaying_hbl);
  if (Q.mlrmth == 2) {                                                                 // :1064-1069
    const double c2_3 = 2. / 3.;
    const double wpup = fmax2(Q.wpup_min, pow_libm(Q.mstar * V.f[F_ustar3][c] + Q.nstar * V.f[F_wstar3][c], c2_3));
    double wpup_tf = V.f[F_wpup_tf][c];
    rmeanfilt(wpup_tf, wpup, Q.wf_growing_hbl, Q.wf_decaying_hbl);
    V.f[F_wpup_tf][c] = wpup_tf;
  }
  rmeanfilt(hml_tf1, V.f[F_mld][c], Q.wf_growing_hbl, Q.wf_decaying_hbl);
  rmeanfilt(hml_tf, hml_tf1, Q.wf_growing_hml, Q.wf_decaying_hml);
  const double hml_tfbnd = fmin2(hml_tf, Q.mlbl_max_ratio * hbl_tf);
  V.f[F_hbl_tf][c] = hbl_tf;
  V.f[F_hml_tf1][c] = hml_tf1;
  V.f[F_hml_tf][c] = hml_tf;
  V.f[F_hml_tfbnd][c] = hml_tfbnd;
  // vertically averaged mixed layer density, :1103-1123
  gcd_t p = V.f[F_p], dp = V.f[F_dp] + (size_t)nn * np, temp = V.f[F_temp] + (size_t)nn * np, saln = V.f[F_saln] + (size_t)nn * np;
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
#define EU 4        /* levels whose loads a pass keeps in flight */

// :1197-1738.  The reference's ten loops over a column are three passes here, each statement as it stands:
//   1. interface pressures puv, the last layer with mass kmax (:1224-1230); the mixed layer base is found on the way -- puv never
//      decreases, so the run of interfaces below the base that the reference walks upwards from kmax (:1244-1251) starts at the
//      first interface above pml;
//   2. GM and submesoscale interface fluxes, their sum, the thicknesses the limiter may deplete (:1255-1301);
//   -- the alternating-sweep limiter (:1306-1394), the sweep's shared interface value carried in a register --
//   3. the split of the limited sum (:1398-1435) and the layer fluxes with the reference's bound checks (:1441-1466).
// What a pass reads at the next EU levels is loaded before the current ones are worked on.
__global__ void k_eda_column(const DevView *__restrict__ Vp, int n, int mm, int nn, EdaPar Q, int *__restrict__ errflag) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const bool isv = by_ == 1;
  if (!(isv ? V.m[I_iv][c] : V.m[I_iu][c])) return;
  const size_t np = V.nplane, xb = c, xa = isv ? c - V.ni : c - 1;
  const int kk = V.kk;
  const double ffac = .0625, fface = .99 * ffac, eps = 1.e-14, c5_21 = 5. / 21., delt1 = V.P.delt1;
  gd_t __restrict__ mfgm = (isv ? V.f[F_vmfltd] : V.f[F_umfltd]) + (size_t)mm * np;
  gd_t __restrict__ mfsm = (isv ? V.f[F_vmflsm] : V.f[F_umflsm]) + (size_t)mm * np;
  gcd_t __restrict__ nslp = isv ? V.f[F_nslpy] : V.f[F_nslpx], dpz = (isv ? V.f[F_dpv] : V.f[F_dpu]) + (size_t)nn * np;
  gcd_t __restrict__ p = V.f[F_p], dp = V.f[F_dp] + (size_t)nn * np, difint = V.f[F_difint];
  gcd_t hmlb = V.f[F_hml_tfbnd], scp2 = V.f[F_scp2];
#define AT(a, x, k) (a)[(x) + (size_t)((k)-1) * np]
#define CLK(k, lo, hi) ((k) < (lo) ? (lo) : ((k) > (hi) ? (hi) : (k)))
  gd_t __restrict__ const wb = global_ptr(V.wk) + ((size_t)(by_ * gridDim.x + bx_) * (kk + 2) * E_NARR) * 64 + threadIdx.x;
#define W(a, k) wb[((size_t)((k)-1) * E_NARR + (a)) * 64]
  const double mfleps = eps * EPSILP * (isv ? V.f[F_scv2] : V.f[F_scu2])[c];           // :1216
  const double et2mf = -GRAV * RHO0 * delt1 * (isv ? V.f[F_scvx] : V.f[F_scuy])[c];    // :1219
  const double pt = fmax2(AT(p, xa, 1), AT(p, xb, 1));                                 // ptu / ptv, :1182-1194
  const double hml = .5 * (hmlb[xa] + hmlb[xb]);                                       // :1233
  const double puv1 = pt, pml_a = puv1 + hml * ONEM;
  // ---- pass 1 ------------------------------------------------------------------------------------------------------------------
  int kmax = 1, ka = kk + 2;                   // ka: first interface k >= 2 with puv(k) > puv(1) + hml*onem
  {
    double a = pt;
    W(E_PUV, 1) = a;
    for (int k0 = 1; k0 <= kk; k0 += EU) {
      double a0[EU], a1[EU], a2[EU];
#pragma unroll
      for (int u = 0; u < EU; u++) { const int kq = CLK(k0 + u, 1, kk); a0[u] = AT(dpz, xb, kq); a1[u] = AT(dp, xa, kq); a2[u] = AT(dp, xb, kq); }
#pragma unroll
      for (int u = 0; u < EU; u++) {
        const int k = k0 + u;
        if (k > kk) break;
        AT(mfgm, xb, k) = 0.; AT(mfsm, xb, k) = 0.;                                    // :1209-1213
        a = a + a0[u];
        W(E_PUV, k + 1) = a;
        if (a1[u] > EPSILP || a2[u] > EPSILP) kmax = k;
        if (a > pml_a && ka > k + 1) ka = k + 1;
      }
    }
  }
  const double pml = fmin2(pml_a, W(E_PUV, kmax + 1));                                 // :1236
  const double dpmli = 1. / (pml - puv1);                                              // :1240
  const int kml = ka <= kmax ? ka : kmax + 1;                                          // :1244-1251
  // the submesoscale transport of the column (upsilon), :1125-1176 (bod23, fox08) or zero (:1030-1044)
  double upssm = 0.;
  if (Q.mlrmth == 2) {
    const double csm = GRAV * ALPHA0 * Q.ce / Q.cl;
    const double hbl = .5 * (V.f[F_hbl_tf][xa] + V.f[F_hbl_tf][xb]);
    const double absf = .5 * fabs(V.f[F_coriop][xa] + V.f[F_coriop][xb]);
    const double wpup = .5 * (V.f[F_wpup_tf][xa] + V.f[F_wpup_tf][xb]);
    const double drho = V.f[F_util1][xb] - V.f[F_util1][xa];
    upssm = csm * absf * hbl * hml * hml * drho / wpup;
  } else if (Q.mlrmth == 1) {
    const double rtau = 1. / Q.tau_mlr, csm = GRAV * ALPHA0 * Q.ce;
    const double f = .5 * (V.f[F_coriop][xa] + V.f[F_coriop][xb]);
    const double absfi = 1. / sqrt(f * f + rtau * rtau);
    const double lfi = 1. / fmax2(sqrt(DBCL82 * hml) * absfi, Q.lfmin);
    const double drho = V.f[F_util1][xb] - V.f[F_util1][xa];
    upssm = csm * hml * hml * drho * lfi * absfi;
  }
  // ---- pass 2 ------------------------------------------------------------------------------------------------------------------
  const double pb = (isv ? V.f[F_pbv] : V.f[F_pbu])[c + (size_t)(n - 1) * np];
  const double scp2a = scp2[xa], scp2b = scp2[xb];
  auto gm_below = [&](int k, double d0, double d1, double d2, double d3, double sl) {  // :1255-1259
    (void)k;
    const double kappa = .25 * (d0 + d1 + d2 + d3);
    return -kappa * sl * et2mf;
  };
  double gml = 0.;                                                                     // mflgm(kml): :1260 when kml = kmax+1
  if (kml <= kmax) gml = gm_below(kml, AT(difint, xa, kml - 1), AT(difint, xb, kml - 1), AT(difint, xa, kml), AT(difint, xb, kml), AT(nslp, xb, kml));
  {
    double pa_k = AT(p, xa, 1), pb_k = AT(p, xb, 1);
    for (int k0 = 1; k0 <= kmax + 1; k0 += EU) {
      double d0[EU], d1[EU], d2[EU], d3[EU], sl[EU], pu[EU], a0[EU], a1[EU];
#pragma unroll
      for (int u = 0; u < EU; u++) {
        const int kq = CLK(k0 + u, 2, kk);                   // difint(k-1), difint(k), nslp(k): read for kml <= k <= kmax only
        d0[u] = AT(difint, xa, kq - 1); d1[u] = AT(difint, xb, kq - 1); d2[u] = AT(difint, xa, kq); d3[u] = AT(difint, xb, kq);
        sl[u] = AT(nslp, xb, kq);
        pu[u] = W(E_PUV, CLK(k0 + u, 1, kk + 1));
        const int kp = CLK(k0 + u, 1, kk);                   // p(k+1) of the two columns: read for k <= kmax only
        a0[u] = AT(p, xa, kp + 1); a1[u] = AT(p, xb, kp + 1);
      }
#pragma unroll
      for (int u = 0; u < EU; u++) {
        const int k = k0 + u;
        if (k > kmax + 1) break;
        double gm, sm;
        if (k == 1) { gm = 0.; sm = 0.; }                                              // :1264, :1272
        else if (k < kml) {
          gm = gml * (pu[u] - puv1) * dpmli;                                           // :1265-1268
          const double r = 2. * (puv1 - pu[u]) * dpmli + 1.;
          const double q = r * r;
          sm = -upssm * (1. - q) * (1. + c5_21 * q) * et2mf;                           // :1273-1277
        } else {
          gm = k <= kmax ? gm_below(k, d0[u], d1[u], d2[u], d3[u], sl[u]) : 0.;        // :1255-1260
          sm = 0.;                                                                     // :1278-1280
        }
        W(E_GM, k) = gm; W(E_SM, k) = sm;
        W(E_MFL, k) = gm + sm;                                                         // :1288-1290
        if (k <= kmax) {                                                               // :1296-1301
          W(E_DLM, k) = fmax2(0., fmin2(a0[u], pb) - fmax2(pa_k, pt));
          W(E_DLP, k) = fmax2(0., fmin2(a1[u], pb) - fmax2(pb_k, pt));
          pa_k = a0[u]; pb_k = a1[u];
        }
      }
    }
  }
  // ---- iterative limiter by alternating sweeps, :1306-1394 ----------------------------------------------------------------------
  bool changed = true;
  int niter = 0, kdir = 1;
  while (changed) {
    niter = niter + 1;
    if (niter == 1000) { atomicOr(errflag, 1); return; }
    changed = false;
    kdir = -kdir;
    // a sweep walks the layers 1..kmax pairwise; the value a step leaves in the interface it shares with the next step travels in
    // a register, the far interface and the two thicknesses of EU steps are loaded ahead (no step writes what a later step's
    // look-ahead reads)
    const bool up = kdir > 0;
    double carry = up ? W(E_MFL, 1) : W(E_MFL, kmax + 1);
    for (int s0 = 0; s0 <= kmax - 1; s0 += EU) {
      double a0[EU], a1[EU], a2[EU];
#pragma unroll
      for (int u = 0; u < EU; u++) {
        const int st = s0 + u <= kmax - 1 ? s0 + u : kmax - 1;
        const int kq = up ? 1 + st : kmax - st;
        a0[u] = up ? W(E_MFL, kq + 1) : W(E_MFL, kq); a1[u] = W(E_DLM, kq); a2[u] = W(E_DLP, kq);
      }
#pragma unroll
      for (int u = 0; u < EU; u++) {
        if (s0 + u > kmax - 1) break;
        const int k = up ? 1 + s0 + u : kmax - s0 - u;
        double lo = up ? carry : a0[u], hi = up ? a0[u] : carry;
        if (fabs(hi - lo) > fmax2(mfleps, eps * fabs(hi + lo))) {
          const double dm = a1[u], dq = a2[u];
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
        carry = up ? hi : lo;
      }
    }
  }
  // ---- pass 3: the two parts follow the limited sum (:1398-1435); layer fluxes and the reference's bound checks (:1441-1466) ------
  {
    double lo = 0., glo = 0., slo = 0.;
    for (int k0 = 1; k0 <= kmax + 1; k0 += EU) {
      double am[EU], ag[EU], as[EU], a1[EU], a2[EU];
#pragma unroll
      for (int u = 0; u < EU; u++) {
        const int kq = CLK(k0 + u, 1, kmax + 1), kl = CLK(k0 + u - 1, 1, kmax);
        am[u] = W(E_MFL, kq); ag[u] = W(E_GM, kq); as[u] = W(E_SM, kq);
        a1[u] = W(E_DLM, kl); a2[u] = W(E_DLP, kl);                 // of the layer above interface k
      }
#pragma unroll
      for (int u = 0; u < EU; u++) {
        const int k = k0 + u;
        if (k > kmax + 1) break;
        double mfl = am[u], gm = ag[u], sm = as[u];
        if (fabs(mfl) < mfleps) {
          mfl = 0.; gm = 0.; sm = 0.;
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
        if (k >= 2) {                                         // layer k-1 between the interfaces k-1 (lo) and k (hi)
          double fg = 0., fs = 0.;
          if (fabs(mfl - lo) > fmax2(mfleps, eps * fabs(mfl + lo))) { fg = gm - glo; fs = sm - slo; }
          AT(mfgm, xb, k - 1) = fg;
          AT(mfsm, xb, k - 1) = fs;
          if (fg + fs > ffac * fmax2(EPSILP, a1[u]) * scp2a || fg + fs < -ffac * fmax2(EPSILP, a2[u]) * scp2b) atomicOr(errflag, 2);
        }
        lo = mfl; glo = gm; slo = sm;
      }
    }
  }
#undef W
#undef CLK
#undef AT
}

// heat and salt components, :1874-1901
__global__ void k_eda_ts(const DevView *__restrict__ Vp, int mm) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const size_t np = V.nplane, o = c + (size_t)(by_ + mm) * np;
  gcd_t temp = V.f[F_temp], saln = V.f[F_saln];
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
  Q.cl = c->eddtra_cl; Q.mstar = c->mstar; Q.nstar = c->nstar; Q.wpup_min = c->wpup_min;
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
    if (c->mlrmth == 2) {                                                               // :1079-1080
      if (int rc = st_xctilr(c, h.f[F_hbl_tf], 1, 1, 1, 1, 1)) return rc;
      if (int rc = st_xctilr(c, h.f[F_wpup_tf], 1, 1, 1, 1, 1)) return rc;
    }
    if (int rc = st_xctilr(c, h.f[F_hml_tfbnd], 1, 1, 1, 1, 1)) return rc;              // :1081 / :1100, halo_ps
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
