// eddtra -- eddy-induced transport for vcoord = isopyc_bulkml, phy/mod_eddtra.F90:1808-1857
// (driver + heat/salt components), :152-226 (interface diffusion), :228-1000 (Gent-McWilliams).
//
// GM: one thread per velocity-point column (blockIdx.y = 0: u, 1: v -- the reference's two halves
// are mirror images), columns of a wavefront adjacent in i.  The private 1-D arrays mfl, dlm, dlp
// become work-space planes (coalesced plane-row accesses); upsilon is evaluated on the fly.  The
// alternating-sweep flux limiter has a data dependent trip count (:529-621); non-convergence and
// a violated final bound are the reference's two xchalt exits and come back as errors.
// intdif: layers are independent once the k-accumulation umfltd(k-1) += q(k); umfltd(k) = -q(k)
// is written as umfltd(k) = -q(k) + q(k+1) (same additions, same order), so it is (i,j,k)-parallel.
// Algorithmic bytes: 14 F; roofline: HBM.
// Parity: the reference module cannot be compiled in this image (mod_difest -> CVMix), so the CPU
// restatement oracle/c/eddtra.c this kernel is checked against is itself unpinned.
#include "blomgpu_internal.h"
#include "eos.h"

#define GRAV 9.806
#define RHO0 1.e3
#define EPSILP 1.e-12

#define THREAD_IJ(V)                                                       \
  unsigned bx_, by_;                                                       \
  xcd_block(bx_, by_);                                                     \
  const int t_ = bx_ * blockDim.x + threadIdx.x;                           \
  (void)by_;                                                               \
  if (t_ >= (V).nplane) return;                                            \
  const int i = t_ % (V).ni - (NBDY - 1), j = t_ / (V).ni - (NBDY - 1);    \
  const size_t c = t_

// work-space slots per component: mfl needs kk+1 levels and borrows the first level of the next slot
enum { G_MFL = 0, G_SPARE, G_DLM, G_DLP, G_NSLOT };

__global__ void k_eddtra_gm(const DevView *__restrict__ Vp, int n, int mm, int nn, int *__restrict__ errflag) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const bool isv = by_ == 1;
  if (!(isv ? V.m[I_iv][c] : V.m[I_iu][c])) return;
  const size_t np = V.nplane, xb = c, xa = isv ? c - V.ni : c - 1;
  const int kk = V.kk, sb = isv ? G_NSLOT : 0;
  const double ffac = .0625, fface = .99 * ffac, eps = 1.e-14, delt1 = V.P.delt1;
  gd_t mf = (isv ? V.f[F_vmfltd] : V.f[F_umfltd]) + (size_t)mm * np;
  gcd_t nslp = isv ? V.f[F_nslpy] : V.f[F_nslpx], dpz = (isv ? V.f[F_dpv] : V.f[F_dpu]) + (size_t)nn * np;
  gcd_t p = V.f[F_p], dp = V.f[F_dp] + (size_t)nn * np, difint = V.f[F_difint];
  gcd_t temp = V.f[F_temp] + (size_t)nn * np, saln = V.f[F_saln] + (size_t)nn * np;
#define AT(a, x, k) (a)[(x) + (size_t)((k)-1) * np]
  // the column's private arrays mfl(kk+1), dlm, dlp, wavefront-major: level k of the 64 columns of a wavefront in three
  // consecutive rows of 64 doubles (one piece of memory per level instead of three that lie a field apart)
  gd_t const wb = global_ptr(V.wk) + ((size_t)(by_ * gridDim.x + bx_) * (kk + 2) * 3) * 64 + threadIdx.x;
  (void)sb;
#define MFL(k) wb[((size_t)((k)-1) * 3 + 0) * 64]
#define DLM(k) wb[((size_t)((k)-1) * 3 + 1) * 64]
#define DLP(k) wb[((size_t)((k)-1) * 3 + 2) * 64]
#define GU 8       /* levels whose loads a sweep keeps in flight */
#define CLK(k, lo, hi) ((k) < (lo) ? (lo) : ((k) > (hi) ? (hi) : (k)))
  for (int k = 1; k <= kk; k++) AT(mf, xb, k) = 0.;                                   // :300-303
  const double et2mf = -GRAV * RHO0 * delt1 * (isv ? V.f[F_scvx] : V.f[F_scuy])[c];   // :306
  int kmax = 1;                                                                       // :310-314
  for (int k0 = 3; k0 <= kk; k0 += GU) {
    double a0[GU], a1[GU];
#pragma unroll
    for (int u = 0; u < GU; u++) { const int kq = CLK(k0 + u, 3, kk); a0[u] = AT(dp, xa, kq); a1[u] = AT(dp, xb, kq); }
#pragma unroll
    for (int u = 0; u < GU; u++)
      if (k0 + u <= kk && (a0[u] > EPSILP || a1[u] > EPSILP)) kmax = k0 + u;
  }
  const int kfa = V.m[I_kfpla][xa + (size_t)(n - 1) * np], kfb = V.m[I_kfpla][xb + (size_t)(n - 1) * np];
  const double scp2a = V.f[F_scp2][xa], scp2b = V.f[F_scp2][xb];
  const double pb = (isv ? V.f[F_pbv] : V.f[F_pbu])[c + (size_t)(n - 1) * np];
  const double s2 = (isv ? V.f[F_scv2] : V.f[F_scu2])[c];
  const double pt = fmax2(AT(p, xa, 1), AT(p, xb, 1));                                // :259-268
  const double pa3 = AT(p, xa, 3), pb3 = AT(p, xb, 3);
  int kintr, kmin;
  if (kfa > kk && kfb > kk) return;                                                   // case 1
  const double ups3 = -(.5 * (AT(difint, xa, 2) + AT(difint, xb, 2))) * AT(nslp, xb, 3);
  if (kfa <= kk && kfb > kk) {                                                        // case 2, :343-382
    kintr = kfa;
    const double rb = eos::rho(pb3, AT(temp, xb, 2), AT(saln, xb, 2));
    while (eos::rho(pb3, AT(temp, xa, kintr), AT(saln, xa, kintr)) < rb || AT(dp, xa, kintr) < EPSILP) {
      kintr = kintr + 1;
      if (kintr == kmax + 1) break;
    }
    if (kintr == kmax + 1) return;
    if (ups3 <= 0.) return;
    kmin = kintr - 1;
    MFL(kmin) = 0.;
    MFL(kintr) = et2mf * ups3;
    for (int k = kintr + 1; k <= kmax + 1; k++) MFL(k) = 0.;
  } else if (kfa > kk && kfb <= kk) {                                                 // case 3, :384-423
    kintr = kfb;
    const double ra = eos::rho(pa3, AT(temp, xa, 2), AT(saln, xa, 2));
    while (eos::rho(pa3, AT(temp, xb, kintr), AT(saln, xb, kintr)) < ra || AT(dp, xb, kintr) < EPSILP) {
      kintr = kintr + 1;
      if (kintr == kmax + 1) break;
    }
    if (kintr == kmax + 1) return;
    if (ups3 >= 0.) return;
    kmin = kintr - 1;
    MFL(kmin) = 0.;
    MFL(kintr) = et2mf * ups3;
    for (int k = kintr + 1; k <= kmax + 1; k++) MFL(k) = 0.;
  } else {                                                                            // case 4, :425-483
    kintr = kfa > kfb ? kfa : kfb;
    // upsilon(k) = -kappa(k)*nslp(k), k = kintr+1..kmax; upsilon(kmax+1) = 0
    auto ups = [&](int k) -> double {
      if (k >= kmax + 1) return 0.;      // upsilon(kmax+1) = 0; beyond it the reference reads an unassigned value
      const double kappa = .25 * (AT(difint, xa, k - 1) + AT(difint, xb, k - 1) + AT(difint, xa, k) + AT(difint, xb, k));
      return -kappa * AT(nslp, xb, k);
    };
    const double ups_k1 = ups(kintr + 1);                       // upsilon(kintr+1) of the unshifted column
    const int kq = kintr - 1;
    bool shift = false;
    if (kfa < kintr && ups3 - ups_k1 > 0. &&
        eos::rho(pb3, AT(temp, xa, kq), AT(saln, xa, kq)) > eos::rho(pb3, AT(temp, xb, 2), AT(saln, xb, 2)))
      shift = true;
    else if (kfb < kintr && ups3 - ups_k1 < 0. &&
             eos::rho(pa3, AT(temp, xb, kq), AT(saln, xb, kq)) > eos::rho(pa3, AT(temp, xa, 2), AT(saln, xa, 2)))
      shift = true;
    if (shift) kintr = kintr - 1;                               // upsilon(kintr+1) = upsilon(kintr+2)
    kmin = kintr - 1;
    MFL(kmin) = 0.;
    // (with kfpla < 3, which valid states never hold, the copy above lands on upsilon(3) itself)
    MFL(kintr) = et2mf * (shift && kintr + 1 == 3 ? ups_k1 : ups3);
    int kf = kintr + 1;
    if (shift) { MFL(kintr + 1) = et2mf * ups_k1; kf = kintr + 2; }
    if (kf <= kmax) {
      // mfl(k) = et2mf * upsilon(k), k = kf..kmax: the difint values of level k-1 are carried over from level k
      double da = AT(difint, xa, kf - 1), db = AT(difint, xb, kf - 1);
      for (int k0 = kf; k0 <= kmax; k0 += GU) {
        double a0[GU], a1[GU], a2[GU];
#pragma unroll
        for (int u = 0; u < GU; u++) { const int kq = CLK(k0 + u, kf, kmax); a0[u] = AT(difint, xa, kq); a1[u] = AT(difint, xb, kq); a2[u] = AT(nslp, xb, kq); }
#pragma unroll
        for (int u = 0; u < GU; u++) {
          const int k = k0 + u;
          if (k > kmax) break;
          const double kappa = .25 * (da + db + a0[u] + a1[u]);
          MFL(k) = et2mf * (-kappa * a2[u]);
          da = a0[u]; db = a1[u];
        }
      }
    }
    MFL(kmax + 1) = 0.;
  }
  // layer thicknesses available for depletion, :493-502
  DLM(kmin) = fmax2(0., fmin2(pa3, pb) - fmax2(AT(p, xa, 1), pt));
  DLP(kmin) = fmax2(0., fmin2(pb3, pb) - fmax2(AT(p, xb, 1), pt));
  {
    double pa_k = AT(p, xa, kintr <= kk ? kintr : kk), pb_k = AT(p, xb, kintr <= kk ? kintr : kk);
    for (int k0 = kintr; k0 <= kmax; k0 += GU) {
      double a0[GU], a1[GU];
#pragma unroll
      for (int u = 0; u < GU; u++) { const int kq = CLK(k0 + u, kintr, kmax); a0[u] = AT(p, xa, kq + 1); a1[u] = AT(p, xb, kq + 1); }
#pragma unroll
      for (int u = 0; u < GU; u++) {
        const int k = k0 + u;
        if (k > kmax) break;
        DLM(k) = fmax2(0., fmin2(a0[u], pb) - fmax2(pa_k, pt));
        DLP(k) = fmax2(0., fmin2(a1[u], pb) - fmax2(pb_k, pt));
        pa_k = a0[u]; pb_k = a1[u];
      }
    }
  }
  {                                                                                   // :507-524
    const double fhi = fface * fmax2(0., fmin2((pa3 - pt) * scp2a, (pb - AT(p, xb, kintr)) * scp2b));
    const double flo = -fface * fmax2(0., fmin2((pb3 - pt) * scp2b, (pb - AT(p, xa, kintr)) * scp2a));
    double m0 = fmin2(fhi, fmax2(flo, MFL(kmin + 1)));
    MFL(kmin + 1) = m0;
    for (int k = kmin + 1; k <= kmax - 1; k++) {
      const double m1 = MFL(k + 1), dm = DLM(k), dq = DLP(k);
      if (m1 - m0 > ffac * fmax2(EPSILP, dm) * scp2a) m0 = m0 + fface * dm * scp2a;
      else if (m1 - m0 < -ffac * fmax2(EPSILP, dq) * scp2b) m0 = m0 - fface * dq * scp2b;
      else break;
      MFL(k + 1) = m0;
    }
  }
  // iterative limiter by alternating sweeps, :529-621
  bool changed = true;
  int niter = 0, kdir = 1;
  while (changed) {
    niter = niter + 1;
    if (niter == 1000) { atomicOr(errflag, 1); return; }
    changed = false;
    kdir = -kdir;
    // A sweep walks the interfaces pairwise; the value a step leaves in the interface it shares with the next step
    // travels in a register (`carry`), the far interface and the two thicknesses of GU steps are loaded ahead: no
    // step of the sweep writes what a later step's look-ahead reads.
    const bool up = kdir > 0;
    double carry = up ? MFL(kmin) : MFL(kmax + 1);
    for (int s0 = 0; s0 <= kmax - kmin; s0 += GU) {
      double a0[GU], a1[GU], a2[GU];
#pragma unroll
      for (int u = 0; u < GU; u++) {
        const int st = s0 + u <= kmax - kmin ? s0 + u : kmax - kmin;
        const int kq = up ? kmin + st : kmax - st;
        a0[u] = up ? MFL(kq + 1) : MFL(kq); a1[u] = DLM(kq); a2[u] = DLP(kq);
      }
#pragma unroll
      for (int u = 0; u < GU; u++) {
        if (s0 + u > kmax - kmin) break;
        const int k = up ? kmin + s0 + u : kmax - s0 - u;
        double lo = up ? carry : a0[u], hi = up ? a0[u] : carry;
        if (fabs(hi - lo) > eps * fmax2(EPSILP * s2, fabs(hi + lo))) {
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
            MFL(k) = lo; MFL(k + 1) = hi;
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
            MFL(k) = lo; MFL(k + 1) = hi;
            changed = true;
          }
        }
        carry = up ? hi : lo;
      }
    }
  }
  // final mass fluxes, :627-661
  {
    const double lo = MFL(kmin), hi = MFL(kmin + 1);
    if (fabs(hi - lo) > eps * fmax2(EPSILP * s2, fabs(hi + lo))) {
      const double d1 = AT(dpz, xb, 1), d2 = AT(dpz, xb, 2);
      const double f2 = hi - lo;
      const double f1 = f2 * d1 / (d1 + d2);
      AT(mf, xb, 1) = f1;
      AT(mf, xb, 2) = f2 - f1;
    }
  }
  {
    double lo = MFL(kintr);
    for (int k0 = kintr; k0 <= kmax; k0 += GU) {
      double a0[GU], a1[GU], a2[GU];
#pragma unroll
      for (int u = 0; u < GU; u++) { const int kq = CLK(k0 + u, kintr, kmax); a0[u] = MFL(kq + 1); a1[u] = DLM(kq); a2[u] = DLP(kq); }
#pragma unroll
      for (int u = 0; u < GU; u++) {
        const int k = k0 + u;
        if (k > kmax) break;
        const double hi = a0[u];
        double f = 0.;
        if (fabs(hi - lo) > eps * fmax2(EPSILP * s2, fabs(hi + lo))) f = hi - lo;
        AT(mf, xb, k) = f;
        if (f > ffac * fmax2(EPSILP, a1[u]) * scp2a || f < -ffac * fmax2(EPSILP, a2[u]) * scp2b) atomicOr(errflag, 2);
        lo = hi;
      }
    }
  }
}

// interface diffusion, :152-226
__global__ void k_eddtra_intdif(const DevView *__restrict__ Vp, int mm, int nn) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const int k = by_ + 1, kk = V.kk, ni = V.ni;
  const size_t np = V.nplane;
  const double delt1 = V.P.delt1;
  gcd_t dp = V.f[F_dp] + (size_t)nn * np, p = V.f[F_p], difint = V.f[F_difint], scp2 = V.f[F_scp2];
  // q of interface k (4 <= k <= kk) between the scalar points xa and c
  auto qk = [&](int kq, size_t xa, double metric) -> double {
    const double flxhi = .125 * fmin2(AT(dp, xa, kq - 1) * scp2[xa], AT(dp, c, kq) * scp2[c]);
    const double flxlo = -.125 * fmin2(AT(dp, c, kq - 1) * scp2[c], AT(dp, xa, kq) * scp2[xa]);
    const double q = .25 * (AT(difint, xa, kq - 1) + AT(difint, c, kq - 1) + AT(difint, xa, kq) + AT(difint, c, kq));
    return fmin2(flxhi, fmax2(flxlo, delt1 * q * (AT(p, xa, kq) - AT(p, c, kq)) * metric));
  };
  for (int comp = 0; comp < 2; comp++) {
    if (!(comp ? V.m[I_iv][c] : V.m[I_iu][c])) continue;
    const size_t xa = comp ? c - ni : c - 1;
    // delt1*q*(dp)*scuy*scuxi: the two metric factors multiply left to right in the reference, so
    // they cannot be pre-multiplied; pass them through a two-step product instead
    gd_t mf = (comp ? V.f[F_vmfltd] : V.f[F_umfltd]) + (size_t)mm * np;
    const double m1 = comp ? V.f[F_scvx][c] : V.f[F_scuy][c], m2 = comp ? V.f[F_scvyi][c] : V.f[F_scuxi][c];
    auto qk2 = [&](int kq) -> double {
      const double flxhi = .125 * fmin2(AT(dp, xa, kq - 1) * scp2[xa], AT(dp, c, kq) * scp2[c]);
      const double flxlo = -.125 * fmin2(AT(dp, c, kq - 1) * scp2[c], AT(dp, xa, kq) * scp2[xa]);
      const double q = .25 * (AT(difint, xa, kq - 1) + AT(difint, c, kq - 1) + AT(difint, xa, kq) + AT(difint, c, kq));
      return fmin2(flxhi, fmax2(flxlo, delt1 * q * (AT(p, xa, kq) - AT(p, c, kq)) * m1 * m2));
    };
    (void)qk;
    double val = 0.;
    if (k >= 4) val = -qk2(k);
    if (k >= 3 && k + 1 <= kk) val = val + qk2(k + 1);
    AT(mf, c, k) = val;
  }
}

// heat and salt components, :1837-1857
__global__ void k_eddtra_ts(const DevView *__restrict__ Vp, int mm) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const size_t np = V.nplane, o = c + (size_t)(by_ + mm) * np;
  gcd_t temp = V.f[F_temp], saln = V.f[F_saln];
  if (V.m[I_iu][c]) {
    const double f = V.f[F_umfltd][o];
    V.f[F_utfltd][o] = .5 * f * (temp[o - 1] + temp[o]);
    V.f[F_usfltd][o] = .5 * f * (saln[o - 1] + saln[o]);
  }
  if (V.m[I_iv][c]) {
    const double f = V.f[F_vmfltd][o];
    V.f[F_vtfltd][o] = .5 * f * (temp[o - V.ni] + temp[o]);
    V.f[F_vsfltd][o] = .5 * f * (saln[o - V.ni] + saln[o]);
  }
}

int st_eddtra(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  const DevView &h = c->h;
  if (h.P.vcoord_tag != 1) return st_eddtra_ale(c, m, n, mm, nn, k1m, k1n);           // :1859-1901 (stage_eddtra_ale.hip)
  if (c->mlrmth == 2) return ctx_fail(c, " init_eddtra: mlrmth = bod23 is unsupported with vcoord = 'isopyc_bulkml'!");   // :1787-1795
  if (h.nwk < 2 * G_NSLOT) return ctx_fail(c, "eddtra: device work space too small");
  if (int rc = ctx_err_words(c)) return rc;
  int *errflag = c->err_dev + 1;
  {
    TimeScope ts(c, "eddtra");
    if (h.P.eitmth == 1) hipLaunchKernelGGL(k_eddtra_intdif, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, mm, nn);
    else if (h.P.eitmth == 2)
      hipLaunchKernelGGL(k_eddtra_gm, plane_grid(h, 2, 64), dim3(64), 0, c->stream, c->d, n, mm, nn, errflag);
    else return ctx_fail(c, " eitmth_opt is unsupported for vcoord = 'isopyc_bulkml'!");   // :1829-1835
    hipLaunchKernelGGL(k_eddtra_ts, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, mm);
  }
  HIPCHK(c, hipGetLastError());
  if (h.P.eitmth == 2 && !c->defer_checks) return ctx_check_errors(c);      // :536-555, :640-660
  return 0;
}
