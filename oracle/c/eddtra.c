/* TEST INFRASTRUCTURE (oracle): restatement of eddtra for vcoord = isopyc_bulkml,
 * phy/mod_eddtra.F90:1808-1857 (driver), :152-226 (eddtra_intdif_isopyc_bulkml) and
 * :228-1000 (eddtra_gm_isopyc_bulkml).
 *
 * PARITY UNPINNED: mod_eddtra cannot be compiled in this image (it uses mod_difest, which needs
 * CVMix), and the reference's tests hold no vectors for it, so this restatement is
 * checked only by construction against the cited lines.  The u- and v-halves of the reference are
 * mirror images ((i-1,j) <-> (i,j-1), scuy <-> scvx, nslpx <-> nslpy, pbu <-> pbv, dpu <-> dpv,
 * scu2 <-> scv2); one routine with the neighbour offset as a parameter restates both.
 */
#include "ostate.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#define GRAV 9.806
#define RHO0 1.e3
#define EPSILP 1.e-12
#define ONEMM 9.806
#define KMAXDIM 256

double eos_rho(double p, double th, double s);

/* flat-index accessors: x = IX(S,i,j); level k is 1-based */
#define L3(a, x, k) (S->a)[(x) + (size_t)S->nplane * ((k)-1)]

/* One velocity-point column of eddtra_gm_isopyc_bulkml, :296-632 (u) == :640-976 (v).
 * xa, xb: flat indices of the scalar points (i-1,j)|(i,j-1) and (i,j); sc = scuy|scvx(i,j);
 * nslp = nslpx|nslpy, pbz = pbu|pbv, dpz = dpu|dpv, s2 = scu2|scv2(i,j), mf = umfltd|vmfltd. */
static int gm_column(OState *S, size_t xa, size_t xb, int n, int mm, int nn, double sc, const double *nslp,
                     const double *pbz, const double *dpz, double s2, double *mf, double pt) {
  const int kk = S->kk;
  const double ffac = .0625, fface = .99 * ffac, eps = 1.e-14, delt1 = S->delt1;
  const size_t np = (size_t)S->nplane;
  /* upsilon(kintr+1) is read at :455/:459 even when kintr >= kmax+1 left it unassigned (stale
   * kfpla); the reference reads an undefined stack value there, this restatement and the device
   * kernel both read 0. */
  double mfl[KMAXDIM + 2], dlm[KMAXDIM + 1], dlp[KMAXDIM + 1], upsilon[KMAXDIM + 2] = {0.};
  int k, km, kn, kintr, kmax, kmin;
#define NSLP(k) nslp[xb + np * ((k)-1)]
#define MF(k) mf[xb + np * ((k)-1)]
#define DPZ(k) dpz[xb + np * ((k)-1)]
  for (k = 1; k <= kk; k++) MF(k + mm) = 0.;                                        /* :300-303 */
  const double et2mf = -GRAV * RHO0 * delt1 * sc;                                   /* :306 */
  kmax = 1;                                                                         /* :310-314 */
  for (k = 3; k <= kk; k++)
    if (L3(dp, xa, k + nn) > EPSILP || L3(dp, xb, k + nn) > EPSILP) kmax = k;
  const int kfa = S->kfpla[xa + np * (n - 1)], kfb = S->kfpla[xb + np * (n - 1)];
  const double scp2a = S->scp2[xa], scp2b = S->scp2[xb], pb = pbz[xb + np * (n - 1)];
  if (kfa > kk && kfb > kk) return 0;                                               /* case 1, :335-341 */
  if (kfa <= kk && kfb > kk) {                                                      /* case 2, :343-382 */
    km = 2 + nn;
    kintr = kfa;
    kn = kintr + nn;
    while (eos_rho(L3(p, xb, 3), L3(temp, xa, kn), L3(saln, xa, kn)) <
               eos_rho(L3(p, xb, 3), L3(temp, xb, km), L3(saln, xb, km)) ||
           L3(dp, xa, kn) < EPSILP) {
      kintr = kintr + 1;
      if (kintr == kmax + 1) break;
      kn = kintr + nn;
    }
    if (kintr == kmax + 1) return 0;
    const double kappa = .5 * (L3(difint, xa, 2) + L3(difint, xb, 2));
    upsilon[3] = -kappa * NSLP(3);
    if (upsilon[3] <= 0.) return 0;
    kmin = kintr - 1;
    mfl[kmin] = 0.;
    mfl[kintr] = et2mf * upsilon[3];
    for (k = kintr + 1; k <= kmax + 1; k++) mfl[k] = 0.;
  } else if (kfa > kk && kfb <= kk) {                                               /* case 3, :384-423 */
    km = 2 + nn;
    kintr = kfb;
    kn = kintr + nn;
    while (eos_rho(L3(p, xa, 3), L3(temp, xb, kn), L3(saln, xb, kn)) <
               eos_rho(L3(p, xa, 3), L3(temp, xa, km), L3(saln, xa, km)) ||
           L3(dp, xb, kn) < EPSILP) {
      kintr = kintr + 1;
      if (kintr == kmax + 1) break;
      kn = kintr + nn;
    }
    if (kintr == kmax + 1) return 0;
    const double kappa = .5 * (L3(difint, xa, 2) + L3(difint, xb, 2));
    upsilon[3] = -kappa * NSLP(3);
    if (upsilon[3] >= 0.) return 0;
    kmin = kintr - 1;
    mfl[kmin] = 0.;
    mfl[kintr] = et2mf * upsilon[3];
    for (k = kintr + 1; k <= kmax + 1; k++) mfl[k] = 0.;
  } else {                                                                          /* case 4, :425-483 */
    kintr = kfa > kfb ? kfa : kfb;
    double kappa = .5 * (L3(difint, xa, 2) + L3(difint, xb, 2));
    upsilon[3] = -kappa * NSLP(3);
    for (k = kintr + 1; k <= kmax; k++) {
      kappa = .25 * (L3(difint, xa, k - 1) + L3(difint, xb, k - 1) + L3(difint, xa, k) + L3(difint, xb, k));
      upsilon[k] = -kappa * NSLP(k);
    }
    upsilon[kmax + 1] = 0.;
    km = 2 + nn;
    kn = kintr - 1 + nn;
    /* Fortran .and./.or. do not short-circuit by rule, but every operand here is a defined
     * value whenever it can decide the result (kintr-1 >= 3 when kf? < kintr), so evaluating
     * left to right with short-circuit gives the same truth value. */
    if ((kfa < kintr && upsilon[3] - upsilon[kintr + 1] > 0. &&
         eos_rho(L3(p, xb, 3), L3(temp, xa, kn), L3(saln, xa, kn)) >
             eos_rho(L3(p, xb, 3), L3(temp, xb, km), L3(saln, xb, km))) ||
        (kfb < kintr && upsilon[3] - upsilon[kintr + 1] < 0. &&
         eos_rho(L3(p, xa, 3), L3(temp, xb, kn), L3(saln, xb, kn)) >
             eos_rho(L3(p, xa, 3), L3(temp, xa, km), L3(saln, xa, km)))) {
      kintr = kintr - 1;
      upsilon[kintr + 1] = upsilon[kintr + 2];
    }
    kmin = kintr - 1;
    mfl[kmin] = 0.;
    mfl[kintr] = et2mf * upsilon[3];
    for (k = kintr + 1; k <= kmax; k++) mfl[k] = et2mf * upsilon[k];
    mfl[kmax + 1] = 0.;
  }
  /* layer thicknesses available for depletion, :493-502 */
  dlm[kmin] = fmax2(0., fmin2(L3(p, xa, 3), pb) - fmax2(L3(p, xa, 1), pt));
  dlp[kmin] = fmax2(0., fmin2(L3(p, xb, 3), pb) - fmax2(L3(p, xb, 1), pt));
  for (k = kintr; k <= kmax; k++) {
    dlm[k] = fmax2(0., fmin2(L3(p, xa, k + 1), pb) - fmax2(L3(p, xa, k), pt));
    dlp[k] = fmax2(0., fmin2(L3(p, xb, k + 1), pb) - fmax2(L3(p, xb, k), pt));
  }
  /* :507-524 */
  const double fhi = fface * fmax2(0., fmin2((L3(p, xa, 3) - pt) * scp2a, (pb - L3(p, xb, kintr)) * scp2b));
  const double flo = -fface * fmax2(0., fmin2((L3(p, xb, 3) - pt) * scp2b, (pb - L3(p, xa, kintr)) * scp2a));
  mfl[kmin + 1] = fmin2(fhi, fmax2(flo, mfl[kmin + 1]));
  for (k = kmin + 1; k <= kmax - 1; k++) {
    if (mfl[k + 1] - mfl[k] > ffac * fmax2(EPSILP, dlm[k]) * scp2a) mfl[k + 1] = mfl[k] + fface * dlm[k] * scp2a;
    else if (mfl[k + 1] - mfl[k] < -ffac * fmax2(EPSILP, dlp[k]) * scp2b) mfl[k + 1] = mfl[k] - fface * dlp[k] * scp2b;
    else break;
  }
  /* iterative limiter by alternating sweeps, :529-621.  NOTE: index kmin+1 .. kintr-1 do not exist
   * as layers (kmin = kintr-1), the sweep runs over k = kmin..kmax exactly as the reference's
   * do-loop bounds ((1-kdir)*kmax+(1+kdir)*kmin)/2 .. step kdir. */
  int changed = 1, niter = 0, kdir = 1;
  while (changed) {
    niter = niter + 1;
    if (niter == 1000) { fprintf(stderr, "oracle eddtra_gm_isopyc_bulkml: no convergence\n"); return 1; }
    changed = 0;
    kdir = -kdir;
    const int k0 = ((1 - kdir) * kmax + (1 + kdir) * kmin) / 2, k1 = ((1 - kdir) * kmin + (1 + kdir) * kmax) / 2;
    for (k = k0; kdir > 0 ? k <= k1 : k >= k1; k += kdir) {
      if (fabs(mfl[k + 1] - mfl[k]) > eps * fmax2(EPSILP * s2, fabs(mfl[k + 1] + mfl[k]))) {
        if (mfl[k + 1] - mfl[k] > ffac * fmax2(EPSILP, dlm[k]) * scp2a) {
          const double q = fface * dlm[k] * scp2a;
          if (mfl[k + 1] > -mfl[k]) {
            if (mfl[k] > -.5 * q) mfl[k + 1] = mfl[k] + q;
            else { mfl[k + 1] = .5 * q; mfl[k] = -mfl[k + 1]; }
          } else {
            if (mfl[k + 1] < .5 * q) mfl[k] = mfl[k + 1] - q;
            else { mfl[k] = -.5 * q; mfl[k + 1] = -mfl[k]; }
          }
          changed = 1;
        } else if (mfl[k + 1] - mfl[k] < -ffac * fmax2(EPSILP, dlp[k]) * scp2b) {
          const double q = fface * dlp[k] * scp2b;
          if (mfl[k + 1] < -mfl[k]) {
            if (mfl[k] < .5 * q) mfl[k + 1] = mfl[k] - q;
            else { mfl[k + 1] = -.5 * q; mfl[k] = -mfl[k + 1]; }
          } else {
            if (mfl[k + 1] > -.5 * q) mfl[k] = mfl[k + 1] + q;
            else { mfl[k] = .5 * q; mfl[k + 1] = -mfl[k]; }
          }
          changed = 1;
        }
      }
    }
  }
  /* final mass fluxes, :627-661 */
  k = kmin;
  if (fabs(mfl[k + 1] - mfl[k]) > eps * fmax2(EPSILP * s2, fabs(mfl[k + 1] + mfl[k]))) {
    MF(2 + mm) = mfl[k + 1] - mfl[k];
    MF(1 + mm) = MF(2 + mm) * DPZ(1 + nn) / (DPZ(1 + nn) + DPZ(2 + nn));
    MF(2 + mm) = MF(2 + mm) - MF(1 + mm);
  } else {
    MF(1 + mm) = 0.;
    MF(2 + mm) = 0.;
  }
  for (k = kintr; k <= kmax; k++) {
    km = k + mm;
    if (fabs(mfl[k + 1] - mfl[k]) > eps * fmax2(EPSILP * s2, fabs(mfl[k + 1] + mfl[k]))) MF(km) = mfl[k + 1] - mfl[k];
    else MF(km) = 0.;
    if (MF(km) > ffac * fmax2(EPSILP, dlm[k]) * scp2a || MF(km) < -ffac * fmax2(EPSILP, dlp[k]) * scp2b) {
      fprintf(stderr, "oracle eddtra_gm_isopyc_bulkml: flux bound violated at k=%d\n", k);
      return 1;
    }
  }
  return 0;
#undef NSLP
#undef MF
#undef DPZ
}

/* mod_eddtra.F90:152-226 */
static void intdif(OState *S, int mm, int nn) {
  const int ii = S->ii, jj = S->jj, kk = S->kk;
  const double delt1 = S->delt1;
  for (int j = 1; j <= jj; j++)
    for (int i = 1; i <= ii; i++) {
      if (A2(S, iu, i, j)) { A3(S, umfltd, i, j, 1 + mm) = 0.; A3(S, umfltd, i, j, 2 + mm) = 0.; A3(S, umfltd, i, j, 3 + mm) = 0.; }
      if (A2(S, iv, i, j)) { A3(S, vmfltd, i, j, 1 + mm) = 0.; A3(S, vmfltd, i, j, 2 + mm) = 0.; A3(S, vmfltd, i, j, 3 + mm) = 0.; }
    }
  for (int k = 4; k <= kk; k++) {
    const int km = k + mm, kn = k + nn;
    for (int j = 1; j <= jj; j++) {
      for (int i = 1; i <= ii; i++) {
        if (!A2(S, iu, i, j)) continue;
        const double flxhi = .125 * fmin2(A3(S, dp, i - 1, j, kn - 1) * A2(S, scp2, i - 1, j), A3(S, dp, i, j, kn) * A2(S, scp2, i, j));
        const double flxlo = -.125 * fmin2(A3(S, dp, i, j, kn - 1) * A2(S, scp2, i, j), A3(S, dp, i - 1, j, kn) * A2(S, scp2, i - 1, j));
        double q = .25 * (A3(S, difint, i - 1, j, k - 1) + A3(S, difint, i, j, k - 1) + A3(S, difint, i - 1, j, k) + A3(S, difint, i, j, k));
        q = fmin2(flxhi, fmax2(flxlo, delt1 * q * (A3(S, p, i - 1, j, k) - A3(S, p, i, j, k)) * A2(S, scuy, i, j) * A2(S, scuxi, i, j)));
        A3(S, umfltd, i, j, km - 1) = A3(S, umfltd, i, j, km - 1) + q;
        A3(S, umfltd, i, j, km) = -q;
      }
      for (int i = 1; i <= ii; i++) {
        if (!A2(S, iv, i, j)) continue;
        const double flxhi = .125 * fmin2(A3(S, dp, i, j - 1, kn - 1) * A2(S, scp2, i, j - 1), A3(S, dp, i, j, kn) * A2(S, scp2, i, j));
        const double flxlo = -.125 * fmin2(A3(S, dp, i, j, kn - 1) * A2(S, scp2, i, j), A3(S, dp, i, j - 1, kn) * A2(S, scp2, i, j - 1));
        double q = .25 * (A3(S, difint, i, j - 1, k - 1) + A3(S, difint, i, j, k - 1) + A3(S, difint, i, j - 1, k) + A3(S, difint, i, j, k));
        q = fmin2(flxhi, fmax2(flxlo, delt1 * q * (A3(S, p, i, j - 1, k) - A3(S, p, i, j, k)) * A2(S, scvx, i, j) * A2(S, scvyi, i, j)));
        A3(S, vmfltd, i, j, km - 1) = A3(S, vmfltd, i, j, km - 1) + q;
        A3(S, vmfltd, i, j, km) = -q;
      }
    }
  }
}

/* eddtra, mod_eddtra.F90:1808-1857 (isopyc_bulkml branch) */
int orc_eddtra(OState *S, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)k1m; (void)k1n;
  const int ii = S->ii, jj = S->jj, kk = S->kk;
  if (kk > KMAXDIM) { fprintf(stderr, "oracle eddtra: kdm too large\n"); return 1; }
  if (S->eitmth == 1) intdif(S, mm, nn);
  else if (S->eitmth == 2) {
    int bad = 0;
    /* the reference's j-loop carries an OpenMP directive (:247); rows are independent (a column writes only its own fluxes) */
#pragma omp parallel for schedule(dynamic, 4) reduction(| : bad)
    for (int j = 1; j <= jj; j++) {
      for (int i = 1; i <= ii; i++)                           /* :259-268: ptu, then the u-column */
        if (A2(S, iu, i, j)) {
          const size_t xb = IX(S, i, j), xa = xb - 1;
          const double ptu = fmax2(L3(p, xa, 1), L3(p, xb, 1));
          bad |= gm_column(S, xa, xb, n, mm, nn, S->scuy[xb], S->nslpx, S->pbu, S->dpu, S->scu2[xb], S->umfltd, ptu);
        }
      for (int i = 1; i <= ii; i++)
        if (A2(S, iv, i, j)) {
          const size_t xb = IX(S, i, j), xa = xb - (size_t)S->ni;
          const double ptv = fmax2(L3(p, xa, 1), L3(p, xb, 1));
          bad |= gm_column(S, xa, xb, n, mm, nn, S->scvx[xb], S->nslpy, S->pbv, S->dpv, S->scv2[xb], S->vmfltd, ptv);
        }
    }
    if (bad) return 1;
  } else {
    fprintf(stderr, "oracle eddtra: eitmth_opt = %d is unsupported for vcoord = 'isopyc_bulkml'!\n", S->eitmth);
    return 1;
  }
  /* heat and salt components, :1837-1857 */
#pragma omp parallel for
  for (int j = 1; j <= jj; j++)
    for (int k = 1; k <= kk; k++) {
      const int km = k + mm;
      for (int i = 1; i <= ii; i++) {
        if (A2(S, iu, i, j)) {
          A3(S, utfltd, i, j, km) = .5 * A3(S, umfltd, i, j, km) * (A3(S, temp, i - 1, j, km) + A3(S, temp, i, j, km));
          A3(S, usfltd, i, j, km) = .5 * A3(S, umfltd, i, j, km) * (A3(S, saln, i - 1, j, km) + A3(S, saln, i, j, km));
        }
        if (A2(S, iv, i, j)) {
          A3(S, vtfltd, i, j, km) = .5 * A3(S, vmfltd, i, j, km) * (A3(S, temp, i, j - 1, km) + A3(S, temp, i, j, km));
          A3(S, vsfltd, i, j, km) = .5 * A3(S, vmfltd, i, j, km) * (A3(S, saln, i, j - 1, km) + A3(S, saln, i, j, km));
        }
      }
    }
  return 0;
}
