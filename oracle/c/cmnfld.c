/* TEST INFRASTRUCTURE (oracle): restatement of cmnfld2 for vcoord = isopyc_bulkml with eitmth = 'gm',
 * phy/mod_cmnfld_routines.F90:1158-1238: halo updates (:1171-1196), cmnfld_bfsqf_isopyc_bulkml (:61-227) and
 * cmnfld_nslope_isopyc_bulkml (:423-652).
 *
 * PARITY UNPINNED: mod_cmnfld_routines uses mod_dia (netCDF) and cannot be compiled in this image, and the
 * reference's tests hold no vectors for it; this restatement is checked by construction only
 * (tests/test_cmnfld.py: N^2 of a linear stratification, the filter leaves a uniform field alone, the slope of a
 * tilted stack of layers).  It follows the Fortran loop by loop with its local arrays.
 */
#include "ostate.h"
#include <math.h>
#include <stdio.h>

#define RHO0 1.e3
#define SLS0 (10. * ONEM)      /* phy/mod_cmnfld.F90:36-46 */
#define SLSMFQ 2.
#define SLSELS 2.
#define BFSQMN 1.e-7
#define KMAXDIM 256

static void bfsqf_column(OState *S, int i, int j, int n, int nn) {
  const int kk = S->kk;
  double delp[KMAXDIM + 2], bfsq[KMAXDIM + 2], sls2[KMAXDIM + 2], atd[KMAXDIM + 2], btd[KMAXDIM + 2], ctd[KMAXDIM + 2],
         rtd[KMAXDIM + 2], gam[KMAXDIM + 2];
  int k;
#define P_(k) A3(S, p, i, j, k)
#define T_(k) A3(S, temp, i, j, (k) + nn)
#define S_(k) A3(S, saln, i, j, (k) + nn)
#define BI(k) A3(S, bfsqi, i, j, k)
#define BL(k) A3(S, bfsql, i, j, k)
#define BF(k) A3(S, bfsqf, i, j, k)
  BI(1) = .5 * GRAV * GRAV * (eos_rho(P_(2), T_(2), S_(2)) - eos_rho(P_(2), T_(1), S_(1))) /
          (A3(S, dp, i, j, 1 + nn) + A3(S, dp, i, j, 2 + nn));
  BI(2) = BI(1);
  BL(1) = BI(1);
  BL(2) = BI(1);
  const int kfpl = A3(S, kfpla, i, j, n);
  if (kfpl > kk) {
    for (k = 3; k <= kk; k++) { BI(k) = BI(1); BL(k) = BI(1); }
    BI(kk + 1) = BI(1);
    for (k = 1; k <= kk + 1; k++) BF(k) = BFSQMN;
    return;
  }
  const double pml = fmax2(.5 * (P_(3) + P_(1)), .5 * (3. * P_(3) - P_(kfpl + 1)));
  delp[kfpl - 1] = pml - P_(1);
  BI(kfpl - 1) = BI(2);
  bfsq[kfpl - 1] = BFSQMN;
  double q = fmax2(SLS0, delp[kfpl - 1] * SLSMFQ);
  sls2[kfpl - 1] = q * q;
  double pup = pml, tup = T_(2), sup = S_(2), plo, tlo, slo;
  for (k = kfpl; k <= kk; k++) {
    if (P_(kk + 1) - P_(k) < EPSILP) {
      delp[k] = ONEMM;
      BI(k) = BI(k - 1);
      bfsq[k] = BFSQMN;
      q = exp(-(P_(kk + 1) - pml) / (SLSELS * delp[kfpl - 1]));
      q = fmax2(SLS0, delp[kfpl - 1] * SLSMFQ * q + SLS0 * (1. - q));
      sls2[k] = q * q;
    } else {
      if (P_(kk + 1) - P_(k + 1) < EPSILP) plo = P_(kk + 1);
      else plo = .5 * (P_(k) + P_(k + 1));
      tlo = T_(k);
      slo = S_(k);
      delp[k] = fmax2(ONEMM, plo - pup);
      BI(k) = GRAV * GRAV * (eos_rho(P_(k), tlo, slo) - eos_rho(P_(k), tup, sup)) / delp[k];
      bfsq[k] = fmax2(BFSQMN, BI(k));
      BI(k) = BI(k) * delp[k] / fmax2(ONEM, delp[k]);
      if (P_(kk + 1) - P_(k) < ONEM) BI(k) = BI(k - 1);
      q = exp(-(P_(k) - pml) / (SLSELS * delp[kfpl - 1]));
      q = fmax2(SLS0, delp[kfpl - 1] * SLSMFQ * q + SLS0 * (1. - q));
      sls2[k] = q * q;
      pup = plo;
      tup = tlo;
      sup = slo;
    }
  }
  for (k = kfpl; k <= kk - 1; k++) BL(k) = .5 * (BI(k) + BI(k + 1));
  BL(kk) = BI(kk);
  for (k = 3; k <= kfpl - 1; k++) { BI(k) = BI(kfpl); BL(k) = BL(kfpl); }
  k = kfpl - 1;
  ctd[k] = -2. * sls2[k] / (delp[k] * (delp[k] + delp[k + 1]));
  btd[k] = 1. - ctd[k];
  rtd[k] = bfsq[k];
  for (k = kfpl; k <= kk - 1; k++) {
    atd[k] = -2. * sls2[k - 1] / (delp[k] * (delp[k - 1] + delp[k]));
    ctd[k] = -2. * sls2[k] / (delp[k] * (delp[k] + delp[k + 1]));
    btd[k] = 1. - atd[k] - ctd[k];
    rtd[k] = bfsq[k];
  }
  k = kk;
  atd[k] = -2. * sls2[k - 1] / (delp[k] * (delp[k - 1] + delp[k]));
  btd[k] = 1. - atd[k];
  rtd[k] = bfsq[k];
  double bei = 1. / btd[kfpl - 1];
  BF(kfpl - 1) = rtd[kfpl - 1] * bei;
  for (k = kfpl; k <= kk; k++) {
    gam[k] = ctd[k - 1] * bei;
    bei = 1. / (btd[k] - atd[k] * gam[k]);
    BF(k) = (rtd[k] - atd[k] * BF(k - 1)) * bei;
  }
  for (k = kk - 1; k >= kfpl - 1; k--) BF(k) = BF(k) - gam[k + 1] * BF(k + 1);
  for (k = 1; k <= kfpl - 2; k++) BF(k) = BF(kfpl - 1);
  BI(kk + 1) = BI(kk);
  BF(kk + 1) = BF(kk);
#undef P_
#undef T_
#undef S_
#undef BI
#undef BL
#undef BF
}

/* one velocity point: (ia,ja) = (i-1,j) | (i,j-1), the u- and v-halves of :465-641 being mirror images */
static void nslope_column(OState *S, int i, int j, int ia, int ja, int n, int nn, double *nslp, double *nnslp, double sci) {
  const int kk = S->kk;
  const size_t x = IX(S, i, j), np = (size_t)S->nplane;
  int k, kn;
#define NS(k) nslp[x + np * ((k)-1)]
#define NN(k) nnslp[x + np * ((k)-1)]
  for (k = 1; k <= kk; k++) { NS(k) = 0.; NN(k) = 0.; }
  const int kfa = A3(S, kfpla, ia, ja, n), kfb = A3(S, kfpla, i, j, n);
  if (!(kfa <= kk || kfb <= kk)) return;
  int kmax = 1;
  for (k = 3; k <= kk; k++) {
    kn = k + nn;
    if (A3(S, dp, ia, ja, kn) > EPSILP || A3(S, dp, i, j, kn) > EPSILP) kmax = k;
  }
  const int kintr = imax2(kfa, kfb);
  int knnsl = 2;
  double pm = .5 * (A3(S, p, ia, ja, 3) + A3(S, p, i, j, 3));
  double rho_x = eos_rho(pm, A3(S, temp, i, j, 2 + nn), A3(S, saln, i, j, 2 + nn)) -
                 eos_rho(pm, A3(S, temp, ia, ja, 2 + nn), A3(S, saln, ia, ja, 2 + nn));
  double phi_x = A3(S, phi, i, j, 3) - A3(S, phi, ia, ja, 3);
  double bfsqm = .5 * (A3(S, bfsqf, ia, ja, 3) + A3(S, bfsqf, i, j, 3));
  NS(3) = (GRAV * rho_x / (RHO0 * bfsqm) + phi_x / GRAV) * sci;
  if (A3(S, phi, i, j, 3) > A3(S, phi, ia, ja, kk + 1) && A3(S, phi, ia, ja, 3) > A3(S, phi, i, j, kk + 1)) {
    NN(3) = sqrt(bfsqm) * NS(3);
    knnsl = 3;
  }
  for (k = kintr + 1; k <= kmax; k++) {
    kn = k + nn;
    pm = .5 * (A3(S, p, ia, ja, k) + A3(S, p, i, j, k));
    rho_x = .5 * (eos_rho(pm, A3(S, temp, i, j, kn - 1), A3(S, saln, i, j, kn - 1)) -
                  eos_rho(pm, A3(S, temp, ia, ja, kn - 1), A3(S, saln, ia, ja, kn - 1)) +
                  eos_rho(pm, A3(S, temp, i, j, kn), A3(S, saln, i, j, kn)) -
                  eos_rho(pm, A3(S, temp, ia, ja, kn), A3(S, saln, ia, ja, kn)));
    phi_x = A3(S, phi, i, j, k) - A3(S, phi, ia, ja, k);
    bfsqm = .5 * (A3(S, bfsqf, ia, ja, k) + A3(S, bfsqf, i, j, k));
    NS(k) = (GRAV * rho_x / (RHO0 * bfsqm) + phi_x / GRAV) * sci;
    if (A3(S, phi, i, j, k) > A3(S, phi, ia, ja, kk + 1) && A3(S, phi, ia, ja, k) > A3(S, phi, i, j, kk + 1)) {
      NN(k) = sqrt(bfsqm) * NS(k);
      knnsl = k;
    }
  }
  for (k = knnsl + 1; k <= kmax; k++) NN(k) = NN(knnsl);
  if (kintr < kmax) {
    for (k = 4; k <= kintr; k++) { NS(k) = NS(kintr + 1); NN(k) = NN(kintr + 1); }
  } else {
    for (k = 4; k <= kmax; k++) { NS(k) = NS(3); NN(k) = NN(3); }
  }
#undef NS
#undef NN
}

int orc_cmnfld2(OState *S, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)mm; (void)k1m; (void)k1n;
  const int ii = S->ii, jj = S->jj, kk = S->kk;
  if (kk > KMAXDIM) { fprintf(stderr, "oracle cmnfld2: kdm too large\n"); return 1; }
  orc_xctilr(S, S->temp, 1, 2 * kk, 3, 3, 1);                     /* :1171-1172 */
  orc_xctilr(S, S->saln, 1, 2 * kk, 3, 3, 1);
  for (int j = 1; j <= jj; j++)                                   /* kfpla halo through util1, :1176-1196 */
    for (int i = 1; i <= ii; i++)
      if (A2(S, ip, i, j)) A2(S, util1, i, j) = (double)A3(S, kfpla, i, j, n);
  orc_xctilr(S, S->util1, 1, 1, 2, 2, 1);
  for (int j = -1; j <= jj + 2; j++)
    for (int i = -1; i <= ii + 2; i++)
      if (A2(S, ip, i, j)) A3(S, kfpla, i, j, n) = (int)lround(A2(S, util1, i, j));
  if (S->eitmth != 2) return 0;
#pragma omp parallel for schedule(dynamic, 4)
  for (int j = -1; j <= jj + 2; j++)                              /* :83-212 */
    for (int i = -1; i <= ii + 2; i++)
      if (A2(S, ip, i, j)) bfsqf_column(S, i, j, n, nn);
#pragma omp parallel for schedule(dynamic, 4)
  for (int j = -1; j <= jj + 2; j++)                              /* :437-455 */
    for (int k = kk; k >= 1; k--)
      for (int i = -1; i <= ii + 2; i++)
        if (A2(S, ip, i, j)) {
          const int kn = k + nn;
          if (A3(S, dp, i, j, kn) < EPSILP) A3(S, phi, i, j, k) = A3(S, phi, i, j, k + 1);
          else A3(S, phi, i, j, k) = A3(S, phi, i, j, k + 1) - eos_p_alpha(A3(S, p, i, j, k + 1), A3(S, p, i, j, k), A3(S, temp, i, j, kn), A3(S, saln, i, j, kn));
        }
#pragma omp parallel for schedule(dynamic, 4)
  for (int j = -1; j <= jj + 2; j++)                              /* :465-550 */
    for (int i = 0; i <= ii + 2; i++)
      if (A2(S, iu, i, j)) nslope_column(S, i, j, i - 1, j, n, nn, S->nslpx, S->nnslpx, A2(S, scuxi, i, j));
#pragma omp parallel for schedule(dynamic, 4)
  for (int j = 0; j <= jj + 2; j++)                               /* :556-641 */
    for (int i = -1; i <= ii + 2; i++)
      if (A2(S, iv, i, j)) nslope_column(S, i, j, i, j - 1, n, nn, S->nslpy, S->nnslpy, A2(S, scvyi, i, j));
  return 0;
}

/* cmnfld1 for vcoord = isopyc_bulkml without the mixed-layer-depth diagnostics: cmnfld_z, depth of the layer interfaces
   and thickness of the layers, phy/mod_cmnfld_routines.F90:885-921 (called at :1106).  The depth of the sea floor is
   minus its geopotential over g; going up, an interface lies p_alpha(p(k+1), p(k), T, S)/g above the one below it,
   layers without mass have no thickness. */
int orc_cmnfld1(OState *S, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)n; (void)nn; (void)k1m; (void)k1n;
  const int ii = S->ii, jj = S->jj, kk = S->kk;
  for (int j = 1; j <= jj; j++)
    for (int i = 1; i <= ii; i++)
      if (A2(S, ip, i, j)) A3(S, z, i, j, kk + 1) = -A3(S, phi, i, j, kk + 1) / GRAV;
  for (int j = 1; j <= jj; j++)
    for (int k = kk; k >= 1; k--) {
      const int km = k + mm;
      for (int i = 1; i <= ii; i++)
        if (A2(S, ip, i, j)) {
          if (A3(S, dp, i, j, km) < EPSILP) A3(S, z, i, j, k) = A3(S, z, i, j, k + 1);
          else
            A3(S, z, i, j, k) = A3(S, z, i, j, k + 1) +
                                eos_p_alpha(A3(S, p, i, j, k + 1), A3(S, p, i, j, k), A3(S, temp, i, j, km), A3(S, saln, i, j, km)) / GRAV;
          A3(S, dz, i, j, k) = A3(S, z, i, j, k + 1) - A3(S, z, i, j, k);
        }
    }
  return 0;
}
