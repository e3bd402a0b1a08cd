/* TEST INFRASTRUCTURE (oracle): restatement of pgforc, phy/mod_pgforc.F90:95-260 and
 * :438-615 (pgfmth = 'geopotential'). */
#include "ostate.h"
#include <stdlib.h>

void orc_p_dpu_dpv(OState *S, int off, int with_pupv);

/* pgforc_geopotential, phy/mod_pgforc.F90:95-260 */
static void pgforc_geopotential(OState *S, int n, int nn) {
  const int ii = S->ii, jj = S->jj, kk = S->kk;
  const size_t lev = (size_t)S->nplane;
  double *phip = (double *)calloc(lev * (kk + 1), sizeof(double));
#define PHIP(i, j, k) phip[IX(S, i, j) + lev * ((k)-1)]
  for (int j = 0; j <= jj; j++) {                                     /* :112-134 */
    for (int i = 0; i <= ii; i++)
      if (A2(S, ip, i, j)) PHIP(i, j, kk + 1) = 0.;
    for (int k = kk; k >= 1; k--) {
      int kn = k + nn;
      for (int i = 0; i <= ii; i++) {
        if (!A2(S, ip, i, j)) continue;
        if (A3(S, dp, i, j, kn) < EPSILP) {
          A3(S, phi, i, j, k) = A3(S, phi, i, j, k + 1);
          PHIP(i, j, k) = PHIP(i, j, k + 1);
        } else {
          double dphi, alpu, alpl;
          eos_delphi(A3(S, p, i, j, k), A3(S, p, i, j, k + 1), A3(S, temp, i, j, kn), A3(S, saln, i, j, kn),
                     &dphi, &alpu, &alpl);
          A3(S, phi, i, j, k) = A3(S, phi, i, j, k + 1) - dphi;
          PHIP(i, j, k) = PHIP(i, j, k + 1) + A3(S, p, i, j, k + 1) * alpl - A3(S, p, i, j, k) * alpu;
        }
      }
    }
  }
  int *kup = (int *)malloc(sizeof(int) * (ii + 1)), *kum = (int *)malloc(sizeof(int) * (ii + 1));
  int *kvp = (int *)malloc(sizeof(int) * (ii + 1)), *kvm = (int *)malloc(sizeof(int) * (ii + 1));
  for (int j = 1; j <= jj; j++) {                                     /* :140-257 */
    for (int i = 1; i <= ii; i++) {
      if (A2(S, iu, i, j)) {
        kup[i] = kk; kum[i] = kk;
        A3(S, xixp, i, j, n) = 0.; A3(S, xixm, i, j, n) = 0.; A3(S, pgfxm, i, j, n) = 0.;
      }
      if (A2(S, iv, i, j)) {
        kvp[i] = kk; kvm[i] = kk;
        A3(S, xiyp, i, j, n) = 0.; A3(S, xiym, i, j, n) = 0.; A3(S, pgfym, i, j, n) = 0.;
      }
    }
    for (int k = kk; k >= 1; k--) {
      int kn = k + nn;
      for (int i = 1; i <= ii; i++) {
        if (!A2(S, iu, i, j)) continue;
        double prs = A3(S, pu, i, j, k + 1) - .5 * A3(S, dpu, i, j, kn);
        while (A3(S, p, i, j, kup[i]) > prs) kup[i]--;
        while (A3(S, p, i - 1, j, kum[i]) > prs) kum[i]--;
        double dphip, alpup, alplp, dphim, alpum, alplm;
        eos_delphi(prs, A3(S, p, i, j, kup[i] + 1), A3(S, temp, i, j, kup[i] + nn), A3(S, saln, i, j, kup[i] + nn),
                   &dphip, &alpup, &alplp);
        eos_delphi(prs, A3(S, p, i - 1, j, kum[i] + 1), A3(S, temp, i - 1, j, kum[i] + nn),
                   A3(S, saln, i - 1, j, kum[i] + nn), &dphim, &alpum, &alplm);
        double cp = .25 * (A3(S, p, i, j, k + 1) + A3(S, p, i, j, k));
        double cm = .25 * (A3(S, p, i - 1, j, k + 1) + A3(S, p, i - 1, j, k));
        double q = prs / (cp + cm);
        cp = q * cp;
        cm = q * cm;
        double phi_p = A3(S, phi, i, j, kup[i] + 1) - dphip;
        A3(S, xixp, i, j, n) = A3(S, xixp, i, j, n) +
                               (PHIP(i, j, kup[i] + 1) + A3(S, p, i, j, kup[i] + 1) * alplp - cp * (alpup - alpum)) *
                                   A3(S, dpu, i, j, kn);
        double phi_m = A3(S, phi, i - 1, j, kum[i] + 1) - dphim;
        A3(S, xixm, i, j, n) = A3(S, xixm, i, j, n) +
                               (PHIP(i - 1, j, kum[i] + 1) + A3(S, p, i - 1, j, kum[i] + 1) * alplm -
                                cm * (alpum - alpup)) * A3(S, dpu, i, j, kn);
        A3(S, pgfx, i, j, kn) = -(phi_p - phi_m);
        A3(S, pgfxm, i, j, n) = A3(S, pgfxm, i, j, n) + A3(S, pgfx, i, j, kn) * A3(S, dpu, i, j, kn);
      }
      for (int i = 1; i <= ii; i++) {
        if (!A2(S, iv, i, j)) continue;
        double prs = A3(S, pv, i, j, k + 1) - .5 * A3(S, dpv, i, j, kn);
        while (A3(S, p, i, j, kvp[i]) > prs) kvp[i]--;
        while (A3(S, p, i, j - 1, kvm[i]) > prs) kvm[i]--;
        double dphip, alpup, alplp, dphim, alpum, alplm;
        eos_delphi(prs, A3(S, p, i, j, kvp[i] + 1), A3(S, temp, i, j, kvp[i] + nn), A3(S, saln, i, j, kvp[i] + nn),
                   &dphip, &alpup, &alplp);
        eos_delphi(prs, A3(S, p, i, j - 1, kvm[i] + 1), A3(S, temp, i, j - 1, kvm[i] + nn),
                   A3(S, saln, i, j - 1, kvm[i] + nn), &dphim, &alpum, &alplm);
        double cp = .25 * (A3(S, p, i, j, k + 1) + A3(S, p, i, j, k));
        double cm = .25 * (A3(S, p, i, j - 1, k + 1) + A3(S, p, i, j - 1, k));
        double q = prs / (cp + cm);
        cp = q * cp;
        cm = q * cm;
        double phi_p = A3(S, phi, i, j, kvp[i] + 1) - dphip;
        A3(S, xiyp, i, j, n) = A3(S, xiyp, i, j, n) +
                               (PHIP(i, j, kvp[i] + 1) + A3(S, p, i, j, kvp[i] + 1) * alplp - cp * (alpup - alpum)) *
                                   A3(S, dpv, i, j, kn);
        double phi_m = A3(S, phi, i, j - 1, kvm[i] + 1) - dphim;
        A3(S, xiym, i, j, n) = A3(S, xiym, i, j, n) +
                               (PHIP(i, j - 1, kvm[i] + 1) + A3(S, p, i, j - 1, kvm[i] + 1) * alplm -
                                cm * (alpum - alpup)) * A3(S, dpv, i, j, kn);
        A3(S, pgfy, i, j, kn) = -(phi_p - phi_m);
        A3(S, pgfym, i, j, n) = A3(S, pgfym, i, j, n) + A3(S, pgfy, i, j, kn) * A3(S, dpv, i, j, kn);
      }
    }
  }
  free(kup); free(kum); free(kvp); free(kvm); free(phip);
#undef PHIP
}

/* pgforc, phy/mod_pgforc.F90:438-615 */
void orc_pgforc(OState *S, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)mm; (void)k1m; (void)k1n;
  const int ii = S->ii, jj = S->jj, kk = S->kk;
  orc_p_dpu_dpv(S, nn, 1);                                            /* :450-485 */
  for (int j = -1; j <= jj + 2; j++)                                  /* :488-505 */
    for (int i = 0; i <= ii + 1; i++) {
      if (A2(S, iu, i, j)) {
        A2(S, xixp_o, i, j) = A3(S, xixp, i, j, n);
        A2(S, xixm_o, i, j) = A3(S, xixm, i, j, n);
        A2(S, pgfxm_o, i, j) = A3(S, pgfxm, i, j, n);
      }
      if (A2(S, iv, i, j)) {
        A2(S, xiyp_o, i, j) = A3(S, xiyp, i, j, n);
        A2(S, xiym_o, i, j) = A3(S, xiym, i, j, n);
        A2(S, pgfym_o, i, j) = A3(S, pgfym, i, j, n);
      }
    }
  for (int j = 1; j <= jj; j++)                                       /* :506-522 */
    for (int k = kk; k >= 1; k--) {
      int kn = k + nn;
      for (int i = 1; i <= ii; i++) {
        if (A2(S, iu, i, j)) A3(S, pgfx_o, i, j, k) = A3(S, pgfx, i, j, kn);
        if (A2(S, iv, i, j)) A3(S, pgfy_o, i, j, k) = A3(S, pgfy, i, j, kn);
      }
    }
  pgforc_geopotential(S, n, nn);                                      /* :525-526 */
  orc_xctilr(S, S->pb_p, 1, 1, 1, 1, 1);                              /* :540 */
  for (int j = 1; j <= jj; j++) {                                     /* :543-597 */
    for (int i = 1; i <= ii; i++) {
      if (A2(S, iu, i, j)) {
        double q = 1. / A2(S, pbu_p, i, j);
        A3(S, pgfxm, i, j, n) = A3(S, pgfxm, i, j, n) * q;
        A3(S, xixp, i, j, n) = A3(S, xixp, i, j, n) * q;
        A3(S, xixm, i, j, n) = A3(S, xixm, i, j, n) * q;
      }
      if (A2(S, iv, i, j)) {
        double q = 1. / A2(S, pbv_p, i, j);
        A3(S, pgfym, i, j, n) = A3(S, pgfym, i, j, n) * q;
        A3(S, xiyp, i, j, n) = A3(S, xiyp, i, j, n) * q;
        A3(S, xiym, i, j, n) = A3(S, xiym, i, j, n) * q;
      }
    }
    for (int k = 1; k <= kk; k++) {
      int kn = k + nn;
      for (int i = 1; i <= ii; i++) {
        if (A2(S, iu, i, j)) A3(S, pgfx, i, j, kn) = A3(S, pgfx, i, j, kn) - A3(S, pgfxm, i, j, n);
        if (A2(S, iv, i, j)) A3(S, pgfy, i, j, kn) = A3(S, pgfy, i, j, kn) - A3(S, pgfym, i, j, n);
      }
    }
    for (int i = 1; i <= ii; i++) {
      if (A2(S, iu, i, j)) {
        A3(S, pgfxm, i, j, n) = A3(S, pgfxm, i, j, n) + A3(S, xixp, i, j, n) - A3(S, xixm, i, j, n);
        A3(S, xixp, i, j, n) = A3(S, xixp, i, j, n) / A2(S, pb_p, i, j);
        A3(S, xixm, i, j, n) = A3(S, xixm, i, j, n) / A2(S, pb_p, i - 1, j);
      }
      if (A2(S, iv, i, j)) {
        A3(S, pgfym, i, j, n) = A3(S, pgfym, i, j, n) + A3(S, xiyp, i, j, n) - A3(S, xiym, i, j, n);
        A3(S, xiyp, i, j, n) = A3(S, xiyp, i, j, n) / A2(S, pb_p, i, j);
        A3(S, xiym, i, j, n) = A3(S, xiym, i, j, n) / A2(S, pb_p, i, j - 1);
      }
      if (A2(S, ip, i, j)) A2(S, sealv, i, j) = A3(S, phi, i, j, 1) / GRAV;
    }
  }
}
