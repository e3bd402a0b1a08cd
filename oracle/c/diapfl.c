/* TEST INFRASTRUCTURE (oracle): restatement of diapfl, phy/mod_diapfl.F90:49-1046
 * (use_TRC without TKE/GLS tracers).  1-based work arrays are declared with one spare
 * element so that the Fortran indices can be used verbatim. */
#include "ostate.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

/* test-side instrumentation: trip-count histograms of the two data-dependent loops */
long orc_diapfl_stat_lim[101], orc_diapfl_stat_niter[101], orc_diapfl_stat_span;

#define KMAXD 130

static void mix_momentum(OState *S, int nn, int k1n, int isv);

void orc_diapfl(OState *S, int n, int nn, int k1n) {
  const int ii = S->ii, jj = S->jj, kk = S->kk, ntr = S->ntr;
  if (kk + 2 > KMAXD) abort();
  const double dsgmnr = .1, fcmxr = .25, dsgcr0 = .25, dfeps = 1.e-12, gbbl = .2, kappa = .4, ustmin = .0001;
  const double c = GRAV * GRAV * S->delt1 / (ALPHA0 * ALPHA0); /* :95 */
  for (int j = 1; j <= jj; j++)
    for (int i = 1; i <= ii; i++) {
      if (!A2(S, ip, i, j)) continue;
      double ttem[KMAXD], ssal[KMAXD], delp[KMAXD], dens[KMAXD], sigr[KMAXD], nu[KMAXD], fpu[KMAXD], fpl[KMAXD],
          fcu[KMAXD], fcl[KMAXD], dsgu[KMAXD], dsgl[KMAXD], dsghm[KMAXD], dsg[KMAXD], dsgui[KMAXD], dsgli[KMAXD],
          fmax[KMAXD], f[KMAXD], f0[KMAXD], fold[KMAXD], h[KMAXD], gtd[KMAXD], pres[KMAXD + 1];
      double ttrc[MAXTR][KMAXD];
      int rstdns[KMAXD];
      for (int k = 1; k <= kk; k++) { /* :114-137 */
        int kn = k + nn;
        ttem[k] = A3(S, temp, i, j, kn);
        ssal[k] = A3(S, saln, i, j, kn);
        delp[k] = A3(S, dp, i, j, kn);
        dens[k] = A3(S, sigma, i, j, kn);
        sigr[k] = A3(S, sigmar, i, j, k);
        nu[k] = A3(S, difdia, i, j, k);
        rstdns[k] = 1;
        for (int nt = 1; nt <= ntr; nt++) ttrc[nt - 1][k] = TRC(S, i, j, kn, nt);
      }
      const int kfpl = A3(S, kfpla, i, j, n); /* :139-144 */
      const int kmin = kfpl - 2;
      int kmax = 1;
      for (int k = 2; k <= kk; k++)
        if (delp[k] > EPSILP) kmax = k;
      if (kmin < kmax) {
        rstdns[kfpl] = 0; /* :150-155 */
        if (kfpl != kmax)
          if (dens[kfpl] > .5 * (sigr[kfpl] + sigr[kfpl + 1])) rstdns[kfpl + 1] = 0;
        delp[kmin + 1] = delp[2]; delp[kmin] = delp[1]; /* :159-172 */
        ttem[kmin + 1] = ttem[2]; ttem[kmin] = ttem[1];
        ssal[kmin + 1] = ssal[2]; ssal[kmin] = ssal[1];
        nu[kmin + 1] = nu[2]; nu[kmin] = nu[1];
        for (int nt = 0; nt < ntr; nt++) { ttrc[nt][kmin + 1] = ttrc[nt][2]; ttrc[nt][kmin] = ttrc[nt][1]; }
        pres[kmin] = 0.; /* :175-178 */
        for (int k = kmin; k <= kmax; k++) pres[k + 1] = pres[k] + delp[k];
        int k = kmin; /* :182-193 */
        fpu[k] = 0.;
        fpl[k] = fmin2(fmin2(pres[k + 1], pres[kmax + 1] - pres[k + 1]),
                       c * nu[k] * (delp[k] + delp[k + 1]) / (2. * delp[k] * delp[k + 1]));
        k = kmin + 1;
        fpu[k] = fpl[k - 1];
        double delpu = fmax2(ONEM, delp[k]), delpl = fmax2(ONEM, delp[k + 1]);
        fpl[k] = fmin2(fmin2(pres[k + 1], pres[kmax + 1] - pres[k + 1]), c * nu[k] * (delpu + delpl) / (2. * delpu * delpl));
        fpl[kmax] = 0.;
        if (kfpl <= kmax) {
          if (kfpl < kmax) { /* :197-209 */
            k = kmax - 1;
            const double us = A2(S, ustarb, i, j);
            double nubbl = gbbl * (us * us * us) *
                           exp(-(delp[k + 1] + .5 * delp[k]) * fabs(A2(S, coriop, i, j)) * ALPHA0 /
                               (kappa * fmax2(ustmin, us) * GRAV)) /
                           (ALPHA0 * GRAV * (sigr[k + 1] - sigr[k]));
            nu[k] = fmax2(nu[k], nubbl);
            A3(S, difdia, i, j, k) = nu[k];
          }
          k = kfpl - 1; /* :217-274 */
          dsgli[k] = 1.;
          fcl[k] = -fpl[k];
          for (k = kfpl; k <= kmax - 1; k++) {
            if (rstdns[k]) {
              double dsgdt = eos_dsigdt(S, ttem[k], ssal[k]), dsgds = eos_dsigds(S, ttem[k], ssal[k]);
              dsgu[k] = fmax2(dsgmnr * (sigr[k] - sigr[k - 1]), dsgdt * (ttem[k] - ttem[k - 1]) + dsgds * (ssal[k] - ssal[k - 1]));
              dsgl[k] = fmax2(dsgmnr * (sigr[k + 1] - sigr[k]), dsgdt * (ttem[k + 1] - ttem[k]) + dsgds * (ssal[k + 1] - ssal[k]));
              dsghm[k] = 2. * dsgu[k] * dsgl[k] / (dsgu[k] + dsgl[k]);
              dsg[k] = .5 * (dsgu[k] + dsgl[k]);
              dsgui[k] = 1. / dsgu[k];
              dsgli[k] = 1. / dsgl[k];
              double fcmx = .25 * (sqrt(delp[k] * delp[k] + 4. * c * nu[k] * dsg[k] * (dsgui[k] + dsgli[k])) - delp[k]) * dsghm[k] * fcmxr;
              double dsgc = dens[k] - sigr[k];
              if (dsgc > 0.) {
                fcl[k] = 0.;
                if (dens[k - 1] < sigr[k]) {
                  double q = fmax2(0., (dens[k] - sigr[k + 1]) / ((sigr[k] - sigr[k + 1]) * (1. - dsgcr0)));
                  q = fmax2(0., 1. - q * q);
                  q = q * q * q;
                  fcu[k] = dsgc * delp[k];
                  fcu[k] = fmin2(q * fcu[k] + (1. - q) * fcmx, fcu[k]);
                } else fcu[k] = 0.;
              } else {
                fcu[k] = 0.;
                if (dens[k + 1] > sigr[k]) {
                  double q = fmax2(0., (dens[k] - sigr[k - 1]) / ((sigr[k] - sigr[k - 1]) * (1. - dsgcr0)));
                  q = fmax2(0., 1. - q * q);
                  q = q * q * q;
                  fcl[k] = dsgc * delp[k];
                  fcl[k] = fmax2(q * fcl[k] - (1. - q) * fcmx, fcl[k]);
                } else fcl[k] = 0.;
              }
            } else {
              dsgu[k] = 1.; dsgl[k] = 1.; dsghm[k] = 1.; dsg[k] = 1.; dsgui[k] = 1.; dsgli[k] = 1.; fcl[k] = 0.; fcu[k] = 0.;
            }
          }
          k = kmax; /* :275-287 */
          {
            double dsgdt = eos_dsigdt(S, ttem[k], ssal[k]), dsgds = eos_dsigds(S, ttem[k], ssal[k]);
            dsgu[k] = fmax2(dsgmnr * (sigr[k] - sigr[k - 1]), dsgdt * (ttem[k] - ttem[k - 1]) + dsgds * (ssal[k] - ssal[k - 1]));
            dsgui[k] = 1. / dsgu[k];
            if (dens[k] > sigr[k] && dens[k - 1] < sigr[k]) fpu[k] = fmin2(delp[k - 1], (dens[k] - sigr[k]) * delp[k] * dsgui[k]);
            else fpu[k] = 0.;
            fcu[k] = fpu[k] * dsgu[k];
          }
          fmax[kfpl - 1] = 0.; /* :292-330 */
          fmax[kmax] = 0.;
          int done = 0, niter = 0, kfmaxu = 0, nlim = 0;
          while (!done) {
            done = 1;
            nlim++;
            for (k = kmax - 1; k >= kfpl; k--) {
              double q = ((fmax[k + 1] + fcu[k + 1]) * dsgui[k + 1] + pres[kmax + 1] - pres[k + 1]) * dsgl[k];
              fcl[k] = fmax2(-q, fcl[k]);
              fmax[k] = q + fcl[k];
            }
            kfmaxu = 0;
            for (k = kfpl; k <= kmax - 1; k++) {
              double q = ((fmax[k - 1] - fcl[k - 1]) * dsgli[k - 1] + pres[k] - pres[kfpl]) * dsgu[k];
              if (fcu[k] > q) { fcu[k] = q; done = 0; }
              if (fmax[k] > q - fcu[k]) { fmax[k] = q - fcu[k]; kfmaxu = k; }
            }
            if (niter == 100) { fprintf(stderr, "oracle diapfl: no convergence in flux limit!\n"); abort(); }
          }
          orc_diapfl_stat_lim[nlim < 100 ? nlim : 100]++;
          k = kfpl - 1; /* :334-353 */
          f0[k] = 0.; f[k] = 0.; gtd[k] = 0.;
          double dflim = 0.;
          for (k = kfpl; k <= kmax - 1; k++) {
            f[k] = fmin2(fmin2(fmax[k], .5 * sqrt(c * nu[k] * dsg[k] * (dsgui[k] + dsgli[k])) * dsghm[k]),
                         c * nu[k] * dsg[k] / fmax2(EPSILP, delp[k]));
            fold[k] = f[k];
            h[k] = fcu[k] * dsgui[k] - fcl[k] * dsgli[k] + fcl[k - 1] * dsgli[k - 1] - fcu[k + 1] * dsgui[k + 1];
            dflim = fmax2(dflim, fmax[k]);
          }
          k = kmax;
          f0[k] = 0.; f[k] = 0.; gtd[k] = 0.;
          dflim = dflim * dfeps;
          niter = 0; /* :357-533 */
          int dwnwrd = 0;
          for (;;) {
            dwnwrd = !dwnwrd;
            double maxdf, ctd, atd, bitd;
            int remfmx = 0;
            if (dwnwrd) {
              ctd = 0.; bitd = 1.;
              for (k = kfpl; k <= kmax - 1; k++) {
                if (remfmx) { gtd[k] = 0.; f0[k] = fmax[k]; f[k] = fmax[k]; }
                else {
                  double q = f0[k - 1] * dsgli[k - 1] + f[k + 1] * dsgui[k + 1] - delp[k] - h[k];
                  double r = 4. * c * nu[k] * dsg[k] * (dsgui[k] + dsgli[k]);
                  double t = .25 * dsghm[k], s, dfdg;
                  if (q < 0.) {
                    s = r / (q * q);
                    if (s < 1.e-3) {
                      r = .00390625 * s;
                      q = -q * r * (128. - s * (32. - s * (16. - s * (10. - s * 7.))));
                      f0[k] = q * t;
                      q = r * (128. - s * (96. - s * (80. - s * (70. - s * 63.))));
                      dfdg = q * t;
                    } else { s = sqrt(q * q + r); f0[k] = (q + s) * t; dfdg = (1. + q / s) * t; }
                  } else { s = sqrt(q * q + r); f0[k] = (q + s) * t; dfdg = (1. + q / s) * t; }
                  if (f0[k] >= fmax[k]) { f0[k] = fmax[k]; dfdg = 0.; if (k > kfmaxu) remfmx = 1; }
                  gtd[k] = ctd * bitd;
                  atd = -dfdg * dsgli[k - 1];
                  ctd = -dfdg * dsgui[k + 1];
                  bitd = 1. / (1. - atd * gtd[k]);
                  f[k] = (f0[k] - atd * (f[k - 1] - f0[k - 1]) + ctd * f[k + 1]) * bitd;
                }
              }
              maxdf = 0.;
              for (k = kmax - 1; k >= kfpl; k--) {
                f[k] = fmin2(fmax[k], f[k] - gtd[k + 1] * f[k + 1]);
                maxdf = fmax2(maxdf, fabs(f[k] - fold[k]));
                fold[k] = f[k];
              }
            } else {
              atd = 0.; bitd = 1.;
              for (k = kmax - 1; k >= kfpl; k--) {
                if (remfmx) { gtd[k] = 0.; f0[k] = fmax[k]; f[k] = fmax[k]; }
                else {
                  double q = f[k - 1] * dsgli[k - 1] + f0[k + 1] * dsgui[k + 1] - delp[k] - h[k];
                  double r = 4. * c * nu[k] * dsg[k] * (dsgui[k] + dsgli[k]);
                  double t = .25 * dsghm[k], s, dfdg;
                  if (q < 0.) {
                    s = r / (q * q);
                    if (s < 1.e-3) {
                      r = .00390625 * s;
                      q = -q * r * (128. - s * (32. - s * (16. - s * (10. - s * 7.))));
                      f0[k] = q * t;
                      q = r * (128. - s * (96. - s * (80. - s * (70. - s * 63.))));
                      dfdg = q * t;
                    } else { s = sqrt(q * q + r); f0[k] = (q + s) * t; dfdg = (1. + q / s) * t; }
                  } else { s = sqrt(q * q + r); f0[k] = (q + s) * t; dfdg = (1. + q / s) * t; }
                  if (f0[k] >= fmax[k]) { f0[k] = fmax[k]; dfdg = 0.; if (k <= kfmaxu) remfmx = 1; }
                  gtd[k] = atd * bitd;
                  atd = -dfdg * dsgli[k - 1];
                  ctd = -dfdg * dsgui[k + 1];
                  bitd = 1. / (1. - ctd * gtd[k]);
                  f[k] = (f0[k] + atd * f[k - 1] - ctd * (f[k + 1] - f0[k + 1])) * bitd;
                }
              }
              maxdf = 0.;
              for (k = kfpl; k <= kmax - 1; k++) {
                f[k] = fmin2(fmax[k], f[k] - gtd[k - 1] * f[k - 1]);
                maxdf = fmax2(maxdf, fabs(f[k] - fold[k]));
                fold[k] = f[k];
              }
            }
            niter = niter + 1;
            if (maxdf <= dflim) { orc_diapfl_stat_niter[niter]++; orc_diapfl_stat_span += kmax - kfpl; break; }
            if (niter == 100) { fprintf(stderr, "oracle diapfl: no convergence in implicit diffusion!\n"); abort(); }
          }
          for (k = kfpl; k <= kmax - 1; k++) { /* :536-540 */
            fpu[k] = (f[k] + fcu[k]) * dsgui[k];
            fpl[k] = (f[k] - fcl[k]) * dsgli[k];
          }
          fpu[kfpl] = fpl[kmin + 1];
        }
        { /* :546-576 */
          double ctd = 0., bitd = 1.;
          for (k = kmin; k <= kmax; k++) {
            gtd[k] = ctd * bitd;
            double q = 1. / (delp[k] + fpu[k] + fpl[k]);
            double atd = -fpu[k] * q;
            ctd = -fpl[k] * q;
            double dtd = delp[k] * q;
            bitd = 1. / (1. - atd * gtd[k]);
            int km1 = imax2(1, k - 1);
            ssal[k] = (dtd * ssal[k] - atd * ssal[km1]) * bitd;
            ttem[k] = (dtd * ttem[k] - atd * ttem[km1]) * bitd;
            for (int nt = 0; nt < ntr; nt++) ttrc[nt][k] = (dtd * ttrc[nt][k] - atd * ttrc[nt][km1]) * bitd;
          }
          for (k = kmax - 1; k >= kmin; k--) {
            ssal[k] = ssal[k] - gtd[k + 1] * ssal[k + 1];
            ttem[k] = ttem[k] - gtd[k + 1] * ttem[k + 1];
            dens[k] = eos_sig(S, ttem[k], ssal[k]);
            for (int nt = 0; nt < ntr; nt++) ttrc[nt][k] = ttrc[nt][k] - gtd[k + 1] * ttrc[nt][k + 1];
          }
          for (k = kfpl; k <= kmax - 1; k++) delp[k] = fmax2(0., delp[k] + fpu[k] + fpl[k] - fpl[k - 1] - fpu[k + 1]);
          delp[kmax] = fmax2(0., delp[kmax] + fpu[kmax] - fpl[kmax - 1]);
        }
        ttem[1] = ttem[kmin]; ttem[2] = ttem[kmin + 1]; /* :580-599 */
        ssal[1] = ssal[kmin]; ssal[2] = ssal[kmin + 1];
        dens[1] = dens[kmin]; dens[2] = dens[kmin + 1];
        if (kmin > 1) {
          if (kmin == 2) { delp[2] = delp[kmin + 1]; delp[kmin + 1] = 0.; }
          else delp[kmin] = 0.;
        }
        for (int nt = 0; nt < ntr; nt++) { ttrc[nt][1] = ttrc[nt][kmin]; ttrc[nt][2] = ttrc[nt][kmin + 1]; }
      }
      if (kfpl > kmax) { /* :605-651 */
        for (int k = 3; k <= kk; k++) {
          ttem[k] = fmax2(ttem[2], A3(S, temmin, i, j, k));
          dens[k] = sigr[k];
          ssal[k] = eos_sofsig(S, dens[k], ttem[k]);
          delp[k] = 0.;
          for (int nt = 0; nt < ntr; nt++) {                               /* :612-626 */
            if (S->itrtke >= 1 && nt + 1 == S->itrtke) ttrc[nt][k] = fmax2(ttrc[nt][2], ORC_TKE_MIN);
            else if (S->itrtke >= 1 && S->gls && nt + 1 == S->itrgls) ttrc[nt][k] = fmax2(ttrc[nt][2], ORC_GLS_PSI_MIN);
            else ttrc[nt][k] = ttrc[nt][2];
          }
        }
      } else {
        for (int k = 3; k <= kfpl - 1; k++) {
          ttem[k] = ttem[kfpl];
          dens[k] = sigr[k];
          ssal[k] = eos_sofsig(S, dens[k], ttem[k]);
          delp[k] = 0.;
          for (int nt = 0; nt < ntr; nt++) ttrc[nt][k] = ttrc[nt][kfpl];
        }
        for (int k = kmax + 1; k <= kk; k++) {
          ttem[k] = ttem[kmax];
          dens[k] = sigr[k];
          ssal[k] = eos_sofsig(S, dens[k], ttem[k]);
          for (int nt = 0; nt < ntr; nt++) ttrc[nt][k] = ttrc[nt][kmax];
        }
      }
      for (int k = 1; k <= kk; k++) { /* :654-678 */
        int kn = k + nn;
        A3(S, temp, i, j, kn) = ttem[k];
        A3(S, saln, i, j, kn) = ssal[k];
        A3(S, dp, i, j, kn) = delp[k];
        A3(S, sigma, i, j, kn) = dens[k];
        A3(S, p, i, j, k + 1) = A3(S, p, i, j, k) + A3(S, dp, i, j, kn);
        for (int nt = 1; nt <= ntr; nt++) {                                /* :662-677 */
          if (S->itrtke >= 1 && nt == S->itrtke) TRC(S, i, j, kn, nt) = fmax2(ttrc[nt - 1][k], ORC_TKE_MIN);
          else if (S->itrtke >= 1 && S->gls && nt == S->itrgls) TRC(S, i, j, kn, nt) = fmax2(ttrc[nt - 1][k], ORC_GLS_PSI_MIN);
          else TRC(S, i, j, kn, nt) = ttrc[nt - 1][k];
        }
      }
      A2(S, kming, i, j) = kmin; /* :681-700 */
      if (kmin < kmax) {
        for (int k = 1; k <= kmin; k++) { A3(S, fpug, i, j, k) = fpl[kmin]; A3(S, fplg, i, j, k) = fpl[kmin]; }
        for (int k = kmin + 1; k <= kmax; k++) { A3(S, fpug, i, j, k) = fpu[k]; A3(S, fplg, i, j, k) = fpl[k]; }
        for (int k = kmax + 1; k <= kk; k++) { A3(S, fpug, i, j, k) = 0.; A3(S, fplg, i, j, k) = 0.; }
      } else
        for (int k = 1; k <= kk; k++) { A3(S, fpug, i, j, k) = 0.; A3(S, fplg, i, j, k) = 0.; }
    }
  /* :711-733 */
  orc_xctilr(S, S->p, 1, kk + 1, 1, 1, 1);
  orc_xctilr(S, S->fpug, 1, kk, 1, 1, 1);
  orc_xctilr(S, S->fplg, 1, kk, 1, 1, 1);
  for (int j = 1; j <= jj; j++)
    for (int i = 1; i <= ii; i++)
      if (A2(S, ip, i, j)) A2(S, util1, i, j) = A2(S, kming, i, j);
  orc_xctilr(S, S->util1, 1, 1, 1, 1, 1);
  for (int j = 0; j <= jj + 1; j++)
    for (int i = 0; i <= ii + 1; i++)
      if (A2(S, ip, i, j)) A2(S, kming, i, j) = (int)nearbyint(A2(S, util1, i, j));
  mix_momentum(S, nn, k1n, 0); /* :738-852 */
  mix_momentum(S, nn, k1n, 1); /* :856-966 */
  for (int j = 1; j <= jj; j++) /* :971-983 */
    for (int k = 1; k <= kk; k++)
      for (int i = 1; i <= ii + 1; i++)
        if (A2(S, iu, i, j)) {
          double q = fmin2(A3(S, p, i, j, kk + 1), A3(S, p, i - 1, j, kk + 1));
          A3(S, dpu, i, j, k + nn) = .5 * ((fmin2(q, A3(S, p, i - 1, j, k + 1)) - fmin2(q, A3(S, p, i - 1, j, k))) +
                                           (fmin2(q, A3(S, p, i, j, k + 1)) - fmin2(q, A3(S, p, i, j, k))));
        }
  for (int j = 1; j <= jj + 1; j++) /* :987-999 */
    for (int k = 1; k <= kk; k++)
      for (int i = 1; i <= ii; i++)
        if (A2(S, iv, i, j)) {
          double q = fmin2(A3(S, p, i, j, kk + 1), A3(S, p, i, j - 1, kk + 1));
          A3(S, dpv, i, j, k + nn) = .5 * ((fmin2(q, A3(S, p, i, j - 1, k + 1)) - fmin2(q, A3(S, p, i, j - 1, k))) +
                                           (fmin2(q, A3(S, p, i, j, k + 1)) - fmin2(q, A3(S, p, i, j, k))));
        }
}

/* diapycnal mixing of one velocity component, :740-852 (u) and :856-966 (v) */
static void mix_momentum(OState *S, int nn, int k1n, int isv) {
  const int ii = S->ii, jj = S->jj, kk = S->kk;
  double *vel = isv ? S->v : S->u, *dpz = isv ? S->dpv : S->dpu, *pz = isv ? S->pv : S->pu;
  const int *msk = isv ? S->iv : S->iu;
  const size_t lev = (size_t)S->nplane;
#define VEL(i, j, k) vel[IX(S, i, j) + lev * ((k)-1)]
#define DPZ(i, j, k) dpz[IX(S, i, j) + lev * ((k)-1)]
  for (int j = 1; j <= jj; j++)
    for (int i = 1; i <= ii; i++) {
      if (!msk[IX(S, i, j)]) continue;
      const int im = isv ? i : i - 1, jm = isv ? j - 1 : j;
      double uc[KMAXD], delp[KMAXD], fpu[KMAXD], fpl[KMAXD], gtd[KMAXD];
      const int kmin = imin2(A2(S, kming, im, jm), A2(S, kming, i, j));
      int kmax = 1;
      for (int k = 2; k <= kk; k++)
        if (DPZ(i, j, k + nn) > 0.) kmax = k;
      if (!(kmin < kmax)) continue;
      uc[kmin + 1] = VEL(i, j, k1n + 1);
      uc[kmin] = VEL(i, j, k1n);
      delp[kmin + 1] = DPZ(i, j, k1n + 1);
      delp[kmin] = DPZ(i, j, k1n);
      for (int k = kmin + 2; k <= kmax; k++) { uc[k] = VEL(i, j, k + nn); delp[k] = DPZ(i, j, k + nn); }
      const double pzb = pz[IX(S, i, j) + lev * kk];
      fpu[kmin] = 0.;
      for (int k = kmin + 1; k <= kmax; k++) {
        double fpum, fplm, fpup, fplp;
        double pold = A3(S, p, im, jm, k) - A3(S, fplg, im, jm, k - 1) + A3(S, fpug, im, jm, k);
        double pnew = A3(S, p, im, jm, k);
        if (pold <= pzb) {
          if (pnew <= pzb) { fpum = A3(S, fpug, im, jm, k); fplm = A3(S, fplg, im, jm, k - 1); }
          else { fpum = A3(S, fpug, im, jm, k); fplm = A3(S, fplg, im, jm, k - 1) - pnew + pzb; }
        } else {
          if (pnew <= pzb) { fpum = A3(S, fpug, im, jm, k) - pold + pzb; fplm = A3(S, fplg, im, jm, k - 1); }
          else { fpum = .5 * (A3(S, fpug, im, jm, k) + A3(S, fplg, im, jm, k - 1)); fplm = fpum; }
        }
        pold = A3(S, p, i, j, k) - A3(S, fplg, i, j, k - 1) + A3(S, fpug, i, j, k);
        pnew = A3(S, p, i, j, k);
        if (pold <= pzb) {
          if (pnew <= pzb) { fpup = A3(S, fpug, i, j, k); fplp = A3(S, fplg, i, j, k - 1); }
          else { fpup = A3(S, fpug, i, j, k); fplp = A3(S, fplg, i, j, k - 1) - pnew + pzb; }
        } else {
          if (pnew <= pzb) { fpup = A3(S, fpug, i, j, k) - pold + pzb; fplp = A3(S, fplg, i, j, k - 1); }
          else { fpup = .5 * (A3(S, fpug, i, j, k) + A3(S, fplg, i, j, k - 1)); fplp = fpup; }
        }
        fpu[k] = .5 * (fpum + fpup);
        fpl[k - 1] = .5 * (fplm + fplp);
      }
      fpl[kmax] = 0.;
      double ctd = 0., bitd = 1.;
      for (int k = kmin; k <= kmax; k++) {
        gtd[k] = ctd * bitd;
        double q = 1. / (delp[k] + fpu[k] + fpl[k]);
        double atd = -fpu[k] * q;
        ctd = -fpl[k] * q;
        double dtd = delp[k] * q;
        bitd = 1. / (1. - atd * gtd[k]);
        uc[k] = (dtd * uc[k] - atd * uc[imax2(kmin, k - 1)]) * bitd;
      }
      for (int k = kmax - 1; k >= kmin; k--) uc[k] = uc[k] - gtd[k + 1] * uc[k + 1];
      VEL(i, j, k1n) = uc[kmin];
      VEL(i, j, k1n + 1) = uc[kmin + 1];
      for (int k = kmin + 2; k <= kmax; k++) VEL(i, j, k + nn) = uc[k];
      for (int k = kmax + 1; k <= kk; k++)
        if (fmin2(A3(S, p, im, jm, k), A3(S, p, i, j, k)) < pzb) VEL(i, j, k + nn) = uc[kmax];
    }
#undef VEL
#undef DPZ
}
