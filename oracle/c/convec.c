/* TEST INFRASTRUCTURE (oracle): restatement of convec, phy/mod_convec.F90:43-449 (removal of static
 * instabilities between the mixed layer and the interior, then redistribution of momentum onto the new
 * layer structure).  1-based work arrays are declared with one spare element so that the Fortran
 * indices can be used verbatim. */
#include "ostate.h"
#include <stdio.h>
#include <stdlib.h>

#define KMAXC 130

static void remap_velocity(OState *S, int nn, int isv);

void orc_convec(OState *S, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)mm; (void)k1m; (void)k1n;
  const int ii = S->ii, jj = S->jj, kk = S->kk, ntr = S->ntr;
  if (kk + 2 > KMAXC) abort();
  for (int j = 1; j <= jj; j++)
    for (int i = 1; i <= ii; i++) {
      if (!A2(S, ip, i, j)) continue;
      double ttem[KMAXC], ssal[KMAXC], delp[KMAXC], dens[KMAXC], densr[KMAXC], ttrc[MAXTR][KMAXC], trdps[MAXTR];
      double tdps, sdps, dps, ttmp, stmp, dtmp, q = 0.;
      for (int k = 1; k <= kk; k++) { /* :76-89 */
        const int kn = k + nn;
        ttem[k] = A3(S, temp, i, j, kn);
        ssal[k] = A3(S, saln, i, j, kn);
        delp[k] = A3(S, dp, i, j, kn);
        dens[k] = A3(S, sigma, i, j, kn);
        densr[k] = A3(S, sigmar, i, j, k);
        for (int nt = 1; nt <= ntr; nt++) ttrc[nt - 1][k] = TRC(S, i, j, kn, nt);
      }
      /* first physical interior layer, :95-109 */
      int k = 3;
      dps = 0.;
      while (delp[k] < EPSILP) {
        dps = dps + delp[k];
        delp[k] = 0.;
        k = k + 1;
        if (k > kk) break;
      }
      if (k > kk) delp[2] = delp[2] + dps;
      else delp[k] = delp[k] + dps;
      int kfpl = k;
      const int kfplo = A3(S, kfpla, i, j, n);
      if (kfpl < kfplo) { /* :110-191 */
        tdps = 0.; sdps = 0.; dps = 0.;
        for (int nt = 0; nt < ntr; nt++) trdps[nt] = 0.;
        if (kfplo <= kk) {
          for (k = kfpl; k <= kfplo; k++) {
            tdps = tdps + ttem[k] * delp[k];
            sdps = sdps + ssal[k] * delp[k];
            dps = dps + delp[k];
            for (int nt = 0; nt < ntr; nt++) trdps[nt] = trdps[nt] + ttrc[nt][k] * delp[k];
          }
          q = 1. / dps;
          ttmp = tdps * q;
          stmp = sdps * q;
          dtmp = eos_sig(S, ttmp, stmp);
          if (dtmp > densr[kfplo]) {
            for (k = kfpl; k <= kfplo - 1; k++) delp[k] = 0.;
            kfpl = kfplo;
            ttem[kfpl] = ttmp; ssal[kfpl] = stmp; dens[kfpl] = dtmp; delp[kfpl] = dps;
            for (int nt = 0; nt < ntr; nt++) ttrc[nt][kfpl] = trdps[nt] * q;
          }
        } else {
          for (k = kfpl; k <= kk; k++) {
            tdps = tdps + ttem[k] * delp[k];
            sdps = sdps + ssal[k] * delp[k];
            dps = dps + delp[k];
            for (int nt = 0; nt < ntr; nt++) trdps[nt] = trdps[nt] + ttrc[nt][k] * delp[k];
            delp[k] = 0.;
          }
          q = 1. / dps;
          ttmp = tdps * q;
          stmp = sdps * q;
          dtmp = eos_sig(S, ttmp, stmp);
          kfpl = kk;
          while (dtmp < densr[kfpl]) {
            if (kfpl == 3) break;
            kfpl = kfpl - 1;
          }
          ttem[kfpl] = ttmp; ssal[kfpl] = stmp; dens[kfpl] = dtmp; delp[kfpl] = dps;
          for (int nt = 0; nt < ntr; nt++) ttrc[nt][kfpl] = trdps[nt] * q;
        }
      }
      if (kfpl <= kk) { /* :193-283 */
        int done = 0, niter = 0;
        while (!done) {
          niter = niter + 1;
          if (niter == 100) {
            printf(" blom: convec: no convergence! %d %d\n", i, j);
            break;
          }
          done = 1;
          tdps = ttem[2] * delp[2];
          sdps = ssal[2] * delp[2];
          dps = delp[2];
          for (int nt = 0; nt < ntr; nt++) trdps[nt] = ttrc[nt][2] * delp[2];
          ttmp = ttem[2];
          stmp = ssal[2];
          k = kfpl;
          while (eos_rho(dps, ttmp, stmp) > eos_rho(dps, ttem[k], ssal[k]) || delp[k] < EPSILP) {
            tdps = tdps + ttem[k] * delp[k];
            sdps = sdps + ssal[k] * delp[k];
            dps = dps + delp[k];
            q = 1. / dps;
            ttmp = tdps * q;
            stmp = sdps * q;
            for (int nt = 0; nt < ntr; nt++) trdps[nt] = trdps[nt] + ttrc[nt][k] * delp[k];
            k = k + 1;
            if (k > kk) break;
          }
          const int kmix = k - 1;
          if (kmix >= kfpl) {
            ttem[2] = ttmp;
            ssal[2] = stmp;
            dens[2] = eos_sig(S, ttem[2], ssal[2]);
            for (int nt = 0; nt < ntr; nt++) ttrc[nt][2] = trdps[nt] * q;
            dps = 0.;
            for (k = kfpl; k <= kmix; k++) {
              dps = dps + delp[k];
              delp[k] = 0.;
            }
            k = kmix;
            while (dens[2] < densr[k]) {
              if (k == 3) break;
              k = k - 1;
            }
            kfpl = k;
            ttem[kfpl] = ttem[2]; ssal[kfpl] = ssal[2]; dens[kfpl] = dens[2]; delp[kfpl] = dps;
            for (int nt = 0; nt < ntr; nt++) ttrc[nt][kfpl] = ttrc[nt][2];
            for (k = kfpl + 1; k <= kmix; k++) {
              ttem[k] = ttem[2];
              dens[k] = densr[k];
              ssal[k] = eos_sofsig(S, dens[k], ttem[k]);
            }
          }
        }
      }
      A3(S, kfpla, i, j, n) = kfpl;
      for (k = 1; k <= kk; k++) { /* :288-302 */
        const int kn = k + nn;
        A3(S, temp, i, j, kn) = ttem[k];
        A3(S, saln, i, j, kn) = ssal[k];
        A3(S, sigma, i, j, kn) = dens[k];
        A3(S, dp, i, j, kn) = delp[k];
        A3(S, p, i, j, k + 1) = A3(S, p, i, j, k) + A3(S, dp, i, j, kn);
        for (int nt = 1; nt <= ntr; nt++) TRC(S, i, j, kn, nt) = ttrc[nt - 1][k];
      }
    }

  orc_xctilr(S, S->p, 1, kk + 1, 1, 1, 1); /* :313, halo_ps */
  remap_velocity(S, nn, 0);
  remap_velocity(S, nn, 1);

  for (int j = 1; j <= jj; j++) /* :393-414 */
    for (int k = 1; k <= kk; k++) {
      const int kn = k + nn;
      for (int i = 1; i <= ii; i++) {
        if (A2(S, iu, i, j)) {
          const double q = fmin2(A3(S, p, i, j, kk + 1), A3(S, p, i - 1, j, kk + 1));
          A3(S, dpu, i, j, kn) = .5 * ((fmin2(q, A3(S, p, i - 1, j, k + 1)) - fmin2(q, A3(S, p, i - 1, j, k))) +
                                       (fmin2(q, A3(S, p, i, j, k + 1)) - fmin2(q, A3(S, p, i, j, k))));
        }
        if (A2(S, iv, i, j)) {
          const double q = fmin2(A3(S, p, i, j, kk + 1), A3(S, p, i, j - 1, kk + 1));
          A3(S, dpv, i, j, kn) = .5 * ((fmin2(q, A3(S, p, i, j - 1, k + 1)) - fmin2(q, A3(S, p, i, j - 1, k))) +
                                       (fmin2(q, A3(S, p, i, j, k + 1)) - fmin2(q, A3(S, p, i, j, k))));
        }
      }
    }
}

/* :315-391: conservative remapping of u (isv = 0) or v (isv = 1) from the old (pu/pv) to the new
 * interface pressures at the velocity point */
static void remap_velocity(OState *S, int nn, int isv) {
  const int ii = S->ii, jj = S->jj, kk = S->kk;
  double *vel = isv ? S->v : S->u, *pvel = isv ? S->pv : S->pu;
  const int *mask = isv ? S->iv : S->iu;
  const int di = isv ? 0 : 1, dj = isv ? 1 : 0;
  for (int j = 1; j <= jj; j++)
    for (int i = 1; i <= ii; i++) {
      if (!mask[IX(S, i, j)]) continue;
      double uo[KMAXC], un[KMAXC], po[KMAXC + 1], pn[KMAXC + 1];
      const size_t c = IX(S, i, j), cm = IX(S, i - di, j - dj), np = S->nplane;
      for (int k = 1; k <= kk; k++) uo[k] = vel[c + np * (size_t)(k + nn - 1)];
      po[1] = 0.;
      pn[1] = 0.;
      const double pbot = pvel[c + np * (size_t)kk];
      for (int k = 2; k <= kk + 1; k++) {
        po[k] = pvel[c + np * (size_t)(k - 1)];
        pn[k] = .5 * (fmin2(pbot, S->p[c + np * (size_t)(k - 1)]) + fmin2(pbot, S->p[cm + np * (size_t)(k - 1)]));
      }
      int ko = 1;
      for (int kn = 1; kn <= kk; kn++) {
        if (pn[kn + 1] - pn[kn] == 0.) {
          un[kn] = 0.;
        } else {
          double udpn = 0.;
          while (pn[kn + 1] > po[ko + 1]) {
            udpn = udpn + uo[ko] * (po[ko + 1] - fmax2(po[ko], pn[kn]));
            ko = ko + 1;
          }
          un[kn] = (udpn + uo[ko] * (pn[kn + 1] - fmax2(po[ko], pn[kn]))) / (pn[kn + 1] - pn[kn]);
        }
      }
      for (int k = 1; k <= kk; k++) vel[c + np * (size_t)(k + nn - 1)] = un[k];
    }
}
