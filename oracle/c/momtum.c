/* TEST INFRASTRUCTURE (oracle): restatement of momtum, phy/mod_momtum.F90:215-1282
 * (isopyc_bulkml branch of the wind-stress term; mommth enscon/enecon/enedis). */
#include "ostate.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define L(a, i, j) (a)[IX(S, i, j)]
static inline double hfharm(double a, double b) { return a * b / (a + b); } /* :131-141 */

/* wet-segment tests over the full halo-extended row/column, phy/mod_bigrid.F90:320-429 */
static int seg_first_i(const OState *S, const int *msk, int i, int j) {
  return msk[IX(S, i, j)] && (i == 1 - NBDY || !msk[IX(S, i - 1, j)]);
}
static int seg_last_i(const OState *S, const int *msk, int i, int j) {
  return msk[IX(S, i, j)] && (i == S->ii + NBDY || !msk[IX(S, i + 1, j)]);
}
static int seg_first_j(const OState *S, const int *msk, int i, int j) {
  return msk[IX(S, i, j)] && (j == 1 - NBDY || !msk[IX(S, i, j - 1)]);
}
static int seg_last_j(const OState *S, const int *msk, int i, int j) {
  return msk[IX(S, i, j)] && (j == S->jj + NBDY || !msk[IX(S, i, j + 1)]);
}

void orc_momtum(OState *S, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)k1m; (void)k1n;
  const int ii = S->ii, jj = S->jj, kk = S->kk;
  const size_t lev = (size_t)S->nplane;
  const double c1 = 1. - 1.5 * .5, c2 = 1. - .5, c3 = 2., slope = .5; /* :220-221 */
  const double slip = -1., thkbot = 10.;                               /* :93-97 */
  const double cutoff = ONEM, thkbop = thkbot * ONEM;
  const double delt1 = S->delt1, dlt = S->dlt;
  const double tsfac = dlt / delt1, dt1inv = 1. / delt1;
  const double wpgf = .25, wuv1 = S->wuv1, wuv2 = S->wuv2; /* mod_pgforc.F90:47 */
  const int *ip = S->ip, *iu = S->iu, *iv = S->iv, *iq = S->iq;
  enum { NLOC = 45 };
  double *loc[NLOC];
  for (int x = 0; x < NLOC; x++) loc[x] = (double *)calloc(lev, sizeof(double));
  double *drag = loc[0], *ubrhs = loc[1], *vbrhs = loc[2], *stress = loc[3], *dpmx = loc[4], *vsc2 = loc[5],
         *vsc4 = loc[6], *potvor = loc[7], *vort = loc[8], *wgtia = loc[9], *wgtib = loc[10], *wgtja = loc[11],
         *wgtjb = loc[12], *dl2u = loc[13], *dl2uja = loc[14], *dl2ujb = loc[15], *dl2v = loc[16], *dl2via = loc[17],
         *dl2vib = loc[18], *ke = loc[19], *uh_min = loc[20], *uh_max = loc[21], *vh_min = loc[22], *vh_max = loc[23],
         *cau = loc[24], *cav = loc[25], *uflux1 = loc[26], *vflux1 = loc[27], *uja = loc[28], *ujb = loc[29],
         *via = loc[30], *vib = loc[31], *defor1 = loc[32], *defor2 = loc[33];
  double *utotm = S->utotm, *vtotm = S->vtotm, *utotn = S->utotn, *vtotn = S->vtotn, *uflux = S->uflux,
         *vflux = S->vflux, *uflux2 = S->uflux2, *vflux2 = S->vflux2, *uflux3 = S->uflux3, *vflux3 = S->vflux3;

  for (int j = -1; j <= jj + 2; j++) /* :245-254 */
    for (int k = 1; k <= kk; k++)
      for (int i = -1; i <= ii + 2; i++)
        if (L(ip, i, j)) A3(S, p, i, j, k + 1) = A3(S, p, i, j, k) + A3(S, dp, i, j, k + mm);

  for (int j = 0; j <= jj; j++) /* :260-292 bottom drag */
    for (int i = 0; i <= ii; i++) {
      if (!L(ip, i, j)) continue;
      double u1 = 0., u2 = 0.;
      for (int k = 1; k <= kk; k++) {
        int kn = k + nn;
        double pbotl = fmax2(A3(S, p, i, j, k + 1), A3(S, p, i, j, kk + 1) - thkbop);
        double ptopl = fmax2(A3(S, p, i, j, k), A3(S, p, i, j, kk + 1) - thkbop);
        u1 = u1 + (A3(S, u, i, j, kn) + A3(S, u, i + 1, j, kn)) * (pbotl - ptopl);
        u2 = u2 + (A3(S, v, i, j, kn) + A3(S, v, i, j + 1, kn)) * (pbotl - ptopl);
      }
      A2(S, util1, i, j) = u1;
      A2(S, util2, i, j) = u2;
      double ubot = (A3(S, ubflxs_p, i, j, n) / fmax2(EPSILPL, A3(S, pbu, i, j, n) * A2(S, scuy, i, j)) +
                     A3(S, ubflxs_p, i + 1, j, n) / fmax2(EPSILPL, A3(S, pbu, i + 1, j, n) * A2(S, scuy, i + 1, j))) * tsfac +
                    u1 / thkbop;
      double vbot = (A3(S, vbflxs_p, i, j, n) / fmax2(EPSILPL, A3(S, pbv, i, j, n) * A2(S, scvx, i, j)) +
                     A3(S, vbflxs_p, i, j + 1, n) / fmax2(EPSILPL, A3(S, pbv, i, j + 1, n) * A2(S, scvx, i, j + 1))) * tsfac +
                    u2 / thkbop;
      double ubbl = .5 * sqrt(ubot * ubot + vbot * vbot);
      double q = S->cb * (ubbl + S->cbar);
      L(drag, i, j) = q * GRAV / (ALPHA0 * thkbop);
      A2(S, ustarb, i, j) = sqrt(q * ubbl);
    }
  for (int j = 1; j <= jj; j++) /* :299-310 */
    for (int i = 1; i <= ii; i++) {
      if (L(iu, i, j)) L(ubrhs, i, j) = A2(S, ubcors_p, i, j) * tsfac;
      if (L(iv, i, j)) L(vbrhs, i, j) = A2(S, vbcors_p, i, j) * tsfac;
    }
  for (int j = 0; j <= jj + 1; j++) /* :314-319 */
    for (int i = 0; i <= ii + 1; i++) { L(dl2u, i, j) = 0.; L(dl2v, i, j) = 0.; }
  for (int k = 1; k <= kk; k++) /* :322-338 */
    for (int j = -1; j <= jj + 2; j++)
      for (int i = -1; i <= ii + 2; i++) {
        if (L(iu, i, j)) A3(S, pu, i, j, k + 1) = A3(S, pu, i, j, k) + A3(S, dpu, i, j, k + mm);
        if (L(iv, i, j)) A3(S, pv, i, j, k + 1) = A3(S, pv, i, j, k) + A3(S, dpv, i, j, k + mm);
      }
  orc_xctilr(S, S->difwgt, 1, 1, 2, 2, 1); /* :340 */

  for (int k = 1; k <= kk; k++) { /* :351-1145 */
    const int km = k + mm, kn = k + nn;
    for (int j = 0; j <= jj + 2; j++) /* :360-396 */
      for (int i = 0; i <= ii + 2; i++) L(dpmx, i, j) = 8. * cutoff;
    for (int j = 0; j <= jj + 2; j++)
      for (int i = 0; i <= ii + 2; i++)
        if (L(iu, i, j)) L(dpmx, i, j) = fmax2(L(dpmx, i, j), A3(S, dp, i, j, km) + A3(S, dp, i - 1, j, km));
    for (int j = -1; j <= jj + 1; j++)
      for (int i = 0; i <= ii + 2; i++)
        if (L(iu, i, j)) L(dpmx, i, j + 1) = fmax2(L(dpmx, i, j + 1), A3(S, dp, i, j, km) + A3(S, dp, i - 1, j, km));
    for (int j = 0; j <= jj + 2; j++)
      for (int i = 0; i <= ii + 2; i++)
        if (L(iv, i, j)) L(dpmx, i, j) = fmax2(L(dpmx, i, j), A3(S, dp, i, j, km) + A3(S, dp, i, j - 1, km));
    for (int j = 0; j <= jj + 2; j++)
      for (int i = -1; i <= ii + 1; i++)
        if (L(iv, i, j)) L(dpmx, i + 1, j) = fmax2(L(dpmx, i + 1, j), A3(S, dp, i, j, km) + A3(S, dp, i, j - 1, km));
    for (int j = 0; j <= jj + 1; j++) /* :398-406 */
      for (int i = 0; i <= ii + 1; i++)
        if (L(iu, i, j)) {
          L(utotm, i, j) = A3(S, u, i, j, km) + A3(S, ubflxs_p, i, j, m) * tsfac / (A3(S, pbu, i, j, m) * A2(S, scuy, i, j));
          L(uflux, i, j) = L(utotm, i, j) * fmax2(A3(S, dpu, i, j, km), cutoff);
        }
    for (int j = -1; j <= jj + 2; j++) /* :408-414 */
      for (int i = -1; i <= ii + 2; i++)
        if (L(iu, i, j))
          L(utotn, i, j) = A3(S, u, i, j, kn) + A3(S, ubflxs_p, i, j, n) * tsfac / (A3(S, pbu, i, j, n) * A2(S, scuy, i, j));
    for (int j = 0; j <= jj + 1; j++) /* :416-423 */
      for (int i = 0; i <= ii + 1; i++)
        if (L(iv, i, j)) {
          L(vtotm, i, j) = A3(S, v, i, j, km) + A3(S, vbflxs_p, i, j, m) * tsfac / (A3(S, pbv, i, j, m) * A2(S, scvx, i, j));
          L(vflux, i, j) = L(vtotm, i, j) * fmax2(A3(S, dpv, i, j, km), cutoff);
        }
    for (int j = -1; j <= jj + 2; j++) /* :425-431 */
      for (int i = -1; i <= ii + 2; i++)
        if (L(iv, i, j))
          L(vtotn, i, j) = A3(S, v, i, j, kn) + A3(S, vbflxs_p, i, j, n) * tsfac / (A3(S, pbv, i, j, n) * A2(S, scvx, i, j));
    for (int j = -1; j <= jj + 2; j++) /* :438-453 */
      for (int i = 0; i <= ii + 2; i++)
        if (L(iu, i, j)) {
          L(wgtja, i, j) = fmax2(0., fmin2(1., (A3(S, pu, i, j, k + 1) - A3(S, pbu, i, j - 1, m)) /
                                               fmax2(A3(S, pu, i, j, k + 1) - A3(S, pu, i, j, k), EPSILP)));
          L(wgtjb, i, j) = fmax2(0., fmin2(1., (A3(S, pu, i, j, k + 1) - A3(S, pbu, i, j + 1, m)) /
                                               fmax2(A3(S, pu, i, j, k + 1) - A3(S, pu, i, j, k), EPSILP)));
          L(uja, i, j) = (1. - L(wgtja, i, j)) * L(utotn, i, j - 1) + L(wgtja, i, j) * slip * L(utotn, i, j);
          L(ujb, i, j) = (1. - L(wgtjb, i, j)) * L(utotn, i, j + 1) + L(wgtjb, i, j) * slip * L(utotn, i, j);
          L(dl2u, i, j) = L(utotn, i, j) - .25 * (L(utotn, i + 1, j) + L(utotn, i - 1, j) + L(uja, i, j) + L(ujb, i, j));
        }
    for (int j = 0; j <= jj + 2; j++) /* :457-472 */
      for (int i = -1; i <= ii + 2; i++)
        if (L(iv, i, j)) {
          L(wgtia, i, j) = fmax2(0., fmin2(1., (A3(S, pv, i, j, k + 1) - A3(S, pbv, i - 1, j, m)) /
                                               fmax2(A3(S, pv, i, j, k + 1) - A3(S, pv, i, j, k), EPSILP)));
          L(wgtib, i, j) = fmax2(0., fmin2(1., (A3(S, pv, i, j, k + 1) - A3(S, pbv, i + 1, j, m)) /
                                               fmax2(A3(S, pv, i, j, k + 1) - A3(S, pv, i, j, k), EPSILP)));
          L(via, i, j) = (1. - L(wgtia, i, j)) * L(vtotn, i - 1, j) + L(wgtia, i, j) * slip * L(vtotn, i, j);
          L(vib, i, j) = (1. - L(wgtib, i, j)) * L(vtotn, i + 1, j) + L(wgtib, i, j) * slip * L(vtotn, i, j);
          L(dl2v, i, j) = L(vtotn, i, j) - .25 * (L(vtotn, i, j + 1) + L(vtotn, i, j - 1) + L(via, i, j) + L(vib, i, j));
        }
    /* :477-496 vorticity at lateral boundary points, v-segments */
    for (int j = 1; j <= jj + 1; j++)
      for (int i = 1 - NBDY; i <= ii + NBDY; i++) {
        if (seg_first_i(S, iv, i, j) && i >= 1 && i <= ii + 1) {
          L(vort, i, j) = L(vtotm, i, j) * (1. - slip) * A2(S, scvy, i, j) * A2(S, scq2i, i, j);
          A3(S, absvor, i, j, k) = L(vort, i, j) + A2(S, corioq, i, j);
          A3(S, dpvor, i, j, k) = .125 * fmax2(fmax2(4. * (A3(S, dp, i, j, km) + A3(S, dp, i, j - 1, km)), L(dpmx, i, j)), L(dpmx, i + 1, j));
          L(potvor, i, j) = A3(S, absvor, i, j, k) / A3(S, dpvor, i, j, k);
        }
        if (seg_last_i(S, iv, i, j) && i >= 0 && i <= ii) {
          L(vort, i + 1, j) = -L(vtotm, i, j) * (1. - slip) * A2(S, scvy, i, j) * A2(S, scq2i, i + 1, j);
          A3(S, absvor, i + 1, j, k) = L(vort, i + 1, j) + A2(S, corioq, i + 1, j);
          A3(S, dpvor, i + 1, j, k) = .125 * fmax2(fmax2(4. * (A3(S, dp, i, j, km) + A3(S, dp, i, j - 1, km)), L(dpmx, i, j)), L(dpmx, i + 1, j));
          L(potvor, i + 1, j) = A3(S, absvor, i + 1, j, k) / A3(S, dpvor, i + 1, j, k);
        }
      }
    for (int j = 0; j <= jj + 2; j++) /* :498-509 */
      for (int i = 1 - NBDY; i <= ii + NBDY; i++) {
        if (seg_first_i(S, iv, i, j) && i >= 0) {
          double t = L(vtotn, i, j) * (1. - slip) * A2(S, scvy, i, j);
          L(defor2, i, j) = t * t * A2(S, scq2i, i, j);
        }
        if (seg_last_i(S, iv, i, j) && i < ii + 2) {
          double t = L(vtotn, i, j) * (1. - slip) * A2(S, scvy, i, j);
          L(defor2, i + 1, j) = t * t * A2(S, scq2i, i + 1, j);
        }
      }
    for (int i = 1; i <= ii + 1; i++) /* :511-530 u-segments (in j) */
      for (int j = 1 - NBDY; j <= jj + NBDY; j++) {
        if (seg_first_j(S, iu, i, j) && j >= 1 && j <= jj + 1) {
          L(vort, i, j) = -L(utotm, i, j) * (1. - slip) * A2(S, scux, i, j) * A2(S, scq2i, i, j);
          A3(S, absvor, i, j, k) = L(vort, i, j) + A2(S, corioq, i, j);
          A3(S, dpvor, i, j, k) = .125 * fmax2(fmax2(4. * (A3(S, dp, i, j, km) + A3(S, dp, i - 1, j, km)), L(dpmx, i, j)), L(dpmx, i, j + 1));
          L(potvor, i, j) = A3(S, absvor, i, j, k) / A3(S, dpvor, i, j, k);
        }
        if (seg_last_j(S, iu, i, j) && j >= 0 && j <= jj) {
          L(vort, i, j + 1) = L(utotm, i, j) * (1. - slip) * A2(S, scux, i, j) * A2(S, scq2i, i, j + 1);
          A3(S, absvor, i, j + 1, k) = L(vort, i, j + 1) + A2(S, corioq, i, j + 1);
          A3(S, dpvor, i, j + 1, k) = .125 * fmax2(fmax2(4. * (A3(S, dp, i, j, km) + A3(S, dp, i - 1, j, km)), L(dpmx, i, j)), L(dpmx, i, j + 1));
          L(potvor, i, j + 1) = A3(S, absvor, i, j + 1, k) / A3(S, dpvor, i, j + 1, k);
        }
      }
    for (int i = 0; i <= ii + 2; i++) /* :532-543 */
      for (int j = 1 - NBDY; j <= jj + NBDY; j++) {
        if (seg_first_j(S, iu, i, j) && j >= 0) {
          double t = L(utotn, i, j) * (1. - slip) * A2(S, scux, i, j);
          L(defor2, i, j) = t * t * A2(S, scq2i, i, j);
        }
        if (seg_last_j(S, iu, i, j) && j < jj + 2) {
          double t = L(utotn, i, j) * (1. - slip) * A2(S, scux, i, j);
          L(defor2, i, j + 1) = t * t * A2(S, scq2i, i, j + 1);
        }
      }
    for (int j = -1; j <= jj + 1; j++) /* :549-559 */
      for (int i = -1; i <= ii + 1; i++)
        if (L(ip, i, j)) {
          double t = (L(utotn, i + 1, j) * A2(S, scuy, i + 1, j) - L(utotn, i, j) * A2(S, scuy, i, j)) -
                     (L(vtotn, i, j + 1) * A2(S, scvx, i, j + 1) - L(vtotn, i, j) * A2(S, scvx, i, j));
          L(defor1, i, j) = t * t * A2(S, scp2i, i, j);
        }
    for (int j = 1; j <= jj + 1; j++) /* :561-575 */
      for (int i = 1; i <= ii + 1; i++)
        if (L(iq, i, j)) {
          L(vort, i, j) = (L(vtotm, i, j) * A2(S, scvy, i, j) - L(vtotm, i - 1, j) * A2(S, scvy, i - 1, j) -
                           L(utotm, i, j) * A2(S, scux, i, j) + L(utotm, i, j - 1) * A2(S, scux, i, j - 1)) * A2(S, scq2i, i, j);
          A3(S, absvor, i, j, k) = L(vort, i, j) + A2(S, corioq, i, j);
          double d = fmax2(2. * (A3(S, dp, i, j, km) + A3(S, dp, i - 1, j, km) + A3(S, dp, i, j - 1, km) + A3(S, dp, i - 1, j - 1, km)),
                           L(dpmx, i, j));
          d = fmax2(d, L(dpmx, i - 1, j));
          d = fmax2(d, L(dpmx, i + 1, j));
          d = fmax2(d, L(dpmx, i, j - 1));
          d = fmax2(d, L(dpmx, i, j + 1));
          A3(S, dpvor, i, j, k) = .125 * d;
          L(potvor, i, j) = A3(S, absvor, i, j, k) / A3(S, dpvor, i, j, k);
        }
    for (int j = 0; j <= jj + 2; j++) /* :577-585 */
      for (int i = 0; i <= ii + 2; i++)
        if (L(iq, i, j)) {
          double t = L(vib, i - 1, j) * A2(S, scvy, i, j) - L(via, i, j) * A2(S, scvy, i - 1, j) +
                     L(ujb, i, j - 1) * A2(S, scux, i, j) - L(uja, i, j) * A2(S, scux, i, j - 1);
          L(defor2, i, j) = t * t * A2(S, scq2i, i, j);
        }
    for (int j = 1; j <= jj; j++) /* :591-608 */
      for (int i = 1; i <= ii; i++) {
        if (L(iu, i, j)) {
          L(dl2uja, i, j) = (1. - L(wgtja, i, j)) * L(dl2u, i, j - 1) + L(wgtja, i, j) * slip * L(dl2u, i, j);
          L(dl2ujb, i, j) = (1. - L(wgtjb, i, j)) * L(dl2u, i, j + 1) + L(wgtjb, i, j) * slip * L(dl2u, i, j);
        }
        if (L(iv, i, j)) {
          L(dl2via, i, j) = (1. - L(wgtia, i, j)) * L(dl2v, i - 1, j) + L(wgtia, i, j) * slip * L(dl2v, i, j);
          L(dl2vib, i, j) = (1. - L(wgtib, i, j)) * L(dl2v, i + 1, j) + L(wgtib, i, j) * slip * L(dl2v, i, j);
        }
      }
    for (int j = 0; j <= jj; j++) /* :613-662 */
      for (int i = 0; i <= ii; i++)
        if (L(ip, i, j))
          L(ke, i, j) = .25 * (A2(S, scu2, i, j) * (L(utotm, i, j) * L(utotm, i, j)) +
                               A2(S, scu2, i + 1, j) * (L(utotm, i + 1, j) * L(utotm, i + 1, j)) +
                               A2(S, scv2, i, j) * (L(vtotm, i, j) * L(vtotm, i, j)) +
                               A2(S, scv2, i, j + 1) * (L(vtotm, i, j + 1) * L(vtotm, i, j + 1))) / A2(S, scp2, i, j);
    if (S->mommth == 2) { /* :664-719 */
      for (int j = 0; j <= jj + 1; j++)
        for (int i = 0; i <= ii + 1; i++) {
          if (L(iu, i, j)) {
            double uhc = .5 * L(utotm, i, j) * (A3(S, dp, i, j, km) + A3(S, dp, i - 1, j, km));
            double uhm = L(uflux, i, j);
            if (fabs(uhc) < .1 * fabs(uhm)) uhm = 10. * uhc;
            else if (fabs(uhc) > c1 * fabs(uhm)) {
              if (fabs(uhc) < c2 * fabs(uhm)) uhc = (3. * uhc + (1. - c2 * 3.) * uhm);
              else if (fabs(uhc) <= c3 * fabs(uhm)) uhc = uhm;
              else uhc = slope * uhc + (1. - c3 * slope) * uhm;
            }
            if (uhc > uhm) { L(uh_min, i, j) = uhm; L(uh_max, i, j) = uhc; }
            else { L(uh_max, i, j) = uhm; L(uh_min, i, j) = uhc; }
          }
          if (L(iv, i, j)) {
            double vhc = .5 * L(vtotm, i, j) * (A3(S, dp, i, j, km) + A3(S, dp, i, j - 1, km));
            double vhm = L(vflux, i, j);
            if (fabs(vhc) < .1 * fabs(vhm)) vhm = 10. * vhc;
            else if (fabs(vhc) > c1 * fabs(vhm)) {
              if (fabs(vhc) < c2 * fabs(vhm)) vhc = (3. * vhc + (1. - c2 * 3.) * vhm);
              else if (fabs(vhc) <= c3 * fabs(vhm)) vhc = vhm;
              else vhc = slope * vhc + (1. - c3 * slope) * vhm;
            }
            if (vhc > vhm) { L(vh_min, i, j) = vhm; L(vh_max, i, j) = vhc; }
            else { L(vh_max, i, j) = vhm; L(vh_min, i, j) = vhc; }
          }
        }
    }
    for (int j = 1; j <= jj; j++) /* :723-813 Coriolis / advection terms */
      for (int i = 1; i <= ii; i++) {
        if (L(iu, i, j)) {
          if (S->mommth == 0)
            L(cau, i, j) = .125 * (L(vflux, i, j) + L(vflux, i, j + 1) + L(vflux, i - 1, j) + L(vflux, i - 1, j + 1)) *
                           (L(potvor, i, j) + L(potvor, i, j + 1));
          else if (S->mommth == 1)
            L(cau, i, j) = .25 * ((L(vflux, i, j) + L(vflux, i - 1, j)) * L(potvor, i, j) +
                                  (L(vflux, i, j + 1) + L(vflux, i - 1, j + 1)) * L(potvor, i, j + 1));
          else {
            double t1, t2;
            if (L(potvor, i, j + 1) * L(utotm, i, j) == 0.)
              t1 = L(potvor, i, j + 1) * ((L(vh_max, i, j + 1) + L(vh_max, i - 1, j + 1)) + (L(vh_min, i, j + 1) + L(vh_min, i - 1, j + 1))) * .5;
            else if (L(potvor, i, j + 1) * L(utotm, i, j) < 0.) t1 = L(potvor, i, j + 1) * (L(vh_max, i, j + 1) + L(vh_max, i - 1, j + 1));
            else t1 = L(potvor, i, j + 1) * (L(vh_min, i, j + 1) + L(vh_min, i - 1, j + 1));
            if (L(potvor, i, j) * L(utotm, i, j) == 0.)
              t2 = L(potvor, i, j) * ((L(vh_max, i, j) + L(vh_max, i - 1, j)) + (L(vh_min, i, j) + L(vh_min, i - 1, j))) * .5;
            else if (L(potvor, i, j) * L(utotm, i, j) < 0.) t2 = L(potvor, i, j) * (L(vh_max, i, j) + L(vh_max, i - 1, j));
            else t2 = L(potvor, i, j) * (L(vh_min, i, j) + L(vh_min, i - 1, j));
            L(cau, i, j) = .25 * (t1 + t2);
          }
        }
        if (L(iv, i, j)) {
          if (S->mommth == 0)
            L(cav, i, j) = -.125 * (L(uflux, i, j) + L(uflux, i + 1, j) + L(uflux, i, j - 1) + L(uflux, i + 1, j - 1)) *
                           (L(potvor, i, j) + L(potvor, i + 1, j));
          else if (S->mommth == 1)
            L(cav, i, j) = -.25 * ((L(uflux, i, j) + L(uflux, i, j - 1)) * L(potvor, i, j) +
                                   (L(uflux, i + 1, j) + L(uflux, i + 1, j - 1)) * L(potvor, i + 1, j));
          else {
            double t1, t2;
            if (L(potvor, i + 1, j) * L(vtotm, i, j) == 0.)
              t1 = L(potvor, i + 1, j) * ((L(uh_max, i + 1, j) + L(uh_max, i + 1, j - 1)) + (L(uh_min, i + 1, j) + L(uh_min, i + 1, j - 1))) * .5;
            else if (L(potvor, i + 1, j) * L(vtotm, i, j) > 0.) t1 = L(potvor, i + 1, j) * (L(uh_max, i + 1, j) + L(uh_max, i + 1, j - 1));
            else t1 = L(potvor, i + 1, j) * (L(uh_min, i + 1, j) + L(uh_min, i + 1, j - 1));
            if (L(potvor, i, j) * L(vtotm, i, j) == 0.)
              t2 = L(potvor, i, j) * ((L(uh_max, i, j) + L(uh_max, i, j - 1)) + (L(uh_min, i, j) + L(uh_min, i, j - 1))) * .5;
            else if (L(potvor, i, j) * L(vtotm, i, j) > 0.) t2 = L(potvor, i, j) * (L(uh_max, i, j) + L(uh_max, i, j - 1));
            else t2 = L(potvor, i, j) * (L(uh_min, i, j) + L(uh_min, i, j - 1));
            L(cav, i, j) = -.25 * (t1 + t2);
          }
        }
      }
    /* ---------- u equation ---------- */
    for (int j = 0; j <= jj + 1; j++) /* :829-841 */
      for (int i = 0; i <= ii + 1; i++)
        if (L(iu, i, j)) {
          double q = .5 * (A2(S, difwgt, i - 1, j) + A2(S, difwgt, i, j));
          double deform = sqrt(.5 * (L(defor1, i, j) + L(defor1, i - 1, j) + L(defor2, i, j) + L(defor2, i, j + 1)));
          L(vsc2, i, j) = fmax2(q * S->mdv2hi + (1. - q) * S->mdv2lo, (q * S->vsc2hi + (1. - q) * S->vsc2lo) * deform);
          L(vsc4, i, j) = fmax2(q * S->mdv4hi + (1. - q) * S->mdv4lo, (q * S->vsc4hi + (1. - q) * S->vsc4lo) * deform);
        }
    for (int j = 1; j <= jj; j++) { /* :843-915 */
      for (int i = 1 - NBDY; i <= ii + NBDY; i++) {
        if (seg_first_i(S, iu, i, j) && i > 0) { L(vsc2, i - 1, j) = L(vsc2, i, j); L(vsc4, i - 1, j) = L(vsc4, i, j); }
        if (seg_last_i(S, iu, i, j) && i < ii + 1) { L(vsc2, i + 1, j) = L(vsc2, i, j); L(vsc4, i + 1, j) = L(vsc4, i, j); }
      }
      for (int i = 0; i <= ii; i++)
        if (L(ip, i, j) && L(iu, i, j) + L(iu, i + 1, j) > 0) {
          double dpxy = fmax2(A3(S, dpu, i, j, km), ONEMM), dpib = fmax2(A3(S, dpu, i + 1, j, km), ONEMM);
          L(uflux1, i, j) = fmin2(A2(S, difmxp, i, j), (L(vsc2, i, j) + L(vsc2, i + 1, j)) * A2(S, scpy, i, j)) *
                                hfharm(dpxy, dpib) * (L(utotn, i, j) - L(utotn, i + 1, j)) +
                            fmin2(.125 * A2(S, difmxp, i, j), (L(vsc4, i, j) + L(vsc4, i + 1, j)) * A2(S, scpy, i, j)) *
                                hfharm(dpxy, dpib) * (L(dl2u, i, j) - L(dl2u, i + 1, j));
        }
      for (int i = 1; i <= ii; i++)
        if (L(iu, i, j)) {
          double dpxy = fmax2(A3(S, dpu, i, j, km), ONEMM);
          double dpja = fmax2(A3(S, dpu, i, j - 1, km), ONEMM);
          dpja = dpja + L(wgtja, i, j) * (dpxy - dpja);
          double dpjb = fmax2(A3(S, dpu, i, j + 1, km), ONEMM);
          dpjb = dpjb + L(wgtjb, i, j) * (dpxy - dpjb);
          double vsc2a, vsc4a, vsc2b, vsc4b;
          if (L(iu, i, j - 1) == 0) { vsc2a = L(vsc2, i, j); vsc4a = L(vsc4, i, j); }
          else { vsc2a = L(vsc2, i, j - 1); vsc4a = L(vsc4, i, j - 1); }
          if (L(iu, i, j + 1) == 0) { vsc2b = L(vsc2, i, j); vsc4b = L(vsc4, i, j); }
          else { vsc2b = L(vsc2, i, j + 1); vsc4b = L(vsc4, i, j + 1); }
          L(uflux2, i, j) = fmin2(A2(S, difmxq, i, j), (L(vsc2, i, j) + vsc2a) * A2(S, scqx, i, j)) * hfharm(dpja, dpxy) *
                                (L(uja, i, j) - L(utotn, i, j)) +
                            fmin2(.125 * A2(S, difmxq, i, j), (L(vsc4, i, j) + vsc4a) * A2(S, scqx, i, j)) * hfharm(dpja, dpxy) *
                                (L(dl2uja, i, j) - L(dl2u, i, j));
          L(uflux3, i, j) = fmin2(A2(S, difmxq, i, j + 1), (L(vsc2, i, j) + vsc2b) * A2(S, scqx, i, j + 1)) * hfharm(dpjb, dpxy) *
                                (L(utotn, i, j) - L(ujb, i, j)) +
                            fmin2(.125 * A2(S, difmxq, i, j + 1), (L(vsc4, i, j) + vsc4b) * A2(S, scqx, i, j + 1)) * hfharm(dpjb, dpxy) *
                                (L(dl2u, i, j) - L(dl2ujb, i, j));
        }
    }
    for (int j = 1; j <= jj; j++) /* :919-980 */
      for (int i = 1; i <= ii; i++)
        if (L(iu, i, j)) {
          if (k == 1) L(stress, i, j) = -2. * A2(S, taux, i, j) * GRAV * A2(S, scux, i, j) / (A3(S, p, i, j, 2) + A3(S, p, i - 1, j, 2));
          else L(stress, i, j) = 0.;
          double ptopl = .5 * (fmin2(A3(S, pbu, i, j, m), A3(S, p, i, j, k)) + fmin2(A3(S, pbu, i, j, m), A3(S, p, i - 1, j, k)));
          double pbotl = .5 * (fmin2(A3(S, pbu, i, j, m), A3(S, p, i, j, k + 1)) + fmin2(A3(S, pbu, i, j, m), A3(S, p, i - 1, j, k + 1)));
          double q = .5 * (L(drag, i, j) + L(drag, i - 1, j)) *
                     (fmax2(A3(S, pbu, i, j, m) - thkbop, pbotl) - fmax2(A3(S, pbu, i, j, m) - thkbop, fmin2(ptopl, pbotl - ONEMM))) /
                     fmax2(A3(S, dpu, i, j, km), ONEMM);
          double botstr = -L(utotn, i, j) * q / (1. + delt1 * q);
          double pgf = (1. - 2. * wpgf) * A3(S, pgfx, i, j, km) + wpgf * (A3(S, pgfx_o, i, j, k) + A3(S, pgfx, i, j, kn));
          A3(S, u, i, j, km) = A3(S, u, i, j, km) * (wuv1 * A3(S, dpu, i, j, km) + ONEMM) + A3(S, u, i, j, kn) * wuv2 * A3(S, dpuold, i, j, k);
          A3(S, u, i, j, kn) = A3(S, u, i, j, kn) +
                               delt1 * (-A2(S, scuxi, i, j) * (-pgf + L(stress, i, j) + (L(ke, i, j) - L(ke, i - 1, j))) + L(cau, i, j) -
                                        L(ubrhs, i, j) + botstr -
                                        (L(uflux1, i, j) - L(uflux1, i - 1, j) + L(uflux3, i, j) - L(uflux2, i, j)) /
                                            (A2(S, scu2, i, j) * fmax2(A3(S, dpu, i, j, km), ONEMM)));
        }
    /* ---------- v equation ---------- */
    for (int j = 0; j <= jj + 1; j++) /* :988-1000 */
      for (int i = 0; i <= ii + 1; i++)
        if (L(iv, i, j)) {
          double q = .5 * (A2(S, difwgt, i, j - 1) + A2(S, difwgt, i, j));
          double deform = sqrt(.5 * (L(defor1, i, j) + L(defor1, i, j - 1) + L(defor2, i, j) + L(defor2, i + 1, j)));
          L(vsc2, i, j) = fmax2(q * S->mdv2hi + (1. - q) * S->mdv2lo, (q * S->vsc2hi + (1. - q) * S->vsc2lo) * deform);
          L(vsc4, i, j) = fmax2(q * S->mdv4hi + (1. - q) * S->mdv4lo, (q * S->vsc4hi + (1. - q) * S->vsc4lo) * deform);
        }
    for (int i = 0; i <= ii + 1; i++) /* :1002-1015 */
      for (int j = 1 - NBDY; j <= jj + NBDY; j++) {
        if (seg_first_j(S, iv, i, j) && j > 0) { L(vsc2, i, j - 1) = L(vsc2, i, j); L(vsc4, i, j - 1) = L(vsc4, i, j); }
        if (seg_last_j(S, iv, i, j) && j < jj + 1) { L(vsc2, i, j + 1) = L(vsc2, i, j); L(vsc4, i, j + 1) = L(vsc4, i, j); }
      }
    for (int j = 0; j <= jj; j++) /* :1019-1034 */
      for (int i = 1; i <= ii; i++)
        if (L(ip, i, j) && L(iv, i, j) + L(iv, i, j + 1) > 0) {
          double dpxy = fmax2(A3(S, dpv, i, j, km), ONEMM), dpjb = fmax2(A3(S, dpv, i, j + 1, km), ONEMM);
          L(vflux1, i, j) = fmin2(A2(S, difmxp, i, j), (L(vsc2, i, j) + L(vsc2, i, j + 1)) * A2(S, scpx, i, j)) *
                                hfharm(dpxy, dpjb) * (L(vtotn, i, j) - L(vtotn, i, j + 1)) +
                            fmin2(.125 * A2(S, difmxp, i, j), (L(vsc4, i, j) + L(vsc4, i, j + 1)) * A2(S, scpx, i, j)) *
                                hfharm(dpxy, dpjb) * (L(dl2v, i, j) - L(dl2v, i, j + 1));
        }
    for (int j = 1; j <= jj; j++) /* :1040-1078 */
      for (int i = 1; i <= ii; i++)
        if (L(iv, i, j)) {
          double dpxy = fmax2(A3(S, dpv, i, j, km), ONEMM);
          double dpia = fmax2(A3(S, dpv, i - 1, j, km), ONEMM);
          dpia = dpia + L(wgtia, i, j) * (dpxy - dpia);
          double dpib = fmax2(A3(S, dpv, i + 1, j, km), ONEMM);
          dpib = dpib + L(wgtib, i, j) * (dpxy - dpib);
          double vsc2a, vsc4a, vsc2b, vsc4b;
          if (L(iv, i - 1, j) == 0) { vsc2a = L(vsc2, i, j); vsc4a = L(vsc4, i, j); }
          else { vsc2a = L(vsc2, i - 1, j); vsc4a = L(vsc4, i - 1, j); }
          if (L(iv, i + 1, j) == 0) { vsc2b = L(vsc2, i, j); vsc4b = L(vsc4, i, j); }
          else { vsc2b = L(vsc2, i + 1, j); vsc4b = L(vsc4, i + 1, j); }
          L(vflux2, i, j) = fmin2(A2(S, difmxq, i, j), (L(vsc2, i, j) + vsc2a) * A2(S, scqy, i, j)) * hfharm(dpia, dpxy) *
                                (L(via, i, j) - L(vtotn, i, j)) +
                            fmin2(.125 * A2(S, difmxq, i, j), (L(vsc4, i, j) + vsc4a) * A2(S, scqy, i, j)) * hfharm(dpia, dpxy) *
                                (L(dl2via, i, j) - L(dl2v, i, j));
          L(vflux3, i, j) = fmin2(A2(S, difmxq, i + 1, j), (L(vsc2, i, j) + vsc2b) * A2(S, scqy, i + 1, j)) * hfharm(dpib, dpxy) *
                                (L(vtotn, i, j) - L(vib, i, j)) +
                            fmin2(.125 * A2(S, difmxq, i + 1, j), (L(vsc4, i, j) + vsc4b) * A2(S, scqy, i + 1, j)) * hfharm(dpib, dpxy) *
                                (L(dl2v, i, j) - L(dl2vib, i, j));
        }
    for (int j = 1; j <= jj; j++) /* :1082-1143 */
      for (int i = 1; i <= ii; i++)
        if (L(iv, i, j)) {
          if (k == 1) L(stress, i, j) = -2. * A2(S, tauy, i, j) * GRAV * A2(S, scvy, i, j) / (A3(S, p, i, j, 2) + A3(S, p, i, j - 1, 2));
          else L(stress, i, j) = 0.;
          double ptopl = .5 * (fmin2(A3(S, pbv, i, j, m), A3(S, p, i, j, k)) + fmin2(A3(S, pbv, i, j, m), A3(S, p, i, j - 1, k)));
          double pbotl = .5 * (fmin2(A3(S, pbv, i, j, m), A3(S, p, i, j, k + 1)) + fmin2(A3(S, pbv, i, j, m), A3(S, p, i, j - 1, k + 1)));
          double q = .5 * (L(drag, i, j) + L(drag, i, j - 1)) *
                     (fmax2(A3(S, pbv, i, j, m) - thkbop, pbotl) - fmax2(A3(S, pbv, i, j, m) - thkbop, fmin2(ptopl, pbotl - ONEMM))) /
                     fmax2(A3(S, dpv, i, j, km), ONEMM);
          double botstr = -L(vtotn, i, j) * q / (1. + delt1 * q);
          double pgf = (1. - 2. * wpgf) * A3(S, pgfy, i, j, km) + wpgf * (A3(S, pgfy_o, i, j, k) + A3(S, pgfy, i, j, kn));
          A3(S, v, i, j, km) = A3(S, v, i, j, km) * (wuv1 * A3(S, dpv, i, j, km) + ONEMM) + A3(S, v, i, j, kn) * wuv2 * A3(S, dpvold, i, j, k);
          A3(S, v, i, j, kn) = A3(S, v, i, j, kn) +
                               delt1 * (-A2(S, scvyi, i, j) * (-pgf + L(stress, i, j) + (L(ke, i, j) - L(ke, i, j - 1))) + L(cav, i, j) -
                                        L(vbrhs, i, j) + botstr -
                                        (L(vflux1, i, j) - L(vflux1, i, j - 1) + L(vflux3, i, j) - L(vflux2, i, j)) /
                                            (A2(S, scv2, i, j) * fmax2(A3(S, dpv, i, j, km), ONEMM)));
        }
  }
  /* :1154-1197 massless-layer fill, barotropic extraction */
  for (int j = 1; j <= jj; j++)
    for (int i = 1; i <= ii; i++) {
      if (L(iu, i, j)) {
        double t = 0.;
        for (int k = 1; k <= kk; k++) {
          int km = k + mm, kn = k + nn, kan = imax2(1, k - 1) + nn;
          double q = fmin2(fmin2(A3(S, dpu, i, j, km), A3(S, dpu, i, j, kn)), ONEM);
          A3(S, u, i, j, kn) = (A3(S, u, i, j, kn) * q + A3(S, u, i, j, kan) * (ONEM - q)) / ONEM;
          A3(S, u, i, j, kn) = fmax2(-A2(S, umax, i, j), fmin2(A2(S, umax, i, j), A3(S, u, i, j, kn) + A3(S, ub, i, j, m))) - A3(S, ub, i, j, m);
          t = t + A3(S, u, i, j, kn) * A3(S, dpu, i, j, kn);
        }
        L(utotn, i, j) = t / A2(S, pbu_p, i, j);
      }
      if (L(iv, i, j)) {
        double t = 0.;
        for (int k = 1; k <= kk; k++) {
          int km = k + mm, kn = k + nn, kan = imax2(1, k - 1) + nn;
          double q = fmin2(fmin2(A3(S, dpv, i, j, km), A3(S, dpv, i, j, kn)), ONEM);
          A3(S, v, i, j, kn) = (A3(S, v, i, j, kn) * q + A3(S, v, i, j, kan) * (ONEM - q)) / ONEM;
          A3(S, v, i, j, kn) = fmax2(-A2(S, vmax, i, j), fmin2(A2(S, vmax, i, j), A3(S, v, i, j, kn) + A3(S, vb, i, j, m))) - A3(S, vb, i, j, m);
          t = t + A3(S, v, i, j, kn) * A3(S, dpv, i, j, kn);
        }
        L(vtotn, i, j) = t / A2(S, pbv_p, i, j);
      }
    }
  for (int k = 1; k <= kk; k++) { /* :1202-1232 time smoothing part 2 */
    int km = k + mm, kn = k + nn;
    for (int j = 1; j <= jj; j++)
      for (int i = 1; i <= ii; i++) {
        if (L(iu, i, j)) {
          A3(S, u, i, j, kn) = A3(S, u, i, j, kn) - L(utotn, i, j);
          A3(S, u, i, j, km) = (A3(S, u, i, j, km) + A3(S, u, i, j, kn) * wuv2 * A3(S, dpu, i, j, kn)) /
                               (wuv1 * A3(S, dpu, i, j, km) + ONEMM + wuv2 * (A3(S, dpuold, i, j, k) + A3(S, dpu, i, j, kn)));
        }
        if (L(iv, i, j)) {
          A3(S, v, i, j, kn) = A3(S, v, i, j, kn) - L(vtotn, i, j);
          A3(S, v, i, j, km) = (A3(S, v, i, j, km) + A3(S, v, i, j, kn) * wuv2 * A3(S, dpv, i, j, kn)) /
                               (wuv1 * A3(S, dpv, i, j, km) + ONEMM + wuv2 * (A3(S, dpvold, i, j, k) + A3(S, dpv, i, j, kn)));
        }
      }
  }
  for (int j = 1; j <= jj; j++) /* :1235-1267 */
    for (int i = 1; i <= ii; i++) {
      if (L(iu, i, j)) L(utotn, i, j) = L(utotn, i, j) * dt1inv;
      if (L(iv, i, j)) L(vtotn, i, j) = L(vtotn, i, j) * dt1inv;
    }
  for (int j = 1; j <= jj; j++)
    for (int k = 1; k <= kk; k++)
      for (int i = 1; i <= ii; i++) {
        if (L(iu, i, j)) A3(S, pu, i, j, k + 1) = A3(S, pu, i, j, k) + A3(S, dpu, i, j, k + nn);
        if (L(iv, i, j)) A3(S, pv, i, j, k + 1) = A3(S, pv, i, j, k) + A3(S, dpv, i, j, k + nn);
      }
  for (int x = 0; x < NLOC; x++) free(loc[x]);
}
