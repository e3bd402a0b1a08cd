/* TEST INFRASTRUCTURE (oracle): restatement of barotp, phy/mod_barotp.F90:148-1003
 * (split-explicit forward-backward barotropic subcycling). */
#include "ostate.h"
#include <math.h>
#include <stdlib.h>

/* continuity, :401-411 (odd, j,i = -1..+2/+1) and :626-636 (even, 0..+1/ii) */
static void cont(OState *S, int ml, int nl, int j0, int j1, int i0, int i1) {
  const double wbaro = S->wbaro, dlt = S->dlt;
  for (int j = j0; j <= j1; j++)
    for (int i = i0; i <= i1; i++)
      if (A2(S, ip, i, j))
        A3(S, pb_t, i, j, nl) = (1. - wbaro) * A3(S, pb_t, i, j, ml) + wbaro * A3(S, pb_t, i, j, nl) -
                                (1. + wbaro) * dlt *
                                    (A3(S, ubflx_t, i + 1, j, ml) - A3(S, ubflx_t, i, j, ml) +
                                     A3(S, vbflx_t, i, j + 1, ml) - A3(S, vbflx_t, i, j, ml)) *
                                    A2(S, scp2i, i, j);
}

/* u momentum, :420-457 (odd: lv = ml) and :745-781 (even: lv = nl); enscon and enecon/enedis */
static void umom(OState *S, int m, int n, int ml, int nl, int lv, double wo, double wm, double wn, int j0, int j1,
                 int i0, int i1) {
  const double wbaro = S->wbaro, dlt = S->dlt;
  for (int j = j0; j <= j1; j++)
    for (int i = i0; i <= i1; i++) {
      if (!A2(S, iu, i, j)) continue;
      A2(S, ubflxs_t, i, j) = A2(S, ubflxs_t, i, j) - wbaro * A3(S, ubflx_t, i, j, nl) +
                              (1. + wbaro) * A3(S, ubflx_t, i, j, ml);
      double q;
      if (S->mommth == 0)
        q = (A3(S, vbflx_t, i, j, lv) * A2(S, scvxi, i, j) + A3(S, vbflx_t, i, j + 1, lv) * A2(S, scvxi, i, j + 1) +
             A3(S, vbflx_t, i - 1, j, lv) * A2(S, scvxi, i - 1, j) +
             A3(S, vbflx_t, i - 1, j + 1, lv) * A2(S, scvxi, i - 1, j + 1)) *
            (wo * (A2(S, pvtrop_o, i, j) + A2(S, pvtrop_o, i, j + 1)) +
             wm * (A3(S, pvtrop, i, j, m) + A3(S, pvtrop, i, j + 1, m)) +
             wn * (A3(S, pvtrop, i, j, n) + A3(S, pvtrop, i, j + 1, n))) * .125;
      else
        q = .25 * ((A3(S, vbflx_t, i, j, lv) * A2(S, scvxi, i, j) + A3(S, vbflx_t, i - 1, j, lv) * A2(S, scvxi, i - 1, j)) *
                       (wo * A2(S, pvtrop_o, i, j) + wm * A3(S, pvtrop, i, j, m) + wn * A3(S, pvtrop, i, j, n)) +
                   (A3(S, vbflx_t, i, j + 1, lv) * A2(S, scvxi, i, j + 1) +
                    A3(S, vbflx_t, i - 1, j + 1, lv) * A2(S, scvxi, i - 1, j + 1)) *
                       (wo * A2(S, pvtrop_o, i, j + 1) + wm * A3(S, pvtrop, i, j + 1, m) + wn * A3(S, pvtrop, i, j + 1, n)));
      A2(S, ubcors_t, i, j) = A2(S, ubcors_t, i, j) + q;
      double utndcy =
          q + (wo * (A2(S, pgfxm_o, i, j) - (A2(S, xixp_o, i, j) * A3(S, pb_t, i, j, nl) - A2(S, xixm_o, i, j) * A3(S, pb_t, i - 1, j, nl))) +
               wm * (A3(S, pgfxm, i, j, m) - (A3(S, xixp, i, j, m) * A3(S, pb_t, i, j, nl) - A3(S, xixm, i, j, m) * A3(S, pb_t, i - 1, j, nl))) +
               wn * (A3(S, pgfxm, i, j, n) - (A3(S, xixp, i, j, n) * A3(S, pb_t, i, j, nl) - A3(S, xixm, i, j, n) * A3(S, pb_t, i - 1, j, nl)))) *
                  A2(S, scuxi, i, j);
      double x = (1. - wbaro) * A3(S, ubflx_t, i, j, ml) + wbaro * A3(S, ubflx_t, i, j, nl) +
                 (1. + wbaro) * dlt *
                     ((utndcy + A2(S, utotn, i, j)) * A2(S, scuy, i, j) * fmin2(A3(S, pb_t, i - 1, j, nl), A3(S, pb_t, i, j, nl)) -
                      A2(S, uglue, i, j) * A3(S, ubflx_t, i, j, ml));
      A3(S, ubflx_t, i, j, nl) = fmax2(-A2(S, uminb, i, j), fmin2(A2(S, umaxb, i, j), x));
    }
}

/* v momentum, :520-557 (odd: lu = nl) and :646-682 (even: lu = ml) */
static void vmom(OState *S, int m, int n, int ml, int nl, int lu, double wo, double wm, double wn, int j0, int j1,
                 int i0, int i1) {
  const double wbaro = S->wbaro, dlt = S->dlt;
  for (int j = j0; j <= j1; j++)
    for (int i = i0; i <= i1; i++) {
      if (!A2(S, iv, i, j)) continue;
      A2(S, vbflxs_t, i, j) = A2(S, vbflxs_t, i, j) - wbaro * A3(S, vbflx_t, i, j, nl) +
                              (1. + wbaro) * A3(S, vbflx_t, i, j, ml);
      double q;
      if (S->mommth == 0)
        q = -(A3(S, ubflx_t, i, j, lu) * A2(S, scuyi, i, j) + A3(S, ubflx_t, i + 1, j, lu) * A2(S, scuyi, i + 1, j) +
              A3(S, ubflx_t, i, j - 1, lu) * A2(S, scuyi, i, j - 1) +
              A3(S, ubflx_t, i + 1, j - 1, lu) * A2(S, scuyi, i + 1, j - 1)) *
            (wo * (A2(S, pvtrop_o, i, j) + A2(S, pvtrop_o, i + 1, j)) +
             wm * (A3(S, pvtrop, i, j, m) + A3(S, pvtrop, i + 1, j, m)) +
             wn * (A3(S, pvtrop, i, j, n) + A3(S, pvtrop, i + 1, j, n))) * .125;
      else
        q = -.25 * ((A3(S, ubflx_t, i, j, lu) * A2(S, scuyi, i, j) + A3(S, ubflx_t, i, j - 1, lu) * A2(S, scuyi, i, j - 1)) *
                        (wo * A2(S, pvtrop_o, i, j) + wm * A3(S, pvtrop, i, j, m) + wn * A3(S, pvtrop, i, j, n)) +
                    (A3(S, ubflx_t, i + 1, j, lu) * A2(S, scuyi, i + 1, j) +
                     A3(S, ubflx_t, i + 1, j - 1, lu) * A2(S, scuyi, i + 1, j - 1)) *
                        (wo * A2(S, pvtrop_o, i + 1, j) + wm * A3(S, pvtrop, i + 1, j, m) + wn * A3(S, pvtrop, i + 1, j, n)));
      A2(S, vbcors_t, i, j) = A2(S, vbcors_t, i, j) + q;
      double vtndcy =
          q + (wo * (A2(S, pgfym_o, i, j) - (A2(S, xiyp_o, i, j) * A3(S, pb_t, i, j, nl) - A2(S, xiym_o, i, j) * A3(S, pb_t, i, j - 1, nl))) +
               wm * (A3(S, pgfym, i, j, m) - (A3(S, xiyp, i, j, m) * A3(S, pb_t, i, j, nl) - A3(S, xiym, i, j, m) * A3(S, pb_t, i, j - 1, nl))) +
               wn * (A3(S, pgfym, i, j, n) - (A3(S, xiyp, i, j, n) * A3(S, pb_t, i, j, nl) - A3(S, xiym, i, j, n) * A3(S, pb_t, i, j - 1, nl)))) *
                  A2(S, scvyi, i, j);
      double x = (1. - wbaro) * A3(S, vbflx_t, i, j, ml) + wbaro * A3(S, vbflx_t, i, j, nl) +
                 (1. + wbaro) * dlt *
                     ((vtndcy + A2(S, vtotn, i, j)) * A2(S, scvx, i, j) * fmin2(A3(S, pb_t, i, j - 1, nl), A3(S, pb_t, i, j, nl)) -
                      A2(S, vglue, i, j) * A3(S, vbflx_t, i, j, ml));
      A3(S, vbflx_t, i, j, nl) = fmax2(-A2(S, vminb, i, j), fmin2(A2(S, vmaxb, i, j), x));
    }
}

void orc_barotp(OState *S, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)mm; (void)k1m; (void)k1n;
  const int ii = S->ii, jj = S->jj, kk = S->kk, lstep = S->lstep;
  const size_t lev = (size_t)S->nplane;
  /* :177-224 */
  for (int j = 1; j <= jj; j++)
    for (int i = 1; i <= ii; i++) {
      if (A2(S, iu, i, j)) {
        double mx = 0., mn = 0.;
        A2(S, uglue, i, j) = S->cwbdts * exp(1. - A3(S, pbu, i, j, m) / (S->cwbdls * ONEM));
        for (int k = 1; k <= kk; k++) {
          mx = fmax2(mx, A3(S, u, i, j, k + nn));
          mn = fmin2(mn, A3(S, u, i, j, k + nn));
        }
        A2(S, umaxb, i, j) = (A2(S, umax, i, j) - mx) * A3(S, pbu, i, j, m) * A2(S, scuy, i, j);
        A2(S, uminb, i, j) = (A2(S, umax, i, j) + mn) * A3(S, pbu, i, j, m) * A2(S, scuy, i, j);
      }
      if (A2(S, iv, i, j)) {
        double mx = 0., mn = 0.;
        A2(S, vglue, i, j) = S->cwbdts * exp(1. - A3(S, pbv, i, j, m) / (S->cwbdls * ONEM));
        for (int k = 1; k <= kk; k++) {
          mx = fmax2(mx, A3(S, v, i, j, k + nn));
          mn = fmin2(mn, A3(S, v, i, j, k + nn));
        }
        A2(S, vmaxb, i, j) = (A2(S, vmax, i, j) - mx) * A3(S, pbv, i, j, m) * A2(S, scvx, i, j);
        A2(S, vminb, i, j) = (A2(S, vmax, i, j) + mn) * A3(S, pbv, i, j, m) * A2(S, scvx, i, j);
      }
    }
  /* :230-268 potential vorticity of barotropic flow */
  for (int j = -2; j <= jj + 3; j++)
    for (int i = 0; i <= ii + 1; i++) A2(S, pvtrop_o, i, j) = A3(S, pvtrop, i, j, n);
  for (int j = 0; j <= jj; j++)
    for (int i = 1; i <= ii; i++)
      if (A2(S, iu, i, j)) {
        double q = 2. / (A2(S, pb_p, i, j) + A2(S, pb_p, i - 1, j));
        A3(S, pvtrop, i, j, n) = A2(S, corioq, i, j) * q;
        A3(S, pvtrop, i, j + 1, n) = A2(S, corioq, i, j + 1) * q;
      }
  for (int j = 1; j <= jj; j++)
    for (int i = 0; i <= ii; i++)
      if (A2(S, iv, i, j)) {
        double q = 2. / (A2(S, pb_p, i, j) + A2(S, pb_p, i, j - 1));
        A3(S, pvtrop, i, j, n) = A2(S, corioq, i, j) * q;
        A3(S, pvtrop, i + 1, j, n) = A2(S, corioq, i + 1, j) * q;
      }
  for (int j = 1; j <= jj; j++)
    for (int i = 1; i <= ii; i++)
      if (A2(S, iq, i, j))
        A3(S, pvtrop, i, j, n) = A2(S, corioq, i, j) * 4. /
                                 (A2(S, pb_p, i, j) + A2(S, pb_p, i - 1, j) + A2(S, pb_p, i, j - 1) + A2(S, pb_p, i - 1, j - 1));
  /* :271-285 */
  orc_xctilr(S, S->uglue, 1, 1, 1, 2, 3);
  orc_xctilr(S, S->utotn, 1, 1, 1, 2, 13);
  orc_xctilr(S, S->umaxb, 1, 1, 1, 2, 3);
  orc_xctilr(S, S->uminb, 1, 1, 1, 2, 3);
  orc_xctilr(S, S->vglue, 1, 1, 1, 2, 4);
  orc_xctilr(S, S->vtotn, 1, 1, 1, 2, 14);
  orc_xctilr(S, S->vmaxb, 1, 1, 1, 2, 4);
  orc_xctilr(S, S->vminb, 1, 1, 1, 2, 4);
  orc_xctilr(S, S->pvtrop + lev * (n - 1), 1, 1, 1, 3, 2);
  orc_xctilr(S, S->pgfxm + lev * (n - 1), 1, 1, 1, 2, 13);
  orc_xctilr(S, S->xixp + lev * (n - 1), 1, 1, 1, 2, 3);
  orc_xctilr(S, S->xixm + lev * (n - 1), 1, 1, 1, 2, 3);
  orc_xctilr(S, S->pgfym + lev * (n - 1), 1, 1, 1, 2, 14);
  orc_xctilr(S, S->xiyp + lev * (n - 1), 1, 1, 1, 2, 4);
  orc_xctilr(S, S->xiym + lev * (n - 1), 1, 1, 1, 2, 4);
  if (S->nreg == 2) { /* arctic patch: the min/max and +/- fields change roles across the seam, :290-325 */
    const int ii_ = S->ii, jj_ = S->jj;
#define SWAP(a, b) { const double q_ = (a); (a) = (b); (b) = q_; }
    for (int j = jj_; j <= jj_ + 2; j++)
      for (int i = 0; i <= ii_ + 1; i++) { SWAP(A2(S, umaxb, i, j), A2(S, uminb, i, j)); SWAP(A3(S, xixp, i, j, n), A3(S, xixm, i, j, n)); }
    for (int i = (ii_ / 2 + 1 > 0 ? ii_ / 2 + 1 : 0); i <= ii_ + 1; i++) {
      SWAP(A2(S, vmaxb, i, jj_), A2(S, vminb, i, jj_)); SWAP(A3(S, xiyp, i, jj_, n), A3(S, xiym, i, jj_, n));
    }
    for (int j = jj_ + 1; j <= jj_ + 2; j++)
      for (int i = 0; i <= ii_ + 1; i++) { SWAP(A2(S, vmaxb, i, j), A2(S, vminb, i, j)); SWAP(A3(S, xiyp, i, j, n), A3(S, xiym, i, j, n)); }
#undef SWAP
  }

  int lll0 = 1, ml = 1, nl = 2;
  double woa = 0., wob = 0., wna = 0., wnb = 0.;
  for (int nb = 1; nb <= 5; nb++) { /* :328 */
    if (nb == 1) {
      lll0 = 1; ml = 1; nl = 2;
      woa = -1. / lstep;
      wob = .5 + (lll0 - .5) / lstep;
      wna = 0.; wnb = 0.;
      for (int j = 1; j <= jj; j++)
        for (int i = 1; i <= ii; i++)
          for (int l = 1; l <= 2; l++) {
            A3(S, pb_t, i, j, l) = A3(S, pb_mn, i, j, l);
            A3(S, ubflx_t, i, j, l) = A3(S, ubflx_mn, i, j, l);
            A3(S, vbflx_t, i, j, l) = A3(S, vbflx_mn, i, j, l);
          }
    } else if (nb == 2) {
      woa = 0.; wob = 0.;
      wna = 1. / lstep;
      wnb = -(lll0 - .5) / lstep;
    } else if (nb == 4) {
      wna = 0.; wnb = 1.;
    }
    for (int j = -1; j <= jj + 2; j++) /* :361-368 */
      for (int i = 0; i <= ii + 1; i++)
        if (A2(S, iu, i, j)) { A2(S, ubflxs_t, i, j) = 0.; A2(S, ubcors_t, i, j) = 0.; }
    for (int j = 0; j <= jj + 2; j++) /* :372-379 */
      for (int i = 0; i <= ii; i++)
        if (A2(S, iv, i, j)) { A2(S, vbflxs_t, i, j) = 0.; A2(S, vbcors_t, i, j) = 0.; }
    for (int lll = lll0; lll <= lll0 + lstep / 2 - 1; lll++) {
      const double wo = woa * lll + wob, wn = wna * lll + wnb;
      const double wm = 1. - wo - wn;
      if (lll % 2 == 1) {
        orc_xctilr(S, S->pb_t, 1, 2, 2, 2, 1); /* :395-397 */
        orc_xctilr(S, S->ubflx_t, 1, 2, 2, 2, 13);
        orc_xctilr(S, S->vbflx_t, 1, 2, 2, 3, 14);
        cont(S, ml, nl, -1, jj + 2, -1, ii + 1);
        umom(S, m, n, ml, nl, ml, wo, wm, wn, -1, jj + 2, 0, ii + 1);
        vmom(S, m, n, ml, nl, nl, wo, wm, wn, 0, jj + 2, 0, ii);
      } else {
        cont(S, ml, nl, 0, jj + 1, 0, ii);
        vmom(S, m, n, ml, nl, ml, wo, wm, wn, 1, jj + 1, 0, ii);
        umom(S, m, n, ml, nl, nl, wo, wm, wn, 1, jj, 1, ii);
      }
      int ll = ml; ml = nl; nl = ll;
    }
    lll0 = lll0 + lstep / 2;
    /* :847-977 */
    for (int j = 1; j <= jj; j++)
      for (int i = 1; i <= ii; i++) {
        const int wp = A2(S, ip, i, j), wu = A2(S, iu, i, j), wv = A2(S, iv, i, j);
        if (nb == 1) {
          if (wp) A3(S, pb, i, j, m) = A3(S, pb_t, i, j, ml);
          if (wu) {
            A3(S, pbu, i, j, m) = fmin2(A3(S, pb_t, i, j, ml), A3(S, pb_t, i - 1, j, ml));
            A3(S, ubflx, i, j, m) = A3(S, ubflx_t, i, j, ml);
            A3(S, ub, i, j, m) = A3(S, ubflx, i, j, m) / (A3(S, pbu, i, j, m) * A2(S, scuy, i, j));
            A3(S, ubflxs, i, j, n) = A3(S, ubflxs, i, j, n) + A2(S, ubflxs_t, i, j);
            A3(S, ubflxs, i, j, m) = A3(S, ubflxs, i, j, 3) + A2(S, ubflxs_t, i, j);
          }
          if (wv) {
            A3(S, pbv, i, j, m) = fmin2(A3(S, pb_t, i, j, ml), A3(S, pb_t, i, j - 1, ml));
            A3(S, vbflx, i, j, m) = A3(S, vbflx_t, i, j, ml);
            A3(S, vb, i, j, m) = A3(S, vbflx, i, j, m) / (A3(S, pbv, i, j, m) * A2(S, scvx, i, j));
            A3(S, vbflxs, i, j, n) = A3(S, vbflxs, i, j, n) + A2(S, vbflxs_t, i, j);
            A3(S, vbflxs, i, j, m) = A3(S, vbflxs, i, j, 3) + A2(S, vbflxs_t, i, j);
          }
        } else if (nb == 2) {
          if (wp) { A3(S, pb_mn, i, j, ml) = A3(S, pb_t, i, j, ml); A3(S, pb_mn, i, j, nl) = A3(S, pb_t, i, j, nl); }
          if (wu) {
            A3(S, ubflx_mn, i, j, ml) = A3(S, ubflx_t, i, j, ml);
            A3(S, ubflx_mn, i, j, nl) = A3(S, ubflx_t, i, j, nl);
            A3(S, ubflxs, i, j, m) = A3(S, ubflxs, i, j, m) + A2(S, ubflxs_t, i, j);
            A3(S, ubflxs, i, j, 3) = A2(S, ubflxs_t, i, j);
            A3(S, ubflxs_p, i, j, n) = A2(S, ubflxs_t, i, j);
            A2(S, ubcors_p, i, j) = A2(S, ubcors_t, i, j);
          }
          if (wv) {
            A3(S, vbflx_mn, i, j, ml) = A3(S, vbflx_t, i, j, ml);
            A3(S, vbflx_mn, i, j, nl) = A3(S, vbflx_t, i, j, nl);
            A3(S, vbflxs, i, j, m) = A3(S, vbflxs, i, j, m) + A2(S, vbflxs_t, i, j);
            A3(S, vbflxs, i, j, 3) = A2(S, vbflxs_t, i, j);
            A3(S, vbflxs_p, i, j, n) = A2(S, vbflxs_t, i, j);
            A2(S, vbcors_p, i, j) = A2(S, vbcors_t, i, j);
          }
        } else if (nb == 3) {
          if (wp) A3(S, pb, i, j, n) = A3(S, pb_t, i, j, ml);
          if (wu) {
            A3(S, pbu, i, j, n) = fmin2(A3(S, pb_t, i, j, ml), A3(S, pb_t, i - 1, j, ml));
            A3(S, ubflx, i, j, n) = A3(S, ubflx_t, i, j, ml);
            A3(S, ub, i, j, n) = A3(S, ubflx, i, j, n) / (A3(S, pbu, i, j, n) * A2(S, scuy, i, j));
            A3(S, ubflxs_p, i, j, m) = A3(S, ubflxs, i, j, m) + A2(S, ubflxs_t, i, j);
            A3(S, ubflxs_p, i, j, n) = A3(S, ubflxs_p, i, j, n) + A2(S, ubflxs_t, i, j);
            A2(S, ubcors_p, i, j) = A2(S, ubcors_p, i, j) + A2(S, ubcors_t, i, j);
          }
          if (wv) {
            A3(S, pbv, i, j, n) = fmin2(A3(S, pb_t, i, j, ml), A3(S, pb_t, i, j - 1, ml));
            A3(S, vbflx, i, j, n) = A3(S, vbflx_t, i, j, ml);
            A3(S, vb, i, j, n) = A3(S, vbflx, i, j, n) / (A3(S, pbv, i, j, n) * A2(S, scvx, i, j));
            A3(S, vbflxs_p, i, j, m) = A3(S, vbflxs, i, j, m) + A2(S, vbflxs_t, i, j);
            A3(S, vbflxs_p, i, j, n) = A3(S, vbflxs_p, i, j, n) + A2(S, vbflxs_t, i, j);
            A2(S, vbcors_p, i, j) = A2(S, vbcors_p, i, j) + A2(S, vbcors_t, i, j);
          }
        } else {
          if (nb == 5) {
            if (wp) A2(S, pb_p, i, j) = A3(S, pb_t, i, j, ml);
            if (wu) A2(S, pbu_p, i, j) = fmin2(A3(S, pb_t, i, j, ml), A3(S, pb_t, i - 1, j, ml));
            if (wv) A2(S, pbv_p, i, j) = fmin2(A3(S, pb_t, i, j, ml), A3(S, pb_t, i, j - 1, ml));
          }
          if (wu) {
            A3(S, ubflxs_p, i, j, n) = A3(S, ubflxs_p, i, j, n) + A2(S, ubflxs_t, i, j);
            A2(S, ubcors_p, i, j) = A2(S, ubcors_p, i, j) + A2(S, ubcors_t, i, j);
          }
          if (wv) {
            A3(S, vbflxs_p, i, j, n) = A3(S, vbflxs_p, i, j, n) + A2(S, vbflxs_t, i, j);
            A2(S, vbcors_p, i, j) = A2(S, vbcors_p, i, j) + A2(S, vbcors_t, i, j);
          }
        }
      }
  }
}
