/* TEST INFRASTRUCTURE (oracle): restatement of pbcor1 and pbcor2, phy/mod_pbcor.F90:66-412 and
 * :416-743 (use_TRC, no TKE tracers; bmcmth 'uc' and 'dluc'). */
#include "ostate.h"
#include <stdlib.h>

#define DPEPS1 1.e-5 /* phy/mod_pbcor.F90:58-59 */
#define DPEPS2 1.e-7

/* common body of pbcor1 (which=1) and pbcor2 (which=2) */
static void pbcor(OState *S, int which, int m, int n, int mm, int nn, int k1m) {
  const int ii = S->ii, jj = S->jj, kk = S->kk, ntr = S->ntr;
  const size_t lev = (size_t)S->nplane;
  const double dlt = S->dlt;
  /* level roles: pbcor1 corrects the new level (kn) with fluxes accumulated at km,
   * pbcor2 corrects the mid level (km) with fluxes accumulated at kn */
  const int offc = which == 1 ? nn : mm; /* corrected level  */
  const int offf = which == 1 ? mm : nn; /* flux level       */
  double *tot_u = which == 1 ? S->utotm : S->utotn, *tot_v = which == 1 ? S->vtotm : S->vtotn;
  const double *bfx_u = which == 1 ? S->ubflxs_p + lev * (m - 1) : S->ubflxs + lev * (n - 1);
  const double *bfx_v = which == 1 ? S->vbflxs_p + lev * (m - 1) : S->vbflxs + lev * (n - 1);
  double *pbu_t = (double *)calloc(lev, sizeof(double)), *pbv_t = (double *)calloc(lev, sizeof(double));
  double *uflxtr = (double *)calloc(lev * (ntr > 0 ? ntr : 1), sizeof(double));
  double *vflxtr = (double *)calloc(lev * (ntr > 0 ? ntr : 1), sizeof(double));
#define UTR(nt, i, j) uflxtr[IX(S, i, j) + lev * ((nt)-1)]
#define VTR(nt, i, j) vflxtr[IX(S, i, j) + lev * ((nt)-1)]

  if (which == 2) { /* :434-440 */
    orc_xctilr(S, S->ubflxs + lev * (n - 1), 1, 1, 1, 1, 13);
    orc_xctilr(S, S->vbflxs + lev * (n - 1), 1, 1, 1, 1, 14);
    for (int nt = 1; nt <= ntr; nt++)
      orc_xctilr(S, S->trc + lev * ((size_t)(k1m - 1) + 2 * kk * (nt - 1)), 1, kk, 1, 1, 1);
  }
  for (int j = 0; j <= jj + 1; j++) /* pbcor1 :84-95, pbcor2 :442-454 */
    for (int k = 1; k <= kk; k++)
      for (int i = 0; i <= ii + 1; i++)
        if (A2(S, ip, i, j)) {
          if (which == 2) A3(S, dp, i, j, k + offc) = fmax2(0., A3(S, dp, i, j, k + offc)) + EPSILP;
          A3(S, p, i, j, k + 1) = A3(S, p, i, j, k) + A3(S, dp, i, j, k + offc);
        }
  for (int j = 1; j <= jj; j++) /* :97-128 / :456-487 */
    for (int i = 1; i <= ii + 1; i++) {
      if (!A2(S, iu, i, j)) continue;
      tot_u[IX(S, i, j)] = dlt * bfx_u[IX(S, i, j)];
      if (S->bmcmth == 1) pbu_t[IX(S, i, j)] = fmin2(A3(S, p, i, j, kk + 1), A3(S, p, i - 1, j, kk + 1));
      for (int k = 1; k <= kk; k++) tot_u[IX(S, i, j)] = tot_u[IX(S, i, j)] - A3(S, uflx, i, j, k + offf);
    }
  for (int j = 1; j <= jj + 1; j++) /* :130-161 / :488-519 */
    for (int i = 1; i <= ii; i++) {
      if (!A2(S, iv, i, j)) continue;
      tot_v[IX(S, i, j)] = dlt * bfx_v[IX(S, i, j)];
      if (S->bmcmth == 1) pbv_t[IX(S, i, j)] = fmin2(A3(S, p, i, j, kk + 1), A3(S, p, i, j - 1, kk + 1));
      for (int k = 1; k <= kk; k++) tot_v[IX(S, i, j)] = tot_v[IX(S, i, j)] - A3(S, vflx, i, j, k + offf);
    }
  for (int k = 1; k <= kk; k++) { /* :163-367 / :521-697 */
    const int kc = k + offc, kf = k + offf;
    for (int j = 1; j <= jj; j++)
      for (int i = 1; i <= ii + 1; i++) {
        if (!A2(S, iu, i, j)) continue;
        const double tot = tot_u[IX(S, i, j)];
        const int iup = tot > 0. ? i - 1 : i;
        double f;
        if (S->bmcmth == 0) f = tot * A3(S, dp, iup, j, kc) / A3(S, p, iup, j, kk + 1);
        else f = tot * fmax2(0., fmin2(pbu_t[IX(S, i, j)], A3(S, p, iup, j, k + 1)) - A3(S, p, iup, j, k)) /
                 pbu_t[IX(S, i, j)];
        A2(S, uflux, i, j) = f;
        A2(S, uflux2, i, j) = f * A3(S, saln, iup, j, kc);
        A2(S, uflux3, i, j) = f * A3(S, temp, iup, j, kc);
        for (int nt = 1; nt <= ntr; nt++) UTR(nt, i, j) = f * TRC(S, iup, j, kc, nt);
        A3(S, uflx, i, j, kf) = A3(S, uflx, i, j, kf) + A2(S, uflux, i, j);
        A3(S, usflx, i, j, kf) = A3(S, usflx, i, j, kf) + A2(S, uflux2, i, j);
        A3(S, utflx, i, j, kf) = A3(S, utflx, i, j, kf) + A2(S, uflux3, i, j);
      }
    for (int j = 1; j <= jj + 1; j++)
      for (int i = 1; i <= ii; i++) {
        if (!A2(S, iv, i, j)) continue;
        const double tot = tot_v[IX(S, i, j)];
        const int jup = tot > 0. ? j - 1 : j;
        double f;
        if (S->bmcmth == 0) f = tot * A3(S, dp, i, jup, kc) / A3(S, p, i, jup, kk + 1);
        else f = tot * fmax2(0., fmin2(pbv_t[IX(S, i, j)], A3(S, p, i, jup, k + 1)) - A3(S, p, i, jup, k)) /
                 pbv_t[IX(S, i, j)];
        A2(S, vflux, i, j) = f;
        A2(S, vflux2, i, j) = f * A3(S, saln, i, jup, kc);
        A2(S, vflux3, i, j) = f * A3(S, temp, i, jup, kc);
        for (int nt = 1; nt <= ntr; nt++) VTR(nt, i, j) = f * TRC(S, i, jup, kc, nt);
        A3(S, vflx, i, j, kf) = A3(S, vflx, i, j, kf) + A2(S, vflux, i, j);
        A3(S, vsflx, i, j, kf) = A3(S, vsflx, i, j, kf) + A2(S, vflux2, i, j);
        A3(S, vtflx, i, j, kf) = A3(S, vtflx, i, j, kf) + A2(S, vflux3, i, j);
      }
    for (int j = 1; j <= jj; j++)
      for (int i = 1; i <= ii; i++) {
        if (!A2(S, ip, i, j)) continue;
        double dpo = A3(S, dp, i, j, kc);
        const double dv = A2(S, uflux, i + 1, j) - A2(S, uflux, i, j) + A2(S, vflux, i, j + 1) - A2(S, vflux, i, j);
        const double dv2 = A2(S, uflux2, i + 1, j) - A2(S, uflux2, i, j) + A2(S, vflux2, i, j + 1) - A2(S, vflux2, i, j);
        const double dv3 = A2(S, uflux3, i + 1, j) - A2(S, uflux3, i, j) + A2(S, vflux3, i, j + 1) - A2(S, vflux3, i, j);
        const double s2i = A2(S, scp2i, i, j);
        if (which == 1) { /* :339-361 */
          A3(S, dp, i, j, kc) = fmax2(0., dpo - dv * s2i);
          dpo = dpo + DPEPS1;
          const double dpni = 1. / (A3(S, dp, i, j, kc) + DPEPS1);
          A3(S, saln, i, j, kc) = (dpo * A3(S, saln, i, j, kc) - dv2 * s2i) * dpni;
          A3(S, temp, i, j, kc) = (dpo * A3(S, temp, i, j, kc) - dv3 * s2i) * dpni;
          for (int nt = 1; nt <= ntr; nt++)
            if (!orc_skip_adv(S, nt)) /* pbcor1 only: :353-355; pbcor2 :684 has no such test */ TRC(S, i, j, kc, nt) = (dpo * TRC(S, i, j, kc, nt) -
                                    (UTR(nt, i + 1, j) - UTR(nt, i, j) + VTR(nt, i, j + 1) - VTR(nt, i, j)) * s2i) * dpni;
          if (A3(S, dp, i, j, kc) < DPEPS2) A3(S, dp, i, j, kc) = 0.;
        } else { /* :671-692 */
          A3(S, dp, i, j, kc) = dpo - s2i * dv;
          const double dpni = 1. / A3(S, dp, i, j, kc);
          A3(S, saln, i, j, kc) = (dpo * A3(S, saln, i, j, kc) - s2i * dv2) * dpni;
          A3(S, temp, i, j, kc) = (dpo * A3(S, temp, i, j, kc) - s2i * dv3) * dpni;
          for (int nt = 1; nt <= ntr; nt++)
            TRC(S, i, j, kc, nt) = (dpo * TRC(S, i, j, kc, nt) -
                                    (UTR(nt, i + 1, j) - UTR(nt, i, j) + VTR(nt, i, j + 1) - VTR(nt, i, j)) * s2i) * dpni;
          A3(S, sigma, i, j, kc) = eos_sig(S, A3(S, temp, i, j, kc), A3(S, saln, i, j, kc));
          A3(S, dp, i, j, kc) = A3(S, dp, i, j, kc) - EPSILP;
          if (A3(S, dp, i, j, kc) < DPEPS2) A3(S, dp, i, j, kc) = 0.;
        }
      }
  }
  for (int j = 1; j <= jj; j++) /* :369-392 / :699-723 */
    for (int i = 1; i <= ii; i++) {
      if (!A2(S, ip, i, j)) continue;
      for (int k = 1; k <= kk; k++) A3(S, p, i, j, k + 1) = A3(S, p, i, j, k) + A3(S, dp, i, j, k + offc);
      const double pbfac = (which == 1 ? A2(S, pb_p, i, j) : A3(S, pb, i, j, m)) / A3(S, p, i, j, kk + 1);
      for (int k = 1; k <= kk; k++) {
        A3(S, dp, i, j, k + offc) = A3(S, dp, i, j, k + offc) * pbfac;
        if (which == 2) A3(S, p, i, j, k + 1) = A3(S, p, i, j, k) + A3(S, dp, i, j, k + offc);
      }
    }
  free(pbu_t); free(pbv_t); free(uflxtr); free(vflxtr);
}

void orc_pbcor1(OState *S, int m, int n, int mm, int nn, int k1m, int k1n) { (void)k1n; pbcor(S, 1, m, n, mm, nn, k1m); }
void orc_pbcor2(OState *S, int m, int n, int mm, int nn, int k1m, int k1n) { (void)k1n; pbcor(S, 2, m, n, mm, nn, k1m); }
