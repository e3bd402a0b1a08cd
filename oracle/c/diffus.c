/* TEST INFRASTRUCTURE (oracle): restatement of diffus, phy/mod_diffus.F90:41-185
 * (ltedtp_opt == ltedtp_layer). */
#include "ostate.h"
#include <stdlib.h>

void orc_diffus(OState *S, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)n; (void)k1m;
  const int ii = S->ii, jj = S->jj, kk = S->kk, ntr = S->ntr;
  const double dpeps = 1.e-5, delt1 = S->delt1;                       /* :55-56 */
  const size_t lev = (size_t)S->nplane;
  orc_xctilr(S, S->dp + lev * (k1n - 1), 1, kk, 3, 3, 1);             /* :58 */
  orc_xctilr(S, S->temp + lev * (k1n - 1), 1, kk, 2, 2, 1);           /* :72-73 */
  orc_xctilr(S, S->saln + lev * (k1n - 1), 1, kk, 2, 2, 1);
  for (int nt = 1; nt <= ntr; nt++)                                   /* :74-80 */
    if (!orc_skip_dif(S, nt)) orc_xctilr(S, S->trc + lev * ((size_t)(k1n - 1) + 2 * kk * (nt - 1)), 1, kk, 2, 2, 1);
  double *uflxtr = (double *)calloc((size_t)(ntr > 0 ? ntr : 1) * S->nplane, sizeof(double));
  double *vflxtr = (double *)calloc((size_t)(ntr > 0 ? ntr : 1) * S->nplane, sizeof(double));
#define UTR(nt, i, j) uflxtr[IX(S, i, j) + lev * ((nt)-1)]
#define VTR(nt, i, j) vflxtr[IX(S, i, j) + lev * ((nt)-1)]
  for (int k = 1; k <= kk; k++) {
    int kn = k + nn, km = k + mm;
    for (int j = 0; j <= jj + 1; j++)                                 /* :89-109 */
      for (int i = 0; i <= ii + 2; i++) {
        if (!A2(S, iu, i, j)) continue;
        double q = delt1 * .5 * (A3(S, difiso, i - 1, j, k) + A3(S, difiso, i, j, k)) * A2(S, scuy, i, j) *
                   A2(S, scuxi, i, j) * fmax2(fmin2(A3(S, dp, i - 1, j, kn), A3(S, dp, i, j, kn)), dpeps);
        A3(S, usflld, i, j, km) = q * (A3(S, saln, i - 1, j, kn) - A3(S, saln, i, j, kn));
        A3(S, utflld, i, j, km) = q * (A3(S, temp, i - 1, j, kn) - A3(S, temp, i, j, kn));
        for (int nt = 1; nt <= ntr; nt++) if (!orc_skip_dif(S, nt)) UTR(nt, i, j) = q * (TRC(S, i - 1, j, kn, nt) - TRC(S, i, j, kn, nt));
        A3(S, usflx, i, j, km) = A3(S, usflx, i, j, km) + A3(S, usflld, i, j, km);
        A3(S, utflx, i, j, km) = A3(S, utflx, i, j, km) + A3(S, utflld, i, j, km);
      }
    for (int j = 0; j <= jj + 2; j++)                                 /* :113-133 */
      for (int i = 0; i <= ii + 1; i++) {
        if (!A2(S, iv, i, j)) continue;
        double q = delt1 * .5 * (A3(S, difiso, i, j - 1, k) + A3(S, difiso, i, j, k)) * A2(S, scvx, i, j) *
                   A2(S, scvyi, i, j) * fmax2(fmin2(A3(S, dp, i, j - 1, kn), A3(S, dp, i, j, kn)), dpeps);
        A3(S, vsflld, i, j, km) = q * (A3(S, saln, i, j - 1, kn) - A3(S, saln, i, j, kn));
        A3(S, vtflld, i, j, km) = q * (A3(S, temp, i, j - 1, kn) - A3(S, temp, i, j, kn));
        for (int nt = 1; nt <= ntr; nt++) if (!orc_skip_dif(S, nt)) VTR(nt, i, j) = q * (TRC(S, i, j - 1, kn, nt) - TRC(S, i, j, kn, nt));
        A3(S, vsflx, i, j, km) = A3(S, vsflx, i, j, km) + A3(S, vsflld, i, j, km);
        A3(S, vtflx, i, j, km) = A3(S, vtflx, i, j, km) + A3(S, vtflld, i, j, km);
      }
    for (int j = 0; j <= jj + 1; j++)                                 /* :137-160 */
      for (int i = 0; i <= ii + 1; i++) {
        if (!A2(S, ip, i, j)) continue;
        double q = 1. / (A2(S, scp2, i, j) * fmax2(A3(S, dp, i, j, kn), dpeps));
        A3(S, saln, i, j, kn) = A3(S, saln, i, j, kn) -
                                q * (A3(S, usflld, i + 1, j, km) - A3(S, usflld, i, j, km) +
                                     A3(S, vsflld, i, j + 1, km) - A3(S, vsflld, i, j, km));
        A3(S, temp, i, j, kn) = A3(S, temp, i, j, kn) -
                                q * (A3(S, utflld, i + 1, j, km) - A3(S, utflld, i, j, km) +
                                     A3(S, vtflld, i, j + 1, km) - A3(S, vtflld, i, j, km));
        for (int nt = 1; nt <= ntr; nt++)
          if (!orc_skip_dif(S, nt)) TRC(S, i, j, kn, nt) = TRC(S, i, j, kn, nt) -
                                 q * (UTR(nt, i + 1, j) - UTR(nt, i, j) + VTR(nt, i, j + 1) - VTR(nt, i, j));
        A3(S, sigma, i, j, kn) = eos_sig(S, A3(S, temp, i, j, kn), A3(S, saln, i, j, kn));
      }
  }
  free(uflxtr);
  free(vflxtr);
}
