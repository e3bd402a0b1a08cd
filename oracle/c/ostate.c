/* TEST INFRASTRUCTURE (oracle): state container, option setters and stage dispatcher of
 * the plain-C restatement (see ostate.h). */
#include "ostate.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

OState *orc_create(int idm, int jdm, int kdm, int ntr, int nreg) {
  OState *S = (OState *)calloc(1, sizeof(OState));
  S->ii = idm; S->jj = jdm; S->kk = kdm; S->ni = idm + 2 * NBDY; S->nj = jdm + 2 * NBDY;
  S->nplane = S->ni * S->nj; S->nreg = nreg; S->ntr = ntr;
  const int K = kdm, NT = ntr > 0 ? ntr : 1;
#define X(name, lev) S->name = (double *)calloc((size_t)(lev) * S->nplane, sizeof(double));
  ORC_REAL_FIELDS(X)
#undef X
#define X(name, lev) S->name = (int *)calloc((size_t)(lev) * S->nplane, sizeof(int));
  ORC_INT_FIELDS(X)
#undef X
  /* module defaults, phy/mod_tmsmt.F90:46-51 */
  S->wuv1 = .75; S->wuv2 = .125; S->wts1 = .875; S->wts2 = .0625; S->wbaro = .125;
  S->vland = 0.; S->vcoord_tag = 1; S->ltedtp_opt = 1; S->eitmth = 2;   /* eitmth default gm, phy/mod_diffusion.F90 */
  eos_set_pref(S, 2000.e4);
  return S;
}

void orc_destroy(OState *S) {
#define X(name, lev) free(S->name);
  ORC_REAL_FIELDS(X)
  ORC_INT_FIELDS(X)
#undef X
  free(S);
}

void *orc_field(OState *S, const char *name, int *nlev, int *is_int) {
  const int K = S->kk, NT = S->ntr > 0 ? S->ntr : 1;
  *is_int = 0;
#define X(nm, lev) if (!strcmp(name, #nm)) { *nlev = (lev); return S->nm; }
  ORC_REAL_FIELDS(X)
#undef X
  *is_int = 1;
#define X(nm, lev) if (!strcmp(name, #nm)) { *nlev = (lev); return S->nm; }
  ORC_INT_FIELDS(X)
#undef X
  *nlev = 0;
  return NULL;
}

int orc_set_real(OState *S, const char *name, double v) {
#define R(nm) if (!strcmp(name, #nm)) { S->nm = v; return 0; }
  R(baclin) R(batrop) R(delt1) R(dlt) R(mdv2hi) R(mdv2lo) R(mdv4hi) R(mdv4lo) R(mdc2hi)
  R(mdc2lo) R(vsc2hi) R(vsc2lo) R(vsc4hi) R(vsc4lo) R(cbar) R(cb) R(cwbdts) R(cwbdls)
  R(wuv1) R(wuv2) R(wts1) R(wts2) R(wbaro) R(bdmc1) R(bdmc2) R(iwdfac) R(nubmin) R(vland)
#undef R
  if (!strcmp(name, "pref")) { eos_set_pref(S, v); return 0; }
  return 1;
}

int orc_set_int(OState *S, const char *name, int v) {
#define R(nm) if (!strcmp(name, #nm)) { S->nm = v; return 0; }
  R(lstep) R(nstep) R(nday_in_year) R(itriag) R(cnsvdi) R(itrtke) R(itrgls) R(tkeadv) R(tkeidf) R(gls) R(vcoord_tag) R(ltedtp_opt) R(bdmtyp) R(iwdflg) R(bdmldp)
#undef R
  return 1;
}

int orc_set_str(OState *S, const char *name, const char *v) {
  if (!strcmp(name, "mommth")) {
    S->mommth = !strcmp(v, "enscon") ? 0 : !strcmp(v, "enecon") ? 1 : !strcmp(v, "enedis") ? 2 : -1;
    return S->mommth < 0;
  }
  if (!strcmp(name, "pgfmth")) {
    S->pgfmth = !strcmp(v, "geopotential") ? 0 : !strcmp(v, "dynamic enthalpy") ? 1 : -1;
    return S->pgfmth < 0;
  }
  if (!strcmp(name, "advmth")) { S->advmth = !strcmp(v, "remap") ? 0 : !strcmp(v, "cppm") ? 1 : -1; return S->advmth < 0; }
  if (!strcmp(name, "bmcmth")) { S->bmcmth = !strcmp(v, "uc") ? 0 : !strcmp(v, "dluc") ? 1 : -1; return S->bmcmth < 0; }
  if (!strcmp(name, "eitmth")) { S->eitmth = !strcmp(v, "intdif") ? 1 : !strcmp(v, "gm") ? 2 : -1; return S->eitmth < 0; }
  if (!strcmp(name, "expcnf")) return 0;
  return 1;
}

int orc_stage(OState *S, const char *st, int m, int n, int mm, int nn, int k1m, int k1n) {
  const int kk = S->kk;
  if (!strcmp(st, "init_fluxes")) orc_init_fluxes(S, m, n, mm, nn, k1m, k1n);
  else if (!strcmp(st, "tmsmt1")) orc_tmsmt1(S, nn);
  else if (!strcmp(st, "tmsmt2")) orc_tmsmt2(S, m, mm, nn, k1m);
  else if (!strcmp(st, "initms")) orc_initms(S, mm);
  else if (!strcmp(st, "diffus")) orc_diffus(S, m, n, mm, nn, k1m, k1n);
  else if (!strcmp(st, "pgforc")) orc_pgforc(S, m, n, mm, nn, k1m, k1n);
  else if (!strcmp(st, "advect")) orc_advect(S, m, n, mm, nn, k1m, k1n);
  else if (!strcmp(st, "pbcor1")) orc_pbcor1(S, m, n, mm, nn, k1m, k1n);
  else if (!strcmp(st, "pbcor2")) orc_pbcor2(S, m, n, mm, nn, k1m, k1n);
  else if (!strcmp(st, "momtum")) orc_momtum(S, m, n, mm, nn, k1m, k1n);
  else if (!strcmp(st, "barotp")) orc_barotp(S, m, n, mm, nn, k1m, k1n);
  else if (!strcmp(st, "convec")) orc_convec(S, m, n, mm, nn, k1m, k1n);
  else if (!strcmp(st, "updtrc")) orc_updtrc(S, m, n, mm, nn, k1m, k1n);
  else if (!strcmp(st, "diapfl")) orc_diapfl(S, n, nn, k1n);
  else if (!strcmp(st, "eddtra")) return orc_eddtra(S, m, n, mm, nn, k1m, k1n);
  else if (!strcmp(st, "cmnfld2")) return orc_cmnfld2(S, m, n, mm, nn, k1m, k1n);
  else if (!strcmp(st, "cmnfld1")) return orc_cmnfld1(S, m, n, mm, nn, k1m, k1n);
  else if (!strcmp(st, "mxlayr_tail")) orc_mxlayr_tail(S, nn, k1n);
  else if (!strcmp(st, "halo_cmnfld2")) {  /* phy/mod_cmnfld_routines.F90:1171-1196 */
    orc_xctilr(S, S->temp, 1, 2 * kk, 3, 3, 1);
    orc_xctilr(S, S->saln, 1, 2 * kk, 3, 3, 1);
    for (int j = 1; j <= S->jj; j++)        /* kfpla halo through util1, :1176-1196 */
      for (int i = 1; i <= S->ii; i++)
        if (A2(S, ip, i, j)) A2(S, util1, i, j) = (double)A3(S, kfpla, i, j, n);
    orc_xctilr(S, S->util1, 1, 1, 2, 2, 1);
    for (int j = -1; j <= S->jj + 2; j++)
      for (int i = -1; i <= S->ii + 2; i++)
        if (A2(S, ip, i, j)) A3(S, kfpla, i, j, n) = (int)lround(A2(S, util1, i, j));
  } else if (!strcmp(st, "halo_difest")) { /* phy/mod_difest.F90:750-772 */
    orc_xctilr(S, S->u, 1, 2 * kk, 2, 2, 13);
    orc_xctilr(S, S->v, 1, 2 * kk, 2, 2, 14);
    orc_xctilr(S, S->ubflxs_p, 1, 2, 2, 2, 13);
    orc_xctilr(S, S->vbflxs_p, 1, 2, 2, 2, 14);
    orc_xctilr(S, S->pbu, 1, 2, 2, 2, 3);
    orc_xctilr(S, S->pbv, 1, 2, 2, 2, 4);
    for (int j = -2; j <= S->jj + 3; j++)   /* phy/mod_difest.F90:761-772 */
      for (int k = 1; k <= kk; k++)
        for (int i = -2; i <= S->ii + 3; i++)
          if (A2(S, ip, i, j)) A3(S, p, i, j, k + 1) = A3(S, p, i, j, k) + A3(S, dp, i, j, k + nn);
  } else return 1;
  return 0;
}

double orc_budget_get(const OState *S, int which, int ncall, int n) { return S->budget[which][ncall - 1][n - 1]; }

/* xcsum of level lev (1-based) of a named field; itype as for xctilr/xccrc selects the mask */
double orc_xcsum_field(OState *S, const char *name, int lev, int itype) {
  int nlev = 0, isint = 0;
  const double *a = (const double *)orc_field(S, name, &nlev, &isint);
  if (!a || isint || lev < 1 || lev > nlev) return 0. / 0.;
  const int g = itype % 10;
  const int *mask = g == 1 ? S->ip : g == 2 ? S->iq : g == 3 ? S->iu : S->iv;
  return orc_xcsum(S, a + (size_t)(lev - 1) * S->nplane, mask, g == 1);
}
