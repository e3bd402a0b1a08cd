/* TEST INFRASTRUCTURE (oracle) -- plain-C restatement of BLOM's dynamical-core stages.
 *
 * This is a CPU restatement of the reference algorithm (each function cites the
 * reference file:line it follows), pinned against the reference's own compiled code
 * (oracle/_ref/<cfg>/libblomref.so) by tests/test_oracle_vs_reference.py and against the
 * committed fixtures under tests/golden/.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it; the product (blom_amd/) never does.
 *
 * Arrays keep the Fortran layout a(1-nbdy:idm+nbdy,1-nbdy:jdm+nbdy,k), i fastest.
 * Arithmetic is written operator-for-operator as in the Fortran so that, compiled with
 * -ffp-contract=off, results are bit-identical to the reference built with its release
 * flags (meson.build:11,18-20).
 */
#ifndef OSTATE_H
#define OSTATE_H
#include <stddef.h>

#define NBDY 4
#define MAXTR 64

/* X(name, levels) -- K = kdm, NT = max(ntr,1) */
#define ORC_REAL_FIELDS(X)                                                                \
  X(u, 2 * K) X(v, 2 * K) X(dp, 2 * K) X(dpu, 2 * K) X(dpv, 2 * K) X(temp, 2 * K)         \
  X(saln, 2 * K) X(sigma, 2 * K) X(uflx, 2 * K) X(vflx, 2 * K) X(utflx, 2 * K)            \
  X(vtflx, 2 * K) X(usflx, 2 * K) X(vsflx, 2 * K)                                         \
  X(p, K + 1) X(pu, K + 1) X(pv, K + 1) X(phi, K + 1) X(cau, K) X(cav, K)                 \
  X(ubflxs, 3) X(vbflxs, 3) X(ub, 2) X(vb, 2) X(pb, 2) X(pbu, 2) X(pbv, 2)                \
  X(ubflxs_p, 2) X(vbflxs_p, 2) X(pb_p, 1) X(pbu_p, 1) X(pbv_p, 1) X(ubcors_p, 1)         \
  X(vbcors_p, 1) X(sealv, 1)                                                              \
  X(scqx, 1) X(scqy, 1) X(scpx, 1) X(scpy, 1) X(scux, 1) X(scuy, 1) X(scvx, 1)            \
  X(scvy, 1) X(scq2, 1) X(scp2, 1) X(scu2, 1) X(scv2, 1) X(scq2i, 1) X(scp2i, 1)          \
  X(scuxi, 1) X(scuyi, 1) X(scvxi, 1) X(scvyi, 1) X(corioq, 1) X(coriop, 1)               \
  X(betafp, 1) X(depths, 1)                                                               \
  X(pgfx, 2 * K) X(pgfy, 2 * K) X(pgfx_o, K) X(pgfy_o, K) X(pgfxm, 2) X(pgfym, 2)         \
  X(xixp, 2) X(xixm, 2) X(xiyp, 2) X(xiym, 2) X(pgfxm_o, 1) X(pgfym_o, 1)                 \
  X(xixp_o, 1) X(xixm_o, 1) X(xiyp_o, 1) X(xiym_o, 1)                                     \
  X(absvor, 2 * K) X(dpvor, 2 * K)                                                        \
  X(ubflx, 2) X(vbflx, 2) X(pb_mn, 2) X(ubflx_mn, 2) X(vbflx_mn, 2) X(pvtrop, 2)          \
  X(pvtrop_o, 1) X(pb_t, 2) X(ubflx_t, 2) X(vbflx_t, 2) X(umaxb, 1) X(uminb, 1)           \
  X(vmaxb, 1) X(vminb, 1) X(uglue, 1) X(vglue, 1) X(ubflxs_t, 1) X(vbflxs_t, 1)           \
  X(ubcors_t, 1) X(vbcors_t, 1)                                                           \
  X(dpold, 2 * K) X(dpuold, K) X(dpvold, K) X(told, K) X(sold, K)                         \
  X(sigmar, K) X(temmin, K) X(difint, K) X(difiso, K) X(difdia, K) X(difmxp, 1) X(difmxq, 1)           \
  X(difwgt, 1) X(umfltd, 2 * K) X(vmfltd, 2 * K) X(umflsm, 2 * K) X(vmflsm, 2 * K)        \
  X(utfltd, 2 * K) X(vtfltd, 2 * K) X(utflsm, 2 * K) X(vtflsm, 2 * K) X(utflld, 2 * K)    \
  X(vtflld, 2 * K) X(usfltd, 2 * K) X(vsfltd, 2 * K) X(usflsm, 2 * K) X(vsflsm, 2 * K)    \
  X(usflld, 2 * K) X(vsflld, 2 * K)                                                       \
  X(utotm, 1) X(vtotm, 1) X(utotn, 1) X(vtotn, 1) X(uflux, 1) X(vflux, 1) X(uflux2, 1)    \
  X(vflux2, 1) X(uflux3, 1) X(vflux3, 1) X(umax, 1) X(vmax, 1) X(util1, 1) X(util2, 1)    \
  X(util3, 1) X(util4, 1) X(taux, 1) X(tauy, 1) X(ustarb, 1)                              \
  X(trc, 2 * K * NT) X(trcold, K * NT) X(fpug, K) X(fplg, K) X(nslpx, K) X(nslpy, K)       \
  X(nnslpx, K) X(nnslpy, K) X(bfsqi, K + 1) X(bfsql, K) X(bfsqf, K + 1) X(z, K + 1) X(dz, K)

#define ORC_INT_FIELDS(X) X(ip, 1) X(iu, 1) X(iv, 1) X(iq, 1) X(kfpla, 2) X(kming, 1)

typedef struct {
  int ii, jj, kk, ni, nj, nplane, nreg, ntr;
  /* mod_time */
  double baclin, batrop, delt1, dlt;
  int lstep, nstep, nday_in_year, itriag, cnsvdi;
  /* the reference's TKE build options (phy/mod_ifdefs.F90:16-35) as run-time switches: itrtke >= 1 <=> use_TKE;
   * itrtke, itrgls = tracer indices (trc/mod_tracers.F90:87-91); tkeadv, tkeidf, gls <=> use_TKEADV, use_TKEIDF, use_GLS */
  int itrtke, itrgls, tkeadv, tkeidf, gls;
  double budget[4][7][2];    /* sdp, tdp, trdp, tkedp of mod_budget (ncall, n) */
  /* mod_eos */
  double pref, ap11, ap12, ap13, ap14, ap15, ap16, ap21, ap22, ap23, ap24, ap25, ap26;
  /* mod_momtum */
  double mdv2hi, mdv2lo, mdv4hi, mdv4lo, mdc2hi, mdc2lo, vsc2hi, vsc2lo, vsc4hi, vsc4lo, cbar, cb;
  /* mod_barotp, mod_tmsmt */
  double cwbdts, cwbdls, wuv1, wuv2, wts1, wts2, wbaro;
  /* mod_diffusion */
  double bdmc1, bdmc2, iwdfac, nubmin;
  int bdmtyp, iwdflg, bdmldp;
  int mommth, pgfmth, advmth, bmcmth, vcoord_tag, ltedtp_opt;
  int eitmth;      /* 1 intdif, 2 gm (phy/mod_diffusion.F90:112-113) */
  double vland;
#define X(name, lev) double *name;
  ORC_REAL_FIELDS(X)
#undef X
#define X(name, lev) int *name;
  ORC_INT_FIELDS(X)
#undef X
} OState;

/* Fortran a(i,j,k), k 1-based */
#define IX(S, i, j) ((size_t)((i) + NBDY - 1) + (size_t)(S)->ni * ((j) + NBDY - 1))
#define A2(S, a, i, j) ((S)->a[IX(S, i, j)])
#define A3(S, a, i, j, k) ((S)->a[IX(S, i, j) + (size_t)(S)->nplane * ((k)-1)])
/* tracer trc(i,j,k,nt): k over 2*kk */
#define TRC(S, i, j, k, nt) ((S)->trc[IX(S, i, j) + (size_t)(S)->nplane * (((k)-1) + 2 * (S)->kk * ((nt)-1))])
#define TRCOLD(S, i, j, k, nt) ((S)->trcold[IX(S, i, j) + (size_t)(S)->nplane * (((k)-1) + (S)->kk * ((nt)-1))])

/* Fortran MAX/MIN as amdflang lowers them (fcmp ogt/olt + select): ties -> 2nd operand */
static inline double fmax2(double a, double b) { return a > b ? a : b; }
static inline double fmin2(double a, double b) { return a < b ? a : b; }
static inline int imax2(int a, int b) { return a > b ? a : b; }
static inline int imin2(int a, int b) { return a < b ? a : b; }

/* constants, phy/mod_constants.F90:31-56 */
#define GRAV 9.806
#define ALPHA0 1.e-3
#define EPSILPL 1.e-14
#define EPSILP 1.e-12
#define SPVAL 1.e33
#define ONEM 9806.
#define ONECM 98.06
#define ONEMM 9.806

/* mod_eos restatement (eos.c) */
double eos_sig(const OState *S, double th, double s);
double eos_rho(double p, double th, double s);
double eos_alp(double p, double th, double s);
void eos_delphi(double p1, double p2, double th, double s, double *dphi, double *alp1, double *alp2);
double eos_p_alpha(double p1, double p2, double th, double s);
double eos_dsigdt(const OState *S, double th, double s);
double eos_dsigds(const OState *S, double th, double s);
double eos_sofsig(const OState *S, double sg, double th);
void eos_set_pref(OState *S, double pref);

/* stages */
void orc_xctilr(OState *S, double *a, int l1, int ld, int mh, int nh, int itype);
void orc_init_fluxes(OState *S, int m, int n, int mm, int nn, int k1m, int k1n);
void orc_tmsmt1(OState *S, int nn);
void orc_tmsmt2(OState *S, int m, int mm, int nn, int k1m);
void orc_initms(OState *S, int mm);
void orc_diffus(OState *S, int m, int n, int mm, int nn, int k1m, int k1n);
void orc_pgforc(OState *S, int m, int n, int mm, int nn, int k1m, int k1n);
void orc_advect(OState *S, int m, int n, int mm, int nn, int k1m, int k1n);
void orc_pbcor1(OState *S, int m, int n, int mm, int nn, int k1m, int k1n);
void orc_pbcor2(OState *S, int m, int n, int mm, int nn, int k1m, int k1n);
void orc_momtum(OState *S, int m, int n, int mm, int nn, int k1m, int k1n);
void orc_barotp(OState *S, int m, int n, int mm, int nn, int k1m, int k1n);
void orc_convec(OState *S, int m, int n, int mm, int nn, int k1m, int k1n);
void orc_updtrc(OState *S, int m, int n, int mm, int nn, int k1m, int k1n);
double orc_xcsum(const OState *S, const double *a, const int *mask, int use_ips);
void orc_budget_sums(OState *S, int ncall, int n, int nn);
/* tracers left out of advection (phy/mod_remap.F90:314-316 ..) and of layer diffusion (phy/mod_diffus.F90:64-66) */
static inline int orc_skip_adv(const OState *S, int nt) { return S->itrtke >= 1 && !S->tkeadv && (nt == S->itrtke || nt == S->itrgls); }
static inline int orc_skip_dif(const OState *S, int nt) { return S->itrtke >= 1 && !S->tkeidf && (nt == S->itrtke || nt == S->itrgls); }
#define ORC_TKE_MIN 7.6e-8      /* phy/mod_tke.F90:61 */
#define ORC_GLS_PSI_MIN 1.e-14  /* phy/mod_tke.F90:62 */
double orc_budget_get(const OState *S, int which, int ncall, int n);
double orc_xcsum_field(OState *S, const char *name, int lev, int itype);
void orc_diapfl(OState *S, int n, int nn, int k1n);
int orc_eddtra(OState *S, int m, int n, int mm, int nn, int k1m, int k1n);
int orc_cmnfld1(OState *S, int m, int n, int mm, int nn, int k1m, int k1n);  /* PARITY UNPINNED, see cmnfld.c */
int orc_cmnfld2(OState *S, int m, int n, int mm, int nn, int k1m, int k1n);  /* PARITY UNPINNED, see cmnfld.c */   /* PARITY UNPINNED, see eddtra.c */
void orc_mxlayr_tail(OState *S, int nn, int k1n);
#endif
