/* TEST INFRASTRUCTURE (oracle): restatement of phy/mod_eos.F90. */
#include "ostate.h"
#include <math.h>

/* phy/mod_eos.F90:36-54 */
static const double a11 = 9.9985372432159340e+02, a12 = 1.0380621928183473e+01,
                    a13 = 1.7073577195684715e+00, a14 = -3.6570490496333680e-02,
                    a15 = -7.3677944503527477e-03, a16 = -3.5529175999643348e-03,
                    b11 = 1.7083494994335439e-06, b12 = 7.1567921402953455e-09,
                    b13 = 1.2821026080049485e-09, a21 = 1.0, a22 = 1.0316374535350838e-02,
                    a23 = 8.9521792365142522e-04, a24 = -2.8438341552142710e-05,
                    a25 = -1.1887778959461776e-05, a26 = -4.0163964812921489e-06,
                    b21 = 1.1995545126831476e-09, b22 = 5.5234008384648383e-12,
                    b23 = 8.4310335919950873e-13;

/* inieos, phy/mod_eos.F90:105-116 */
void eos_set_pref(OState *S, double pref) {
  S->pref = pref;
  S->ap21 = a21 + b21 * pref;
  S->ap22 = a22 + b22 * pref;
  S->ap23 = a23 + b23 * pref;
  S->ap24 = a24;
  S->ap25 = a25;
  S->ap26 = a26;
  S->ap11 = a11 + b11 * pref - S->ap21 / ALPHA0;
  S->ap12 = a12 + b12 * pref - S->ap22 / ALPHA0;
  S->ap13 = a13 + b13 * pref - S->ap23 / ALPHA0;
  S->ap14 = a14 - S->ap24 / ALPHA0;
  S->ap15 = a15 - S->ap25 / ALPHA0;
  S->ap16 = a16 - S->ap26 / ALPHA0;
}

/* sig, phy/mod_eos.F90:191-203 */
double eos_sig(const OState *S, double th, double s) {
  return (S->ap11 + (S->ap12 + S->ap14 * th + S->ap15 * s) * th + (S->ap13 + S->ap16 * s) * s) /
         (S->ap21 + (S->ap22 + S->ap24 * th + S->ap25 * s) * th + (S->ap23 + S->ap26 * s) * s);
}

/* rho, phy/mod_eos.F90:157-172 */
double eos_rho(double p, double th, double s) {
  return (a11 + (a12 + a14 * th + a15 * s) * th + (a13 + a16 * s) * s + (b11 + b12 * th + b13 * s) * p) /
         (a21 + (a22 + a24 * th + a25 * s) * th + (a23 + a26 * s) * s + (b21 + b22 * th + b23 * s) * p);
}

/* alp, phy/mod_eos.F90:174-189 */
double eos_alp(double p, double th, double s) {
  return (a21 + (a22 + a24 * th + a25 * s) * th + (a23 + a26 * s) * s + (b21 + b22 * th + b23 * s) * p) /
         (a11 + (a12 + a14 * th + a15 * s) * th + (a13 + a16 * s) * s + (b11 + b12 * th + b13 * s) * p);
}

/* p_alpha, phy/mod_eos.F90:386-428 */
double eos_p_alpha(double p1, double p2, double th, double s) {
  const double r1_3 = 1. / 3., r1_5 = 1. / 5., r1_7 = 1. / 7., r1_9 = 1. / 9.;
  double a1 = a11 + (a12 + a14 * th + a15 * s) * th + (a13 + a16 * s) * s;
  double a2 = a21 + (a22 + a24 * th + a25 * s) * th + (a23 + a26 * s) * s;
  double b1 = b11 + b12 * th + b13 * s;
  double b2 = b21 + b22 * th + b23 * s;
  double pm = .5 * (p2 + p1);
  double r = .5 * (p2 - p1) / (a1 + b1 * pm);
  double q = b1 * r;
  double qq = q * q;
  return 2. * r * (a2 + b2 * pm + (a2 - a1 * b2 / b1) * qq * (r1_3 + qq * (r1_5 + qq * (r1_7 + qq * r1_9))));
}

/* delphi, phy/mod_eos.F90:478-529 */
void eos_delphi(double p1, double p2, double th, double s, double *dphi, double *alp1, double *alp2) {
  const double r1_3 = 1. / 3., r1_5 = 1. / 5., r1_7 = 1. / 7., r1_9 = 1. / 9.;
  double a1 = a11 + (a12 + a14 * th + a15 * s) * th + (a13 + a16 * s) * s;
  double a2 = a21 + (a22 + a24 * th + a25 * s) * th + (a23 + a26 * s) * s;
  double b1 = b11 + b12 * th + b13 * s;
  double b2 = b21 + b22 * th + b23 * s;
  double pm = .5 * (p2 + p1);
  double r = .5 * (p2 - p1) / (a1 + b1 * pm);
  double q = b1 * r;
  double qq = q * q;
  *dphi = -2. * r * (a2 + b2 * pm + (a2 - a1 * b2 / b1) * qq * (r1_3 + qq * (r1_5 + qq * (r1_7 + qq * r1_9))));
  *alp1 = (a2 + b2 * p1) / (a1 + b1 * p1);
  *alp2 = (a2 + b2 * p2) / (a1 + b1 * p2);
}

/* dsigdt, phy/mod_eos.F90:243-261 */
double eos_dsigdt(const OState *S, double th, double s) {
  double r1 = S->ap11 + (S->ap12 + S->ap14 * th + S->ap15 * s) * th + (S->ap13 + S->ap16 * s) * s;
  double r2i = 1. / (S->ap21 + (S->ap22 + S->ap24 * th + S->ap25 * s) * th + (S->ap23 + S->ap26 * s) * s);
  return (S->ap12 + 2. * S->ap14 * th + S->ap15 * s - (S->ap22 + 2. * S->ap24 * th + S->ap25 * s) * r1 * r2i) * r2i;
}

/* dsigds, phy/mod_eos.F90:306-323 */
double eos_dsigds(const OState *S, double th, double s) {
  double r1 = S->ap11 + (S->ap12 + S->ap14 * th + S->ap15 * s) * th + (S->ap13 + S->ap16 * s) * s;
  double r2i = 1. / (S->ap21 + (S->ap22 + S->ap24 * th + S->ap25 * s) * th + (S->ap23 + S->ap26 * s) * s);
  return (S->ap13 + S->ap15 * th + 2. * S->ap16 * s - (S->ap23 + S->ap25 * th + 2. * S->ap26 * s) * r1 * r2i) * r2i;
}

/* sofsig, phy/mod_eos.F90:366-384 */
double eos_sofsig(const OState *S, double sg, double th) {
  double a = S->ap16 - S->ap26 * sg;
  double b = S->ap13 - S->ap23 * sg + (S->ap15 - S->ap25 * sg) * th;
  double c = S->ap11 - S->ap21 * sg + (S->ap12 - S->ap22 * sg + (S->ap14 - S->ap24 * sg) * th) * th;
  return (-b + sqrt(b * b - 4. * a * c)) / (2. * a);
}
