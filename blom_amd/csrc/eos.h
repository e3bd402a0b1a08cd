// Device functions of the equation of state, phy/mod_eos.F90.  Only + - * / (no
// transcendentals), written operator-for-operator as in the Fortran so that with
// -ffp-contract=off the results are bit-identical to the reference.
#pragma once
#include "blomgpu_internal.h"

namespace eos {
// phy/mod_eos.F90:36-54
__device__ constexpr double a11 = 9.9985372432159340e+02, a12 = 1.0380621928183473e+01,
                            a13 = 1.7073577195684715e+00, a14 = -3.6570490496333680e-02,
                            a15 = -7.3677944503527477e-03, a16 = -3.5529175999643348e-03,
                            b11 = 1.7083494994335439e-06, b12 = 7.1567921402953455e-09,
                            b13 = 1.2821026080049485e-09, a21 = 1.0, a22 = 1.0316374535350838e-02,
                            a23 = 8.9521792365142522e-04, a24 = -2.8438341552142710e-05,
                            a25 = -1.1887778959461776e-05, a26 = -4.0163964812921489e-06,
                            b21 = 1.1995545126831476e-09, b22 = 5.5234008384648383e-12,
                            b23 = 8.4310335919950873e-13;

// sig(th,s), phy/mod_eos.F90:191-203
__device__ inline double sig(const Params &P, double th, double s) {
  return (P.ap11 + (P.ap12 + P.ap14 * th + P.ap15 * s) * th + (P.ap13 + P.ap16 * s) * s) /
         (P.ap21 + (P.ap22 + P.ap24 * th + P.ap25 * s) * th + (P.ap23 + P.ap26 * s) * s);
}

// rho(p,th,s), phy/mod_eos.F90:157-172
__device__ inline double rho(double p, double th, double s) {
  return (a11 + (a12 + a14 * th + a15 * s) * th + (a13 + a16 * s) * s + (b11 + b12 * th + b13 * s) * p) /
         (a21 + (a22 + a24 * th + a25 * s) * th + (a23 + a26 * s) * s + (b21 + b22 * th + b23 * s) * p);
}

// alp(p,th,s), phy/mod_eos.F90:174-189
__device__ inline double alp(double p, double th, double s) {
  return (a21 + (a22 + a24 * th + a25 * s) * th + (a23 + a26 * s) * s + (b21 + b22 * th + b23 * s) * p) /
         (a11 + (a12 + a14 * th + a15 * s) * th + (a13 + a16 * s) * s + (b11 + b12 * th + b13 * s) * p);
}

// delphi(p1,p2,th,s,dphi,alp1,alp2), phy/mod_eos.F90:478-529
__device__ inline void delphi(double p1, double p2, double th, double s, double &dphi, double &alp1,
                              double &alp2) {
  const double r1_3 = 1. / 3., r1_5 = 1. / 5., r1_7 = 1. / 7., r1_9 = 1. / 9.;
  const double a1 = a11 + (a12 + a14 * th + a15 * s) * th + (a13 + a16 * s) * s;
  const double a2 = a21 + (a22 + a24 * th + a25 * s) * th + (a23 + a26 * s) * s;
  const double b1 = b11 + b12 * th + b13 * s;
  const double b2 = b21 + b22 * th + b23 * s;
  const double pm = .5 * (p2 + p1);
  const double r = .5 * (p2 - p1) / (a1 + b1 * pm);
  const double q = b1 * r;
  const double qq = q * q;
  dphi = -2. * r * (a2 + b2 * pm + (a2 - a1 * b2 / b1) * qq * (r1_3 + qq * (r1_5 + qq * (r1_7 + qq * r1_9))));
  alp1 = (a2 + b2 * p1) / (a1 + b1 * p1);
  alp2 = (a2 + b2 * p2) / (a1 + b1 * p2);
}

// p_alpha(p1,p2,th,s), phy/mod_eos.F90:386-428
__device__ inline double p_alpha(double p1, double p2, double th, double s) {
  const double r1_3 = 1. / 3., r1_5 = 1. / 5., r1_7 = 1. / 7., r1_9 = 1. / 9.;
  const double a1 = a11 + (a12 + a14 * th + a15 * s) * th + (a13 + a16 * s) * s;
  const double a2 = a21 + (a22 + a24 * th + a25 * s) * th + (a23 + a26 * s) * s;
  const double b1 = b11 + b12 * th + b13 * s;
  const double b2 = b21 + b22 * th + b23 * s;
  const double pm = .5 * (p2 + p1);
  const double r = .5 * (p2 - p1) / (a1 + b1 * pm);
  const double q = b1 * r;
  const double qq = q * q;
  return 2. * r * (a2 + b2 * pm + (a2 - a1 * b2 / b1) * qq * (r1_3 + qq * (r1_5 + qq * (r1_7 + qq * r1_9))));
}
}  // namespace eos
