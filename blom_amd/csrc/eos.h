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

// sig0(th,s), phy/mod_eos.F90:205-218: sig with the reference pressure at the surface (coefficients :118-129)
__device__ inline double sig0(double th, double s) {
  constexpr double alpha0 = 1.e-3;
  constexpr double ap110 = a11 - a21 / alpha0, ap120 = a12 - a22 / alpha0, ap130 = a13 - a23 / alpha0,
                   ap140 = a14 - a24 / alpha0, ap150 = a15 - a25 / alpha0, ap160 = a16 - a26 / alpha0;
  return (ap110 + (ap120 + ap140 * th + ap150 * s) * th + (ap130 + ap160 * s) * s) /
         (a21 + (a22 + a24 * th + a25 * s) * th + (a23 + a26 * s) * s);
}

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

// dalpdt(p,th,s), phy/mod_eos.F90:531-552
__device__ inline double dalpdt(double p, double th, double s) {
  const double r1 = a21 + (a22 + a24 * th + a25 * s) * th + (a23 + a26 * s) * s + (b21 + b22 * th + b23 * s) * p;
  const double r2i = 1. / (a11 + (a12 + a14 * th + a15 * s) * th + (a13 + a16 * s) * s + (b11 + b12 * th + b13 * s) * p);
  return (a22 + 2. * a24 * th + a25 * s + b22 * p - (a12 + 2. * a14 * th + a15 * s + b12 * p) * r1 * r2i) * r2i;
}

// dalpds(p,th,s), phy/mod_eos.F90:554-574
__device__ inline double dalpds(double p, double th, double s) {
  const double r1 = a21 + (a22 + a24 * th + a25 * s) * th + (a23 + a26 * s) * s + (b21 + b22 * th + b23 * s) * p;
  const double r2i = 1. / (a11 + (a12 + a14 * th + a15 * s) * th + (a13 + a16 * s) * s + (b11 + b12 * th + b13 * s) * p);
  return (a23 + a25 * th + 2. * a26 * s + b23 * p - (a13 + a15 * th + 2. * a16 * s + b13 * p) * r1 * r2i) * r2i;
}

// dynh_derivatives(p0,p1,p2,th,s,dynh_th,dynh_s), phy/mod_eos.F90:576-695 (truncated series form)
__device__ inline void dynh_derivatives(double p0, double p1, double p2, double th, double s, double &dynh_th, double &dynh_s) {
  const double r1_2 = 1. / 2., r1_3 = 1. / 3., r1_4 = 1. / 4., r1_5 = 1. / 5., r1_6 = 1. / 6., r1_7 = 1. / 7., r1_8 = 1. / 8.,
               r1_9 = 1. / 9., r1_10 = 1. / 10., r1_11 = 1. / 11.;
  const double b1i = 1. / (b11 + b12 * th + b13 * s);
  const double A1 = (a11 + (a12 + a14 * th + a15 * s) * th + (a13 + a16 * s) * s) * b1i;
  const double A2 = (a21 + (a22 + a24 * th + a25 * s) * th + (a23 + a26 * s) * s) * b1i;
  const double B2 = (b21 + b22 * th + b23 * s) * b1i;
  const double a1_th = (a12 + 2. * a14 * th + a15 * s - A1 * b12) * b1i;
  const double a2_th = (a22 + 2. * a24 * th + a25 * s - A2 * b12) * b1i;
  const double b2_th = (b22 - B2 * b12) * b1i;
  const double a1_s = (a13 + a15 * th + 2. * a16 * s - A1 * b13) * b1i;
  const double a2_s = (a23 + a25 * th + 2. * a26 * s - A2 * b13) * b1i;
  const double b2_s = (b23 - B2 * b13) * b1i;
  const double pm1 = r1_2 * (p2 + p1), pp1 = r1_2 * (p2 - p1);
  const double pm0 = r1_2 * (pm1 + p0), pp0 = r1_2 * (pm1 - p0);
  const double t1 = 1. / (A1 + pm1), t0 = 1. / (A1 + pm0);
  const double q1 = pp1 * t1, q0 = pp0 * t0;
  const double qq1 = q1 * q1, qq0 = q0 * q0;
  double f = (A2 - A1 * B2) * a1_th;
  double c1 = a2_th - A1 * b2_th - B2 * a1_th;
  double c2 = f * t1, c3 = f * t0;
  dynh_th = 2. * (pp0 * b2_th +
                  ((((((r1_11 * c1 - c3) * qq0 + (r1_9 * c1 - c3)) * qq0 + (r1_7 * c1 - c3)) * qq0 + (r1_5 * c1 - c3)) * qq0 +
                    (r1_3 * c1 - c3)) * qq0 + (c1 - c3)) * q0) -
            ((((r1_11 * (r1_10 * c1 - c2) * qq1 + r1_9 * (r1_8 * c1 - c2)) * qq1 + r1_7 * (r1_6 * c1 - c2)) * qq1 +
               r1_5 * (r1_4 * c1 - c2)) * qq1 + r1_3 * (r1_2 * c1 - c2)) * qq1;
  f = (A2 - A1 * B2) * a1_s;
  c1 = a2_s - A1 * b2_s - B2 * a1_s;
  c2 = f * t1;
  c3 = f * t0;
  dynh_s = 2. * (pp0 * b2_s +
                 ((((((r1_11 * c1 - c3) * qq0 + (r1_9 * c1 - c3)) * qq0 + (r1_7 * c1 - c3)) * qq0 + (r1_5 * c1 - c3)) * qq0 +
                   (r1_3 * c1 - c3)) * qq0 + (c1 - c3)) * q0) -
           ((((r1_11 * (r1_10 * c1 - c2) * qq1 + r1_9 * (r1_8 * c1 - c2)) * qq1 + r1_7 * (r1_6 * c1 - c2)) * qq1 +
              r1_5 * (r1_4 * c1 - c2)) * qq1 + r1_3 * (r1_2 * c1 - c2)) * qq1;
}
}  // namespace eos
