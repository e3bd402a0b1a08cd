// Shared device helpers of the diapfl kernels (stage_diapfl.hip, stage_diapfl_col3.hip).
#pragma once
#include "blomgpu_internal.h"
#include "eos.h"

#define DIAPFL_GRAV 9.806
#define DIAPFL_ALPHA0 1.e-3
#define DIAPFL_EPSILP 1.e-12
#define DIAPFL_ONEM 9806.

namespace eosd {
// dsigdt / dsigds, phy/mod_eos.F90:243-261, :306-323
__device__ inline double dsigdt(const Params &P, double th, double s) {
  const double r1 = P.ap11 + (P.ap12 + P.ap14 * th + P.ap15 * s) * th + (P.ap13 + P.ap16 * s) * s;
  const double r2i = 1. / (P.ap21 + (P.ap22 + P.ap24 * th + P.ap25 * s) * th + (P.ap23 + P.ap26 * s) * s);
  return (P.ap12 + 2. * P.ap14 * th + P.ap15 * s - (P.ap22 + 2. * P.ap24 * th + P.ap25 * s) * r1 * r2i) * r2i;
}
__device__ inline double dsigds(const Params &P, double th, double s) {
  const double r1 = P.ap11 + (P.ap12 + P.ap14 * th + P.ap15 * s) * th + (P.ap13 + P.ap16 * s) * s;
  const double r2i = 1. / (P.ap21 + (P.ap22 + P.ap24 * th + P.ap25 * s) * th + (P.ap23 + P.ap26 * s) * s);
  return (P.ap13 + P.ap15 * th + 2. * P.ap16 * s - (P.ap23 + P.ap25 * th + 2. * P.ap26 * s) * r1 * r2i) * r2i;
}
// sofsig, phy/mod_eos.F90:366-384
__device__ inline double sofsig(const Params &P, double sg, double th) {
  const double a = P.ap16 - P.ap26 * sg;
  const double b = P.ap13 - P.ap23 * sg + (P.ap15 - P.ap25 * sg) * th;
  const double cc = P.ap11 - P.ap21 * sg + (P.ap12 - P.ap22 * sg + (P.ap14 - P.ap24 * sg) * th) * th;
  return (-b + sqrt(b * b - 4. * a * cc)) / (2. * a);
}
}  // namespace eosd

// flux of the backward solution and its sensitivity, :382-406 == :457-481
__device__ inline void flux_solution(double q, double r, double t, double &f0, double &dfdg) {
  if (q < 0.) {
    double s = r / (q * q);
    if (s < 1.e-3) {
      r = .00390625 * s;
      q = -q * r * (128. - s * (32. - s * (16. - s * (10. - s * 7.))));
      f0 = q * t;
      q = r * (128. - s * (96. - s * (80. - s * (70. - s * 63.))));
      dfdg = q * t;
    } else {
      s = sqrt(q * q + r);
      f0 = (q + s) * t;
      dfdg = (1. + q / s) * t;
    }
  } else {
    const double s = sqrt(q * q + r);
    f0 = (q + s) * t;
    dfdg = (1. + q / s) * t;
  }
}

