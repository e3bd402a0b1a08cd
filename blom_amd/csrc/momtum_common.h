// Constants and point functions shared by the momtum kernels (stage_momtum.hip: one kernel per sweep;
// stage_momtum_fused.hip: row-marching fused kernels).  phy/mod_momtum.F90.
#pragma once
#include "blomgpu_internal.h"

#define GRAV 9.806
#define ALPHA0 1.e-3
#define EPSILPL 1.e-14
#define EPSILP 1.e-12
#define ONEM 9806.
#define ONEMM 9.806
#define SLIP (-1.)      // phy/mod_momtum.F90:94
#define THKBOT 10.      // phy/mod_momtum.F90:97
#define WPGF .25        // phy/mod_pgforc.F90:47

__device__ inline double hfharm(double a, double b) { return a * b / (a + b); }   // :131-141

// ---- :662-715 min,max transports for the energy conserving scheme with dissipation (mommth = 'enedis') --
__device__ inline void enedis_minmax(double hc, double hm, double &hmin, double &hmax) {
  const double c1 = 1. - 1.5 * .5, c2 = 1. - .5, c3 = 2., slope = .5;                 // :221
  if (fabs(hc) < .1 * fabs(hm)) hm = 10. * hc;
  else if (fabs(hc) > c1 * fabs(hm)) {
    if (fabs(hc) < c2 * fabs(hm)) hc = (3. * hc + (1. - c2 * 3.) * hm);
    else if (fabs(hc) <= c3 * fabs(hm)) hc = hm;
    else hc = slope * hc + (1. - c3 * slope) * hm;
  }
  if (hc > hm) { hmin = hm; hmax = hc; }
  else { hmax = hm; hmin = hc; }
}

// viscosity at x along a row, extended one point beyond wet u-segments (:845-856): a land point
// takes the value of the u-point to its right (segment start, written last) else of the one to its left
__device__ inline double ext_i(const int *msk, const double *f, size_t x) {
  return msk[x] ? f[x] : (msk[x + 1] ? f[x + 1] : f[x - 1]);
}
__device__ inline double ext_j(const int *msk, const double *f, size_t x, int ni) {
  return msk[x] ? f[x] : (msk[x + ni] ? f[x + ni] : f[x - ni]);
}

