// atan2_libm.h -- double precision atan2() with the bits of the host libm's.
//
// With rhsctp the reference evaluates `atan2(vbc + vbt, ubc + ubt)` through glibc's libm (phy/mod_difest.F90:2331, :1751).  As for sin
// (sin_libm.h) the device math library's result agrees to an ulp, not to the bit.  glibc's atan2 (sysdeps/ieee754/dbl-64/e_atan2.c, the
// IBM Accurate Mathematical Library routine without its multi-precision fallback since 2.35) is restated here with the fused
// multiply-adds exactly where the x86-64 FMA build has them (read off `objdump -d libm.so.6`, Ubuntu GLIBC 2.35):
//   u = min(|x|,|y|) / max(|x|,|y|) with its error du = ((num - v) - fma(den, u, -v)) / den, v = den u        (both scaled by 2^+-500 when extreme)
//   u < 1/16:  zz = u v poly(v) (+ du), v = u u, poly = d3 + v (d5 + v (d7 + v (d9 + v (d11 + v d13)))) by Horner in fma
//   else    :  i = (int)(fma(u, 256, 2^52) - 2^52) - 16 picks the sample point cij[i][0] of the table (atan2_libm_table.h), v = (u - cij[i][0]) + du,
//              a degree-4 polynomial in v with cij[i][2..6] around atan(cij[i][0]) = cij[i][1]
//   (i)   x > 0, |y| <  |x|:  atan(u)               z = u + fma(u v, poly, du)                     |  t1 + fma(v, t2, fma(dv, t2, v v p3(v)))   [v, dv = EADD(t3, du)]
//   (ii)  x > 0, |x| <= |y|:  pi/2 - atan(u)        t2 + (((hpi1 + cor) - du) - zz), ESUB(hpi, u)  |  (hpi - cij[i][1]) + fma(-v, p2(v), hpi1)
//   (iii) x < 0, |x| <  |y|:  pi/2 + atan(u)        t2 + (((hpi1 + cor) + du) + zz), EADD(hpi, u)  |  (hpi + cij[i][1]) + fma(v, p2(v), hpi1)
//   (iv)  x < 0, |y| <= |x|:  pi - atan(u)          t2 + (((opi1 + cor) - du) - zz), ESUB(opi, u)  |  (opi - cij[i][1]) + fma(-v, p2(v), opi1)
//   result copysign(z, y); the exponent difference of y and x beyond +-57 and the zero / infinity / NaN cases as in the source.
// tests/test_sin_atan2_libm.py compares the host build of this very function, and the device through blomgpu_atan2, with the host's
// atan2() bit for bit (> 500 000 pairs: the four quadrants, the octant edges, tiny and huge ratios, zeros, infinities, NaN).
#pragma once
#include "exp_libm.h"
#include "atan2_libm_table.h"

#if defined(__HIPCC__)
static __device__ const uint64_t atl_tab_dev[241 * 7] = {ATAN2_LIBM_TABLE};
#endif
static const uint64_t atl_tab_host[241 * 7] = {ATAN2_LIBM_TABLE};

EXPL_HD static inline double atl_cij(int i, int j) {
#if defined(__HIP_DEVICE_COMPILE__)
  return expl_from_bits(atl_tab_dev[7 * i + j]);
#else
  return expl_from_bits(atl_tab_host[7 * i + j]);
#endif
}

#define ATL_HPI 0x1.921fb54442d18p+0
#define ATL_HPI1 0x1.1a62633145c07p-54
#define ATL_OPI 0x1.921fb54442d18p+1
#define ATL_OPI1 0x1.1a62633145c07p-53
#define ATL_QPI 0x1.921fb54442d18p-1
#define ATL_TQPI 0x1.2d97c7f3321d2p+1

// d3 + v (d5 + v (d7 + v (d9 + v (d11 + v d13))))
EXPL_HD static inline double atl_poly(double v) {
  double p = __builtin_fma(v, 0x1.375f08b31cbcep-4, -0x1.7458022b13c25p-4);
  p = __builtin_fma(v, p, 0x1.c71c6e5129a3bp-4);
  p = __builtin_fma(v, p, -0x1.24924923f7603p-3);
  p = __builtin_fma(v, p, 0x1.99999999997fdp-3);
  return __builtin_fma(v, p, -0x1.5555555555555p-2);
}
EXPL_HD static inline int atl_index(double u) {
  const double two52 = 0x1p52;
  return (int)(__builtin_fma(u, 256., two52) - two52) - 16;
}
// c2 + v (c3 + v (c4 + v (c5 + v c6)))
EXPL_HD static inline double atl_p2(int i, double v) {
  double p = __builtin_fma(v, atl_cij(i, 6), atl_cij(i, 5));
  p = __builtin_fma(v, p, atl_cij(i, 4));
  p = __builtin_fma(v, p, atl_cij(i, 3));
  return __builtin_fma(v, p, atl_cij(i, 2));
}

EXPL_HD static inline double atan2_libm(double y, double x) {
  const uint64_t bx = expl_to_bits(x), by = expl_to_bits(y);
  const uint32_t ux = (uint32_t)(bx >> 32), dx = (uint32_t)bx, uy = (uint32_t)(by >> 32), dy = (uint32_t)by;
  // NaN
  if ((ux & 0x7ff00000u) == 0x7ff00000u && ((ux & 0x000fffffu) | dx) != 0u) return x + y;
  if ((uy & 0x7ff00000u) == 0x7ff00000u && ((uy & 0x000fffffu) | dy) != 0u) return y + y;
  // y = +-0
  if (uy == 0x00000000u) { if (dy == 0u) return (ux & 0x80000000u) == 0u ? 0. : ATL_OPI; }
  else if (uy == 0x80000000u) { if (dy == 0u) return (ux & 0x80000000u) == 0u ? -0. : -ATL_OPI; }
  // x = +-0
  if (x == 0.) return (uy & 0x80000000u) == 0u ? ATL_HPI : -ATL_HPI;
  // x = +-inf
  if (ux == 0x7ff00000u) {
    if (dx == 0u) {
      if (uy == 0x7ff00000u) { if (dy == 0u) return ATL_QPI; }
      else if (uy == 0xfff00000u) { if (dy == 0u) return -ATL_QPI; }
      else return (uy & 0x80000000u) == 0u ? 0. : -0.;
    }
  } else if (ux == 0xfff00000u) {
    if (dx == 0u) {
      if (uy == 0x7ff00000u) { if (dy == 0u) return ATL_TQPI; }
      else if (uy == 0xfff00000u) { if (dy == 0u) return -ATL_TQPI; }
      else return (uy & 0x80000000u) == 0u ? ATL_OPI : -ATL_OPI;
    }
  }
  // y = +-inf
  if (uy == 0x7ff00000u) { if (dy == 0u) return ATL_HPI; }
  else if (uy == 0xfff00000u) { if (dy == 0u) return -ATL_HPI; }

  double ax = x < 0. ? -x : x, ay = y < 0. ? -y : y;
  const int de = (int)(uy & 0x7ff00000u) - (int)(ux & 0x7ff00000u);
  if (de >= 0x3900000) return y > 0. ? ATL_HPI : -ATL_HPI;
  if (de <= -0x3900000) {
    if (x > 0.) return __builtin_copysign(ay / ax, y);
    return y > 0. ? ATL_OPI : -ATL_OPI;
  }
  if (ax < 0x1p-500 || ay < 0x1p-500) { ax *= 0x1p500; ay *= 0x1p500; }
  if (ax > 0x1p500 || ay > 0x1p500) { ax *= 0x1p-500; ay *= 0x1p-500; }

  double u, du;
  const int ylt = ay < ax;
  if (ylt) {
    u = ay / ax;
    const double v = ax * u, vv = __builtin_fma(ax, u, -v);
    du = ((ay - v) - vv) / ax;
  } else {
    u = ax / ay;
    const double v = ay * u, vv = __builtin_fma(ay, u, -v);
    du = ((ax - v) - vv) / ay;
  }
  const int small = u < 0.0625;
  double z;
  if (x > 0.) {
    if (ylt) {                                                    // (i)
      if (small) {
        const double v = u * u;
        z = u + __builtin_fma(u * v, atl_poly(v), du);
      } else {
        const int i = atl_index(u);
        const double t3 = u - atl_cij(i, 0);
        const double v = t3 + du;
        const double dv = __builtin_fabs(t3) > __builtin_fabs(du) ? (t3 - v) + du : (du - v) + t3;
        const double t2 = atl_cij(i, 2);
        double p = __builtin_fma(v, atl_cij(i, 6), atl_cij(i, 5));
        p = __builtin_fma(v, p, atl_cij(i, 4));
        p = __builtin_fma(v, p, atl_cij(i, 3));
        const double zz = __builtin_fma(v, t2, __builtin_fma(dv, t2, (v * v) * p));
        z = zz + atl_cij(i, 1);
      }
    } else {                                                      // (ii)
      if (small) {
        const double v = u * u;
        const double zz = (u * v) * atl_poly(v);
        const double t2 = ATL_HPI - u;
        const double cor = ATL_HPI > __builtin_fabs(u) ? (ATL_HPI - t2) - u : ATL_HPI - (u + t2);
        const double t3 = ((cor + ATL_HPI1) - du) - zz;
        z = t3 + t2;
      } else {
        const int i = atl_index(u);
        const double v = (u - atl_cij(i, 0)) + du;
        const double zz = __builtin_fma(-v, atl_p2(i, v), ATL_HPI1);
        z = (ATL_HPI - atl_cij(i, 1)) + zz;
      }
    }
  } else if (ax < ay) {                                           // (iii)
    if (small) {
      const double v = u * u;
      const double zz = (v * u) * atl_poly(v);
      const double t2 = u + ATL_HPI;
      const double cor = ATL_HPI > __builtin_fabs(u) ? (ATL_HPI - t2) + u : (u - t2) + ATL_HPI;
      const double t3 = ((cor + ATL_HPI1) + du) + zz;
      z = t3 + t2;
    } else {
      const int i = atl_index(u);
      const double v = (u - atl_cij(i, 0)) + du;
      const double zz = __builtin_fma(v, atl_p2(i, v), ATL_HPI1);
      z = (ATL_HPI + atl_cij(i, 1)) + zz;
    }
  } else {                                                        // (iv)
    if (small) {
      const double v = u * u;
      const double zz = (v * u) * atl_poly(v);
      const double t2 = ATL_OPI - u;
      const double cor = ATL_OPI > __builtin_fabs(u) ? (ATL_OPI - t2) - u : ATL_OPI - (t2 + u);
      const double t3 = ((cor + ATL_OPI1) - du) - zz;
      z = t3 + t2;
    } else {
      const int i = atl_index(u);
      const double v = (u - atl_cij(i, 0)) + du;
      const double zz = __builtin_fma(-v, atl_p2(i, v), ATL_OPI1);
      z = (ATL_OPI - atl_cij(i, 1)) + zz;
    }
  }
  return __builtin_copysign(z, y);
}
