// sin_libm.h -- double precision sin() with the bits of the host libm's.
//
// The reference's compiled Fortran evaluates `sin(atan2(v, u) - hangle)**10` through glibc's libm when rhsctp is set
// (phy/mod_difest.F90:2320-2340, :1738-1760: the alignment of the flow with the topographic beta).  The device math library's sin
// agrees with it to an ulp, not to the bit, and this library compares with `==`.  glibc's sin (sysdeps/ieee754/dbl-64/s_sin.c, the IBM
// Accurate Mathematical Library routine, no multi-precision fallback since 2.28; < 0.55 ulp, NOT correctly rounded, so a correctly
// rounded sin would differ on a small fraction of arguments) is restated here with the fused multiply-adds exactly where the x86-64
// FMA build has them (read off `objdump -d libm.so.6`, Ubuntu GLIBC 2.35, the variant its resolver picks with FMA + AVX2):
//   |x| < 2^-26            x
//   |x| < 0.855469         do_sin(x, 0)
//   |x| < 2.426265         copysign(do_cos(hp0 - |x|, hp1), x)
//   |x| < 105414350        n = reduce_sincos(x, a, da): t = fma(x, 2/pi, toint), xn = t - toint, n = low bits of t,
//                          y = fma(-xn, mp2, fma(-xn, mp1, x)), t2 = fma(-xn, pp3, y), db = fma(-pp3, xn, y - t2),
//                          a = fma(-xn, pp4, t2), da = db + fma(-xn, pp4, t2 - a);
//                          result = (n & 1 ? do_cos(a, da) : do_sin(a, da)), negated when n & 2
//   do_sin(x, dx): |x| < 0.126: t = fma(fma(poly(xx), x, -0.5 dx), xx, dx), x + t with poly by Horner in fma;
//                  else u = big + |x| (its low mantissa bits k = round(128 |x|) index the table {sn, ssn, cs, ccs} of sin / cos(k/128)),
//                  xr = |x| - (u - big), s = xr + fma(xr xx, fma(xx, sn5, sn3), dx), c = fma(xr, dx, xx cspoly(xx)),
//                  cor = fma(s, cs, fma(-c, sn, fma(s, ccs, ssn))), copysign(sn + cor, x)            (dx negated for x <= 0)
//   do_cos(x, dx): xr = (|x| - (u - big)) + dx, s = fma(xr xx, fma(xx, sn5, sn3), xr), c = xx cspoly(xx),
//                  cor = fma(-s, sn, fma(-c, cs, fma(-s, ssn, ccs))), cs + cor                         (dx negated for x < 0)
// Arguments of 105414350 and beyond (glibc's __branred) and non-finite ones return NaN here: the model's argument is an angle of a
// few radians.  The table: tools/gen_sincos_table.py (first principles + the 18 low words where glibc's published table is not the
// correctly rounded one; checked against the table inside this machine's libm).  tests/test_sin_atan2_libm.py compares the host build
// of this very function, and the device through blomgpu_sin, with the host's sin() bit for bit.
#pragma once
#include "exp_libm.h"
#include "sin_libm_table.h"

#if defined(__HIPCC__)
static __device__ const uint64_t sinl_tab_dev[440] = {SIN_LIBM_TABLE};
#endif
static const uint64_t sinl_tab_host[440] = {SIN_LIBM_TABLE};

EXPL_HD static inline double sinl_tab(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
  return expl_from_bits(sinl_tab_dev[i]);
#else
  return expl_from_bits(sinl_tab_host[i]);
#endif
}

#define SINL_BIG 0x1.8p45
#define SINL_SN3 -0x1.5555555555515p-3
#define SINL_SN5 0x1.11110e829872fp-7
#define SINL_CS2 0.5
#define SINL_CS4 -0x1.5555555555535p-5
#define SINL_CS6 0x1.6c16bedd9e239p-10
#define SINL_S1 -0x1.5555555555555p-3
#define SINL_S2 0x1.1111111110ecep-7
#define SINL_S3 -0x1.a01a019db08b8p-13
#define SINL_S4 0x1.71de27b9a7ed9p-19
#define SINL_S5 -0x1.addffc2fcdf59p-26

EXPL_HD static inline double sinl_taylor(double x, double dx) {
  const double xx = x * x;
  double p = __builtin_fma(xx, SINL_S5, SINL_S4);
  p = __builtin_fma(xx, p, SINL_S3);
  p = __builtin_fma(xx, p, SINL_S2);
  p = __builtin_fma(xx, p, SINL_S1);
  const double t = __builtin_fma(__builtin_fma(p, x, -(0.5 * dx)), xx, dx);
  return x + t;
}

// the table entry of |x| and the remainder: u = big + |x|, k = 4 round(128 |x|)
EXPL_HD static inline int sinl_split(double ax, double *xr) {
  const double u = SINL_BIG + ax;
  *xr = ax - (u - SINL_BIG);
  return (int)(uint32_t)expl_to_bits(u) << 2;
}

EXPL_HD static inline double sinl_do_sin(double x, double dx) {
  const double ax = __builtin_fabs(x);
  if (ax < 0.126) return sinl_taylor(x, dx);
  if (x <= 0.) dx = -dx;
  double xr;
  const int k = sinl_split(ax, &xr);
  const double xx = xr * xr;
  const double s = xr + __builtin_fma(xr * xx, __builtin_fma(xx, SINL_SN5, SINL_SN3), dx);
  const double c = __builtin_fma(xr, dx, xx * __builtin_fma(xx, __builtin_fma(xx, SINL_CS6, SINL_CS4), SINL_CS2));
  const double sn = sinl_tab(k), ssn = sinl_tab(k + 1), cs = sinl_tab(k + 2), ccs = sinl_tab(k + 3);
  const double cor = __builtin_fma(s, cs, __builtin_fma(-c, sn, __builtin_fma(s, ccs, ssn)));
  return __builtin_copysign(sn + cor, x);
}

EXPL_HD static inline double sinl_do_cos(double x, double dx) {
  if (x < 0.) dx = -dx;
  double xr;
  const int k = sinl_split(__builtin_fabs(x), &xr);
  xr = xr + dx;
  const double xx = xr * xr;
  const double s = __builtin_fma(xr * xx, __builtin_fma(xx, SINL_SN5, SINL_SN3), xr);
  const double c = xx * __builtin_fma(xx, __builtin_fma(xx, SINL_CS6, SINL_CS4), SINL_CS2);
  const double sn = sinl_tab(k), ssn = sinl_tab(k + 1), cs = sinl_tab(k + 2), ccs = sinl_tab(k + 3);
  const double cor = __builtin_fma(-s, sn, __builtin_fma(-c, cs, __builtin_fma(-s, ssn, ccs)));
  return cs + cor;
}

EXPL_HD static inline double sin_libm(double x) {
  const uint32_t k = (uint32_t)(expl_to_bits(x) >> 32) & 0x7fffffffu;
  if (k < 0x3e500000u) return x;                                     // |x| < 2^-26
  if (k < 0x3feb6000u) return sinl_do_sin(x, 0.);                    // |x| < 0.855469
  if (k < 0x400368fdu) {                                             // |x| < 2.426265
    const double t = 0x1.921fb54442d18p+0 - __builtin_fabs(x);       // hp0 - |x|
    return __builtin_copysign(sinl_do_cos(t, 0x1.1a62633145c07p-54), x);
  }
  if (k < 0x419921fbu) {                                             // |x| < 105414350
    const double toint = 0x1.8p52;
    const double t = __builtin_fma(x, 0x1.45f306dc9c883p-1, toint);
    const double xn = t - toint;
    const uint32_t n = (uint32_t)expl_to_bits(t);
    const double y = __builtin_fma(-xn, -0x1.dde973c000000p-27, __builtin_fma(-xn, 0x1.921fb58000000p+0, x));
    const double pp3 = -0x1.cb3b398000000p-55, pp4 = -0x1.d747f23e32ed7p-83;
    const double t2 = __builtin_fma(-xn, pp3, y);
    double db = __builtin_fma(-pp3, xn, y - t2);
    const double a = __builtin_fma(-xn, pp4, t2);
    db = db + __builtin_fma(-xn, pp4, t2 - a);
    const double r = (n & 1u) ? sinl_do_cos(a, db) : sinl_do_sin(a, db);
    return (n & 2u) ? -r : r;
  }
  return expl_from_bits(0x7ff8000000000000ull);                      // (__branred's range and non-finite arguments: not restated)
}
