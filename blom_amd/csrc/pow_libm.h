// pow_libm.h -- double precision pow() with the bits of the host libm's.
//
// The reference's compiled Fortran evaluates real powers through pow() of glibc's libm (phy/mod_difest.F90:2881-2886, :2989-2994:
// the TKE closure's length scales; :3056-3058: the stability function of the surface layer).  As with exp (exp_libm.h), the device
// math library's pow agrees with it to an ulp, not to the bit, and this library compares with `==`.  So pow is evaluated the way
// glibc >= 2.28 does on x86-64 with FMA (its ifunc choice on every AVX2 host): sysdeps/ieee754/dbl-64/e_pow.c --
//     log(x) as k ln2 + log(c) + log1p(z/c - 1), z/c - 1 exact thanks to the 8-bit 1/c of a 128-entry table, a degree-7 polynomial,
//     the result as a double-double (hi, lo);  exp(y hi + y lo) with the exp algorithm of exp_libm.h carrying the low part
// with the fused multiply-adds exactly where the compiled libm has them (read off `objdump -d libm.so.6`, Ubuntu GLIBC 2.35, the
// variant its resolver picks with FMA + AVX2):
//     r   = fma(z, invc, -1);  t1 = fma(kd, Ln2hi, logc);  lo1 = fma(kd, Ln2lo, logctail)
//     ar = A0 r; ar2 = r ar; ar3 = r ar2;  lo3 = fma(ar, r, -ar2)
//     p   = fma(ar2, fma(ar2, fma(r, A6, A5), fma(r, A4, A3)), fma(r, A2, A1))
//     lo  = fma(p, ar3, ((lo1 + lo2) + lo3) + lo4)
//     ehi = y hi;  elo = fma(y, lo, fma(hi, y, -ehi))
//     exp: kd = fma(ehi, InvLn2N, Shift); r = fma(kd, NegLn2loN, fma(kd, NegLn2hiN, ehi)) + elo
//          tmp = fma(fma(r, C5, C4), r2 r2, fma(fma(r, C3, C2), r2, tail + r));  result = fma(tmp, scale, scale)
// The log table is regenerated from the construction its authors publish (tools/gen_pow_log_table.py, 80-digit logarithms, checked
// against the table inside this machine's libm); the 2^(k/128) table is exp_libm.h's.  tests/test_pow_libm.py compares the host build
// of this very function, and the device through blomgpu_pow, with the host's pow() bit for bit.
#pragma once
#include "exp_libm.h"
#include "pow_libm_table.h"

#if defined(__HIPCC__)
static __device__ const uint64_t powl_tab_dev[384] = {POW_LIBM_LOG_TABLE};
#endif
static const uint64_t powl_tab_host[384] = {POW_LIBM_LOG_TABLE};

// 0: y is not an integer, 1: an odd integer, 2: an even integer (e_pow.c: checkint)
EXPL_HD static inline int powl_checkint(uint64_t iy) {
  const int e = (int)(iy >> 52) & 0x7ff;
  if (e < 0x3ff) return 0;
  if (e > 0x3ff + 52) return 2;
  if (iy & ((1ull << (0x3ff + 52 - e)) - 1)) return 0;
  if (iy & (1ull << (0x3ff + 52 - e))) return 1;
  return 2;
}
EXPL_HD static inline int powl_zeroinfnan(uint64_t i) { return 2 * i - 1 >= 2 * 0x7ff0000000000000ull - 1; }

// exp(x + xtail) with the sign of the result in sign_bias (e_pow.c: exp_inline, specialcase)
EXPL_HD static inline double powl_exp(double x, double xtail, uint32_t sign_bias) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint64_t *tab = expl_tab_dev;
#else
  const uint64_t *tab = expl_tab_host;
#endif
  const double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p52;
  const double NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
  const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
  const uint64_t ix = expl_to_bits(x);
  unsigned abstop = (unsigned)(ix >> 52) & 0x7ff;
  if (abstop - 0x3c9u > 0x3eu) {
    if ((int)(abstop - 0x3c9u) < 0) {                 // tiny
      const double one = 1.0 + x;
      return sign_bias ? -one : one;
    }
    if (abstop >= 0x409) {                            // |x| >= 1024 (inf and nan have been dealt with)
      if (ix >> 63) return sign_bias ? -0.0 : 0.0;    // underflow
      const double inf = expl_from_bits(0x7ff0000000000000ull);
      return sign_bias ? -inf : inf;                  // overflow
    }
    abstop = 0;
  }
  double kd = __builtin_fma(x, InvLn2N, Shift);
  const uint64_t ki = expl_to_bits(kd);
  kd = kd - Shift;
  double r = __builtin_fma(kd, NegLn2loN, __builtin_fma(kd, NegLn2hiN, x));
  r = xtail + r;
  const unsigned idx = 2 * (unsigned)(ki % 128);
  const uint64_t top = (ki + sign_bias) << 45;
  const double tail = expl_from_bits(tab[idx]);
  uint64_t sbits = tab[idx + 1] + top;
  const double r2 = r * r;
  const double tmp = __builtin_fma(__builtin_fma(r, C5, C4), r2 * r2, __builtin_fma(__builtin_fma(r, C3, C2), r2, r + tail));
  if (abstop == 0) {                                  // specialcase()
    if ((ki & 0x80000000ull) == 0) {
      sbits -= 1009ull << 52;
      const double scale = expl_from_bits(sbits);
      return 0x1p1009 * __builtin_fma(scale, tmp, scale);
    }
    sbits += 1022ull << 52;
    const double scale = expl_from_bits(sbits);
    const double st = tmp * scale;
    double y = scale + st;
    if (__builtin_fabs(y) < 1.0) {
      const double one = y < 0.0 ? -1.0 : 1.0;
      double lo = scale - y + st;
      const double hi = y + one;
      lo = one - hi + y + lo;
      y = (lo + hi) - one;
      if (y == 0.0) y = expl_from_bits(sbits & 0x8000000000000000ull);
    }
    return 0x1p-1022 * y;
  }
  const double scale = expl_from_bits(sbits);
  return __builtin_fma(tmp, scale, scale);
}

EXPL_HD static inline double pow_libm(double x, double y) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint64_t *T = powl_tab_dev;
#else
  const uint64_t *T = powl_tab_host;
#endif
  const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
  const double A0 = -0x1p-1, A1 = -0x1.555555555556p-1, A2 = 0x1.0000000000006p-1, A3 = 0x1.999999959554ep-1,
               A4 = -0x1.555555529a47ap-1, A5 = -0x1.2495b9b4845e9p+0, A6 = 0x1.0002b8b263fc3p+0;
  uint32_t sign_bias = 0;
  uint64_t ix = expl_to_bits(x);
  const uint64_t iy = expl_to_bits(y);
  uint32_t topx = (uint32_t)(ix >> 52);
  const uint32_t topy = (uint32_t)(iy >> 52);
  if (topx - 0x001u >= 0x7ffu - 0x001u || (topy & 0x7ff) - 0x3beu >= 0x43eu - 0x3beu) {
    // x < 0x1p-1022 or inf or nan, or |y| < 0x1p-65 or |y| >= 0x1p63 or nan
    const uint64_t one = 0x3ff0000000000000ull, inf = 0x7ff0000000000000ull;
    if (powl_zeroinfnan(iy)) {
      if (2 * iy == 0) return 1.0;
      if (ix == one) return 1.0;
      if (2 * ix > 2 * inf || 2 * iy > 2 * inf) return x + y;
      if (2 * ix == 2 * one) return 1.0;
      if ((2 * ix < 2 * one) == !(iy >> 63)) return 0.0;     // |x| < 1 and y = inf, or |x| > 1 and y = -inf
      return y * y;
    }
    if (powl_zeroinfnan(ix)) {
      double x2 = x * x;
      if (ix >> 63 && powl_checkint(iy) == 1) { x2 = -x2; sign_bias = 1; }
      if (2 * ix == 0 && iy >> 63) return sign_bias ? -expl_from_bits(inf) : expl_from_bits(inf);
      return iy >> 63 ? 1 / x2 : x2;
    }
    if (ix >> 63) {                                           // finite x < 0
      const int yint = powl_checkint(iy);
      if (yint == 0) return (x - x) / (x - x);
      if (yint == 1) sign_bias = 0x800u << 7;
      ix &= 0x7fffffffffffffffull;
      topx &= 0x7ff;
    }
    if ((topy & 0x7ff) - 0x3beu >= 0x43eu - 0x3beu) {
      if (ix == one) return 1.0;
      if ((topy & 0x7ff) < 0x3be) return ix > one ? 1.0 + y : 1.0 - y;   // |y| < 2^-65
      return (ix > one) == (topy < 0x800) ? expl_from_bits(inf) : 0.0;   // overflow : underflow
    }
    if (topx == 0) {                                          // subnormal x: normalise
      ix = expl_to_bits(x * 0x1p52);
      ix &= 0x7fffffffffffffffull;
      ix -= 52ull << 52;
    }
  }
  // log_inline
  const uint64_t tmp = ix - 0x3fe6955500000000ull;
  const int i = (int)((tmp >> 45) % 128);
  const int k = (int)((int64_t)tmp >> 52);
  const uint64_t iz = ix - (tmp & 0xfffull << 52);
  const double z = expl_from_bits(iz);
  const double kd = (double)k;
  const double invc = expl_from_bits(T[3 * i]), logc = expl_from_bits(T[3 * i + 1]), logctail = expl_from_bits(T[3 * i + 2]);
  const double r = __builtin_fma(z, invc, -1.0);
  const double t1 = __builtin_fma(kd, Ln2hi, logc);
  const double t2 = r + t1;
  const double lo1 = __builtin_fma(kd, Ln2lo, logctail);
  const double lo2 = t1 - t2 + r;
  const double ar = r * A0;
  const double ar2 = r * ar;
  const double ar3 = r * ar2;
  const double hi = t2 + ar2;
  const double lo3 = __builtin_fma(ar, r, -ar2);
  const double lo4 = t2 - hi + ar2;
  const double p = __builtin_fma(ar2, __builtin_fma(__builtin_fma(r, A6, A5), ar2, __builtin_fma(r, A4, A3)), __builtin_fma(r, A2, A1));
  const double lo = __builtin_fma(p, ar3, lo1 + lo2 + lo3 + lo4);
  const double lhi = hi + lo;
  const double llo = hi - lhi + lo;
  const double ehi = y * lhi;
  const double elo = __builtin_fma(y, llo, __builtin_fma(lhi, y, -ehi));
  return powl_exp(ehi, elo, sign_bias);
}
