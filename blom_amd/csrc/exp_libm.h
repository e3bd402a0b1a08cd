// exp_libm.h -- double precision exp() with the bits of the host libm's.
//
// The reference's compiled Fortran calls exp() of glibc's libm (phy/mod_barotp.F90:183,205 coastal damping,
// phy/mod_diapfl.F90:204 bottom boundary layer term).  The device math library's exp agrees with it to an ulp, not
// to the bit, and one ulp in diapfl's mass flux becomes one ulp in p, which flips `p < pu(kk+1)` comparisons
// (phy/mod_diapfl.F90:845) and with them whole velocity columns of massless layers.  So the kernels evaluate exp
// the way glibc >= 2.28 does on x86-64 with FMA (its ifunc choice on every AVX2 host): the table-driven algorithm
// of sysdeps/ieee754/dbl-64/e_exp.c (N = 128, degree-5 polynomial), with the fused multiply-adds exactly where the
// compiled libm has them (read off `objdump -d libm.so.6`, Ubuntu GLIBC 2.35):
//     kd  = fma(x, InvLn2N, Shift);  ki = bits(kd);  kd -= Shift
//     r   = fma(kd, NegLn2loN, fma(kd, NegLn2hiN, x))
//     tmp = fma(r2 * r2, fma(r, C5, C4), fma(fma(r, C3, C2), r2, tail + r))
//     exp = fma(scale, tmp, scale)
// The table is regenerated from first principles by tools/gen_exp_table.py; tests/test_exp_libm.py compares the
// host build of this very function, and the device through blomgpu_exp, with the host's exp() bit for bit.
#pragma once
#include <stdint.h>
#include <string.h>
#include "exp_libm_table.h"

#if defined(__HIPCC__)
#define EXPL_HD __host__ __device__
static __device__ const uint64_t expl_tab_dev[256] = {EXP_LIBM_TABLE};
#else
#define EXPL_HD
#endif
static const uint64_t expl_tab_host[256] = {EXP_LIBM_TABLE};

EXPL_HD static inline double expl_from_bits(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
EXPL_HD static inline uint64_t expl_to_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

EXPL_HD static inline double exp_libm(double x) {
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
  if (abstop - 0x3c9u > 0x3eu) {                    // |x| < 2^-54 or |x| >= 512
    if ((int)(abstop - 0x3c9u) < 0) return 1.0 + x; // tiny
    if (abstop >= 0x409) {                          // |x| >= 1024, inf, nan
      if (ix == 0xfff0000000000000ull) return 0.0;
      if (abstop >= 0x7ff) return 1.0 + x;
      return (ix >> 63) ? 0.0 : expl_from_bits(0x7ff0000000000000ull);      // underflow : overflow
    }
    abstop = 0;                                     // 512 <= |x| < 1024: the scale needs care below
  }
  double kd = __builtin_fma(x, InvLn2N, Shift);
  const uint64_t ki = expl_to_bits(kd);
  kd = kd - Shift;
  const double r = __builtin_fma(kd, NegLn2loN, __builtin_fma(kd, NegLn2hiN, x));
  const unsigned idx = 2 * (unsigned)(ki % 128);
  const uint64_t top = ki << 45;
  const double tail = expl_from_bits(tab[idx]);
  uint64_t sbits = tab[idx + 1] + top;
  const double r2 = r * r;
  const double tmp = __builtin_fma(r2 * r2, __builtin_fma(r, C5, C4), __builtin_fma(__builtin_fma(r, C3, C2), r2, tail + r));
  if (abstop == 0) {                                // specialcase()
    if ((ki & 0x80000000ull) == 0) {                // k > 0: the exponent of scale may have overflowed
      sbits -= 1009ull << 52;
      const double scale = expl_from_bits(sbits);
      return 0x1p1009 * __builtin_fma(scale, tmp, scale);
    }
    sbits += 1022ull << 52;                         // k < 0: subnormal range
    const double scale = expl_from_bits(sbits);
    const double st = scale * tmp;
    double y = scale + st;
    if (y < 1.0) {
      double lo = scale - y + st;
      const double hi = 1.0 + y;
      lo = 1.0 - hi + y + lo;
      y = (hi + lo) - 1.0;
      if (y == 0.0) y = 0.0;
    }
    return 0x1p-1022 * y;
  }
  const double scale = expl_from_bits(sbits);
  return __builtin_fma(scale, tmp, scale);
}
