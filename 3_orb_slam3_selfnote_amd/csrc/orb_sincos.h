// orb_sincos.h -- bit-exact replica of the libm cosf/sinf that the reference calls at ORBextractor.cc:111
// (`float a = (float)cos(angle), b = (float)sin(angle)` resolves to std::cos(float) -> glibc cosf).
//
// The device has no glibc; ROCm's device libm differs from it in the last ulp, and a 1-ulp change of a/b can move
// a cvRound()-ed rBRIEF sample to the neighbouring pixel.  This file restates the
// published algorithm glibc >= 2.28 uses for sinf/cosf (ARM "optimized-routines" sincosf: double-precision
// argument reduction by pi/2 and degree-8/9 minimax polynomials; glibc sysdeps/ieee754/flt-32/s_sinf.c,
// s_cosf.c, sincosf.h) in the x86-64 FMA variant that libm's ifunc selects on FMA-capable hosts.  Every operation
// is an IEEE-754 double mul / fma / cvt, so host and gfx950 produce identical bits.
// tests/test_sincos.py checks the replica against the host libm (exhaustively over all 1 086 953 884 floats in
// [0, 6.3] when run with ORB_EXHAUSTIVE=1; a strided sample otherwise).  Only |x| < 120 is supported -- the
// extractor only ever passes angles in [0, 2*pi].
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define ORB_HD __host__ __device__ inline
#else
#define ORB_HD inline
#endif

namespace orbsc {

ORB_HD double madd(double a, double b, double c) { return __builtin_fma(a, b, c); }

ORB_HD uint32_t top12(float f) {
  uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
  u = __float_as_uint(f);
#else
  memcpy(&u, &f, 4);
#endif
  return (u >> 20) & 0x7ff;
}

// polynomial of sincosf.h:sinf_poly; neg selects __sincosf_table[1] (cosine coefficients negated)
ORB_HD float poly(double x, double x2, bool neg, int n) {
  const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
  const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
               C4 = 0x1.99343027bf8c3p-16;
  if ((n & 1) == 0) {
    double x3 = x * x2;
    double s1 = madd(x2, S3, S2);
    double x7 = x3 * x2;
    double s = madd(x3, S1, x);
    return (float)madd(x7, s1, s);
  } else {
    const double sg = neg ? -1.0 : 1.0;
    double x4 = x2 * x2;
    double c2 = madd(x2, sg * C4, sg * C3);
    double c1 = madd(x2, sg * C1, sg * C0);
    double x6 = x4 * x2;
    double c = madd(x4, sg * C2, c1);
    return (float)madd(x6, c2, c);
  }
}

ORB_HD void reduce_fast(double &x, int &n) {
  const double HPI_INV = 0x1.45f306dc9c883p+23; /* 2/pi * 2^24 */
  const double HPI = 0x1.921fb54442d18p+0;      /* pi/2 */
  double r = x * HPI_INV;
  n = ((int32_t)r + 0x800000) >> 24;
  x = madd(-(double)n, HPI, x);
}

ORB_HD double quadrant_sign(int n) { return ((n + 1) & 2) ? -1.0 : 1.0; } /* {1,-1,-1,1}[n&3] */

ORB_HD float ref_sinf(float y) {
  double x = (double)y;
  if (top12(y) < 0x3f4) { /* |y| < pi/4 */
    double s = x * x;
    if (top12(y) < 0x398) return y; /* |y| < 2^-12 */
    return poly(x, s, false, 0);
  }
  int n;
  reduce_fast(x, n);
  return poly(x * quadrant_sign(n), x * x, (n & 2) != 0, n);
}

ORB_HD float ref_cosf(float y) {
  double x = (double)y;
  if (top12(y) < 0x3f4) {
    double s = x * x;
    if (top12(y) < 0x398) return 1.0f;
    return poly(x, s, false, 1);
  }
  int n;
  reduce_fast(x, n);
  return poly(x * quadrant_sign(n), x * x, (n & 2) != 0, n ^ 1);
}

}  // namespace orbsc
