// orb_atan2f.h -- bit-exact replica of the libm atan2f / atanf that KannalaBrandt8::project calls
// (`theta = atan2f(sqrtf(x2_plus_y2), p3D.z)`, `psi = atan2f(p3D.y, p3D.x)`, KannalaBrandt8.cpp:31-32), so that the
// projection of a last-frame search can run on the device and still place its window exactly where the reference does.
//
// glibc up to 2.40 implements both in single precision after fdlibm (sysdeps/ieee754/flt-32/e_atan2f.c, s_atanf.c:
// argument reduction to |x| < 7/16 by the four classic breakpoints, an 11-term odd/even split polynomial, hi/lo table
// of atan(0.5), atan(1), atan(1.5), atan(inf)); x86-64 has no FMA / ifunc variant of these two, so every operation is
// a plain IEEE-754 single add / mul / div in source order and host and gfx950 agree bit for bit when nothing is
// contracted (-ffp-contract=off, correctly rounded fp32 division).
// tests/test_libm_replicas.py checks the replica against the host libm: atanf over every float (ORB_EXHAUSTIVE=1) or a
// strided sample, atan2f on the quadrant / zero / infinity cases and on random pairs.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define ORBAT_HD __host__ __device__ inline
#else
#define ORBAT_HD inline
#endif

namespace orbat {

ORBAT_HD int32_t fbits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (int32_t)__float_as_uint(f);
#else
  int32_t u;
  memcpy(&u, &f, 4);
  return u;
#endif
}
ORBAT_HD float bitsf(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(u);
#else
  float f;
  memcpy(&f, &u, 4);
  return f;
#endif
}

// s_atanf.c
ORBAT_HD float ref_atanf(float x) {
  const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};   // 0x3eed6338 0x3f490fda 0x3f7b985e 0x3fc90fda
  const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};   // 0x31ac3769 0x33222168 0x33140fb4 0x33a22168
  const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f,
              aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f,
              aT10 = 1.6285819933e-02f;
  const float one = 1.0f;
  const int32_t hx = fbits(x), ix = hx & 0x7fffffff;
  int id;
  if (ix >= 0x4c000000) {  // |x| >= 2^25
    if (ix > 0x7f800000) return x + x;  // NaN
    return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
  }
  if (ix < 0x3ee00000) {            // |x| < 0.4375
    if (ix < 0x31000000) return x;  // |x| < 2^-29 (huge + x > one raises inexact only)
    id = -1;
  } else {
    x = bitsf((uint32_t)ix);        // fabsf
    if (ix < 0x3f980000) {          // |x| < 1.1875
      if (ix < 0x3f300000) { id = 0; x = (2.0f * x - one) / (2.0f + x); }   // 7/16 <= |x| < 11/16
      else { id = 1; x = (x - one) / (x + one); }                           // 11/16 <= |x| < 19/16
    } else {
      if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (one + 1.5f * x); }   // |x| < 2.4375
      else { id = 3; x = -1.0f / x; }                                       // 2.4375 <= |x| < 2^25
    }
  }
  const float z = x * x;
  const float w = z * z;
  const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  if (id < 0) return x - x * (s1 + s2);
  const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
  return hx < 0 ? -r : r;
}

// e_atan2f.c
ORBAT_HD float ref_atan2f(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  const int32_t hx = fbits(x), hy = fbits(y);
  const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;  // NaN
  if (hx == 0x3f800000) return ref_atanf(y);              // x == 1.0
  const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);      // 2 * sign(x) + sign(y)
  if (iy == 0) {                                          // y == 0
    switch (m) {
      case 0:
      case 1: return y;
      case 2: return pi + tiny;
      default: return -pi - tiny;
    }
  }
  if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;  // x == 0
  if (ix == 0x7f800000) {                                       // x == INF
    if (iy == 0x7f800000) {
      switch (m) {
        case 0: return pi_o_4 + tiny;
        case 1: return -pi_o_4 - tiny;
        case 2: return 3.0f * pi_o_4 + tiny;
        default: return -3.0f * pi_o_4 - tiny;
      }
    } else {
      switch (m) {
        case 0: return 0.0f;
        case 1: return -0.0f;
        case 2: return pi + tiny;
        default: return -pi - tiny;
      }
    }
  }
  if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;  // y == INF
  const int32_t k = (iy - ix) >> 23;
  float z;
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;       // |y/x| > 2^60
  else if (hx < 0 && k < -60) z = 0.0f;        // |y|/x < -2^60
  else z = ref_atanf(bitsf((uint32_t)fbits(y / x) & 0x7fffffffu));
  switch (m) {
    case 0: return z;
    case 1: return bitsf((uint32_t)fbits(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}

}  // namespace orbat
