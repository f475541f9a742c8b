/*
 * orb_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).  See orb_oracle.h.
 *
 * PARITY UNPINNED: no golden vectors exist in the reference; OpenCV primitive semantics are
 * restated from the OpenCV 3.4.x pure-C++ code paths.
 *
 * Build: gcc -O2 -std=gnu11 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  All float
 * expressions are written so that every intermediate has the type it has in the reference.
 */
#include "orb_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* rounding helpers                                                             */
/* ------------------------------------------------------------------------------------------ */
int orc_cvRound(double v) { return (int)lrint(v); } /* round-half-even in the default FP mode */
static int cvFloor_(double v) { return (int)floor(v); }
static int cvCeil_(double v) { return (int)ceil(v); }

float orc_libm_cosf(float x) { return cosf(x); }
float orc_libm_sinf(float x) { return sinf(x); }
float orc_libm_atanf(float x) { return atanf(x); }
float orc_libm_atan2f(float y, float x) { return atan2f(y, x); }

/* Native sweeps for tests/test_libm_replicas.py: the product's device-side replicas of cosf / sinf / atanf / atan2f (passed in
 * as function pointers; evaluated on the host from the same source the kernels compile) against THIS host's libm.
 * which: 0 cosf, 1 sinf, 2 atanf; bit patterns lo..hi (inclusive) in steps of `step`, both signs when both_signs.
 * Returns the number of mismatching results (NaN inputs skipped). */
long orc_sweep_unary(float (*replica)(float), int which, uint32_t lo, uint32_t hi, uint32_t step, int both_signs) {
  long bad = 0;
  if (step == 0) step = 1;
  for (uint64_t u = lo; u <= hi; u += step) {
    for (int sg = 0; sg <= (both_signs ? 1 : 0); sg++) {
      const uint32_t b = (uint32_t)u | (sg ? 0x80000000u : 0u);
      float x, a, r;
      memcpy(&x, &b, 4);
      if (x != x) continue;
      a = which == 0 ? cosf(x) : which == 1 ? sinf(x) : atanf(x);
      r = replica(x);
      if (memcmp(&a, &r, 4) != 0) bad++;
    }
  }
  return bad;
}
/* n random pairs (xorshift64*, seeded): odd draws are raw bit patterns (all magnitudes, infinities, zeros, denormals),
 * even draws camera-like coordinates in [-scale, scale]. */
long orc_sweep_atan2f(float (*replica)(float, float), uint64_t seed, long n, float scale) {
  long bad = 0;
  uint64_t s = seed ? seed : 88172645463325252ull;
  for (long i = 0; i < n; i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    float y, x, a, r;
    if (i & 1) {
      uint32_t by = (uint32_t)s, bx = (uint32_t)(s >> 32);
      memcpy(&y, &by, 4); memcpy(&x, &bx, 4);
    } else {
      y = (float)((double)(int32_t)(uint32_t)s / 2147483648.0 * (double)scale);
      x = (float)((double)(int32_t)(uint32_t)(s >> 32) / 2147483648.0 * (double)scale);
    }
    if (x != x || y != y) continue;
    a = atan2f(y, x);
    r = replica(y, x);
    if (memcmp(&a, &r, 4) != 0) bad++;
  }
  return bad;
}

static const int8_t kPattern[1024] = {
#include "../include/orb_pattern_data.inc"
};

#define PATCH_SIZE 31      /* ORBextractor.cc:70 */
#define HALF_PATCH_SIZE 15 /* ORBextractor.cc:71 */
#define EDGE_THRESHOLD 19  /* ORBextractor.cc:72 */

/* ------------------------------------------------------------------------------------------ */
/* X0: constructor, ORBextractor.cc:408-468                                                     */
/* ------------------------------------------------------------------------------------------ */
void orc_extractor_init(orc_extractor *e, int nfeatures, float scaleFactor_, int nlevels, int iniTh, int minTh) {
  memset(e, 0, sizeof(*e));
  e->nfeatures = nfeatures;
  e->scaleFactor = (double)scaleFactor_; /* member is double, initialised from float (ORBextractor.h:96) */
  e->nlevels = nlevels;
  e->iniThFAST = iniTh;
  e->minThFAST = minTh;
  e->mvScaleFactor[0] = 1.0f;
  e->mvLevelSigma2[0] = 1.0f;
  for (int i = 1; i < nlevels; i++) {
    e->mvScaleFactor[i] = (float)((double)e->mvScaleFactor[i - 1] * e->scaleFactor); /* :419 */
    e->mvLevelSigma2[i] = e->mvScaleFactor[i] * e->mvScaleFactor[i];                 /* :420 */
  }
  for (int i = 0; i < nlevels; i++) {
    e->mvInvScaleFactor[i] = 1.0f / e->mvScaleFactor[i];
    e->mvInvLevelSigma2[i] = 1.0f / e->mvLevelSigma2[i];
  }
  float factor = (float)(1.0 / e->scaleFactor); /* 1.0f / double -> double -> float, :433 */
  float nDesired = (float)nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels)); /* :434 */
  int sum = 0;
  for (int level = 0; level < nlevels - 1; level++) {
    e->mnFeaturesPerLevel[level] = orc_cvRound(nDesired);
    sum += e->mnFeaturesPerLevel[level];
    nDesired *= factor;
  }
  e->mnFeaturesPerLevel[nlevels - 1] = (nfeatures - sum) > 0 ? (nfeatures - sum) : 0;

  /* umax, :452-467 */
  int v, v0;
  int vmax = cvFloor_(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
  int vmin = cvCeil_(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
  const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
  for (v = 0; v <= vmax; ++v) e->umax[v] = orc_cvRound(sqrt(hp2 - v * v));
  for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
    while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
    e->umax[v] = v0;
    ++v0;
  }
}

void orc_level_size(const orc_extractor *e, int level, int cols, int rows, int *lcols, int *lrows) {
  float scale = e->mvInvScaleFactor[level];
  *lcols = orc_cvRound((float)cols * scale); /* :1192-1193 */
  *lrows = orc_cvRound((float)rows * scale);
}

/* ------------------------------------------------------------------------------------------ */
/* A.3 cv::resize INTER_LINEAR, 8UC1 (fixed point, 11 coefficient bits)                          */
/* ------------------------------------------------------------------------------------------ */
static short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, size_t sstride, uint8_t *dst, int dw, int dh,
                          size_t dstride) {
  /* hal::resize: inv_scale = dsize/ssize (double); scale = 1./inv_scale */
  double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  int *xofs = (int *)malloc(sizeof(int) * dw);
  short *ialpha = (short *)malloc(sizeof(short) * 2 * dw);
  int *yofs = (int *)malloc(sizeof(int) * dh);
  short *ibeta = (short *)malloc(sizeof(short) * 2 * dh);
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cvFloor_(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    ialpha[2 * dx] = sat_short(orc_cvRound((1.f - fx) * 2048));
    ialpha[2 * dx + 1] = sat_short(orc_cvRound(fx * 2048));
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cvFloor_(fy);
    fy -= sy;
    yofs[dy] = sy;
    ibeta[2 * dy] = sat_short(orc_cvRound((1.f - fy) * 2048));
    ibeta[2 * dy + 1] = sat_short(orc_cvRound(fy * 2048));
  }
  int *row0 = (int *)malloc(sizeof(int) * dw), *row1 = (int *)malloc(sizeof(int) * dw);
  for (int dy = 0; dy < dh; dy++) {
    int sy0 = yofs[dy], sy1 = yofs[dy] + 1;
    if (sy0 < 0) sy0 = 0;
    if (sy0 > sh - 1) sy0 = sh - 1;
    if (sy1 < 0) sy1 = 0;
    if (sy1 > sh - 1) sy1 = sh - 1;
    const uint8_t *S0 = src + (size_t)sy0 * sstride, *S1 = src + (size_t)sy1 * sstride;
    for (int dx = 0; dx < dw; dx++) {
      int sx = xofs[dx];
      int sx1 = sx + 1 < sw ? sx + 1 : sx; /* right tap has weight 0 when clamped */
      row0[dx] = S0[sx] * ialpha[2 * dx] + S0[sx1] * ialpha[2 * dx + 1];
      row1[dx] = S1[sx] * ialpha[2 * dx] + S1[sx1] * ialpha[2 * dx + 1];
    }
    int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
    uint8_t *D = dst + (size_t)dy * dstride;
    for (int dx = 0; dx < dw; dx++) {
      int v = (((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2;
      D[dx] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  }
  free(xofs); free(ialpha); free(yofs); free(ibeta); free(row0); free(row1);
}

/* ------------------------------------------------------------------------------------------ */
/* A.2 copyMakeBorder REFLECT_101                                                               */
/* ------------------------------------------------------------------------------------------ */
static int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) {
    if (p < 0) p = -p;
    else p = 2 * (n - 1) - p;
  }
  return p;
}

void orc_copy_make_border101(const uint8_t *src, int w, int h, size_t sstride, uint8_t *dst, int border, size_t dstride) {
  for (int y = -border; y < h + border; y++) {
    const uint8_t *S = src + (size_t)reflect101(y, h) * sstride;
    uint8_t *D = dst + (size_t)(y + border) * dstride;
    for (int x = -border; x < w + border; x++) D[x + border] = S[reflect101(x, w)];
  }
}

/* ------------------------------------------------------------------------------------------ */
/* A.5 GaussianBlur 7x7 sigma=2, fixed point                                                    */
/* ------------------------------------------------------------------------------------------ */
void orc_gauss7_kernel(int k[7]) {
  /* getGaussianKernel(7, 2): exp(-(i-3)^2/(2 sigma^2)) normalised, then round(g*256) per tap */
  double g[7], sum = 0;
  const double sigma = 2.0, scale2X = -0.5 / (sigma * sigma);
  for (int i = 0; i < 7; i++) {
    double x = i - 3.0;
    g[i] = exp(scale2X * x * x);
    sum += g[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < 7; i++) k[i] = orc_cvRound(g[i] * sum * 256.0);
}

void orc_gaussian_blur7(const uint8_t *src, int w, int h, size_t sstride, uint8_t *dst, size_t dstride) {
  int k[7];
  orc_gauss7_kernel(k);
  uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)w * h);
  const uint32_t k0 = (uint32_t)k[0], k1 = (uint32_t)k[1], k2 = (uint32_t)k[2], k3 = (uint32_t)k[3], k4 = (uint32_t)k[4], k5 = (uint32_t)k[5],
                 k6 = (uint32_t)k[6];
  /* row pass: the interior without the border arithmetic (same sums, same order), the 3 + 3 edge columns through reflect101 */
  for (int y = 0; y < h; y++) {
    const uint8_t *S = src + (size_t)y * sstride;
    uint32_t *T = tmp + (size_t)y * w;
    for (int x = 0; x < w; x++) {
      if (x == 3 && w > 6) {
        for (; x < w - 3; x++)
          T[x] = k0 * S[x - 3] + k1 * S[x - 2] + k2 * S[x - 1] + k3 * S[x] + k4 * S[x + 1] + k5 * S[x + 2] + k6 * S[x + 3];
        x--;
        continue;
      }
      uint32_t s = 0;
      for (int i = 0; i < 7; i++) s += (uint32_t)k[i] * S[reflect101(x + i - 3, w)];
      T[x] = s;
    }
  }
  /* column pass: seven row pointers per output row */
  for (int y = 0; y < h; y++) {
    uint8_t *D = dst + (size_t)y * dstride;
    const uint32_t *R[7];
    for (int j = 0; j < 7; j++) R[j] = tmp + (size_t)reflect101(y + j - 3, h) * w;
    for (int x = 0; x < w; x++) {
      uint32_t s = k0 * R[0][x] + k1 * R[1][x] + k2 * R[2][x] + k3 * R[3][x] + k4 * R[4][x] + k5 * R[5][x] + k6 * R[6][x];
      uint32_t v = (s + 32768u) >> 16;
      D[x] = (uint8_t)(v > 255 ? 255 : v);
    }
  }
  free(tmp);
}

/* ------------------------------------------------------------------------------------------ */
/* A.1 cv::FAST 9_16 with non-max suppression                                                   */
/* ------------------------------------------------------------------------------------------ */
static const int kCircle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                   {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

int orc_fast_corner_score(const uint8_t *p, size_t stride, int threshold) {
  /* cornerScore<16>: d[k] = v - circle[k], 25 entries (wrap) */
  int d[25];
  int v = p[0];
  for (int k = 0; k < 25; k++) {
    int kk = k & 15;
    d[k] = v - p[(ptrdiff_t)kCircle[kk][1] * (ptrdiff_t)stride + kCircle[kk][0]];
  }
  int a0 = threshold;
  for (int k = 0; k < 16; k += 2) {
    int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
    a = a < d[k + 3] ? a : d[k + 3];
    if (a <= a0) continue;
    for (int j = 4; j <= 8; j++) a = a < d[k + j] ? a : d[k + j];
    int t = a < d[k] ? a : d[k];
    if (t > a0) a0 = t;
    t = a < d[k + 9] ? a : d[k + 9];
    if (t > a0) a0 = t;
  }
  int b0 = -a0;
  for (int k = 0; k < 16; k += 2) {
    int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
    for (int j = 3; j <= 5; j++) b = b > d[k + j] ? b : d[k + j];
    if (b >= b0) continue;
    for (int j = 6; j <= 8; j++) b = b > d[k + j] ? b : d[k + j];
    int t = b > d[k] ? b : d[k];
    if (t < b0) b0 = t;
    t = b > d[k + 9] ? b : d[k + 9];
    if (t < b0) b0 = t;
  }
  return -b0 - 1;
}

int orc_fast9_16(const uint8_t *img, int w, int h, size_t stride, int threshold, int *xys, int cap) {
  const int K = 8, N = 25;
  int n = 0;
  if (threshold < 0) threshold = 0;
  if (threshold > 255) threshold = 255;
  if (w < 7 || h < 7) return 0;
  /* three rolling score rows + corner position lists, exactly like FAST_t<16> */
  uint8_t *buf = (uint8_t *)calloc((size_t)w * 3, 1);
  int *cp = (int *)malloc(sizeof(int) * ((size_t)w + 1) * 3);
  uint8_t *sc[3] = {buf, buf + w, buf + 2 * w};
  int *cpbuf[3] = {cp + 1, cp + 1 + (w + 1), cp + 1 + 2 * (w + 1)};
  ptrdiff_t off[16];
  for (int k = 0; k < 16; k++) off[k] = (ptrdiff_t)kCircle[k][1] * (ptrdiff_t)stride + kCircle[k][0];
  for (int i = 3; i < h - 2; i++) {
    const uint8_t *ptr = img + (size_t)i * stride + 3;
    uint8_t *curr = sc[(i - 3) % 3];
    int *cornerpos = cpbuf[(i - 3) % 3];
    memset(curr, 0, (size_t)w);
    int ncorners = 0;
    if (i < h - 3) {
      for (int j = 3; j < w - 3; j++, ptr++) {
        int v = ptr[0];
        int vt_lo = v - threshold, vt_hi = v + threshold;
        int is_corner = 0;
        /* high-speed rejection test of FAST_t<16>: every 9-arc contains one pixel of each opposite pair
         * (k, k+8), so all 8 pairs must have a darker (bit 1) resp. brighter (bit 2) member */
        int dmask = 3;
        for (int k = 0; k < 8 && dmask; k++) {
          int xa = ptr[off[k]], xb = ptr[off[k + 8]];
          int m = (xa < vt_lo ? 1 : xa > vt_hi ? 2 : 0) | (xb < vt_lo ? 1 : xb > vt_hi ? 2 : 0);
          dmask &= m;
        }
        if (!dmask) continue;
        /* darker arc: x < v - t ; brighter arc: x > v + t ; 9 contiguous among 25 (wrap) */
        if (dmask & 1) {
          int count = 0;
          for (int k = 0; k < N; k++) {
            int x = ptr[off[k & 15]];
            if (x < vt_lo) { if (++count > K) { is_corner = 1; break; } }
            else count = 0;
          }
        }
        if (!is_corner && (dmask & 2)) {
          int count = 0;
          for (int k = 0; k < N; k++) {
            int x = ptr[off[k & 15]];
            if (x > vt_hi) { if (++count > K) { is_corner = 1; break; } }
            else count = 0;
          }
        }
        if (is_corner) {
          cornerpos[ncorners++] = j;
          curr[j] = (uint8_t)orc_fast_corner_score(ptr, stride, threshold);
        }
      }
    }
    cornerpos[-1] = ncorners;
    if (i == 3) continue;
    const uint8_t *prev = sc[(i - 4 + 3) % 3];
    const uint8_t *pprev = sc[(i - 5 + 3) % 3];
    cornerpos = cpbuf[(i - 4 + 3) % 3];
    ncorners = cornerpos[-1];
    for (int k = 0; k < ncorners; k++) {
      int j = cornerpos[k];
      int score = prev[j];
      if (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
          score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1]) {
        if (n < cap) {
          xys[3 * n] = j;
          xys[3 * n + 1] = i - 1;
          xys[3 * n + 2] = score;
        }
        n++;
      }
    }
  }
  free(buf);
  free(cp);
  return n;
}

/* ------------------------------------------------------------------------------------------ */
/* A.6 cv::fastAtan2                                                                            */
/* ------------------------------------------------------------------------------------------ */
float orc_fast_atan2(float y, float x) {
  const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
  const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
  const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
  float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

/* ------------------------------------------------------------------------------------------ */
/* X5 IC_Angle, ORBextractor.cc:75-102                                                          */
/* ------------------------------------------------------------------------------------------ */
float orc_ic_angle(const orc_extractor *e, const uint8_t *img, size_t stride, float ptx, float pty) {
  int m_01 = 0, m_10 = 0;
  const uint8_t *center = img + (ptrdiff_t)orc_cvRound(pty) * (ptrdiff_t)stride + orc_cvRound(ptx);
  for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
  int step = (int)stride;
  for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
    int v_sum = 0;
    int d = e->umax[v];
    for (int u = -d; u <= d; ++u) {
      int val_plus = center[u + v * step], val_minus = center[u - v * step];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return orc_fast_atan2((float)m_01, (float)m_10);
}

/* ------------------------------------------------------------------------------------------ */
/* X7 computeOrbDescriptor, ORBextractor.cc:106-145                                             */
/* ------------------------------------------------------------------------------------------ */
void orc_compute_descriptor(const uint8_t *img, size_t stride, float ptx, float pty, float angle_deg, uint8_t desc[32]) {
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f); /* :105 */
  float angle = angle_deg * factorPI;
  float a = cosf(angle), b = sinf(angle); /* std::cos(float) via `using namespace std` */
  const uint8_t *center = img + (ptrdiff_t)orc_cvRound(pty) * (ptrdiff_t)stride + orc_cvRound(ptx);
  const int step = (int)stride;
  const int8_t *pat = kPattern;
#define GET_VALUE(idx)                                                                            \
  center[orc_cvRound((float)pat[2 * (idx)] * b + (float)pat[2 * (idx) + 1] * a) * step +          \
         orc_cvRound((float)pat[2 * (idx)] * a - (float)pat[2 * (idx) + 1] * b)]
  for (int i = 0; i < 32; ++i, pat += 32) {
    int val = 0;
    for (int j = 0; j < 8; j++) {
      int t0 = GET_VALUE(2 * j), t1 = GET_VALUE(2 * j + 1);
      val |= (t0 < t1) << j;
    }
    desc[i] = (uint8_t)val;
  }
#undef GET_VALUE
}

/* ------------------------------------------------------------------------------------------ */
/* X2 candidate generation, ORBextractor.cc:771-854                                             */
/* ------------------------------------------------------------------------------------------ */
int orc_level_candidates(const orc_extractor *e, const uint8_t *img, int cols, int rows, size_t stride, float *xyr, int cap) {
  const float W = 30;
  const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
  const int maxBorderX = cols - EDGE_THRESHOLD + 3, maxBorderY = rows - EDGE_THRESHOLD + 3;
  const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
  const int nCols = (int)(width / W), nRows = (int)(height / W);
  if (nCols <= 0 || nRows <= 0) return 0; /* reference divides by zero here; defined as "no keypoints" */
  const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
  int n = 0;
  int cellcap = 64 * 64;
  int *cell = (int *)malloc(sizeof(int) * 3 * cellcap);
  for (int i = 0; i < nRows; i++) {
    const float iniY = (float)(minBorderY + i * hCell);
    float maxY = iniY + hCell + 6;
    if (iniY >= maxBorderY - 3) continue;
    if (maxY > maxBorderY) maxY = (float)maxBorderY;
    for (int j = 0; j < nCols; j++) {
      const float iniX = (float)(minBorderX + j * wCell);
      float maxX = iniX + wCell + 6;
      if (iniX >= maxBorderX - 6) continue;
      if (maxX > maxBorderX) maxX = (float)maxBorderX;
      int x0 = (int)iniX, x1 = (int)maxX, y0 = (int)iniY, y1 = (int)maxY;
      const uint8_t *sub = img + (size_t)y0 * stride + x0;
      int nc = orc_fast9_16(sub, x1 - x0, y1 - y0, stride, e->iniThFAST, cell, cellcap);
      if (nc == 0) nc = orc_fast9_16(sub, x1 - x0, y1 - y0, stride, e->minThFAST, cell, cellcap);
      for (int k = 0; k < nc; k++) {
        if (n < cap) {
          xyr[3 * n] = (float)cell[3 * k] + (float)(j * wCell);
          xyr[3 * n + 1] = (float)cell[3 * k + 1] + (float)(i * hCell);
          xyr[3 * n + 2] = (float)cell[3 * k + 2];
        }
        n++;
      }
    }
  }
  free(cell);
  return n;
}

/* ------------------------------------------------------------------------------------------ */
/* X4 DistributeOctTree, ORBextractor.cc:479-761                                                */
/* Nodes live in a pool; the std::list is a doubly linked index list.  The reference sorts
 * pair<int, ExtractorNode*>; pointer ties are defined here as creation order (a later-created
 * node compares greater), i.e. what a monotone allocator gives.      */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
  int *keys;
  int nkeys;
  int bNoMore;
  int prev, next; /* list links */
} onode;

typedef struct {
  onode *nodes;
  int count, capn;
  int head, tail, size;
} olist;

static int ol_new(olist *L) {
  if (L->count == L->capn) {
    L->capn *= 2;
    L->nodes = (onode *)realloc(L->nodes, sizeof(onode) * (size_t)L->capn);
  }
  onode *n = &L->nodes[L->count];
  memset(n, 0, sizeof(*n));
  n->prev = n->next = -1;
  return L->count++;
}
static void ol_push_front(olist *L, int id) {
  L->nodes[id].prev = -1;
  L->nodes[id].next = L->head;
  if (L->head >= 0) L->nodes[L->head].prev = id;
  L->head = id;
  if (L->tail < 0) L->tail = id;
  L->size++;
}
static void ol_push_back(olist *L, int id) {
  L->nodes[id].next = -1;
  L->nodes[id].prev = L->tail;
  if (L->tail >= 0) L->nodes[L->tail].next = id;
  L->tail = id;
  if (L->head < 0) L->head = id;
  L->size++;
}
static int ol_erase(olist *L, int id) { /* returns next */
  int p = L->nodes[id].prev, nx = L->nodes[id].next;
  if (p >= 0) L->nodes[p].next = nx; else L->head = nx;
  if (nx >= 0) L->nodes[nx].prev = p; else L->tail = p;
  L->size--;
  return nx;
}

/* ExtractorNode::DivideNode, ORBextractor.cc:479-535.  Creates 4 pool nodes (not linked). */
static void divide_node(olist *L, int id, const float *xyr, int ch[4]) {
  for (int c = 0; c < 4; c++) ch[c] = ol_new(L);
  onode *p = &L->nodes[id];
  const int halfX = (int)ceilf((float)(p->URx - p->ULx) / 2);
  const int halfY = (int)ceilf((float)(p->BRy - p->ULy) / 2);
  onode *n1 = &L->nodes[ch[0]], *n2 = &L->nodes[ch[1]], *n3 = &L->nodes[ch[2]], *n4 = &L->nodes[ch[3]];
  n1->ULx = p->ULx; n1->ULy = p->ULy;
  n1->URx = p->ULx + halfX; n1->URy = p->ULy;
  n1->BLx = p->ULx; n1->BLy = p->ULy + halfY;
  n1->BRx = p->ULx + halfX; n1->BRy = p->ULy + halfY;
  n2->ULx = n1->URx; n2->ULy = n1->URy;
  n2->URx = p->URx; n2->URy = p->URy;
  n2->BLx = n1->BRx; n2->BLy = n1->BRy;
  n2->BRx = p->URx; n2->BRy = p->ULy + halfY;
  n3->ULx = n1->BLx; n3->ULy = n1->BLy;
  n3->URx = n1->BRx; n3->URy = n1->BRy;
  n3->BLx = p->BLx; n3->BLy = p->BLy;
  n3->BRx = n1->BRx; n3->BRy = p->BLy;
  n4->ULx = n3->URx; n4->ULy = n3->URy;
  n4->URx = n2->BRx; n4->URy = n2->BRy;
  n4->BLx = n3->BRx; n4->BLy = n3->BRy;
  n4->BRx = p->BRx; n4->BRy = p->BRy;
  for (int c = 0; c < 4; c++) {
    L->nodes[ch[c]].keys = (int *)malloc(sizeof(int) * (size_t)(p->nkeys > 0 ? p->nkeys : 1));
    L->nodes[ch[c]].nkeys = 0;
  }
  for (int i = 0; i < p->nkeys; i++) {
    int k = p->keys[i];
    float kx = xyr[3 * k], ky = xyr[3 * k + 1];
    onode *t;
    if (kx < (float)n1->URx) t = (ky < (float)n1->BRy) ? n1 : n3;
    else t = (ky < (float)n1->BRy) ? n2 : n4;
    t->keys[t->nkeys++] = k;
  }
  for (int c = 0; c < 4; c++)
    if (L->nodes[ch[c]].nkeys == 1) L->nodes[ch[c]].bNoMore = 1;
}

typedef struct { int size; int id; } szid;
static int cmp_szid(const void *a, const void *b) {
  const szid *x = (const szid *)a, *y = (const szid *)b;
  if (x->size != y->size) return x->size < y->size ? -1 : 1;
  return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}

int orc_distribute_octtree(const float *xyr, int n, int minX, int maxX, int minY, int maxY, int N, float *out, int cap) {
  const int nIni = (int)roundf((float)(maxX - minX) / (float)(maxY - minY)); /* :541 */
  if (nIni <= 0) return 0; /* reference: division by zero / empty vpIniNodes (undefined); defined as empty */
  const float hX = (float)(maxX - minX) / (float)nIni;
  olist L;
  L.capn = 64; L.count = 0; L.head = L.tail = -1; L.size = 0;
  L.nodes = (onode *)malloc(sizeof(onode) * (size_t)L.capn);
  int *ini = (int *)malloc(sizeof(int) * (size_t)nIni);
  for (int i = 0; i < nIni; i++) {
    int id = ol_new(&L);
    onode *ni = &L.nodes[id];
    ni->ULx = (int)(hX * (float)i); ni->ULy = 0;
    ni->URx = (int)(hX * (float)(i + 1)); ni->URy = 0;
    ni->BLx = ni->ULx; ni->BLy = maxY - minY;
    ni->BRx = ni->URx; ni->BRy = maxY - minY;
    ni->keys = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    ni->nkeys = 0;
    ol_push_back(&L, id);
    ini[i] = id;
  }
  for (int i = 0; i < n; i++) {
    int r = (int)(xyr[3 * i] / hX); /* :567 */
    if (r < 0) r = 0;
    if (r >= nIni) r = nIni - 1; /* reference would index out of range; cannot happen for in-range x */
    onode *t = &L.nodes[ini[r]];
    t->keys[t->nkeys++] = i;
  }
  free(ini);
  for (int lit = L.head; lit >= 0;) {
    if (L.nodes[lit].nkeys == 1) { L.nodes[lit].bNoMore = 1; lit = L.nodes[lit].next; }
    else if (L.nodes[lit].nkeys == 0) lit = ol_erase(&L, lit);
    else lit = L.nodes[lit].next;
  }
  int bFinish = 0;
  szid *vSize = (szid *)malloc(sizeof(szid) * (size_t)(4 * (n + 8)));
  szid *vPrev = (szid *)malloc(sizeof(szid) * (size_t)(4 * (n + 8)));
  int nSize = 0;
  while (!bFinish) {
    int prevSize = L.size;
    int lit = L.head;
    int nToExpand = 0;
    nSize = 0;
    while (lit >= 0) {
      if (L.nodes[lit].bNoMore) { lit = L.nodes[lit].next; continue; }
      int ch[4];
      divide_node(&L, lit, xyr, ch);
      for (int c = 0; c < 4; c++) {
        if (L.nodes[ch[c]].nkeys > 0) {
          ol_push_front(&L, ch[c]);
          if (L.nodes[ch[c]].nkeys > 1) {
            nToExpand++;
            vSize[nSize].size = L.nodes[ch[c]].nkeys;
            vSize[nSize].id = ch[c];
            nSize++;
          }
        }
      }
      lit = ol_erase(&L, lit);
    }
    if (L.size >= N || L.size == prevSize) {
      bFinish = 1;
    } else if (L.size + nToExpand * 3 > N) {
      while (!bFinish) {
        prevSize = L.size;
        int nPrev = nSize;
        memcpy(vPrev, vSize, sizeof(szid) * (size_t)nSize);
        nSize = 0;
        qsort(vPrev, (size_t)nPrev, sizeof(szid), cmp_szid);
        for (int j = nPrev - 1; j >= 0; j--) {
          int ch[4];
          divide_node(&L, vPrev[j].id, xyr, ch);
          for (int c = 0; c < 4; c++) {
            if (L.nodes[ch[c]].nkeys > 0) {
              ol_push_front(&L, ch[c]);
              if (L.nodes[ch[c]].nkeys > 1) {
                vSize[nSize].size = L.nodes[ch[c]].nkeys;
                vSize[nSize].id = ch[c];
                nSize++;
              }
            }
          }
          ol_erase(&L, vPrev[j].id);
          if (L.size >= N) break;
        }
        if (L.size >= N || L.size == prevSize) bFinish = 1;
      }
    }
  }
  int nout = 0;
  for (int lit = L.head; lit >= 0; lit = L.nodes[lit].next) {
    onode *nd = &L.nodes[lit];
    int best = nd->keys[0];
    float maxResponse = xyr[3 * best + 2];
    for (int k = 1; k < nd->nkeys; k++) {
      if (xyr[3 * nd->keys[k] + 2] > maxResponse) {
        best = nd->keys[k];
        maxResponse = xyr[3 * best + 2];
      }
    }
    if (nout < cap) {
      out[3 * nout] = xyr[3 * best];
      out[3 * nout + 1] = xyr[3 * best + 1];
      out[3 * nout + 2] = xyr[3 * best + 2];
    }
    nout++;
  }
  for (int i = 0; i < L.count; i++) free(L.nodes[i].keys);
  free(L.nodes);
  free(vSize);
  free(vPrev);
  return nout;
}

/* ------------------------------------------------------------------------------------------ */
/* X1 ComputePyramid, ORBextractor.cc:1186-1219 (ROI content only; the 19-px border is never
 * read by the extractor and is produced on demand by orc_copy_make_border101).                 */
/* ------------------------------------------------------------------------------------------ */
void orc_compute_pyramid(const orc_extractor *e, const uint8_t *img, int cols, int rows, size_t stride, uint8_t **levels) {
  int pw = cols, ph = rows;
  for (int y = 0; y < rows; y++) memcpy(levels[0] + (size_t)y * cols, img + (size_t)y * stride, (size_t)cols);
  for (int l = 1; l < e->nlevels; l++) {
    int lw, lh;
    orc_level_size(e, l, cols, rows, &lw, &lh);
    orc_resize_linear_u8(levels[l - 1], pw, ph, (size_t)pw, levels[l], lw, lh, (size_t)lw);
    pw = lw; ph = lh;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* X8 operator(), ORBextractor.cc:1071-1184                                                     */
/* ------------------------------------------------------------------------------------------ */
int orc_extract(const orc_extractor *e, const uint8_t *img, int rows, int cols, size_t stride, int lap0, int lap1,
                orc_keypoint *kps, uint8_t *desc, int cap, int *n_out) {
  *n_out = 0;
  if (!img || rows <= 0 || cols <= 0) return -1;
  const int nl = e->nlevels;
  uint8_t *levels[ORC_MAX_LEVELS];
  int lw[ORC_MAX_LEVELS], lh[ORC_MAX_LEVELS];
  for (int l = 0; l < nl; l++) {
    orc_level_size(e, l, cols, rows, &lw[l], &lh[l]);
    levels[l] = (uint8_t *)malloc((size_t)lw[l] * lh[l] + 1);
  }
  orc_compute_pyramid(e, img, cols, rows, stride, levels);

  /* ComputeKeyPointsOctTree */
  float *lvl_kp[ORC_MAX_LEVELS];   /* x, y, response, angle per keypoint (level coordinates) */
  int lvl_n[ORC_MAX_LEVELS];
  int nkeypoints = 0;
  for (int l = 0; l < nl; l++) {
    const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
    const int maxBorderX = lw[l] - EDGE_THRESHOLD + 3, maxBorderY = lh[l] - EDGE_THRESHOLD + 3;
    int candcap = (lw[l] * lh[l]) / 4 + 16;
    float *cand = (float *)malloc(sizeof(float) * 3 * (size_t)candcap);
    int nc = 0;
    if (maxBorderX > minBorderX && maxBorderY > minBorderY)
      nc = orc_level_candidates(e, levels[l], lw[l], lh[l], (size_t)lw[l], cand, candcap);
    int outcap = nc + 8;
    float *sel = (float *)malloc(sizeof(float) * 3 * (size_t)outcap);
    int ns = 0;
    if (nc > 0)
      ns = orc_distribute_octtree(cand, nc, minBorderX, maxBorderX, minBorderY, maxBorderY, e->mnFeaturesPerLevel[l], sel, outcap);
    lvl_kp[l] = (float *)malloc(sizeof(float) * 4 * (size_t)(ns + 1));
    for (int i = 0; i < ns; i++) {
      lvl_kp[l][4 * i] = sel[3 * i] + (float)minBorderX;
      lvl_kp[l][4 * i + 1] = sel[3 * i + 1] + (float)minBorderY;
      lvl_kp[l][4 * i + 2] = sel[3 * i + 2];
    }
    lvl_n[l] = ns;
    nkeypoints += ns;
    free(cand);
    free(sel);
  }
  for (int l = 0; l < nl; l++)
    for (int i = 0; i < lvl_n[l]; i++)
      lvl_kp[l][4 * i + 3] = orc_ic_angle(e, levels[l], (size_t)lw[l], lvl_kp[l][4 * i], lvl_kp[l][4 * i + 1]);

  *n_out = nkeypoints;
  int monoIndex = 0, stereoIndex = nkeypoints - 1;
  if (nkeypoints <= cap) {
    for (int l = 0; l < nl; l++) {
      if (lvl_n[l] == 0) continue;
      uint8_t *blur = (uint8_t *)malloc((size_t)lw[l] * lh[l]);
      orc_gaussian_blur7(levels[l], lw[l], lh[l], (size_t)lw[l], blur, (size_t)lw[l]);
      const int scaledPatchSize = (int)((float)PATCH_SIZE * e->mvScaleFactor[l]); /* :862 */
      float scale = e->mvScaleFactor[l];
      for (int i = 0; i < lvl_n[l]; i++) {
        float x = lvl_kp[l][4 * i], y = lvl_kp[l][4 * i + 1];
        uint8_t d[32];
        orc_compute_descriptor(blur, (size_t)lw[l], x, y, lvl_kp[l][4 * i + 3], d);
        orc_keypoint kp;
        kp.x = x; kp.y = y;
        kp.size = (float)scaledPatchSize;
        kp.angle = lvl_kp[l][4 * i + 3];
        kp.response = lvl_kp[l][4 * i + 2];
        kp.octave = l;
        kp.class_id = -1;
        if (l != 0) { kp.x *= scale; kp.y *= scale; }
        int dst;
        if (kp.x >= (float)lap0 && kp.x <= (float)lap1) dst = stereoIndex--;
        else dst = monoIndex++;
        kps[dst] = kp;
        memcpy(desc + 32 * (size_t)dst, d, 32);
      }
      free(blur);
    }
  } else {
    monoIndex = -2; /* capacity too small */
  }
  for (int l = 0; l < nl; l++) { free(levels[l]); free(lvl_kp[l]); }
  return monoIndex;
}

/* ------------------------------------------------------------------------------------------ */
/* M1 DescriptorDistance, ORBmatcher.cc:2463-2483                                               */
/* ------------------------------------------------------------------------------------------ */
int orc_descriptor_distance(const uint8_t *a, const uint8_t *b) {
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t pa, pb;
    memcpy(&pa, a + 4 * i, 4);
    memcpy(&pb, b + 4 * i, 4);
    uint32_t v = pa ^ pb;
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (int)((((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24);
  }
  return dist;
}

/* ------------------------------------------------------------------------------------------ */
/* G1 grid, Frame.cc:379-380, 434-465, 744-825                                                  */
/* ------------------------------------------------------------------------------------------ */
static int pos_in_grid(const orc_frame *f, float x, float y, int *px, int *py) {
  *px = (int)roundf((x - f->mnMinX) * f->mfGridElementWidthInv);
  *py = (int)roundf((y - f->mnMinY) * f->mfGridElementHeightInv);
  if (*px < 0 || *px >= ORC_GRID_COLS || *py < 0 || *py >= ORC_GRID_ROWS) return 0;
  return 1;
}

void orc_frame_init(orc_frame *f, int N, const float *kx, const float *ky, const int32_t *octave, const float *angle,
                    const uint8_t *desc, const float *uRight, float minX, float maxX, float minY, float maxY,
                    const float *scaleFactors, int nlevels) {
  memset(f, 0, sizeof(*f));
  f->N = N; f->kx = kx; f->ky = ky; f->octave = octave; f->angle = angle; f->desc = desc; f->uRight = uRight;
  f->mnMinX = minX; f->mnMaxX = maxX; f->mnMinY = minY; f->mnMaxY = maxY;
  f->mfGridElementWidthInv = (float)ORC_GRID_COLS / (maxX - minX);
  f->mfGridElementHeightInv = (float)ORC_GRID_ROWS / (maxY - minY);
  f->mvScaleFactors = scaleFactors; f->nlevels = nlevels;
  const int nc = ORC_GRID_COLS * ORC_GRID_ROWS;
  int *cnt = (int *)calloc((size_t)nc + 1, sizeof(int));
  int *cellof = (int *)malloc(sizeof(int) * (size_t)(N > 0 ? N : 1));
  for (int i = 0; i < N; i++) {
    int px, py;
    if (pos_in_grid(f, kx[i], ky[i], &px, &py)) { cellof[i] = px * ORC_GRID_ROWS + py; cnt[cellof[i]]++; }
    else cellof[i] = -1;
  }
  f->cell_start[0] = 0;
  for (int c = 0; c < nc; c++) f->cell_start[c + 1] = f->cell_start[c] + cnt[c];
  f->cell_idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
  memset(cnt, 0, sizeof(int) * (size_t)nc);
  for (int i = 0; i < N; i++)
    if (cellof[i] >= 0) f->cell_idx[f->cell_start[cellof[i]] + cnt[cellof[i]]++] = i;
  free(cnt);
  free(cellof);
}

void orc_frame_free(orc_frame *f) { free(f->cell_idx); f->cell_idx = NULL; }

int orc_get_features_in_area(const orc_frame *f, float x, float y, float r, int minLevel, int maxLevel, int32_t *out) {
  int n = 0;
  float factorX = r, factorY = r;
  int nMinCellX = (int)floorf((x - f->mnMinX - factorX) * f->mfGridElementWidthInv);
  if (nMinCellX < 0) nMinCellX = 0;
  if (nMinCellX >= ORC_GRID_COLS) return 0;
  int nMaxCellX = (int)ceilf((x - f->mnMinX + factorX) * f->mfGridElementWidthInv);
  if (nMaxCellX > ORC_GRID_COLS - 1) nMaxCellX = ORC_GRID_COLS - 1;
  if (nMaxCellX < 0) return 0;
  int nMinCellY = (int)floorf((y - f->mnMinY - factorY) * f->mfGridElementHeightInv);
  if (nMinCellY < 0) nMinCellY = 0;
  if (nMinCellY >= ORC_GRID_ROWS) return 0;
  int nMaxCellY = (int)ceilf((y - f->mnMinY + factorY) * f->mfGridElementHeightInv);
  if (nMaxCellY > ORC_GRID_ROWS - 1) nMaxCellY = ORC_GRID_ROWS - 1;
  if (nMaxCellY < 0) return 0;
  const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
  for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
    for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
      int c = ix * ORC_GRID_ROWS + iy;
      for (int j = f->cell_start[c]; j < f->cell_start[c + 1]; j++) {
        int idx = f->cell_idx[j];
        if (bCheckLevels) {
          if (f->octave[idx] < minLevel) continue;
          if (maxLevel >= 0)
            if (f->octave[idx] > maxLevel) continue;
        }
        const float distx = f->kx[idx] - x;
        const float disty = f->ky[idx] - y;
        if (fabsf(distx) < factorX && fabsf(disty) < factorY) out[n++] = idx;
      }
    }
  }
  return n;
}

float orc_radius_by_viewing_cos(float viewCos) {
  if ((double)viewCos > 0.998) return 2.5f; /* float compared with a double literal, ORBmatcher.cc:218 */
  else return 4.0f;
}

/* ------------------------------------------------------------------------------------------ */
/* M2, ORBmatcher.cc:44-143 (mono / rectified-stereo half; Nleft == -1)                         */
/* ------------------------------------------------------------------------------------------ */
int orc_search_by_projection_mp(orc_frame *f, int nq, const uint8_t *in_view, const uint8_t *qdesc, const float *projX,
                                const float *projY, const float *projXR, const float *viewCos, const int32_t *level,
                                const uint8_t *qobs, float th, float nnratio, int32_t *slot, uint8_t *slot_obs,
                                int32_t *match_of_query) {
  int nmatches = 0;
  const int bFactor = ((double)th != 1.0);
  int32_t *vIndices = (int32_t *)malloc(sizeof(int32_t) * (size_t)(f->N > 0 ? f->N : 1));
  for (int q = 0; q < nq; q++) {
    if (match_of_query) match_of_query[q] = -1;
    if (!in_view[q]) continue;
    const int nPredictedLevel = level[q];
    float r = orc_radius_by_viewing_cos(viewCos[q]);
    if (bFactor) r *= th;
    int nv = orc_get_features_in_area(f, projX[q], projY[q], r * f->mvScaleFactors[nPredictedLevel], nPredictedLevel - 1,
                                      nPredictedLevel, vIndices);
    if (nv == 0) continue;
    const uint8_t *MPdescriptor = qdesc + 32 * (size_t)q;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int k = 0; k < nv; k++) {
      const int idx = vIndices[k];
      if (slot[idx] >= 0)
        if (slot_obs[idx]) continue;
      if (f->uRight && f->uRight[idx] > 0) {
        const float er = fabsf(projXR[q] - f->uRight[idx]);
        if (er > r * f->mvScaleFactors[nPredictedLevel]) continue;
      }
      const int dist = orc_descriptor_distance(MPdescriptor, f->desc + 32 * (size_t)idx);
      if (dist < bestDist) {
        bestDist2 = bestDist; bestDist = dist;
        bestLevel2 = bestLevel; bestLevel = f->octave[idx];
        bestIdx = idx;
      } else if (dist < bestDist2) {
        bestLevel2 = f->octave[idx];
        bestDist2 = dist;
      }
    }
    if (bestDist <= 100 /* TH_HIGH */) {
      if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
      slot[bestIdx] = q;
      slot_obs[bestIdx] = qobs ? qobs[q] : 1;
      if (match_of_query) match_of_query[q] = bestIdx;
      nmatches++;
    }
  }
  free(vIndices);
  return nmatches;
}

int orc_search_by_projection_win(orc_frame *f, int nq, const uint8_t *in_view, const uint8_t *qdesc, const float *u,
                                 const float *v, const float *radius, const int32_t *minLevel, const int32_t *maxLevel,
                                 const uint8_t *qobs, float nnratio, int th_high, int mode_second, int32_t *slot,
                                 uint8_t *slot_obs, int32_t *match_of_query, int32_t *best_dist_out) {
  int nmatches = 0;
  int32_t *vIndices = (int32_t *)malloc(sizeof(int32_t) * (size_t)(f->N > 0 ? f->N : 1));
  for (int q = 0; q < nq; q++) {
    if (match_of_query) match_of_query[q] = -1;
    if (best_dist_out) best_dist_out[q] = 256;
    if (in_view && !in_view[q]) continue;
    int nv = orc_get_features_in_area(f, u[q], v[q], radius[q], minLevel[q], maxLevel[q], vIndices);
    if (nv == 0) continue;
    const uint8_t *d = qdesc + 32 * (size_t)q;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int k = 0; k < nv; k++) {
      const int idx = vIndices[k];
      if (slot[idx] >= 0)
        if (slot_obs[idx]) continue;
      const int dist = orc_descriptor_distance(d, f->desc + 32 * (size_t)idx);
      if (dist < bestDist) {
        bestDist2 = bestDist; bestDist = dist;
        bestLevel2 = bestLevel; bestLevel = f->octave[idx];
        bestIdx = idx;
      } else if (dist < bestDist2) {
        bestLevel2 = f->octave[idx];
        bestDist2 = dist;
      }
    }
    if (best_dist_out) best_dist_out[q] = bestDist <= th_high ? bestDist : 256; /* distance of the best candidate if within the threshold */
    if (bestDist <= th_high) {
      if (mode_second && bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
      slot[bestIdx] = q;
      slot_obs[bestIdx] = qobs ? qobs[q] : 1;
      if (match_of_query) match_of_query[q] = bestIdx;
      nmatches++;
    }
  }
  free(vIndices);
  return nmatches;
}

/* ------------------------------------------------------------------------------------------ */
/* C1 / C2 projections                                                                          */
/* ------------------------------------------------------------------------------------------ */
void orc_project(int type, const float *p, float X, float Y, float Z, float *u, float *v) {
  if (type == 0) { /* Pinhole.cpp:46-49 */
    *u = p[0] * X / Z + p[2];
    *v = p[1] * Y / Z + p[3];
  } else { /* KannalaBrandt8.cpp:29-45 */
    const float x2_plus_y2 = X * X + Y * Y;
    const float theta = atan2f(sqrtf(x2_plus_y2), Z);
    const float psi = atan2f(Y, X);
    const float theta2 = theta * theta;
    const float theta3 = theta * theta2;
    const float theta5 = theta3 * theta2;
    const float theta7 = theta5 * theta2;
    const float theta9 = theta7 * theta2;
    const float r = theta + p[4] * theta3 + p[5] * theta5 + p[6] * theta7 + p[7] * theta9;
    /* Un-suffixed cos / sin on a float.  KannalaBrandt8.cpp has no `using namespace std`, so the overload depends on whether
     * the C++ <math.h> wrapper (libstdc++ >= 6: `using std::cos;` -> float overloads in the global namespace) is in the
     * translation unit.  It is: KannalaBrandt8.h -> TwoViewReconstruction.h:22 includes <opencv2/opencv.hpp>, whose FLANN
     * headers include <math.h>.  The same closure is what makes `log(ratio)` in MapPoint::PredictScale (MapPoint.cc:578, :595;
     * restated with logf below) a float call.  Hence cosf / sinf and float products.  (With GCC 5 or a libc++ without the
     * wrapper this would be ::cos(double): an ambiguity of the reference's build, stated in DESIGN.md.) */
    *u = p[0] * r * cosf(psi) + p[2];
    *v = p[1] * r * sinf(psi) + p[3];
  }
}

/* ------------------------------------------------------------------------------------------ */
/* M7 ComputeThreeMaxima, ORBmatcher.cc:2416-2458                                               */
/* ------------------------------------------------------------------------------------------ */
void orc_three_maxima(const int *histo, int L, int *ind1, int *ind2, int *ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  *ind1 = *ind2 = *ind3 = -1;
  for (int i = 0; i < L; i++) {
    const int s = histo[i];
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
    else if (s > max3) { max3 = s; *ind3 = i; }
  }
  if ((float)max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
  else if ((float)max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

/* ------------------------------------------------------------------------------------------ */
/* M3, ORBmatcher.cc:2027-2289 (Nleft == -1 path)                                               */
/* cv::Mat products restated per SURVEY.md A.8: the 3x3*3x1 `A*B+C` MatExpr is one cv::gemm whose
 * small-matrix float path forms a0*b0+a1*b1+a2*b2 in float then adds C; the transposed,
 * alpha=-1 product goes through the generic path that accumulates in double. [OPENCV-UNVERIFIED] */
/* ------------------------------------------------------------------------------------------ */
static void mat3_mul_add(const float *R /*row-major 3x3, row stride rs*/, int rs, const float *x, const float *t, float *out) {
  for (int i = 0; i < 3; i++) {
    float t0 = R[i * rs + 0] * x[0] + R[i * rs + 1] * x[1] + R[i * rs + 2] * x[2];
    out[i] = (float)((double)t0 * 1.0 + (double)t[i] * 1.0);
  }
}

int orc_search_by_projection_ff(orc_frame *cur, int nLast, const uint8_t *has_mp, const float *Xw, const uint8_t *mpdesc,
                                const int32_t *lastOctave, const float *lastAngle, const uint8_t *qobs, const float *Tcw,
                                const float *Tlw, int camType, const float *camParams, float mb, float mbf, float th,
                                int bMono, int checkOri, int32_t *slot, uint8_t *slot_obs) {
  int nmatches = 0;
  const int HISTO_LENGTH = 30;
  int *rotHist[30];
  int rotN[30];
  for (int i = 0; i < HISTO_LENGTH; i++) { rotHist[i] = (int *)malloc(sizeof(int) * (size_t)(nLast + 1)); rotN[i] = 0; }
  const float factor = 1.0f / HISTO_LENGTH;
  float tcw[3] = {Tcw[3], Tcw[7], Tcw[11]};
  float tlw[3] = {Tlw[3], Tlw[7], Tlw[11]};
  /* twc = -Rcw.t()*tcw : generic gemm path, double accumulation, alpha = -1 */
  float twc[3];
  for (int i = 0; i < 3; i++) {
    double s = 0;
    for (int k = 0; k < 3; k++) s += (double)Tcw[k * 4 + i] * (double)tcw[k];
    twc[i] = (float)(s * -1.0);
  }
  float tlc[3];
  mat3_mul_add(Tlw, 4, twc, tlw, tlc);
  const int bForward = tlc[2] > mb && !bMono;
  const int bBackward = -tlc[2] > mb && !bMono;
  int32_t *vIndices2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(cur->N > 0 ? cur->N : 1));
  for (int i = 0; i < nLast; i++) {
    if (!has_mp[i]) continue;
    float x3Dc[3];
    mat3_mul_add(Tcw, 4, Xw + 3 * i, tcw, x3Dc);
    const float invzc = (float)(1.0 / (double)x3Dc[2]); /* `1.0/x3Dc.at<float>(2)` is double, stored to float */
    if (invzc < 0) continue;
    float uvx, uvy;
    orc_project(camType, camParams, x3Dc[0], x3Dc[1], x3Dc[2], &uvx, &uvy);
    if (uvx < cur->mnMinX || uvx > cur->mnMaxX) continue;
    if (uvy < cur->mnMinY || uvy > cur->mnMaxY) continue;
    int nLastOctave = lastOctave[i];
    float radius = th * cur->mvScaleFactors[nLastOctave];
    int nv;
    if (bForward) nv = orc_get_features_in_area(cur, uvx, uvy, radius, nLastOctave, -1, vIndices2);
    else if (bBackward) nv = orc_get_features_in_area(cur, uvx, uvy, radius, 0, nLastOctave, vIndices2);
    else nv = orc_get_features_in_area(cur, uvx, uvy, radius, nLastOctave - 1, nLastOctave + 1, vIndices2);
    if (nv == 0) continue;
    const uint8_t *dMP = mpdesc + 32 * (size_t)i;
    int bestDist = 256, bestIdx2 = -1;
    for (int k = 0; k < nv; k++) {
      const int i2 = vIndices2[k];
      if (slot[i2] >= 0)
        if (slot_obs[i2]) continue;
      if (cur->uRight && cur->uRight[i2] > 0) {
        const float ur = uvx - mbf * invzc;
        const float er = fabsf(ur - cur->uRight[i2]);
        if (er > radius) continue;
      }
      const int dist = orc_descriptor_distance(dMP, cur->desc + 32 * (size_t)i2);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= 100 /* TH_HIGH */) {
      slot[bestIdx2] = i;
      slot_obs[bestIdx2] = qobs ? qobs[i] : 1;
      nmatches++;
      if (checkOri) {
        float rot = lastAngle[i] - cur->angle[bestIdx2];
        if ((double)rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin][rotN[bin]++] = bestIdx2;
      }
    }
  }
  if (checkOri) {
    int ind1, ind2, ind3;
    orc_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i != ind1 && i != ind2 && i != ind3) {
        for (int j = 0; j < rotN[i]; j++) {
          slot[rotHist[i][j]] = -1;
          slot_obs[rotHist[i][j]] = 0;
          nmatches--;
        }
      }
    }
  }
  for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
  free(vIndices2);
  return nmatches;
}

/* ------------------------------------------------------------------------------------------ */
/* M2 with a fisheye-stereo frame (Nleft != -1), ORBmatcher.cc:44-214 complete: left half on    */
/* mGrid / mvKeys, right half (:145-211) on mGridRight / mvKeysRight, the partner writes through */
/* mvLeftToRightMatch / mvRightToLeftMatch (:128-132, :199-203) and the `continue` of :125 that  */
/* also skips the right half of the same map point.  slot / slot_obs hold Nleft + Nright entries */
/* (F.mvpMapPoints); right keypoint j is entry Nleft + j.                                        */
/* ------------------------------------------------------------------------------------------ */
int orc_search_by_projection_mp_fisheye(orc_frame *fl, orc_frame *fr, const int32_t *leftToRight,
                                        const int32_t *rightToLeft, int nmp, const uint8_t *in_view,
                                        const uint8_t *in_view_r, const uint8_t *qdesc, const float *projX,
                                        const float *projY, const float *viewCos, const int32_t *level,
                                        const float *projXR, const float *projYR, const float *viewCosR,
                                        const int32_t *levelR, const uint8_t *qobs, float th, float nnratio,
                                        int32_t *slot, uint8_t *slot_obs, int32_t *match_left, int32_t *match_right) {
  int nmatches = 0;
  const int Nleft = fl->N;
  const int bFactor = ((double)th != 1.0);
  const int cap = (fl->N > fr->N ? fl->N : fr->N) + 1;
  int32_t *vIndices = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
  for (int q = 0; q < nmp; q++) {
    if (match_left) match_left[q] = -1;
    if (match_right) match_right[q] = -1;
    if (!in_view[q] && !in_view_r[q]) continue;
    const uint8_t ob = qobs ? qobs[q] : 1;
    const uint8_t *MPdescriptor = qdesc + 32 * (size_t)q;
    if (in_view[q]) {
      const int nPredictedLevel = level[q];
      float r = orc_radius_by_viewing_cos(viewCos[q]);
      if (bFactor) r *= th;
      int nv = orc_get_features_in_area(fl, projX[q], projY[q], r * fl->mvScaleFactors[nPredictedLevel],
                                        nPredictedLevel - 1, nPredictedLevel, vIndices);
      if (nv != 0) {
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int k = 0; k < nv; k++) {
          const int idx = vIndices[k];
          if (slot[idx] >= 0)
            if (slot_obs[idx]) continue;
          /* F.Nleft != -1: no mvuRight test (:93) */
          const int dist = orc_descriptor_distance(MPdescriptor, fl->desc + 32 * (size_t)idx);
          if (dist < bestDist) {
            bestDist2 = bestDist; bestDist = dist;
            bestLevel2 = bestLevel; bestLevel = fl->octave[idx];
            bestIdx = idx;
          } else if (dist < bestDist2) {
            bestLevel2 = fl->octave[idx];
            bestDist2 = dist;
          }
        }
        if (bestDist <= 100 /* TH_HIGH */) {
          if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue; /* :125, skips the right half too */
          slot[bestIdx] = q;
          slot_obs[bestIdx] = ob;
          if (match_left) match_left[q] = bestIdx;
          if (leftToRight[bestIdx] != -1) { /* :128-132, unconditional overwrite of the partner's slot */
            slot[leftToRight[bestIdx] + Nleft] = q;
            slot_obs[leftToRight[bestIdx] + Nleft] = ob;
            nmatches++;
          }
          nmatches++;
        }
      }
    }
    if (in_view_r[q]) {
      const int nPredictedLevel = levelR[q];
      if (nPredictedLevel != -1) {
        float r = orc_radius_by_viewing_cos(viewCosR[q]); /* :148, not multiplied by th */
        int nv = orc_get_features_in_area(fr, projXR[q], projYR[q], r * fr->mvScaleFactors[nPredictedLevel],
                                          nPredictedLevel - 1, nPredictedLevel, vIndices);
        if (nv == 0) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int k = 0; k < nv; k++) {
          const int idx = vIndices[k];
          if (slot[idx + Nleft] >= 0)
            if (slot_obs[idx + Nleft]) continue;
          const int dist = orc_descriptor_distance(MPdescriptor, fr->desc + 32 * (size_t)idx);
          if (dist < bestDist) {
            bestDist2 = bestDist; bestDist = dist;
            bestLevel2 = bestLevel; bestLevel = fr->octave[idx];
            bestIdx = idx;
          } else if (dist < bestDist2) {
            bestLevel2 = fr->octave[idx];
            bestDist2 = dist;
          }
        }
        if (bestDist <= 100 /* TH_HIGH */) {
          if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
          if (rightToLeft[bestIdx] != -1) { /* :199-203 */
            slot[rightToLeft[bestIdx]] = q;
            slot_obs[rightToLeft[bestIdx]] = ob;
            nmatches++;
          }
          slot[bestIdx + Nleft] = q;
          slot_obs[bestIdx + Nleft] = ob;
          if (match_right) match_right[q] = bestIdx;
          nmatches++;
        }
      }
    }
  }
  free(vIndices);
  return nmatches;
}

/* ------------------------------------------------------------------------------------------ */
/* M3 with a fisheye-stereo current frame, ORBmatcher.cc:2027-2289 complete: the left search and  */
/* the extra right-camera pass (:2189-2256, x3Dr = Rrl * x3Dc + trl projected with the same      */
/* camera); `if(vIndices2.empty()) continue;` (:2126) of the left search skips the right pass.    */
/* curAngle[]: angles of mvKeys (left, Nleft) followed by mvKeysRight (right).                    */
/* ------------------------------------------------------------------------------------------ */
int orc_search_by_projection_ff_fisheye(orc_frame *cl, orc_frame *cr, int nLast, const uint8_t *has_mp, const float *Xw,
                                        const uint8_t *mpdesc, const int32_t *lastOctave, const float *lastAngle,
                                        const uint8_t *qobs, const float *Tcw, const float *Tlw, const float *Trl,
                                        int camType, const float *camParams, float mb, float th, int bMono, int checkOri,
                                        int32_t *slot, uint8_t *slot_obs) {
  int nmatches = 0;
  const int HISTO_LENGTH = 30;
  const int Nleft = cl->N;
  int *rotHist[30];
  int rotN[30];
  for (int i = 0; i < HISTO_LENGTH; i++) { rotHist[i] = (int *)malloc(sizeof(int) * (size_t)(2 * nLast + 1)); rotN[i] = 0; }
  const float factor = 1.0f / HISTO_LENGTH;
  float tcw[3] = {Tcw[3], Tcw[7], Tcw[11]};
  float tlw[3] = {Tlw[3], Tlw[7], Tlw[11]};
  float trl[3] = {Trl[3], Trl[7], Trl[11]};
  float twc[3];
  for (int i = 0; i < 3; i++) {
    double s = 0;
    for (int k = 0; k < 3; k++) s += (double)Tcw[k * 4 + i] * (double)tcw[k];
    twc[i] = (float)(s * -1.0);
  }
  float tlc[3];
  mat3_mul_add(Tlw, 4, twc, tlw, tlc);
  const int bForward = tlc[2] > mb && !bMono;
  const int bBackward = -tlc[2] > mb && !bMono;
  const int cap = (cl->N > cr->N ? cl->N : cr->N) + 1;
  int32_t *vIndices2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
  for (int i = 0; i < nLast; i++) {
    if (!has_mp[i]) continue;
    float x3Dc[3];
    mat3_mul_add(Tcw, 4, Xw + 3 * i, tcw, x3Dc);
    const float invzc = (float)(1.0 / (double)x3Dc[2]);
    if (invzc < 0) continue;
    float uvx, uvy;
    orc_project(camType, camParams, x3Dc[0], x3Dc[1], x3Dc[2], &uvx, &uvy);
    if (uvx < cl->mnMinX || uvx > cl->mnMaxX) continue;
    if (uvy < cl->mnMinY || uvy > cl->mnMaxY) continue;
    const int nLastOctave = lastOctave[i];
    const float radius = th * cl->mvScaleFactors[nLastOctave];
    const uint8_t ob = qobs ? qobs[i] : 1;
    const uint8_t *dMP = mpdesc + 32 * (size_t)i;
    int nv;
    if (bForward) nv = orc_get_features_in_area(cl, uvx, uvy, radius, nLastOctave, -1, vIndices2);
    else if (bBackward) nv = orc_get_features_in_area(cl, uvx, uvy, radius, 0, nLastOctave, vIndices2);
    else nv = orc_get_features_in_area(cl, uvx, uvy, radius, nLastOctave - 1, nLastOctave + 1, vIndices2);
    if (nv == 0) continue; /* :2126 - the right pass of this map point is skipped as well */
    {
      int bestDist = 256, bestIdx2 = -1;
      for (int k = 0; k < nv; k++) {
        const int i2 = vIndices2[k];
        if (slot[i2] >= 0)
          if (slot_obs[i2]) continue;
        /* CurrentFrame.Nleft != -1: no mvuRight test (:2139) */
        const int dist = orc_descriptor_distance(dMP, cl->desc + 32 * (size_t)i2);
        if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
      }
      if (bestDist <= 100 /* TH_HIGH */) {
        slot[bestIdx2] = i;
        slot_obs[bestIdx2] = ob;
        nmatches++;
        if (checkOri) {
          float rot = lastAngle[i] - cl->angle[bestIdx2];
          if ((double)rot < 0.0) rot += 360.0f;
          int bin = (int)roundf(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          rotHist[bin][rotN[bin]++] = bestIdx2;
        }
      }
    }
    {
      float x3Dr[3];
      mat3_mul_add(Trl, 4, x3Dc, trl, x3Dr); /* :2190 */
      float uvxr, uvyr;
      orc_project(camType, camParams, x3Dr[0], x3Dr[1], x3Dr[2], &uvxr, &uvyr);
      if (bForward) nv = orc_get_features_in_area(cr, uvxr, uvyr, radius, nLastOctave, -1, vIndices2);
      else if (bBackward) nv = orc_get_features_in_area(cr, uvxr, uvyr, radius, 0, nLastOctave, vIndices2);
      else nv = orc_get_features_in_area(cr, uvxr, uvyr, radius, nLastOctave - 1, nLastOctave + 1, vIndices2);
      int bestDist = 256, bestIdx2 = -1;
      for (int k = 0; k < nv; k++) {
        const int i2 = vIndices2[k];
        if (slot[i2 + Nleft] >= 0)
          if (slot_obs[i2 + Nleft]) continue;
        const int dist = orc_descriptor_distance(dMP, cr->desc + 32 * (size_t)i2);
        if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
      }
      if (bestDist <= 100 /* TH_HIGH */) {
        slot[bestIdx2 + Nleft] = i;
        slot_obs[bestIdx2 + Nleft] = ob;
        nmatches++;
        if (checkOri) {
          float rot = lastAngle[i] - cr->angle[bestIdx2];
          if ((double)rot < 0.0) rot += 360.0f;
          int bin = (int)roundf(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          rotHist[bin][rotN[bin]++] = bestIdx2 + Nleft;
        }
      }
    }
  }
  if (checkOri) {
    int ind1, ind2, ind3;
    orc_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i != ind1 && i != ind2 && i != ind3) {
        for (int j = 0; j < rotN[i]; j++) {
          slot[rotHist[i][j]] = -1;
          slot_obs[rotHist[i][j]] = 0;
          nmatches--;
        }
      }
    }
  }
  for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
  free(vIndices2);
  return nmatches;
}

/* ------------------------------------------------------------------------------------------ */
/* N1: Frame::ComputeStereoMatches, Frame.cc:901-1079 (rectified stereo).                          */
/* levelsL / levelsR: pyramid ROIs (mvImagePyramid of the two extractors), level l has size          */
/* orc_level_size(l) and stride = its width.  kx/ky/oct = mvKeys / mvKeysRight.                       */
/* cv::Mat::rowRange / colRange throw outside the matrix; such keypoints are skipped here (defined). */
/* With no accepted match the reference indexes an empty vector (:1062); defined as "nothing to do". */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int dist; int idx; } orc_distidx;
static int cmp_distidx(const void *a, const void *b) {
  const orc_distidx *x = (const orc_distidx *)a, *y = (const orc_distidx *)b;
  if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

void orc_compute_stereo_matches(const orc_extractor *e, uint8_t *const *levelsL, uint8_t *const *levelsR, int cols, int rows,
                                int N, const float *kxL, const float *kyL, const int32_t *octL, const uint8_t *descL,
                                int Nr, const float *kxR, const float *kyR, const int32_t *octR, const uint8_t *descR,
                                float mb, float mbf, float *mvuRight, float *mvDepth) {
  for (int i = 0; i < N; i++) { mvuRight[i] = -1.0f; mvDepth[i] = -1.0f; }
  const int thOrbDist = (100 + 50) / 2;
  const int nRows = rows;
  /* row table, :912-930: CSR in push order */
  int *cnt = (int *)calloc((size_t)nRows + 1, sizeof(int));
  for (int iR = 0; iR < Nr; iR++) {
    const float r = 2.0f * e->mvScaleFactor[octR[iR]];
    const int maxr = (int)ceilf(kyR[iR] + r), minr = (int)floorf(kyR[iR] - r);
    for (int yi = minr; yi <= maxr; yi++) if (yi >= 0 && yi < nRows) cnt[yi + 1]++;
  }
  for (int y = 0; y < nRows; y++) cnt[y + 1] += cnt[y];
  int *rowIdx = (int *)malloc(sizeof(int) * (size_t)(cnt[nRows] + 1));
  int *fill = (int *)calloc((size_t)nRows + 1, sizeof(int));
  for (int iR = 0; iR < Nr; iR++) {
    const float r = 2.0f * e->mvScaleFactor[octR[iR]];
    const int maxr = (int)ceilf(kyR[iR] + r), minr = (int)floorf(kyR[iR] - r);
    for (int yi = minr; yi <= maxr; yi++) if (yi >= 0 && yi < nRows) rowIdx[cnt[yi] + fill[yi]++] = iR;
  }
  const float minZ = mb, minD = 0, maxD = mbf / minZ;
  orc_distidx *vDistIdx = (orc_distidx *)malloc(sizeof(orc_distidx) * (size_t)(N + 1));
  int nDI = 0;
  for (int iL = 0; iL < N; iL++) {
    const int levelL = octL[iL];
    const float vL = kyL[iL], uL = kxL[iL];
    const int row = (int)vL;
    if (row < 0 || row >= nRows) continue;
    const int c0 = cnt[row], c1 = cnt[row + 1];
    if (c0 == c1) continue;
    const float minU = uL - maxD, maxU = uL - minD;
    if (maxU < 0) continue;
    int bestDist = 100; /* TH_HIGH */
    int bestIdxR = 0;
    const uint8_t *dL = descL + 32 * (size_t)iL;
    for (int k = c0; k < c1; k++) {
      const int iR = rowIdx[k];
      if (octR[iR] < levelL - 1 || octR[iR] > levelL + 1) continue;
      const float uR = kxR[iR];
      if (uR >= minU && uR <= maxU) {
        const int dist = orc_descriptor_distance(dL, descR + 32 * (size_t)iR);
        if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
      }
    }
    if (bestDist < thOrbDist) {
      const float uR0 = kxR[bestIdxR];
      const float scaleFactor = e->mvInvScaleFactor[levelL];
      const float scaleduL = roundf(uL * scaleFactor), scaledvL = roundf(vL * scaleFactor), scaleduR0 = roundf(uR0 * scaleFactor);
      const int w = 5, L = 5;
      int lw, lh;
      orc_level_size(e, levelL, cols, rows, &lw, &lh);
      const int cuL = (int)scaleduL, cvL = (int)scaledvL, cuR = (int)scaleduR0;
      if (cvL - w < 0 || cvL + w + 1 > lh || cuL - w < 0 || cuL + w + 1 > lw) continue;   /* rowRange/colRange would throw */
      const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
      if (iniu < 0 || endu >= (float)lw) continue;
      if (cuR - L - w < 0) continue;                                                     /* colRange would throw */
      const uint8_t *IL = levelsL[levelL], *IRm = levelsR[levelL];
      const int cL = IL[(size_t)cvL * lw + cuL];
      int bestSad = 2147483647, bestincR = 0;
      float vDists[11];
      for (int incR = -L; incR <= L; incR++) {
        const int cR = IRm[(size_t)cvL * lw + cuR + incR];
        int sad = 0;
        for (int py = -w; py <= w; py++)
          for (int px = -w; px <= w; px++) {
            const int a = (int)IL[(size_t)(cvL + py) * lw + cuL + px] - cL;
            const int b = (int)IRm[(size_t)(cvL + py) * lw + cuR + incR + px] - cR;
            sad += abs(a - b);
          }
        const float dist = (float)sad;
        if (dist < (float)bestSad) { bestSad = (int)dist; bestincR = incR; }
        vDists[L + incR] = dist;
      }
      if (bestincR == -L || bestincR == L) continue;
      const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
      const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
      if (deltaR < -1 || deltaR > 1) continue;
      float bestuR = e->mvScaleFactor[levelL] * ((float)scaleduR0 + (float)bestincR + deltaR);
      float disparity = (uL - bestuR);
      if (disparity >= minD && disparity < maxD) {
        if (disparity <= 0) { disparity = (float)0.01; bestuR = (float)((double)uL - 0.01); }
        mvDepth[iL] = mbf / disparity;
        mvuRight[iL] = bestuR;
        vDistIdx[nDI].dist = bestSad; vDistIdx[nDI].idx = iL; nDI++;
      }
    }
  }
  if (nDI > 0) {
    qsort(vDistIdx, (size_t)nDI, sizeof(orc_distidx), cmp_distidx);
    const float median = (float)vDistIdx[nDI / 2].dist;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = nDI - 1; i >= 0; i--) {
      if ((float)vDistIdx[i].dist < thDist) break;
      mvuRight[vDistIdx[i].idx] = -1;
      mvDepth[vDistIdx[i].idx] = -1;
    }
  }
  free(cnt); free(rowIdx); free(fill); free(vDistIdx);
}

/* ------------------------------------------------------------------------------------------ */
/* N2: ORBmatcher::SearchForInitialization, ORBmatcher.cc:722-837                                 */
/* F1 side: octave / angle / descriptors of mvKeysUn; F2 = orc_frame; prevMatched[2*n1] in/out.    */
/* ------------------------------------------------------------------------------------------ */
int orc_search_for_initialization(int n1, const int32_t *octave1, const float *angle1, const uint8_t *desc1, orc_frame *F2,
                                  float *prevMatched, int windowSize, float nnratio, int checkOri, int32_t *vnMatches12) {
  int nmatches = 0;
  const int HISTO_LENGTH = 30, TH_LOW = 50;
  const int n2 = F2->N;
  for (int i = 0; i < n1; i++) vnMatches12[i] = -1;
  int *rotHist[30];
  int rotN[30];
  for (int i = 0; i < HISTO_LENGTH; i++) { rotHist[i] = (int *)malloc(sizeof(int) * (size_t)(n1 + 1)); rotN[i] = 0; }
  const float factor = 1.0f / HISTO_LENGTH;
  int *vMatchedDistance = (int *)malloc(sizeof(int) * (size_t)(n2 + 1));
  int *vnMatches21 = (int *)malloc(sizeof(int) * (size_t)(n2 + 1));
  for (int i = 0; i < n2; i++) { vMatchedDistance[i] = 2147483647; vnMatches21[i] = -1; }
  int32_t *vIndices2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 + 1));
  for (int i1 = 0; i1 < n1; i1++) {
    const int level1 = octave1[i1];
    if (level1 > 0) continue;
    const int nv = orc_get_features_in_area(F2, prevMatched[2 * i1], prevMatched[2 * i1 + 1], (float)windowSize, level1, level1, vIndices2);
    if (nv == 0) continue;
    const uint8_t *d1 = desc1 + 32 * (size_t)i1;
    int bestDist = 2147483647, bestDist2 = 2147483647, bestIdx2 = -1;
    for (int k = 0; k < nv; k++) {
      const int i2 = vIndices2[k];
      const int dist = orc_descriptor_distance(d1, F2->desc + 32 * (size_t)i2);
      if (vMatchedDistance[i2] <= dist) continue;
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
      else if (dist < bestDist2) bestDist2 = dist;
    }
    if (bestDist <= TH_LOW) {
      if ((float)bestDist < (float)bestDist2 * nnratio) {
        if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
        vnMatches12[i1] = bestIdx2;
        vnMatches21[bestIdx2] = i1;
        vMatchedDistance[bestIdx2] = bestDist;
        nmatches++;
        if (checkOri) {
          float rot = angle1[i1] - F2->angle[bestIdx2];
          if ((double)rot < 0.0) rot += 360.0f;
          int bin = (int)roundf(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          rotHist[bin][rotN[bin]++] = i1;
        }
      }
    }
  }
  if (checkOri) {
    int ind1, ind2, ind3;
    orc_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int j = 0; j < rotN[i]; j++) {
        const int idx1 = rotHist[i][j];
        if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
      }
    }
  }
  for (int i1 = 0; i1 < n1; i1++)
    if (vnMatches12[i1] >= 0) { prevMatched[2 * i1] = F2->kx[vnMatches12[i1]]; prevMatched[2 * i1 + 1] = F2->ky[vnMatches12[i1]]; }
  for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
  free(vMatchedDistance); free(vnMatches21); free(vIndices2);
  return nmatches;
}

/* ------------------------------------------------------------------------------------------ */
/* M6, ORBmatcher.cc:981-1222 + Pinhole::epipolarConstrain, Pinhole.cpp:143-165                 */
/* cv::Mat algebra restated per SURVEY.md A.8 [OPENCV-UNVERIFIED]:                              */
/*  - 3x3 * 3x3 / 3x3 * 3x1 products without flags: float dot products (a0*b0+a1*b1+a2*b2 in    */
/*    float), then (float)(t*alpha + c*beta) in double;                                         */
/*  - products with a transposed operand or a scalar factor go through the generic gemm that    */
/*    accumulates in double;                                                                    */
/*  - cv::invert of a 3x3 CV_32F matrix: cofactor formula evaluated in double, stored as float. */
/* ------------------------------------------------------------------------------------------ */
static void m33_mul(const float *A, const float *B, float *D) { /* small-matrix float path */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      float t = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
      D[i * 3 + j] = (float)((double)t * 1.0);
    }
}
static void m33_inv(const float *S, float *D) {
#define Sf(y, x) ((double)S[(y) * 3 + (x)])
  double d = Sf(0, 0) * (Sf(1, 1) * Sf(2, 2) - Sf(1, 2) * Sf(2, 1)) - Sf(0, 1) * (Sf(1, 0) * Sf(2, 2) - Sf(1, 2) * Sf(2, 0)) +
             Sf(0, 2) * (Sf(1, 0) * Sf(2, 1) - Sf(1, 1) * Sf(2, 0));
  if (d == 0.) { memset(D, 0, 9 * sizeof(float)); return; }
  d = 1. / d;
  double t[9];
  t[0] = (Sf(1, 1) * Sf(2, 2) - Sf(1, 2) * Sf(2, 1)) * d;
  t[1] = (Sf(0, 2) * Sf(2, 1) - Sf(0, 1) * Sf(2, 2)) * d;
  t[2] = (Sf(0, 1) * Sf(1, 2) - Sf(0, 2) * Sf(1, 1)) * d;
  t[3] = (Sf(1, 2) * Sf(2, 0) - Sf(1, 0) * Sf(2, 2)) * d;
  t[4] = (Sf(0, 0) * Sf(2, 2) - Sf(0, 2) * Sf(2, 0)) * d;
  t[5] = (Sf(0, 2) * Sf(1, 0) - Sf(0, 0) * Sf(1, 2)) * d;
  t[6] = (Sf(1, 0) * Sf(2, 1) - Sf(1, 1) * Sf(2, 0)) * d;
  t[7] = (Sf(0, 1) * Sf(2, 0) - Sf(0, 0) * Sf(2, 1)) * d;
  t[8] = (Sf(0, 0) * Sf(1, 1) - Sf(0, 1) * Sf(1, 0)) * d;
#undef Sf
  for (int i = 0; i < 9; i++) D[i] = (float)t[i];
}

void orc_pinhole_F12(const float *R12, const float *t12, const float *cam1, const float *cam2, float *F12) {
  const float t12x[9] = {0, -t12[2], t12[1], t12[2], 0, -t12[0], -t12[1], t12[0], 0};        /* Pinhole.cpp:183-188 */
  const float K1t[9] = {cam1[0], 0.f, 0.f, 0.f, cam1[1], 0.f, cam1[2], cam1[3], 1.f};        /* K1.t() */
  const float K2[9] = {cam2[0], 0.f, cam2[2], 0.f, cam2[1], cam2[3], 0.f, 0.f, 1.f};
  float K1ti[9], K2i[9], A[9], B[9];
  m33_inv(K1t, K1ti);
  m33_inv(K2, K2i);
  m33_mul(K1ti, t12x, A);  /* ((K1.t().inv() * t12x) * R12) * K2.inv() */
  m33_mul(A, R12, B);
  m33_mul(B, K2i, F12);
}

static int epipolar_constrain(const float *F12, float x1, float y1, float x2, float y2, float unc) { /* Pinhole.cpp:150-164 */
  const float a = x1 * F12[0] + y1 * F12[3] + F12[6];
  const float b = x1 * F12[1] + y1 * F12[4] + F12[7];
  const float c = x1 * F12[2] + y1 * F12[5] + F12[8];
  const float num = a * x2 + b * y2 + c;
  const float den = a * a + b * b;
  if (den == 0) return 0;
  const float dsqr = num * num / den;
  return (double)dsqr < 3.84 * (double)unc;
}

/* The member with its epipolar test abstract (ORBmatcher.cc:1148: pCamera1->epipolarConstrain(...) is a virtual call; Pinhole's is   */
/* restated above, KannalaBrandt8's rests on cv::SVD and stays outside the oracle): pred(user, idx1, idx2).  ep = the epipole in image */
/* 2 as the caller's camera model projects it (:992); epipole_gate = !pKF1->mpCamera2 (:1105).                                          */
typedef int (*orc_pair_predicate)(void *user, int idx1, int idx2);
int orc_search_for_triangulation_pred(const orc_keyframe *k1, const orc_keyframe *k2, float epx, float epy, int epipole_gate,
                                      int bOnlyStereo, int bCoarse, int checkOri, orc_pair_predicate pred, void *user, int32_t *vMatches12) {
  int nmatches = 0;
  for (int i = 0; i < k1->N; i++) vMatches12[i] = -1;
  const int HISTO_LENGTH = 30;
  int *rotHist[30];
  int rotN[30];
  for (int i = 0; i < HISTO_LENGTH; i++) { rotHist[i] = (int *)malloc(sizeof(int) * (size_t)(k1->N + 1)); rotN[i] = 0; }
  const float factor = 1.0f / HISTO_LENGTH;
  int f1 = 0, f2 = 0;
  while (f1 < k1->n_nodes && f2 < k2->n_nodes) {
    if (k1->node_id[f1] == k2->node_id[f2]) {
      for (int i1 = k1->node_start[f1]; i1 < k1->node_start[f1 + 1]; i1++) {
        const int idx1 = k1->node_idx[i1];
        if (k1->has_mp[idx1]) continue;
        const int bStereo1 = k1->uRight[idx1] >= 0;
        if (bOnlyStereo && !bStereo1) continue;
        int bestDist = 50 /* TH_LOW */, bestIdx2 = -1;
        for (int i2 = k2->node_start[f2]; i2 < k2->node_start[f2 + 1]; i2++) {
          const int idx2 = k2->node_idx[i2];
          if (k2->has_mp[idx2]) continue;                 /* vbMatched2 is never set (:1027) */
          const int bStereo2 = k2->uRight[idx2] >= 0;
          if (bOnlyStereo && !bStereo2) continue;
          const int dist = orc_descriptor_distance(k1->desc + 32 * (size_t)idx1, k2->desc + 32 * (size_t)idx2);
          if (dist > 50 || dist > bestDist) continue;
          if (!bStereo1 && !bStereo2 && epipole_gate) {
            const float distex = epx - k2->kx[idx2], distey = epy - k2->ky[idx2];
            if (distex * distex + distey * distey < 100 * k2->scaleFactors[k2->octave[idx2]]) continue;
          }
          if (pred(user, idx1, idx2) || bCoarse) {        /* :1148 (the predicate is evaluated first, as there) */
            bestIdx2 = idx2;
            bestDist = dist;
          }
        }
        if (bestIdx2 >= 0) {
          vMatches12[idx1] = bestIdx2;
          nmatches++;
          if (checkOri) {
            float rot = k1->angle[idx1] - k2->angle[bestIdx2];
            if ((double)rot < 0.0) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            rotHist[bin][rotN[bin]++] = idx1;
          }
        }
      }
      f1++; f2++;
    } else if (k1->node_id[f1] < k2->node_id[f2]) {
      while (f1 < k1->n_nodes && k1->node_id[f1] < k2->node_id[f2]) f1++;   /* lower_bound */
    } else {
      while (f2 < k2->n_nodes && k2->node_id[f2] < k1->node_id[f1]) f2++;
    }
  }
  if (checkOri) {
    int ind1, ind2, ind3;
    orc_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int j = 0; j < rotN[i]; j++) { vMatches12[rotHist[i][j]] = -1; nmatches--; }
    }
  }
  for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
  return nmatches;
}

/* Pinhole keyframes (mpCamera2 == NULL): the predicate is Pinhole::epipolarConstrain with the pair's F12 */
typedef struct { const orc_keyframe *k1, *k2; float F12[9]; } pinhole_pred_ctx;
static int pinhole_pred(void *user, int idx1, int idx2) {
  const pinhole_pred_ctx *c = (const pinhole_pred_ctx *)user;
  return epipolar_constrain(c->F12, c->k1->kx[idx1], c->k1->ky[idx1], c->k2->kx[idx2], c->k2->ky[idx2], c->k2->levelSigma2[c->k2->octave[idx2]]);
}
/* exposed for tests that inject the same predicate into the product (tests/test_gpu_triangulation_pred.py) */
int orc_pinhole_epipolar_constrain(const float *F12, float x1, float y1, float x2, float y2, float unc) { return epipolar_constrain(F12, x1, y1, x2, y2, unc); }

/* epipole and F12 of a Pinhole pair: C2 = R2w*Cw+t2w, ep = project(C2) (:988-994); R12 = R1w*R2w.t(), t12 = -R1w*R2w.t()*t2w+t1w (:1008-1010) */
void orc_pinhole_pair_geometry(const float *R1w, const float *t1w, const float *R2w, const float *t2w, const float *Cw1, const float *cam1,
                               const float *cam2, float *ep, float *F12) {
  float C2[3];
  for (int i = 0; i < 3; i++) {
    float t0 = R2w[i * 3 + 0] * Cw1[0] + R2w[i * 3 + 1] * Cw1[1] + R2w[i * 3 + 2] * Cw1[2];
    C2[i] = (float)((double)t0 + (double)t2w[i]);
  }
  orc_project(0, cam2, C2[0], C2[1], C2[2], &ep[0], &ep[1]);
  float R12[9], Rm[9], t12[3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double sacc = 0;
      for (int k = 0; k < 3; k++) sacc += (double)R1w[i * 3 + k] * (double)R2w[j * 3 + k];
      R12[i * 3 + j] = (float)(sacc * 1.0);
      Rm[i * 3 + j] = (float)(sacc * -1.0);
    }
  for (int i = 0; i < 3; i++) {
    float t0 = Rm[i * 3 + 0] * t2w[0] + Rm[i * 3 + 1] * t2w[1] + Rm[i * 3 + 2] * t2w[2];
    t12[i] = (float)((double)t0 + (double)t1w[i]);
  }
  orc_pinhole_F12(R12, t12, cam1, cam2, F12);
}

int orc_search_for_triangulation(const orc_keyframe *k1, const orc_keyframe *k2, const float *R1w, const float *t1w,
                                 const float *R2w, const float *t2w, const float *Cw1, const float *cam1, const float *cam2,
                                 int bOnlyStereo, int bCoarse, int checkOri, int32_t *vMatches12) {
  pinhole_pred_ctx c;
  c.k1 = k1; c.k2 = k2;
  float ep[2];
  orc_pinhole_pair_geometry(R1w, t1w, R2w, t2w, Cw1, cam1, cam2, ep, c.F12);
  return orc_search_for_triangulation_pred(k1, k2, ep[0], ep[1], 1, bOnlyStereo, bCoarse, checkOri, pinhole_pred, &c, vMatches12);
}

/* The lists orbm_triangulation_candidates documents (include/orbhip.h), from the same loops: per idx1 every idx2 that reaches the   */
/* predicate when bestDist never drops (dist <= TH_LOW and the gates), ordered (dist ascending, node position descending).            */
int orc_triangulation_candidates(const orc_keyframe *k1, const orc_keyframe *k2, float epx, float epy, int epipole_gate, int bOnlyStereo,
                                 int32_t *start, int32_t *cidx2, int32_t *cdist, int cap) {
  int total = 0;
  for (int i = 0; i <= k1->N; i++) start[i] = 0;
  /* lists are built per idx1 in idx1 order: first the counts, then a second merge-walk that fills */
  for (int pass = 0; pass < 2; pass++) {
    int f1 = 0, f2 = 0;
    while (f1 < k1->n_nodes && f2 < k2->n_nodes) {
      if (k1->node_id[f1] == k2->node_id[f2]) {
        for (int i1 = k1->node_start[f1]; i1 < k1->node_start[f1 + 1]; i1++) {
          const int idx1 = k1->node_idx[i1];
          if (k1->has_mp[idx1]) continue;
          const int bStereo1 = k1->uRight[idx1] >= 0;
          if (bOnlyStereo && !bStereo1) continue;
          int n = 0;
          for (int i2 = k2->node_start[f2]; i2 < k2->node_start[f2 + 1]; i2++) {
            const int idx2 = k2->node_idx[i2];
            if (k2->has_mp[idx2]) continue;
            const int bStereo2 = k2->uRight[idx2] >= 0;
            if (bOnlyStereo && !bStereo2) continue;
            const int dist = orc_descriptor_distance(k1->desc + 32 * (size_t)idx1, k2->desc + 32 * (size_t)idx2);
            if (dist > 50) continue;
            if (!bStereo1 && !bStereo2 && epipole_gate) {
              const float distex = epx - k2->kx[idx2], distey = epy - k2->ky[idx2];
              if (distex * distex + distey * distey < 100 * k2->scaleFactors[k2->octave[idx2]]) continue;
            }
            if (pass == 1 && start[idx1] + n < cap) {   /* insertion sort by (dist asc, position desc): a later equal distance goes first */
              int o = start[idx1] + n;
              while (o > start[idx1] && cdist[o - 1] >= dist) { cidx2[o] = cidx2[o - 1]; cdist[o] = cdist[o - 1]; o--; }
              cidx2[o] = idx2; cdist[o] = dist;
            }
            n++;
          }
          if (pass == 0) { start[idx1 + 1] = n; total += n; }
        }
        f1++; f2++;
      } else if (k1->node_id[f1] < k2->node_id[f2]) {
        while (f1 < k1->n_nodes && k1->node_id[f1] < k2->node_id[f2]) f1++;
      } else {
        while (f2 < k2->n_nodes && k2->node_id[f2] < k1->node_id[f1]) f2++;
      }
    }
    if (pass == 0) {
      for (int i = 0; i < k1->N; i++) start[i + 1] += start[i];
      if (total > cap) return total;
    }
  }
  return total;
}

/* ------------------------------------------------------------------------------------------ */
/* M4, ORBmatcher.cc:2291-2413                                                                  */
/* ------------------------------------------------------------------------------------------ */
int orc_search_by_projection_kf(orc_frame *cur, int nKF, const uint8_t *valid, const float *Xw, const uint8_t *mpdesc,
                                const float *kfAngle, const float *maxDist, const float *minDist, const float *Tcw,
                                int camType, const float *camParams, float logScaleFactor, float th, int ORBdist,
                                int checkOri, int32_t *slot, uint8_t *slot_obs) {
  int nmatches = 0;
  const int HISTO_LENGTH = 30;
  int *rotHist[30];
  int rotN[30];
  for (int i = 0; i < HISTO_LENGTH; i++) { rotHist[i] = (int *)malloc(sizeof(int) * (size_t)(nKF + 1)); rotN[i] = 0; }
  const float factor = 1.0f / HISTO_LENGTH;
  const float tcw[3] = {Tcw[3], Tcw[7], Tcw[11]};
  float Ow[3]; /* Ow = -Rcw.t()*tcw : generic gemm, double accumulation, alpha -1 (:2297) */
  for (int i = 0; i < 3; i++) {
    double sacc = 0;
    for (int k = 0; k < 3; k++) sacc += (double)Tcw[k * 4 + i] * (double)tcw[k];
    Ow[i] = (float)(sacc * -1.0);
  }
  int32_t *vIndices2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(cur->N > 0 ? cur->N : 1));
  for (int i = 0; i < nKF; i++) {
    if (!valid[i]) continue;
    const float *x3Dw = Xw + 3 * i;
    float x3Dc[3];
    mat3_mul_add(Tcw, 4, x3Dw, tcw, x3Dc);
    float uvx, uvy;
    orc_project(camType, camParams, x3Dc[0], x3Dc[1], x3Dc[2], &uvx, &uvy);
    if (uvx < cur->mnMinX || uvx > cur->mnMaxX) continue;
    if (uvy < cur->mnMinY || uvy > cur->mnMaxY) continue;
    /* PO = x3Dw-Ow; dist3D = cv::norm(PO): float subtraction, L2 norm accumulated in double (:2321-2322) */
    double n2 = 0;
    for (int k = 0; k < 3; k++) { float po = x3Dw[k] - Ow[k]; n2 += (double)po * (double)po; }
    const float dist3D = (float)sqrt(n2);
    const float maxDistance = 1.2f * maxDist[i], minDistance = 0.8f * minDist[i]; /* MapPoint.cc:552-563 */
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    /* MapPoint::PredictScale(dist3D, Frame*), MapPoint.cc:587-602 */
    const float ratio = maxDist[i] / dist3D;
    int nPredictedLevel = (int)ceilf(logf(ratio) / logScaleFactor);
    if (nPredictedLevel < 0) nPredictedLevel = 0;
    else if (nPredictedLevel >= cur->nlevels) nPredictedLevel = cur->nlevels - 1;
    const float radius = th * cur->mvScaleFactors[nPredictedLevel];
    int nv = orc_get_features_in_area(cur, uvx, uvy, radius, nPredictedLevel - 1, nPredictedLevel + 1, vIndices2);
    if (nv == 0) continue;
    const uint8_t *dMP = mpdesc + 32 * (size_t)i;
    int bestDist = 256, bestIdx2 = -1;
    for (int k = 0; k < nv; k++) {
      const int i2 = vIndices2[k];
      if (slot[i2] >= 0) continue; /* any occupant blocks (:2355-2356) */
      const int dist = orc_descriptor_distance(dMP, cur->desc + 32 * (size_t)i2);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= ORBdist) {
      slot[bestIdx2] = i;
      slot_obs[bestIdx2] = 1;
      nmatches++;
      if (checkOri) {
        float rot = kfAngle[i] - cur->angle[bestIdx2];
        if ((double)rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin][rotN[bin]++] = bestIdx2;
      }
    }
  }
  if (checkOri) {
    int ind1, ind2, ind3;
    orc_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i != ind1 && i != ind2 && i != ind3)
        for (int j = 0; j < rotN[i]; j++) { slot[rotHist[i][j]] = -1; slot_obs[rotHist[i][j]] = 0; nmatches--; }
    }
  }
  for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
  free(vIndices2);
  return nmatches;
}

/* ------------------------------------------------------------------------------------------ */
/* M5, ORBmatcher.cc:489-602                                                                    */
/* ------------------------------------------------------------------------------------------ */
int orc_search_by_projection_sim3(orc_frame *kf, int nP, const uint8_t *valid, const float *Xw, const float *normal,
                                  const uint8_t *mpdesc, const float *maxDist, const float *minDist, const float *Scw,
                                  const float *cam, float logScaleFactor, int th, float ratioHamming, int32_t *slot,
                                  uint8_t *slot_obs) {
  return orc_search_by_projection_sim3_cam(kf, nP, valid, Xw, normal, mpdesc, maxDist, minDist, Scw, 0, cam, logScaleFactor, th, ratioHamming, slot, slot_obs);
}
/* pKF->mpCamera->project (:534) for either camera model */
int orc_search_by_projection_sim3_cam(orc_frame *kf, int nP, const uint8_t *valid, const float *Xw, const float *normal,
                                      const uint8_t *mpdesc, const float *maxDist, const float *minDist, const float *Scw, int camType,
                                      const float *cam, float logScaleFactor, int th, float ratioHamming, int32_t *slot,
                                      uint8_t *slot_obs) {
  /* Decompose Scw (:498-503): Mat::dot accumulates in double; Mat / float = convert with scale 1./s in double */
  double dot = 0;
  for (int k = 0; k < 3; k++) dot += (double)Scw[k] * (double)Scw[k];
  const float scw = (float)sqrt(dot);
  const double inv = 1. / (double)scw;
  float Rcw[9], tcw[3], Ow[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) Rcw[i * 3 + j] = (float)((double)Scw[i * 4 + j] * inv);
    tcw[i] = (float)((double)Scw[i * 4 + 3] * inv);
  }
  for (int i = 0; i < 3; i++) {
    double sacc = 0;
    for (int k = 0; k < 3; k++) sacc += (double)Rcw[k * 3 + i] * (double)tcw[k];
    Ow[i] = (float)(sacc * -1.0);
  }
  int nmatches = 0;
  int32_t *vIndices = (int32_t *)malloc(sizeof(int32_t) * (size_t)(kf->N > 0 ? kf->N : 1));
  for (int iMP = 0; iMP < nP; iMP++) {
    if (!valid[iMP]) continue;
    const float *p3Dw = Xw + 3 * iMP;
    float p3Dc[3];
    mat3_mul_add(Rcw, 3, p3Dw, tcw, p3Dc);
    if ((double)p3Dc[2] < 0.0) continue;
    float uvx, uvy;
    orc_project(camType, cam, p3Dc[0], p3Dc[1], p3Dc[2], &uvx, &uvy);
    if (!(uvx >= kf->mnMinX && uvx < kf->mnMaxX && uvy >= kf->mnMinY && uvy < kf->mnMaxY)) continue; /* KeyFrame::IsInImage */
    const float maxDistance = 1.2f * maxDist[iMP], minDistance = 0.8f * minDist[iMP];
    float PO[3];
    double n2 = 0, pd = 0;
    for (int k = 0; k < 3; k++) { PO[k] = p3Dw[k] - Ow[k]; n2 += (double)PO[k] * (double)PO[k]; }
    const float dist = (float)sqrt(n2);
    if (dist < minDistance || dist > maxDistance) continue;
    for (int k = 0; k < 3; k++) pd += (double)PO[k] * (double)normal[3 * iMP + k];
    if (pd < 0.5 * (double)dist) continue;                     /* viewing angle < 60 deg, :553 */
    const float ratio = maxDist[iMP] / dist;                   /* PredictScale(dist, pKF), MapPoint.cc:570-585 */
    int nPredictedLevel = (int)ceilf(logf(ratio) / logScaleFactor);
    if (nPredictedLevel < 0) nPredictedLevel = 0;
    else if (nPredictedLevel >= kf->nlevels) nPredictedLevel = kf->nlevels - 1;
    const float radius = (float)th * kf->mvScaleFactors[nPredictedLevel];
    int nv = orc_get_features_in_area(kf, uvx, uvy, radius, -1, -1, vIndices); /* KeyFrame::GetFeaturesInArea: no level filter */
    if (nv == 0) continue;
    const uint8_t *dMP = mpdesc + 32 * (size_t)iMP;
    int bestDist = 256, bestIdx = -1;
    for (int k = 0; k < nv; k++) {
      const int idx = vIndices[k];
      if (slot[idx] >= 0) continue;
      const int kpLevel = kf->octave[idx];
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      const int d = orc_descriptor_distance(dMP, kf->desc + 32 * (size_t)idx);
      if (d < bestDist) { bestDist = d; bestIdx = idx; }
    }
    if ((float)bestDist <= 50 /* TH_LOW */ * ratioHamming) {
      slot[bestIdx] = iMP;
      slot_obs[bestIdx] = 1;
      nmatches++;
    }
  }
  free(vIndices);
  return nmatches;
}

/* ------------------------------------------------------------------------------------------ */
/* G0: cv::undistortPoints as used by Frame.cc:856 / :883 [OPENCV-UNVERIFIED]   */
/* ------------------------------------------------------------------------------------------ */
void orc_undistort_points(int n, const float *xy_in, const float *K, const float *D, int nD, float *xy_out) {
  if (D[0] == 0.0f) { memcpy(xy_out, xy_in, sizeof(float) * 2 * (size_t)n); return; }
  const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
  const double k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = nD > 4 ? D[4] : 0.0;
  const double ifx = 1. / fx, ify = 1. / fy;
  for (int i = 0; i < n; i++) {
    double x = ((double)xy_in[2 * i] - cx) * ifx, y = ((double)xy_in[2 * i + 1] - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
      double r2 = x * x + y * y;
      double icdist = 1. / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
      double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
      double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    xy_out[2 * i] = (float)(x * fx + cx);      /* P = K */
    xy_out[2 * i + 1] = (float)(y * fy + cy);
  }
}

void orc_image_bounds(int cols, int rows, const float *K, const float *D, int nD, float *minX, float *maxX, float *minY, float *maxY) {
  if (D[0] != 0.0f) {
    float in[8] = {0.f, 0.f, (float)cols, 0.f, 0.f, (float)rows, (float)cols, (float)rows}, out[8];
    orc_undistort_points(4, in, K, D, nD, out);
    *minX = out[0] < out[4] ? out[0] : out[4];
    *maxX = out[2] > out[6] ? out[2] : out[6];
    *minY = out[1] < out[3] ? out[1] : out[3];
    *maxY = out[5] > out[7] ? out[5] : out[7];
  } else {
    *minX = 0.0f; *maxX = (float)cols; *minY = 0.0f; *maxY = (float)rows;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* N3 (first member): ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&),           */
/* ORBmatcher.cc:273-469, Frame::Nleft == -1.  Both sides as orc_keyframe (the frame's has_mp    */
/* is unused); matchF[F.N] = keyframe keypoint index whose map point the frame keypoint got.    */
/* ------------------------------------------------------------------------------------------ */
int orc_search_by_bow_kf_frame_stereo(const orc_keyframe *KF, const orc_keyframe *F, int Nleft, float nnratio, int checkOri, int32_t *matchF);
int orc_search_by_bow_kf_frame(const orc_keyframe *KF, const orc_keyframe *F, float nnratio, int checkOri, int32_t *matchF) {
  return orc_search_by_bow_kf_frame_stereo(KF, F, -1, nnratio, checkOri, matchF);
}

/* Nleft = F.Nleft (-1: mono / rectified frame).  For Nleft != -1 F holds mvKeys ++ mvKeysRight and the branch :338-363 /
 * :405-436 applies: separate best for the right image, accepted inside `if(bestDist1<=TH_LOW)` with `|| true` (:407). */
int orc_search_by_bow_kf_frame_stereo(const orc_keyframe *KF, const orc_keyframe *F, int Nleft, float nnratio, int checkOri, int32_t *matchF) {
  const int HISTO_LENGTH = 30, TH_LOW = 50;
  int nmatches = 0;
  for (int i = 0; i < F->N; i++) matchF[i] = -1;
  int *rotHist[30];
  int rotN[30];
  for (int i = 0; i < HISTO_LENGTH; i++) { rotHist[i] = (int *)malloc(sizeof(int) * (size_t)(2 * F->N + 1)); rotN[i] = 0; }
  const float factor = 1.0f / HISTO_LENGTH;
  int a = 0, b = 0;
  while (a < KF->n_nodes && b < F->n_nodes) {
    if (KF->node_id[a] == F->node_id[b]) {
      for (int iKF = KF->node_start[a]; iKF < KF->node_start[a + 1]; iKF++) {
        const int realIdxKF = KF->node_idx[iKF];
        if (!KF->has_mp[realIdxKF]) continue;
        const uint8_t *dKF = KF->desc + 32 * (size_t)realIdxKF;
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
        int bestDist1R = 256, bestIdxFR = -1, bestDist2R = 256;
        for (int iF = F->node_start[b]; iF < F->node_start[b + 1]; iF++) {
          const int realIdxF = F->node_idx[iF];
          if (matchF[realIdxF] >= 0) continue;
          const int dist = orc_descriptor_distance(dKF, F->desc + 32 * (size_t)realIdxF);
          if (Nleft == -1) {
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
            else if (dist < bestDist2) bestDist2 = dist;
          } else {
            if (realIdxF < Nleft && dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
            else if (realIdxF < Nleft && dist < bestDist2) bestDist2 = dist;
            if (realIdxF >= Nleft && dist < bestDist1R) { bestDist2R = bestDist1R; bestDist1R = dist; bestIdxFR = realIdxF; }
            else if (realIdxF >= Nleft && dist < bestDist2R) bestDist2R = dist;
          }
        }
        if (bestDist1 <= TH_LOW) {
          if ((float)bestDist1 < nnratio * (float)bestDist2) {
            matchF[bestIdxF] = realIdxKF;
            if (checkOri) {
              float rot = KF->angle[realIdxKF] - F->angle[bestIdxF];
              if ((double)rot < 0.0) rot += 360.0f;
              int bin = (int)roundf(rot * factor);
              if (bin == HISTO_LENGTH) bin = 0;
              rotHist[bin][rotN[bin]++] = bestIdxF;
            }
            nmatches++;
          }
          if (bestDist1R <= TH_LOW) {
            if ((float)bestDist1R < nnratio * (float)bestDist2R || 1) {
              matchF[bestIdxFR] = realIdxKF;
              if (checkOri) {
                float rot = KF->angle[realIdxKF] - F->angle[bestIdxFR];
                if ((double)rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin][rotN[bin]++] = bestIdxFR;
              }
              nmatches++;
            }
          }
        }
      }
      a++; b++;
    } else if (KF->node_id[a] < F->node_id[b]) {
      while (a < KF->n_nodes && KF->node_id[a] < F->node_id[b]) a++;
    } else {
      while (b < F->n_nodes && F->node_id[b] < KF->node_id[a]) b++;
    }
  }
  if (checkOri) {
    int ind1, ind2, ind3;
    orc_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int j = 0; j < rotN[i]; j++) { matchF[rotHist[i][j]] = -1; nmatches--; }
    }
  }
  for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
  return nmatches;
}

/* N3: ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&), ORBmatcher.cc:839-979 (NLeft == -1). */
int orc_search_by_bow_kf_kf(const orc_keyframe *K1, const orc_keyframe *K2, float nnratio, int checkOri, int32_t *matches12) {
  const int HISTO_LENGTH = 30, TH_LOW = 50;
  int nmatches = 0;
  for (int i = 0; i < K1->N; i++) matches12[i] = -1;
  uint8_t *vbMatched2 = (uint8_t *)calloc((size_t)K2->N + 1, 1);
  int *rotHist[30];
  int rotN[30];
  for (int i = 0; i < HISTO_LENGTH; i++) { rotHist[i] = (int *)malloc(sizeof(int) * (size_t)(K1->N + 1)); rotN[i] = 0; }
  const float factor = 1.0f / HISTO_LENGTH;
  int a = 0, b = 0;
  while (a < K1->n_nodes && b < K2->n_nodes) {
    if (K1->node_id[a] == K2->node_id[b]) {
      for (int i1 = K1->node_start[a]; i1 < K1->node_start[a + 1]; i1++) {
        const int idx1 = K1->node_idx[i1];
        if (!K1->has_mp[idx1]) continue;
        const uint8_t *d1 = K1->desc + 32 * (size_t)idx1;
        int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
        for (int i2 = K2->node_start[b]; i2 < K2->node_start[b + 1]; i2++) {
          const int idx2 = K2->node_idx[i2];
          if (vbMatched2[idx2] || !K2->has_mp[idx2]) continue;
          const int dist = orc_descriptor_distance(d1, K2->desc + 32 * (size_t)idx2);
          if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
          else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist1 < TH_LOW) {
          if ((float)bestDist1 < nnratio * (float)bestDist2) {
            matches12[idx1] = bestIdx2;
            vbMatched2[bestIdx2] = 1;
            if (checkOri) {
              float rot = K1->angle[idx1] - K2->angle[bestIdx2];
              if ((double)rot < 0.0) rot += 360.0f;
              int bin = (int)roundf(rot * factor);
              if (bin == HISTO_LENGTH) bin = 0;
              rotHist[bin][rotN[bin]++] = idx1;
            }
            nmatches++;
          }
        }
      }
      a++; b++;
    } else if (K1->node_id[a] < K2->node_id[b]) {
      while (a < K1->n_nodes && K1->node_id[a] < K2->node_id[b]) a++;
    } else {
      while (b < K2->n_nodes && K2->node_id[b] < K1->node_id[a]) b++;
    }
  }
  if (checkOri) {
    int ind1, ind2, ind3;
    orc_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int j = 0; j < rotN[i]; j++) { matches12[rotHist[i][j]] = -1; nmatches--; }
    }
  }
  for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
  free(vbMatched2);
  return nmatches;
}

/* ------------------------------------------------------------------------------------------ */
/* N3: ORBmatcher::Fuse.  chi2 != 0: Fuse(KeyFrame*, const vector<MapPoint*>&, th, bRight=false),  */
/* ORBmatcher.cc:1425-1658 (T = [Rcw | tcw] row-major 3x3 + 3, Ow, camera, bf, invLevelSigma2).     */
/* chi2 == 0: Fuse(KeyFrame*, cv::Mat Scw, vpPoints, th, vpReplacePoint), :1660-1786, with T / Ow   */
/* from the Sim3 decomposition (orc_fuse_sim3).  Only the per-point search is restated: bestIdx[i]  */
/* (bestDist <= TH_LOW) or -1, bestDist[i]; the Replace / AddObservation bookkeeping acts on objects */
/* that do not feed back into the search.                                                            */
/* ------------------------------------------------------------------------------------------ */
static int fuse_impl(orc_frame *kf, int nP, const uint8_t *valid, const float *Xw, const float *normal, const uint8_t *mpdesc,
                     const float *maxDist, const float *minDist, const float *Rcw, const float *tcw, const float *Ow, int camType,
                     const float *cam, float bf, const float *invLevelSigma2, float logScaleFactor, float th, int chi2,
                     int32_t *bestIdxOut, int32_t *bestDistOut) {
  int nFused = 0;
  int32_t *vIndices = (int32_t *)malloc(sizeof(int32_t) * (size_t)(kf->N > 0 ? kf->N : 1));
  for (int i = 0; i < nP; i++) {
    bestIdxOut[i] = -1; bestDistOut[i] = 256;
    if (!valid[i]) continue;
    const float *p3Dw = Xw + 3 * i;
    float p3Dc[3];
    mat3_mul_add(Rcw, 3, p3Dw, tcw, p3Dc);
    if (p3Dc[2] < 0.0f) continue;
    const float invz = 1 / p3Dc[2];
    float uvx, uvy;
    orc_project(camType, cam, p3Dc[0], p3Dc[1], p3Dc[2], &uvx, &uvy);
    if (!(uvx >= kf->mnMinX && uvx < kf->mnMaxX && uvy >= kf->mnMinY && uvy < kf->mnMaxY)) continue;
    const float ur = uvx - bf * invz;
    const float maxDistance = 1.2f * maxDist[i], minDistance = 0.8f * minDist[i];
    float PO[3];
    double n2 = 0, pd = 0;
    for (int k = 0; k < 3; k++) { PO[k] = p3Dw[k] - Ow[k]; n2 += (double)PO[k] * (double)PO[k]; }
    const float dist3D = (float)sqrt(n2);
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    for (int k = 0; k < 3; k++) pd += (double)PO[k] * (double)normal[3 * i + k];
    if (pd < 0.5 * (double)dist3D) continue;
    const float ratio = maxDist[i] / dist3D;
    int nPredictedLevel = (int)ceilf(logf(ratio) / logScaleFactor);
    if (nPredictedLevel < 0) nPredictedLevel = 0;
    else if (nPredictedLevel >= kf->nlevels) nPredictedLevel = kf->nlevels - 1;
    const float radius = th * kf->mvScaleFactors[nPredictedLevel];
    const int nv = orc_get_features_in_area(kf, uvx, uvy, radius, -1, -1, vIndices);
    if (nv == 0) continue;
    const uint8_t *dMP = mpdesc + 32 * (size_t)i;
    int bestDist = 256, bestIdx = -1;
    for (int k = 0; k < nv; k++) {
      const int idx = vIndices[k];
      const int kpLevel = kf->octave[idx];
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      if (chi2) {
        const float kpx = kf->kx[idx], kpy = kf->ky[idx];
        const float kur = kf->uRight ? kf->uRight[idx] : -1.0f;
        if (kur >= 0) {
          const float ex = uvx - kpx, ey = uvy - kpy, er = ur - kur;
          const float e2 = ex * ex + ey * ey + er * er;
          if ((double)(e2 * invLevelSigma2[kpLevel]) > 7.8) continue;
        } else {
          const float ex = uvx - kpx, ey = uvy - kpy;
          const float e2 = ex * ex + ey * ey;
          if ((double)(e2 * invLevelSigma2[kpLevel]) > 5.99) continue;
        }
      }
      const int dist = orc_descriptor_distance(dMP, kf->desc + 32 * (size_t)idx);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    if (bestDist <= 50 /* TH_LOW */) {
      bestIdxOut[i] = bestIdx;
      bestDistOut[i] = bestDist;
      nFused++;
    }
  }
  free(vIndices);
  return nFused;
}

int orc_fuse(orc_frame *kf, int nP, const uint8_t *valid, const float *Xw, const float *normal, const uint8_t *mpdesc,
             const float *maxDist, const float *minDist, const float *Tcw /* row-major 4x4 */, const float *Ow, int camType,
             const float *cam, float bf, const float *invLevelSigma2, float logScaleFactor, float th, int32_t *bestIdx, int32_t *bestDist) {
  float Rcw[9], tcw[3];
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Rcw[3 * i + j] = Tcw[4 * i + j]; tcw[i] = Tcw[4 * i + 3]; }
  return fuse_impl(kf, nP, valid, Xw, normal, mpdesc, maxDist, minDist, Rcw, tcw, Ow, camType, cam, bf, invLevelSigma2, logScaleFactor, th, 1,
                   bestIdx, bestDist);
}

int orc_fuse_sim3(orc_frame *kf, int nP, const uint8_t *valid, const float *Xw, const float *normal, const uint8_t *mpdesc,
                  const float *maxDist, const float *minDist, const float *Scw, const float *cam, float logScaleFactor, float th,
                  int32_t *bestIdx, int32_t *bestDist) {
  return orc_fuse_sim3_cam(kf, nP, valid, Xw, normal, mpdesc, maxDist, minDist, Scw, 0, cam, logScaleFactor, th, bestIdx, bestDist);
}
int orc_fuse_sim3_cam(orc_frame *kf, int nP, const uint8_t *valid, const float *Xw, const float *normal, const uint8_t *mpdesc,
                      const float *maxDist, const float *minDist, const float *Scw, int camType, const float *cam, float logScaleFactor, float th,
                      int32_t *bestIdx, int32_t *bestDist) {
  double dot = 0;
  for (int k = 0; k < 3; k++) dot += (double)Scw[k] * (double)Scw[k];
  const float scw = (float)sqrt(dot);
  const double inv = 1. / (double)scw;
  float Rcw[9], tcw[3], Ow[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) Rcw[i * 3 + j] = (float)((double)Scw[i * 4 + j] * inv);
    tcw[i] = (float)((double)Scw[i * 4 + 3] * inv);
  }
  for (int i = 0; i < 3; i++) {
    double sacc = 0;
    for (int k = 0; k < 3; k++) sacc += (double)Rcw[k * 3 + i] * (double)tcw[k];
    Ow[i] = (float)(sacc * -1.0);
  }
  return fuse_impl(kf, nP, valid, Xw, normal, mpdesc, maxDist, minDist, Rcw, tcw, Ow, camType, cam, 0.f, NULL, logScaleFactor, th, 0, bestIdx, bestDist);
}

/* ------------------------------------------------------------------------------------------ */
/* N3: ORBmatcher::SearchBySim3, ORBmatcher.cc:1788-2012.  cv::Mat algebra per SURVEY.md A.8      */
/* [OPENCV-UNVERIFIED]: scalar * Mat in double, products with a transposed / scaled operand       */
/* accumulate in double, plain 3x3*3x1+3x1 on the small-matrix float path.                        */
/* ------------------------------------------------------------------------------------------ */
static void sim3_dir(orc_frame *kfB, float logSfB, int nA, const uint8_t *valid, const float *Xw, const uint8_t *mpdesc,
                     const float *maxDist, const float *minDist, const float *RAw, const float *tAw, const float *sRBA,
                     const float *tBA, const float *cam, float th, int32_t *vnMatch) {
  int32_t *vIndices = (int32_t *)malloc(sizeof(int32_t) * (size_t)(kfB->N > 0 ? kfB->N : 1));
  for (int i = 0; i < nA; i++) {
    vnMatch[i] = -1;
    if (!valid[i]) continue;
    float pA[3], pB[3];
    mat3_mul_add(RAw, 3, Xw + 3 * i, tAw, pA);
    mat3_mul_add(sRBA, 3, pA, tBA, pB);
    if ((double)pB[2] < 0.0) continue;
    const float invz = (float)(1.0 / (double)pB[2]);
    const float x = pB[0] * invz, y = pB[1] * invz;
    const float u = cam[0] * x + cam[2], v = cam[1] * y + cam[3];
    if (!(u >= kfB->mnMinX && u < kfB->mnMaxX && v >= kfB->mnMinY && v < kfB->mnMaxY)) continue;
    const float maxDistance = 1.2f * maxDist[i], minDistance = 0.8f * minDist[i];
    double n2 = 0;
    for (int k = 0; k < 3; k++) n2 += (double)pB[k] * (double)pB[k];
    const float dist3D = (float)sqrt(n2);
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    const float ratio = maxDist[i] / dist3D;
    int nPredictedLevel = (int)ceilf(logf(ratio) / logSfB);
    if (nPredictedLevel < 0) nPredictedLevel = 0;
    else if (nPredictedLevel >= kfB->nlevels) nPredictedLevel = kfB->nlevels - 1;
    const float radius = th * kfB->mvScaleFactors[nPredictedLevel];
    const int nv = orc_get_features_in_area(kfB, u, v, radius, -1, -1, vIndices);
    if (nv == 0) continue;
    const uint8_t *dMP = mpdesc + 32 * (size_t)i;
    int bestDist = 2147483647, bestIdx = -1;
    for (int k = 0; k < nv; k++) {
      const int idx = vIndices[k];
      if (kfB->octave[idx] < nPredictedLevel - 1 || kfB->octave[idx] > nPredictedLevel) continue;
      const int dist = orc_descriptor_distance(dMP, kfB->desc + 32 * (size_t)idx);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    if (bestDist <= 100 /* TH_HIGH */) vnMatch[i] = bestIdx;
  }
  free(vIndices);
}

int orc_search_by_sim3(orc_frame *kf1, float logSf1, const uint8_t *valid1, const float *Xw1, const uint8_t *mpdesc1,
                       const float *maxDist1, const float *minDist1, const float *R1w, const float *t1w, orc_frame *kf2, float logSf2,
                       const uint8_t *valid2, const float *Xw2, const uint8_t *mpdesc2, const float *maxDist2, const float *minDist2,
                       const float *R2w, const float *t2w, float s12, const float *R12, const float *t12, const float *cam1, float th,
                       int32_t *matches12) {
  float sR12[9], sR21[9], t21[3];
  const double a21 = 1.0 / (double)s12;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      sR12[3 * i + j] = (float)((double)R12[3 * i + j] * (double)s12);
      sR21[3 * i + j] = (float)((double)R12[3 * j + i] * a21);
    }
  for (int i = 0; i < 3; i++) {
    double acc = 0;
    for (int k = 0; k < 3; k++) acc += (double)sR21[3 * i + k] * (double)t12[k];
    t21[i] = (float)(acc * -1.0);
  }
  const int N1 = kf1->N, N2 = kf2->N;
  int32_t *vnMatch1 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N1 + 1)), *vnMatch2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N2 + 1));
  sim3_dir(kf2, logSf2, N1, valid1, Xw1, mpdesc1, maxDist1, minDist1, R1w, t1w, sR21, t21, cam1, th, vnMatch1);
  sim3_dir(kf1, logSf1, N2, valid2, Xw2, mpdesc2, maxDist2, minDist2, R2w, t2w, sR12, t12, cam1, th, vnMatch2);
  int nFound = 0;
  for (int i1 = 0; i1 < N1; i1++) {
    matches12[i1] = -1;
    const int idx2 = vnMatch1[i1];
    if (idx2 >= 0) {
      const int idx1 = vnMatch2[idx2];
      if (idx1 == i1) { matches12[i1] = idx2; nFound++; }
    }
  }
  free(vnMatch1); free(vnMatch2);
  return nFound;
}

/* N3: MapPoint::ComputeDistinctiveDescriptors, MapPoint.cc:350-436: BestIdx of one group of N descriptors. */
static int cmp_int(const void *a, const void *b) { const int x = *(const int *)a, y = *(const int *)b; return x < y ? -1 : (x > y ? 1 : 0); }
int orc_distinctive_descriptor(const uint8_t *desc, int N) {
  if (N <= 0) return -1;
  float *D = (float *)malloc(sizeof(float) * (size_t)N * (size_t)N);
  for (int i = 0; i < N; i++) {
    D[(size_t)i * N + i] = 0;
    for (int j = i + 1; j < N; j++) {
      const int distij = orc_descriptor_distance(desc + 32 * (size_t)i, desc + 32 * (size_t)j);
      D[(size_t)i * N + j] = (float)distij;
      D[(size_t)j * N + i] = (float)distij;
    }
  }
  int BestMedian = 2147483647, BestIdx = 0;
  int *v = (int *)malloc(sizeof(int) * (size_t)N);
  for (int i = 0; i < N; i++) {
    for (int j = 0; j < N; j++) v[j] = (int)D[(size_t)i * N + j];
    qsort(v, (size_t)N, sizeof(int), cmp_int);
    const int median = v[(int)(0.5 * (double)(N - 1))];
    if (median < BestMedian) { BestMedian = median; BestIdx = i; }
  }
  free(v); free(D);
  return BestIdx;
}


/* ---- N4: the image operations the Examples apply before Track* --------------------------------------------------------------
 * Both are OpenCV (third-party, absent here) algorithms, restated from their published sources: [OPENCV-UNVERIFIED]. */

/* cv::createCLAHE(clipLimit, Size(tilesX, tilesY))->apply(src, dst), CV_8UC1 (Examples/Monocular/mono_tum_vi.cc:101-109).
 * OpenCV 3.4 / 4.x modules/imgproc/src/clahe.cpp: CLAHE_Impl::apply, CLAHE_CalcLut_Body, CLAHE_Interpolation_Body. */
static uint8_t sat_u8_f(float v) {
  long r = lrintf(v);   /* cvRound: round half to even */
  return (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
}
int orc_clahe(const uint8_t *src, int rows, int cols, size_t sstride, double clipLimit_, int tilesX, int tilesY, uint8_t *dst, size_t dstride) {
  enum { histSize = 256 };
  if (rows <= 0 || cols <= 0 || tilesX <= 0 || tilesY <= 0 || cols < tilesX || rows < tilesY) return -1;
  /* apply(): extend to a multiple of the tile grid with BORDER_REFLECT_101 (bottom / right only) when needed */
  const int ecols = cols % tilesX == 0 ? cols : cols + (tilesX - cols % tilesX);
  const int erows = rows % tilesY == 0 ? rows : rows + (tilesY - rows % tilesY);
  const int tw = ecols / tilesX, th = erows / tilesY;
  const int tileSizeTotal = tw * th;
  const float lutScale = (float)(histSize - 1) / tileSizeTotal;
  int clipLimit = 0;
  if (clipLimit_ > 0.0) {
    clipLimit = (int)(clipLimit_ * tileSizeTotal / histSize);
    if (clipLimit < 1) clipLimit = 1;
  }
  uint8_t *lut = (uint8_t *)malloc((size_t)tilesX * tilesY * histSize);
  for (int k = 0; k < tilesX * tilesY; k++) {
    const int ty = k / tilesX, tx = k % tilesX;
    int tileHist[histSize];
    memset(tileHist, 0, sizeof(tileHist));
    for (int y = ty * th; y < (ty + 1) * th; y++)
      for (int x = tx * tw; x < (tx + 1) * tw; x++) tileHist[src[(size_t)reflect101(y, rows) * sstride + reflect101(x, cols)]]++;
    if (clipLimit > 0) {
      int clipped = 0;
      for (int i = 0; i < histSize; i++)
        if (tileHist[i] > clipLimit) { clipped += tileHist[i] - clipLimit; tileHist[i] = clipLimit; }
      const int redistBatch = clipped / histSize;
      int residual = clipped - redistBatch * histSize;
      for (int i = 0; i < histSize; i++) tileHist[i] += redistBatch;
      if (residual != 0) {
        const int residualStep = histSize / residual > 1 ? histSize / residual : 1;
        for (int i = 0; i < histSize && residual > 0; i += residualStep, residual--) tileHist[i]++;
      }
    }
    int sum = 0;
    for (int i = 0; i < histSize; i++) { sum += tileHist[i]; lut[(size_t)k * histSize + i] = sat_u8_f(sum * lutScale); }
  }
  const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
  for (int y = 0; y < rows; y++) {
    const float tyf = y * inv_th - 0.5f;
    int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
    const float ya = tyf - ty1, ya1 = 1.0f - ya;
    if (ty1 < 0) ty1 = 0;
    if (ty2 > tilesY - 1) ty2 = tilesY - 1;
    const uint8_t *lutPlane1 = lut + (size_t)ty1 * tilesX * histSize, *lutPlane2 = lut + (size_t)ty2 * tilesX * histSize;
    for (int x = 0; x < cols; x++) {
      const float txf = x * inv_tw - 0.5f;
      int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
      const float xa = txf - tx1, xa1 = 1.0f - xa;
      if (tx1 < 0) tx1 = 0;
      if (tx2 > tilesX - 1) tx2 = tilesX - 1;
      const int srcVal = src[(size_t)y * sstride + x];
      const int ind1 = tx1 * histSize + srcVal, ind2 = tx2 * histSize + srcVal;
      const float res = (lutPlane1[ind1] * xa1 + lutPlane1[ind2] * xa) * ya1 + (lutPlane2[ind1] * xa1 + lutPlane2[ind2] * xa) * ya;
      dst[(size_t)y * dstride + x] = sat_u8_f(res);
    }
  }
  free(lut);
  return 0;
}

/* cv::remap(src, dst, map1 (CV_32FC1 x), map2 (CV_32FC1 y), INTER_LINEAR, BORDER_CONSTANT, Scalar()) for CV_8UC1
 * (Examples/Stereo/stereo_euroc.cc:166-167).  OpenCV imgwarp.cpp: RemapInvoker converts the float maps to fixed point
 * (INTER_BITS = 5), remapBilinear<FixedPtCast<int, uchar, 15>> blends with the BilinearTab_i weights.  The table is built here
 * the way initInterTab2D does (float products, saturate_cast<short>(v * 32768)); every product is a multiple of 1/1024, so the
 * weights are exact and sum to 1 << 15 except at fx = fy = 0 (see below). */
static int cv_round_f(float v) {
  if (!(v > -2147483648.0f && v < 2147483648.0f)) return INT32_MIN;   /* cvtss2si "integer indefinite" */
  return (int)lrintf(v);
}
static short sat_s16(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
int orc_remap_linear(const uint8_t *src, int srows, int scols, size_t sstride, const float *mapx, const float *mapy, size_t mstride, int rows,
                     int cols, uint8_t *dst, size_t dstride) {
  enum { INTER_BITS = 5, INTER_TAB_SIZE = 32, COEF_BITS = 15, COEF_SCALE = 1 << 15 };
  static short wtab[INTER_TAB_SIZE * INTER_TAB_SIZE][4];
  float t1[INTER_TAB_SIZE][2];
  for (int i = 0; i < INTER_TAB_SIZE; i++) { const float x = i * (1.f / INTER_TAB_SIZE); t1[i][0] = 1.f - x; t1[i][1] = x; }
  for (int i = 0; i < INTER_TAB_SIZE; i++)
    for (int j = 0; j < INTER_TAB_SIZE; j++) {
      int isum = 0;
      for (int k1 = 0; k1 < 2; k1++)
        for (int k2 = 0; k2 < 2; k2++) {
          const float v = t1[i][k1] * t1[j][k2];
          isum += wtab[i * INTER_TAB_SIZE + j][k1 * 2 + k2] = sat_s16((int)lrintf(v * COEF_SCALE));
        }
      /* initInterTab2D's correction step: only the entry fx = fy = 0 needs it (1.0 * 32768 saturates to 32767, isum = 32767);
       * it looks for the extreme weight among table positions [ksize/2, ksize/2 + 2)^2, which for ksize = 2 starts at the
       * entry's last weight, and adds the missing 1 there: {32767, 0, 0, 1}.  No output depends on it: with taps a, b in 0..255,
       * (32767 a + b + 16384) >> 15 == a. */
      if (isum != COEF_SCALE) {
        if (i != 0 || j != 0 || isum != COEF_SCALE - 1) return -2;
        wtab[0][3] = (short)(wtab[0][3] + 1);
      }
    }
  for (int y = 0; y < rows; y++)
    for (int x = 0; x < cols; x++) {
      const int sxq = cv_round_f(mapx[(size_t)y * mstride + x] * INTER_TAB_SIZE), syq = cv_round_f(mapy[(size_t)y * mstride + x] * INTER_TAB_SIZE);
      const int v = (syq & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (sxq & (INTER_TAB_SIZE - 1));
      const int sx = sat_s16(sxq >> INTER_BITS), sy = sat_s16(syq >> INTER_BITS);
      const short *w = wtab[v];
      int val;
      if ((unsigned)sx < (unsigned)(scols - 1 > 0 ? scols - 1 : 0) && (unsigned)sy < (unsigned)(srows - 1 > 0 ? srows - 1 : 0)) {
        const uint8_t *S = src + (size_t)sy * sstride + sx;
        val = S[0] * w[0] + S[1] * w[1] + S[sstride] * w[2] + S[sstride + 1] * w[3];
      } else if (sx >= scols || sx + 1 < 0 || sy >= srows || sy + 1 < 0) {
        dst[(size_t)y * dstride + x] = 0;
        continue;
      } else {
        const int sx0 = sx, sx1 = sx + 1, sy0 = sy, sy1 = sy + 1;
        const int v0 = (sx0 >= 0 && sy0 >= 0 && sx0 < scols && sy0 < srows) ? src[(size_t)sy0 * sstride + sx0] : 0;
        const int v1 = (sx1 >= 0 && sy0 >= 0 && sx1 < scols && sy0 < srows) ? src[(size_t)sy0 * sstride + sx1] : 0;
        const int v2 = (sx0 >= 0 && sy1 >= 0 && sx0 < scols && sy1 < srows) ? src[(size_t)sy1 * sstride + sx0] : 0;
        const int v3 = (sx1 >= 0 && sy1 >= 0 && sx1 < scols && sy1 < srows) ? src[(size_t)sy1 * sstride + sx1] : 0;
        val = v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3];
      }
      val = (val + (1 << (COEF_BITS - 1))) >> COEF_BITS;
      dst[(size_t)y * dstride + x] = (uint8_t)(val < 0 ? 0 : val > 255 ? 255 : val);
    }
  return 0;
}
