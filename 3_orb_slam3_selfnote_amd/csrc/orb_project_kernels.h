// orb_project_kernels.h -- the parts of SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono)
// (ORBmatcher.cc:2027-2289, Nleft == -1) that surround the Hamming search: the projection of the last frame's map points into
// the current frame (:2038-2118) in front of k_match_scan / k_match_resolve, and the rotation-histogram pruning (:2177-2185,
// :2263-2286) behind them.  With both on the device a whole batch of frame pairs runs without touching the host
// (BASELINE config 5: KannalaBrandt8 projection inside the search).
//
// Every operation is the IEEE-754 single / double operation of the reference's expression in source order (no contraction),
// and the libm calls of KannalaBrandt8::project go through the bit-exact replicas of glibc's atan2f (orb_atan2f.h) and
// sinf / cosf (orb_sincos.h), so u, v - and with them the search windows - carry the same bits as on the host.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "orb_atan2f.h"
#include "orb_sincos.h"

struct LastFrameParams {
  // last-frame side; problem p is at element offset p * last_stride
  const uint8_t *has_mp;   // LastFrame.mvpMapPoints[i] && !LastFrame.mvbOutlier[i]
  const float *Xw;         // pMP->GetWorldPos(), 3 floats per keypoint
  const float *last_kp;    // LastFrame.mvKeys as 7 floats per keypoint (octave = word 5, angle = word 3)
  const uint8_t *obs;      // pMP->Observations() > 0, or NULL = all 1
  const float *Tcw, *Tlw;  // row-major 4x4 per problem (16 floats)
  int last_stride;
  const int32_t *last_n; int last_n_stride; int last_n_const;
  // current frame: image bounds, scale factors, camera
  float min_x, max_x, min_y, max_y;
  float sf[16]; int nlevels;
  int cam_type; float cam[8];
  float mb, mbf, th; int bMono;
  // out: the query arrays of the projection search (same stride)
  float *qu, *qv, *qr, *qur;
  int32_t *qminl, *qmaxl;
  uint8_t *qflags;
};

// cv::Mat `A*B + C` for a 3x3 * 3x1 product: float products summed in float, then C added (as on the host path)
__device__ __forceinline__ void dev_mat3_mul_add(const float *R, const float *x, const float *t, float *out) {  // R: row stride 4
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float t0 = R[i * 4 + 0] * x[0] + R[i * 4 + 1] * x[1] + R[i * 4 + 2] * x[2];
    out[i] = (float)((double)t0 + (double)t[i]);
  }
}

// GeometricCamera::project(cv::Point3f): 0 Pinhole (Pinhole.cpp:46-49), 1 KannalaBrandt8 (KannalaBrandt8.cpp:29-45)
__device__ __forceinline__ void dev_project(int cam_type, const float *p, float X, float Y, float Z, float &u, float &v) {
  if (cam_type == 0) {
    u = p[0] * X / Z + p[2];
    v = p[1] * Y / Z + p[3];
  } else {
    const float x2_plus_y2 = X * X + Y * Y;
    const float theta = orbat::ref_atan2f(sqrtf(x2_plus_y2), Z);
    const float psi = orbat::ref_atan2f(Y, X);
    const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
    const float r = theta + p[4] * theta3 + p[5] * theta5 + p[6] * theta7 + p[7] * theta9;
#ifdef ORB_KB8_DOUBLE_TRIG   // see orbm_project (orbhip.hip).  The device's fp64 cos / sin are ROCm's, within 1 ulp of glibc's but NOT verified equal:
                             // with this switch the windows of config 5 are no longer covered by the bit-exact replicas (parity unpinned)
    u = (float)((double)(p[0] * r) * cos((double)psi) + (double)p[2]);
    v = (float)((double)(p[1] * r) * sin((double)psi) + (double)p[3]);
#else
    u = p[0] * r * orbsc::ref_cosf(psi) + p[2];   // cos / sin on a float: the <math.h> overloads -> cosf / sinf (libm choices)
    v = p[1] * r * orbsc::ref_sinf(psi) + p[3];
#endif
  }
}

// One thread per last-frame keypoint: ORBmatcher.cc:2038-2052 (per pair, recomputed by every thread from scalar loads) and
// :2062-2118 -> (u, v, radius, level window, right coordinate, flags) of query i.
__global__ __launch_bounds__(256) void k_lastframe_project(LastFrameParams P) {
  const int p = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = P.last_n ? min(P.last_n[(size_t)p * P.last_n_stride], P.last_stride) : P.last_n_const;
  if (i >= P.last_stride) return;
  const size_t o = (size_t)p * P.last_stride + i;
  float u = 0.f, v = 0.f, rad = 0.f, ur = 0.f;
  int minl = -1, maxl = -1;
  uint8_t fl = 0;
  if (i < n && P.has_mp[o]) {
    const float *Tcw = P.Tcw + (size_t)p * 16, *Tlw = P.Tlw + (size_t)p * 16;
    const float tcw[3] = {Tcw[3], Tcw[7], Tcw[11]}, tlw[3] = {Tlw[3], Tlw[7], Tlw[11]};
    float twc[3], tlc[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {  // twc = -Rcw.t()*tcw (:2041): generic gemm path, double accumulation, alpha = -1
      double s = 0;
#pragma unroll
      for (int k = 0; k < 3; k++) s += (double)Tcw[k * 4 + a] * (double)tcw[k];
      twc[a] = (float)(s * -1.0);
    }
    dev_mat3_mul_add(Tlw, twc, tlw, tlc);  // :2047
    const bool bForward = tlc[2] > P.mb && !P.bMono, bBackward = -tlc[2] > P.mb && !P.bMono;  // :2051-2052
    const float xw[3] = {P.Xw[3 * o], P.Xw[3 * o + 1], P.Xw[3 * o + 2]};
    float xc[3];
    dev_mat3_mul_add(Tcw, xw, tcw, xc);                      // :2072
    const float invzc = (float)(1.0 / (double)xc[2]);        // :2076
    if (!(invzc < 0)) {
      float ux, vy;
      dev_project(P.cam_type, P.cam, xc[0], xc[1], xc[2], ux, vy);  // :2091
      const bool inside = !(ux < P.min_x || ux > P.max_x) && !(vy < P.min_y || vy > P.max_y);  // :2094-2097
      const int oct = __float_as_int(P.last_kp[7 * o + 5]);
      if (inside && oct >= 0 && oct < P.nlevels) {
        u = ux; v = vy;
        rad = P.th * P.sf[oct];                                   // :2105
        if (bForward) { minl = oct; maxl = -1; }                  // :2113-2118
        else if (bBackward) { minl = 0; maxl = oct; }
        else { minl = oct - 1; maxl = oct + 1; }
        ur = ux - P.mbf * invzc;                                  // :2141
        fl = (uint8_t)(1u | ((P.obs ? (P.obs[o] & 1u) : 1u) << 1));
      }
    }
  }
  P.qu[o] = u; P.qv[o] = v; P.qr[o] = rad; P.qur[o] = ur;
  P.qminl[o] = minl; P.qmaxl[o] = maxl; P.qflags[o] = fl;
}

struct RotPruneParams {
  const float *last_kp;  // query side: angle of LastFrame.mvKeysUn[i] (word 3 of 7)
  int last_stride;
  const int32_t *last_n; int last_n_stride; int last_n_const;
  const float *cur_kp;   // CurrentFrame.mvKeysUn, 7 floats per keypoint
  int frame_stride;
  int32_t *moq;          // match_of_query (in/out: pruned matches become -1)
  int32_t *slot; uint8_t *slot_obs;
  int32_t *nmatches;     // per problem (in/out)
};

// ORBmatcher.cc:2177-2185 (histogram of the rotation between the matched keypoints) + ComputeThreeMaxima (:2416-2458) +
// :2263-2286 (matches outside the three dominant bins are undone).  One workgroup per frame pair; the bins only need their
// sizes (integer LDS atomics), the order inside a bin does not matter for what is kept.
__global__ __launch_bounds__(256) void k_rot_prune(RotPruneParams R) {
  __shared__ int hist[32];
  __shared__ int keep[3];
  __shared__ int removed;
  const int p = blockIdx.x, t = threadIdx.x;
  const int n = R.last_n ? min(R.last_n[(size_t)p * R.last_n_stride], R.last_stride) : R.last_n_const;
  const size_t qo = (size_t)p * R.last_stride, ko = (size_t)p * R.frame_stride;
  if (t < 32) hist[t] = 0;
  if (t == 0) removed = 0;
  __syncthreads();
  const float factor = 1.0f / 30;
  for (int i = t; i < n; i += 256) {
    const int m = R.moq[qo + i];
    if (m < 0) continue;
    float rot = R.last_kp[7 * (qo + i) + 3] - R.cur_kp[7 * (ko + m) + 3];
    if ((double)rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == 30) bin = 0;
    if (bin >= 0 && bin < 30) atomicAdd(&hist[bin], 1);
  }
  __syncthreads();
  if (t == 0) {
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int i = 0; i < 30; i++) {
      const int s = hist[i];
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
      else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
      else if (s > max3) { max3 = s; ind3 = i; }
    }
    if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
    keep[0] = ind1; keep[1] = ind2; keep[2] = ind3;
  }
  __syncthreads();
  const int k0 = keep[0], k1 = keep[1], k2 = keep[2];
  int mine = 0;
  for (int i = t; i < n; i += 256) {
    const int m = R.moq[qo + i];
    if (m < 0) continue;
    float rot = R.last_kp[7 * (qo + i) + 3] - R.cur_kp[7 * (ko + m) + 3];
    if ((double)rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == 30) bin = 0;
    if (bin < 0 || bin >= 30 || bin == k0 || bin == k1 || bin == k2) continue;
    R.slot[ko + m] = -1;
    R.slot_obs[ko + m] = 0;
    R.moq[qo + i] = -1;
    mine++;
  }
  if (mine) atomicAdd(&removed, mine);
  __syncthreads();
  if (t == 0 && removed) R.nmatches[p] -= removed;
}

// ------------------------------------------------------------------------------------------------------------
// G0: Frame::UndistortKeyPoints (Frame.cc:837-870) for frames resident in HBM - the step between operator() and the
// searches in the Frame constructor.  cv::undistortPoints(pts, K, D, R = I, P = K) restated: five
// fixed-point iterations in double, every operation in source order (no contraction; fp64 division and the
// double -> float conversions are IEEE on gfx950), i.e. the same bits as the host function orbm_undistort_keypoints,
// which tests/test_gpu_distort.py checks.  One thread per keypoint; only pt changes (:862-868).
// ------------------------------------------------------------------------------------------------------------
struct UndistortParams {
  const float *keys;           // orbx_keypoint_t AoS viewed as floats, frame f at element offset f * key_stride
  float *keys_un;              // may alias keys
  int key_stride;
  const int32_t *counts; int count_stride; int count_const;
  float K[4], D[5]; int nD;
};

__host__ __device__ __forceinline__ void undistort_point(double u, double v, const float *K, const float *D, int nD, float *ou, float *ov) {
  const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
  const double k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = nD > 4 ? D[4] : 0.0;
  double x = (u - cx) * (1. / fx), y = (v - cy) * (1. / fy);
  const double x0 = x, y0 = y;
  for (int it = 0; it < 5; it++) {
    const double r2 = x * x + y * y;
    const double icdist = 1. / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
    const double dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
    const double dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
    x = (x0 - dx) * icdist;
    y = (y0 - dy) * icdist;
  }
  *ou = (float)(x * fx + cx);
  *ov = (float)(y * fy + cy);
}

__global__ __launch_bounds__(256) void k_undistort(UndistortParams U) {
  const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  const int n = U.counts ? U.counts[(size_t)f * U.count_stride] : U.count_const;
  if (i >= n) return;
  const size_t o = ((size_t)f * U.key_stride + i) * 7;
  float k[7];
#pragma unroll
  for (int t = 0; t < 7; t++) k[t] = U.keys[o + t];
  if (U.D[0] != 0.0f) undistort_point((double)k[0], (double)k[1], U.K, U.D, U.nD, &k[0], &k[1]);   // :839-843: D[0] == 0 copies
#pragma unroll
  for (int t = 0; t < 7; t++) U.keys_un[o + t] = k[t];
}
