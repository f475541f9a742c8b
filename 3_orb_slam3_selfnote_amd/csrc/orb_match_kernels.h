// orb_match_kernels.h -- Hamming / projection-search kernels of the ORBmatcher path (K8, K9 of SURVEY.md 2.2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MATCH_NT 256

struct MatchProblemSet {
  // frame side (keypoint-indexed, element offset p*frame_stride per problem)
  const float *kp;            // orbx_keypoint_t AoS viewed as floats (7 per keypoint)
  const uint8_t *desc;        // n x 32
  const float *u_right;       // or NULL
  int frame_stride;
  const int32_t *frame_n; int frame_n_stride; int frame_n_const;  // live count: device array or constant
  float min_x, min_y, inv_w, inv_h;  // mnMinX, mnMinY, mfGridElementWidthInv/HeightInv (Frame.cc:379-380)
  // query side
  const uint8_t *qdesc;
  const float *qu, *qv, *qr, *qur;
  const int32_t *qminl, *qmaxl;
  const uint8_t *qflags;
  int query_stride;
  const int32_t *query_n; int query_n_stride; int query_n_const;
  // options
  float nnratio; int th_dist; int use_second;
  // in/out
  int32_t *slot; uint8_t *slot_obs; int32_t *match_of_query; int32_t *best_dist; int32_t *nmatches;
};

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int o) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  lo = __shfl_xor(lo, o, 64);
  hi = __shfl_xor(hi, o, 64);
  return ((unsigned long long)hi << 32) | lo;
}

// merge two (best <= second) pairs
__device__ __forceinline__ void merge2(unsigned long long &b, unsigned long long &s, unsigned long long ob,
                                       unsigned long long os) {
  unsigned long long nb = b < ob ? b : ob;
  unsigned long long mx = b < ob ? ob : b;
  unsigned long long ms = s < os ? s : os;
  s = mx < ms ? mx : ms;
  b = nb;
}

// One workgroup per (frame, query set).  Each thread keeps CPT candidate keypoints (descriptor, position, octave,
// grid cell) in registers; queries are resolved strictly in order -- the reference's loops carry a dependency
// through F.mvpMapPoints (ORBmatcher.cc:89-91/:130, :2135-2137/:2162) -- with one workgroup-wide
// (best, second) reduction per query.
//
// Candidate enumeration order of Frame::GetFeaturesInArea (Frame.cc:781-809: ix outer, iy inner, insertion order)
// decides argmin ties (strict <, first minimum wins): it is carried in the reduction key
//   key = dist<<40 | (ix*48+iy)<<28 | idx<<8 | octave.
template <int CPT>
__global__ __launch_bounds__(MATCH_NT) void k_search_by_projection(MatchProblemSet M) {
  __shared__ unsigned long long sRed[2][2 * (MATCH_NT / 64)];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int p = blockIdx.x;
  const int n = M.frame_n ? M.frame_n[(size_t)p * M.frame_n_stride] : M.frame_n_const;
  const int nq = M.query_n ? M.query_n[(size_t)p * M.query_n_stride] : M.query_n_const;
  const size_t fo = (size_t)p * M.frame_stride, qo = (size_t)p * M.query_stride;
  const float *kp = M.kp + fo * 7;
  const uint32_t *desc = reinterpret_cast<const uint32_t *>(M.desc + fo * 32);
  int32_t *slot = M.slot + fo;
  uint8_t *slot_obs = M.slot_obs + fo;

  uint32_t cd[CPT][8];
  float cx[CPT], cy[CPT], cur[CPT];
  int coct[CPT], ccell[CPT];
  uint32_t claimed = 0;
#pragma unroll
  for (int j = 0; j < CPT; j++) {
    const int i = j * MATCH_NT + tid;
    ccell[j] = -1;
    cx[j] = cy[j] = 0.f; cur[j] = -1.f; coct[j] = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) cd[j][w] = 0;
    if (i < n) {
      cx[j] = kp[(size_t)i * 7];
      cy[j] = kp[(size_t)i * 7 + 1];
      coct[j] = __float_as_int(kp[(size_t)i * 7 + 5]);
      if (M.u_right) cur[j] = M.u_right[fo + i];
#pragma unroll
      for (int w = 0; w < 8; w++) cd[j][w] = desc[(size_t)i * 8 + w];
      // Frame::PosInGrid, Frame.cc:815-825
      int gx = (int)roundf((cx[j] - M.min_x) * M.inv_w), gy = (int)roundf((cy[j] - M.min_y) * M.inv_h);
      if (gx >= 0 && gx < 64 && gy >= 0 && gy < 48) ccell[j] = gx * 48 + gy;
      if (slot[i] >= 0 && slot_obs[i]) claimed |= 1u << j;
    }
  }
  int nmatches = 0;
  for (int q = 0; q < nq; q++) {
    const uint8_t fl = M.qflags ? M.qflags[qo + q] : (uint8_t)3;
    unsigned long long best = ~0ull, second = ~0ull;
    const float u = M.qu[qo + q], v = M.qv[qo + q], r = M.qr[qo + q];
    const int minl = M.qminl[qo + q], maxl = M.qmaxl[qo + q];
    // Frame::GetFeaturesInArea cell window, Frame.cc:755-777
    int cx0 = max(0, (int)floorf((u - M.min_x - r) * M.inv_w));
    int cx1 = min(63, (int)ceilf((u - M.min_x + r) * M.inv_w));
    int cy0 = max(0, (int)floorf((v - M.min_y - r) * M.inv_h));
    int cy1 = min(47, (int)ceilf((v - M.min_y + r) * M.inv_h));
    const bool live = (fl & 1) && cx0 < 64 && cx1 >= 0 && cy0 < 48 && cy1 >= 0;
    const bool checkLevels = (minl > 0) || (maxl >= 0);
    if (live) {
      const uint32_t *qd = reinterpret_cast<const uint32_t *>(M.qdesc + (qo + q) * 32);
      uint32_t q0 = qd[0], q1 = qd[1], q2 = qd[2], q3 = qd[3], q4 = qd[4], q5 = qd[5], q6 = qd[6], q7 = qd[7];
      const float ur = M.qur ? M.qur[qo + q] : 0.f;
#pragma unroll
      for (int j = 0; j < CPT; j++) {
        const int cell = ccell[j];
        if (cell < 0) continue;
        const int gx = cell / 48, gy = cell - gx * 48;
        bool ok = gx >= cx0 && gx <= cx1 && gy >= cy0 && gy <= cy1;
        if (checkLevels) ok = ok && coct[j] >= minl && (maxl < 0 || coct[j] <= maxl);
        ok = ok && fabsf(cx[j] - u) < r && fabsf(cy[j] - v) < r;
        ok = ok && !((claimed >> j) & 1u);
        if (M.u_right && cur[j] > 0.f) ok = ok && !(fabsf(ur - cur[j]) > r);  // ORBmatcher.cc:93-98, :2139-2146
        if (ok) {
          int dist = __popc(cd[j][0] ^ q0) + __popc(cd[j][1] ^ q1) + __popc(cd[j][2] ^ q2) + __popc(cd[j][3] ^ q3) +
                     __popc(cd[j][4] ^ q4) + __popc(cd[j][5] ^ q5) + __popc(cd[j][6] ^ q6) + __popc(cd[j][7] ^ q7);
          unsigned long long key = ((unsigned long long)dist << 40) | ((unsigned long long)cell << 28) |
                                   ((unsigned long long)(j * MATCH_NT + tid) << 8) | (unsigned long long)(coct[j] & 0xff);
          if (key < best) { second = best; best = key; }
          else if (key < second) second = key;
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      unsigned long long ob = shfl_xor_u64(best, o), os = shfl_xor_u64(second, o);
      merge2(best, second, ob, os);
    }
    const int par = q & 1;
    if (lane == 0) { sRed[par][2 * wid] = best; sRed[par][2 * wid + 1] = second; }
    __syncthreads();
    best = sRed[par][0]; second = sRed[par][1];
#pragma unroll
    for (int w = 1; w < MATCH_NT / 64; w++) merge2(best, second, sRed[par][2 * w], sRed[par][2 * w + 1]);
    // decision, identical in every thread
    int bestDist = 256, bestIdx = -1, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1;
    if (best != ~0ull) { bestDist = (int)(best >> 40); bestIdx = (int)((best >> 8) & 0xfffff); bestLevel = (int)(best & 0xff); }
    if (second != ~0ull) { bestDist2 = (int)(second >> 40); bestLevel2 = (int)(second & 0xff); }
    bool accept = bestIdx >= 0 && bestDist <= M.th_dist;
    if (accept && M.use_second && bestLevel == bestLevel2 && (float)bestDist > M.nnratio * (float)bestDist2) accept = false;
    if (accept) {
      nmatches++;
      if ((bestIdx % MATCH_NT) == tid) {
        const int j = bestIdx / MATCH_NT;
        const uint32_t ob = (fl >> 1) & 1u;
        claimed = (claimed & ~(1u << j)) | (ob << j);
        slot[bestIdx] = q;
        slot_obs[bestIdx] = (uint8_t)ob;
      }
    }
    if (tid == 0) {
      if (M.match_of_query) M.match_of_query[qo + q] = accept ? bestIdx : -1;
      if (M.best_dist) M.best_dist[qo + q] = live ? bestDist : 256;
    }
  }
  if (tid == 0 && M.nmatches) M.nmatches[p] = nmatches;
}

// K8 brute force: dist[i][j] = popcount(q_i ^ c_j).  Candidates staged through LDS in 256-descriptor (8 KB) chunks.
__global__ __launch_bounds__(256) void k_hamming_matrix(const uint32_t *q, int nq, const uint32_t *c, int nc, uint16_t *dist) {
  __shared__ uint32_t sC[256 * 9];  // +1 word pad per descriptor: conflict-free column reads
  const int tid = threadIdx.x;
  const int qi = blockIdx.x * 256 + tid;
  uint32_t qd[8];
#pragma unroll
  for (int w = 0; w < 8; w++) qd[w] = qi < nq ? q[(size_t)qi * 8 + w] : 0u;
  for (int base = 0; base < nc; base += 256) {
    const int m = min(256, nc - base);
    for (int idx = tid; idx < m * 8; idx += 256) sC[(idx >> 3) * 9 + (idx & 7)] = c[(size_t)base * 8 + idx];
    __syncthreads();
    if (qi < nq) {
      for (int j = 0; j < m; j++) {
        const uint32_t *cc = &sC[j * 9];
        int d = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) d += __popc(qd[w] ^ cc[w]);
        dist[(size_t)qi * nc + base + j] = (uint16_t)d;
      }
    }
    __syncthreads();
  }
}
