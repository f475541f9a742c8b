// orb_match_kernels.h -- Hamming / projection-search kernels of the ORBmatcher path (K8, K9 of SURVEY.md 2.2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "orb_mfma_util.h"

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
  // fisheye-stereo frames (Nleft != -1), all NULL / 0 otherwise: keypoints [nleft, n) are the right image's; qside[q] = 1
  // restricts query q to them (0: to the left ones); partner[k] = stereo partner of keypoint k in the other image or -1
  // (mvLeftToRightMatch / mvRightToLeftMatch); queries come in (left, right) pairs of one map point at (2j, 2j+1):
  // couple 1 = the right query is dropped when the left one failed the ratio test (ORBmatcher.cc:125 `continue`),
  // couple 2 = ... when the left one's window was empty (:2126); qany[q] = window of query q non-empty (scan -> resolve);
  // serial = resolve one query at a time from a fresh list (exact when a partner write can release a claim).
  int nleft; const uint8_t *qside; const int32_t *partner; int couple; uint8_t *qany; int serial;
  // ORBmatcher::Fuse (ORBmatcher.cc:1425-1658): per-candidate chi-square gate on the reprojection error, :1585-1608
  // (k_match_scan mode SCAN_FUSE); u_right = mvuRight of the keyframe, qur = projected right coordinate
  float inv_sigma2[16];
  int npairs, scan_qblocks;  // problems of the launch; 256-query blocks per problem (k_match_scan's 1-D, XCD-aware grid)
  long long *dbg;  // diagnostic builds (-DRESOLVE_STAMPS) only: per-problem cycle sums; never read by the product
};

// ---------------------------------------------------------------------------------------------------------------
// Projection search = candidate lists (k_match_scan, or k_match_walk for tracking-sized windows) + k_match_resolve.
//
// The reference resolves queries strictly in order because a keypoint claimed by an earlier map point is skipped by
// later ones (ORBmatcher.cc:89-91/:130, :2135-2137/:2162).  Claims only ever REMOVE candidates, so:
//   k_match_scan     (fully parallel, all the Hamming work): for every query the TOPK smallest reduction keys over all
//                    candidates that pass the static tests (grid-cell window, level window, |dx|,|dy| < r, stereo
//                    check, not held by an occupant with observations).
//   k_match_resolve  (one workgroup per frame pair, in query order): the first two entries of a query's list that are
//                    still unclaimed ARE its best / second-best at its turn; the query is rescanned (exact, all waves
//                    of the workgroup split the keypoints) only if fewer than two survive AND an unlisted candidate
//                    could still change the decision.  Then the reference's accept rule (:124-130 resp. :2159-2162)
//                    and the claim.
//
// Candidate enumeration order of Frame::GetFeaturesInArea (Frame.cc:781-809: ix outer, iy inner, insertion order)
// decides argmin ties (strict <, first minimum wins); it is carried in the key below the distance:
//   Key32 (frames of <= 2048 keypoints): dist<<23 | (ix*48+iy)<<11 | idx     -- one v_min_u32 per comparison
//   Key64 (up to 32768 keypoints):       dist<<40 | (ix*48+iy)<<28 | idx<<8
// ---------------------------------------------------------------------------------------------------------------
#define MATCH_TOPK 8
#define MATCH_CH 128
// Wavefronts of a k_match_resolve workgroup.  The resolver itself is ONE wavefront (the claims are sequential); the others only serve its
// list requests.  Four, not eight: the fused form needs 248 vector registers per lane, so a workgroup of eight wavefronts (two per SIMD) took
// the whole register file of its CU and nothing of the other pipelines could run beside it for the kernel's 0.3 ms - one workgroup per CU,
// 256 frame pairs = every CU of the chip.  With one wavefront per SIMD half the registers stay free (four k_fast wavefronts per SIMD): alone
// the kernel is 5 % slower (0.281 -> 0.296 ms, the list build is spread over half the wavefronts), the four-pipeline bench 5 % faster
// (190.8 k -> 200.5 k frames/s); two wavefronts: 0.39 ms, 193 k.  The other forms (115-119 registers: two wavefronts per SIMD leave half the file
// free as well) keep eight: the wide form's lanes ARE the workgroup's threads (512 queries per super-chunk).
#define RESOLVE_NW_OF(fused) ((fused) ? 4 : 8)

struct Key32 {
  typedef uint32_t T;
  static constexpr T NONE = 0xffffffffu;
  static __device__ __forceinline__ T make(int dist, uint32_t cell, int idx) { return ((uint32_t)dist << 23) | (cell << 11) | (uint32_t)idx; }
  static __device__ __forceinline__ int dist(T k) { return (int)(k >> 23); }
  static __device__ __forceinline__ int idx(T k) { return (int)(k & 0x7ffu); }
  static __device__ __forceinline__ T readlane(T v, int l) { return (T)__builtin_amdgcn_readlane((int)v, l); }
};
struct Key64 {
  typedef unsigned long long T;
  static constexpr T NONE = ~0ull;
  static __device__ __forceinline__ T make(int dist, uint32_t cell, int idx) { return ((T)dist << 40) | ((T)cell << 28) | ((T)idx << 8); }
  static __device__ __forceinline__ int dist(T k) { return (int)(k >> 40); }
  static __device__ __forceinline__ int idx(T k) { return (int)((k >> 8) & 0xfffff); }
  static __device__ __forceinline__ T readlane(T v, int l) {
    uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, l), hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((T)hi << 32) | lo;
  }
};

// popcount(x) + acc in one instruction (v_bcnt_u32_b32 D = bcnt(S0) + S1)
__device__ __forceinline__ int popc_acc(uint32_t x, int acc) {
  int d;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(acc));
  return d;
}

struct CandMeta { float x, y; uint32_t bits; float ur; };  // bits: octave | gx<<8 | gy<<16 | usable<<24 | in-grid<<25

__device__ __forceinline__ uint32_t cand_bits(float x, float y, int oct, bool claimed, const MatchProblemSet &M) {
  // Frame::PosInGrid, Frame.cc:815-825
  int gx = (int)roundf((x - M.min_x) * M.inv_w), gy = (int)roundf((y - M.min_y) * M.inv_h);
  bool in = gx >= 0 && gx < 64 && gy >= 0 && gy < 48;
  return (uint32_t)(oct & 0xff) | ((uint32_t)(gx & 0xff) << 8) | ((uint32_t)(gy & 0xff) << 16) | ((in && !claimed) ? (1u << 24) : 0u) |
         (in ? (1u << 25) : 0u);
}
__device__ __forceinline__ uint32_t cell_of(uint32_t bits) { return ((bits >> 8) & 0xff) * 48u + ((bits >> 16) & 0xff); }

struct QueryWin { float u, v, r, ur; int minl, maxl, cx0, cx1, cy0, cy1; bool live, checkLevels, stereo; };

__device__ __forceinline__ QueryWin load_query(const MatchProblemSet &M, size_t qo, int q) {
  QueryWin w;
  const uint8_t fl = M.qflags ? M.qflags[qo + q] : (uint8_t)3;
  w.u = M.qu[qo + q]; w.v = M.qv[qo + q]; w.r = M.qr[qo + q];
  w.minl = M.qminl[qo + q]; w.maxl = M.qmaxl[qo + q];
  w.ur = M.qur ? M.qur[qo + q] : 0.f;
  // Frame::GetFeaturesInArea cell window, Frame.cc:755-777
  w.cx0 = max(0, (int)floorf((w.u - M.min_x - w.r) * M.inv_w));
  w.cx1 = min(63, (int)ceilf((w.u - M.min_x + w.r) * M.inv_w));
  w.cy0 = max(0, (int)floorf((w.v - M.min_y - w.r) * M.inv_h));
  w.cy1 = min(47, (int)ceilf((w.v - M.min_y + w.r) * M.inv_h));
  w.live = (fl & 1) && w.cx0 < 64 && w.cx1 >= 0 && w.cy0 < 48 && w.cy1 >= 0;
  w.checkLevels = (w.minl > 0) || (w.maxl >= 0);
  w.stereo = M.u_right != nullptr;
  return w;
}

__device__ __forceinline__ bool cand_passes(const QueryWin &w, float x, float y, uint32_t bits, float cur) {
  const int oct = bits & 0xff, gx = (bits >> 8) & 0xff, gy = (bits >> 16) & 0xff;
  bool ok = (bits >> 24) & 1u;
  ok = ok && gx >= w.cx0 && gx <= w.cx1 && gy >= w.cy0 && gy <= w.cy1;
  if (w.checkLevels) ok = ok && oct >= w.minl && (w.maxl < 0 || oct <= w.maxl);
  ok = ok && fabsf(x - w.u) < w.r && fabsf(y - w.v) < w.r;
  if (w.stereo && cur > 0.f) ok = ok && !(fabsf(w.ur - cur) > w.r);  // ORBmatcher.cc:93-98, :2139-2146
  return ok;
}

// "Open" query: the cell window is the whole 64 x 48 grid, no level filter, and the float window reaches more than one grid
// cell beyond the image bounds on every side, so |x - u| < r and |y - v| < r hold for every keypoint PosInGrid accepts
// (those lie within half a cell of the bounds): GetFeaturesInArea returns every in-grid keypoint.
__device__ __forceinline__ bool query_is_open(const MatchProblemSet &M, const QueryWin &w) {
  const float cellw = 1.0f / M.inv_w, cellh = 1.0f / M.inv_h;
  return w.live && !w.checkLevels && w.cx0 == 0 && w.cx1 == 63 && w.cy0 == 0 && w.cy1 == 47 &&
         w.u - w.r < M.min_x - cellw && w.u + w.r > M.min_x + 65.0f * cellw && w.v - w.r < M.min_y - cellh &&
         w.v + w.r > M.min_y + 49.0f * cellh;
}
// Workgroup-wide vote (every thread of the workgroup calls it with its own query's flags): true <=> the block has a live query
// and all its live queries are open - such blocks of monocular Key32 problems are served by k_match_scan_mfma (orb_match_mfma.h).
__device__ __forceinline__ bool block_is_open(bool live, bool open) {
  const int notOpen = __syncthreads_or(live && !open);
  const int anyLive = __syncthreads_or(live);
  return anyLive && !notOpen;
}

// ---- walk or scan: see k_match_walk below
#define GRID_CELLS (64 * 48)
#define WALK_MAX_CELLS 256
#define WALK_MAX_N 2048
enum { SCAN_AUTO = 0, SCAN_DENSE = 1, SCAN_WALK = 2 };

// Workgroup-wide vote (all NT threads of the workgroup must call it): true <=> pair p is served by k_match_walk.
template <int NT = MATCH_NT>
__device__ __forceinline__ bool pair_walks(const MatchProblemSet &M, int p, int n, int nq, int force, int *sVote /* one LDS word */) {
  if (force != SCAN_AUTO) return force == SCAN_WALK;
  if (n > WALK_MAX_N) return false;
  const size_t qo = (size_t)p * M.query_stride;
  if (threadIdx.x == 0) *sVote = 0;
  __syncthreads();
  int big = 0;
  for (int q = threadIdx.x; q < nq; q += NT) {
    const uint8_t fl = M.qflags ? M.qflags[qo + q] : (uint8_t)3;
    const float u = M.qu[qo + q], v = M.qv[qo + q], r = M.qr[qo + q];
    const int cx0 = max(0, (int)floorf((u - M.min_x - r) * M.inv_w)), cx1 = min(63, (int)ceilf((u - M.min_x + r) * M.inv_w));
    const int cy0 = max(0, (int)floorf((v - M.min_y - r) * M.inv_h)), cy1 = min(47, (int)ceilf((v - M.min_y + r) * M.inv_h));
    const bool live = (fl & 1) && cx0 < 64 && cx1 >= 0 && cy0 < 48 && cy1 >= 0;
    if (live && (cx1 - cx0 + 1) * (cy1 - cy0 + 1) > WALK_MAX_CELLS) big = 1;
  }
  if (__builtin_amdgcn_ballot_w64(big != 0) && (threadIdx.x & 63) == 0) *sVote = 1;
  __syncthreads();
  const bool walks = *sVote == 0;
  __syncthreads();   // the caller may reuse the word
  return walks;
}

// MODE selects what is compiled into the inner loop besides the window test:
//   SCAN_PLAIN    nothing (monocular frames)
//   SCAN_UR       the right-coordinate test of frames with mvuRight (rectified stereo / RGB-D), ORBmatcher.cc:93-98
//   SCAN_FISHEYE  fisheye-stereo problem: image restriction per query, window-non-empty flags
//   SCAN_FUSE     ORBmatcher::Fuse's chi-square gate on the reprojection error, ORBmatcher.cc:1585-1608
enum { SCAN_PLAIN = 0, SCAN_UR = 1, SCAN_FISHEYE = 2, SCAN_FUSE = 3 };
// Latency mode (few problems in flight): the candidate chunks of one problem are split over gridDim.z workgroups per query
// block ("slices"), each writing its own sorted top-8 per query at topk + slice * slice_stride; k_topk_merge folds them.
template <typename KT, int MODE>
__global__ __launch_bounds__(MATCH_NT) void k_match_scan(MatchProblemSet M, typename KT::T *topk, size_t slice_stride, int force, int mfma /* open blocks belong to k_match_scan_mfma */,
                                                         const uint32_t *pairflag /* or NULL; pairflag[p] != 0: the fused k_match_resolve makes this pair's lists */) {
  typedef typename KT::T K;
  __shared__ uint4 sDesc[MATCH_CH * 2];
  __shared__ CandMeta sMeta[MATCH_CH];
  __shared__ int sVote;
  const int tid = threadIdx.x;
  // XCD-aware placement: workgroups are dealt round-robin over the 8 XCDs by linear index, so problem p = (b % 8) + 8 * (...) and
  // query block qb = (b / 8) % qblocks put ALL query blocks of one problem on one XCD - its candidates are fetched into ONE L2
  // instead of up to four (a (query block, problem) grid spread a problem's four query blocks over four XCDs: 4.5x the bytes).
  const unsigned b = blockIdx.x, qblocks = (unsigned)M.scan_qblocks;   // launch: grid = (8 * qblocks * ceil(npairs / 8), 1, slices)
  const int qb = (int)((b >> 3) % qblocks), p = (int)((b & 7u) + 8u * (b / (8u * qblocks)));
  if (p >= M.npairs) return;
  const int n = M.frame_n ? M.frame_n[(size_t)p * M.frame_n_stride] : M.frame_n_const;
  const int nq = M.query_n ? M.query_n[(size_t)p * M.query_n_stride] : M.query_n_const;
  if ((int)(qb * MATCH_NT) >= nq) return;
  if (pairflag && pairflag[p]) return;
  if (force != SCAN_DENSE && pair_walks(M, p, n, nq, force, &sVote)) return;   // k_match_walk serves this pair
  const size_t fo = (size_t)p * M.frame_stride, qo = (size_t)p * M.query_stride;
  const float *kp = M.kp + fo * 7;
  const uint4 *desc = reinterpret_cast<const uint4 *>(M.desc + fo * 32);
  const int q = qb * MATCH_NT + tid;
  QueryWin w;
  w.live = false;
  uint32_t qd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (q < nq) {
    w = load_query(M, qo, q);
    const uint4 *qp = reinterpret_cast<const uint4 *>(M.qdesc + (qo + q) * 32);
    uint4 a = qp[0], b = qp[1];
    qd[0] = a.x; qd[1] = a.y; qd[2] = a.z; qd[3] = a.w; qd[4] = b.x; qd[5] = b.y; qd[6] = b.z; qd[7] = b.w;
  }
  const bool open = w.live && query_is_open(M, w);
  if (MODE == SCAN_PLAIN && sizeof(K) == 4 && mfma && block_is_open(w.live, open)) return;
  K top[MATCH_TOPK];
#pragma unroll
  for (int j = 0; j < MATCH_TOPK; j++) top[j] = KT::NONE;
  // fisheye-stereo: a query sees only the keypoints of its own image
  constexpr bool STEREO = MODE == SCAN_FISHEYE, UR = MODE == SCAN_UR;
  const int nleft = STEREO && M.qside ? M.nleft : n;
  const bool sideR = STEREO && M.qside && q < nq && M.qside[qo + q] != 0;
  const bool wantAny = STEREO && M.qany != nullptr;
  bool any = false;
  // GetFeaturesInArea's level filter (Frame.cc:779, :794-801) as a closed interval; open ends when it is disabled
  const int minlE = w.live && w.checkLevels ? w.minl : -1000;
  const int maxlE = w.live && w.checkLevels && w.maxl >= 0 ? w.maxl : 1000;
  // The three range tests (octave, grid column, grid row) as ONE byte-parallel comparison: candidates carry
  // octave | gx << 8 | gy << 16 in `bits`; with bit 7 of every byte pre-set a byte-wise subtraction keeps that bit iff
  // no borrow occurred, i.e. iff the byte is >= its bound.  All operands are < 128 (gx < 64, gy < 48, octave < 16; open
  // ends clamp to 0 / 127), so no borrow crosses a byte.  Keypoints outside the grid may carry larger bytes; they are
  // rejected by their usable / in-grid bit whatever the comparison yields.
  auto c7 = [](int v) { return (uint32_t)min(max(v, 0), 127); };
  const uint32_t LO = c7(minlE) | (c7(w.cx0) << 8) | (c7(w.cy0) << 16);
  const uint32_t HI7 = (c7(maxlE) | (c7(w.cx1) << 8) | (c7(w.cy1) << 16)) | 0x808080u;
  K pend = KT::NONE;
  auto insert = [&](K t) {   // sorted insertion into top[]; a no-op for t = NONE
#pragma unroll
    for (int j = 0; j < MATCH_TOPK; j++) {
      const K lo = t < top[j] ? t : top[j];
      const K hi = t < top[j] ? top[j] : t;
      top[j] = lo;
      t = hi;
    }
  };
  int base = 0;   // first candidate of the chunk in LDS
  // Parked insertion (both candidate loops below): a key below the lane's current 8th best is parked in `pend`; the sorted
  // insertion (16 min/max) runs for the whole wavefront only when some lane would have to park a second one.  With 64 lanes
  // nearly every candidate improves SOME lane's list, so inserting on the spot executes the network for almost every candidate;
  // parked, it runs about once per ten such events.  top[7] is an upper bound of the true 8th best meanwhile, so nothing is lost.
  // A wavefront whose live lanes are all open (query_is_open) skips the per-candidate tests.
  const bool allOpen = MODE == SCAN_PLAIN && __builtin_amdgcn_ballot_w64(w.live && !open) == 0ull;
  const int nchunks = (n + MATCH_CH - 1) / MATCH_CH;
  const int chunk0 = (int)(((long long)nchunks * blockIdx.z) / gridDim.z), chunk1 = (int)(((long long)nchunks * (blockIdx.z + 1)) / gridDim.z);
#ifdef SCAN_STAMPS
  long long st_stage = 0, st_comp = 0, st_t = __builtin_readcyclecounter();
  const long long st_begin = st_t;
#define SSTAMP(acc) do { const long long t_ = __builtin_readcyclecounter(); acc += t_ - st_t; st_t = t_; } while (0)
#else
#define SSTAMP(acc) do {} while (0)
#endif
  for (base = chunk0 * MATCH_CH; base < min(n, chunk1 * MATCH_CH); base += MATCH_CH) {
    const int m = min(MATCH_CH, n - base);
    SSTAMP(st_comp);
    __syncthreads();
    if (tid < 2 * m) sDesc[tid] = desc[(size_t)base * 2 + tid];
    const int m4 = (m + 3) & ~3;   // the candidate loop runs four at a time; the tail is padded with unusable entries (bits = 0)
    if (tid < m4) {
      CandMeta c = {0.f, 0.f, 0u, -1.f};
      if (tid < m) {
        const int i = base + tid;
        c.x = kp[(size_t)i * 7];
        c.y = kp[(size_t)i * 7 + 1];
        const int oct = __float_as_int(kp[(size_t)i * 7 + 5]);
        c.ur = M.u_right ? M.u_right[fo + i] : -1.f;
        const bool claimed = M.slot[fo + i] >= 0 && M.slot_obs[fo + i];
        c.bits = cand_bits(c.x, c.y, oct, claimed, M);
      }
      sMeta[tid] = c;
    }
    __syncthreads();
    SSTAMP(st_stage);
    if (w.live) {
      const int m4u = __builtin_amdgcn_readfirstlane(m4);
      if (MODE == SCAN_PLAIN && allOpen) {
        // Every live lane's window contains the whole grid and has no level filter (BASELINE's 1000x1000 stress setting, or a
        // relocalisation-style wide search): GetFeaturesInArea returns every in-grid keypoint, so the only test left is the
        // candidate's own usable bit, which is the same for all lanes (a scalar branch).
        // Four candidates per trip, nothing conditional in front of their distances: eight LDS broadcast reads in flight together
        // (one candidate at a time left each wavefront waiting out its own reads), and ONE vote per trip on "does any of the four
        // enter some lane's list" - after the first few hundred candidates it hardly ever does.  19 vector instructions per
        // candidate and wavefront (8 xor, 8 accumulating popcounts, 3 for the key); tools/scan_stamps.py.  The descriptors are
        // the same for all 64 lanes and could come through the scalar cache into SGPRs instead (tried: s_load_dwordx8 x 4 per
        // trip), but a vector instruction that reads a scalar register issues at the half rate (profiles/valu_calib.json), which
        // costs the eight xors more than the LDS reads do: 0.262 ms against 0.238.
        const int baseu = __builtin_amdgcn_readfirstlane(base);
        auto hamming = [&](const uint4 a, const uint4 b) {
          int d0 = popc_acc(a.x ^ qd[0], 0), d1 = popc_acc(b.x ^ qd[4], 0);
          d0 = popc_acc(a.y ^ qd[1], d0); d1 = popc_acc(b.y ^ qd[5], d1);
          d0 = popc_acc(a.z ^ qd[2], d0); d1 = popc_acc(b.z ^ qd[6], d1);
          d0 = popc_acc(a.w ^ qd[3], d0); d1 = popc_acc(b.w ^ qd[7], d1);
          return d0 + d1;
        };
        auto key_of = [&](int dist, uint32_t bits, int idx) -> K {   // NONE for a keypoint outside the grid or already held (uniform)
          const K unusable = (K)(((bits >> 24) & 1u)) - (K)1;                      // 0 or all ones; scalar arithmetic: bits and idx are uniform
          return KT::make(dist, 0u, 0) | KT::make(0, cell_of(bits), idx) | unusable;
        };
        auto park = [&](K t) {
          const bool pass = t < top[MATCH_TOPK - 1];
          const unsigned long long passMask = __builtin_amdgcn_ballot_w64(pass);   // parked insertion, see above
          if (passMask) {
            if (passMask & __builtin_amdgcn_ballot_w64(pend != KT::NONE)) {
              insert(pend);
              pend = KT::NONE;
            }
            pend = pass ? t : pend;
          }
        };
        for (int c = 0; c < m4u; c += 4) {
          // (entries past the chunk's last candidate are unusable, bits = 0: whatever their LDS rows hold yields the key NONE)
          const uint4 a0 = sDesc[2 * c], b0 = sDesc[2 * c + 1], a1 = sDesc[2 * c + 2], b1 = sDesc[2 * c + 3];
          const uint4 a2 = sDesc[2 * c + 4], b2 = sDesc[2 * c + 5], a3 = sDesc[2 * c + 6], b3 = sDesc[2 * c + 7];
          const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sMeta[c].bits), s1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sMeta[c + 1].bits);
          const uint32_t s2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sMeta[c + 2].bits), s3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sMeta[c + 3].bits);
          const K t0 = key_of(hamming(a0, b0), s0, baseu + c), t1 = key_of(hamming(a1, b1), s1, baseu + c + 1);
          const K t2 = key_of(hamming(a2, b2), s2, baseu + c + 2), t3 = key_of(hamming(a3, b3), s3, baseu + c + 3);
          const K tmin = min(min(t0, t1), min(t2, t3));
          if (__builtin_amdgcn_ballot_w64(tmin < top[MATCH_TOPK - 1])) { park(t0); park(t1); park(t2); park(t3); }
        }
      } else {
      for (int c0 = 0; c0 < m4u; c0 += 4)
#pragma unroll
      for (int cu = 0; cu < 4; cu++) {
        const int c = c0 + cu;
        const CandMeta cm = sMeta[c];
        // cand_passes() in sign-bit arithmetic (one compare at the end instead of a dozen compare/and chains):
        // an integer term is negative iff its range test is violated; |d| - r is negative iff the window test passes.
        const int oct = cm.bits & 0xff;
        const uint32_t G = cm.bits & 0xffffffu;
        const uint32_t inRange = ((G | 0x808080u) - LO) & (HI7 - G) & 0x808080u;   // bit 7 of byte b set iff LO_b <= G_b <= HI_b
        const int geo = (int)(inRange ^ 0x808080u) - 1;                              // negative iff all three ranges hold
        const int fpass = __float_as_int(fabsf(cm.x - w.u) - w.r) & __float_as_int(fabsf(cm.y - w.v) - w.r);
        const bool sok = !STEREO || (base + c >= nleft) == sideR;       // candidate index is uniform: a mask select
        if (STEREO && wantAny) any = any || (sok && (fpass & geo & (int)(cm.bits << 6)) < 0);  // GetFeaturesInArea alone (bit 25 = in grid)
        bool ok = sok && (fpass & geo & (int)(cm.bits << 7)) < 0;      // bit 24 = usable (in grid, not pre-occupied)
        if (UR) ok = ok && !(cm.ur > 0.f && fabsf(w.ur - cm.ur) > w.r);        // ORBmatcher.cc:93-98, :2139-2146
        if (MODE == SCAN_FUSE) {                                          // ORBmatcher.cc:1585-1608
          const float ex = w.u - cm.x, ey = w.v - cm.y;
          float e2 = ex * ex + ey * ey;
          double lim = 5.99;
          if (cm.ur >= 0.f) { const float er = w.ur - cm.ur; e2 = e2 + er * er; lim = 7.8; }
          ok = ok && !((double)(e2 * M.inv_sigma2[oct & 15]) > lim);
        }
        // Uniform branch: a divergent region would cost the same issue slots, and the wavefront-wide votes below need all lanes.
        if (__builtin_amdgcn_ballot_w64(ok)) {
          const uint4 a = sDesc[2 * c], b = sDesc[2 * c + 1];
          // one accumulating v_bcnt_u32_b32 per word: two independent chains of four, one add (the compiler's own
          // choice is eight zero-based counts plus an add tree: three more vector instructions per candidate)
          int d0 = popc_acc(a.x ^ qd[0], 0), d1 = popc_acc(b.x ^ qd[4], 0);
          d0 = popc_acc(a.y ^ qd[1], d0); d1 = popc_acc(b.y ^ qd[5], d1);
          d0 = popc_acc(a.z ^ qd[2], d0); d1 = popc_acc(b.z ^ qd[6], d1);
          d0 = popc_acc(a.w ^ qd[3], d0); d1 = popc_acc(b.w ^ qd[7], d1);
          const int dist = d0 + d1;
          const K t = ok ? KT::make(dist, cell_of(cm.bits), base + c) : KT::NONE;
          const bool pass = t < top[MATCH_TOPK - 1];
          const unsigned long long passMask = __builtin_amdgcn_ballot_w64(pass);   // parked insertion, see above
          if (passMask) {
            if (passMask & __builtin_amdgcn_ballot_w64(pend != KT::NONE)) {
              insert(pend);
              pend = KT::NONE;
            }
            pend = pass ? t : pend;
          }
        }
      }
      }
    }
  }
  SSTAMP(st_comp);
#ifdef SCAN_STAMPS
  if (tid == 0 && M.dbg) { long long *d = M.dbg + 4 * (size_t)blockIdx.x; d[0] = st_stage; d[1] = st_comp; d[2] = st_t - st_begin; d[3] = st_begin; }
#endif
  insert(pend);
  if (q < nq) {
    K *o = topk + (size_t)blockIdx.z * slice_stride + (qo + q) * MATCH_TOPK;
#pragma unroll
    for (int j = 0; j < MATCH_TOPK; j++) o[j] = top[j];
    if (STEREO && wantAny) {
      if (gridDim.z == 1) M.qany[qo + q] = any ? 1 : 0;
      else if (any) M.qany[qo + q] = 1;      // sliced: the flags were zeroed before the launch; every writer stores 1
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Window walk: the candidate enumeration of Frame::GetFeaturesInArea (Frame.cc:744-813) itself instead of a scan of every keypoint.
//
// k_match_scan streams ALL keypoints of the frame past every query: right for BASELINE's 1000x1000 stress setting (and for
// relocalisation-style searches) whose windows cover the frame, 100x too much work for a tracking search whose windows
// (radius 15 * scale factor) hold a few dozen keypoints.  k_match_walk stages the frame once per workgroup in the order of
// the reference's mGrid - cells column-major (ix * 48 + iy), i.e. a counting sort by cell built in LDS - with records,
// descriptors and original indices alongside; then every query (one per lane) walks the columns of its cell window: a column's
// cells iy0..iy1 are ONE contiguous range of that order.  The per-candidate tests and the key are k_match_scan's, so the
// top-8 lists - and everything k_match_resolve derives from them - are the same whichever kernel made them (a key is unique
// per keypoint, so the list does not depend on the order candidates are met in).
//
// Which kernel serves a frame pair is decided on the device, per pair, by the same rule in both kernels (and k_topk_merge):
// the walk when no query's cell window exceeds WALK_MAX_CELLS cells and the frame has at most 2048 keypoints (Key32; 50 bytes of
// LDS per keypoint), else the scan.  `force` (orbm_set_scan_mode, the host-pointer entry points) skips the vote.
// ---------------------------------------------------------------------------------------------------------------
#ifdef WALK_STAMPS   // diagnostic builds only (tools/walk_stamps.py): cycles per phase, thread 0 of every workgroup
#define WSTAMP(i) do { const long long t_ = __builtin_readcyclecounter(); if (threadIdx.x == 0 && M.dbg) M.dbg[(size_t)blockIdx.x * 8 + (i)] = t_ - wt0; wt0 = t_; } while (0)
#else
#define WSTAMP(i) do {} while (0)
#endif
#define WALK_LIST 16   // passing candidates a lane collects before it computes their distances
struct WalkRec { float x, y; uint32_t w; };   // w = octave | gx << 4 | gy << 10 | usable << 16 | keypoint index << 17

// LPQ = lanes per query.  1: a batch (one lane walks a query's whole window).  4: few pairs in flight (single-frame calls) - the
// columns of a window are dealt to four adjacent lanes, whose lists lane 0 merges: the walk is a serial chain per lane, and
// a quarter of the columns is a quarter of the chain; four times the workgroups stage the frame, on CUs that are idle anyway.
template <typename KT, int MODE, int LPQ>
__global__ __launch_bounds__(MATCH_NT) void k_match_walk(MatchProblemSet M, typename KT::T *topk, int force, int capn /* LDS rows: >= every n, <= WALK_MAX_N */,
                                                         int wqblocks /* query blocks of MATCH_NT / LPQ queries per pair */) {
  typedef typename KT::T K;
  extern __shared__ __align__(16) uint32_t smem_walk[];
  __shared__ int sVote;
  __shared__ uint32_t sWaveSum[MATCH_NT / 64];
#ifdef WALK_STAMPS
  long long wt0 = __builtin_readcyclecounter();
#endif
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  constexpr int QW = MATCH_NT / LPQ;   // queries per workgroup
  const unsigned b = blockIdx.x, qblocks = (unsigned)wqblocks;   // same XCD-aware order as k_match_scan
  const int qb = (int)((b >> 3) % qblocks), p = (int)((b & 7u) + 8u * (b / (8u * qblocks)));
  if (p >= M.npairs) return;
  const int n = M.frame_n ? M.frame_n[(size_t)p * M.frame_n_stride] : M.frame_n_const;
  const int nq = M.query_n ? M.query_n[(size_t)p * M.query_n_stride] : M.query_n_const;
  if ((int)(qb * QW) >= nq) return;
  if (!pair_walks(M, p, n, nq, force, &sVote)) return;
  WSTAMP(0);
  const size_t fo = (size_t)p * M.frame_stride, qo = (size_t)p * M.query_stride;
  const float *kp = M.kp + fo * 7;
  const uint4 *desc = reinterpret_cast<const uint4 *>(M.desc + fo * 32);
  constexpr bool STEREO = MODE == SCAN_FISHEYE, UR = MODE == SCAN_UR, NEEDUR = MODE == SCAN_UR || MODE == SCAN_FUSE;
  // LDS: cell starts (GRID_CELLS + 1 words, +3 pad); per keypoint IN CELL ORDER a 12-byte record (+ its right coordinate for the
  // modes that test it); per lane a short list of passing candidates.  Descriptors stay in global memory: only the few
  // candidates that pass every test are compared, and 32 more bytes per keypoint would halve the workgroups per CU.
  uint32_t *sCell = smem_walk;
  WalkRec *sRec = reinterpret_cast<WalkRec *>(smem_walk + GRID_CELLS + 4);
  float *sUr = reinterpret_cast<float *>(sRec + capn);
  uint16_t *sList = reinterpret_cast<uint16_t *>(sUr + (NEEDUR ? capn : 0));   // [WALK_LIST][MATCH_NT]
  K *sMerge = reinterpret_cast<K *>(sList + WALK_LIST * MATCH_NT);              // [MATCH_NT][MATCH_TOPK], LPQ > 1 only
  for (int c = tid; c < GRID_CELLS + 4; c += MATCH_NT) sCell[c] = 0u;
  __syncthreads();
  // ---- counting sort by cell: histogram (the atomic's return value is the keypoint's rank inside its cell) ...
  constexpr int PER = WALK_MAX_N / MATCH_NT;   // keypoints per thread
  float kx[PER], ky[PER], kur[PER];
  uint32_t kbits[PER], krank[PER];
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const int i = tid + j * MATCH_NT;
    kbits[j] = 0u; krank[j] = 0u; kx[j] = 0.f; ky[j] = 0.f; kur[j] = -1.f;
    if (i < n) {
      kx[j] = kp[(size_t)i * 7];
      ky[j] = kp[(size_t)i * 7 + 1];
      const int oct = __float_as_int(kp[(size_t)i * 7 + 5]);
      if (NEEDUR) kur[j] = M.u_right ? M.u_right[fo + i] : -1.f;
      const bool claimed = M.slot[fo + i] >= 0 && M.slot_obs[fo + i];
      kbits[j] = cand_bits(kx[j], ky[j], oct, claimed, M);
      if ((kbits[j] >> 25) & 1u) krank[j] = atomicAdd(&sCell[cell_of(kbits[j])], 1u);
    }
  }
  __syncthreads();
  WSTAMP(1);
  // ---- ... exclusive prefix sums over the 3072 cells (12 consecutive cells per thread) ...
  {
    constexpr int CPT = GRID_CELLS / MATCH_NT;
    uint32_t loc[CPT], sum = 0;
#pragma unroll
    for (int j = 0; j < CPT; j++) { loc[j] = sCell[tid * CPT + j]; sum += loc[j]; }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    if (lane == 63) sWaveSum[wid] = inc;
    __syncthreads();
    uint32_t off = inc - sum;
    for (int w2 = 0; w2 < wid; w2++) off += sWaveSum[w2];
#pragma unroll
    for (int j = 0; j < CPT; j++) { sCell[tid * CPT + j] = off; off += loc[j]; }
    if (tid == MATCH_NT - 1) sCell[GRID_CELLS] = off;
  }
  __syncthreads();
  WSTAMP(2);
  // ---- ... scatter the records to start[cell] + rank
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const int i = tid + j * MATCH_NT;
    if (i < n && ((kbits[j] >> 25) & 1u)) {
      const uint32_t pos = sCell[cell_of(kbits[j])] + krank[j];
      const uint32_t wv = (kbits[j] & 0xfu) | (((kbits[j] >> 8) & 0x3fu) << 4) | (((kbits[j] >> 16) & 0x3fu) << 10) | (((kbits[j] >> 24) & 1u) << 16) | ((uint32_t)i << 17);
      sRec[pos] = WalkRec{kx[j], ky[j], wv};
      if (NEEDUR) sUr[pos] = kur[j];
    }
  }
  __syncthreads();
  WSTAMP(3);
  // ---- every query walks its window
  const int q = qb * QW + tid / LPQ, sub = tid % LPQ;
  if (q >= nq) return;
  const QueryWin w = load_query(M, qo, q);
  const int nleft = STEREO && M.qside ? M.nleft : n;
  const bool sideR = STEREO && M.qside && M.qside[qo + q] != 0;
  const bool wantAny = STEREO && M.qany != nullptr;
  bool any = false;
  K top[MATCH_TOPK];
#pragma unroll
  for (int j = 0; j < MATCH_TOPK; j++) top[j] = KT::NONE;
  if (w.live) {
    const uint4 *qp = reinterpret_cast<const uint4 *>(M.qdesc + (qo + q) * 32);
    const uint4 qa = qp[0], qb4 = qp[1];
    uint16_t *myList = sList + tid;
    // One flat loop over the window's candidates (column after column; the cells iy0..iy1 of a column are one range of the cell
    // order): lanes advance independently, so a wavefront runs as long as its busiest lane, not as the sum over columns of the
    // busiest lane per column.  Candidates that pass every test are only noted; their descriptors are fetched and compared
    // WALK_LIST at a time - the Hamming distance and the top-8 insertion run for the few that pass, not once per visited cell entry.
    int ix = w.cx0 + sub, cnt = 0;
    bool more = ix <= w.cx1;      // (a NaN or negative radius gives an empty column range: no candidates, as in the reference's loops)
    int j = more ? (int)sCell[ix * 48 + w.cy0] : 0, j1 = more ? (int)sCell[ix * 48 + w.cy1 + 1] : 0;
    for (;;) {
      if (more) {
        if (j >= j1) {
          if ((ix += LPQ) > w.cx1) more = false;
          else { j = (int)sCell[ix * 48 + w.cy0]; j1 = (int)sCell[ix * 48 + w.cy1 + 1]; }
        } else {
          const WalkRec rec = sRec[j];
          const int oct = rec.w & 0xf, i = (int)(rec.w >> 17);
          // GetFeaturesInArea's own tests (level range, |dx| < r, |dy| < r; the cell window is the loop itself), Frame.cc:794-806
          bool in = fabsf(rec.x - w.u) < w.r && fabsf(rec.y - w.v) < w.r;
          if (w.checkLevels) in = in && oct >= w.minl && (w.maxl < 0 || oct <= w.maxl);
          if (STEREO) in = in && ((i >= nleft) == sideR);
          if (STEREO && wantAny) any = any || in;
          bool ok = in && ((rec.w >> 16) & 1u);                                  // not held by a map point with observations
          if (NEEDUR) {
            const float cur = sUr[j];
            if (UR) ok = ok && !(cur > 0.f && fabsf(w.ur - cur) > w.r);          // ORBmatcher.cc:93-98, :2139-2146
            if (MODE == SCAN_FUSE) {                                             // ORBmatcher.cc:1585-1608
              const float ex = w.u - rec.x, ey = w.v - rec.y;
              float e2 = ex * ex + ey * ey;
              double lim = 5.99;
              if (cur >= 0.f) { const float er = w.ur - cur; e2 = e2 + er * er; lim = 7.8; }
              ok = ok && !((double)(e2 * M.inv_sigma2[oct & 15]) > lim);
            }
          }
          if (ok) { myList[cnt * MATCH_NT] = (uint16_t)j; cnt++; }
          j++;
        }
      }
      // The lists are emptied by the whole wavefront at once - when some lane's list is full, and when every lane is through -
      // so that the descriptor fetches of all lanes overlap (lane-by-lane flushes would each wait out their own memory latency).
      const bool anyMore = __builtin_amdgcn_ballot_w64(more) != 0ull;
      if (__builtin_amdgcn_ballot_w64(cnt == WALK_LIST) != 0ull || !anyMore) {
        for (int e = 0; __builtin_amdgcn_ballot_w64(e < cnt) != 0ull; e++) {
          if (e < cnt) {
            const uint32_t wv = sRec[myList[e * MATCH_NT]].w;
            const int i = (int)(wv >> 17);
            const uint4 a = desc[(size_t)i * 2], b2 = desc[(size_t)i * 2 + 1];
            const int dist = __popc(a.x ^ qa.x) + __popc(a.y ^ qa.y) + __popc(a.z ^ qa.z) + __popc(a.w ^ qa.w) + __popc(b2.x ^ qb4.x) +
                             __popc(b2.y ^ qb4.y) + __popc(b2.z ^ qb4.z) + __popc(b2.w ^ qb4.w);
            K t = KT::make(dist, ((wv >> 4) & 0x3fu) * 48u + ((wv >> 10) & 0x3fu), i);
            if (t < top[MATCH_TOPK - 1]) {
#pragma unroll
              for (int k2 = 0; k2 < MATCH_TOPK; k2++) {
                const K lo = t < top[k2] ? t : top[k2];
                const K hi = t < top[k2] ? top[k2] : t;
                top[k2] = lo;
                t = hi;
              }
            }
          }
        }
        cnt = 0;
      }
      if (!anyMore) break;
    }
  }
  WSTAMP(4);
  if (LPQ > 1) {
    // the LPQ lanes of a query sit side by side in one wavefront and left the loop together: lists through LDS (same-wavefront
    // LDS traffic is ordered), lane 0 of the group folds the others' sorted lists into its own and stops at the first key that
    // cannot enter
#pragma unroll
    for (int j = 0; j < MATCH_TOPK; j++) sMerge[tid * MATCH_TOPK + j] = top[j];
    const unsigned long long anyMask = __builtin_amdgcn_ballot_w64(any);
    any = ((anyMask >> (lane & ~(LPQ - 1))) & ((1ull << LPQ) - 1ull)) != 0ull;
    __builtin_amdgcn_wave_barrier();
    if (sub != 0) return;
    for (int sl = 1; sl < LPQ; sl++)
      for (int j = 0; j < MATCH_TOPK; j++) {
        K t = sMerge[(tid + sl) * MATCH_TOPK + j];
        if (t >= top[MATCH_TOPK - 1]) break;
#pragma unroll
        for (int k2 = 0; k2 < MATCH_TOPK; k2++) {
          const K lo = t < top[k2] ? t : top[k2];
          const K hi = t < top[k2] ? top[k2] : t;
          top[k2] = lo;
          t = hi;
        }
      }
  }
  K *o = topk + (qo + q) * MATCH_TOPK;
#pragma unroll
  for (int j = 0; j < MATCH_TOPK; j++) o[j] = top[j];
  if (STEREO && wantAny) M.qany[qo + q] = any ? 1 : 0;
  WSTAMP(5);
}

// Folds the per-slice sorted top-8 lists of every query into slice 0's list (the S * 8 smallest keys' first 8, sorted).
template <typename KT>
__global__ __launch_bounds__(MATCH_NT) void k_topk_merge(MatchProblemSet M, typename KT::T *topk, size_t slice_stride, int nslices, int force) {
  typedef typename KT::T K;
  __shared__ int sVote;
  const int p = blockIdx.y;
  const int nq = M.query_n ? M.query_n[(size_t)p * M.query_n_stride] : M.query_n_const;
  const int n = M.frame_n ? M.frame_n[(size_t)p * M.frame_n_stride] : M.frame_n_const;
  if ((int)(blockIdx.x * MATCH_NT) >= nq) return;
  if (force != SCAN_DENSE && pair_walks(M, p, n, nq, force, &sVote)) return;   // the walk wrote one list per query: nothing to fold
  const int q = blockIdx.x * MATCH_NT + threadIdx.x;
  if (q >= nq) return;
  K *o = topk + ((size_t)p * M.query_stride + q) * MATCH_TOPK;
  K top[MATCH_TOPK];
#pragma unroll
  for (int j = 0; j < MATCH_TOPK; j++) top[j] = o[j];
  for (int sl = 1; sl < nslices; sl++) {
    const K *in = o + (size_t)sl * slice_stride;
#pragma unroll
    for (int j = 0; j < MATCH_TOPK; j++) {
      K t = in[j];
      if (t >= top[MATCH_TOPK - 1]) break;          // the slice's list is sorted: nothing further can enter
#pragma unroll
      for (int k = 0; k < MATCH_TOPK; k++) {
        const K lo = t < top[k] ? t : top[k];
        const K hi = t < top[k] ? top[k] : t;
        top[k] = lo;
        t = hi;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < MATCH_TOPK; j++) o[j] = top[j];
}

// Accept rule shared by both paths (ORBmatcher.cc:124-130 resp. :2159-2162).
__device__ __forceinline__ bool accept_rule(const MatchProblemSet &M, bool has1, int bd, int lvl1, bool has2, int d2, int lvl2) {
  if (!has1 || bd > M.th_dist) return false;
  if (M.use_second && has2 && lvl1 == lvl2 && (float)bd > M.nnratio * (float)d2) return false;
  return true;
}

// ---- wave-wide minimum through DPP (row_shr 1/2/4/8, row_bcast15, row_bcast31); the result is uniform ---------
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp_or_ones(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, ROWMASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_min_key(uint32_t v) {
  uint32_t t;
  t = dpp_or_ones<0x111, 0xf>(v); v = t < v ? t : v;
  t = dpp_or_ones<0x112, 0xf>(v); v = t < v ? t : v;
  t = dpp_or_ones<0x114, 0xf>(v); v = t < v ? t : v;
  t = dpp_or_ones<0x118, 0xf>(v); v = t < v ? t : v;
  t = dpp_or_ones<0x142, 0xa>(v); v = t < v ? t : v;
  t = dpp_or_ones<0x143, 0xc>(v); v = t < v ? t : v;
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned long long wave_min_key(unsigned long long v) {
#define ORB_DPP64_STEP(CTRL, RM)                                                                           \
  {                                                                                                        \
    const uint32_t lo = dpp_or_ones<CTRL, RM>((uint32_t)v), hi = dpp_or_ones<CTRL, RM>((uint32_t)(v >> 32)); \
    const unsigned long long t = ((unsigned long long)hi << 32) | lo;                                      \
    v = t < v ? t : v;                                                                                     \
  }
  ORB_DPP64_STEP(0x111, 0xf) ORB_DPP64_STEP(0x112, 0xf) ORB_DPP64_STEP(0x114, 0xf) ORB_DPP64_STEP(0x118, 0xf)
  ORB_DPP64_STEP(0x142, 0xa) ORB_DPP64_STEP(0x143, 0xc)
#undef ORB_DPP64_STEP
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}

// Per-lane decision from a query's sorted candidate list (column `lane` of sTk, entry j at sTkLane[64*j]).
//   vm   bit j: entry j exists;  cm  bit j: entry j is claimed by an earlier query;  oct4: 4-bit octave per entry.
// Returns accept<<31 | rescan<<30 | ratioFail<<29 | bestIdx; *bd_out = distance of the best surviving entry (256 if none).
//
// Every candidate that is not in the list has key >= tk[TOPK-1], i.e. distance >= lbDist.  An exact rescan is needed
// only when an unlisted candidate could change the decision:
//   no survivor          and an unlisted one could be within th_dist;
//   one survivor (best)  and the unknown second-best could make the ratio test fail (ORBmatcher.cc:126).
template <typename KT, int STRIDE = 64>
__device__ __forceinline__ uint32_t decide(const MatchProblemSet &M, const typename KT::T *sTkLane, uint32_t vm, uint32_t cm,
                                           uint32_t oct4, bool truncated, int lbDist, int *bd_out) {
  typedef typename KT::T K;
  const uint32_t av = vm & ~cm, av2 = av & (av - 1u);
  const int found = __popc(av);
  const int j1 = (__ffs((int)av) - 1) & (MATCH_TOPK - 1), j2 = (__ffs((int)av2) - 1) & (MATCH_TOPK - 1);
  const K best = sTkLane[STRIDE * j1], second = sTkLane[STRIDE * j2];
  const int l1 = (int)((oct4 >> (4 * j1)) & 0xfu), l2 = (int)((oct4 >> (4 * j2)) & 0xfu);
  const int bd = found > 0 ? KT::dist(best) : 256;
  bool rescan = false;
  if (truncated) {
    if (found == 0) rescan = lbDist <= M.th_dist;
    else if (found == 1 && M.use_second) rescan = bd <= M.th_dist && (float)bd > M.nnratio * (float)lbDist;
  }
  const bool acc = !rescan && accept_rule(M, found > 0, bd, l1, found > 1, KT::dist(second), l2);
  const bool ratioFail = !rescan && !acc && found > 0 && bd <= M.th_dist;  // the `continue` of ORBmatcher.cc:125
  *bd_out = bd;
  return (acc ? 0x80000000u : 0u) | (rescan ? 0x40000000u : 0u) | (ratioFail ? 0x20000000u : 0u) | (found > 0 ? (uint32_t)KT::idx(best) : 0u);
}

// One workgroup per problem.  All wavefronts build the LDS state; wavefront 0 then resolves the queries, 64 at a time.
//
// The reference's loop is sequential only through the claims: the decision of query i is a function f_i of the claims
// made by queries < i.  Within a chunk of 64 queries (one per lane) the decisions D_0..D_63 are therefore the unique
// solution of D_i = f_i(D_0..D_i-1), and that solution is reached by iterating all lanes in parallel:
//   round:  every lane that currently accepts (and whose map point has observations) posts its claim into sOwner with
//           ds_min(lane+1); every lane reads the owner words of its list entries - "claimed for me" means owner <= lane,
//           i.e. committed (0) or posted by an EARLIER lane; the posts are withdrawn; every lane re-decides.
// After round t the first t pending lanes are final, and a round in which no lane of a prefix changed proves that
// prefix final.  Conflicts are sparse, so a prefix settles in two or three rounds instead of one turn per query.
// A lane whose list is exhausted (decide(): an unlisted keypoint could change its decision) cuts the prefix: everything
// before it is committed, then the lane - together with every other exhausted lane, one wavefront each, up to
// NW_ at a time - gets a FRESH list: the REFRESH_K best keypoints of its window that no committed claim holds,
// found by scanning only the grid columns the window touches (sPerm = keypoints sorted by grid column, sCol = column
// starts).  A list is valid as long as it was the head of the unclaimed candidates when it was made - later claims
// are seen through sOwner - so refreshing is always safe, and right after it the lane is first in line and decides.
// tests/resolve_model.py is an executable restatement, tested against the plain in-order loop.
//
// LDS: sOwner[k] = 0 claimed (committed, or held by a map point with observations on entry), 1..64 posted by lane-1
//                  during a round, 0xffffffff free; word n is a dummy that stays free (target of empty list entries).
//      sSlot[k]  = max over accepted queries of (query<<1 | obs), i.e. the LAST query that took keypoint k; -1 none.
//      sTk[j][lane] = entry j of lane's current list.
//      LDSCAND: descriptors + positions staged in LDS once (48 B per keypoint) so that refreshes never leave the CU.
#define RESOLVE_FREE 0xffffffffu
#define REFRESH_K 4
#define REQ_WORDS 16  // lane, flags, u, v, r, ur, minl, maxl, descriptor[8]
// FUSED (Key32, LDSCAND; orb_mfma_util.h): a frame pair ALL of whose live queries are open (pairflag[p], voted by k_match_rank) gets
// no lists from the scan kernels.  It is resolved in the chunked form, and the lists of a chunk of 64 queries are made at the
// chunk's turn by the whole workgroup on the matrix pipe (serve_mfma): every wavefront takes its share of the keypoints, keeps
// only those NO COMMITTED CLAIM HOLDS (compacted, so the work shrinks as the frame fills up: by the last chunks nine keypoints in
// ten are taken), runs them as 32-row A tiles against the chunk's queries (two B tiles built from the posted descriptors) with
// the keypoint's rank as accumulator seed, and keeps a top-4 per query and wavefront; the requesting lane folds the wavefronts' sorted
// shares.  The same pass serves the (now rare) lists exhausted inside a chunk.  The keypoints sit in LDS at position = rank, so
// a key's low 11 bits are the position and sPerm gives the index.  What this replaces: the all-pairs scan kernel (0.23 ms per 256
// frame pairs) and the 36 refresh passes per pair that re-scanned the frame on the vector ALU (70 % of k_match_resolve's 0.29 ms).
template <typename KT, bool LDSCAND, bool FUSED = false>
__global__ __launch_bounds__(64 * RESOLVE_NW_OF(FUSED)) void k_match_resolve(MatchProblemSet M, const typename KT::T *topk, int maxn, int rforce,
                                                                   const uint32_t *rec = nullptr, const uint32_t *keyrec = nullptr, const uint32_t *pairflag = nullptr) {
  constexpr int NW_ = RESOLVE_NW_OF(FUSED);   // wavefronts of this workgroup (see RESOLVE_NW_OF)
  typedef typename KT::T K;
  extern __shared__ __align__(16) uint32_t smem_resolve[];
  __shared__ K sTk[MATCH_TOPK * 64];
  __shared__ int sCol[66 + 66];  // column starts (65 bins + end), then the scatter cursors
  __shared__ int sCmd;           // number of refresh requests posted, -1 = exit
  __shared__ uint32_t sReq[NW_][REQ_WORDS];
  __shared__ K sPart[NW_][NW_][REFRESH_K];   // [request][share]: each serving wavefront's REFRESH_K best keys
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int p = blockIdx.x;
  const int n = M.frame_n ? M.frame_n[(size_t)p * M.frame_n_stride] : M.frame_n_const;
  const int nq = M.query_n ? M.query_n[(size_t)p * M.query_n_stride] : M.query_n_const;
  const size_t fo = (size_t)p * M.frame_stride, qo = (size_t)p * M.query_stride;
  const float *kp = M.kp + fo * 7;
  const uint32_t *desc = reinterpret_cast<const uint32_t *>(M.desc + fo * 32);
  int32_t *slot = M.slot + fo;
  uint8_t *slot_obs = M.slot_obs + fo;
  // carve: [desc 8*maxn words][meta 4*maxn words] (LDSCAND only, both in sPerm order: the refresh scan then reads them
  // sequentially, conflict-free, instead of at random) [owner maxn+1][slot maxn][perm: maxn u16][partner][octave: maxn u8 (LDSCAND only)]
  uint4 *sDesc = reinterpret_cast<uint4 *>(smem_resolve);
  CandMeta *sMeta = reinterpret_cast<CandMeta *>(smem_resolve + (LDSCAND ? 8 * (size_t)maxn : 0));
  uint32_t *sOwner = smem_resolve + (LDSCAND ? 12 * (size_t)maxn : 0);
  int32_t *sSlot = reinterpret_cast<int32_t *>(sOwner + maxn + 1);
  uint16_t *sPerm = reinterpret_cast<uint16_t *>(sSlot + maxn);
  uint16_t *sPartner = sPerm + ((maxn + 1) & ~1);   // stereo partner of each keypoint, 0xffff = none (only if M.partner)
  uint8_t *sOct = reinterpret_cast<uint8_t *>(reinterpret_cast<uint32_t *>(sPartner) + (M.partner ? (maxn + 1) / 2 + 1 : 0));
  uint32_t *sRank = reinterpret_cast<uint32_t *>(sOct + ((maxn + 3) & ~3));   // FUSED only: accumulator seed of the keypoint at sorted position pos
  bool fusedPair = false;
  if constexpr (FUSED) fusedPair = pairflag[p] != 0u;
  int *sFill = sCol + 66;
  for (int i = tid; i < 132; i += 64 * NW_) sCol[i] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += 64 * NW_) {
    sOwner[i] = (slot[i] >= 0 && slot_obs[i]) ? 0u : RESOLVE_FREE;
    sSlot[i] = -1;
    const float x = kp[(size_t)i * 7], y = kp[(size_t)i * 7 + 1];
    const uint32_t bits = cand_bits(x, y, __float_as_int(kp[(size_t)i * 7 + 5]), false, M);
    if (LDSCAND) sOct[i] = (uint8_t)(bits & 0xffu);
    atomicAdd(&sCol[((bits >> 24) & 1u) ? (int)((bits >> 8) & 0xff) + 1 : 65], 1);  // histogram shifted by one
  }
  if (tid == 0) sOwner[n] = RESOLVE_FREE;
  if (M.partner)
    for (int i = tid; i < n; i += 64 * NW_) { const int32_t pr = M.partner[fo + i]; sPartner[i] = (uint16_t)(pr >= 0 && pr < n ? pr : 0xffff); }
  __syncthreads();
  if (tid == 0) {  // exclusive prefix: sCol[c] = first position of column c, sCol[64] = end of the in-grid keypoints
    int acc = 0;
    for (int c = 1; c <= 65; c++) { acc += sCol[c]; sCol[c] = acc; }
  }
  __syncthreads();
  for (int i = tid; i < n; i += 64 * NW_) {
    const float x = kp[(size_t)i * 7], y = kp[(size_t)i * 7 + 1];
    const uint32_t bits = cand_bits(x, y, __float_as_int(kp[(size_t)i * 7 + 5]), false, M);
    const int col = ((bits >> 24) & 1u) ? (int)((bits >> 8) & 0xff) : 64;
    int pos;
    uint32_t seed = 0u;
    if (FUSED && fusedPair) {      // exact (cell, index) order: k_match_rank's rank (column-major cells: the columns stay contiguous)
      seed = rec[fo + i];
      pos = col < 64 ? (int)(seed & 0x7ffu) : sCol[64] + atomicAdd(&sFill[64], 1);
    } else pos = sCol[col] + atomicAdd(&sFill[col], 1);
    sPerm[pos] = (uint16_t)i;
    if (LDSCAND) {   // record and descriptor of keypoint i live at its sorted position
      CandMeta c;
      c.x = x; c.y = y; c.bits = bits;
      c.ur = M.u_right ? M.u_right[fo + i] : -1.f;
      sMeta[pos] = c;
      const uint4 *gd = reinterpret_cast<const uint4 *>(M.desc + fo * 32);
      sDesc[2 * pos] = gd[2 * i];
      sDesc[2 * pos + 1] = gd[2 * i + 1];
    }
    if (FUSED && fusedPair) sRank[pos] = seed;
  }
  __syncthreads();
  auto octave_of = [&](int idx) -> int {
    if (LDSCAND) return (int)sOct[idx];
    return __float_as_int(kp[(size_t)idx * 7 + 5]) & 0xff;
  };
  // One wavefront serves refresh request `rq`: the REFRESH_K smallest keys of the query's window among the keypoints
  // no committed claim holds -> column `target lane` of sTk.
  // A pass serves m <= NW_ requests with all NW_ wavefronts: mp = m rounded up to a power of two, every request
  // gets NW_ / mp wavefronts, each scanning an interleaved share of the window's keypoints; the requesting lane merges
  // the shares' sorted lists.  Most passes carry one or two requests, which then finish 8 or 4 times sooner.
  auto shares_of = [](int m) { return m <= 1 ? NW_ : m <= 2 ? NW_ / 2 : m <= 4 ? NW_ / 4 : 1; };
  auto serve = [&](int rq, int part, int nparts) {
    const uint32_t *R = sReq[rq];
    const int target = (int)R[0];
    QueryWin w;
    w.u = __uint_as_float(R[2]); w.v = __uint_as_float(R[3]); w.r = __uint_as_float(R[4]); w.ur = __uint_as_float(R[5]);
    w.minl = (int)R[6]; w.maxl = (int)R[7];
    w.cx0 = max(0, (int)floorf((w.u - M.min_x - w.r) * M.inv_w));   // Frame::GetFeaturesInArea, Frame.cc:755-777
    w.cx1 = min(63, (int)ceilf((w.u - M.min_x + w.r) * M.inv_w));
    w.cy0 = max(0, (int)floorf((w.v - M.min_y - w.r) * M.inv_h));
    w.cy1 = min(47, (int)ceilf((w.v - M.min_y + w.r) * M.inv_h));
    w.live = (R[1] & 1u) && w.cx0 < 64 && w.cx1 >= 0 && w.cy0 < 48 && w.cy1 >= 0 && w.cx0 <= w.cx1;
    const int nleft = M.qside ? M.nleft : n;
    const bool sideR = (R[1] >> 8) & 1u;             // fisheye-stereo: the query sees only the keypoints of its own image
    w.checkLevels = (w.minl > 0) || (w.maxl >= 0);
    w.stereo = M.u_right != nullptr;
    uint32_t q8[8];
#pragma unroll
    for (int t = 0; t < 8; t++) q8[t] = R[8 + t];
    K l[REFRESH_K];
#pragma unroll
    for (int j = 0; j < REFRESH_K; j++) l[j] = KT::NONE;
    // level filter as a closed interval (open ends when disabled), window test in sign-bit arithmetic as in k_match_scan
    const int minlE = w.checkLevels ? w.minl : -1000, maxlE = w.checkLevels && w.maxl >= 0 ? w.maxl : 1000;
    // Two steps per keypoint: its 16-byte record (position, cell, octave, usable bit) and owner word decide whether it is a
    // candidate at all; only then are the 32 descriptor bytes read.  Late in a frame most keypoints are claimed, and the scan
    // is bound by LDS throughput (random 16-byte reads), so not touching their descriptors is most of the saving.
    struct Probe { float x, y, ur; uint32_t bits; int c, pos; };
    auto probe = [&](int pos, bool valid) {
      Probe k;
      k.c = valid ? (int)sPerm[pos] : -1;
      k.pos = valid ? pos : 0;
      const int c = valid ? k.c : 0;
      const bool cl = sOwner[valid ? c : n] == 0u;
      if (LDSCAND) {
        const CandMeta cmeta = sMeta[k.pos];
        k.x = cmeta.x; k.y = cmeta.y; k.ur = cmeta.ur;
        k.bits = cl ? (cmeta.bits & ~(1u << 24)) : cmeta.bits;
      } else {
        k.x = kp[(size_t)c * 7]; k.y = kp[(size_t)c * 7 + 1];
        k.bits = cand_bits(k.x, k.y, __float_as_int(kp[(size_t)c * 7 + 5]), cl, M);
        k.ur = M.u_right ? M.u_right[fo + c] : -1.f;
      }
      return k;
    };
    auto passes = [&](const Probe &k) {
      const int gx = (k.bits >> 8) & 0xff, gy = (k.bits >> 16) & 0xff, oct = k.bits & 0xff;
      int viol = (gx - w.cx0) | (w.cx1 - gx) | (gy - w.cy0) | (w.cy1 - gy) | (oct - minlE) | (maxlE - oct) | k.c;
      viol |= ((k.c >= nleft) == sideR) ? 0 : -1;
      viol |= ~(int)(k.bits << 7);                                    // bit 24 = usable (in grid, not claimed)
      const int fpass = __float_as_int(fabsf(k.x - w.u) - w.r) & __float_as_int(fabsf(k.y - w.v) - w.r);
      bool ok = (fpass & ~viol) < 0;
      if (w.stereo) ok = ok && !(k.ur > 0.f && fabsf(w.ur - k.ur) > w.r);   // ORBmatcher.cc:93-98, :2139-2146
      return ok;
    };
    auto score = [&](const Probe &k) {
      uint32_t d[8];
      if (LDSCAND) {
        const uint4 a = sDesc[2 * k.pos], b = sDesc[2 * k.pos + 1];
        d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
      } else {
#pragma unroll
        for (int t = 0; t < 8; t++) d[t] = desc[(size_t)k.c * 8 + t];
      }
      int dist = 0;
#pragma unroll
      for (int t = 0; t < 8; t++) dist += __popc(d[t] ^ q8[t]);
      K t = KT::make(dist, cell_of(k.bits), k.c);
      if (t < l[REFRESH_K - 1]) {
#pragma unroll
        for (int j = 0; j < REFRESH_K; j++) {
          const K lo = t < l[j] ? t : l[j];
          const K hi = t < l[j] ? l[j] : t;
          l[j] = lo;
          t = hi;
        }
      }
    };
    const int end = w.live ? sCol[w.cx1 + 1] : 0;
    for (int pos = (w.live ? sCol[w.cx0] : 0) + lane + 128 * part; pos < end; pos += 128 * nparts) {  // two keypoints in flight per lane
      const Probe k0 = probe(pos, true);
      const Probe k1 = probe(pos + 64, pos + 64 < end);
      const bool ok0 = passes(k0), ok1 = passes(k1);
      if (ok0) score(k0);
      if (ok1) score(k1);
    }
    // REFRESH_K extractions of the wave minimum; keys are unique, so exactly one lane pops per extraction
    K out[REFRESH_K];
#pragma unroll
    for (int j = 0; j < REFRESH_K; j++) {
      out[j] = wave_min_key(l[0]);
      if (l[0] == out[j] && out[j] != KT::NONE) {
#pragma unroll
        for (int t = 0; t + 1 < REFRESH_K; t++) l[t] = l[t + 1];
        l[REFRESH_K - 1] = KT::NONE;
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < REFRESH_K; j++) sPart[rq][part][j] = out[j];
    }
    (void)target;
  };
  // ---- Wide form (tracking-sized windows, monocular / rectified-stereo problems) ------------------------------------------------------
  // The chunked form below resolves 64 queries at a time on ONE wavefront: with sparse conflicts its time is the number of
  // chunks times (a dozen global loads + two or three rounds of LDS round trips), all of it latency on a single wavefront while
  // seven wait.  The same fix-point holds for any number of lanes: here every thread of the workgroup is a lane (64 * NW_ queries per
  // super-chunk), "claimed for me" is still owner <= my index, a round is separated by workgroup barriers instead of wavefront
  // ones, and the settled prefix / first exhausted lane are found through two LDS words.  Conflict chains are short, so a
  // super-chunk settles in three or four rounds - a 1000-query frame in 8 rounds instead of 48.  Exhausted lists are refreshed
  // by the whole workgroup exactly as below (serve()).  Used when no query's window exceeds WALK_MAX_CELLS grid cells (the rule
  // of k_match_walk: few candidates per query, few exhausted lists); the stress setting keeps the chunked form, whose refresh
  // passes are cheaper per pass.
  __shared__ int sVoteR;
  constexpr bool WIDE_OK = sizeof(K) == 4;   // Key32 only: frames of at most 2048 keypoints (the wide list array is 16 KiB)
  constexpr int WN = 64 * NW_;
  constexpr int TKW_WORDS = FUSED && MATCH_TOPK * WN < 3072 ? 3072 : MATCH_TOPK * WN;   // the fused form's work areas live in it as well
  __shared__ __align__(16) K sTkW[WIDE_OK ? TKW_WORDS : 1];
  __shared__ int sRmin[4], sChg[4], sTake, sNm;   // sRmin / sChg: a ring over the rounds (see below)
  bool wide = false;
  if (FUSED && fusedPair) wide = false;   // fused pairs take the chunked form below
  else if (WIDE_OK && !M.serial && !M.partner && M.couple == 0 && !M.qside && rforce != SCAN_DENSE) {
    wide = pair_walks<WN>(M, p, n, nq, rforce, &sVoteR);   // k_match_walk's rule, voted by this kernel's threads
  }
  if (wide) {
    if (tid == 0) { for (int i = 0; i < 4; i++) { sRmin[i] = 0x7fffffff; sChg[i] = 0; } sTake = 0; sNm = 0; }
    __syncthreads();
    int nmatches = 0;
#ifdef WIDE_STAMPS   // diagnostic builds only (tools/wide_stamps.py): cycles per phase of the wide form, thread 0 of the workgroup
    long long ws_build = 0, ws_round = 0, ws_refresh = 0, ws_nround = 0, ws_nrefresh = 0, ws_ncut = 0, ws_t = __builtin_readcyclecounter();
    const long long ws_begin = ws_t;
#define WSTMP(acc) do { const long long t_ = __builtin_readcyclecounter(); acc += t_ - ws_t; ws_t = t_; } while (0)
#else
#define WSTMP(acc) do {} while (0)
#endif
    for (int base = 0; base < nq; base += WN) {
      const int q = base + tid;
      const int cnt = min(WN, nq - base);
      const uint32_t myfl = q < nq ? (M.qflags ? M.qflags[qo + q] : 3u) : 0u;
      const bool ob = (myfl >> 1) & 1u;
      uint32_t qpar[6] = {0, 0, 0, 0, 0, 0}, qd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < MATCH_TOPK; j++) sTkW[WN * j + tid] = q < nq ? topk[(qo + q) * MATCH_TOPK + j] : KT::NONE;
      if (q < nq) {
        qpar[0] = __float_as_uint(M.qu[qo + q]); qpar[1] = __float_as_uint(M.qv[qo + q]); qpar[2] = __float_as_uint(M.qr[qo + q]);
        qpar[3] = __float_as_uint(M.qur ? M.qur[qo + q] : 0.f);
        qpar[4] = (uint32_t)M.qminl[qo + q]; qpar[5] = (uint32_t)M.qmaxl[qo + q];
        const uint4 *qp = reinterpret_cast<const uint4 *>(M.qdesc + (qo + q) * 32);
        const uint4 a = qp[0], b = qp[1];
        qd[0] = a.x; qd[1] = a.y; qd[2] = a.z; qd[3] = a.w; qd[4] = b.x; qd[5] = b.y; qd[6] = b.z; qd[7] = b.w;
      }
      int eidx[MATCH_TOPK];
      uint32_t vm, oct4;
      int capm1 = MATCH_TOPK - 1, lbDist;
      bool truncated;
      auto load_list = [&]() {   // own column only: no barrier needed between its write and this read
        vm = 0; oct4 = 0;
#pragma unroll
        for (int j = 0; j < MATCH_TOPK; j++) {
          const K t = sTkW[WN * j + tid];
          eidx[j] = n;
          if (t != KT::NONE) {
            const int idx = KT::idx(t);
            eidx[j] = idx;
            vm |= 1u << j;
            oct4 |= (uint32_t)(octave_of(idx) & 0xf) << (4 * j);
          }
        }
        const K last = sTkW[WN * capm1 + tid];
        truncated = last != KT::NONE;
        lbDist = KT::dist(last);
        if (!(myfl & 1u)) { vm = 0; truncated = false; }
      };
      load_list();
      // fresh lists for every lane with `want`, NW_ requests per pass, all wavefronts serving (serve() as in the chunked form)
      auto refresh = [&](bool want) {
        bool todo = want;
        WSTMP(ws_round);
        for (;;) {
          __syncthreads();                       // sTake == 0 here (reset at the end of the previous pass / at start)
          int slot = -1;
          if (todo) slot = atomicAdd(&sTake, 1);
          const bool take = todo && slot < NW_;
          if (take) {
            uint32_t *R = sReq[slot];
            R[0] = (uint32_t)tid; R[1] = myfl;
#pragma unroll
            for (int t = 0; t < 6; t++) R[2 + t] = qpar[t];
#pragma unroll
            for (int t = 0; t < 8; t++) R[8 + t] = qd[t];
          }
          __syncthreads();
          const int asked = sTake;
          if (asked == 0) break;                 // uniform: nobody left
          const int m = min(NW_, asked);
          {
            const int np = shares_of(m), mp = NW_ / np, rq = wid & (mp - 1);
            if (rq < m) serve(rq, wid / mp, np);
          }
          __syncthreads();
          if (tid == 0) sTake = 0;
          if (take) {
            static_assert(REFRESH_K == 4, "the merge network below is written for 4 keys");
            auto cx = [](K &lo, K &hi) { const K l = lo < hi ? lo : hi, h = lo < hi ? hi : lo; lo = l; hi = h; };
            const int np = shares_of(m);
            K a[REFRESH_K];
#pragma unroll
            for (int j = 0; j < REFRESH_K; j++) a[j] = sPart[slot][0][j];
            for (int sh = 1; sh < np; sh++) {
#pragma unroll
              for (int j = 0; j < REFRESH_K; j++) {
                const K y = sPart[slot][sh][REFRESH_K - 1 - j];
                a[j] = a[j] < y ? a[j] : y;
              }
              cx(a[0], a[2]); cx(a[1], a[3]);
              cx(a[0], a[1]); cx(a[2], a[3]);
            }
#pragma unroll
            for (int j = 0; j < MATCH_TOPK; j++) sTkW[WN * j + tid] = j < REFRESH_K ? a[j] : KT::NONE;
            todo = false;
            capm1 = REFRESH_K - 1;
            load_list();
          }
#ifdef WIDE_STAMPS
          ws_nrefresh++;
#endif
        }
        WSTMP(ws_refresh);
      };
      uint32_t D = 0;
      int my_bd = 256, res_idx = -1, res_bd = 256;
      int s = 0, par = 0;
      bool first_round = true;
      WSTMP(ws_build);
      while (s < cnt) {
#ifdef WIDE_STAMPS
        ws_ncut++;
#endif
        int r;
        for (;;) {
          const bool pend = tid >= s && tid < cnt;
          const bool post = pend && (D >> 31) && ob;
          const int bidx = (int)(D & 0xfffffu);
          if (post) __hip_atomic_fetch_min(&sOwner[bidx], (uint32_t)(tid + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __syncthreads();
          uint32_t cm = 0;
#pragma unroll
          for (int j = 0; j < MATCH_TOPK; j++) cm |= (sOwner[eidx[j]] <= (uint32_t)tid ? 1u : 0u) << j;
          const uint32_t mine = sOwner[post ? bidx : n];
          __syncthreads();
          if (post && mine != 0u) sOwner[bidx] = RESOLVE_FREE;     // withdraw (a claim committed meanwhile, 0, must survive)
          const uint32_t nD = pend ? decide<KT, WN>(M, sTkW + tid, vm, cm, oct4, truncated, lbDist, &my_bd) : D;
          const bool changed = nD != D;
          D = nD;
          const bool flagged = pend && ((D >> 30) & 1u);
          // first exhausted lane, then "did anything at or before it change": two LDS words per round, taken from a ring of four -
          // a round's words are cleared two rounds later, when every thread is past the barriers behind which it read them
          const int cur = par & 3;
          par++;
#ifdef WIDE_STAMPS
          ws_nround++;
#endif
          if (tid == 0) { sRmin[(cur + 2) & 3] = 0x7fffffff; sChg[(cur + 2) & 3] = 0; }
          if (flagged) atomicMin(&sRmin[cur], tid);
          __syncthreads();
          r = min(sRmin[cur], cnt);
          if (first_round) {
            first_round = false;
            if (r < cnt) {             // exhausted by the claims of earlier super-chunks: refresh them all at once, decide again
              refresh(flagged);
              continue;
            }
          }
          if (changed && tid <= r) sChg[cur] = 1;
          __syncthreads();
          if (sChg[cur] == 0) break;
        }
        // commit the settled prefix [s, r)
        {
          const bool inpre = tid >= s && tid < r;
          const bool acc = inpre && (D >> 31);
          const int bidx = (int)(D & 0xfffffu);
          nmatches += __popcll(__ballot(acc));
          if (acc) {
            const int32_t sv = (int32_t)(((uint32_t)q << 1) | (ob ? 1u : 0u));
            if (ob) sOwner[bidx] = 0u;
            __hip_atomic_fetch_max(&sSlot[bidx], sv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          if (inpre) { res_idx = acc ? bidx : -1; res_bd = my_bd <= M.th_dist ? my_bd : 256; }
        }
        s = r;
        if (r < cnt) refresh(tid >= r && tid < cnt && ((D >> 30) & 1u));   // lane r is first in line now: with its fresh list it decides next round
        else __syncthreads();                                              // commits visible before the next super-chunk's first round
      }
      if (q < nq) {
        if (M.match_of_query) M.match_of_query[qo + q] = res_idx;
        if (M.best_dist) M.best_dist[qo + q] = res_bd;
      }
    }
    WSTMP(ws_round);
#ifdef WIDE_STAMPS
    if (tid == 0 && M.dbg) { long long *d = M.dbg + 16 * (size_t)p; d[0] = ws_build; d[1] = ws_round; d[2] = ws_refresh; d[3] = ws_nround; d[4] = ws_nrefresh; d[5] = ws_ncut; d[6] = ws_t - ws_begin; d[7] = nq; }
#endif
    if (lane == 0 && nmatches) atomicAdd(&sNm, nmatches);
    __syncthreads();
    if (tid == 0 && M.nmatches) M.nmatches[p] = sNm;
    for (int i = tid; i < n; i += 64 * NW_) {
      const int32_t v = sSlot[i];
      if (v >= 0) { slot[i] = v >> 1; slot_obs[i] = (uint8_t)(v & 1); }
    }
    return;
  }
  // ---- FUSED: lists on the matrix pipe (see the comment above the kernel).  Work areas in the wide form's list array, which a fused
  // pair does not use: [request descriptors 64 x 8 words][shares 64 x NW_ wavefronts x 4 keys][compacted positions NW_ x CMP_CAP u16][seeds NW_ x 32]
  constexpr int CMP_CAP = 2048 / NW_;                  // a wavefront's share of at most 2048 keypoints
  static_assert(!FUSED || (64 * 8 + 64 * NW_ * REFRESH_K + NW_ * CMP_CAP / 2 + NW_ * MF_TILE) * 4 <= TKW_WORDS * (int)sizeof(K), "work areas exceed the list array");
  uint32_t *sReqD = reinterpret_cast<uint32_t *>(sTkW);
  uint32_t *sPartF = sReqD + 64 * 8;
  uint16_t *sCmpF = reinterpret_cast<uint16_t *>(sPartF + 64 * NW_ * REFRESH_K);
  uint32_t *sSeedF = reinterpret_cast<uint32_t *>(sCmpF + NW_ * CMP_CAP);
  auto serve_mfma = [&](int m) {   // all wavefronts; m <= 64 requests posted in sReqD (their queries are open: every usable keypoint is a candidate)
    if constexpr (FUSED) {
      const int col = lane & 31, h = lane >> 5;
      const int nin = sCol[64];                               // the keypoints PosInGrid accepts are positions [0, nin)
      const int lo = (int)(((long long)nin * wid) / NW_), hi = (int)(((long long)nin * (wid + 1)) / NW_);
      uint16_t *cmp = sCmpF + wid * CMP_CAP;
      int kw = 0;
      for (int b0 = lo; b0 < hi; b0 += 64) {                  // my share, compacted to the keypoints no committed claim holds
        const int pos = b0 + lane;
        bool keep = false;
        if (pos < hi) keep = !(sRank[pos] & MF_REC_HELD) && sOwner[sPerm[pos]] != 0u;
        const unsigned long long mk = __ballot(keep);
        if (keep) cmp[kw + __popcll(mk & ((1ull << lane) - 1ull))] = (uint16_t)pos;
        kw += __popcll(mk);
      }
      const bool two = m > 32;
      mf_v4i B0[8], B1[8];
      {
        const uint4 *r0 = reinterpret_cast<const uint4 *>(sReqD + col * 8), *r1 = reinterpret_cast<const uint4 *>(sReqD + (32 + col) * 8);
        const uint4 a0 = r0[0], b0 = r0[1], a1 = r1[0], b1 = r1[1];
        const uint32_t d0[8] = {a0.x, a0.y, a0.z, a0.w, b0.x, b0.y, b0.z, b0.w}, d1[8] = {a1.x, a1.y, a1.z, a1.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int s = 0; s < 8; s++) {
          B0[s] = mf_expand16((d0[s] >> (16 * h)) & 0xffffu, MF_LUT_QUERY);
          B1[s] = mf_expand16((d1[s] >> (16 * h)) & 0xffffu, MF_LUT_QUERY);
        }
      }
      MfList<REFRESH_K> L0, L1;
      L0.init(); L1.init();
      uint32_t *seedw = sSeedF + wid * MF_TILE;
      for (int t0 = 0; t0 < kw; t0 += MF_TILE) {
        const int e = t0 + col;
        const bool have = e < kw;
        const int pos = have ? (int)cmp[e] : 0;
        if (h == 0) seedw[col] = have ? sRank[pos] : MF_REC_UNUSABLE;
        // A fragments straight from the keypoint's descriptor: bits [32 s + 16 h, 32 s + 16 h + 16) of row `col` per K-step s
        const uint4 da = sDesc[2 * pos], db = sDesc[2 * pos + 1];
        const uint32_t dd[8] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
        mf_v4i A[8];
#pragma unroll
        for (int s = 0; s < 8; s++) A[s] = mf_expand16((dd[s] >> (16 * h)) & 0xffffu, MF_LUT_CAND);
        mf_v16i c;
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const mf_v4i v4 = *reinterpret_cast<const mf_v4i *>(&seedw[8 * g + 4 * h]);
          c[4 * g] = v4[0]; c[4 * g + 1] = v4[1]; c[4 * g + 2] = v4[2]; c[4 * g + 3] = v4[3];
        }
        mf_v16i acc0 = c;
#pragma unroll
        for (int s = 0; s < 8; s++) acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[s], B0[s], acc0, 0, 0, 0);
        L0.take(acc0);
        if (two) {
          mf_v16i acc1 = c;
#pragma unroll
          for (int s = 0; s < 8; s++) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[s], B1[s], acc1, 0, 0, 0);
          L1.take(acc1);
        }
      }
      // lane h of a column reports request h * 32 + col: it sends the partner the other tile's list and merges what it receives
      // (two sorted 4-lists: min(a[i], b[3 - i]) is bitonic, two compare-exchange stages sort it)
      static_assert(REFRESH_K == 4, "the merge network below is written for 4 keys");
      uint32_t mg[REFRESH_K];
#pragma unroll
      for (int j = 0; j < REFRESH_K; j++) {
        const uint32_t send = h ? L0.top[REFRESH_K - 1 - j] : L1.top[REFRESH_K - 1 - j];
        const uint32_t other = (uint32_t)__shfl_xor((int)send, 32, 64);
        mg[j] = min(h ? L1.top[j] : L0.top[j], other);
      }
      auto cxu = [](uint32_t &a, uint32_t &b) { const uint32_t l = min(a, b), g2 = max(a, b); a = l; b = g2; };
      cxu(mg[0], mg[2]); cxu(mg[1], mg[3]);
      cxu(mg[0], mg[1]); cxu(mg[2], mg[3]);
      const int req = h * 32 + col;
      if (req < m) *reinterpret_cast<uint4 *>(sPartF + ((size_t)req * NW_ + wid) * REFRESH_K) = make_uint4(mg[0], mg[1], mg[2], mg[3]);
    }
  };
  // requesting lane: fold the NW_ sorted shares of request `rank`, turn the keys (distance << 11 | position) into list entries
  auto fold_mfma = [&](int rank, K *colp /* column of the chunk's list array, stride 64 */) {
    if constexpr (FUSED) {
      auto cxu = [](uint32_t &a, uint32_t &b) { const uint32_t l = min(a, b), g2 = max(a, b); a = l; b = g2; };
      const uint4 *sh = reinterpret_cast<const uint4 *>(sPartF + (size_t)rank * NW_ * REFRESH_K);
      uint4 v = sh[0];
      uint32_t a[REFRESH_K] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int w2 = 1; w2 < NW_; w2++) {
        v = sh[w2];
        a[0] = min(a[0], v.w); a[1] = min(a[1], v.z); a[2] = min(a[2], v.y); a[3] = min(a[3], v.x);
        cxu(a[0], a[2]); cxu(a[1], a[3]);
        cxu(a[0], a[1]); cxu(a[2], a[3]);
      }
#pragma unroll
      for (int j = 0; j < MATCH_TOPK; j++) {
        K o = KT::NONE;
        if (j < REFRESH_K && a[j] < MF_KEY_LIMIT) o = (K)(((a[j] >> 11) << 23) | (uint32_t)sPerm[a[j] & 0x7ffu]);
        colp[64 * j] = o;
      }
    }
  };
#ifdef RESOLVE_STAMPS
  long long t_round = 0, t_refresh = 0, n_refresh = 0, n_batch = 0, t_chunk = 0, n_round = 0, t_sub[6] = {0, 0, 0, 0, 0, 0}, sub_m = 0;
  long long t0 = __builtin_readcyclecounter();
#endif
  if (wid != 0) {
    // helper waves: sleep at the barrier until wave 0 posts refresh requests (or the exit command)
    for (;;) {
      __syncthreads();                 // (A) requests posted
      const int m = sCmd;
      if (m < 0) break;
      if (FUSED && fusedPair) serve_mfma(m);
      else {
        const int np = shares_of(m), mp = NW_ / np, rq = wid & (mp - 1);
        if (rq < m) serve(rq, wid / mp, np);
      }
      __syncthreads();                 // (B) shares written
    }
  } else {
    int nmatches = 0;
    for (int base = 0; base < nq; base += 64) {
#ifdef RESOLVE_STAMPS
      long long tc0 = __builtin_readcyclecounter();
#endif
      const int q = base + lane;
      const int cnt = min(64, nq - base);
      // my query: parameters and descriptor stay in registers (they become a refresh request if my list runs out)
      uint32_t myfl = q < nq ? (M.qflags ? M.qflags[qo + q] : 3u) : 0u;
      if (M.qside && q < nq && M.qside[qo + q]) myfl |= 0x100u;                       // bit 8: right-image query
      if (M.couple == 2 && (q & 1) && q < nq && !M.qany[qo + q - 1]) myfl &= ~1u;     // :2126, left window empty
      const bool ob = (myfl >> 1) & 1u;
      uint32_t qpar[6] = {0, 0, 0, 0, 0, 0}, qd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < MATCH_TOPK; j++) sTk[64 * j + lane] = (q < nq && !(FUSED && fusedPair)) ? topk[(qo + q) * MATCH_TOPK + j] : KT::NONE;
      if (q < nq) {
        qpar[0] = __float_as_uint(M.qu[qo + q]); qpar[1] = __float_as_uint(M.qv[qo + q]); qpar[2] = __float_as_uint(M.qr[qo + q]);
        qpar[3] = __float_as_uint(M.qur ? M.qur[qo + q] : 0.f);
        qpar[4] = (uint32_t)M.qminl[qo + q]; qpar[5] = (uint32_t)M.qmaxl[qo + q];
        const uint4 *qp = reinterpret_cast<const uint4 *>(M.qdesc + (qo + q) * 32);
        const uint4 a = qp[0], b = qp[1];
        qd[0] = a.x; qd[1] = a.y; qd[2] = a.z; qd[3] = a.w; qd[4] = b.x; qd[5] = b.y; qd[6] = b.z; qd[7] = b.w;
        // FUSED: no scan kernel has looked at this query's window.  A query in view whose window lies outside the grid
        // (GetFeaturesInArea's early returns, Frame.cc:757-777) has no candidates: it is not a request (the pair's vote counts only
        // queries with a window INSIDE the grid as live, so such a query does not keep the pair from being fused)
        if (FUSED && fusedPair && !load_query(M, qo, q).live) myfl &= ~1u;
      }
      // my list as the rounds need it: keypoint index per entry (n = the dummy for an empty entry), validity mask,
      // octaves, and the lower bound every unlisted candidate obeys (entry capm1, the last one the list can hold)
      int eidx[MATCH_TOPK];
      uint32_t vm, oct4;
      int capm1 = MATCH_TOPK - 1, lbDist;
      bool truncated;
      auto load_list = [&]() {
        vm = 0; oct4 = 0;
#pragma unroll
        for (int j = 0; j < MATCH_TOPK; j++) {
          const K t = sTk[64 * j + lane];
          eidx[j] = n;
          if (t != KT::NONE) {
            const int idx = KT::idx(t);
            eidx[j] = idx;
            vm |= 1u << j;
            oct4 |= (uint32_t)(octave_of(idx) & 0xf) << (4 * j);
          }
        }
        const K last = sTk[64 * capm1 + lane];
        truncated = last != KT::NONE;
        lbDist = KT::dist(last);
        if (!(myfl & 1u)) { vm = 0; truncated = false; }   // dropped query (scan may have listed candidates for it)
      };
      load_list();
      // refresh the lists of the lanes in F, NW_ per pass (one wavefront each)
      auto refresh = [&](unsigned long long F) {
#ifdef RESOLVE_STAMPS
        long long ts0 = __builtin_readcyclecounter(), ts1 = ts0;
#endif
        const bool mineF = (F >> lane) & 1ull;
        unsigned long long todo = F;
        if (FUSED && fusedPair && todo) {    // one pass of the whole workgroup on the matrix pipe serves every request
          const int rank = __popcll(todo & ((1ull << lane) - 1ull));
          if (mineF) {
            uint4 *R = reinterpret_cast<uint4 *>(sReqD + rank * 8);
            R[0] = make_uint4(qd[0], qd[1], qd[2], qd[3]);
            R[1] = make_uint4(qd[4], qd[5], qd[6], qd[7]);
          }
          const int m = (int)__popcll(todo);
          if (lane == 0) sCmd = m;
          __syncthreads();             // (A)
          serve_mfma(m);
          __syncthreads();             // (B)
          if (mineF) fold_mfma(rank, sTk + lane);
          todo = 0ull;
#ifdef RESOLVE_STAMPS
          n_batch++;
#endif
        }
        while (todo) {
          const int rank = __popcll(todo & ((1ull << lane) - 1ull));
          const bool take = ((todo >> lane) & 1ull) && rank < NW_;
          if (take) {
            uint32_t *R = sReq[rank];
            R[0] = (uint32_t)lane; R[1] = myfl;
#pragma unroll
            for (int t = 0; t < 6; t++) R[2 + t] = qpar[t];
#pragma unroll
            for (int t = 0; t < 8; t++) R[8 + t] = qd[t];
          }
          const int m = min(NW_, (int)__popcll(todo));
          if (lane == 0) sCmd = m;
#ifdef RESOLVE_STAMPS
          long long tq = __builtin_readcyclecounter(); t_sub[0] += tq - ts1; ts1 = tq; sub_m += m;
#endif
          __syncthreads();             // (A)
#ifdef RESOLVE_STAMPS
          tq = __builtin_readcyclecounter(); t_sub[1] += tq - ts1; ts1 = tq;
#endif
          const int np = shares_of(m);
          serve(0, 0, np);
#ifdef RESOLVE_STAMPS
          tq = __builtin_readcyclecounter(); t_sub[2] += tq - ts1; ts1 = tq;
#endif
          __syncthreads();             // (B)
#ifdef RESOLVE_STAMPS
          tq = __builtin_readcyclecounter(); t_sub[3] += tq - ts1; ts1 = tq;
#endif
          if (take) {
            // merge the np sorted shares of my request: min(A[i], B[K-1-i]) are the K smallest of a pair's union (a bitonic
            // sequence), two compare-exchange stages sort them; the rest of my column is empty
            static_assert(REFRESH_K == 4, "the merge network below is written for 4 keys");
            auto cx = [](K &lo, K &hi) { const K l = lo < hi ? lo : hi, h = lo < hi ? hi : lo; lo = l; hi = h; };
            K a[REFRESH_K];
#pragma unroll
            for (int j = 0; j < REFRESH_K; j++) a[j] = sPart[rank][0][j];
            for (int sh = 1; sh < np; sh++) {
#pragma unroll
              for (int j = 0; j < REFRESH_K; j++) {
                const K y = sPart[rank][sh][REFRESH_K - 1 - j];
                a[j] = a[j] < y ? a[j] : y;
              }
              cx(a[0], a[2]); cx(a[1], a[3]);
              cx(a[0], a[1]); cx(a[2], a[3]);
            }
#pragma unroll
            for (int j = 0; j < MATCH_TOPK; j++) sTk[64 * j + lane] = j < REFRESH_K ? a[j] : KT::NONE;
          }
          todo &= ~__ballot(take);
#ifdef RESOLVE_STAMPS
          n_batch++;
          tq = __builtin_readcyclecounter(); t_sub[4] += tq - ts1; ts1 = tq;
#endif
        }
        if (mineF) capm1 = REFRESH_K - 1;
        load_list();
#ifdef RESOLVE_STAMPS
        t_sub[5] += __builtin_readcyclecounter() - ts1;
        t_refresh += __builtin_readcyclecounter() - ts0; n_refresh += __popcll(F);
#endif
      };
      if (FUSED && fusedPair) {              // the chunk's lists, made now: only keypoints without a committed claim enter
        const unsigned long long want = __ballot(q < nq && (myfl & 1u));
        if (want) refresh(want);
      }
      uint32_t D = 0;
      int my_bd = 256, res_idx = -1, res_bd = 256;
#ifdef RESOLVE_STAMPS
      t_chunk += __builtin_readcyclecounter() - tc0;
#endif
      int s = 0;
      bool first_round = !M.serial;
      while (s < cnt) {
#ifdef RESOLVE_STAMPS
        long long tr0 = __builtin_readcyclecounter();
#endif
        // serial mode: one query at a time, decided from a list made at its turn (sees released claims as well)
        const int hi = M.serial ? s + 1 : cnt;
        if (M.serial) refresh(1ull << s);
        int r;
        unsigned long long flagged;
        for (;;) {
          const bool pend = lane >= s && lane < hi;
          const bool post = pend && (D >> 31) && ob;
          const int bidx = (int)(D & 0xfffffu);
          // a claim also takes the keypoint's stereo partner (ORBmatcher.cc:128-132, :199-203)
          int pidx = n;
          if (M.partner && post) { const int pr = sPartner[bidx]; pidx = pr == 0xffff ? n : pr; }
          if (post) __hip_atomic_fetch_min(&sOwner[bidx], (uint32_t)(lane + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (M.partner && pidx != n) __hip_atomic_fetch_min(&sOwner[pidx], (uint32_t)(lane + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __builtin_amdgcn_wave_barrier();
          uint32_t cm = 0;
#pragma unroll
          for (int j = 0; j < MATCH_TOPK; j++) cm |= (sOwner[eidx[j]] <= (uint32_t)lane ? 1u : 0u) << j;
          // withdraw the posts - unless the keypoint was committed since my (stale) decision was taken: a committed
          // claim (0) must survive
          const uint32_t mine = sOwner[post ? bidx : n];
          uint32_t mineP = 0u;
          if (M.partner) mineP = sOwner[pidx];
          __builtin_amdgcn_wave_barrier();
          if (post && mine != 0u) sOwner[bidx] = RESOLVE_FREE;
          if (M.partner && pidx != n && mineP != 0u) sOwner[pidx] = RESOLVE_FREE;
          __builtin_amdgcn_wave_barrier();
          uint32_t nD = pend ? decide<KT>(M, sTk + lane, vm, cm, oct4, truncated, lbDist, &my_bd) : D;
          if (M.couple == 1) {  // the right-image query of a map point whose left-image query failed the ratio test is dropped
            const uint32_t leftD = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)D, 0x111, 0xf, 0xf, true);  // D of lane - 1
            if (pend && (lane & 1) && ((leftD >> 29) & 1u)) { nD = 0; my_bd = 256; }
          }
          const bool changed = nD != D;
          D = nD;
          flagged = __ballot(pend && ((D >> 30) & 1u));
#ifdef RESOLVE_STAMPS
          n_round++;
#endif
          if (first_round) {
            first_round = false;
            if (flagged) {             // exhausted by the claims of earlier chunks: refresh them all at once
#ifdef RESOLVE_STAMPS
              t_round += __builtin_readcyclecounter() - tr0;
#endif
              refresh(flagged);
#ifdef RESOLVE_STAMPS
              tr0 = __builtin_readcyclecounter();
#endif
              continue;
            }
          }
          r = flagged ? (int)__builtin_ctzll(flagged) : hi;
          if (!__ballot(changed && lane <= r)) break;
        }
        // commit the settled prefix [s, r)
        {
          const bool inpre = lane >= s && lane < r;
          const bool acc = inpre && (D >> 31);
          const int bidx = (int)(D & 0xfffffu);
          nmatches += __popcll(__ballot(acc));
          if (acc) {
            const int32_t sv = (int32_t)(((uint32_t)q << 1) | (ob ? 1u : 0u));
            if (ob) sOwner[bidx] = 0u;
            __hip_atomic_fetch_max(&sSlot[bidx], sv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (M.partner) {
              const int pr = sPartner[bidx];
              if (pr != 0xffff) {
                // the partner's slot is overwritten unconditionally: a holder without observations RELEASES a claim
                // (only reachable in serial mode; the host selects it whenever such a query exists)
                sOwner[pr] = ob ? 0u : RESOLVE_FREE;
                __hip_atomic_fetch_max(&sSlot[pr], sv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
            }
          }
          if (M.partner) nmatches += __popcll(__ballot(acc && sPartner[acc ? bidx : 0] != 0xffff));
          if (inpre) { res_idx = acc ? bidx : -1; res_bd = my_bd <= M.th_dist ? my_bd : 256; }
        }
        __builtin_amdgcn_wave_barrier();
#ifdef RESOLVE_STAMPS
        t_round += __builtin_readcyclecounter() - tr0;
#endif
        s = r;
        if (r < hi) refresh(flagged);   // lane r is now first in line: with its fresh list it decides in the next round
      }
      if (q < nq) {
        if (M.match_of_query) M.match_of_query[qo + q] = res_idx;
        if (M.best_dist) M.best_dist[qo + q] = res_bd;
      }
    }
#ifdef RESOLVE_STAMPS
    if (lane == 0 && M.dbg) { long long *d = M.dbg + 16 * (size_t)p; d[0] = n_batch; d[1] = t_chunk; d[2] = t_round; d[3] = t_refresh; d[4] = n_refresh; d[5] = __builtin_readcyclecounter() - t0; d[6] = nq; d[7] = n_round; for (int i = 0; i < 6; i++) d[8 + i] = t_sub[i]; d[14] = sub_m; }
#endif
    if (lane == 0) { sCmd = -1; if (M.nmatches) M.nmatches[p] = nmatches; }
    __syncthreads();                   // (A) exit command
  }
  __syncthreads();                     // sSlot complete
  for (int i = tid; i < n; i += 64 * NW_) {
    const int32_t v = sSlot[i];
    if (v >= 0) { slot[i] = v >> 1; slot_obs[i] = (uint8_t)(v & 1); }
  }
}

// ORBmatcher::SearchForInitialization (ORBmatcher.cc:722-837): the sequential state is vMatchedDistance[i2], the
// distance at which keypoint i2 of F2 is currently matched; a later query sees i2 only while its own distance is
// SMALLER (:762), so visibility shrinks monotonically and k_match_scan's sorted top-8 list of a query stays the head of
// its visible candidates.  One wavefront walks the queries in order: lanes 0..7 test the visibility of the 8 entries
// (one ballot), the first two visible ones are best / second (:765-774); the accept rule is :777-779.  If fewer than two
// are visible and an unlisted candidate could still change the decision, the wave rescans all keypoints of F2 exactly.
// Runs once per monocular initialisation attempt (Tracking.cc:2471), so one wavefront per problem is enough.
// Output: match_of_query[q] = keypoint accepted AT q's turn (or -1), best_dist[q] = its distance; the caller replays the
// steals (:781-785) and the rotation histogram, which need the accept-time pairs.
template <typename KT>
__global__ __launch_bounds__(64) void k_init_resolve(MatchProblemSet M, const typename KT::T *topk, int th_low) {
  typedef typename KT::T K;
  extern __shared__ __align__(16) uint32_t smem_init[];
  __shared__ K sTk[MATCH_TOPK * 64];
  const int lane = threadIdx.x;
  const int p = blockIdx.x;
  const int n = M.frame_n ? M.frame_n[(size_t)p * M.frame_n_stride] : M.frame_n_const;
  const int nq = M.query_n ? M.query_n[(size_t)p * M.query_n_stride] : M.query_n_const;
  const size_t fo = (size_t)p * M.frame_stride, qo = (size_t)p * M.query_stride;
  const float *kp = M.kp + fo * 7;
  const uint32_t *desc = reinterpret_cast<const uint32_t *>(M.desc + fo * 32);
  uint16_t *sVMD = reinterpret_cast<uint16_t *>(smem_init);  // vMatchedDistance, 0xffff = INT_MAX (:733)
  for (int i = lane; i < n; i += 64) sVMD[i] = 0xffff;
  __builtin_amdgcn_s_barrier();
  int nacc = 0;
  for (int base = 0; base < nq; base += 64) {
    const int q = base + lane, cnt = min(64, nq - base);
#pragma unroll
    for (int j = 0; j < MATCH_TOPK; j++) sTk[64 * j + lane] = q < nq ? topk[(qo + q) * MATCH_TOPK + j] : KT::NONE;
    int res_idx = -1, res_bd = 256;
    for (int i = 0; i < cnt; i++) {
      // lanes 0..7: entry `lane` of query base+i
      const K k = lane < MATCH_TOPK ? sTk[64 * lane + i] : KT::NONE;
      const bool valid = k != KT::NONE;
      const int c = valid ? KT::idx(k) : 0, d = KT::dist(k);
      const bool vis = valid && (int)sVMD[c] > d;                                   // :762
      const uint32_t b = (uint32_t)__ballot(vis) & ((1u << MATCH_TOPK) - 1u);
      const int found = __popc(b);
      const int j1 = b ? __builtin_ctz(b) : 0, j2 = (b & (b - 1u)) ? __builtin_ctz(b & (b - 1u)) : 0;
      int bd = found > 0 ? __builtin_amdgcn_readlane(d, j1) : 0x7fffffff;
      int bd2 = found > 1 ? __builtin_amdgcn_readlane(d, j2) : 0x7fffffff;
      int best = found > 0 ? __builtin_amdgcn_readlane(c, j1) : -1;
      const K last = KT::readlane(k, MATCH_TOPK - 1);
      const bool truncated = last != KT::NONE;
      const int lbDist = KT::dist(last);
      bool rescan = false;
      if (truncated) {
        if (found == 0) rescan = lbDist <= th_low;
        else if (found == 1) rescan = bd <= th_low && !((float)bd < (float)lbDist * M.nnratio);
      }
      if (rescan) {  // exact best / second among the visible keypoints of the window, whole wavefront
        const QueryWin w = load_query(M, qo, base + i);
        const uint32_t *qd = reinterpret_cast<const uint32_t *>(M.qdesc + (qo + base + i) * 32);
        uint32_t q8[8];
#pragma unroll
        for (int t = 0; t < 8; t++) q8[t] = qd[t];
        K b1 = KT::NONE, b2 = KT::NONE;
        if (w.live) {
          for (int cc = lane; cc < n; cc += 64) {
            const float x = kp[(size_t)cc * 7], y = kp[(size_t)cc * 7 + 1];
            const uint32_t bits = cand_bits(x, y, __float_as_int(kp[(size_t)cc * 7 + 5]), false, M);
            if (cand_passes(w, x, y, bits, -1.f)) {
              int dist = 0;
#pragma unroll
              for (int t = 0; t < 8; t++) dist += __popc(desc[(size_t)cc * 8 + t] ^ q8[t]);
              if ((int)sVMD[cc] > dist) {
                const K key = KT::make(dist, cell_of(bits), cc);
                if (key < b1) { b2 = b1; b1 = key; }
                else if (key < b2) b2 = key;
              }
            }
          }
        }
        const K g1 = wave_min_key(b1);
        const K g2 = wave_min_key((b1 == g1) ? b2 : b1);
        bd = g1 != KT::NONE ? KT::dist(g1) : 0x7fffffff;
        bd2 = g2 != KT::NONE ? KT::dist(g2) : 0x7fffffff;
        best = g1 != KT::NONE ? KT::idx(g1) : -1;
      }
      // :777-779  bestDist<=TH_LOW && bestDist<(float)bestDist2*mfNNratio   (bestDist2 == INT_MAX when there is no second)
      const bool accept = best >= 0 && bd <= th_low && (float)bd < (float)bd2 * M.nnratio;
      if (accept) {
        if (lane == 0) sVMD[best] = (uint16_t)bd;                                   // :788
        nacc++;
      }
      if (lane == i) { res_idx = accept ? best : -1; res_bd = accept ? bd : 256; }
      __builtin_amdgcn_wave_barrier();
    }
    if (q < nq) {
      if (M.match_of_query) M.match_of_query[qo + q] = res_idx;
      if (M.best_dist) M.best_dist[qo + q] = res_bd;
    }
  }
  if (lane == 0 && M.nmatches) M.nmatches[p] = nacc;
}

// SearchForTriangulation inner loops (ORBmatcher.cc:1080-1153): one wavefront per unmatched keypoint of KF1, lanes over
// the keypoints of KF2 in the same vocabulary node.  The reference keeps a running best with `dist>bestDist -> skip`
// and updates it only when the geometric gates pass, i.e. the result is the LAST minimum among the gated candidates
// with dist <= TH_LOW: key = dist<<16 | (0xffff - position), wave minimum.
struct TriItem { int32_t idx1, start2, len2; };
struct TriParams {
  const float *kp1, *kp2;                 // keypoint AoS (7 floats each)
  const uint32_t *desc1, *desc2;
  const float *ur1, *ur2;
  const uint8_t *hasmp2;
  const int32_t *node_idx2;
  const TriItem *items; int nitems;
  float sf2[ORBX_MAX_LEVELS], sigma2_2[ORBX_MAX_LEVELS];
  float F12[9];
  float epx, epy;
  int bOnlyStereo, bCoarse;
  int32_t *matches12;
};

__global__ __launch_bounds__(256) void k_triangulation_match(TriParams T) {
  const int lane = threadIdx.x & 63;
  const int it = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (it >= T.nitems) return;
  const TriItem item = T.items[it];
  const int idx1 = item.idx1;
  uint32_t d1[8];
#pragma unroll
  for (int w = 0; w < 8; w++) d1[w] = T.desc1[(size_t)idx1 * 8 + w];
  const float x1 = T.kp1[(size_t)idx1 * 7], y1 = T.kp1[(size_t)idx1 * 7 + 1];
  const bool bStereo1 = T.ur1[idx1] >= 0.f;
  // epipolar line of kp1 in image 2, Pinhole.cpp:150-153
  const float a = x1 * T.F12[0] + y1 * T.F12[3] + T.F12[6];
  const float b = x1 * T.F12[1] + y1 * T.F12[4] + T.F12[7];
  const float c = x1 * T.F12[2] + y1 * T.F12[5] + T.F12[8];
  const float den = a * a + b * b;
  uint32_t best = 0xffffffffu;
  for (int j = lane; j < item.len2; j += 64) {
    const int idx2 = T.node_idx2[item.start2 + j];
    if (T.hasmp2[idx2]) continue;                                   // :1083 (vbMatched2 is never set)
    const bool bStereo2 = T.ur2[idx2] >= 0.f;
    if (T.bOnlyStereo && !bStereo2) continue;                       // :1088-1090
    int dist = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) dist += __popc(d1[w] ^ T.desc2[(size_t)idx2 * 8 + w]);
    if (dist > ORBM_TH_LOW) continue;                               // :1096
    const float x2 = T.kp2[(size_t)idx2 * 7], y2 = T.kp2[(size_t)idx2 * 7 + 1];
    const int oct2 = __float_as_int(T.kp2[(size_t)idx2 * 7 + 5]);
    if (!bStereo1 && !bStereo2) {                                   // :1105-1113
      const float distex = T.epx - x2, distey = T.epy - y2;
      if (distex * distex + distey * distey < 100 * T.sf2[oct2]) continue;
    }
    bool ok = T.bCoarse != 0;
    if (!ok && den != 0) {                                          // Pinhole.cpp:155-164
      const float num = a * x2 + b * y2 + c;
      const float dsqr = num * num / den;
      ok = (double)dsqr < 3.84 * (double)T.sigma2_2[oct2];
    }
    if (ok) {
      const uint32_t key = ((uint32_t)dist << 16) | (uint32_t)(0xffff - j);
      best = key < best ? key : best;
    }
  }
  best = wave_min_key(best);
  if (lane == 0) T.matches12[idx1] = best != 0xffffffffu ? T.node_idx2[item.start2 + (0xffff - (int)(best & 0xffff))] : -1;
}

// SearchForTriangulation for keyframes whose epipolar test is NOT Pinhole's (KannalaBrandt8, two-camera rigs; ORBmatcher.cc:1096-1153
// with pCamera1->epipolarConstrain(...) a virtual call into code outside this path): the device delivers, per unmatched keypoint of KF1,
// EVERY keypoint of KF2 in the same vocabulary node that passes the gates in front of the predicate - no map point (:1083), stereo
// filter (:1088-1090), distance <= TH_LOW (:1096), epipole distance (:1105-1113) - as keys dist << 16 | (0xffff - position); the host
// orders them and the caller's predicate picks (orbm_search_for_triangulation_pred).  One wavefront per item; the segment of the
// output buffer comes from one atomicAdd per wavefront, a segment that would not fit is counted but not written.
struct TriCandParams {
  const float *kp2;
  const uint32_t *desc1, *desc2;
  const float *ur1, *ur2;
  const uint8_t *hasmp2;
  const int32_t *node_idx2;
  const TriItem *items; int nitems;
  float sf2[ORBX_MAX_LEVELS];
  float epx, epy;
  int epipole_gate, bOnlyStereo;
  int32_t *item_off, *item_cnt;   // per item
  uint32_t *keys; int cap;        // output buffer
  int32_t *total;                 // running total (allocation counter)
};

__global__ __launch_bounds__(256) void k_triangulation_candidates(TriCandParams T) {
  const int lane = threadIdx.x & 63;
  const int it = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (it >= T.nitems) return;
  const TriItem item = T.items[it];
  const int idx1 = item.idx1;
  uint32_t d1[8];
#pragma unroll
  for (int w = 0; w < 8; w++) d1[w] = T.desc1[(size_t)idx1 * 8 + w];
  const bool bStereo1 = T.ur1[idx1] >= 0.f;
  auto key_of = [&](int j) -> uint32_t {      // 0xffffffff: not a candidate
    const int idx2 = T.node_idx2[item.start2 + j];
    if (T.hasmp2[idx2]) return 0xffffffffu;                               // :1083 (vbMatched2 is never set)
    const bool bStereo2 = T.ur2[idx2] >= 0.f;
    if (T.bOnlyStereo && !bStereo2) return 0xffffffffu;                   // :1088-1090
    int dist = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) dist += __popc(d1[w] ^ T.desc2[(size_t)idx2 * 8 + w]);
    if (dist > ORBM_TH_LOW) return 0xffffffffu;                           // :1096
    if (T.epipole_gate && !bStereo1 && !bStereo2) {                       // :1105-1113
      const float x2 = T.kp2[(size_t)idx2 * 7], y2 = T.kp2[(size_t)idx2 * 7 + 1];
      const int oct2 = __float_as_int(T.kp2[(size_t)idx2 * 7 + 5]);
      const float distex = T.epx - x2, distey = T.epy - y2;
      if (distex * distex + distey * distey < 100 * T.sf2[oct2 & (ORBX_MAX_LEVELS - 1)]) return 0xffffffffu;
    }
    return ((uint32_t)dist << 16) | (uint32_t)(0xffff - j);
  };
  int cnt = 0;
  for (int j0 = 0; j0 < item.len2; j0 += 64) {
    const int j = j0 + lane;
    cnt += __popcll(__ballot(j < item.len2 && key_of(j) != 0xffffffffu));
  }
  int base = 0;
  if (lane == 0) { base = cnt ? atomicAdd(T.total, cnt) : 0; T.item_off[it] = base; T.item_cnt[it] = cnt; }
  base = __builtin_amdgcn_readfirstlane(base);
  if (cnt == 0 || base + cnt > T.cap) return;
  int w0 = 0;
  for (int j0 = 0; j0 < item.len2; j0 += 64) {
    const int j = j0 + lane;
    const uint32_t k = j < item.len2 ? key_of(j) : 0xffffffffu;
    const unsigned long long mk = __ballot(k != 0xffffffffu);
    if (k != 0xffffffffu) T.keys[base + w0 + __popcll(mk & ((1ull << lane) - 1ull))] = k;
    w0 += __popcll(mk);
  }
}

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) inner loops (ORBmatcher.cc:303-438, Nleft == -1): the
// keypoints of the frame that fall into a vocabulary node can only be taken by keyframe keypoints of the same node, so
// the shared nodes are independent problems: one wavefront per shared node.  It walks the node's keyframe keypoints
// in order (:303); lanes hold the node's frame keypoints (position p = lane + 64 t, "taken" bit t in a lane register);
// best / second = two wave minima of (dist << 16 | position) - strict `<`, first minimum in position order (:326-335);
// accept rule :385-387, the taken frame keypoint is skipped by the rest of the node (:321).
struct BowItem { int32_t startKF, lenKF, startF, lenF; };
struct BowParams {
  const uint32_t *descKF, *descF;
  const uint8_t *hasmpKF;                 // pMP && !pMP->isBad()
  const int32_t *node_idxKF, *node_idxF;
  const BowItem *items; int nitems;
  float nnratio;
  int32_t *matchF;                        // [F.N] keyframe keypoint index or -1 (pre-filled with -1)
  // SearchByBoW(KeyFrame*, KeyFrame*, ...) (ORBmatcher.cc:839-979) runs on the same kernel with: hasmpF = the second
  // keyframe's keypoints that may be taken (:896-903), strict = accept needs bestDist1 < TH_LOW (:923, not <=),
  // match12[KF.N] = taken keypoint of the second keyframe per keypoint of the first (:927); all 0 / NULL otherwise
  const uint8_t *hasmpF; int strict; int32_t *match12;
  // fisheye-stereo frame (Frame::Nleft != -1, :338-363, :405-436): frame keypoints >= nleftF belong to the right image and
  // are matched separately (accepted whenever the LEFT best is within TH_LOW, `|| true` at :407); 0x7fffffff otherwise
  int nleftF;
};

__global__ __launch_bounds__(256) void k_bow_match(BowParams B) {
  const int lane = threadIdx.x & 63;
  const int it = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (it >= B.nitems) return;
  const BowItem item = B.items[it];
  const int nt = (item.lenF + 63) >> 6;     // <= 32 (host splits nothing: larger nodes are refused there)
  uint32_t taken = 0;
  for (int k = 0; k < item.lenKF; k++) {
    const int idxKF = B.node_idxKF[item.startKF + k];
    if (!B.hasmpKF[idxKF]) continue;                                 // :307-313
    uint32_t dk[8];
#pragma unroll
    for (int w = 0; w < 8; w++) dk[w] = B.descKF[(size_t)idxKF * 8 + w];
    const bool stereoF = B.nleftF != 0x7fffffff;
    uint32_t b1 = 0xffffffffu, b2 = 0xffffffffu, r1 = 0xffffffffu;
    for (int t = 0; t < nt; t++) {
      const int pos = lane + 64 * t;
      if (pos < item.lenF && !((taken >> t) & 1u)) {                 // :321
        const int idxF = B.node_idxF[item.startF + pos];
        if (B.hasmpF && !B.hasmpF[idxF]) continue;                     // :896-903
        int dist = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) dist += __popc(dk[w] ^ B.descF[(size_t)idxF * 8 + w]);
        const uint32_t key = ((uint32_t)dist << 16) | (uint32_t)pos;
        if (idxF < B.nleftF) {                                         // :326-335 resp. :347-354
          if (key < b1) { b2 = b1; b1 = key; }
          else if (key < b2) b2 = key;
        } else {
          r1 = key < r1 ? key : r1;                                    // :356-363 (the right second-best is never used, :407)
        }
      }
    }
    const uint32_t g1 = wave_min_key(b1);
    const uint32_t g2 = wave_min_key(b1 == g1 ? b2 : b1);
    const int bestDist1 = g1 != 0xffffffffu ? (int)(g1 >> 16) : 256, bestDist2 = g2 != 0xffffffffu ? (int)(g2 >> 16) : 256;
    const bool within = B.strict ? bestDist1 < ORBM_TH_LOW : bestDist1 <= ORBM_TH_LOW;
    if (within && (float)bestDist1 < B.nnratio * (float)bestDist2) {   // :385-387
      const int pos = (int)(g1 & 0xffffu);
      if (lane == (pos & 63)) {
        taken |= 1u << (pos >> 6);
        const int idxF = B.node_idxF[item.startF + pos];
        if (B.matchF) B.matchF[idxF] = idxKF;                          // :389
        if (B.match12) B.match12[idxKF] = idxF;                        // :927
      }
    }
    if (stereoF && within) {                                           // :405-436, inside `if(bestDist1<=TH_LOW)`
      const uint32_t gr = wave_min_key(r1);
      if (gr != 0xffffffffu && (int)(gr >> 16) <= ORBM_TH_LOW) {
        const int pos = (int)(gr & 0xffffu);
        if (lane == (pos & 63)) {
          taken |= 1u << (pos >> 6);
          if (B.matchF) B.matchF[B.node_idxF[item.startF + pos]] = idxKF;   // :409
        }
      }
    }
  }
}

// MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:350-436): among the N descriptors observing a map point, the one
// with the least MEDIAN Hamming distance to the others (median = sorted row [int(0.5*(N-1))], first minimum wins).
// One wavefront per map point: rows in order; the lanes hold the row's distances (j = lane + 64 t) and the k-th smallest
// is found by bisection on the value with ballot counts - no sort, no N x N matrix.
#define DISTINCT_MAXT 16   // N <= 1024 observations per map point
__global__ __launch_bounds__(256) void k_distinctive(const uint32_t *desc, const int32_t *start, int nmp, int32_t *best) {
  const int lane = threadIdx.x & 63;
  const int mp = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (mp >= nmp) return;
  const int s0 = start[mp], N = start[mp + 1] - s0;
  if (N <= 0) { if (lane == 0) best[mp] = -1; return; }
  const int nt = (N + 63) >> 6;
  const int k = (int)(0.5 * (double)(N - 1));                     // vDists[0.5*(N-1)], MapPoint.cc:423
  int bestMedian = 0x7fffffff, bestIdx = 0;
  for (int i = 0; i < N; i++) {
    uint32_t di[8];
#pragma unroll
    for (int w = 0; w < 8; w++) di[w] = desc[(size_t)(s0 + i) * 8 + w];
    int d[DISTINCT_MAXT];
#pragma unroll
    for (int t = 0; t < DISTINCT_MAXT; t++) {
      d[t] = 0x7fff;                                              // beyond every real distance: never counted
      const int j = lane + 64 * t;
      if (t < nt && j < N) {
        int dist = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) dist += __popc(di[w] ^ desc[(size_t)(s0 + j) * 8 + w]);
        d[t] = dist;
      }
    }
    int lo = 0, hi = 256;                                         // smallest v with #{j : d_ij <= v} >= k + 1
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      int cnt = 0;
#pragma unroll
      for (int t = 0; t < DISTINCT_MAXT; t++)
        if (t < nt) cnt += __popcll(__ballot(d[t] <= mid));
      if (cnt >= k + 1) hi = mid; else lo = mid + 1;
    }
    if (lo < bestMedian) { bestMedian = lo; bestIdx = i; }        // :424-428
  }
  if (lane == 0) best[mp] = bestIdx;
}

// cv::BFMatcher(NORM_HAMMING).knnMatch(query, train, matches, 2) as used by Frame::ComputeStereoFishEyeMatches
// (Frame.cc:1246): the two nearest train descriptors of every query descriptor.  One wavefront per query; ties in
// distance are ordered by train index (the caller's Lowe ratio test, :1252, cannot tell the orders apart).
__global__ __launch_bounds__(256) void k_knn2(const uint32_t *q, int nq, const uint32_t *c, int nc, int32_t *idx2, int32_t *dist2) {
  const int lane = threadIdx.x & 63;
  const int qi = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (qi >= nq) return;
  uint32_t qd[8];
#pragma unroll
  for (int w = 0; w < 8; w++) qd[w] = q[(size_t)qi * 8 + w];
  unsigned long long b1 = ~0ull, b2 = ~0ull;
  for (int j = lane; j < nc; j += 64) {
    int d = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) d += __popc(qd[w] ^ c[(size_t)j * 8 + w]);
    const unsigned long long key = ((unsigned long long)d << 32) | (unsigned long long)(uint32_t)j;
    if (key < b1) { b2 = b1; b1 = key; }
    else if (key < b2) b2 = key;
  }
  const unsigned long long g1 = wave_min_key(b1);
  const unsigned long long g2 = wave_min_key(b1 == g1 ? b2 : b1);
  if (lane == 0) {
    idx2[2 * qi] = g1 != ~0ull ? (int32_t)(uint32_t)g1 : -1; dist2[2 * qi] = g1 != ~0ull ? (int32_t)(g1 >> 32) : -1;
    idx2[2 * qi + 1] = g2 != ~0ull ? (int32_t)(uint32_t)g2 : -1; dist2[2 * qi + 1] = g2 != ~0ull ? (int32_t)(g2 >> 32) : -1;
  }
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

// ------------------------------------------------------------------------------------------------------------
// N1: Frame::ComputeStereoMatches (Frame.cc:901-1079), rectified stereo: one wavefront per LEFT keypoint.
//   1. lanes over the right keypoints: row-band test (the reference's vRowIndices table, :912-930, restated as
//      floor(yR - r) <= (int)yL <= ceil(yR + r), r = 2*scale[octR]), octave window, disparity window, Hamming;
//      wave minimum of (dist << 16 | iR) = first minimum in iR order (:966);
//   2. 11x11 SAD at 11 horizontal offsets on the two level images (lanes over the 121 pixels, DPP wave sums);
//   3. parabola fit, disparity, depth in the reference's fp32 expressions (:1021-1047).
// The median filter over the accepted matches (:1060-1073) needs a sort of <= N pairs and stays on the host.
// ------------------------------------------------------------------------------------------------------------
struct StereoParams {
  const uint8_t *imgL0, *imgR0; size_t strideL0, strideR0;   // level 0 of the chosen frames
  const uint8_t *pyrL, *pyrR;                                 // pyramid blocks of the chosen frames (levels >= 1)
  int w[ORB_MAXL], h[ORB_MAXL], pitch[ORB_MAXL]; size_t off[ORB_MAXL];
  float sf[ORB_MAXL], invsf[ORB_MAXL];
  int nlevels, rows;
  const float *kpL, *kpR;                                     // 7 floats per keypoint (mvKeys / mvKeysRight)
  const uint32_t *descL, *descR;
  int nL, nR;
  float mb, mbf;
  float *uRight, *depth; int32_t *sad;                        // per left keypoint; sad = -1: no match
};

__global__ __launch_bounds__(256) void k_stereo_match(StereoParams S) {
  const int lane = threadIdx.x & 63;
  const int iL = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (iL >= S.nL) return;
  const float uL = S.kpL[(size_t)iL * 7], vL = S.kpL[(size_t)iL * 7 + 1];
  const int levelL = __float_as_int(S.kpL[(size_t)iL * 7 + 5]);
  float outU = -1.0f, outD = -1.0f;
  int outSad = -1;
  const float minD = 0.f, maxD = S.mbf / S.mb;                // :933-935
  const float minU = uL - maxD, maxU = uL - minD;
  const int row = (int)vL;
  uint32_t best = 0xffffffffu;
  if (row >= 0 && row < S.rows && !(maxU < 0) && levelL >= 0 && levelL < S.nlevels) {
    uint32_t dl[8];
#pragma unroll
    for (int t = 0; t < 8; t++) dl[t] = S.descL[(size_t)iL * 8 + t];
    for (int iR = lane; iR < S.nR; iR += 64) {
      const float uR = S.kpR[(size_t)iR * 7], yR = S.kpR[(size_t)iR * 7 + 1];
      const int oR = __float_as_int(S.kpR[(size_t)iR * 7 + 5]);
      const float r = 2.0f * S.sf[min(max(oR, 0), ORB_MAXL - 1)];
      const int maxr = (int)ceilf(yR + r), minr = (int)floorf(yR - r);
      if (row < minr || row > maxr) continue;
      if (oR < levelL - 1 || oR > levelL + 1) continue;         // :954
      if (!(uR >= minU && uR <= maxU)) continue;                // :959
      int dist = 0;
#pragma unroll
      for (int t = 0; t < 8; t++) dist += __popc(dl[t] ^ S.descR[(size_t)iR * 8 + t]);
      if (dist < ORBM_TH_HIGH) best = min(best, ((uint32_t)dist << 16) | (uint32_t)iR);   // bestDist starts at TH_HIGH, strict <
    }
  }
  best = wave_min_key(best);
  const int thOrbDist = (ORBM_TH_HIGH + ORBM_TH_LOW) / 2;
  if (best != 0xffffffffu && (int)(best >> 16) < thOrbDist) {
    const int bestIdxR = (int)(best & 0xffffu);
    const float uR0 = S.kpR[(size_t)bestIdxR * 7];
    const float scaleFactor = S.invsf[levelL];
    const float scaleduL = roundf(uL * scaleFactor), scaledvL = roundf(vL * scaleFactor), scaleduR0 = roundf(uR0 * scaleFactor);
    constexpr int w = 5, L = 5;
    const int lw = S.w[levelL], lh = S.h[levelL];
    const int cuL = (int)scaleduL, cvL = (int)scaledvL, cuR = (int)scaleduR0;
    const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
    // cv::Mat::rowRange / colRange throw outside the matrix: such keypoints are skipped (same rule in the test oracle)
    const bool inside = cvL - w >= 0 && cvL + w + 1 <= lh && cuL - w >= 0 && cuL + w + 1 <= lw && !(iniu < 0 || endu >= (float)lw) && cuR - L - w >= 0;
    if (inside) {
      const uint8_t *IL, *IR;
      int pL, pR;
      if (levelL == 0) { IL = S.imgL0; IR = S.imgR0; pL = (int)S.strideL0; pR = (int)S.strideR0; }
      else { IL = S.pyrL + S.off[levelL]; IR = S.pyrR + S.off[levelL]; pL = pR = S.pitch[levelL]; }
      const int cL = IL[(size_t)cvL * pL + cuL];
      // my pixels of the 11x11 window: p = lane and lane + 64
      const int p0 = lane, p1 = lane + 64;
      const int y0 = p0 / 11 - w, x0 = p0 % 11 - w, y1 = p1 / 11 - w, x1 = p1 % 11 - w;
      const bool has1 = p1 < 121;
      const int a0 = (int)IL[(size_t)(cvL + y0) * pL + cuL + x0] - cL;
      const int a1 = has1 ? (int)IL[(size_t)(cvL + y1) * pL + cuL + x1] - cL : 0;
      int sadv[2 * L + 1];
#pragma unroll
      for (int k = 0; k <= 2 * L; k++) {
        const int incR = k - L;
        const int cR = IR[(size_t)cvL * pR + cuR + incR];
        const int b0 = (int)IR[(size_t)(cvL + y0) * pR + cuR + incR + x0] - cR;
        const int b1 = has1 ? (int)IR[(size_t)(cvL + y1) * pR + cuR + incR + x1] - cR : 0;
        sadv[k] = wave_sum_i32(abs(a0 - b0) + (has1 ? abs(a1 - b1) : 0));
      }
      int bestSad = 0x7fffffff, bestincR = 0;
#pragma unroll
      for (int k = 0; k <= 2 * L; k++)
        if ((float)sadv[k] < (float)bestSad) { bestSad = sadv[k]; bestincR = k - L; }       // :1006
      if (!(bestincR == -L || bestincR == L)) {
        float dist1 = 0.f, dist2 = 0.f, dist3 = 0.f;
#pragma unroll
        for (int k = 1; k < 2 * L; k++)
          if (k == bestincR + L) { dist1 = (float)sadv[k - 1]; dist2 = (float)sadv[k]; dist3 = (float)sadv[k + 1]; }
        const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));    // :1024
        if (!(deltaR < -1 || deltaR > 1)) {
          float bestuR = S.sf[levelL] * ((float)scaleduR0 + (float)bestincR + deltaR);       // :1030
          float disparity = uL - bestuR;
          if (disparity >= minD && disparity < maxD) {
            if (disparity <= 0) { disparity = (float)0.01; bestuR = (float)((double)uL - 0.01); }
            outD = S.mbf / disparity;
            outU = bestuR;
            outSad = bestSad;
          }
        }
      }
    }
  }
  if (lane == 0) { S.uRight[iL] = outU; S.depth[iL] = outD; S.sad[iL] = outSad; }
}

