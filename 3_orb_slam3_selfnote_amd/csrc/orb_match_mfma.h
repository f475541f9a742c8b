// orb_match_mfma.h -- k_match_rank and k_match_scan_mfma: the open-window candidate scan of the projection searches on the matrix
// pipe (formulation, operand layout and selection: orb_mfma_util.h).
#pragma once
#include "orb_mfma_util.h"

// One workgroup per frame pair: rank of every keypoint in (grid cell, index) order among the keypoints PosInGrid accepts.
//   rec[i]       = 2^18 + rank                     keypoint i usable (in the grid, not held by a map point with observations)
//                  2^18 + rank + 2^30              in the grid, but held: bit 30 lifts its keys above every real key
//                  2^18 + 2^30                     outside the grid
//   keyrec[rank] = cell << 11 | i                  the Key32 tie-break bits of the keypoint with that rank
// and the pair's vote: pairflag[p] = 1 <=> fuse_ok, 1 <= n <= 2048, the pair has a live query and every live query is open - the
// fused form of k_match_resolve then builds this pair's lists itself and the scan kernels leave it alone.
__global__ __launch_bounds__(MF_NT) void k_match_rank(MatchProblemSet M, uint32_t *rec, uint32_t *keyrec, uint32_t *pairflag, int fuse_ok) {
  __shared__ uint32_t sCell[GRID_CELLS + 4];
  __shared__ uint16_t sSorted[WALK_MAX_N];
  __shared__ uint32_t sWaveSum[MF_NT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int p = blockIdx.x;
  const int n = M.frame_n ? M.frame_n[(size_t)p * M.frame_n_stride] : M.frame_n_const;
  {
    const int nq = M.query_n ? M.query_n[(size_t)p * M.query_n_stride] : M.query_n_const;
    const size_t qo = (size_t)p * M.query_stride;
    bool anyLive = false, notOpen = false;
    if (fuse_ok && n >= 1 && n <= WALK_MAX_N)
      for (int q = tid; q < nq; q += MF_NT) {
        const QueryWin w = load_query(M, qo, q);
        anyLive = anyLive || w.live;
        notOpen = notOpen || (w.live && !query_is_open(M, w));
      }
    const int bad = __syncthreads_or(notOpen), alive = __syncthreads_or(anyLive);
    if (tid == 0) pairflag[p] = (alive && !bad) ? 1u : 0u;
  }
  if (n <= 0 || n > WALK_MAX_N) return;
  const size_t fo = (size_t)p * M.frame_stride;
  const float *kp = M.kp + fo * 7;
  for (int c = tid; c < GRID_CELLS + 4; c += MF_NT) sCell[c] = 0u;
  __syncthreads();
  constexpr int PER = WALK_MAX_N / MF_NT;
  uint32_t kbits[PER], krank[PER];
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const int i = tid + j * MF_NT;
    kbits[j] = 0u; krank[j] = 0u;
    if (i < n) {
      const float x = kp[(size_t)i * 7], y = kp[(size_t)i * 7 + 1];
      const int oct = __float_as_int(kp[(size_t)i * 7 + 5]);
      const bool claimed = M.slot[fo + i] >= 0 && M.slot_obs[fo + i];
      kbits[j] = cand_bits(x, y, oct, claimed, M);
      if ((kbits[j] >> 25) & 1u) krank[j] = atomicAdd(&sCell[cell_of(kbits[j])], 1u);
    }
  }
  __syncthreads();
  {  // exclusive prefix sums over the 3072 cells (12 consecutive cells per thread), as k_match_walk
    constexpr int CPT = GRID_CELLS / MF_NT;
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
    if (tid == MF_NT - 1) sCell[GRID_CELLS] = off;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const int i = tid + j * MF_NT;
    if (i < n && ((kbits[j] >> 25) & 1u)) sSorted[sCell[cell_of(kbits[j])] + krank[j]] = (uint16_t)i;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const int i = tid + j * MF_NT;
    if (i >= n) continue;
    uint32_t r = MF_REC_UNUSABLE;
    if ((kbits[j] >> 25) & 1u) {
      const uint32_t cell = cell_of(kbits[j]);
      const int s = (int)sCell[cell], e = (int)sCell[cell + 1];
      int before = 0;                                   // insertion order inside a cell = index order (Frame.cc:434-465)
      for (int t = s; t < e; t++) before += (int)sSorted[t] < i ? 1 : 0;
      const uint32_t rank = (uint32_t)(s + before);
      keyrec[fo + rank] = (cell << 11) | (uint32_t)i;
      r = MF_REC_BASE + rank + (((kbits[j] >> 24) & 1u) ? 0u : MF_REC_HELD);
    }
    rec[fo + i] = r;
  }
}

__global__ __launch_bounds__(MF_NT) void k_match_scan_mfma(MatchProblemSet M, uint32_t *topk, const uint32_t *rec, const uint32_t *keyrec, const uint32_t *pairflag) {
  // candidate tile in LDS: 32 rows x 16 chunks of 16 bytes, chunk c of row r at position r * 16 + ((c + r) & 15)
  __shared__ __align__(16) uint8_t sA[2][MF_TILE * 256];
  __shared__ __align__(16) uint32_t sRec[2][MF_TILE];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  const unsigned b = blockIdx.x, qblocks = (unsigned)M.scan_qblocks;   // same XCD-aware order as k_match_scan
  const int qb = (int)((b >> 3) % qblocks), p = (int)((b & 7u) + 8u * (b / (8u * qblocks)));
  if (p >= M.npairs) return;
  const int n = M.frame_n ? M.frame_n[(size_t)p * M.frame_n_stride] : M.frame_n_const;
  const int nq = M.query_n ? M.query_n[(size_t)p * M.query_n_stride] : M.query_n_const;
  if ((int)(qb * MF_NT) >= nq) return;
  if (pairflag[p]) return;                 // the fused k_match_resolve builds this pair's lists itself
  const size_t fo = (size_t)p * M.frame_stride, qo = (size_t)p * M.query_stride;
  const int q = qb * MF_NT + tid;          // = tile (lane >> 5), column (lane & 31) of this wavefront: the query this lane reports
  bool live = false, open = false;
  if (q < nq) {
    const QueryWin w = load_query(M, qo, q);
    live = w.live;
    open = query_is_open(M, w);
  }
  if (!block_is_open(live, open)) return;   // k_match_scan serves this block
  // ---- my two queries' B fragments: column `col` of query tile 0 and of tile 1, bits [32 s + 16 h, 32 s + 16 h + 16) per K-step s
  mf_v4i B0[8], B1[8];
  {
    const int q0 = qb * MF_NT + wid * 64 + col, q1 = q0 + 32;
    const uint4 *g0 = reinterpret_cast<const uint4 *>(M.qdesc + (qo + (size_t)min(q0, nq - 1)) * 32);
    const uint4 *g1 = reinterpret_cast<const uint4 *>(M.qdesc + (qo + (size_t)min(q1, nq - 1)) * 32);
    const uint4 a0 = g0[0], b0 = g0[1], a1 = g1[0], b1 = g1[1];
    const uint32_t d0[8] = {a0.x, a0.y, a0.z, a0.w, b0.x, b0.y, b0.z, b0.w}, d1[8] = {a1.x, a1.y, a1.z, a1.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int s = 0; s < 8; s++) {
      B0[s] = mf_expand16((d0[s] >> (16 * h)) & 0xffffu, MF_LUT_QUERY);
      B1[s] = mf_expand16((d1[s] >> (16 * h)) & 0xffffu, MF_LUT_QUERY);
    }
  }
  MfList<MATCH_TOPK> L0, L1;
  L0.init(); L1.init();
  const uint32_t *desc = reinterpret_cast<const uint32_t *>(M.desc + fo * 32);
  const uint32_t *recp = rec + fo;
  const int ntiles = (n + MF_TILE - 1) / MF_TILE;
  // staging: 512 (row, chunk) items per tile, two per thread; chunk c = 2 s + h' is halfword h' of descriptor word s
  const int it_r = tid >> 4, it_c = tid & 15;                         // item tid: row it_r (and it_r + 16), chunk it_c
  auto load_raw = [&](int t, uint32_t raw[2]) {
    const int base = t * MF_TILE;
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int r = it_r + 16 * k;
      raw[k] = base + r < n ? desc[(size_t)(base + r) * 8 + (it_c >> 1)] : 0u;
    }
  };
  auto stage = [&](int t, int buf, const uint32_t raw[2]) {
    const int base = t * MF_TILE;
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int r = it_r + 16 * k;
      const mf_v4i x = mf_expand16((raw[k] >> (16 * (it_c & 1))) & 0xffffu, MF_LUT_CAND);
      *reinterpret_cast<mf_v4i *>(&sA[buf][(r * 16 + ((it_c + r) & 15)) * 16]) = x;
    }
    if (tid < MF_TILE) sRec[buf][tid] = base + tid < n ? recp[base + tid] : MF_REC_UNUSABLE;
  };
#ifdef MF_STAMPS   // diagnostic builds only (tools/mfma_stamps.py): cycles per phase, thread 0 of every workgroup
  long long st_stage = 0, st_mfma = 0, st_sel = 0, st_sync = 0, st_t = __builtin_readcyclecounter();
  const long long st_begin = st_t;
#define MSTAMP(acc) do { const long long t_ = __builtin_readcyclecounter(); acc += t_ - st_t; st_t = t_; } while (0)
#else
#define MSTAMP(acc) do {} while (0)
#endif
  uint32_t raw[2];
  if (ntiles > 0) {
    load_raw(0, raw);
    stage(0, 0, raw);
    if (ntiles > 1) load_raw(1, raw);
  }
  __syncthreads();
  for (int t = 0; t < ntiles; t++) {
    const int buf = t & 1;
    if (t + 1 < ntiles) stage(t + 1, buf ^ 1, raw);
    if (t + 2 < ntiles) load_raw(t + 2, raw);
    MSTAMP(st_stage);
    mf_v16i c;
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const mf_v4i v = *reinterpret_cast<const mf_v4i *>(&sRec[buf][8 * g + 4 * h]);
      c[4 * g] = v[0]; c[4 * g + 1] = v[1]; c[4 * g + 2] = v[2]; c[4 * g + 3] = v[3];
    }
    mf_v16i acc0 = c, acc1 = c;
#pragma unroll
    for (int s = 0; s < 8; s++) {
      const mf_v4i a = *reinterpret_cast<const mf_v4i *>(&sA[buf][(col * 16 + ((2 * s + h + col) & 15)) * 16]);
      acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B0[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B1[s], acc1, 0, 0, 0);
    }
#ifdef MF_STAMPS
    asm volatile("" :: "v"(acc0), "v"(acc1));
    MSTAMP(st_mfma);
#endif
    L0.take(acc0);
    L1.take(acc1);
    MSTAMP(st_sel);
    __syncthreads();
    MSTAMP(st_sync);
  }
#ifdef MF_STAMPS
  if (tid == 0 && M.dbg) { long long *d = M.dbg + 8 * (size_t)blockIdx.x; d[0] = st_stage; d[1] = st_mfma; d[2] = st_sel; d[3] = st_sync; d[4] = st_t - st_begin; d[5] = st_begin; }
#endif
  // ---- the two lanes of a column hold the lists of the two halves of the frame: lane h reports tile h, so it sends the partner the
  // list of the OTHER tile and merges what it receives into its own tile's list
  uint32_t mine[MATCH_TOPK], other[MATCH_TOPK];
#pragma unroll
  for (int j = 0; j < MATCH_TOPK; j++) {
    mine[j] = h ? L1.top[j] : L0.top[j];
    const uint32_t send = h ? L0.top[j] : L1.top[j];
    other[j] = (uint32_t)__shfl_xor((int)send, 32, 64);
  }
  // min(a[i], b[7 - i]) are the 8 smallest of the union (a bitonic sequence), three compare-exchange stages sort them
  uint32_t mg[MATCH_TOPK];
#pragma unroll
  for (int j = 0; j < MATCH_TOPK; j++) mg[j] = min(mine[j], other[MATCH_TOPK - 1 - j]);
  auto cx = [](uint32_t &lo, uint32_t &hi) { const uint32_t l = min(lo, hi), g2 = max(lo, hi); lo = l; hi = g2; };
  cx(mg[0], mg[4]); cx(mg[1], mg[5]); cx(mg[2], mg[6]); cx(mg[3], mg[7]);
  cx(mg[0], mg[2]); cx(mg[1], mg[3]); cx(mg[4], mg[6]); cx(mg[5], mg[7]);
  cx(mg[0], mg[1]); cx(mg[2], mg[3]); cx(mg[4], mg[5]); cx(mg[6], mg[7]);
  if (q < nq) {
    uint32_t out[MATCH_TOPK];
#pragma unroll
    for (int j = 0; j < MATCH_TOPK; j++) {
      const uint32_t k = mg[j];
      const bool real = live && k < MF_KEY_LIMIT;
      const uint32_t kr = keyrec[fo + (real ? (k & 0x7ffu) : 0u)];
      out[j] = real ? ((k >> 11) << 23) | kr : 0xffffffffu;
    }
    uint4 *o = reinterpret_cast<uint4 *>(topk + (qo + q) * MATCH_TOPK);
    o[0] = make_uint4(out[0], out[1], out[2], out[3]);
    o[1] = make_uint4(out[4], out[5], out[6], out[7]);
  }
}

// ------------------------------------------------------------------------------------------------------------
// The same product behind the two brute-force entries (orbm_hamming_matrix, orbm_knn_match2): one operand streams through LDS in
// 32-descriptor tiles (A, rows), the other stays in registers (B, a wavefront's 64 columns), 16 MFMAs per tile and wavefront.
// ------------------------------------------------------------------------------------------------------------
// B fragments of one descriptor (column `col` of a 32-column tile), this lane's half h of every K-step
__device__ __forceinline__ void mf_b_frags(const uint32_t *desc8, int h, mf_v4i B[8]) {
  const uint4 a = reinterpret_cast<const uint4 *>(desc8)[0], b = reinterpret_cast<const uint4 *>(desc8)[1];
  const uint32_t d[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int s = 0; s < 8; s++) B[s] = mf_expand16((d[s] >> (16 * h)) & 0xffffu, MF_LUT_QUERY);
}
// A tile: rows [base, base + 32) of `desc` (n rows of 8 words; rows past n are zero descriptors) expanded into sA (8 KiB, chunk c of
// row r at position r * 16 + ((c + r) & 15)); all MF_NT threads, two (row, chunk) items each
__device__ __forceinline__ void mf_stage_rows(const uint32_t *desc, int n, int base, uint8_t *sA) {
  const int tid = threadIdx.x, it_r = tid >> 4, it_c = tid & 15;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int r = it_r + 16 * k;
    const uint32_t raw = base + r < n ? desc[(size_t)(base + r) * 8 + (it_c >> 1)] : 0u;
    *reinterpret_cast<mf_v4i *>(&sA[(r * 16 + ((it_c + r) & 15)) * 16]) = mf_expand16((raw >> (16 * (it_c & 1))) & 0xffffu, MF_LUT_CAND);
  }
}
__device__ __forceinline__ void mf_tile_product(const uint8_t *sA, int col, int h, const mf_v4i B0[8], const mf_v4i B1[8], mf_v16i &acc0, mf_v16i &acc1) {
#pragma unroll
  for (int s = 0; s < 8; s++) {
    const mf_v4i a = *reinterpret_cast<const mf_v4i *>(&sA[(col * 16 + ((2 * s + h + col) & 15)) * 16]);
    acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B0[s], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B1[s], acc1, 0, 0, 0);
  }
}

// K8 brute force on the matrix pipe: dist[i][j] = popcount(q_i ^ c_j).  Rows = queries (streamed), columns = candidates (a workgroup's
// 256, resident): a store instruction then writes two runs of 32 consecutive candidates of two query rows.  The accumulator starts at
// 2^18, so D = 2048 * ham.  Grid: (ceil(nc / 256), query slabs of qslab rows).
__global__ __launch_bounds__(MF_NT) void k_hamming_matrix_mfma(const uint32_t *q, int nq, const uint32_t *c, int nc, uint16_t *dist, int qslab) {
  __shared__ __align__(16) uint8_t sA[2][MF_TILE * 256];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, col = lane & 31, h = lane >> 5;
  const int c0 = blockIdx.x * MF_NT + wid * 64 + col, c1 = c0 + 32;
  mf_v4i B0[8], B1[8];
  mf_b_frags(c + (size_t)min(c0, nc - 1) * 8, h, B0);
  mf_b_frags(c + (size_t)min(c1, nc - 1) * 8, h, B1);
  const int qBeg = blockIdx.y * qslab, qEnd = min(qBeg + qslab, nq);
  const int ntiles = (qEnd - qBeg + MF_TILE - 1) / MF_TILE;
  if (ntiles > 0) mf_stage_rows(q, qEnd, qBeg, sA[0]);
  __syncthreads();
  for (int t = 0; t < ntiles; t++) {
    const int buf = t & 1, base = qBeg + t * MF_TILE;
    if (t + 1 < ntiles) mf_stage_rows(q, qEnd, base + MF_TILE, sA[buf ^ 1]);
    mf_v16i acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = (int)MF_REC_BASE; acc1[r] = (int)MF_REC_BASE; }
    mf_tile_product(sA[buf], col, h, B0, B1, acc0, acc1);
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = base + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < qEnd) {
        uint16_t *o = dist + (size_t)row * nc;
        if (c0 < nc) o[c0] = (uint16_t)((uint32_t)acc0[r] >> 11);
        if (c1 < nc) o[c1] = (uint16_t)((uint32_t)acc1[r] >> 11);
      }
    }
    __syncthreads();
  }
}

// knnMatch(k = 2) on the matrix pipe: columns = queries (a lane owns one), rows = train descriptors streamed in super-blocks of 2048:
// inside a super-block the accumulator seed 2^18 + (row index in the block) makes D = ham << 11 | index, a ready (distance, train
// index) key for the per-lane sorted pair (MfList<2>: one v_min and one v_med3 per key); at the end of a super-block the pair is
// rewritten with the global train index (ham << 20 | index, nc < 2^20) and merged into the running pair.
__global__ __launch_bounds__(MF_NT) void k_knn2_mfma(const uint32_t *q, int nq, const uint32_t *c, int nc, int32_t *idx2, int32_t *dist2) {
  __shared__ __align__(16) uint8_t sA[2][MF_TILE * 256];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, col = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * MF_NT + wid * 64 + col, q1 = q0 + 32;
  mf_v4i B0[8], B1[8];
  mf_b_frags(q + (size_t)min(q0, nq - 1) * 8, h, B0);
  mf_b_frags(q + (size_t)min(q1, nq - 1) * 8, h, B1);
  uint32_t g0[2] = {0xffffffffu, 0xffffffffu}, g1[2] = {0xffffffffu, 0xffffffffu};   // running pairs, global keys
  auto fold = [](uint32_t g[2], const MfList<2> &L, int sb) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const uint32_t k = L.top[j];
      const uint32_t key = k < MF_KEY_LIMIT ? ((k >> 11) << 20) | (uint32_t)(sb + (int)(k & 0x7ffu)) : 0xffffffffu;
      g[1] = mf_med3(g[0], g[1], key);
      g[0] = min(g[0], key);
    }
  };
  for (int sb = 0; sb < nc; sb += 2048) {
    const int sbEnd = min(sb + 2048, nc), ntiles = (sbEnd - sb + MF_TILE - 1) / MF_TILE;
    MfList<2> L0, L1;
    L0.init(); L1.init();
    __syncthreads();                       // the previous super-block's last tile is no longer read
    mf_stage_rows(c, sbEnd, sb, sA[0]);
    __syncthreads();
    for (int t = 0; t < ntiles; t++) {
      const int buf = t & 1, base = sb + t * MF_TILE;
      if (t + 1 < ntiles) mf_stage_rows(c, sbEnd, base + MF_TILE, sA[buf ^ 1]);
      mf_v16i acc0;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = t * MF_TILE + (r & 3) + 8 * (r >> 2) + 4 * h;      // index inside the super-block
        acc0[r] = (int)(sb + row < sbEnd ? MF_REC_BASE + (uint32_t)row : MF_REC_UNUSABLE);
      }
      mf_v16i acc1 = acc0;
      mf_tile_product(sA[buf], col, h, B0, B1, acc0, acc1);
      L0.take(acc0);
      L1.take(acc1);
      __syncthreads();
    }
    fold(g0, L0, sb);
    fold(g1, L1, sb);
  }
  // the two lanes of a column saw the two halves of every tile: lane h reports tile h's query and merges the partner's pair for it
  uint32_t mine[2], other[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    mine[j] = h ? g1[j] : g0[j];
    other[j] = (uint32_t)__shfl_xor((int)(h ? g0[j] : g1[j]), 32, 64);
  }
#pragma unroll
  for (int j = 0; j < 2; j++) { mine[1] = mf_med3(mine[0], mine[1], other[j]); mine[0] = min(mine[0], other[j]); }
  const int qi = h ? q1 : q0;
  if (qi < nq) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const bool real = mine[j] != 0xffffffffu;
      idx2[2 * qi + j] = real ? (int32_t)(mine[j] & 0xfffffu) : -1;
      dist2[2 * qi + j] = real ? (int32_t)(mine[j] >> 20) : -1;
    }
  }
}
