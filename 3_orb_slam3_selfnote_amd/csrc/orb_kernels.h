// orb_kernels.h -- hand-written gfx950 kernels of the ORB extractor (included once, by orbhip.hip).
//
// Kernel inventory (roofline per kernel in DESIGN.md):
//   k_resize    K1  cv::resize INTER_LINEAR 8U, one pyramid level of every frame per launch
//   k_pyramid_chain K1  the same planes in ONE launch for single-frame calls (a tile recomputes the levels below it in LDS)
//   k_fast      K2  FAST-9/16 score + per-cell NMS, cell detected at iniThFAST and again at minThFAST if empty, one workgroup per 30-px cell
//   k_octree    K3  DistributeOctTree, one workgroup per (frame, level), node list in LDS
//   k_blur      K5  7x7 sigma-2 fixed-point Gaussian as two int8 MFMA products, 128x32 tiles staged through LDS
//   k_describe  K4+K6+K7  IC_Angle + steered BRIEF + lapping-order scatter + frame totals, one wavefront (64 lanes) per keypoint
// All arithmetic is integer or non-contracted IEEE fp32/fp64 (hipcc -ffp-contract=off) so results are bit-exact
// against the CPU restatement the tests use.  No MFMA: this is byte/bit work bound by HBM, LDS and VALU integer rate.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "orb_common.h"
#include "orb_sincos.h"

#define WAVE 64

__constant__ __attribute__((aligned(16))) int8_t c_pattern[1024] = {
#include "../../include/orb_pattern_data.inc"
};

// Offsets (u, v) of the 749 pixels of the radius-15 intensity-centroid disc, rows v = -15..15, u = -umax[|v|]..umax[|v|]
// (ORBextractor.cc:452-467 always yields umax = 15,15,15,15,14,14,14,13,13,12,11,10,9,8,6,3 for HALF_PATCH_SIZE 15;
// orbx_create re-derives umax at run time and refuses to start if it differs).  Padded to 768 with (0,0), which
// contributes u*I = v*I = 0 to the moments.
// Stored per lane: lane L handles disc pixels L, 64+L, ..., 704+L; its twelve u offsets are bytes 0..11 and its twelve
// v offsets bytes 12..23 of w[L][0..5], so one lane fetches its share of the table with two wide loads.
struct DiscTab { uint32_t w[64][6]; };
constexpr DiscTab make_disc_tab() {
  DiscTab t{};
  const int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
  int n = 0;
  for (int v = -15; v <= 15; v++) {
    const int d = umax[v < 0 ? -v : v];
    for (int u = -d; u <= d; u++) {
      const int lane = n & 63, k = n >> 6;  // pixel n = k * 64 + lane
      t.w[lane][k >> 2] |= (uint32_t)(uint8_t)(int8_t)u << (8 * (k & 3));
      t.w[lane][3 + (k >> 2)] |= (uint32_t)(uint8_t)(int8_t)v << (8 * (k & 3));
      n++;
    }
  }
  return t;  // pixels 749..767 stay (0, 0)
}
__constant__ __attribute__((aligned(16))) DiscTab c_disc = make_disc_tab();

// ------------------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
  return p;
}

__device__ __forceinline__ const uint8_t *level_plane(const FrameParams &P, int frame, int level, int &pitch) {
  if (level == 0) {
    pitch = (int)P.img0_stride;
    return P.img0 + (size_t)frame * P.img0_frame_stride;
  }
  pitch = P.geom[level].pitch;
  return P.pyr + (size_t)frame * P.pyr_fs + P.geom[level].off;
}

// XCD-aware block -> (frame, item) mapping.  Workgroups are dealt round-robin over the 8 XCDs (block b lands on XCD
// b % 8, MI355X_MICROARCH.md "Workgroup dispatch"), and each XCD has its own 4 MiB L2.  All items of frame f are
// therefore placed on XCD f % 8: neighbouring FAST cells / blur tiles / keypoint patches of one frame share cache
// lines and halo rows, and with this mapping they are fetched into ONE L2 instead of up to eight.  Placement only
// affects speed, never results.  Grid: 1-D, nframes * nitems blocks.
// magic = floor(2^32 / nitems) (host): the quotient estimate is at most one too small, one correction makes it exact.
__device__ __forceinline__ void xcd_map(int nitems, uint32_t magic, int nframes, int &frame, int &item) {
  const unsigned b = blockIdx.x;
  const unsigned full = (unsigned)(nframes & ~7);
  const unsigned nfull = full * (unsigned)nitems;
  const bool inFull = b < nfull;
  const unsigned q = inFull ? (b >> 3) : (b - nfull);
  unsigned g = __umulhi(q, magic);
  unsigned r = q - g * (unsigned)nitems;
  if (r >= (unsigned)nitems) { g++; r -= (unsigned)nitems; }
  frame = (int)(inFull ? (b & 7u) + 8u * g : full + g);
  item = (int)r;
}

// Wave-wide sum through DPP (row_shr 1/2/4/8, row_bcast15, row_bcast31): no LDS traffic; the result is uniform.
// All 64 lanes must be active.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_or_zero(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROWMASK, 0xf, false); }
// Low 32 bits of a 24 x 24-bit product: v_mul_u32_u24, full rate (v_mul_lo_u32 / v_mul_hi_u32 issue at a quarter of it).
// Inline assembly because the instruction selector only forms it when it can prove both operands 24-bit, which it cannot
// for loop indices bounded by a compare.  mul24(a, u): u wave-uniform (scalar operand).
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t u) {
  uint32_t d;
  asm("v_mul_u32_u24 %0, %1, %2" : "=v"(d) : "s"(u), "v"(a));
  return d;
}
// both operands per lane; the caller guarantees 24 bits
__device__ __forceinline__ uint32_t mul24_vv(uint32_t a, uint32_t b) { return __umul24(a, b); }

__device__ __forceinline__ int wave_sum_i32(int v) {
  v += dpp_or_zero<0x111, 0xf>(v);
  v += dpp_or_zero<0x112, 0xf>(v);
  v += dpp_or_zero<0x114, 0xf>(v);
  v += dpp_or_zero<0x118, 0xf>(v);
  v += dpp_or_zero<0x142, 0xa>(v);
  v += dpp_or_zero<0x143, 0xc>(v);
  return __builtin_amdgcn_readlane(v, 63);
}

// inclusive scan inside a wavefront
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
  // DPP: row_shr 1, 2, 4, 8 scan each row of 16 lanes, row_bcast15 / row_bcast31 carry the row totals on (six vector instructions;
  // the __shfl_up form is six trips through the LDS crossbar)
  int v = (int)x;
  v += dpp_or_zero<0x111, 0xf>(v);
  v += dpp_or_zero<0x112, 0xf>(v);
  v += dpp_or_zero<0x114, 0xf>(v);
  v += dpp_or_zero<0x118, 0xf>(v);
  v += dpp_or_zero<0x142, 0xa>(v);
  v += dpp_or_zero<0x143, 0xc>(v);
  return (uint32_t)v;
}

// Exclusive scan of an LDS array a[0..n) in place using all NT threads (blockDim.x == NT, 1-D block).
// Returns the total.  sw: LDS scratch of NT/64 + 2 words.
template <int NT>
__device__ uint32_t lds_excl_scan(uint32_t *a, int n, uint32_t *sw) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  constexpr int NW = NT / 64;
  uint32_t carry = 0;
  for (int base = 0; base < n; base += NT) {
    int i = base + tid;
    uint32_t v = i < n ? a[i] : 0u;
    uint32_t inc = wave_incl_scan(v);
    if (lane == 63) sw[wid] = inc;
    __syncthreads();
    // the NW wave totals are scanned in registers by every wavefront (one LDS read per lane instead of NW per thread)
    const uint32_t wsum = lane < NW ? sw[lane] : 0u;
    const uint32_t wincl = wave_incl_scan(wsum);
    const int widu = __builtin_amdgcn_readfirstlane(wid);
    const uint32_t woff = (uint32_t)__builtin_amdgcn_readlane((int)(wincl - wsum), widu);
    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)wincl, NW - 1);
    if (i < n) a[i] = carry + woff + inc - v;
    carry += tot;
    __syncthreads();
  }
  return carry;
}

// ------------------------------------------------------------------------------------------------------------
// K1: cv::resize INTER_LINEAR 8UC1.  Level `level` of every frame from level-1.
// Tables {sx, a0|a1<<16} / {sy, b0|b1<<16} are built on the host (orbx_configure); the result is
//   (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2,   r = S[sy][sx] * a0 + S[sy][sx + 1] * a1   for rows sy, sy + 1.
// Workgroup = G.resizeRows (16 or 8) output rows x full width.
//
// Two-pass form (tPitch > 0, the host's choice whenever a tile's source rows fit the LDS stage): the horizontal interpolation r >> 4
// of a source row is shared by the two output rows that straddle it, so it is evaluated once per (source row, output column) -
// about 1.35 per output pixel for 16-row tiles at scale 1.2 instead of 2 - into a 16-bit LDS plane T, and the vertical pass reads
// four finished values with one 8-byte LDS load (SQ_INSTS_VALU per launch 13.3 M -> 9.8 M against the one-pass gather form).
// One-pass form (tPitch == 0): extreme scale factors, straight from global memory.
// ------------------------------------------------------------------------------------------------------------
// Output rows per workgroup: 16 (late round 3; 8 before) - a tile's source rows come in with ONE HBM round trip, and twice the rows
// behind it is twice the work per wait (the same reasoning as k_blur's 64-row tiles); alone on the chip the kernel is as fast as before,
// beside the other pipelines the step gains 2 % (16: 208-211 k frames/s, 8: 204-207 k, 24: 206 k, 32: 201 k).  Levels whose 16-row stage
// would not fit the LDS (images wider than ~2600 columns) take 8-row tiles (G.resizeRows, decided in orbx_configure).
#define RESIZE_ROWS_MAX 16
#define RESIZE_MAXSRC 32   // most source rows per tile the two-pass form stages (16 rows at scale 2: 32)
__global__ __launch_bounds__(256) void k_resize(FrameParams P, int level, int smemRowBytes, int tPitch, int maxSrc) {
  extern __shared__ __align__(16) uint8_t smem_rs[];
  const LevelGeom G = P.geom[level];
  const LevelGeom Gs = P.geom[level - 1];
  const int tid = threadIdx.x;
  int frame, rowTile;
  xcd_map((G.h + G.resizeRows - 1) / G.resizeRows, G.rowTileMagic, P.nframes, frame, rowTile);
  const int dy0 = rowTile * G.resizeRows;
  const int nrows = min(G.resizeRows, G.h - dy0);
  int spitch;
  const uint8_t *src = level_plane(P, frame, level - 1, spitch);
  uint8_t *dstplane = P.pyr + (size_t)frame * P.pyr_fs + G.off;
  const int wq = (G.w + 3) & ~3, qw = wq >> 2;   // the x table is padded to whole output quads with copies of its last entry (host)
  // the tile's y-coefficients go through LDS: the row code below then has no global load in front of it
  __shared__ int2 sY[RESIZE_ROWS_MAX];
  if (tid < nrows) sY[tid] = P.ytab[G.ytabBase + dy0 + tid];

  if (tPitch > 0) {
    // source row span of this tile (at most maxSrc rows: the host derived maxSrc from the same table)
    const int syFirst = min(max(P.ytab[G.ytabBase + dy0].x, 0), Gs.h - 1);
    const int syLast = min(max(P.ytab[G.ytabBase + dy0 + nrows - 1].x + 1, 0), Gs.h - 1);
    const int nsrc = min(syLast - syFirst + 1, maxSrc);
    uint8_t *sRows = smem_rs;                                  // [maxSrc][smemRowBytes]
    uint8_t *sT = sRows + (size_t)maxSrc * smemRowBytes;       // [maxSrc][tPitch] bytes, u16 entries
    uint4 *sRowRec = reinterpret_cast<uint4 *>(sT + (size_t)maxSrc * tPitch);   // [RESIZE_ROWS_MAX]
    // this thread's first two column pairs of pass 1, loaded ahead of the staging (a level up to 1024 columns wide needs no more)
    const int4 *xtab4 = reinterpret_cast<const int4 *>(P.xtab + G.xtabBase);
    int4 xtPre[2];
#pragma unroll
    for (int k = 0; k < 2; k++) { const int p = tid + 256 * k; xtPre[k] = 2 * p < wq ? xtab4[p] : make_int4(0, 0, 0, 0); }
    const bool aligned = ((((uintptr_t)src) | (uintptr_t)spitch) & 3u) == 0;
    // Index arithmetic: 32-bit integer multiplies issue at a quarter of the full rate, so idx / ndw is taken in float --
    // (idx + 0.5) / ndw stays at least 0.5 / ndw >= 2^-11 away from an integer while the rounding error of the product is
    // below 2^-19 for idx < 2^14, so truncation yields the exact quotient -- and offsets use 24-bit multiplies (mul24).
    const uint8_t *src0 = src + (size_t)syFirst * spitch;
    if (aligned) {
      // Full 16-byte chunks of every source row by LDS-DMA (global_load_lds_dwordx4: no register, no ds_write).  The LDS rows
      // are smemRowBytes = 16 * cpr long; the loop runs over all cpr chunk positions so that the LDS image stays lane-linear
      // (destination = wave base + lane * 16), the positions past the last full chunk simply stay inactive.
      const int cpr = smemRowBytes >> 4, nfull = Gs.w >> 4;
      const float inv_cpr = 1.0f / (float)cpr;
      for (int idx = tid; idx < cpr * nsrc; idx += 256) {
        const int r = (int)(((float)idx + 0.5f) * inv_cpr), c = idx - (int)mul24((uint32_t)r, (uint32_t)cpr);
        if (c < nfull)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src0 + (mul24((uint32_t)r, (uint32_t)spitch) + 16u * (uint32_t)c)),
                                           (__attribute__((address_space(3))) void *)(sRows + (size_t)(idx - (tid & 63)) * 16), 16, 0, 0);
      }
      // the tail of every row (< 16 bytes, up to 4 dwords) through registers
      const int ndw = (Gs.w + 3) >> 2, d0 = 4 * nfull, ntail = ndw - d0;   // the last dword may read up to 3 bytes of row padding / next row: inside the plane
      const bool lastRowPartial = (Gs.w & 3) != 0;
      for (int idx = tid; idx < ntail * nsrc; idx += 256) {
        const int r = idx / ntail, d = d0 + (idx - r * ntail);
        const uint8_t *rowp = src0 + mul24((uint32_t)r, (uint32_t)spitch);
        uint32_t v;
        if (lastRowPartial && d == ndw - 1 && syFirst + r == Gs.h - 1 && spitch < 4 * ndw) {
          v = 0;  // very last bytes of the plane: do not read past the allocation
          for (int b = 0; 4 * d + b < Gs.w; b++) v |= (uint32_t)rowp[4 * d + b] << (8 * b);
        } else {
          v = *reinterpret_cast<const uint32_t *>(rowp + 4 * d);
        }
        *reinterpret_cast<uint32_t *>(sRows + mul24((uint32_t)r, (uint32_t)smemRowBytes) + 4 * d) = v;
      }
    } else {
      const float inv_w = 1.0f / (float)Gs.w;
      for (int idx = tid; idx < Gs.w * nsrc; idx += 256) {
        const int r = (int)(((float)idx + 0.5f) * inv_w), c = idx - (int)mul24((uint32_t)r, (uint32_t)Gs.w);
        sRows[mul24((uint32_t)r, (uint32_t)smemRowBytes) + c] = src0[mul24((uint32_t)r, (uint32_t)spitch) + c];
      }
    }
    // per output row of the tile: where its two T rows start, its coefficient pair, where it goes
    if (tid < nrows) {
      const int2 yt = P.ytab[G.ytabBase + dy0 + tid];
      const int sy0 = min(max(yt.x, 0), Gs.h - 1), sy1 = min(max(yt.x + 1, 0), Gs.h - 1);
      sRowRec[tid] = make_uint4((uint32_t)((sy0 - syFirst) * tPitch), (uint32_t)((sy1 - syFirst) * tPitch), (uint32_t)yt.y, (uint32_t)((dy0 + tid) * G.pitch));
    }
    __syncthreads();
    // Pass 1: T[r][c] = (S[r][sx] * a0 + S[r][sx + 1] * a1) >> 4, at most 255 * 2048 / 16 = 32640: 16 bits.  Thread = two neighbouring
    // output columns, loop over the staged source rows.  Per row ONE aligned 8-byte window [base, base + 8), base = sx_A & ~3, holds all
    // four source bytes (sx_B - sx_A <= 3: the host takes this form for horizontal scale factors below 3 only): v_perm_b32 picks each
    // column's byte pair out of it as two 16-bit halves, v_dot2_u32_u16 multiplies by the column's coefficient pair - pre-scaled by 16
    // (a << 4 <= 2^15 still fits its half), which moves the >> 4 to a byte boundary: the product's bytes 1..2 are T - and a third
    // v_perm_b32 packs the two T values into one 32-bit store: five vector instructions per row for two columns, and no byte access
    // that straddles a dword (the straightforward byte-pair read compiles to an unaligned ds_read_u16, which the LDS serialises:
    // measured 25 cycles per LDS instruction, 3x the whole kernel's time).
    // Column sx + 1 = w exists in the LDS rows (they are padded to 16-byte chunks) and is only ever read with weight a1 = 0 (the table
    // clamps sx to w - 1 with fx = 0 there); the window's other bytes may be anything, they are never selected.
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    auto dot2 = [](uint32_t pix, uint32_t coef) -> uint32_t {
      u16x2 x, y;
      __builtin_memcpy(&x, &pix, 4); __builtin_memcpy(&y, &coef, 4);
      return __builtin_amdgcn_udot2(x, y, 0u, false);
    };
    int trip = 0;
    for (int p = tid; 2 * p < wq; p += 256, trip++) {
      const int4 xt = trip == 0 ? xtPre[0] : trip == 1 ? xtPre[1] : xtab4[p];     // {sx_A, coef_A, sx_B, coef_B}
      const uint32_t base = (uint32_t)xt.x & ~3u, oA = (uint32_t)xt.x & 3u, oB = (uint32_t)xt.z - base;
      const uint32_t selA = 0x0c000c00u | oA | ((oA + 1u) << 16), selB = 0x0c000c00u | oB | ((oB + 1u) << 16);
      const uint32_t kA = (uint32_t)xt.y << 4, kB = (uint32_t)xt.w << 4;
      const uint8_t *sp = sRows + base;
      uint8_t *tp = sT + 4 * p;
      // four rows per trip, all loads first: the compiler may not move an LDS load over an LDS store (T and the staged rows could alias
      // for all it knows), and one load -> store chain per row is one LDS latency per row
      for (int r = 0; r < nsrc; r += 4) {
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t *q = reinterpret_cast<const uint32_t *>(sp + min(r + k, nsrc - 1) * smemRowBytes);       // uniform: scalar arithmetic
          lo[k] = q[0]; hi[k] = q[1];
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (r + k < nsrc) {
            const uint32_t vA = dot2(__builtin_amdgcn_perm(hi[k], lo[k], selA), kA), vB = dot2(__builtin_amdgcn_perm(hi[k], lo[k], selB), kB);
            *reinterpret_cast<uint32_t *>(tp + (r + k) * tPitch) = __builtin_amdgcn_perm(vB, vA, 0x06050201u);
          }
      }
    }
    __syncthreads();
    // Pass 2: 32 threads per output row, eight rows at a time, each 4 consecutive output pixels per trip: everything that
    // depends on the row alone is read once, a trip is two 8-byte LDS loads, the arithmetic and one 32-bit store.  No saturate_cast
    // needed: the coefficients are non-negative and each pair sums to 2048, so v is a convex combination of four bytes, rounded
    // down-ish: always inside [0, 255].
    for (int ry = tid >> 5; ry < nrows; ry += 8) {
      const uint4 rec = sRowRec[ry];
      const uint32_t b0 = rec.z & 0xffffu, b1 = rec.z >> 16;
      const uint8_t *T0 = sT + rec.x, *T1 = sT + rec.y;
      uint8_t *dst = dstplane + rec.w;
      for (int qx = tid & 31; qx < qw; qx += 32) {
        const uint2 t0 = *reinterpret_cast<const uint2 *>(T0 + 8 * qx), t1 = *reinterpret_cast<const uint2 *>(T1 + 8 * qx);
        const uint32_t u0[4] = {t0.x & 0xffffu, t0.x >> 16, t0.y & 0xffffu, t0.y >> 16};
        const uint32_t u1[4] = {t1.x & 0xffffu, t1.x >> 16, t1.y & 0xffffu, t1.y >> 16};
        uint32_t packed = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const uint32_t v = ((mul24_vv(b0, u0[j]) >> 16) + (mul24_vv(b1, u1[j]) >> 16) + 2) >> 2;
          packed |= v << (8 * j);
        }
        const int dx0 = 4 * qx;
        if (dx0 + 3 < G.w) *reinterpret_cast<uint32_t *>(dst + dx0) = packed;
        else for (int j = 0; j < 4 && dx0 + j < G.w; j++) dst[dx0 + j] = (uint8_t)(packed >> (8 * j));
      }
    }
    return;
  }

  // One-pass form: the x table in LDS, the four source bytes of every output pixel straight from global memory.
  int2 *sX = reinterpret_cast<int2 *>(smem_rs);                       // [wq]
  for (int i = tid; i < wq; i += 256) sX[i] = P.xtab[G.xtabBase + i];
  __syncthreads();
  // q / qw in float: (q + 0.5) / qw is at least 0.5 / qw >= 2^-11 away from an integer, the product's rounding error is
  // below 2^-19 for q < 8 * 1024, so truncation gives the exact quotient (four full-rate operations)
  const float inv_qw = 1.0f / (float)qw;
  for (int q = tid; q < qw * nrows; q += 256) {
    const int ry = (int)(((float)q + 0.5f) * inv_qw), dx0 = (q - (int)mul24((uint32_t)ry, (uint32_t)qw)) * 4, dy = dy0 + ry;
    const int2 yt = sY[ry];
    const int sy0 = min(max(yt.x, 0), Gs.h - 1), sy1 = min(max(yt.x + 1, 0), Gs.h - 1);
    const int b0 = yt.y & 0xffff, b1 = (yt.y >> 16) & 0xffff;
    const uint8_t *S0 = src + mul24((uint32_t)sy0, (uint32_t)spitch), *S1 = src + mul24((uint32_t)sy1, (uint32_t)spitch);
    uint32_t packed = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int2 xt = sX[dx0 + j];            // columns past the row end repeat the last one (padded table); only [0, w) is stored
      const int sx = xt.x, sx1 = min(sx + 1, Gs.w - 1);
      const int a0 = xt.y & 0xffff, a1 = (xt.y >> 16) & 0xffff;
      const int r0 = S0[sx] * a0 + S0[sx1] * a1;
      const int r1 = S1[sx] * a0 + S1[sx1] * a1;
      const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
      packed |= (uint32_t)v << (8 * j);
    }
    uint8_t *dst = dstplane + mul24((uint32_t)dy, (uint32_t)G.pitch);
    if (dx0 + 3 < G.w) *reinterpret_cast<uint32_t *>(dst + dx0) = packed;
    else for (int j = 0; j < 4 && dx0 + j < G.w; j++) dst[dx0 + j] = (uint8_t)(packed >> (8 * j));
  }
}

// ------------------------------------------------------------------------------------------------------------
// K1, single-frame form.  A frame on its own is latency: seven dependent k_resize launches of a few microseconds each cost more
// in launch-to-launch gaps than in work.  Here ONE launch makes every level: a workgroup owns a 32x32 tile of level L and
// recomputes, in LDS, the rectangles of levels 1..L-1 that tile depends on (sizes grow 1.2x per level down: 141x141 of level 0
// for a level-7 tile), each from the one below with exactly k_resize's arithmetic on exactly its table entries - a level's pixel
// is the same function of the same bytes whoever computes it, so the planes are byte-identical.  About 10x the arithmetic of
// the level-by-level form, on a chip that a single frame leaves empty; batches keep k_resize.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CHAIN_NT) void k_pyramid_chain(FrameParams P) {
  extern __shared__ __align__(16) uint8_t smem_ch[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  constexpr int NW = CHAIN_NT / 64;
  const int ntile = (int)gridDim.x / P.nframes;
  const int frame = (int)blockIdx.x / ntile, tile = (int)blockIdx.x - frame * ntile;
  __shared__ ChainTile T;   // indexed by level at run time: LDS, not registers
  static_assert(sizeof(ChainTile) % 4 == 0 && sizeof(ChainTile) / 4 <= CHAIN_NT, "one dword per thread");
  if (tid < (int)(sizeof(ChainTile) / 4)) reinterpret_cast<uint32_t *>(&T)[tid] = reinterpret_cast<const uint32_t *>(&P.chain[tile])[tid];
  __syncthreads();
  const int L = T.level;
  uint8_t *buf[2] = {smem_ch, smem_ch + P.chainBuf0};
  int2 *sTab = reinterpret_cast<int2 *>(smem_ch + P.chainBuf0 + P.chainBuf1);   // per level 1..L: x entries of rect[l], then y entries
  // table slices of every stage, one memory latency for all of them
  {
    int off = 0;
    for (int l = 1; l <= L; l++) {
      const LevelGeom &G = P.geom[l];
      for (int i = tid; i < T.w[l] + T.h[l]; i += CHAIN_NT)
        sTab[off + i] = i < T.w[l] ? P.xtab[G.xtabBase + T.x[l] + i] : P.ytab[G.ytabBase + T.y[l] + (i - T.w[l])];
      off += T.w[l] + T.h[l];
    }
  }
  // rect[0] of the caller's image -> buffer 0 (rows by wavefront, columns by lane)
  {
    const uint8_t *img = P.img0 + (size_t)frame * P.img0_frame_stride + (size_t)T.y[0] * P.img0_stride + T.x[0];
    const int rw = T.w[0], rh = T.h[0];
    for (int r = wid; r < rh; r += NW)
      for (int c = lane; c < rw; c += 64) buf[0][r * rw + c] = img[(size_t)r * P.img0_stride + c];
  }
  __syncthreads();
  int toff = 0;
  for (int l = 1; l <= L; l++) {
    const LevelGeom &G = P.geom[l];
    const LevelGeom &Gs = P.geom[l - 1];
    const int rw = T.w[l], rh = T.h[l], sw = T.w[l - 1], sx0 = T.x[l - 1], sy0 = T.y[l - 1];
    const uint8_t *S = buf[(l - 1) & 1];
    uint8_t *D = buf[l & 1];
    const int2 *sX = sTab + toff, *sY = sX + rw;
    toff += rw + rh;
    uint8_t *plane = P.pyr + (size_t)frame * P.pyr_fs + G.off + (size_t)T.y[l] * G.pitch + T.x[l];
    const float inv_rw = 1.0f / (float)rw;
    for (int idx = tid; idx < rw * rh; idx += CHAIN_NT) {       // idx < 2^14: (idx + 0.5) / rw truncates to the exact quotient (see k_resize)
      const int ry = (int)(((float)idx + 0.5f) * inv_rw), rx = idx - ry * rw;
      const int2 xt = sX[rx], yt = sY[ry];
      const int sx = xt.x, sx1 = min(sx + 1, Gs.w - 1);
      const int sya = min(max(yt.x, 0), Gs.h - 1), syb = min(max(yt.x + 1, 0), Gs.h - 1);
      const int a0 = xt.y & 0xffff, a1 = (xt.y >> 16) & 0xffff, b0 = yt.y & 0xffff, b1 = (yt.y >> 16) & 0xffff;
      const uint8_t *S0 = S + (sya - sy0) * sw - sx0, *S1 = S + (syb - sy0) * sw - sx0;
      const int r0 = S0[sx] * a0 + S0[sx1] * a1;
      const int r1 = S1[sx] * a0 + S1[sx1] * a1;
      const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
      if (l == L) plane[(size_t)ry * G.pitch + rx] = (uint8_t)v;
      else D[idx] = (uint8_t)v;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------
// K2: FAST-9/16 per cell (ORBextractor.cc:787-854 + cv::FAST).
//
// Threshold-free formulation.  For a pixel with centre v and circle c_k let
//   S = max( max_arcs min_k (v - c_k), max_arcs min_k (c_k - v) )   over the 16 arcs of 9 contiguous pixels.
// Then "corner at threshold t"  <=>  S > t, and cornerScore = S - 1 for every corner, independent of t.
// A corner survives cv::FAST's 3x3 NMS iff its S is strictly greater than the S of its 8 neighbours inside the
// same cell (neighbours outside the cell's detection interior count as 0), again independent of t.  So one pass
// gives the keypoints for iniThFAST and, if that set is empty, for minThFAST (decided per cell, after NMS).
//
// One 256-thread workgroup per cell: tile (<=65x65 B) in LDS, score plane in LDS, wave ballots for the
// "any corner at iniTh" vote and for the raster-order compaction.  Output: per-cell slot list, packed
// (response<<24 | y<<12 | x) in detection-rectangle coordinates, raster order inside the cell.
// ------------------------------------------------------------------------------------------------------------
// number of set bits of a wave ballot below this lane (v_mbcnt_lo / v_mbcnt_hi)
__device__ __forceinline__ int lane_rank(unsigned long long b) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
}
typedef short pk16 __attribute__((ext_vector_type(2)));  // two signed 16-bit lanes in one VGPR (v_pk_*_i16)
__device__ __forceinline__ int fast_score_S(const uint8_t *c /* tile centre, pitch FAST_TILE_PITCH */) {
  constexpr int Pt = FAST_TILE_PITCH;
  const int v = c[0];
  // d[k] = (v - c_k, c_k - v) as two signed 16-bit halves: ONE packed min/max network yields the bright-centre margin
  // (low half) and the dark-centre margin (high half).
  const pk16 V2 = {(short)v, (short)-v};
  const pk16 K = {(short)-1, (short)1};
  pk16 d[16];
#define ORB_RING(k, off) { const short cc = (short)c[off]; const pk16 C = {cc, cc}; d[k] = C * K + V2; }
  ORB_RING(0, 3 * Pt + 0)   ORB_RING(1, 3 * Pt + 1)   ORB_RING(2, 2 * Pt + 2)    ORB_RING(3, 1 * Pt + 3)
  ORB_RING(4, 3)            ORB_RING(5, -1 * Pt + 3)  ORB_RING(6, -2 * Pt + 2)   ORB_RING(7, -3 * Pt + 1)
  ORB_RING(8, -3 * Pt)      ORB_RING(9, -3 * Pt - 1)  ORB_RING(10, -2 * Pt - 2)  ORB_RING(11, -1 * Pt - 3)
  ORB_RING(12, -3)          ORB_RING(13, 1 * Pt - 3)  ORB_RING(14, 2 * Pt - 2)   ORB_RING(15, 3 * Pt - 1)
#undef ORB_RING
  // sliding window minimum of length 9 on the circular sequence by doubling: 2, 4, then three windows of 4 (k, k+4, k+5)
  pk16 m2[16], m4[16];
#pragma unroll
  for (int k = 0; k < 16; k++) m2[k] = __builtin_elementwise_min(d[k], d[(k + 1) & 15]);
#pragma unroll
  for (int k = 0; k < 16; k++) m4[k] = __builtin_elementwise_min(m2[k], m2[(k + 2) & 15]);
  pk16 A = {(short)-255, (short)-255};
#pragma unroll
  for (int k = 0; k < 16; k++)
    A = __builtin_elementwise_max(A, __builtin_elementwise_min(__builtin_elementwise_min(m4[k], m4[(k + 4) & 15]), m4[(k + 5) & 15]));   // k .. k+8: the third window overlaps the second, so d and m2 are dead by now (registers)
  const int S = max((int)A.x, (int)A.y);  // all of some arc darker by A.x, or brighter by A.y
  return min(max(S, 0), 255);
}

// Necessary condition for "corner at threshold t" on the 4 compass pixels (k = 0, 4, 8, 12): every arc of 9 contiguous
// circle pixels contains two ADJACENT compass pixels, so two adjacent ones must both be darker than v-t or both
// brighter than v+t.  Pixels that fail it at minThFAST have S <= minThFAST and can never be emitted nor suppress a
// neighbour, so their score is left at 0 and the 16-pixel score is only evaluated for the survivors.
__device__ __forceinline__ int fast_compass_sign(const uint8_t *c, int t) {  // negative <=> the pixel passes
  constexpr int Pt = FAST_TILE_PITCH;
  const int v = c[0];
  const uint16_t p0 = c[3 * Pt], p4 = c[3], p8 = c[-3 * Pt], p12 = c[-3];
  // "two ADJACENT compass pixels both darker than v - t": (dark0 | dark8) & (dark4 | dark12), and dark_a | dark_b is
  // min(a, b) < v - t: the whole dark test is max(min(p0, p8), min(p4, p12)) < v - t, the bright one
  // min(max(p0, p8), max(p4, p12)) > v + t.  16-bit min / max issue at the full rate on gfx950 (the 32-bit ones at half,
  // profiles/valu_calib.json); the two comparisons are sign bits of differences, so the caller's ballot is one compare.
  const int D = (int)(uint16_t)max((uint16_t)min(p0, p8), (uint16_t)min(p4, p12));
  const int B = (int)(uint16_t)min((uint16_t)max(p0, p8), (uint16_t)max(p4, p12));
  return (D - (v - t)) | ((v + t) - B);
}

#ifndef FAST_NT
#define FAST_NT 256
#endif
#define FAST_LIST_SEG 888   // list entries per wavefront: 15 rows x 59 columns (64-lane rows), 16 rows x 32 (32-lane rows)
static_assert(FAST_NT == 256, "k_fast's list segments assume four wavefronts");
// Diagnostic builds (-DFAST_STAMPS, tools/fast_stamps.py): cycles per section, thread 0 of every workgroup.
#if defined(FAST_STAMPS) || defined(OCT_STAMPS)
__device__ unsigned int *g_fast_stamps;  // [workgroup][8] cycle deltas, set by orbx_debug_fast_stamps
#endif
#ifdef FAST_STAMPS
#define FSTAMP(i) do { const long long t_ = __builtin_readcyclecounter(); if (threadIdx.x == 0 && g_fast_stamps) g_fast_stamps[(size_t)blockIdx.x * 8 + (i)] = (unsigned int)(t_ - ft0); ft0 = t_; } while (0)
#else
#define FSTAMP(i) do {} while (0)
#endif
__global__ __launch_bounds__(FAST_NT) void k_fast(FrameParams P) {
#ifdef FAST_STAMPS
  long long ft0 = __builtin_readcyclecounter();
#endif
  __shared__ __align__(16) uint8_t sTile[FAST_TILE_ROWS * FAST_TILE_PITCH];
  __shared__ __align__(16) uint8_t sS[62 * FAST_S_PITCH];
  __shared__ uint16_t sList[4 * FAST_LIST_SEG];
  __shared__ __align__(16) uint32_t sWCount[12];   // [0..3] pass-1 list segments, [4..11] two count buffers of the ordered compactions
  const int tid = threadIdx.x, lane = tid & 63;
  // A workgroup takes a GROUP of up to 2 x 2 neighbouring cells (late round 3; one cell before): their windows overlap by six pixels and
  // the 80-byte tile rows always carried the right-hand neighbour's columns, so one staged tile - one HBM round trip, one set of
  // address arithmetic - now serves four cells, which are then detected one after the other with the per-cell code unchanged (the
  // reference decides iniThFAST / minThFAST per cell, ORBextractor.cc:820-828).  Group records come from orbx_configure.
  int groupId, frame;
  xcd_map(P.totalGroups, P.magicGroups, P.nframes, frame, groupId);
  const uint4 gr = reinterpret_cast<const uint4 *>(P.groups)[groupId];
  const int firstCell = (int)gr.x, gx = (int)(gr.y & 0xffu), gy = (int)((gr.y >> 8) & 0xffu), gStride = (int)(gr.y >> 16);
  const int tileX = (int)(gr.z & 0xffffu), tileY = (int)(gr.z >> 16);
  const int gtw = (int)(gr.w & 0xffu), gth = (int)((gr.w >> 8) & 0xffu), level = (int)((gr.w >> 16) & 0xffu);
  const bool gvalid = (gr.w >> 24) != 0u;
  int pitch = 0;
  const uint8_t *img = nullptr;
  const int ax = tileX & ~3;                          // tile column 0 = image column ax, tile row 0 = image row tileY
  if (gvalid) {
    const uint4 r1f = reinterpret_cast<const uint4 *>(P.cells)[2 * firstCell + 1];
    if (level == 0) { pitch = (int)P.img0_stride; img = P.img0 + (size_t)frame * P.img0_frame_stride; }
    else { pitch = (int)r1f.y; img = P.pyr + (size_t)frame * P.pyr_fs + (((size_t)r1f.w << 32) | r1f.z); }
    // ---- tile -> LDS.  Aligned path: the columns [tileX & ~3, ...) of the group's rows.
    const bool aligned = ((((uintptr_t)img) | (uintptr_t)pitch) & 3u) == 0;
    if (aligned) {
      // LDS-DMA: one global_load_lds_dwordx4 per lane moves 16 bytes straight into the tile (no register, no ds_write); the
      // tile is lane-linear, chunk idx = 5 * row + chunk-in-row.  Chunks past the group's columns read the bytes that follow
      // in the plane (a cell window ends >= 13 rows above the plane's last row, so they exist) and are never looked at.
      // Index arithmetic in 24-bit multiplies (full rate): idx / 5 as (idx * 13108) >> 16, exact below 5 * FAST_TILE_ROWS.
      const int nch = 5 * gth;
      const uint8_t *img0 = img + mul24((uint32_t)tileY, (uint32_t)pitch) + ax;
      for (int idx = tid; idx < nch; idx += FAST_NT) {
        const uint32_t r = mul24((uint32_t)idx, 13108u) >> 16, c = (uint32_t)idx - 5u * r;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(img0 + (mul24(r, (uint32_t)pitch) + 16u * c)),
                                         (__attribute__((address_space(3))) void *)&sTile[idx * 16], 16, 0, 0);
      }
    } else {
      const uint32_t magic = (1u << 20) / (uint32_t)gtw + 1u;
      const int ox0 = tileX - ax;
      const uint8_t *img0 = img + mul24((uint32_t)tileY, (uint32_t)pitch) + tileX;
      for (int idx = tid; idx < gtw * gth; idx += FAST_NT) {
        const uint32_t r = mul24((uint32_t)idx, magic) >> 20, cc = (uint32_t)idx - mul24(r, (uint32_t)gtw);
        sTile[r * FAST_TILE_PITCH + ox0 + cc] = img0[mul24(r, (uint32_t)pitch) + cc];
      }
    }
  }
  auto do_cell = [&](const int cellId) {
  // cell record (orbx_configure, ORBextractor.cc:787-803): one 32-byte scalar load instead of a level search plus
  // a dozen dependent geometry loads per workgroup
  const uint4 r0 = reinterpret_cast<const uint4 *>(P.cells)[2 * cellId], r1 = reinterpret_cast<const uint4 *>(P.cells)[2 * cellId + 1];
  const int iniX = (int)(r0.x & 0xffffu), iniY = (int)(r0.x >> 16);
  const int tw = (int)(r0.y & 0xffu), th = (int)((r0.y >> 8) & 0xffu), cw = tw - 6, ch = th - 6;
  const uint32_t baseX = r0.z & 0xffffu, baseY = r0.z >> 16;  // cj * wCell, ci * hCell
  const uint32_t cellCap = r1.x;
  uint32_t *cellCnt = P.cellCnt + (size_t)frame * P.cell_fs + cellId;
  if (!(r0.y >> 24)) {  // cell outside the detection rectangle
    if (tid == 0) *cellCnt = 0;
    return;
  }
  const int ox = iniX - ax;                                          // offset of the cell's first column inside the tile
  const uint8_t *sT = sTile + (iniY - tileY) * FAST_TILE_PITCH;      // the cell's first row inside the tile
  // score plane rows 0 .. ch+1 to zero: (ch + 2) * 64 bytes = at most 244 sixteen-byte stores, one per thread, no loop
  static_assert(62 * FAST_S_PITCH / 16 <= FAST_NT, "one 16-byte store per thread must cover the score plane");
  if (tid < ((ch + 2) * FAST_S_PITCH) / 16) reinterpret_cast<uint4 *>(sS)[tid] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  FSTAMP(0);
  // The reference runs cv::FAST on the cell at iniThFAST and, only if that returns nothing, again at minThFAST
  // (ORBextractor.cc:820-828).  The score S is threshold-free, so "FAST at threshold t with non-maximum suppression" is: S > t
  // and S strictly above the S of the 8 neighbours (a neighbour that is no corner at t has S <= t < S and can be left at 0).
  // The kernel follows the same order - detect at iniThFAST, and again at minThFAST when the cell stayed empty - because the
  // compass pre-test at the higher threshold passes far fewer pixels to the 16-pixel score (synthetic EuRoC frames: 81-125 per
  // 30x30 cell instead of 227-360 at minThFAST 7, and 2 % of the cells need the second detection).
  const int xsh = cw <= 32 ? 5 : 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction; tells the compiler so (scalar loop bounds)
  const int px = lane & ((1 << xsh) - 1), sub = lane >> xsh, rpi = 64 >> xsh;   // rows per wave-instruction: 2 or 1
  const int R = (ch + 3) >> 2;
  const int rowBeg = wave * R, rowEnd = min(rowBeg + R, ch);
  const bool xin = px < cw;
  uint16_t *myList = sList + wave * FAST_LIST_SEG;
  uint32_t *sCnt2 = sWCount + 4;   // [2][4]
  uint32_t *slots = P.slots + (size_t)frame * P.slot_fs + r0.w;
  int turn = 0, nkept = 0;
  int t = P.iniTh;
  for (int detection = 0;; detection++) {
    // ---- pass 1: compass pre-test at t, survivors go to a dense work list as y << 6 | x (raster order = numeric order).
    // Threads form rows of 32 (cells up to 32 interior columns, the usual 30) or 64 lanes and step down the cell, so the
    // per-pixel index arithmetic is one add; a linear pixel index would need a division per pixel.
    // Wavefront w owns the CONTIGUOUS rows [w * R, (w + 1) * R), R = ceil(ch / 4): its list segment is in raster order and so
    // is the concatenation of the four segments - every later pass can keep that order with ballots and never has to sort.
    // Every wavefront appends to its own segment of the list (at most R rows x cw entries <= 15 x 59 <= FAST_LIST_SEG) and
    // keeps its count in a scalar register: no atomics.
    int wcount = 0;
    {
      const uint8_t *c = &sT[(rowBeg + sub + 3) * FAST_TILE_PITCH + ox + px + 3];
      int yv = ((rowBeg + sub) << 6) | px;
      // validity as a sign bit: row - rowEnd is negative inside the wavefront's rows; columns outside the cell start from a value
      // that stays positive.  The pass condition is then ONE integer compare, which is also the ballot (no select / re-compare).
      int yr = xin ? rowBeg + sub - rowEnd : 0x40000000;
      for (int y0 = rowBeg; y0 < rowEnd; y0 += rpi, c += rpi * FAST_TILE_PITCH, yv += rpi << 6, yr += rpi) {
        // lanes outside the cell read LDS bytes that mean nothing (or zero past the allocation) and are masked out by yr
#ifdef FAST_PAD   // diagnostic builds only (tools/fast_sensitivity.sh): 16 extra instructions of one class per pass-1 trip, results unused
        {
          int pad0 = yv, pad1 = yr;
#if FAST_PAD == 1   // full-rate VALU
          asm volatile("v_add_u32 %0, %0, %1\nv_add_u32 %1, %1, %0\nv_add_u32 %0, %0, %1\nv_add_u32 %1, %1, %0\nv_add_u32 %0, %0, %1\nv_add_u32 %1, %1, %0\nv_add_u32 %0, %0, %1\nv_add_u32 %1, %1, %0\n"
                       "v_add_u32 %0, %0, %1\nv_add_u32 %1, %1, %0\nv_add_u32 %0, %0, %1\nv_add_u32 %1, %1, %0\nv_add_u32 %0, %0, %1\nv_add_u32 %1, %1, %0\nv_add_u32 %0, %0, %1\nv_add_u32 %1, %1, %0\n" : "+v"(pad0), "+v"(pad1));
#elif FAST_PAD == 2   // half-rate VALU
          asm volatile("v_pk_max_i16 %0, %0, %1\nv_pk_max_i16 %1, %1, %0\nv_pk_max_i16 %0, %0, %1\nv_pk_max_i16 %1, %1, %0\nv_pk_max_i16 %0, %0, %1\nv_pk_max_i16 %1, %1, %0\nv_pk_max_i16 %0, %0, %1\nv_pk_max_i16 %1, %1, %0\n"
                       "v_pk_max_i16 %0, %0, %1\nv_pk_max_i16 %1, %1, %0\nv_pk_max_i16 %0, %0, %1\nv_pk_max_i16 %1, %1, %0\nv_pk_max_i16 %0, %0, %1\nv_pk_max_i16 %1, %1, %0\nv_pk_max_i16 %0, %0, %1\nv_pk_max_i16 %1, %1, %0\n" : "+v"(pad0), "+v"(pad1));
#elif FAST_PAD == 3   // scalar ALU
          asm volatile("s_add_u32 s40, s40, 1\ns_add_u32 s41, s41, 1\ns_add_u32 s42, s42, 1\ns_add_u32 s43, s43, 1\ns_add_u32 s40, s40, 1\ns_add_u32 s41, s41, 1\ns_add_u32 s42, s42, 1\ns_add_u32 s43, s43, 1\n"
                       "s_add_u32 s40, s40, 1\ns_add_u32 s41, s41, 1\ns_add_u32 s42, s42, 1\ns_add_u32 s43, s43, 1\ns_add_u32 s40, s40, 1\ns_add_u32 s41, s41, 1\ns_add_u32 s42, s42, 1\ns_add_u32 s43, s43, 1\n" ::: "s40", "s41", "s42", "s43", "scc");
#endif
          asm volatile("" ::"v"(pad0), "v"(pad1));
        }
#endif
        const bool pass = (fast_compass_sign(c, t) & yr) < 0;
        const unsigned long long b = __builtin_amdgcn_ballot_w64(pass);
        if (pass) myList[wcount + lane_rank(b)] = (uint16_t)yv;
        wcount += __popcll(b);
      }
    }
    if (lane == 0) sWCount[wave] = (uint32_t)wcount;
    __syncthreads();
    FSTAMP(1);
    // The four segments are walked as one list by all 256 threads (a wavefront walking only its own segment would need a
    // second trip whenever that segment alone exceeds 64 entries): entry e lives in segment #(prefix sums <= e).
    const uint4 wc = *reinterpret_cast<const uint4 *>(sWCount);
    const int pre1 = (int)wc.x, pre2 = pre1 + (int)wc.y, pre3 = pre2 + (int)wc.z, nlist = pre3 + (int)wc.w;
    auto entry = [&](int e) -> int {
      const int seg = (e >= pre1) + (e >= pre2) + (e >= pre3);
      const int start = e >= pre2 ? (e >= pre3 ? pre3 : pre2) : (e >= pre1 ? pre1 : 0);
      return sList[seg * FAST_LIST_SEG + (e - start)];
    };
    // ---- pass 2: full 16-pixel score for the survivors only (a second detection recomputes the first one's entries: same
    // values, the plane needs no clearing in between)
    for (int e = tid; e < nlist; e += FAST_NT) {
      const int p = entry(e);
      const int y = p >> 6, x = p & 63;
      const int S = fast_score_S(&sT[(y + 3) * FAST_TILE_PITCH + ox + x + 3]);
      sS[(y + 1) * FAST_S_PITCH + x + 1] = (uint8_t)S;
    }
    __syncthreads();
    FSTAMP(2);
    // ---- pass 3 (survivors only): S > t and 3x3 strict maximum inside the cell; cv::FAST emits rows ascending, x ascending = the
    // list's own order, so a kept entry's output slot is its rank among the kept ones and it is written on the spot.
    // Non-survivors have score 0 in the plane, exactly what cv::FAST's NMS sees for non-corners.  Order-preserving ranks of
    // each 256-entry chunk: a ballot + lane rank inside a wavefront, the four wavefronts' counts through LDS - two count
    // buffers in turn, so one barrier per chunk.  No atomics, no sorting afterwards, no kept list.
    for (int e0 = 0; e0 < nlist; e0 += FAST_NT) {
      const int e = e0 + tid;
      bool keep = false;
      uint32_t kv = 0;
      if (e < nlist) {
        const int p = entry(e);
        const int y = p >> 6, x = p & 63;
        const uint8_t *s = &sS[(y + 1) * FAST_S_PITCH + x + 1];
        const int S = s[0];
        const int m0 = max(max((int)s[-FAST_S_PITCH - 1], (int)s[-FAST_S_PITCH]), (int)s[-FAST_S_PITCH + 1]);
        const int m1 = max(max((int)s[FAST_S_PITCH - 1], (int)s[FAST_S_PITCH]), (int)s[FAST_S_PITCH + 1]);
        const int m2 = max(max((int)s[-1], (int)s[1]), max(m0, m1));
        keep = (S > t) & (S >= 2) & (S > m2);
        kv = ((uint32_t)(S - 1) << 24) | ((baseY + (uint32_t)y + 3u) << 12) | (baseX + (uint32_t)x + 3u);
      }
      const unsigned long long bK = __builtin_amdgcn_ballot_w64(keep);
      uint32_t *cnt = sCnt2 + 4 * turn;
      turn ^= 1;
      if (lane == 0) cnt[wave] = (uint32_t)__popcll(bK);
      __syncthreads();
      const uint4 c4 = *reinterpret_cast<const uint4 *>(cnt);
      const uint32_t pre = (wave > 0 ? c4.x : 0u) + (wave > 1 ? c4.y : 0u) + (wave > 2 ? c4.z : 0u);
      const uint32_t rank = (uint32_t)nkept + pre + (uint32_t)lane_rank(bK);
      if (keep && rank < cellCap) slots[rank] = kv;
      nkept += (int)(c4.x + c4.y + c4.z + c4.w);
    }
    FSTAMP(3);
    // ORBextractor.cc:825: the second detection runs only if the first returned nothing (with minThFAST >= iniThFAST it could
    // only return a subset of nothing)
    if (nkept > 0 || detection == 1 || P.minTh >= P.iniTh) break;
    t = P.minTh;
  }
  if (tid == 0) *cellCnt = (uint32_t)min(nkept, (int)cellCap);
  FSTAMP(7);
  };
  for (int cy = 0; cy < gy; cy++)
    for (int cx = 0; cx < gx; cx++) {
      if (cy | cx) __syncthreads();      // the previous cell is done with the score plane, the lists and the counters
      do_cell(firstCell + cy * gStride + cx);
    }
}

// ------------------------------------------------------------------------------------------------------------
// K3: DistributeOctTree (ORBextractor.cc:479-761), one 256-thread workgroup per (frame, level).
//
// Data-parallel restatement (model + proof-by-test in tests/octree_model.py):
//  * the std::list is an array in list order; a round builds the next array with prefix sums:
//      new list = [children of the nodes processed this round, in reverse processing order, each as n4,n3,n2,n1]
//                 ++ [unprocessed nodes in their old order]            (push_front / erase semantics, :619-661)
//  * keys never move: each candidate carries the list position of its node (knode) and is re-labelled per round;
//  * full rounds process every node with >1 key in list order (:598-663); once size + 3*nToExpand > N (:671)
//    rounds process nodes by (size desc, creation order desc) = (size desc, list position asc) and stop at the
//    first prefix that reaches N nodes (:682-729) -- the cut is found with a scan over the ranked child counts;
//  * the winner of a node is max response, first in vKeys order on ties (:745-757) = smallest dense index,
//    taken with a 64-bit LDS atomicMax over (response, ~index).
// Candidates are first compacted from the per-cell slot lists into the reference's vToDistributeKeys order.
// ------------------------------------------------------------------------------------------------------------
#ifdef OCT_STAMPS   // diagnostic builds only (tools/octree_stamps.py): cycles per phase, thread 0 of every workgroup, summed over the rounds
#define OSTAMP(i) do { const long long t_ = __builtin_readcyclecounter(); if (threadIdx.x == 0 && g_fast_stamps) g_fast_stamps[(size_t)blockIdx.x * 8 + (i)] += (unsigned int)(t_ - ot0); ot0 = t_; } while (0)
#else
#define OSTAMP(i) do {} while (0)
#endif
struct OctNode { short ulx, urx, uly, bry; };

#ifndef OCT_REGKEYS
#define OCT_REGKEYS 4096   // keys of a level held in registers (the rest goes through the global cand / knode arrays)
#endif
template <int NT, bool CELLS_LDS>
__global__ __launch_bounds__(NT) void k_octree(FrameParams P, uint32_t *cellOffScratch) {
  extern __shared__ __align__(16) uint8_t smem[];
#ifdef OCT_STAMPS
  long long ot0 = __builtin_readcyclecounter();
#endif
  const int CAP = P.octCap;
  // carve (all offsets multiples of 8)
  unsigned long long *best = reinterpret_cast<unsigned long long *>(smem);          // aliases chcnt/chpos
  uint32_t *chcnt = reinterpret_cast<uint32_t *>(smem);                               // 4*CAP
  int32_t *chpos = reinterpret_cast<int32_t *>(smem + 16 * (size_t)CAP);              // 4*CAP
  OctNode *nodeA = reinterpret_cast<OctNode *>(smem + 32 * (size_t)CAP);
  OctNode *nodeB = nodeA + CAP;
  uint32_t *cntA = reinterpret_cast<uint32_t *>(nodeB + CAP);
  uint32_t *cntB = cntA + CAP;
  int32_t *npos = reinterpret_cast<int32_t *>(cntB + CAP);
  int32_t *rankOf = npos + CAP;
  uint32_t *incl = reinterpret_cast<uint32_t *>(rankOf + CAP);  // child counts by rank -> inclusive sums
  uint32_t *sflag = incl + CAP;                                  // unprocessed flags -> exclusive sums
  uint32_t *sw = sflag + CAP;                                    // scan scratch (NT/64+2)
  __shared__ int shI[8];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  constexpr int NW = NT / 64;
  // 1-D grid, level-major: workgroups are dealt round-robin over the 8 XCDs by block index, so frame f of every level
  // lands on XCD f % 8 (a (level, frame) 2-D grid put ALL level-0 workgroups, the long ones, on one XCD: 2.2x slower),
  // and the long level-0 workgroups are dispatched first.
  const int level = (int)(blockIdx.x / (unsigned)P.nframes), frame = (int)(blockIdx.x - (unsigned)level * (unsigned)P.nframes);
  const LevelGeom G = P.geom[level];
  int32_t *lcnt = P.lcnt + ((size_t)frame * P.nlevels + level) * 2;
  uint32_t *cand = P.cand + (size_t)frame * P.cand_fs + G.candBase;
  uint16_t *knode = P.knode + (size_t)frame * P.cand_fs + G.candBase;
  // The keys never move and every pass over them uses the same thread <-> key assignment (k = tid + j * NT), so the first 4096 keys
  // of a level - all of them on any frame seen so far (EuRoC: 600 .. 3300 per level) - live in REGISTERS: the packed candidate and its
  // node label, PER of each per thread.  The rounds then touch no memory but the node tables in LDS; the global cand / knode arrays
  // only serve keys beyond 4096 (round 2 re-read and re-wrote both arrays every round: 4.8x the kernel's algorithmic traffic, and a
  // global round trip in front of both key passes of every round).
  constexpr int PER = OCT_REGKEYS / NT, REGN = PER * NT;
  uint32_t rc[PER > 0 ? PER : 1], rk[PER > 0 ? PER : 1];
#pragma unroll
  for (int j = 0; j < PER; j++) { rc[j] = 0u; rk[j] = 0u; }
  // f(k, c, nd): k dense key index, c packed candidate, nd node label (both by reference)
  auto for_keys = [&](int n_, auto f) {
#pragma unroll
    for (int j = 0; j < PER; j++) {
      if (j * NT >= n_) break;
      const int k = tid + j * NT;
      if (k < n_) f(k, rc[j], rk[j]);
    }
    for (int k = REGN + tid; k < n_; k += NT) {
      uint32_t c = cand[k], nd = knode[k];
      f(k, c, nd);
      knode[k] = (uint16_t)nd;
    }
  };

  // ---- A. compact the per-cell slot lists into vToDistributeKeys order (cells row-major, raster inside) ----
  // Cell offsets by a workgroup scan (kept in LDS when they fit: CELLS_LDS), then one thread per candidate finds its
  // cell by bisection and copies its slot: every global access of this phase is independent of the others, so the
  // phase costs a few memory latencies instead of one per cell.
  const int ncell = G.nCols * G.nRows;
  const uint32_t *cellCnt = P.cellCnt + (size_t)frame * P.cell_fs + G.cellBase;
  uint32_t *cellOffG = cellOffScratch + (size_t)frame * P.cell_fs + G.cellBase;
  uint32_t *cellOffL = sw + (NT / 64 + 2);          // ncell + 1 words behind the scan scratch (CELLS_LDS only)
  uint32_t carry = 0;
  for (int base = 0; base < ncell; base += NT) {
    int i = base + tid;
    uint32_t v = i < ncell ? cellCnt[i] : 0u;
    uint32_t inc = wave_incl_scan(v);
    if (lane == 63) sw[wid] = inc;
    __syncthreads();
    const uint32_t wsum = lane < NW ? sw[lane] : 0u;       // wave totals scanned in registers, as in lds_excl_scan
    const uint32_t wincl = wave_incl_scan(wsum);
    const uint32_t woff = (uint32_t)__builtin_amdgcn_readlane((int)(wincl - wsum), __builtin_amdgcn_readfirstlane(wid));
    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)wincl, NW - 1);
    if (i < ncell) { if (CELLS_LDS) cellOffL[i] = carry + woff + inc - v; else cellOffG[i] = carry + woff + inc - v; }
    carry += tot;
    __syncthreads();
  }
  const int n = min((int)carry, G.candCap);
  if (tid == 0) P.candCnt[(size_t)frame * P.nlevels + level] = n;
  if (n == 0) {
    if (tid == 0) { lcnt[0] = 0; lcnt[1] = 0; }
    return;
  }
  __syncthreads();  // cell offsets visible to the whole workgroup
  {
    const uint32_t *slots = P.slots + (size_t)frame * P.slot_fs + G.slotBase;
    if (CELLS_LDS) {
      auto fetch = [&](int k) -> uint32_t {
        int lo = 0, hi = ncell - 1;                  // last cell whose offset is <= k (empty cells share offsets)
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (cellOffL[mid] <= (uint32_t)k) lo = mid; else hi = mid - 1;
        }
        return slots[(size_t)lo * G.cellCap + ((uint32_t)k - cellOffL[lo])];
      };
#pragma unroll
      for (int j = 0; j < PER; j++) {
        if (j * NT >= n) break;
        const int k = tid + j * NT;
        if (k < n) { rc[j] = fetch(k); cand[k] = rc[j]; }   // the global copy is written once and never read back (orbx_debug_candidates, stage parity tests)
      }
      for (int k = REGN + tid; k < n; k += NT) cand[k] = fetch(k);
      __syncthreads();
    } else {
      for (int cidx = wid; cidx < ncell; cidx += NW) {
        uint32_t cnt = cellCnt[cidx], off = cellOffG[cidx];
        for (uint32_t j = lane; j < cnt; j += 64)
          if (off + j < (uint32_t)n) cand[off + j] = slots[(size_t)cidx * G.cellCap + j];
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < PER; j++) {
        if (j * NT >= n) break;
        const int k = tid + j * NT;
        if (k < n) rc[j] = cand[k];
      }
    }
  }
  OSTAMP(0);

  // ---- B. root nodes (ORBextractor.cc:541-585) ----
  const int nIni = G.nIni;
  const float hX = G.hX;
  const int Hrect = G.maxBorderY - ORB_MIN_BORDER;
  for (int i = tid; i < nIni; i += NT) chcnt[i] = 0;
  __syncthreads();
  for_keys(n, [&](int, uint32_t &c, uint32_t &nd) {
    int r = (int)((float)(c & 0xfff) / hX);  // vpIniNodes[kp.pt.x/hX], :567
    r = min(max(r, 0), nIni - 1);
    nd = (uint32_t)r;
    atomicAdd(&chcnt[r], 1u);
  });
  __syncthreads();
  if (tid == 0) {
    int L = 0;
    for (int r = 0; r < nIni; r++) {
      uint32_t cn = chcnt[r];
      if (cn > 0) {
        OctNode nd;
        nd.ulx = (short)(int)(hX * (float)r);
        nd.urx = (short)(int)(hX * (float)(r + 1));
        nd.uly = 0;
        nd.bry = (short)Hrect;
        nodeA[L] = nd;
        cntA[L] = cn;
        chpos[r] = L;
        L++;
      }
    }
    shI[0] = L;
  }
  __syncthreads();
  for_keys(n, [&](int, uint32_t &, uint32_t &nd) { nd = (uint32_t)chpos[nd]; });
  int L = shI[0];
  __syncthreads();
  OSTAMP(1);

  // ---- C. rounds ----
  const int N = G.N;
  bool careful = false;
  OctNode *nodes = nodeA, *nodesN = nodeB;
  uint32_t *cnts = cntA, *cntsN = cntB;
  for (int guard = 0; guard < 64; guard++) {
    const int prevSize = L;
    for (int i = tid; i < 4 * L; i += NT) chcnt[i] = 0;
    if (tid == 0) { shI[1] = 0; shI[2] = 0x7fffffff; }
    __syncthreads();
    // child key counts of every expandable node
    for_keys(n, [&](int, uint32_t &c, uint32_t &nd) {
      if (cnts[nd] > 1) {
        OctNode o = nodes[nd];
        int x = c & 0xfff, y = (c >> 12) & 0xfff;
        int hx = (o.urx - o.ulx + 1) >> 1, hy = (o.bry - o.uly + 1) >> 1;  // ceil(/2), :481-482
        int q = (x < o.ulx + hx ? 0 : 1) + (y < o.uly + hy ? 0 : 2);
        atomicAdd(&chcnt[nd * 4 + q], 1u);
      }
    });
    __syncthreads();
    OSTAMP(2);
    // A full round (every node with more than one key is split, in list order: rank order = list order) needs three prefix sums
    // over the list - expandable nodes before me, their children before me, unprocessed nodes before me - and gets them from ONE
    // scan of packed counters (10 + 12 + 10 bits: lists below 1024 nodes); careful rounds, which rank by size, keep three scans.
    int mstar, Eproc, Lnew;
    const bool fusedScan = !careful && L < 1024;
    if (fusedScan) {
      for (int i = tid; i < L; i += NT) {
        const bool e = cnts[i] > 1;
        const uint32_t c = e ? (chcnt[i * 4] > 0) + (chcnt[i * 4 + 1] > 0) + (chcnt[i * 4 + 2] > 0) + (chcnt[i * 4 + 3] > 0) : 0u;
        sflag[i] = (e ? 1u : 0u) | (c << 10) | ((e ? 0u : 1u) << 22);
      }
      __syncthreads();
      const uint32_t tot = lds_excl_scan<NT>(sflag, L, sw);
      mstar = (int)(tot & 1023u) - 1;
      Eproc = (int)((tot >> 10) & 4095u);
      Lnew = Eproc + (int)(tot >> 22);
    } else {
    // rank of the expandable nodes in processing order
    if (!careful) {
      for (int i = tid; i < L; i += NT) sflag[i] = cnts[i] > 1 ? 1u : 0u;
      __syncthreads();
      uint32_t nX = lds_excl_scan<NT>(sflag, L, sw);
      for (int i = tid; i < L; i += NT) rankOf[i] = cnts[i] > 1 ? (int)sflag[i] : -1;
      if (tid == 0) shI[3] = (int)nX;
    } else {
      // rank by (size descending, list position ascending) = the number of expandable nodes that go first: four adjacent lanes
      // per node, each counting over a quarter of the list, summed across the quad (DPP)
      uint32_t myX = 0;
      for (int i0 = 0; i0 < L; i0 += NT / 4) {
        const int i = i0 + (tid >> 2), part = tid & 3;
        const uint32_t ci = i < L ? cnts[i] : 0u;
        int r = 0;
        if (ci > 1)
          for (int j = part; j < L; j += 4) {
            const uint32_t cj = cnts[j];
            r += (cj > 1 && (cj > ci || (cj == ci && j < i))) ? 1 : 0;
          }
        r += __builtin_amdgcn_update_dpp(0, r, 0xb1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
        r += __builtin_amdgcn_update_dpp(0, r, 0x4e, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
        if (i < L && part == 0) {
          rankOf[i] = ci > 1 ? r : -1;
          if (ci > 1) myX++;
        }
      }
      myX = (uint32_t)wave_sum_i32((int)myX);
      if (lane == 0 && myX) atomicAdd(&shI[1], (int)myX);
      __syncthreads();
      if (tid == 0) { shI[3] = shI[1]; shI[1] = 0; }
    }
    __syncthreads();
    const int nX = shI[3];
    // child counts in rank order -> inclusive sums
    for (int i = tid; i < L; i += NT) {
      int r = rankOf[i];
      if (r >= 0) {
        uint32_t c = (chcnt[i * 4] > 0) + (chcnt[i * 4 + 1] > 0) + (chcnt[i * 4 + 2] > 0) + (chcnt[i * 4 + 3] > 0);
        incl[r] = c;
      }
    }
    __syncthreads();
    lds_excl_scan<NT>(incl, nX, sw);  // exclusive; inclusive = excl + c, recomputed below
    // cut of a careful round: first rank r with L + incl(r) - (r+1) >= N  (:728)
    if (careful) {
      for (int i = tid; i < L; i += NT) {
        int r = rankOf[i];
        if (r >= 0) {
          uint32_t c = (chcnt[i * 4] > 0) + (chcnt[i * 4 + 1] > 0) + (chcnt[i * 4 + 2] > 0) + (chcnt[i * 4 + 3] > 0);
          int after = L + (int)(incl[r] + c) - (r + 1);
          if (after >= N) atomicMin(&shI[2], r);
        }
      }
    }
    __syncthreads();
    mstar = nX - 1;
    if (careful && shI[2] != 0x7fffffff) mstar = shI[2];
    // Eproc = inclusive sum at mstar; unprocessed flags
    for (int i = tid; i < L; i += NT) {
      int r = rankOf[i];
      bool proc = r >= 0 && r <= mstar;
      sflag[i] = proc ? 0u : 1u;
      if (r == mstar && r >= 0) {
        uint32_t c = (chcnt[i * 4] > 0) + (chcnt[i * 4 + 1] > 0) + (chcnt[i * 4 + 2] > 0) + (chcnt[i * 4 + 3] > 0);
        shI[4] = (int)(incl[r] + c);
      }
    }
    if (tid == 0 && nX == 0) shI[4] = 0;
    __syncthreads();
    Eproc = shI[4];
    uint32_t nUnproc = lds_excl_scan<NT>(sflag, L, sw);
    Lnew = Eproc + (int)nUnproc;
    }
    OSTAMP(3);
    // build the next list
    for (int i = tid; i < L; i += NT) {
      OctNode o = nodes[i];
      bool proc;
      uint32_t childrenBefore, unprocBefore;
      if (fusedScan) {
        const uint32_t pk = sflag[i];
        proc = cnts[i] > 1; childrenBefore = (pk >> 10) & 4095u; unprocBefore = pk >> 22;
      } else {
        const int r = rankOf[i];
        proc = r >= 0 && r <= mstar; childrenBefore = proc ? incl[r] : 0u; unprocBefore = sflag[i];
      }
      if (proc) {
        uint32_t c0 = chcnt[i * 4], c1 = chcnt[i * 4 + 1], c2 = chcnt[i * 4 + 2], c3 = chcnt[i * 4 + 3];
        uint32_t c = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
        int pos = Eproc - (int)(childrenBefore + c);  // block start; inside the block n4,n3,n2,n1
        int hx = (o.urx - o.ulx + 1) >> 1, hy = (o.bry - o.uly + 1) >> 1;
        int nexp = 0;
        uint32_t cc[4] = {c0, c1, c2, c3};
#pragma unroll
        for (int q = 3; q >= 0; q--) {
          if (cc[q] > 0) {
            OctNode ch;
            ch.ulx = (short)((q & 1) ? o.ulx + hx : o.ulx);
            ch.urx = (short)((q & 1) ? o.urx : o.ulx + hx);
            ch.uly = (short)((q & 2) ? o.uly + hy : o.uly);
            ch.bry = (short)((q & 2) ? o.bry : o.uly + hy);
            if (pos < CAP) { nodesN[pos] = ch; cntsN[pos] = cc[q]; }
            chpos[i * 4 + q] = pos;
            nexp += cc[q] > 1 ? 1 : 0;
            pos++;
          } else {
            chpos[i * 4 + q] = -1;
          }
        }
        if (nexp) atomicAdd(&shI[1], nexp);
        npos[i] = -1;
      } else {
        int pos = Eproc + (int)unprocBefore;
        if (pos < CAP) { nodesN[pos] = o; cntsN[pos] = cnts[i]; }
        npos[i] = pos;
      }
    }
    __syncthreads();
    OSTAMP(4);
    // re-label the keys
    for_keys(n, [&](int, uint32_t &c, uint32_t &nd) {
      int np = npos[nd];
      if (np < 0) {
        OctNode o = nodes[nd];
        int x = c & 0xfff, y = (c >> 12) & 0xfff;
        int hx = (o.urx - o.ulx + 1) >> 1, hy = (o.bry - o.uly + 1) >> 1;
        int q = (x < o.ulx + hx ? 0 : 1) + (y < o.uly + hy ? 0 : 2);
        np = chpos[nd * 4 + q];
      }
      nd = (uint32_t)(uint16_t)np;
    });
    const int nToExpand = shI[1];
    __syncthreads();
    OSTAMP(5);
    { OctNode *t = nodes; nodes = nodesN; nodesN = t; }
    { uint32_t *t = cnts; cnts = cntsN; cntsN = t; }
    L = min(Lnew, CAP);
    if (Lnew >= N || Lnew == prevSize) break;                      // :667, :731
    if (!careful && Lnew + nToExpand * 3 > N) careful = true;      // :671
  }

  // ---- D. best key per node, list order out, lapping ranks (ORBextractor.cc:742-758, :1169-1178) ----
  for (int i = tid; i < L; i += NT) best[i] = 0ull;
  __syncthreads();
  for_keys(n, [&](int k, uint32_t &c, uint32_t &nd) {
    const unsigned long long key = ((unsigned long long)(c >> 24) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)k);
    atomicMax(&best[nd], key);
  });
  __syncthreads();
  uint32_t *lkp = P.lkp + (size_t)frame * P.lkp_fs + G.kpBase;
  uint16_t *lrank = P.lrank + (size_t)frame * P.lkp_fs + G.kpBase;
  const int Lout = min(L, G.kpCap);
  // every node holds at least one key and exactly one key equals its maximum: that key's thread writes the node's keypoint
  for_keys(n, [&](int k, uint32_t &c, uint32_t &nd) {
    const unsigned long long key = ((unsigned long long)(c >> 24) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)k);
    if ((int)nd < Lout && best[nd] == key) {
      lkp[nd] = c;
      float xs = (float)((int)(c & 0xfff) + ORB_MIN_BORDER);
      if (level != 0) xs = xs * G.scale;  // keypoint->pt *= scale, :1164-1166
      const bool lap = xs >= (float)P.lap0 && xs <= (float)P.lap1;
      sflag[nd] = lap ? 1u : 0u;
    }
  });
  __syncthreads();
  // sflag was read as packed flags; keep a copy of the flag in incl before the scan overwrites it
  for (int i = tid; i < Lout; i += NT) incl[i] = sflag[i];
  __syncthreads();
  uint32_t nLap = lds_excl_scan<NT>(sflag, Lout, sw);
  for (int i = tid; i < Lout; i += NT) {
    uint32_t f = incl[i];
    uint32_t rank = f ? sflag[i] : (uint32_t)i - sflag[i];
    lrank[i] = (uint16_t)((f << 15) | (rank & 0x7fff));
  }
  if (tid == 0) { lcnt[0] = Lout; lcnt[1] = (int)nLap; }
  OSTAMP(6);
}

// ------------------------------------------------------------------------------------------------------------
// K5: cv::GaussianBlur 7x7 sigma 2 BORDER_REFLECT_101, 8U fixed point: taps {18,34,49,55,49,34,18} - on the matrix pipe.
// The 8-bit GaussianBlur is an exact integer computation - row sums of at most 257 * 255 (16 bits, no rounding), then
// (sum of 7 weighted row sums + 32768) >> 16 - i.e. two products with banded constant matrices:
//     T = I H      (H[k][x] = tap[k - x - 13]: 64 input columns -> 32 output columns)
//     O = V T      (V[y][k] = tap[k - y]:      64 rows of T     -> 32 output rows)
// v_mfma_i32_32x32x32_i8 takes SIGNED bytes, so pixels enter as I - 128 (one xor per dword) with the correction 128 * 257 in the
// accumulator seed, and the 16-bit row sums are split into their high and low byte (again minus 128) for the second product:
//     sum V T = 256 (sum V hi' ) + (sum V lo') + 257 * 128 * 257.
// The accumulator of the first product (row index in the registers, column on the lane) IS the B operand of the second one - the
// guide's "accumulator tile as the next MFMA's operand": element j of lane half h is row (j & 3) + 8 (j >> 2) + 4 h, and V's
// fragments are laid out in that order - so nothing moves between the passes but the byte split.  Per 32 x 32 outputs: 4 + 4
// MFMAs and about 130 vector instructions per wavefront for the arithmetic (the v_dot4 / v_dot2 form of rounds 1-2: 220), and with
// the border tiles loaded by LDS-DMA like the interior ones (half of all tiles touch an edge; their byte-wise loops were a third
// of the old kernel's instructions) 0.247 -> 0.172 ms per 256 frames, byte for byte the same planes.
// Workgroup = the same 128 x 32 tile as before, wavefront w its columns 32 w .. 32 w + 31; the results go through a 4 KB LDS
// image so that the stores are 16-byte row segments.
// ------------------------------------------------------------------------------------------------------------
// per lane (column / row n = lane & 31, half h = lane >> 5): H fragments of K-steps 0, 1, then V fragments of K-steps 0, 1
struct BlurTab { uint32_t v[64][16]; };
constexpr uint32_t blur_tap(int d) { return d == 0 || d == 6 ? 18u : d == 1 || d == 5 ? 34u : d == 2 || d == 4 ? 49u : d == 3 ? 55u : 0u; }
constexpr BlurTab make_blur_tab() {
  BlurTab t{};
  for (int lane = 0; lane < 64; lane++) {
    const int n = lane & 31, h = lane >> 5;
    for (int s = 0; s < 2; s++)
      for (int d = 0; d < 4; d++) {
        uint32_t hv = 0, vv = 0;
        for (int b = 0; b < 4; b++) {
          const int j = 4 * d + b;
          hv |= blur_tap(32 * s + 16 * h + j - n - 13) << (8 * b);                       // H[k = 32 s + 16 h + j][n]: input column k -> output column n
          vv |= blur_tap(32 * s + (j & 3) + 8 * (j >> 2) + 4 * h - n) << (8 * b);        // V[m = n][k = 32 s + rho(h, j)]: row sum k -> output row m
        }
        t.v[lane][4 * s + d] = hv;
        t.v[lane][8 + 4 * s + d] = vv;
      }
  }
  return t;
}
__device__ const BlurTab g_blurTab = make_blur_tab();

#define BLUR_TX 128
#ifndef BLUR_TY
#define BLUR_TY 64               // output rows per workgroup: 32-row products, neighbours share a block of row sums
#endif
#define BLUR_ROWS_IN (BLUR_TY + 6)
#define BLURM_PITCH 176          // bytes per input tile row: 160 used (columns x0 - 16 .. x0 + 143) + one 16-byte chunk of padding
                                 // (44 dwords: 16 consecutive rows start in 16 different 4-bank groups, the A fragments read conflict-free)
#define BLURM_OPITCH 144         // bytes per output staging row
// Workgroup = 128 x 64 outputs (late round 3; 128 x 32 before): input rows y0 - 3 .. y0 + 66 form three 32-row blocks of row sums
// T0, T1, T2 (rows 70 .. 95 of the third only meet zero taps), output rows 0 .. 31 = V (T0, T1), rows 32 .. 63 = V (T1, T2): 6 + 8
// products per 64 rows instead of 8 + 8, one block of row sums and byte splits less, and - what the kernel's time is made of - one HBM
// round trip in front of twice the work.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_blur(FrameParams P) {   // seven workgroups per CU by LDS (21.5 KB each): no more than 72 registers
  typedef int v4i __attribute__((ext_vector_type(4)));
  typedef int v16i __attribute__((ext_vector_type(16)));
  // input rows 0 .. 69, then the output staging rows: the A fragments of the third block read "rows" 70 .. 95, i.e. into the staging
  // area - any bytes will do there, they are multiplied by zero taps
  __shared__ __align__(16) uint8_t sLds[BLUR_ROWS_IN * BLURM_PITCH + BLUR_TY * BLURM_OPITCH];
  static_assert(BLUR_TY % 32 == 0 && (BLUR_TY + 32) * BLURM_PITCH <= BLUR_ROWS_IN * BLURM_PITCH + BLUR_TY * BLURM_OPITCH, "the last block's rows must stay inside the allocation");
  uint8_t *sIn = sLds, *sOut = sLds + BLUR_ROWS_IN * BLURM_PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int tile, frame;
  xcd_map(P.totalTiles, P.magicTiles, P.nframes, frame, tile);
  const uint4 t0 = reinterpret_cast<const uint4 *>(P.tiles)[2 * tile], t1 = reinterpret_cast<const uint4 *>(P.tiles)[2 * tile + 1];
  const int x0 = (int)(t0.x & 0xffffu), y0 = (int)(t0.x >> 16), level = (int)t0.y;
  struct { int w, h, bpitch; size_t boff; } G;
  G.w = (int)(t0.z & 0xffffu); G.h = (int)(t0.z >> 16); G.bpitch = (int)(t0.w >> 16);
  G.boff = ((size_t)t1.w << 32) | t1.z;
  int pitch;
  const uint8_t *img;
  if (level == 0) { pitch = (int)P.img0_stride; img = P.img0 + (size_t)frame * P.img0_frame_stride; }
  else { pitch = (int)(t0.w & 0xffffu); img = P.pyr + (size_t)frame * P.pyr_fs + (((size_t)t1.y << 32) | t1.x); }
  const bool aligned = ((((uintptr_t)img) | (uintptr_t)pitch) & 3u) == 0;
  // ---- input tile: 70 rows x 11 chunks of 16 bytes from column x0 - 16, by LDS-DMA for EVERY tile of a level that is larger than
  // the halo: BORDER_REFLECT_101 across the top / bottom edge is a source ROW (an address), across the left / right edge it is at
  // most three bytes per row, which a few threads copy inside the tile once the DMA has landed.  Chunks that would start outside
  // the row's memory are fetched from a clamped address (their bytes never meet a non-zero tap of a stored output, except the
  // reflected ones); the same holds for rows more than three below the image (clamped).  Tiny levels and unaligned level-0 images
  // take the byte-wise loop.
  const int wlim = level == 0 ? G.w : pitch;                 // bytes of a row that may be read (level 0 is the caller's image: not past its rows)
  const bool dma = aligned && G.w >= 160 && G.h >= 40 && (wlim & 15) == 0;
  const bool edgeL = x0 == 0, edgeR = x0 + 131 >= G.w;       // some stored output of this tile needs a column left of 0 / right of w - 1
  if (dma) {
    for (int idx = tid; idx < BLUR_ROWS_IN * 11; idx += 256) {
      const uint32_t r = mul24((uint32_t)idx, 5958u) >> 16, c = (uint32_t)idx - 11u * r;      // idx / 11, exact below 770
      int yy = y0 - 3 + (int)r;
      yy = yy < 0 ? -yy : yy;
      yy = yy >= G.h ? 2 * (G.h - 1) - yy : yy;
      yy = max(yy, 0);
      int xb = x0 - 16 + 16 * (int)c;
      xb = min(max(xb, 0), wlim - 16);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(img + (mul24((uint32_t)yy, (uint32_t)pitch) + (uint32_t)xb)),
                                       (__attribute__((address_space(3))) void *)(sIn + idx * 16), 16, 0, 0);
    }
    if (edgeL || edgeR) {
      __syncthreads();                                        // the tile has landed (the compiler drains the DMA in front of the barrier)
      for (int i = tid; i < BLUR_ROWS_IN * 6; i += 256) {
        const int r = (int)(mul24((uint32_t)i, 10923u) >> 16), k = i - 6 * r;      // i / 6, exact below 420
        const int x = k < 3 ? -1 - k : G.w + (k - 3);         // the column to synthesise
        const int xs = k < 3 ? 1 + k : G.w - 2 - (k - 3);     // its BORDER_REFLECT_101 source
        const bool need = k < 3 ? edgeL : (edgeR && x - (x0 - 16) < 160);
        if (need) sIn[r * BLURM_PITCH + (x - (x0 - 16))] = sIn[r * BLURM_PITCH + (xs - (x0 - 16))];
      }
    }
  } else {
    // byte-wise loop.  Only columns x0 - 3 .. x0 + 130 meet non-zero taps.
    const bool small = G.w < 16 || G.h < 16;
    auto srcRow = [&](int r) {
      int yy = y0 + r - 3;
      if (small) yy = reflect101(yy, G.h);
      else { yy = yy < 0 ? -yy : yy; yy = yy >= G.h ? 2 * (G.h - 1) - yy : yy; yy = max(yy, 0); }
      return img + mul24((uint32_t)yy, (uint32_t)pitch);
    };
    for (int idx = tid; idx < BLUR_ROWS_IN * 36; idx += 256) {          // dword columns 3 .. 38 of the 40: bytes x0 - 4 .. x0 + 139
      const uint32_t r = mul24((uint32_t)idx, 3641u) >> 17, c = 3u + ((uint32_t)idx - r * 36u);   // idx / 36, exact below 4824
      const int xb = x0 - 16 + 4 * (int)c;
      const uint8_t *row = srcRow((int)r);
      uint32_t v;
      if (aligned && xb >= 0 && xb + 3 < G.w) v = *reinterpret_cast<const uint32_t *>(row + xb);
      else {
        v = 0;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          int xx = xb + kk;
          if (small) xx = reflect101(xx, G.w);
          else { xx = xx < 0 ? -xx : xx; xx = xx >= G.w ? 2 * (G.w - 1) - xx : xx; xx = max(xx, 0); }   // beyond w + w - 1: never under a stored output
          v |= (uint32_t)row[xx] << (8 * kk);
        }
      }
      *reinterpret_cast<uint32_t *>(sIn + r * BLURM_PITCH + 4u * c) = v;
    }
  }
  // ---- constant operands of this lane: H (B of the first product), V (A of the second), in the operands' element order (g_blurTab)
  const int n = lane & 31, hh = lane >> 5;
  v4i Hb[2], Va[2];
  {
    const uint4 *tab = reinterpret_cast<const uint4 *>(g_blurTab.v[lane]);
    const uint4 q0 = tab[0], q1 = tab[1], q2 = tab[2], q3 = tab[3];
    Hb[0] = v4i{(int)q0.x, (int)q0.y, (int)q0.z, (int)q0.w}; Hb[1] = v4i{(int)q1.x, (int)q1.y, (int)q1.z, (int)q1.w};
    Va[0] = v4i{(int)q2.x, (int)q2.y, (int)q2.z, (int)q2.w}; Va[1] = v4i{(int)q3.x, (int)q3.y, (int)q3.z, (int)q3.w};
  }
  __syncthreads();
  // ---- T = I H for a 32-row block of the tile's rows, seeded with 128 * 257 so that the accumulators are the row sums 0 .. 65535,
  // split into (high byte - 128, low byte - 128): the second product's B fragments (element j = register j)
  auto row_sums = [&](int mt, v4i &Bhi, v4i &Blo) {
    // (the seed is made opaque so that every block re-materialises its 16 copies: kept alive as one constant vector across the
    // kernel they cost 16 registers, i.e. resident wavefronts)
    int seedT = 128 * 257;
    asm volatile("" : "+v"(seedT));
    v16i acc;
#pragma unroll
    for (int j = 0; j < 16; j++) acc[j] = seedT;
#pragma unroll
    for (int s = 0; s < 2; s++) {
      v4i a = *reinterpret_cast<const v4i *>(sIn + (32 * mt + n) * BLURM_PITCH + 32 * wid + 32 * s + 16 * hh);
      a[0] ^= (int)0x80808080; a[1] ^= (int)0x80808080; a[2] ^= (int)0x80808080; a[3] ^= (int)0x80808080;
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, Hb[s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int d = 0; d < 4; d++) {
      const uint32_t p01 = __builtin_amdgcn_perm((uint32_t)acc[4 * d + 1], (uint32_t)acc[4 * d], 0x05010400u);   // lo0 lo1 hi0 hi1
      const uint32_t p23 = __builtin_amdgcn_perm((uint32_t)acc[4 * d + 3], (uint32_t)acc[4 * d + 2], 0x05010400u);
      Blo[d] = (int)(__builtin_amdgcn_perm(p23, p01, 0x05040100u) ^ 0x80808080u);
      Bhi[d] = (int)(__builtin_amdgcn_perm(p23, p01, 0x07060302u) ^ 0x80808080u);
    }
  };
  // ---- O = V T over two blocks of row sums, high and low bytes separately; (256 aH + aL) >> 16 saturated to a byte into the staging
  // rows row0 .. row0 + 31 (register j is output row rho(h, j), the lane's column is 32 w + n)
  auto outputs = [&](const v4i &BhiA, const v4i &BloA, const v4i &BhiB, const v4i &BloB, int row0) {
    // ONE accumulator: the high-byte product first, then 256 * aH + the rounding constant is the C operand of the low-byte product
    // (two accumulators side by side cost 16 more registers, i.e. two resident wavefronts per SIMD)
    v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(Va[0], BhiA, v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(Va[1], BhiB, acc, 0, 0, 0);
    int seedL = 257 * 128 * 257 + 32768;
    asm volatile("" : "+v"(seedL));
#pragma unroll
    for (int j = 0; j < 16; j++) acc[j] = (int)(((uint32_t)acc[j] << 8) + (uint32_t)seedL);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(Va[0], BloA, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(Va[1], BloB, acc, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t b = min((uint32_t)acc[j] >> 16, 255u);
      sOut[(row0 + (j & 3) + 8 * (j >> 2) + 4 * hh) * BLURM_OPITCH + 32 * wid + n] = (uint8_t)b;
    }
  };
  // (the scheduling barriers keep the phases apart: interleaved, their accumulators need 84 registers instead of 64 and cost three
  // resident wavefronts per SIMD)
  v4i Hp, Lp, Hc, Lc;
  row_sums(0, Hp, Lp);
#pragma unroll
  for (int k = 0; k < BLUR_TY / 32; k++) {
    if (k > 0 && y0 + 32 * k >= G.h) break;      // no output row down there (uniform)
    __builtin_amdgcn_sched_barrier(0);
    row_sums(k + 1, Hc, Lc);
    __builtin_amdgcn_sched_barrier(0);
    outputs(Hp, Lp, Hc, Lc, 32 * k);
    Hp = Hc; Lp = Lc;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < BLUR_TY / 32; t++) {
    const int row = (tid >> 3) + 32 * t, seg = tid & 7;
    const int y = y0 + row, x = x0 + 16 * seg;
    if (y < G.h && x < G.w) {
      uint8_t *out = P.blur + (size_t)frame * P.blur_fs + G.boff + (size_t)y * G.bpitch + x;
      const uint4 v = *reinterpret_cast<const uint4 *>(sOut + row * BLURM_OPITCH + 16 * seg);
      if (x + 16 <= G.bpitch) *reinterpret_cast<uint4 *>(out) = v;
      else {
        const uint8_t *b = sOut + row * BLURM_OPITCH + 16 * seg;
        for (int cc = 0; cc < 16 && x + cc < G.w; cc++) out[cc] = b[cc];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// K4+K6+K7: IC_Angle (ORBextractor.cc:75-102), computeOrbDescriptor (:106-145) and the lapping-order scatter
// (:1159-1180).  One wavefront per keypoint: 749-pixel disc split over 64 lanes and reduced with shuffles;
// 256 binary tests = 4 rounds of 64 lanes, each round packed with one 64-bit ballot (= 8 descriptor bytes).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {  // cv::fastAtan2, SURVEY.md A.6
  const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
  const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
  const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
  float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

struct KpOut { float x, y, size, angle, response; int32_t octave, class_id; };

__global__ __launch_bounds__(256) void k_describe(FrameParams P) {
  const int lane = threadIdx.x & 63;
  int frame, blk;
  xcd_map((P.totalKp + 3) / 4, P.magicKpBlk, P.nframes, frame, blk);
  const int j = blk * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: scalar loads below
  if (j >= P.totalKp) return;
  // Everything that does not depend on the keypoint is requested first (one memory latency for all of it): my share of
  // the disc and pattern tables, the level counts, the keypoint itself (its slot j is level-independent).
  const uint32_t packed = P.lkp[(size_t)frame * P.lkp_fs + j];
  const uint32_t rk = P.lrank[(size_t)frame * P.lkp_fs + j];
  const uint4 dw0 = *reinterpret_cast<const uint4 *>(&c_disc.w[lane][0]);
  const uint2 dw1 = *reinterpret_cast<const uint2 *>(&c_disc.w[lane][4]);
  uint32_t pw[4];
#pragma unroll
  for (int r = 0; r < 4; r++) pw[r] = reinterpret_cast<const uint32_t *>(c_pattern)[r * 64 + lane];  // x0, y0, x1, y1 as int8
  const int32_t *lc = P.lcnt + (size_t)frame * P.nlevels * 2;
  int level = 0;
#pragma unroll
  for (int l = 1; l < ORB_MAXL; l++)
    if (l < P.nlevels && j >= P.geom[l].kpBase) level = l;
  int nTot = 0, lapTot = 0, lapBefore = 0, monoBefore = 0, myCount = 0;
#pragma unroll
  for (int l = 0; l < ORB_MAXL; l++)
    if (l < P.nlevels) {
      const int c = lc[l * 2], lp = lc[l * 2 + 1];
      nTot += c;
      lapTot += lp;
      if (l < level) { lapBefore += lp; monoBefore += c - lp; }
      if (l == level) myCount = c;
    }
  // the frame's totals (keypoints, monoIndex of operator(): ORBextractor.cc:1169-1182) are written by its first wavefront
  if (j == 0 && lane == 0) { P.out_counts[frame * 2] = nTot; P.out_counts[frame * 2 + 1] = nTot - lapTot; }
  const LevelGeom G = P.geom[level];
  const int i = j - G.kpBase;
  if (i >= myCount) return;
  const int X = (int)(packed & 0xfff) + ORB_MIN_BORDER, Y = (int)((packed >> 12) & 0xfff) + ORB_MIN_BORDER;
  const int dst = (rk & 0x8000u) ? nTot - 1 - (lapBefore + (int)(rk & 0x7fff)) : monoBefore + (int)(rk & 0x7fff);

  // IC_Angle on the unblurred level: 12 independent gathers per lane, all in flight together
  int pitch;
  const uint8_t *img = level_plane(P, frame, level, pitch);
  const uint8_t *centre = img + (size_t)Y * pitch + X;
  // The 37x37 blurred patch the 512 steered test points can fall into (|rotated offset| <= 18) is requested NOW, together
  // with the disc gathers, as 37 rows x 10 aligned dwords into this wave's LDS slice: the BRIEF gathers then never leave
  // the CU and the wave pays one global round trip less.  Keypoints are >= 19 px inside the level, so rows Y-18..Y+18 and
  // columns X-18..X+18 exist; the up to 3 extra bytes of the aligned dwords stay inside the plane's pitch padding / next row.
  __shared__ __align__(16) uint8_t sPatch[4][37 * 48];
  uint8_t *myPatch = sPatch[threadIdx.x >> 6];
  const int px0 = (X - 18) & ~3, pox = (X - 18) - px0;
  {
    // LDS-DMA, 16 bytes per lane straight into the patch (no register, no ds_write): 37 rows x 3 chunks, lane-linear;
    // idx / 3 as (idx * 21846) >> 16, exact below 111.  The 48-byte rows reach at most 11 bytes past column X+18+3: inside
    // the plane (the row below exists: keypoints stay 19 px away from the border).
    const uint8_t *brow = P.blur + (size_t)frame * P.blur_fs + G.boff + (size_t)(Y - 18) * G.bpitch + px0;
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const uint32_t idx = (uint32_t)(lane + 64 * t), r = (idx * 21846u) >> 16, c = idx - 3u * r;
      if (idx < 111u)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(brow + (mul24(r, (uint32_t)G.bpitch) + 16u * c)),
                                         (__attribute__((address_space(3))) void *)&myPatch[idx * 16], 16, 0, 0);
    }
  }
  int m10 = 0, m01 = 0;
  int dv[12], du[12], dval[12];
  {
    const uint32_t uw[3] = {dw0.x, dw0.y, dw0.z}, vw[3] = {dw0.w, dw1.x, dw1.y};
#pragma unroll
    for (int t = 0; t < 12; t++) {
      du[t] = (int)(int8_t)(uw[t >> 2] >> (8 * (t & 3)));
      dv[t] = (int)(int8_t)(vw[t >> 2] >> (8 * (t & 3)));
    }
  }
  // The disc pixels: when the level plane is dword-aligned, its 31 rows x 48 bytes around the keypoint come in by LDS-DMA
  // as well (two 16-byte chunks per lane instead of twelve scattered byte gathers) and the disc is read from LDS.
  __shared__ __align__(16) uint8_t sDisc[4][31 * 48];
  uint8_t *myDisc = sDisc[threadIdx.x >> 6];
  if (((((uintptr_t)img) | (uintptr_t)pitch) & 3u) == 0) {
    const int qx0 = (X - 15) & ~3, qox = (X - 15) - qx0;
    const uint8_t *irow = img + (size_t)(Y - 15) * pitch + qx0;
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const uint32_t idx = (uint32_t)(lane + 64 * t), r = (idx * 21846u) >> 16, c = idx - 3u * r;
      if (idx < 93u)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(irow + (mul24(r, (uint32_t)pitch) + 16u * c)),
                                         (__attribute__((address_space(3))) void *)&myDisc[idx * 16], 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // both patches have landed
    const uint8_t *dc = myDisc + 15 * 48 + qox + 15;
#pragma unroll
    for (int t = 0; t < 12; t++) dval[t] = dc[dv[t] * 48 + du[t]];
  } else {
#pragma unroll
    for (int t = 0; t < 12; t++) dval[t] = centre[dv[t] * pitch + du[t]];
  }
#pragma unroll
  for (int t = 0; t < 12; t++) { m10 += du[t] * dval[t]; m01 += dv[t] * dval[t]; }
  m10 = wave_sum_i32(m10);
  m01 = wave_sum_i32(m01);
  const float angle = fast_atan2_deg((float)m01, (float)m10);

  // steered BRIEF on the blurred level
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  const float arad = angle * factorPI;
  const float a = orbsc::ref_cosf(arad), b = orbsc::ref_sinf(arad);
  const uint8_t *bc = myPatch + 18 * 48 + pox + 18;   // patch centre; row pitch 48 B
  constexpr int bp = 48;
  unsigned long long bits[4];
  int o0[4], o1[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const float x0 = (float)(int8_t)(pw[r] & 0xff), y0 = (float)(int8_t)((pw[r] >> 8) & 0xff);
    const float x1 = (float)(int8_t)((pw[r] >> 16) & 0xff), y1 = (float)(int8_t)(pw[r] >> 24);
    o0[r] = __float2int_rn(x0 * b + y0 * a) * bp + __float2int_rn(x0 * a - y0 * b);
    o1[r] = __float2int_rn(x1 * b + y1 * a) * bp + __float2int_rn(x1 * a - y1 * b);
  }
  // the patch was written by this wavefront's own LDS-DMA loads (VM counter): long retired by now, but say so
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int v0[4], v1[4];
#pragma unroll
  for (int r = 0; r < 4; r++) { v0[r] = bc[o0[r]]; v1[r] = bc[o1[r]]; }
#pragma unroll
  for (int r = 0; r < 4; r++) bits[r] = __ballot(v0[r] < v1[r]);
  if (dst < P.cap) {
    uint32_t *dd = reinterpret_cast<uint32_t *>(P.out_desc + ((size_t)frame * P.cap + dst) * 32);
    if (lane < 8) {
      unsigned long long w = bits[lane >> 1];
      dd[lane] = (uint32_t)((lane & 1) ? (w >> 32) : w);
    }
    if (lane == 8) {
      KpOut k;
      k.x = (float)X;
      k.y = (float)Y;
      if (level != 0) { k.x = k.x * G.scale; k.y = k.y * G.scale; }
      k.size = G.kpsize;
      k.angle = angle;
      k.response = (float)(packed >> 24);
      k.octave = level;
      k.class_id = -1;
      reinterpret_cast<KpOut *>(P.out_kps)[(size_t)frame * P.cap + dst] = k;
    }
  }
}

// cv::cvtColor(im, gray, CV_RGB2GRAY / CV_BGR2GRAY / CV_RGBA2GRAY / CV_BGRA2GRAY) as called by Tracking::GrabImage*
// (Tracking.cc:1122-1135) for 8-bit images: OpenCV's fixed-point RGB2Gray, (R*4899 + G*9617 + B*1868 + 8192) >> 14
// (R2Y, G2Y, B2Y with yuv_shift 14; alpha ignored).  Four output pixels per thread, dword store.
__global__ __launch_bounds__(256) void k_cvt_gray(const uint8_t *src, int rows, int cols, size_t sstride, int ch, int rgb, uint8_t *dst,
                                                  size_t dstride) {
  const int q = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  const int x0 = 4 * q;
  if (x0 >= cols || y >= rows) return;
  const uint8_t *s = src + (size_t)y * sstride + (size_t)x0 * ch;
  const int cr = rgb ? 0 : 2, cb = rgb ? 2 : 0;
  uint32_t packed = 0;
  const int n = min(4, cols - x0);
  for (int j = 0; j < n; j++) {
    const uint32_t g = ((uint32_t)s[j * ch + cr] * 4899u + (uint32_t)s[j * ch + 1] * 9617u + (uint32_t)s[j * ch + cb] * 1868u + 8192u) >> 14;
    packed |= g << (8 * j);
  }
  uint8_t *d = dst + (size_t)y * dstride + x0;
  if (n == 4 && ((((uintptr_t)dst) | dstride) & 3u) == 0) *reinterpret_cast<uint32_t *>(d) = packed;
  else for (int j = 0; j < n; j++) d[j] = (uint8_t)(packed >> (8 * j));
}

// cv::CLAHE::apply as the TUM-VI examples run it on every image before TrackMonocular / TrackStereo
// (Examples/Monocular/mono_tum_vi.cc:101-109, createCLAHE(3.0, Size(8, 8))), OpenCV 3.4/4.x clahe.cpp for CV_8UC1:
// per-tile histogram -> clip at clipLimit, the excess spread evenly (batch + every residualStep-th bin) -> cumulative
// LUT saturate_cast<uchar>(sum * lutScale); then every pixel blends the LUTs of its four neighbouring tiles bilinearly
// in float.  When the image size is not a multiple of the tile grid the histogram runs over the image extended to the
// next multiple with BORDER_REFLECT_101 (tw, th are the tile sizes of the extended image).
// k_clahe_lut: one workgroup per tile, thread b owns histogram bin b.
__global__ __launch_bounds__(256) void k_clahe_lut(const uint8_t *src, int rows, int cols, size_t sstride, int tilesX, int tw, int th, int clipLimit,
                                                   float lutScale, uint8_t *lut) {
  __shared__ uint32_t hist[256];
  __shared__ int part[4];
  const int t = threadIdx.x, tile = blockIdx.x, tx = tile % tilesX, ty = tile / tilesX;
  hist[t] = 0;
  __syncthreads();
  const int x0 = tx * tw, y0 = ty * th;
  for (int i = t; i < tw * th; i += 256) {
    int y = y0 + i / tw, x = x0 + i % tw;
    if (y >= rows) y = 2 * (rows - 1) - y;   // BORDER_REFLECT_101, the extension is narrower than the image
    if (x >= cols) x = 2 * (cols - 1) - x;
    atomicAdd(&hist[src[(size_t)y * sstride + x]], 1u);
  }
  __syncthreads();
  int v = (int)hist[t];
  if (clipLimit > 0) {
    int excess = v > clipLimit ? v - clipLimit : 0;
    if (v > clipLimit) v = clipLimit;
    const int ws = wave_sum_i32(excess);
    if ((t & 63) == 0) part[t >> 6] = ws;
    __syncthreads();
    const int clipped = part[0] + part[1] + part[2] + part[3];
    const int redistBatch = clipped / 256;
    const int residual = clipped - redistBatch * 256;
    v += redistBatch;
    if (residual != 0) {
      const int step = max(256 / residual, 1);
      if (t % step == 0 && t / step < residual) v++;
    }
  }
  __syncthreads();
  hist[t] = (uint32_t)v;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {   // inclusive scan over the 256 bins
    const uint32_t a = t >= d ? hist[t - d] : 0u;
    __syncthreads();
    hist[t] += a;
    __syncthreads();
  }
  const int r = __float2int_rn(__fmul_rn((float)(int)hist[t], lutScale));
  lut[(size_t)tile * 256 + t] = (uint8_t)min(max(r, 0), 255);
}

// k_clahe_interp: the whole LUT (tilesX*tilesY*256 bytes, 16 KB for 8x8) sits in LDS; four pixels per thread.
// Every product and sum is rounded on its own, as the reference's scalar float expression is (no fused multiply-add).
__global__ __launch_bounds__(256) void k_clahe_interp(const uint8_t *src, int rows, int cols, size_t sstride, int tilesX, int tilesY, float inv_tw,
                                                      float inv_th, const uint8_t *lut, uint8_t *dst, size_t dstride, int rowsPerBlock) {
  extern __shared__ uint8_t sLut[];
  const int nl = tilesX * tilesY * 256;
  for (int i = threadIdx.x * 4; i < nl; i += 1024) *reinterpret_cast<uint32_t *>(sLut + i) = *reinterpret_cast<const uint32_t *>(lut + i);
  __syncthreads();
  const int yb = blockIdx.x * rowsPerBlock, ye = min(yb + rowsPerBlock, rows);
  const int nq = (cols + 3) >> 2;
  for (int i = threadIdx.x; i < (ye - yb) * nq; i += 256) {
    const int y = yb + i / nq, xq = (i % nq) * 4;
    const float tyf = __fsub_rn(__fmul_rn((float)y, inv_th), 0.5f);
    int ty1 = (int)floorf(tyf);
    const float ya = __fsub_rn(tyf, (float)ty1), ya1 = __fsub_rn(1.0f, ya);
    const int ty2 = min(ty1 + 1, tilesY - 1);
    ty1 = max(ty1, 0);
    const uint8_t *p1 = sLut + ty1 * tilesX * 256, *p2 = sLut + ty2 * tilesX * 256;
    const uint8_t *srow = src + (size_t)y * sstride;
    uint8_t *drow = dst + (size_t)y * dstride;
    const int n = min(4, cols - xq);
    uint32_t packed = 0;
    for (int j = 0; j < n; j++) {
      const int x = xq + j;
      const float txf = __fsub_rn(__fmul_rn((float)x, inv_tw), 0.5f);
      int tx1 = (int)floorf(txf);
      const float xa = __fsub_rn(txf, (float)tx1), xa1 = __fsub_rn(1.0f, xa);
      const int tx2 = min(tx1 + 1, tilesX - 1);
      tx1 = max(tx1, 0);
      const int sv = srow[x];
      const int i1 = tx1 * 256 + sv, i2 = tx2 * 256 + sv;
      const float top = __fadd_rn(__fmul_rn((float)p1[i1], xa1), __fmul_rn((float)p1[i2], xa));
      const float bot = __fadd_rn(__fmul_rn((float)p2[i1], xa1), __fmul_rn((float)p2[i2], xa));
      const float res = __fadd_rn(__fmul_rn(top, ya1), __fmul_rn(bot, ya));
      packed |= (uint32_t)min(max(__float2int_rn(res), 0), 255) << (8 * j);
    }
    if (n == 4 && ((((uintptr_t)dst) | dstride) & 3u) == 0) *reinterpret_cast<uint32_t *>(drow + xq) = packed;
    else for (int j = 0; j < n; j++) drow[xq + j] = (uint8_t)(packed >> (8 * j));
  }
}

// cv::remap(im, imRect, M1, M2, cv::INTER_LINEAR) of the stereo examples (Examples/Stereo/stereo_euroc.cc:166-167), with the
// CV_32FC1 maps initUndistortRectifyMap(..., CV_32F, M1, M2) produced (:113-114) and the default BORDER_CONSTANT (0).  OpenCV
// quantises the source coordinate to 1/32 pixel (cvRound(map * INTER_TAB_SIZE)), takes the bilinear weights from a table of
// 15-bit fixed-point products -- for the linear kernel exactly 32 * (32-fx|fx) * (32-fy|fy); the one saturated entry of the table
// (fx = fy = 0: {32767, 0, 0, 1} after its correction step) yields the same pixel as {32768, 0, 0, 0} for 8-bit taps -- and rounds
// with (sum + (1 << 14)) >> 15.  A tap outside the source contributes the border value.
__global__ __launch_bounds__(256) void k_remap_linear(const uint8_t *src, int srows, int scols, size_t sstride, const float *mapx, const float *mapy,
                                                      size_t mstride, int rows, int cols, uint8_t *dst, size_t dstride) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= cols || y >= rows) return;
  const float mx = mapx[(size_t)y * mstride + x], my = mapy[(size_t)y * mstride + x];
  // cvRound = cvtss2si: out-of-range and NaN inputs give INT_MIN there; __float2int_rn saturates, NaN is patched to match
  const int sx = (mx != mx || fabsf(mx) >= 67108864.0f) ? INT_MIN : __float2int_rn(__fmul_rn(mx, 32.0f));
  const int sy = (my != my || fabsf(my) >= 67108864.0f) ? INT_MIN : __float2int_rn(__fmul_rn(my, 32.0f));
  const int fx = sx & 31, fy = sy & 31;
  const int ix = min(max(sx >> 5, -32768), 32767), iy = min(max(sy >> 5, -32768), 32767);   // saturate_cast<short>
  auto tap = [&](int yy, int xx) -> int { return ((unsigned)xx < (unsigned)scols && (unsigned)yy < (unsigned)srows) ? src[(size_t)yy * sstride + xx] : 0; };
  const int w00 = 32 * (32 - fx) * (32 - fy), w01 = 32 * fx * (32 - fy), w10 = 32 * (32 - fx) * fy, w11 = 32 * fx * fy;
  const int acc = tap(iy, ix) * w00 + tap(iy, ix + 1) * w01 + tap(iy + 1, ix) * w10 + tap(iy + 1, ix + 1) * w11;
  dst[(size_t)y * dstride + x] = (uint8_t)((acc + (1 << 14)) >> 15);
}
