// orb_mfma_util.h -- Hamming distances of the projection searches on the matrix pipe: the shared pieces (operand expansion,
// per-lane sorted lists) of k_match_scan_mfma (orb_match_mfma.h) and of the list build inside k_match_resolve's fused form.
//
// BASELINE's "1000 x 1000" setting (and every relocalisation-style search whose windows cover the frame) asks for ALL
// Hamming distances between the queries and the keypoints of a frame (ORBmatcher::DescriptorDistance, ORBmatcher.cc:2463-2483,
// inside the loops :99-120 / :2148-2156).  k_match_scan does that with 8 xor + 8 v_bcnt per query-candidate pair and wavefront;
// the popcount issues at a quarter of the VALU rate (profiles/valu_calib.json), so the scan was vector-issue bound while the
// matrix pipe idled.  A 256-bit Hamming distance is an exact int8 dot product:
//
//     with a_k = bit ? -32 : +32 (candidate) and b_k = bit ? +32 : -32 (query):   sum_k a_k b_k = 1024 (2 ham - 256)
//
// so  v_mfma_i32_32x32x32_i8  over the 8 K-steps of 32 bits, started from the accumulator  C[row] = 2^18 + rank[row], leaves
//
//     D[row = candidate][col = query] = ham << 11 | rank[candidate]
//
// i.e. the REDUCTION KEY ITSELF, without a single vector instruction: rank = position of the candidate in the enumeration order
// of Frame::GetFeaturesInArea (Frame.cc:781-809: grid column, then row, then insertion order), which decides ties between equal
// distances (strict <, first minimum wins) exactly as the 23 bits cell << 11 | idx of Key32 do - the rank is the same order,
// compressed to 11 bits so that it fits below a product-scaled distance (k_match_rank computes it once per frame pair).
// A candidate that is outside the grid or already held gets C = 2^18 + 2^30: its keys are larger than every real key.
//
// Layout of a 32 x 32 tile (guide, MFMA C/D map): col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) - a lane owns ONE
// query (its column) and sees 16 of the tile's 32 candidates; the two lanes of a column keep separate sorted top-8 lists of
// their halves of the frame, merged at the end.  Every wavefront holds two query tiles (64 queries) as B fragments in registers
// (the 1024 expanded bytes of a K-step pair never leave them), the candidate tiles are expanded once per workgroup into LDS
// (bit -> byte by a 4-bit multiply spread and one v_perm_b32) and read back as A fragments by conflict-free ds_read_b128
// (16-byte chunks of a row rotated by the row index).
//
// What is left on the VALU is the selection: every key goes through a branch-free sorted insertion, 1 v_min + (K - 1) v_med3
// (new[j] = med3(top[j-1], top[j], t): independent instructions, half the count of a compare-exchange chain).
//
// Users.  The brute-force entries (k_hamming_matrix_mfma, k_knn2_mfma: orb_match_mfma.h) take the product as it is.  k_match_scan_mfma (orb_match_mfma.h): 256-query blocks whose live queries are ALL "open" (window = whole grid, no
// level filter; query_is_open) of monocular problems on frames of at most 2048 keypoints get their top-8 lists from it and
// k_match_scan skips exactly those blocks (same vote); the lists are the same Key32 lists either way.  k_match_resolve's FUSED
// form (orb_match_kernels.h): frame pairs all of whose queries are open build the lists of each 64-query chunk inside the
// resolve kernel, with every keypoint a committed claim holds masked out through its accumulator seed - a list made that way
// cannot be exhausted by the claims of earlier chunks, which is what the refresh passes of the separate-kernel form
// spent their time on.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef int mf_v4i __attribute__((ext_vector_type(4)));
typedef int mf_v16i __attribute__((ext_vector_type(16)));

#define MF_NT 256
#define MF_TILE 32
#define MF_REC_BASE (1u << 18)
#define MF_REC_HELD (1u << 30)
#define MF_REC_UNUSABLE ((1u << 18) + MF_REC_HELD)
#define MF_KEY_LIMIT (1u << 20)   // real keys are < 257 << 11

// median of three as the min / max expression the backend selects v_med3_u32 for.  NOT inline assembly: the compiler pads the
// wait states between an MFMA and the first vector instruction that reads its result only for instructions it emitted itself.
__device__ __forceinline__ uint32_t mf_med3(uint32_t a, uint32_t b, uint32_t c) { return max(min(a, b), min(max(a, b), c)); }

// 16 descriptor bits -> 16 bytes: LUT byte 0 for a 0 bit, LUT byte 1 for a 1 bit (v_perm_b32 selectors 0 / 1 pick bytes of src1)
__device__ __forceinline__ mf_v4i mf_expand16(uint32_t hw, uint32_t lut) {
  mf_v4i o;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t nib = (hw >> (4 * i)) & 0xfu;
    const uint32_t t = (nib * 0x00204081u) & 0x01010101u;          // bit i of the nibble -> byte i (0 / 1)
    o[i] = (int)__builtin_amdgcn_perm(0u, lut, t);
  }
  return o;
}
#define MF_LUT_CAND 0x0000E020u    // candidate: 0 -> +32, 1 -> -32
#define MF_LUT_QUERY 0x000020E0u   // query:     0 -> -32, 1 -> +32

template <int K>
struct MfList {
  uint32_t top[K];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int j = 0; j < K; j++) top[j] = 0xffffffffu;
  }
  __device__ __forceinline__ void insert(uint32_t t) {   // sorted insertion; a no-op for t = 0xffffffff
#pragma unroll
    for (int j = K - 1; j >= 1; j--) top[j] = mf_med3(top[j - 1], top[j], t);
    top[0] = min(top[0], t);
  }
  // Every key of the tile goes through the sorted insertion, unconditionally: K independent instructions per key and no vote,
  // no branch, no mask.  With 64 lanes x 16 keys behind every vote "does any key of the tile enter some lane's list" the answer
  // is yes for nearly every tile of a 1000-keypoint frame (a lane's top-8 list changes 8 ln(m / 8) + 8 = 41 times over its 500
  // candidates, 2600 times per wavefront and list), and k_match_scan's parked insertion then pays two ballots, two scalar
  // branches and a VCC select per key ON TOP of the insertions (measured with it: 3700 cycles per tile and wavefront in the
  // selection against 1000 for the MFMAs, tools/mfma_stamps.py; branch-free: 0.174 -> 0.125 ms for 256 frame pairs).
  __device__ __forceinline__ void take(const mf_v16i &k) {
#pragma unroll
    for (int r = 0; r < 16; r++) insert((uint32_t)k[r]);
  }
};

