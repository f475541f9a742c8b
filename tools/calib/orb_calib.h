// orb_calib.h -- measurement aids, not part of the product path.
//
// k_calib_valu<OP>: the vector-issue ceiling of one opcode class on this chip, measured the way k_calib_copy_u32 pins
// the HBM counters: a stream of INDEPENDENT instructions of that class (16 accumulator chains per lane, so that no
// instruction waits for its predecessor), 128 per loop trip, at 1 / 2 / 4 / 8 resident wavefronts per SIMD on every
// CU.  Residency is fixed by the dynamic LDS size of the 256-thread workgroups (160 KiB / waves-per-SIMD each), the
// grid is several times what is resident so that every CU stays full.  The hardware guide gives 2 cycles per wave64
// VALU instruction when several waves share a SIMD and 4 for one wave alone (MI355X_MICROARCH.md, cycle constants);
// the per-class numbers measured here replace that assumption in bench.py's `valu_issue` object
// (profiles/valu_calib.json, written by tools/collect_valu_calib.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum CalibOp {
  CAL_ADD_U32 = 0,    // v_add_u32            VOP2, 32-bit integer
  CAL_XOR_B32,        // v_xor_b32            VOP2 logic
  CAL_MINMAX_U32,     // v_min_u32 / v_max_u32 alternating (top-8 insertion network of k_match_scan)
  CAL_PK_MINMAX_I16,  // v_pk_min_i16 / v_pk_max_i16 alternating (k_fast score network)
  CAL_PK_MAD_U16,     // v_pk_mad_u16
  CAL_BCNT,           // v_bcnt_u32_b32       Hamming popcount with accumulate
  CAL_DOT4_U8,        // v_dot4_u32_u8        k_blur horizontal pass
  CAL_DOT2_U16,       // v_dot2_u32_u16       k_blur vertical pass
  CAL_PERM,           // v_perm_b32           byte gathers
  CAL_ALIGNBYTE,      // v_alignbyte_b32
  CAL_MUL_U24,        // v_mul_u32_u24
  CAL_MAD_U24,        // v_mad_u32_u24        VOP3, three sources
  CAL_MUL_LO_U32,     // v_mul_lo_u32         full 32-bit multiply
  CAL_LSHL_ADD_U32,   // v_lshl_add_u32       VOP3 address arithmetic
  CAL_LSHL_ADD_U64,   // v_lshl_add_u64       64-bit address arithmetic
  CAL_CNDMASK,        // v_cndmask_b32        reads VCC
  CAL_CMP,            // v_cmp_lt_u32         writes VCC
  CAL_SDWA,           // v_add_u32_sdwa       byte-select operand forms (k_fast, k_resize)
  CAL_DPP,            // v_add_u32_dpp row_shr:1
  CAL_MOV,            // v_mov_b32
  CAL_FMA_F32,        // v_fma_f32            the guide's reference instruction
  CAL_FMA_F64,        // v_fma_f64            k_describe's sin/cos polynomial
  CAL_FAST_MIX,       // the k_fast score network's own mix: 4 v_pk_min_i16 : 1 v_pk_max_i16 : 1 v_pk_mad_u16 : 2 v_add_u32
  CAL_AND_B32,        // v_and_b32
  CAL_OR_B32,         // v_or_b32
  CAL_LSHLREV,        // v_lshlrev_b32
  CAL_LSHRREV,        // v_lshrrev_b32
  CAL_SUB_U32,        // v_sub_u32
  CAL_ADD3_U32,       // v_add3_u32
  CAL_AND_OR,         // v_and_or_b32
  CAL_LSHL_OR,        // v_lshl_or_b32
  CAL_BFE_U32,        // v_bfe_u32
  CAL_MINMAX_I32,     // v_min_i32 / v_max_i32
  CAL_MIN3_U32,       // v_min3_u32
  CAL_MED3_I32,       // v_med3_i32
  CAL_PK_ADD_U16,     // v_pk_add_u16
  CAL_PK_SUB_I16,     // v_pk_sub_i16
  CAL_SAD_U8,         // v_sad_u8
  CAL_BITOP3,         // v_bitop3_b32 (gfx950 three-input logic)
  CAL_CNDMASK_SGPR,   // v_cndmask_b32 with the mask in an SGPR pair written once by the scalar unit
  CAL_CNDMASK_VCC_S,  // v_cndmask_b32 ..., vcc with VCC written once by the scalar unit
  CAL_ADD_F32,        // v_add_f32
  CAL_MUL_F32,        // v_mul_f32
  CAL_CVT_F32_U32,    // v_cvt_f32_u32
  CAL_ADD_CO_U32,     // v_add_co_u32 (carry out to VCC)
  CAL_MINMAX_F32,     // v_min_f32 / v_max_f32
  CAL_MIN3_F32,       // v_min3_f32
  CAL_MAX3_F32,       // v_max3_f32
  CAL_MED3_F32,       // v_med3_f32
  CAL_PK_MINMAX_F16,  // v_pk_min_f16 / v_pk_max_f16
  CAL_PK_FMA_F16,     // v_pk_fma_f16
  CAL_PK_ADD_F16,     // v_pk_add_f16
  CAL_MINMAX_F16,     // v_min_f16 / v_max_f16
  CAL_MINMAX_U16,     // v_min_u16 / v_max_u16
  CAL_CVT_UBYTE,      // v_cvt_f32_ubyte0
  CAL_SUB_F32,        // v_sub_f32
  CAL_MAX3_U32,       // v_max3_u32
  CAL_PK_MINMAX_U16,  // v_pk_min_u16 / v_pk_max_u16
  CAL_ADDC,           // v_addc_co_u32 (reads and writes VCC)
  CAL_SALU_ADD,       // s_add_u32 (scalar unit: one per CU?)
  CAL_SALU_AND64,     // s_and_b64 / s_bcnt1_i32_b64 (the mask arithmetic around ballots)
  CAL_DS_READ_U8,     // ds_read_u8 with a per-lane address (LDS issue rate as k_fast's gathers see it)
  CAL_XOR_SGPR,       // v_xor_b32 with a scalar-register source (k_match_scan: candidate descriptor words in SGPRs)
  CAL_BCNT_SGPR,      // v_bcnt_u32_b32 fed by such a xor is VGPR-only; this one counts a scalar source directly
  CAL_NUM_OPS
};

static const char *const kCalibOpNames[CAL_NUM_OPS] = {
    "v_add_u32", "v_xor_b32", "v_min_u32/v_max_u32", "v_pk_min_i16/v_pk_max_i16", "v_pk_mad_u16", "v_bcnt_u32_b32", "v_dot4_u32_u8",
    "v_dot2_u32_u16", "v_perm_b32", "v_alignbyte_b32", "v_mul_u32_u24", "v_mad_u32_u24", "v_mul_lo_u32", "v_lshl_add_u32",
    "v_lshl_add_u64", "v_cndmask_b32", "v_cmp_lt_u32", "v_add_u32_sdwa", "v_add_u32_dpp", "v_mov_b32", "v_fma_f32", "v_fma_f64",
    "k_fast score mix (4 pk_min:1 pk_max:1 pk_mad:2 add)", "v_and_b32", "v_or_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_sub_u32", "v_add3_u32",
    "v_and_or_b32", "v_lshl_or_b32", "v_bfe_u32", "v_min_i32/v_max_i32", "v_min3_u32", "v_med3_i32", "v_pk_add_u16", "v_pk_sub_i16", "v_sad_u8",
    "v_bitop3_b32", "v_cndmask_b32 (SGPR-pair mask)", "v_cndmask_b32 (vcc, scalar-written)", "v_add_f32", "v_mul_f32", "v_cvt_f32_u32",
    "v_add_co_u32", "v_min_f32/v_max_f32", "v_min3_f32", "v_max3_f32", "v_med3_f32", "v_pk_min_f16/v_pk_max_f16", "v_pk_fma_f16", "v_pk_add_f16",
    "v_min_f16/v_max_f16", "v_min_u16/v_max_u16", "v_cvt_f32_ubyte0", "v_sub_f32", "v_max3_u32", "v_pk_min_u16/v_pk_max_u16", "v_addc_co_u32",
    "s_add_u32", "s_and_b64/s_bcnt1_i32_b64", "ds_read_u8", "v_xor_b32 (SGPR source)", "v_bcnt_u32_b32 (SGPR source)"};

#define CAL_INSTR_PER_TRIP 128

// one instruction per accumulator d: operands %0..%15 accumulators, %16 / %17 loop-invariant sources
#define CAL_R16(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(8) I(9) I(10) I(11) I(12) I(13) I(14) I(15)
#define CAL_TRIP(I) CAL_R16(I) CAL_R16(I) CAL_R16(I) CAL_R16(I) CAL_R16(I) CAL_R16(I) CAL_R16(I) CAL_R16(I)
#define CAL_TRIP2(I, J) CAL_R16(I) CAL_R16(J) CAL_R16(I) CAL_R16(J) CAL_R16(I) CAL_R16(J) CAL_R16(I) CAL_R16(J)
#define CAL_OPERANDS                                                                                                                \
  : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), \
    "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])                                                                 \
  : "v"(a), "v"(b)                                                                                                                  \
  : "vcc"

#define I_ADD(d) "v_add_u32 %" #d ", %" #d ", %16\n"
#define I_XOR(d) "v_xor_b32 %" #d ", %" #d ", %16\n"
#define I_XORS(d) "v_xor_b32 %" #d ", s20, %" #d "\n"
#define I_BCNTS(d) "v_bcnt_u32_b32 %" #d ", s20, %" #d "\n"
#define I_MINU(d) "v_min_u32 %" #d ", %" #d ", %16\n"
#define I_MAXU(d) "v_max_u32 %" #d ", %" #d ", %17\n"
#define I_PKMIN(d) "v_pk_min_i16 %" #d ", %" #d ", %16\n"
#define I_PKMAX(d) "v_pk_max_i16 %" #d ", %" #d ", %17\n"
#define I_PKMAD(d) "v_pk_mad_u16 %" #d ", %" #d ", %16, %17\n"
#define I_BCNT(d) "v_bcnt_u32_b32 %" #d ", %16, %" #d "\n"
#define I_DOT4(d) "v_dot4_u32_u8 %" #d ", %16, %17, %" #d "\n"
#define I_DOT2(d) "v_dot2_u32_u16 %" #d ", %16, %17, %" #d "\n"
#define I_PERM(d) "v_perm_b32 %" #d ", %" #d ", %16, %17\n"
#define I_ALIGNB(d) "v_alignbyte_b32 %" #d ", %" #d ", %16, 1\n"
#define I_MUL24(d) "v_mul_u32_u24 %" #d ", %" #d ", %16\n"
#define I_MAD24(d) "v_mad_u32_u24 %" #d ", %" #d ", %16, %17\n"
#define I_MULLO(d) "v_mul_lo_u32 %" #d ", %" #d ", %16\n"
#define I_LSHLADD(d) "v_lshl_add_u32 %" #d ", %" #d ", 1, %16\n"
#define I_CNDMASK(d) "v_cndmask_b32 %" #d ", %" #d ", %16, vcc\n"
#define I_CMP(d) "v_cmp_lt_u32 vcc, %" #d ", %16\n"
#define I_SDWA(d) "v_add_u32_sdwa %" #d ", %" #d ", %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n"
#define I_DPP(d) "v_add_u32_dpp %" #d ", %16, %" #d " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_MOV(d) "v_mov_b32 %" #d ", %16\n"
#define I_FMA32(d) "v_fma_f32 %" #d ", %" #d ", %16, %17\n"
#define I_FMA64(d) "v_fma_f64 %" #d ", %" #d ", %16, %17\n"
#define I_AND(d) "v_and_b32 %" #d ", %" #d ", %16\n"
#define I_OR(d) "v_or_b32 %" #d ", %" #d ", %16\n"
#define I_SHL(d) "v_lshlrev_b32 %" #d ", 1, %" #d "\n"
#define I_SHR(d) "v_lshrrev_b32 %" #d ", 1, %" #d "\n"
#define I_SUB(d) "v_sub_u32 %" #d ", %" #d ", %16\n"
#define I_ADD3(d) "v_add3_u32 %" #d ", %" #d ", %16, %17\n"
#define I_ANDOR(d) "v_and_or_b32 %" #d ", %" #d ", %16, %17\n"
#define I_LSHLOR(d) "v_lshl_or_b32 %" #d ", %" #d ", 1, %16\n"
#define I_BFE(d) "v_bfe_u32 %" #d ", %" #d ", 1, 31\n"
#define I_MINI(d) "v_min_i32 %" #d ", %" #d ", %16\n"
#define I_MAXI(d) "v_max_i32 %" #d ", %" #d ", %17\n"
#define I_MIN3(d) "v_min3_u32 %" #d ", %" #d ", %16, %17\n"
#define I_MED3(d) "v_med3_i32 %" #d ", %" #d ", %16, %17\n"
#define I_PKADD(d) "v_pk_add_u16 %" #d ", %" #d ", %16\n"
#define I_PKSUB(d) "v_pk_sub_i16 %" #d ", %" #d ", %16\n"
#define I_SAD(d) "v_sad_u8 %" #d ", %16, %17, %" #d "\n"
#define I_BITOP3(d) "v_bitop3_b32 %" #d ", %" #d ", %16, %17 bitop3:0x96\n"
#define I_CNDS(d) "v_cndmask_b32 %" #d ", %" #d ", %16, s[20:21]\n"
#define I_ADDF(d) "v_add_f32 %" #d ", %" #d ", %16\n"
#define I_MULF(d) "v_mul_f32 %" #d ", %" #d ", %16\n"
#define I_CVTF(d) "v_cvt_f32_u32 %" #d ", %" #d "\n"
#define I_ADDCO(d) "v_add_co_u32 %" #d ", vcc, %" #d ", %16\n"
#define I_MINF(d) "v_min_f32 %" #d ", %" #d ", %16\n"
#define I_MAXF(d) "v_max_f32 %" #d ", %" #d ", %17\n"
#define I_MIN3F(d) "v_min3_f32 %" #d ", %" #d ", %16, %17\n"
#define I_MAX3F(d) "v_max3_f32 %" #d ", %" #d ", %16, %17\n"
#define I_MED3F(d) "v_med3_f32 %" #d ", %" #d ", %16, %17\n"
#define I_PKMINH(d) "v_pk_min_f16 %" #d ", %" #d ", %16\n"
#define I_PKMAXH(d) "v_pk_max_f16 %" #d ", %" #d ", %17\n"
#define I_PKFMAH(d) "v_pk_fma_f16 %" #d ", %" #d ", %16, %17\n"
#define I_PKADDH(d) "v_pk_add_f16 %" #d ", %" #d ", %16\n"
#define I_MINH(d) "v_min_f16 %" #d ", %" #d ", %16\n"
#define I_MAXH(d) "v_max_f16 %" #d ", %" #d ", %17\n"
#define I_MINU16(d) "v_min_u16 %" #d ", %" #d ", %16\n"
#define I_MAXU16(d) "v_max_u16 %" #d ", %" #d ", %17\n"
#define I_CVTUB(d) "v_cvt_f32_ubyte0 %" #d ", %" #d "\n"
#define I_SUBF(d) "v_sub_f32 %" #d ", %" #d ", %16\n"
#define I_MAX3U(d) "v_max3_u32 %" #d ", %" #d ", %16, %17\n"
#define I_PKMINU(d) "v_pk_min_u16 %" #d ", %" #d ", %16\n"
#define I_PKMAXU(d) "v_pk_max_u16 %" #d ", %" #d ", %17\n"
#define I_ADDC(d) "v_addc_co_u32 %" #d ", vcc, %" #d ", %16, vcc\n"
#define I_LSHLADD64(d) "v_lshl_add_u64 %" #d ", %" #d ", 1, %16\n"

// stamps[block] = {shader-clock ticks, 100 MHz reference ticks} of the block's first wavefront around its loop
template <int OP>
__global__ __launch_bounds__(256) void k_calib_valu(uint32_t *sink, unsigned long long *stamps, int trips) {
  extern __shared__ uint32_t calib_lds[];  // sized by the launch to fix the number of resident workgroups per CU; never touched
  const uint32_t t = threadIdx.x + blockIdx.x * 256u;
  unsigned long long t0 = 0, w0 = 0;
  if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); w0 = __builtin_amdgcn_s_memrealtime(); }
  uint32_t acc = 0;
  if constexpr (OP == CAL_FMA_F64) {
    double r[16], a = 1.0000001, b = 1e-9;
    for (int i = 0; i < 16; i++) r[i] = 1.0 + (double)(t + i) * 1e-6;
    for (int it = 0; it < trips; ++it) asm volatile(CAL_TRIP(I_FMA64) CAL_OPERANDS);
    double s = 0;
    for (int i = 0; i < 16; i++) s += r[i];
    acc = (uint32_t)__double2ll_rn(s);
  } else if constexpr (OP == CAL_LSHL_ADD_U64) {
    unsigned long long r[16], a = t | 1, b = 0;
    for (int i = 0; i < 16; i++) r[i] = t + i;
    for (int it = 0; it < trips; ++it) asm volatile(CAL_TRIP(I_LSHLADD64) CAL_OPERANDS);
    for (int i = 0; i < 16; i++) acc ^= (uint32_t)(r[i] ^ (r[i] >> 32));
    (void)b;
  } else if constexpr (OP == CAL_FMA_F32) {
    float r[16], a = 1.0000001f, b = 1e-9f;
    for (int i = 0; i < 16; i++) r[i] = 1.0f + (float)(t + i) * 1e-6f;
    for (int it = 0; it < trips; ++it) asm volatile(CAL_TRIP(I_FMA32) CAL_OPERANDS);
    float s = 0;
    for (int i = 0; i < 16; i++) s += r[i];
    acc = __float_as_uint(s);
  } else {
    uint32_t r[16], a = t * 2654435761u + 12345u, b = (t ^ 0x9e3779b9u) * 40503u + 7u;
    for (int i = 0; i < 16; i++) r[i] = (t + i) * 747796405u + 2891336453u;
    if constexpr (OP == CAL_PERM) b = 0x06010400u | ((t & 1) << 8);  // a valid byte selector
    for (int it = 0; it < trips; ++it) {
      if constexpr (OP == CAL_ADD_U32) asm volatile(CAL_TRIP(I_ADD) CAL_OPERANDS);
      if constexpr (OP == CAL_XOR_B32) asm volatile(CAL_TRIP(I_XOR) CAL_OPERANDS);
      if constexpr (OP == CAL_MINMAX_U32) asm volatile(CAL_TRIP2(I_MINU, I_MAXU) CAL_OPERANDS);
      if constexpr (OP == CAL_PK_MINMAX_I16) asm volatile(CAL_TRIP2(I_PKMIN, I_PKMAX) CAL_OPERANDS);
      if constexpr (OP == CAL_PK_MAD_U16) asm volatile(CAL_TRIP(I_PKMAD) CAL_OPERANDS);
      if constexpr (OP == CAL_BCNT) asm volatile(CAL_TRIP(I_BCNT) CAL_OPERANDS);
      if constexpr (OP == CAL_DOT4_U8) asm volatile(CAL_TRIP(I_DOT4) CAL_OPERANDS);
      if constexpr (OP == CAL_DOT2_U16) asm volatile(CAL_TRIP(I_DOT2) CAL_OPERANDS);
      if constexpr (OP == CAL_PERM) asm volatile(CAL_TRIP(I_PERM) CAL_OPERANDS);
      if constexpr (OP == CAL_ALIGNBYTE) asm volatile(CAL_TRIP(I_ALIGNB) CAL_OPERANDS);
      if constexpr (OP == CAL_MUL_U24) asm volatile(CAL_TRIP(I_MUL24) CAL_OPERANDS);
      if constexpr (OP == CAL_MAD_U24) asm volatile(CAL_TRIP(I_MAD24) CAL_OPERANDS);
      if constexpr (OP == CAL_MUL_LO_U32) asm volatile(CAL_TRIP(I_MULLO) CAL_OPERANDS);
      if constexpr (OP == CAL_LSHL_ADD_U32) asm volatile(CAL_TRIP(I_LSHLADD) CAL_OPERANDS);
      if constexpr (OP == CAL_CNDMASK) asm volatile("v_cmp_lt_u32 vcc, %16, %17\n" CAL_TRIP(I_CNDMASK) CAL_OPERANDS);
      if constexpr (OP == CAL_CMP) asm volatile(CAL_TRIP(I_CMP) CAL_OPERANDS);
      if constexpr (OP == CAL_SDWA) asm volatile(CAL_TRIP(I_SDWA) CAL_OPERANDS);
      if constexpr (OP == CAL_DPP) asm volatile(CAL_TRIP(I_DPP) CAL_OPERANDS);
      if constexpr (OP == CAL_MOV) asm volatile(CAL_TRIP(I_MOV) CAL_OPERANDS);
      if constexpr (OP == CAL_AND_B32) asm volatile(CAL_TRIP(I_AND) CAL_OPERANDS);
      if constexpr (OP == CAL_OR_B32) asm volatile(CAL_TRIP(I_OR) CAL_OPERANDS);
      if constexpr (OP == CAL_LSHLREV) asm volatile(CAL_TRIP(I_SHL) CAL_OPERANDS);
      if constexpr (OP == CAL_LSHRREV) asm volatile(CAL_TRIP(I_SHR) CAL_OPERANDS);
      if constexpr (OP == CAL_SUB_U32) asm volatile(CAL_TRIP(I_SUB) CAL_OPERANDS);
      if constexpr (OP == CAL_ADD3_U32) asm volatile(CAL_TRIP(I_ADD3) CAL_OPERANDS);
      if constexpr (OP == CAL_AND_OR) asm volatile(CAL_TRIP(I_ANDOR) CAL_OPERANDS);
      if constexpr (OP == CAL_LSHL_OR) asm volatile(CAL_TRIP(I_LSHLOR) CAL_OPERANDS);
      if constexpr (OP == CAL_BFE_U32) asm volatile(CAL_TRIP(I_BFE) CAL_OPERANDS);
      if constexpr (OP == CAL_MINMAX_I32) asm volatile(CAL_TRIP2(I_MINI, I_MAXI) CAL_OPERANDS);
      if constexpr (OP == CAL_MIN3_U32) asm volatile(CAL_TRIP(I_MIN3) CAL_OPERANDS);
      if constexpr (OP == CAL_MED3_I32) asm volatile(CAL_TRIP(I_MED3) CAL_OPERANDS);
      if constexpr (OP == CAL_PK_ADD_U16) asm volatile(CAL_TRIP(I_PKADD) CAL_OPERANDS);
      if constexpr (OP == CAL_PK_SUB_I16) asm volatile(CAL_TRIP(I_PKSUB) CAL_OPERANDS);
      if constexpr (OP == CAL_SAD_U8) asm volatile(CAL_TRIP(I_SAD) CAL_OPERANDS);
      if constexpr (OP == CAL_BITOP3) asm volatile(CAL_TRIP(I_BITOP3) CAL_OPERANDS);
      if constexpr (OP == CAL_CNDMASK_SGPR) asm volatile("s_mov_b32 s20, 0x55555555\ns_mov_b32 s21, 0x55555555\n" CAL_TRIP(I_CNDS) CAL_OPERANDS, "s20", "s21");
      if constexpr (OP == CAL_CNDMASK_VCC_S) asm volatile("s_mov_b32 vcc_lo, 0x55555555\ns_mov_b32 vcc_hi, 0x55555555\n" CAL_TRIP(I_CNDMASK) CAL_OPERANDS);
      if constexpr (OP == CAL_ADD_F32) asm volatile(CAL_TRIP(I_ADDF) CAL_OPERANDS);
      if constexpr (OP == CAL_MUL_F32) asm volatile(CAL_TRIP(I_MULF) CAL_OPERANDS);
      if constexpr (OP == CAL_CVT_F32_U32) asm volatile(CAL_TRIP(I_CVTF) CAL_OPERANDS);
      if constexpr (OP == CAL_ADD_CO_U32) asm volatile(CAL_TRIP(I_ADDCO) CAL_OPERANDS);
      if constexpr (OP == CAL_MINMAX_F32) asm volatile(CAL_TRIP2(I_MINF, I_MAXF) CAL_OPERANDS);
      if constexpr (OP == CAL_MIN3_F32) asm volatile(CAL_TRIP(I_MIN3F) CAL_OPERANDS);
      if constexpr (OP == CAL_MAX3_F32) asm volatile(CAL_TRIP(I_MAX3F) CAL_OPERANDS);
      if constexpr (OP == CAL_MED3_F32) asm volatile(CAL_TRIP(I_MED3F) CAL_OPERANDS);
      if constexpr (OP == CAL_PK_MINMAX_F16) asm volatile(CAL_TRIP2(I_PKMINH, I_PKMAXH) CAL_OPERANDS);
      if constexpr (OP == CAL_PK_FMA_F16) asm volatile(CAL_TRIP(I_PKFMAH) CAL_OPERANDS);
      if constexpr (OP == CAL_PK_ADD_F16) asm volatile(CAL_TRIP(I_PKADDH) CAL_OPERANDS);
      if constexpr (OP == CAL_MINMAX_F16) asm volatile(CAL_TRIP2(I_MINH, I_MAXH) CAL_OPERANDS);
      if constexpr (OP == CAL_MINMAX_U16) asm volatile(CAL_TRIP2(I_MINU16, I_MAXU16) CAL_OPERANDS);
      if constexpr (OP == CAL_CVT_UBYTE) asm volatile(CAL_TRIP(I_CVTUB) CAL_OPERANDS);
      if constexpr (OP == CAL_SUB_F32) asm volatile(CAL_TRIP(I_SUBF) CAL_OPERANDS);
      if constexpr (OP == CAL_MAX3_U32) asm volatile(CAL_TRIP(I_MAX3U) CAL_OPERANDS);
      if constexpr (OP == CAL_PK_MINMAX_U16) asm volatile(CAL_TRIP2(I_PKMINU, I_PKMAXU) CAL_OPERANDS);
      if constexpr (OP == CAL_ADDC) asm volatile(CAL_TRIP(I_ADDC) CAL_OPERANDS);
      if constexpr (OP == CAL_SALU_ADD)
        asm volatile(
#define S8 "s_add_u32 s20, s20, 1\ns_add_u32 s21, s21, 1\ns_add_u32 s22, s22, 1\ns_add_u32 s23, s23, 1\ns_add_u32 s24, s24, 1\ns_add_u32 s25, s25, 1\ns_add_u32 s26, s26, 1\ns_add_u32 s27, s27, 1\n"
            S8 S8 S8 S8 S8 S8 S8 S8 S8 S8 S8 S8 S8 S8 S8 S8
#undef S8
            CAL_OPERANDS, "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
      if constexpr (OP == CAL_SALU_AND64)
        asm volatile(
#define S8 "s_and_b64 s[20:21], s[20:21], exec\ns_bcnt1_i32_b64 s24, s[20:21]\ns_and_b64 s[22:23], s[22:23], exec\ns_bcnt1_i32_b64 s25, s[22:23]\ns_and_b64 s[26:27], s[26:27], exec\ns_bcnt1_i32_b64 s28, s[26:27]\ns_and_b64 s[30:31], s[30:31], exec\ns_bcnt1_i32_b64 s29, s[30:31]\n"
            S8 S8 S8 S8 S8 S8 S8 S8 S8 S8 S8 S8 S8 S8 S8 S8
#undef S8
            CAL_OPERANDS, "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "scc");
      if constexpr (OP == CAL_DS_READ_U8) {
        // 128 byte gathers per trip from the first 16 KiB of the workgroup's LDS (contents irrelevant), 16 in flight
        const uint32_t base = (a & 0x3fffu);
        asm volatile(
#define D16 "ds_read_u8 %0, %16\nds_read_u8 %1, %16 offset:80\nds_read_u8 %2, %16 offset:160\nds_read_u8 %3, %16 offset:240\nds_read_u8 %4, %16 offset:3\nds_read_u8 %5, %16 offset:83\nds_read_u8 %6, %16 offset:163\nds_read_u8 %7, %16 offset:243\nds_read_u8 %8, %16 offset:320\nds_read_u8 %9, %16 offset:400\nds_read_u8 %10, %16 offset:480\nds_read_u8 %11, %16 offset:560\nds_read_u8 %12, %16 offset:323\nds_read_u8 %13, %16 offset:403\nds_read_u8 %14, %16 offset:483\nds_read_u8 %15, %16 offset:563\ns_waitcnt lgkmcnt(0)\n"
            D16 D16 D16 D16 D16 D16 D16 D16
#undef D16
            : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), "=v"(r[8]), "=v"(r[9]), "=v"(r[10]),
              "=v"(r[11]), "=v"(r[12]), "=v"(r[13]), "=v"(r[14]), "=v"(r[15])
            : "v"(base), "v"(b)
            : "memory");
      }
      if constexpr (OP == CAL_XOR_SGPR) asm volatile("s_mov_b32 s20, 0x5bd1e995\n" CAL_TRIP(I_XORS) CAL_OPERANDS, "s20");
      if constexpr (OP == CAL_BCNT_SGPR) asm volatile("s_mov_b32 s20, 0x5bd1e995\n" CAL_TRIP(I_BCNTS) CAL_OPERANDS, "s20");
      if constexpr (OP == CAL_FAST_MIX)
        asm volatile(CAL_R16(I_PKMIN) CAL_R16(I_PKMIN) CAL_R16(I_PKMAX) CAL_R16(I_PKMIN) CAL_R16(I_PKMAD) CAL_R16(I_PKMIN) CAL_R16(I_ADD)
                         CAL_R16(I_ADD) CAL_OPERANDS);
    }
    for (int i = 0; i < 16; i++) acc ^= r[i];
  }
  if (threadIdx.x == 0) {
    asm volatile("" ::"v"(acc));  // the loop's results are complete before the closing stamp is taken
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    stamps[2 * blockIdx.x] = t1 - t0;
    stamps[2 * blockIdx.x + 1] = w1 - w0;
  }
  if (acc == 0x5bd1e995u && trips < 0) sink[t] = acc;  // never true for trips >= 0: keeps the chains observable
}

namespace {
template <int OP>
hipError_t calib_launch(int grid, size_t lds, uint32_t *sink, unsigned long long *stamps, int trips) {
  hipError_t e = hipFuncSetAttribute((const void *)k_calib_valu<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_calib_valu<OP>, dim3(grid), dim3(256), lds, (hipStream_t)0, sink, stamps, trips);
  return hipGetLastError();
}
template <int OP>
hipError_t calib_dispatch(int op, int grid, size_t lds, uint32_t *sink, unsigned long long *stamps, int trips) {
  if (op == OP) return calib_launch<OP>(grid, lds, sink, stamps, trips);
  if constexpr (OP + 1 < CAL_NUM_OPS) return calib_dispatch<OP + 1>(op, grid, lds, sink, stamps, trips);
  return hipErrorInvalidValue;
}
}  // namespace

