// ita_device.h -- device-side helpers shared by the gfx950 kernels.
//
// Conventions used by every int8 kernel in this directory
// -------------------------------------------------------
// * All int8 GEMMs are "NT": C[m][n] = sum_k A[m][k] * Bt[n][k], both operands k-contiguous,
//   so one MFMA fragment is 16 contiguous bytes of one row.  The integer sum does not care
//   which k a fragment byte really is, only that A and Bt use the same slot -> k mapping,
//   which lets the attention kernel feed softmax outputs straight from registers (see
//   ita_int8_kernels.h).
// * int8 operand matrices that live in LDS are stored "chunk-major": [k/16][row][16 bytes].
//   A wave's ds_read_b128 of one fragment column then touches 16 consecutive 16-byte slots per
//   lane group -> conflict-free for both the 32x32x32 and the 16x16x64 MFMA access patterns,
//   with no padding.
// * v_mfma_i32_32x32x32_i8 : A lane l -> row l&31, k-slot group l>>5 (16 bytes);
//                            C lane l -> col l&31, row (i&3) + 8*(i>>2) + 4*(l>>5), i = 0..15
//   v_mfma_i32_16x16x64_i8 : A lane l -> row l&15, k-slot group l>>4 (16 bytes);
//                            C lane l -> col l&15, row 4*(l>>4) + i, i = 0..3
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ITA_MAGIC_F 12582912.0f   /* 1.5 * 2^23: adding it rounds to nearest-even integer */
#define ITA_MAGIC_I 0x4B400000

// byte offset of (row, k-byte) in a chunk-major int8 matrix with ROWS rows
__device__ __forceinline__ int cm_off(int row, int kb, int rows) {
  return ((((kb >> 4) * rows) + row) << 4) | (kb & 15);
}

// nnq requantisation: clamp(rne(float(acc) * mult), lo, 127).  The clamp is applied before the
// rounding (same result, the bounds are integers); the rounding is the fp32 add of 1.5*2^23,
// whose mantissa then holds the two's-complement integer: returns those bits.
__device__ __forceinline__ unsigned rq_bits(int acc, float mult, float lo = -128.0f) {
  float f = (float)acc * mult;
  f = __builtin_amdgcn_fmed3f(f, lo, 127.0f);
  f = f + ITA_MAGIC_F;
  return __float_as_uint(f);
}
__device__ __forceinline__ unsigned q_bits(float x, float inv_scale) {   // torch.quantize_per_tensor
  float f = x * inv_scale;
  f = __builtin_amdgcn_fmed3f(f, -128.0f, 127.0f);
  f = f + ITA_MAGIC_F;
  return __float_as_uint(f);
}
__device__ __forceinline__ int bits_to_int(unsigned b) { return (int)b - ITA_MAGIC_I; }

// Round sixteen already-clamped floats to nearest-even integers and pack their low bytes into
// four dwords: pk[g] byte b = rne(f[4g + b]).  The rounding is the fp32 add of 1.5*2^23 (the
// mantissa then holds the two's-complement integer); SDWA writes the low byte of each sum straight
// into byte b of the destination, so the add IS the pack: 16 VALU instructions for 16 bytes instead
// of 16 adds + 20 and/shift/or.  gfx950 hazard (measured, tools/... sdwa micro test): a VALU that
// reads a VGPR written by the immediately preceding dst_sel instruction sees stale bytes, and
// inline asm is not padded by hipcc -- so the four destinations are interleaved (3 independent
// instructions between two writes of one register) and the statement ends with one wait state.
__device__ __forceinline__ void round_pack16(const float (&f)[16], unsigned (&pk)[4]) {
  const float mg = ITA_MAGIC_F;
  asm(
      "v_add_f32_sdwa %0, %4, %20 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %1, %8, %20 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %2, %12, %20 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %3, %16, %20 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %0, %5, %20 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %1, %9, %20 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %2, %13, %20 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %3, %17, %20 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %0, %6, %20 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %1, %10, %20 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %2, %14, %20 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %3, %18, %20 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %0, %7, %20 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %1, %11, %20 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %2, %15, %20 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %3, %19, %20 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "s_nop 0"
      : "=&v"(pk[0]), "=&v"(pk[1]), "=&v"(pk[2]), "=&v"(pk[3])
      : "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(f[6]), "v"(f[7]), "v"(f[8]), "v"(f[9]),
        "v"(f[10]), "v"(f[11]), "v"(f[12]), "v"(f[13]), "v"(f[14]), "v"(f[15]), "v"(mg));
}
__device__ __forceinline__ float rq_clamped(int acc, float mult, float lo) {
  return __builtin_amdgcn_fmed3f((float)acc * mult, lo, 127.0f);
}
// requantise + pack a 16-accumulator fragment: pk[g] = bytes of acc[4g .. 4g+3]
typedef float f32x2 __attribute__((ext_vector_type(2)));
// f[i] = clamp(float(acc[i]) * mult, lo, 127) for N accumulators.  Non-packed VALU issues one
// wave-instruction per 4 cycles per SIMD on gfx950; v_pk_mul_f32 does two multiplies in that slot, so the
// scale step is written on float2 values.
template <int N, typename ACC>
__device__ __forceinline__ void scale_clamp(const ACC& acc, float mult, float lo, float (&f)[N]) {
#ifdef ITA_SCALAR_SCALE
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float v = (float)acc[i];
    asm("v_mul_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(mult));   // asm: keeps the SLP vectoriser from re-packing it
    f[i] = __builtin_amdgcn_fmed3f(v, lo, 127.0f);
  }
#else
  const f32x2 m2 = {mult, mult};
#pragma unroll
  for (int i = 0; i < N; i += 2) {
    f32x2 v = {(float)acc[i], (float)acc[i + 1]};
    v = v * m2;
    f[i] = __builtin_amdgcn_fmed3f(v.x, lo, 127.0f);
    f[i + 1] = __builtin_amdgcn_fmed3f(v.y, lo, 127.0f);
  }
#endif
}
// The same for accumulators that were started at ITA_ACC_BIAS (+ bias) instead of (bias): while |sum| < 2^22 the
// int32 bit pattern 0x4B400000 + sum IS the float 1.5 * 2^23 + sum, so the int -> float conversion becomes an exact
// float subtraction.  On gfx950 v_sub_f32 / v_mul_f32 issue at twice the rate of v_cvt_f32_i32 / v_pk_mul_f32 when two
// waves share a SIMD (tools/microbench/valu_ops.hip).
#define ITA_ACC_BIAS 0x4B400000
template <int N, typename ACC>
__device__ __forceinline__ void scale_clamp_b(const ACC& acc, float mult, float lo, float (&f)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float x = __int_as_float(acc[i]) - ITA_MAGIC_F;
    f[i] = __builtin_amdgcn_fmed3f(x * mult, lo, 127.0f);
  }
}
// ---- fixed-schedule requantisation of biased accumulators (ITA_ACC_BIAS).  What limits the encoder kernel is the
// NUMBER of VALU instructions a SIMD issues (measured: 4.0 cycles of SQ_ACTIVE_INST_VALU per instruction whatever its
// class, two waves per SIMD), so the steps that exist in packed form run packed -- v_pk_add_f32 / v_pk_mul_f32 handle two
// values per instruction -- and every block is one asm statement in an order where consecutive instructions never depend
// on each other (left to itself hipcc funnels all values through one temporary, and a dependent VALU instruction
// issues ~1.7x slower).  mult / lo / scale must be wave-uniform.
// Hazard: hipcc pads MFMA -> VALU read wait states for its own instructions, not for inline asm; an accumulator may come
// from a 4-pass MFMA issued immediately before (7-8 wait states on gfx950), so the first block opens with ten.
template <int NP>   // NP register pairs: x = (x - 1.5 * 2^23) * mult
__device__ __forceinline__ void pk_unbias_scale(f32x2 (&x)[NP], float mult) {
  static_assert(NP == 8 || NP == 4, "");
  const f32x2 c2 = {-ITA_MAGIC_F, -ITA_MAGIC_F}, m2 = {mult, mult};
  if constexpr (NP == 8) {
    asm("s_nop 7\n\ts_nop 1\n\t"
      "v_pk_add_f32 %0, %0, %8\n\t"
      "v_pk_add_f32 %1, %1, %8\n\t"
      "v_pk_add_f32 %2, %2, %8\n\t"
      "v_pk_add_f32 %3, %3, %8\n\t"
      "v_pk_add_f32 %4, %4, %8\n\t"
      "v_pk_add_f32 %5, %5, %8\n\t"
      "v_pk_add_f32 %6, %6, %8\n\t"
      "v_pk_add_f32 %7, %7, %8\n\t"
      "v_pk_mul_f32 %0, %0, %9\n\t"
      "v_pk_mul_f32 %1, %1, %9\n\t"
      "v_pk_mul_f32 %2, %2, %9\n\t"
      "v_pk_mul_f32 %3, %3, %9\n\t"
      "v_pk_mul_f32 %4, %4, %9\n\t"
      "v_pk_mul_f32 %5, %5, %9\n\t"
      "v_pk_mul_f32 %6, %6, %9\n\t"
      "v_pk_mul_f32 %7, %7, %9"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
        : "s"(c2), "s"(m2));
  } else {
    asm("s_nop 7\n\ts_nop 1\n\t"
      "v_pk_add_f32 %0, %0, %4\n\t"
      "v_pk_add_f32 %1, %1, %4\n\t"
      "v_pk_add_f32 %2, %2, %4\n\t"
      "v_pk_add_f32 %3, %3, %4\n\t"
      "v_pk_mul_f32 %0, %0, %5\n\t"
      "v_pk_mul_f32 %1, %1, %5\n\t"
      "v_pk_mul_f32 %2, %2, %5\n\t"
      "v_pk_mul_f32 %3, %3, %5"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])
        : "s"(c2), "s"(m2));
  }
}
// clamp sixteen floats, round them to nearest-even integers and pack their low bytes: pk[g] byte b = rne(clamp(f[4g+b]))
__device__ __forceinline__ i32x4 clamp_round_pack16(float (&f)[16], float lo) {
  unsigned p0, p1, p2, p3;
  const float hi = 127.0f, mg = ITA_MAGIC_F;
  asm(
      "v_med3_f32 %4, %4, %20, %21\n\t"
      "v_med3_f32 %5, %5, %20, %21\n\t"
      "v_med3_f32 %6, %6, %20, %21\n\t"
      "v_med3_f32 %7, %7, %20, %21\n\t"
      "v_med3_f32 %8, %8, %20, %21\n\t"
      "v_med3_f32 %9, %9, %20, %21\n\t"
      "v_med3_f32 %10, %10, %20, %21\n\t"
      "v_med3_f32 %11, %11, %20, %21\n\t"
      "v_med3_f32 %12, %12, %20, %21\n\t"
      "v_med3_f32 %13, %13, %20, %21\n\t"
      "v_med3_f32 %14, %14, %20, %21\n\t"
      "v_med3_f32 %15, %15, %20, %21\n\t"
      "v_med3_f32 %16, %16, %20, %21\n\t"
      "v_med3_f32 %17, %17, %20, %21\n\t"
      "v_med3_f32 %18, %18, %20, %21\n\t"
      "v_med3_f32 %19, %19, %20, %21\n\t"
      "v_add_f32_sdwa %0, %4, %22 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %1, %8, %22 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %2, %12, %22 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %3, %16, %22 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %0, %5, %22 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %1, %9, %22 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %2, %13, %22 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %3, %17, %22 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %0, %6, %22 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %1, %10, %22 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %2, %14, %22 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %3, %18, %22 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %0, %7, %22 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %1, %11, %22 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %2, %15, %22 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "v_add_f32_sdwa %3, %19, %22 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
      "s_nop 0"
      : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]),
        "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]), "+v"(f[12]), "+v"(f[13]), "+v"(f[14]),
        "+v"(f[15])
      : "s"(lo), "v"(hi), "v"(mg));
  return (i32x4){(int)p0, (int)p1, (int)p2, (int)p3};
}
// the same arithmetic left to the compiler's scheduler (it interleaves the VALU work with the MFMAs and LDS reads
// around it; only the byte-insert block is fixed): experiment switch ITA_RQ_STYLE=1
__device__ __forceinline__ i32x4 rq_pack16_c(const i32x4 (&acc)[4], float mult, float lo) {
  const f32x2 c2 = {-ITA_MAGIC_F, -ITA_MAGIC_F}, m2 = {mult, mult};
  float f[16];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    f32x2 v = {__int_as_float(acc[k >> 1][2 * (k & 1)]), __int_as_float(acc[k >> 1][2 * (k & 1) + 1])};
    v = (v + c2) * m2;
    f[2 * k] = __builtin_amdgcn_fmed3f(v.x, lo, 127.0f);
    f[2 * k + 1] = __builtin_amdgcn_fmed3f(v.y, lo, 127.0f);
  }
  unsigned p4[4];
  round_pack16(f, p4);
  return (i32x4){(int)p4[0], (int)p4[1], (int)p4[2], (int)p4[3]};
}
__device__ __forceinline__ void dq16_c(const i32x4 (&acc)[4], float mult, float scale, float (&d)[16]) {
  const f32x2 c2 = {-ITA_MAGIC_F, -ITA_MAGIC_F}, m2 = {mult, mult}, s2 = {scale, scale};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    f32x2 v = {__int_as_float(acc[k >> 1][2 * (k & 1)]), __int_as_float(acc[k >> 1][2 * (k & 1) + 1])};
    v = (v + c2) * m2;
    f32x2 r = {__builtin_rintf(__builtin_amdgcn_fmed3f(v.x, -128.0f, 127.0f)), __builtin_rintf(__builtin_amdgcn_fmed3f(v.y, -128.0f, 127.0f))};
    r = r * s2;
    d[2 * k] = r.x; d[2 * k + 1] = r.y;
  }
}
__device__ __forceinline__ void lg8_c(const i32x4 (&acc)[2], float mult, unsigned (&bits)[8]) {
  const f32x2 c2 = {-ITA_MAGIC_F, -ITA_MAGIC_F}, m2 = {mult, mult}, mg2 = {ITA_MAGIC_F, ITA_MAGIC_F};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f32x2 v = {__int_as_float(acc[k >> 1][2 * (k & 1)]), __int_as_float(acc[k >> 1][2 * (k & 1) + 1])};
    v = (v + c2) * m2;
    f32x2 r = {__builtin_amdgcn_fmed3f(v.x, -128.0f, 127.0f), __builtin_amdgcn_fmed3f(v.y, -128.0f, 127.0f)};
    r = r + mg2;
    bits[2 * k] = __float_as_uint(r.x); bits[2 * k + 1] = __float_as_uint(r.y);
  }
}
// requantise + pack sixteen biased accumulators (four 16x16 tiles): 8 + 8 + 16 + 16 VALU instructions
__device__ __forceinline__ i32x4 rq_pack16_b(const i32x4 (&acc)[4], float mult, float lo) {
  f32x2 x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = (f32x2){__int_as_float(acc[k >> 1][2 * (k & 1)]), __int_as_float(acc[k >> 1][2 * (k & 1) + 1])};
  pk_unbias_scale<8>(x, mult);
  float f[16];
#pragma unroll
  for (int k = 0; k < 8; ++k) { f[2 * k] = x[k].x; f[2 * k + 1] = x[k].y; }
  return clamp_round_pack16(f, lo);
}
// block output: d[4t+i] = float(int8 code) * scale for sixteen biased accumulators: 8 + 8 + 16 + 16 + 8
__device__ __forceinline__ void dq16_b(const i32x4 (&acc)[4], float mult, float scale, float (&d)[16]) {
  f32x2 x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = (f32x2){__int_as_float(acc[k >> 1][2 * (k & 1)]), __int_as_float(acc[k >> 1][2 * (k & 1) + 1])};
  pk_unbias_scale<8>(x, mult);
  float f[16];
#pragma unroll
  for (int k = 0; k < 8; ++k) { f[2 * k] = x[k].x; f[2 * k + 1] = x[k].y; }
  const float lo = -128.0f, hi = 127.0f;
  asm(
      "v_med3_f32 %0, %0, %16, %17\n\t"
      "v_med3_f32 %1, %1, %16, %17\n\t"
      "v_med3_f32 %2, %2, %16, %17\n\t"
      "v_med3_f32 %3, %3, %16, %17\n\t"
      "v_med3_f32 %4, %4, %16, %17\n\t"
      "v_med3_f32 %5, %5, %16, %17\n\t"
      "v_med3_f32 %6, %6, %16, %17\n\t"
      "v_med3_f32 %7, %7, %16, %17\n\t"
      "v_med3_f32 %8, %8, %16, %17\n\t"
      "v_med3_f32 %9, %9, %16, %17\n\t"
      "v_med3_f32 %10, %10, %16, %17\n\t"
      "v_med3_f32 %11, %11, %16, %17\n\t"
      "v_med3_f32 %12, %12, %16, %17\n\t"
      "v_med3_f32 %13, %13, %16, %17\n\t"
      "v_med3_f32 %14, %14, %16, %17\n\t"
      "v_med3_f32 %15, %15, %16, %17\n\t"
      "v_rndne_f32 %0, %0\n\t"
      "v_rndne_f32 %1, %1\n\t"
      "v_rndne_f32 %2, %2\n\t"
      "v_rndne_f32 %3, %3\n\t"
      "v_rndne_f32 %4, %4\n\t"
      "v_rndne_f32 %5, %5\n\t"
      "v_rndne_f32 %6, %6\n\t"
      "v_rndne_f32 %7, %7\n\t"
      "v_rndne_f32 %8, %8\n\t"
      "v_rndne_f32 %9, %9\n\t"
      "v_rndne_f32 %10, %10\n\t"
      "v_rndne_f32 %11, %11\n\t"
      "v_rndne_f32 %12, %12\n\t"
      "v_rndne_f32 %13, %13\n\t"
      "v_rndne_f32 %14, %14\n\t"
      "v_rndne_f32 %15, %15"
      : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]),
        "+v"(f[10]), "+v"(f[11]), "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15])
      : "s"(lo), "v"(hi));
  const f32x2 s2 = {scale, scale};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const f32x2 v = (f32x2){f[2 * k], f[2 * k + 1]} * s2;
    d[2 * k] = v.x; d[2 * k + 1] = v.y;
  }
}
// logits: eight biased accumulators (two key tiles) -> float bit patterns whose low 16 bits hold rne(clamp(acc * mult))
__device__ __forceinline__ void lg8_b(const i32x4 (&acc)[2], float mult, unsigned (&bits)[8]) {
  f32x2 x[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) x[k] = (f32x2){__int_as_float(acc[k >> 1][2 * (k & 1)]), __int_as_float(acc[k >> 1][2 * (k & 1) + 1])};
  pk_unbias_scale<4>(x, mult);
  float f[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) { f[2 * k] = x[k].x; f[2 * k + 1] = x[k].y; }
  const float lo = -128.0f, hi = 127.0f;
  asm(
      "v_med3_f32 %0, %0, %8, %9\n\t"
      "v_med3_f32 %1, %1, %8, %9\n\t"
      "v_med3_f32 %2, %2, %8, %9\n\t"
      "v_med3_f32 %3, %3, %8, %9\n\t"
      "v_med3_f32 %4, %4, %8, %9\n\t"
      "v_med3_f32 %5, %5, %8, %9\n\t"
      "v_med3_f32 %6, %6, %8, %9\n\t"
      "v_med3_f32 %7, %7, %8, %9"
      : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])
      : "s"(lo), "v"(hi));
  const f32x2 mg2 = {ITA_MAGIC_F, ITA_MAGIC_F};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f32x2 v = (f32x2){f[2 * k], f[2 * k + 1]} + mg2;
    bits[2 * k] = __float_as_uint(v.x); bits[2 * k + 1] = __float_as_uint(v.y);
  }
}
// ---- round-3 requantisation of biased accumulators: the clamp moves behind the rounding, into the integer domain, where
// ONE instruction saturates two values (v_sat_pk_u8_i16) -- 2.75 VALU instructions per value instead of 3.0, and 2.25 where a
// load-time proof allows the single-rounding form:
//   exact : x = acc_bits - 1.5*2^23 (v_pk_add_f32, exact) ; y = fl(x * m) (v_pk_mul_f32: the reference's first rounding) ;
//           t = y + (1.5*2^23 + 128) (v_pk_add_f32: rounds to nearest-even INTEGER, the reference's second rounding; the low
//           16 bits of t's pattern are then r + 128 in two's complement, r = rne(y), provided |r + 128| < 2^15 -- checked per
//           weight row at load time, stream_range_ok) ; pairs of low halves -> one dword (v_perm_b32) ; u8 saturation of
//           both halves = clamp(r, -128, 127) + 128 (v_sat_pk_u8_i16, SDWA places the two bytes) ; ^ 0x80808080 per dword.
//   fast  : t = fma(x, m, 1.5*2^23 + 128) in ONE v_pk_fma_f32 -- a single rounding of the exact product.  It differs from
//           the reference's two roundings only for accumulator values whose exact product lies within half an ulp of a
//           rounding tie without being one; ita_load_weights enumerates every accumulator value of the clamp range for the
//           site's multiplier (fast_site_ok) and enables this form only when none differs.
//   relu  : the exact form with the lower clamp moved into the multiply: y/256 = fl(x * (m/256)) with the VOP3P clamp
//           modifier ([0, 1] <-> y in [0, 256]), t = y/256 + (1.5*2^23 + 128)/256 (ulp 2^-8: same integer rounding, same low
//           bits), the u8 saturation then gives min(r, 127) + 128.
#define ITA_MAGIC128_F 12583040.0f        /* 1.5 * 2^23 + 128 */
#define ITA_MAGIC32K_F 12615680.0f        /* 1.5 * 2^23 + 32768: low 16 bits = r + 32768 = r ^ 0x8000 (order-preserving u16) */
enum { ITA_RQ_EXACT = 0, ITA_RQ_FAST = 1, ITA_RQ_RELU = 2 };

template <int NP, int MODE>   // NP register pairs: acc bit patterns -> float patterns whose low 16 bits hold rne(acc * mult) + bias
__device__ __forceinline__ void pk_requant_round(f32x2 (&x)[NP], float mult, float magic) {
  static_assert(NP == 8 || NP == 4, "");
  const f32x2 c2 = {-ITA_MAGIC_F, -ITA_MAGIC_F};
  if constexpr (MODE == ITA_RQ_FAST) {
    // (an instruction reads at most one SGPR operand: the magic constant is copied into a VGPR pair inside the block -- as a
    //  loop-invariant VGPR pair the compiler would hoist it out of the frame loop and, at 256 registers, spill it)
    const f32x2 m2 = {mult, mult}, g2s = {magic, magic};
    f32x2 g2;
    if constexpr (NP == 8) {
      asm("s_nop 7\n\ts_nop 1\n\t"
        "v_pk_mov_b32 %8, %11, %11\n\t"
        "v_pk_add_f32 %0, %0, %9\n\t"
        "v_pk_add_f32 %1, %1, %9\n\t"
        "v_pk_add_f32 %2, %2, %9\n\t"
        "v_pk_add_f32 %3, %3, %9\n\t"
        "v_pk_add_f32 %4, %4, %9\n\t"
        "v_pk_add_f32 %5, %5, %9\n\t"
        "v_pk_add_f32 %6, %6, %9\n\t"
        "v_pk_add_f32 %7, %7, %9\n\t"
        "v_pk_fma_f32 %0, %0, %10, %8\n\t"
        "v_pk_fma_f32 %1, %1, %10, %8\n\t"
        "v_pk_fma_f32 %2, %2, %10, %8\n\t"
        "v_pk_fma_f32 %3, %3, %10, %8\n\t"
        "v_pk_fma_f32 %4, %4, %10, %8\n\t"
        "v_pk_fma_f32 %5, %5, %10, %8\n\t"
        "v_pk_fma_f32 %6, %6, %10, %8\n\t"
        "v_pk_fma_f32 %7, %7, %10, %8"
          : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "=&v"(g2)
          : "s"(c2), "s"(m2), "s"(g2s));
    } else {
      asm("s_nop 7\n\ts_nop 1\n\t"
        "v_pk_mov_b32 %4, %7, %7\n\t"
        "v_pk_add_f32 %0, %0, %5\n\t"
        "v_pk_add_f32 %1, %1, %5\n\t"
        "v_pk_add_f32 %2, %2, %5\n\t"
        "v_pk_add_f32 %3, %3, %5\n\t"
        "v_pk_fma_f32 %0, %0, %6, %4\n\t"
        "v_pk_fma_f32 %1, %1, %6, %4\n\t"
        "v_pk_fma_f32 %2, %2, %6, %4\n\t"
        "v_pk_fma_f32 %3, %3, %6, %4"
          : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "=&v"(g2)
          : "s"(c2), "s"(m2), "s"(g2s));
    }
  } else {
    // (RELU: the caller passes mult / 256 and magic / 256 -- both exact power-of-two scalings)
    const f32x2 m2 = {mult, mult}, g2 = {magic, magic};
    if constexpr (NP == 8) {
      if constexpr (MODE == ITA_RQ_RELU) {
        asm("s_nop 7\n\ts_nop 1\n\t"
          "v_pk_add_f32 %0, %0, %8\n\t"
          "v_pk_add_f32 %1, %1, %8\n\t"
          "v_pk_add_f32 %2, %2, %8\n\t"
          "v_pk_add_f32 %3, %3, %8\n\t"
          "v_pk_add_f32 %4, %4, %8\n\t"
          "v_pk_add_f32 %5, %5, %8\n\t"
          "v_pk_add_f32 %6, %6, %8\n\t"
          "v_pk_add_f32 %7, %7, %8\n\t"
          "v_pk_mul_f32 %0, %0, %9 clamp\n\t"
          "v_pk_mul_f32 %1, %1, %9 clamp\n\t"
          "v_pk_mul_f32 %2, %2, %9 clamp\n\t"
          "v_pk_mul_f32 %3, %3, %9 clamp\n\t"
          "v_pk_mul_f32 %4, %4, %9 clamp\n\t"
          "v_pk_mul_f32 %5, %5, %9 clamp\n\t"
          "v_pk_mul_f32 %6, %6, %9 clamp\n\t"
          "v_pk_mul_f32 %7, %7, %9 clamp\n\t"
          "v_pk_add_f32 %0, %0, %10\n\t"
          "v_pk_add_f32 %1, %1, %10\n\t"
          "v_pk_add_f32 %2, %2, %10\n\t"
          "v_pk_add_f32 %3, %3, %10\n\t"
          "v_pk_add_f32 %4, %4, %10\n\t"
          "v_pk_add_f32 %5, %5, %10\n\t"
          "v_pk_add_f32 %6, %6, %10\n\t"
          "v_pk_add_f32 %7, %7, %10"
            : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
            : "s"(c2), "s"(m2), "s"(g2));
      } else {
        asm("s_nop 7\n\ts_nop 1\n\t"
          "v_pk_add_f32 %0, %0, %8\n\t"
          "v_pk_add_f32 %1, %1, %8\n\t"
          "v_pk_add_f32 %2, %2, %8\n\t"
          "v_pk_add_f32 %3, %3, %8\n\t"
          "v_pk_add_f32 %4, %4, %8\n\t"
          "v_pk_add_f32 %5, %5, %8\n\t"
          "v_pk_add_f32 %6, %6, %8\n\t"
          "v_pk_add_f32 %7, %7, %8\n\t"
          "v_pk_mul_f32 %0, %0, %9\n\t"
          "v_pk_mul_f32 %1, %1, %9\n\t"
          "v_pk_mul_f32 %2, %2, %9\n\t"
          "v_pk_mul_f32 %3, %3, %9\n\t"
          "v_pk_mul_f32 %4, %4, %9\n\t"
          "v_pk_mul_f32 %5, %5, %9\n\t"
          "v_pk_mul_f32 %6, %6, %9\n\t"
          "v_pk_mul_f32 %7, %7, %9\n\t"
          "v_pk_add_f32 %0, %0, %10\n\t"
          "v_pk_add_f32 %1, %1, %10\n\t"
          "v_pk_add_f32 %2, %2, %10\n\t"
          "v_pk_add_f32 %3, %3, %10\n\t"
          "v_pk_add_f32 %4, %4, %10\n\t"
          "v_pk_add_f32 %5, %5, %10\n\t"
          "v_pk_add_f32 %6, %6, %10\n\t"
          "v_pk_add_f32 %7, %7, %10"
            : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
            : "s"(c2), "s"(m2), "s"(g2));
      }
    } else {
      static_assert(MODE != ITA_RQ_RELU, "the ReLU site has 8 pairs");
      asm("s_nop 7\n\ts_nop 1\n\t"
        "v_pk_add_f32 %0, %0, %4\n\t"
        "v_pk_add_f32 %1, %1, %4\n\t"
        "v_pk_add_f32 %2, %2, %4\n\t"
        "v_pk_add_f32 %3, %3, %4\n\t"
        "v_pk_mul_f32 %0, %0, %5\n\t"
        "v_pk_mul_f32 %1, %1, %5\n\t"
        "v_pk_mul_f32 %2, %2, %5\n\t"
        "v_pk_mul_f32 %3, %3, %5\n\t"
        "v_pk_add_f32 %0, %0, %6\n\t"
        "v_pk_add_f32 %1, %1, %6\n\t"
        "v_pk_add_f32 %2, %2, %6\n\t"
        "v_pk_add_f32 %3, %3, %6"
          : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])
          : "s"(c2), "s"(m2), "s"(g2));
    }
  }
}
// eight dwords of two 16-bit (r + 128) values each -> four dwords of four int8 codes: u8 saturation of both halves
// (= clamp(r, -128, 127) + 128), SDWA places the byte pair in the low / high word, ^ 0x80 per byte.  Same SDWA hazard rule
// as round_pack16: four destinations interleaved, one wait state at the end.
__device__ __forceinline__ i32x4 sat_pack16(const unsigned (&w)[8]) {
  unsigned p0, p1, p2, p3;
  const unsigned flip = 0x80808080u;
  asm(
      "v_sat_pk_u8_i16_sdwa %0, %4 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD\n\t"
      "v_sat_pk_u8_i16_sdwa %1, %6 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD\n\t"
      "v_sat_pk_u8_i16_sdwa %2, %8 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD\n\t"
      "v_sat_pk_u8_i16_sdwa %3, %10 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD\n\t"
      "v_sat_pk_u8_i16_sdwa %0, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
      "v_sat_pk_u8_i16_sdwa %1, %7 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
      "v_sat_pk_u8_i16_sdwa %2, %9 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
      "v_sat_pk_u8_i16_sdwa %3, %11 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
      "v_xor_b32 %0, %12, %0\n\t"
      "v_xor_b32 %1, %12, %1\n\t"
      "v_xor_b32 %2, %12, %2\n\t"
      "v_xor_b32 %3, %12, %3\n\t"
      "s_nop 0"
      : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3)
      : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "s"(flip));
  return (i32x4){(int)p0, (int)p1, (int)p2, (int)p3};
}
// requantise + pack sixteen biased accumulators (four 16x16 tiles): byte 4t + i of the result <-> acc[t][i]
template <int MODE>
__device__ __forceinline__ i32x4 rq_pack16_v3(const i32x4 (&acc)[4], float mult) {
  f32x2 x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = (f32x2){__int_as_float(acc[k >> 1][2 * (k & 1)]), __int_as_float(acc[k >> 1][2 * (k & 1) + 1])};
  if constexpr (MODE == ITA_RQ_RELU) pk_requant_round<8, MODE>(x, mult * 0.00390625f, ITA_MAGIC128_F * 0.00390625f);
  else pk_requant_round<8, MODE>(x, mult, ITA_MAGIC128_F);
  unsigned w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)   // {lo16(x.x), lo16(x.y)}
    w[k] = __builtin_amdgcn_perm(__float_as_uint(x[k].y), __float_as_uint(x[k].x), 0x05040100u);
  return sat_pack16(w);
}
// logits: eight biased accumulators (two key tiles) -> four dwords of two ORDER-PRESERVING u16 each: rne(acc * mult) + 32768,
// unclamped (the integer softmax clamps: ita_softmax_packed16).  Pair k <-> acc[k >> 1][2 (k & 1)], [.. + 1].
template <int MODE>
__device__ __forceinline__ void lg8_v3(const i32x4 (&acc)[2], float mult, unsigned (&w)[4]) {
  f32x2 x[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) x[k] = (f32x2){__int_as_float(acc[k >> 1][2 * (k & 1)]), __int_as_float(acc[k >> 1][2 * (k & 1) + 1])};
  pk_requant_round<4, MODE>(x, mult, ITA_MAGIC32K_F);
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = __builtin_amdgcn_perm(__float_as_uint(x[k].y), __float_as_uint(x[k].x), 0x05040100u);
}
template <typename ACC>
__device__ __forceinline__ void rq_pack16(const ACC& acc, float mult, float lo, unsigned (&pk)[4]) {
  float f[16];
  scale_clamp<16>(acc, mult, lo, f);
  round_pack16(f, pk);
}
// sum of the four signed bytes of each of four packed dwords (v_dot4_i32_i8 against 0x01010101)
__device__ __forceinline__ int sum_bytes16(const unsigned (&pk)[4]) {
  int s = 0;
#pragma unroll
  for (int g = 0; g < 4; ++g) s = __builtin_amdgcn_sdot4((int)pk[g], 0x01010101, s, false);
  return s;
}
// torch.quantize_per_tensor on sixteen floats
__device__ __forceinline__ void q_pack16(const float* x, float inv_scale, unsigned (&pk)[4]) {
  float f[16];
  const f32x2 m2 = {inv_scale, inv_scale};
#pragma unroll
  for (int i = 0; i < 16; i += 2) {
    f32x2 v = {x[i], x[i + 1]};
    v = v * m2;
    f[i] = __builtin_amdgcn_fmed3f(v.x, -128.0f, 127.0f);
    f[i + 1] = __builtin_amdgcn_fmed3f(v.y, -128.0f, 127.0f);
  }
  round_pack16(f, pk);
}

__device__ __forceinline__ unsigned pack4(unsigned b0, unsigned b1, unsigned b2, unsigned b3) {
  return (b0 & 0xffu) | ((b1 & 0xffu) << 8) | ((b2 & 0xffu) << 16) | (b3 << 24);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it
// waits for every outstanding global store and prefetch load at each phase boundary.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// cross-lane exchanges on the VALU (no LDS round trip)
__device__ __forceinline__ int xor1_i(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ int xor2_i(int v) { return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true); }   // quad_perm [2,3,0,1]
__device__ __forceinline__ float xor1_f(float v) { return __int_as_float(xor1_i(__float_as_int(v))); }
__device__ __forceinline__ float xor2_f(float v) { return __int_as_float(xor2_i(__float_as_int(v))); }
// value of lane ^ 16 / lane ^ 32
__device__ __forceinline__ int xor16_i(int v) {
  const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
  return (int)(((threadIdx.x >> 4) & 1) ? r[0] : r[1]);
}
__device__ __forceinline__ int xor32_i(int v) {
  const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  return (int)(((threadIdx.x >> 5) & 1) ? r[0] : r[1]);
}

// exp with a fixed arithmetic, identical to oracle/ita_oracle.c:ita_oracle_expf
__device__ __forceinline__ float ita_expf(float x) {
  x = fminf(fmaxf(x, -87.0f), 88.0f);
  float n = rintf(x * 1.44269504088896341f);
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500E-4f;
  p = fmaf(p, r, 1.3981999507E-3f);
  p = fmaf(p, r, 8.3334519073E-3f);
  p = fmaf(p, r, 4.1665795894E-2f);
  p = fmaf(p, r, 1.6666665459E-1f);
  p = fmaf(p, r, 5.0000001201E-1f);
  float r2 = r * r;
  p = fmaf(p, r2, r) + 1.0f;
  float s = __uint_as_float((unsigned)((int)n + 127) << 23);
  return p * s;
}
__device__ __forceinline__ float ita_sigmoid(float x) { return 1.0f / (1.0f + ita_expf(-x)); }
__device__ __forceinline__ float ita_tanh(float x) { return 1.0f - 2.0f / (ita_expf(2.0f * x) + 1.0f); }

// source row / column and weight of F.interpolate(mode='bilinear', align_corners=False)
__device__ __forceinline__ void bilinear_src_dev(int dst, float scale, int in, int& i0, int& ip, float& l1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.0f) src = 0.0f;
  int i = (int)src;
  if (i > in - 1) i = in - 1;
  i0 = i;
  ip = (i < in - 1) ? 1 : 0;
  l1 = src - (float)i;
}

// LayerNorm over E channels spread over NT adjacent lanes (NT = 2 or 4), each holding EC = E/NT
// consecutive channels, in the oracle's summation order: 4 blocks of E/4 consecutive channels
// summed sequentially, combined (p0+p1)+(p2+p3).   r[] is overwritten with the result.
template <int E, int NT, typename WP, typename BP>
__device__ __forceinline__ void layernorm_lanes(float (&r)[E / NT], const WP& w, const BP& b, int c0) {
  constexpr int EC = E / NT, Q = E / 4;
  const float inv_e = 1.0f / (float)E;
  float tot;
  if constexpr (NT == 4) {
    float p = 0.0f;
#pragma unroll
    for (int i = 0; i < EC; ++i) p = p + r[i];
    float s1 = p + xor1_f(p);
    tot = s1 + xor2_f(s1);
  } else {
    float pa = 0.0f, pb = 0.0f;
#pragma unroll
    for (int i = 0; i < Q; ++i) pa = pa + r[i];
#pragma unroll
    for (int i = 0; i < Q; ++i) pb = pb + r[Q + i];
    float s1 = pa + pb;
    tot = s1 + xor1_f(s1);
  }
  const float mean = tot * inv_e;
  if constexpr (NT == 4) {
    float p = 0.0f;
#pragma unroll
    for (int i = 0; i < EC; ++i) { float d = r[i] - mean; p = fmaf(d, d, p); }
    float s1 = p + xor1_f(p);
    tot = s1 + xor2_f(s1);
  } else {
    float pa = 0.0f, pb = 0.0f;
#pragma unroll
    for (int i = 0; i < Q; ++i) { float d = r[i] - mean; pa = fmaf(d, d, pa); }
#pragma unroll
    for (int i = 0; i < Q; ++i) { float d = r[Q + i] - mean; pb = fmaf(d, d, pb); }
    float s1 = pa + pb;
    tot = s1 + xor1_f(s1);
  }
  const float var = tot * inv_e;
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
  for (int i = 0; i < EC; ++i) r[i] = fmaf((r[i] - mean) * rstd, w[c0 + i], b[c0 + i]);
}
