// ita_stream_kernel.h -- the ITA encoder layer as eight wave-private token streams per CU.
//
//   x1 = LayerNorm1(x + ITASelfAttention_QAT(x));  y = LayerNorm2(x1 + ITAFeedForward_QAT(x1))
//   (reference models/ITA_single_layer_upsample_shuffle/QAT/model.py:100-113,
//    models/ITA/QAT/layers.py:39-45,61-75,101-127, models/ITA/QAT/ITA_softmax.py:51-61)
//
// Same arithmetic, bit for bit, as ita_mha_kernel + ita_ffn_kernel (ita_int8_kernels.h) and the CPU
// oracle; what is different is who owns what.  One persistent 512-thread workgroup per CU; wave w owns
// tokens 16w .. 16w+15 of the current frame (token row w of the 8 x 16 grid) from the pixels to the
// LayerNorm2 output.  Lane (qi = lane & 15, kq = lane >> 4) holds channels (E/4)kq .. (E/4)kq + E/4 - 1 of
// token 16w + qi in f32 registers for the residuals.
//
// Every GEMM of the layer is a 16x16x64 int8 MFMA with the WEIGHTS as the A operand (read from a
// resident LDS image) and the wave's OWN activations as the B operand, taken from registers: the C
// layout of that MFMA (lane = token column, four consecutive feature rows) packs, four tiles at a time,
// into exactly one B fragment of the next GEMM -- the integer sum does not care which k a fragment byte
// is, only that the weight image uses the same slot -> k mapping, and that mapping is baked into the
// image at load time (ita_plugin.hip: build_stream_image).  So x_q, Q, the context, the FFN hidden layer
// and both block outputs never touch LDS; only K and V^T do, because every wave needs every key.
// Two barriers per frame (K/V^T complete; K/V^T free again) instead of eight, and no token of a frame
// ever waits for another wave's tile.
//
// LDS (E = 64, fused tokenizer): weight image 80 KB + biases/LayerNorm 5.3 KB + conv weights 13.3 KB
// (f32 MFMA fragments) | K 24 KB | V^T 24 KB | column sums | eight private 9 x 96 byte windows of the next
// frame = 161 744 B of the CU's 160 KB (163 840).  E = 128 (no tokenizer; fc1 / fc2 weights stay in global memory): 158 464 B.
#pragma once
#include "ita_int8_kernels.h"
#ifndef ITA_ABLATE
#define ITA_ABLATE 0
#endif

struct ItaStreamArgs {
  const char* image;      // device copy of the LDS image (ItaStreamLds<...>::IMAGE bytes, built at load time)
  const float* x;         // (B,128,E) tokens (TOK == 0)
  float* y;               // (B,128,E) f32 output, may be null when planes are given
  _Float16 *y_hi, *y_lo;  // optional f16 hi/lo planes of y, row stride ld_planes per frame
  int ld_planes;
  float* x1_tap;          // optional (B,128,E): LayerNorm1 output
  float inv_sx, mq, mk, mv, ml, mc, mo, so;      // attention scalars (ita_weights.h)
  float f_inv_sx, m1, m2, s2;                    // FFN scalars
  int B;
  int fuse_ln;            // attention-only form: y = fuse_ln ? LayerNorm1(x + attn(x)) : attn(x)
  // diagnostic only (null in production): waves 0 and 4 of each workgroup store s_memtime at the phase
  // boundaries of the first 8 frames: stamps[((block * 8 + frame) * 2 + (wave >> 2)) * 16 + slot]
  unsigned long long* stamps;
  // optional side copy for the LSTM that follows: layer-0 hidden state of frame b (row slots[b] or b of h0_src) ->
  // h0_dst[b].  Layer 0 reads whole rows of h while other workgroups overwrite parts of the same row when the state
  // is updated in place; the staged copy removes that race.
  const float* h0_src;
  float* h0_dst;
  const int* slots;
  // int8 in / int8 out form of the attention block (IO8: the accelerator-native boundary):
  const int8_t* xq;       // (B,128,E) quantised block input
  int8_t* yq;             // (B,128,E) out_proj codes (no dequantisation, no residual)
  const void* img;        // (B,60,90) u8 wire frames (TOK == 1)
  float* tok_tap;         // optional (B,128,E): the tokens
  // bit s set: requantisation site s may use the single-rounding form (one v_pk_fma_f32 instead of multiply + magic add):
  // ita_load_weights has enumerated every accumulator value of the site's clamp range and found the two forms equal.
  // Informational inside the kernel: the launcher picks the FAST instantiation when all six bits are set.
  unsigned fast_sites;
};
enum { ITA_SITE_Q = 1, ITA_SITE_K = 2, ITA_SITE_V = 4, ITA_SITE_L = 8, ITA_SITE_C = 16, ITA_SITE_O = 32, ITA_SITES_ALL = 63 };

#ifdef ITA_NO_SCHEDBAR
#define ITA_SCHED_BARRIER() do {} while (0)
#else
#define ITA_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif
// Where the four conv tiles of the NEXT frame's tokenizer run inside a frame (each: six int8 MFMAs + 24 VALU + 8 LDS reads):
//   0 (default) after LayerNorm2 and the output stores, where the kernel has registers to spare;
//   1 spread over the out_proj / fc1 steps (needs ~20 more live registers at the kernel's tightest point: 7 spilled);
//   2 all four right after the blend, in front of out_proj.
// Measured on one box, encoder launch of 1024 frames: round-2 kernel 58.4 us, placement 0 with the blend at the end too
// (ITA_TOK_BLEND_END, default) 52.1, placement 0 with the blend after B2 52.5, placement 2 52.9, placement 1 (another box) +1.5.
#ifndef ITA_TOK_PLACE
#define ITA_TOK_PLACE 0
#endif
#define ITA_TOK_MID (ITA_TOK_PLACE == 1)
#ifndef ITA_TOK_BLEND_END   // placement 0 only: blend the taps at the end of the frame too (nothing of the tokenizer is then live across the FFN: 244 VGPRs)
#define ITA_TOK_BLEND_END 1
#endif

// ---- the conv7x7 of the u8 tokenizer as int8 MFMA (oracle/ita_oracle.c: ita_oracle_tokenizer_u8).  The blended tap is the
// exact integer B256 = 256 a1 + a0 <= 65280, the conv weight of a channel 23-bit fixed point Wq = 65536 w2 + 256 w1 + w0 with
// balanced digits, and the 49-tap sum splits into byte products that int8 MFMAs accumulate exactly:
//     S0 = sum a0 w0,  S1 = sum (a0 w1 + a1 w0),  S2 = sum (a0 w2 + a1 w1),  S3 = sum a1 w2,   L = S0 + 256 S1,  H = S2 + 256 S3
//     pre = fma((float)H, 65536 s, fma((float)L, s, bias))
// The unsigned bytes a0, a1 ride the signed MFMA as a ^ 0x80 = a - 128; the 128 * sum(w) terms sit in the accumulators'
// initial values (I0 for S0 -- it also carries 256 x the term of S1 --, I2 for S2 / S3).  K = 64 slots of the 16x16x64 MFMA:
// lane (token qi, k-group kq) holds its own 13 taps 4 j + kq in bytes j = 0..12 of its B fragment (the weight image has
// the same slot -> tap mapping, zero weights in the three pad slots).  Six MFMAs per 16-channel tile instead of thirteen
// v_mfma_f32_16x16x4_f32, which run at the f32 vector rate and hold the SIMD's VALU meanwhile (-5 us per 1024-frame launch).
template <int E>
struct ItaTokTab {
  static constexpr int NCT = E / 16;
  static constexpr int TW = 0;                          // int8 [NCT][3 byte planes][64 lanes][16]: A fragments, row rho <-> channel (E/4)(rho>>2) + 4 ct + (rho&3)
  static constexpr int TI = TW + NCT * 3 * 1024;        // int32 [2][E]: I0 | I2
  static constexpr int TS = TI + 2 * E * 4;             // f32 [3][E]: s | 65536 s | conv bias
  static constexpr int BYTES = TS + 3 * E * 4;
};
// thirteen blended taps (exact integers) -> the two B fragments (low bytes, high bytes; both ^ 0x80)
__device__ __forceinline__ void tok_u8_fragments(const unsigned (&pb)[13], i32x4& a0, i32x4& a1) {
  unsigned p16[8];
#pragma unroll
  for (int k = 0; k < 6; ++k) p16[k] = __builtin_amdgcn_perm(pb[2 * k + 1], pb[2 * k], 0x05040100u) ^ 0x80808080u;
  p16[6] = (pb[12] & 0xffffu) ^ 0x80808080u;
  p16[7] = 0;                                           // pad slots: their weights are zero
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    a0[d] = (int)__builtin_amdgcn_perm(p16[2 * d + 1], p16[2 * d], 0x06040200u);
    a1[d] = (int)__builtin_amdgcn_perm(p16[2 * d + 1], p16[2 * d], 0x07050301u);
  }
}
// one 16-channel tile: out[i] = pre-LayerNorm conv output of channel (E/4) kq + 4 ct + i of this lane's token
template <int E>
__device__ __forceinline__ void tok_u8_tile(const char* tab, int ct, int lane, int kq, const i32x4& a0, const i32x4& a1, float (&out)[4]) {
  using T = ItaTokTab<E>;
  const int c0 = (E / 4) * kq + 4 * ct;
  // (one weight fragment live at a time, the sums combined in place: the tile runs where the encoder kernel has no register to spare)
  const i32x4 z = {0, 0, 0, 0};
  i32x4 s0 = *(const i32x4*)(tab + T::TI + c0 * 4), s1, s2 = *(const i32x4*)(tab + T::TI + (E + c0) * 4), s3;
  {
    const i32x4 w0 = *(const i32x4*)(tab + T::TW + ((ct * 3 + 0) * 64 + lane) * 16);
    s0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0, a0, s0, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0, a1, z, 0, 0, 0);
  }
  {
    const i32x4 w1 = *(const i32x4*)(tab + T::TW + ((ct * 3 + 1) * 64 + lane) * 16);
    s2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w1, a1, s2, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w1, a0, s1, 0, 0, 0);
  }
  {
    const i32x4 w2 = *(const i32x4*)(tab + T::TW + ((ct * 3 + 2) * 64 + lane) * 16);
    s3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w2, a1, z, 0, 0, 0);
    s2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w2, a0, s2, 0, 0, 0);
  }
  float lf[4], hf[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { lf[i] = (float)(s0[i] + (s1[i] << 8)); hf[i] = (float)(s2[i] + (s3[i] << 8)); }
  {
    const f32x4 sc = *(const f32x4*)(tab + T::TS + c0 * 4), cb = *(const f32x4*)(tab + T::TS + (2 * E + c0) * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) lf[i] = fmaf(lf[i], sc[i], cb[i]);
  }
  {
    const f32x4 sc16 = *(const f32x4*)(tab + T::TS + (E + c0) * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = fmaf(hf[i], sc16[i], lf[i]);
  }
}

template <int E, bool FFN, bool TOK>
struct ItaStreamLds {
  static constexpr int S = 128, P = 192, F = 256;
  // ---- image: copied verbatim from global memory once per workgroup
  static constexpr int WQ = 0;                       // int8 [E/16][192][16]  chunk-major, natural k
  static constexpr int WK = WQ + P * E;
  static constexpr int WV = WK + P * E;
  static constexpr int WO = WV + P * E;              // int8 [12][E][16]     fragment order (see build_stream_image)
  // E = 128: attention weights (98 KB) + K / V^T (49 KB) fill the LDS, so fc1 / fc2 (64 KB) stay in GLOBAL memory (the image
  // continues behind the part that is copied to LDS) and are read as fragments, L2-resident, once per frame
  static constexpr bool W12G = FFN && E > 64;
  static constexpr int W1 = WO + E * P;              // int8 [E/16][256][16] natural k
  static constexpr int W2 = W1 + ((FFN && !W12G) ? F * E : 0);  // int8 [16][E][16]     fragment order
  static constexpr int BIAS = W2 + ((FFN && !W12G) ? E * F : 0);   // int32: bq | bk | bv | bo | b1 | b2
  static constexpr int NBIAS = 3 * P + E + (FFN ? F + E : 0);
  static constexpr int LNP = BIAS + NBIAS * 4;       // f32: n1w | n1b | n2w | n2b | tok_lnw | tok_lnb
  static constexpr int NLN = 2 * E + (FFN ? 2 * E : 0) + (TOK ? 2 * E : 0);
  static constexpr int CW = LNP + NLN * 4;           // integer conv tables (ItaTokTab<E>): weight byte planes | init sums | scales, bias
  static constexpr int TAP = CW + (TOK ? ItaTokTab<E>::BYTES : 0);   // int32 [52]: window offset ky * 96 + kx of tap t (0 for t >= 49)
  static constexpr int VB4 = TAP + (TOK ? 52 * 4 : 0);          // int32 [192][4]: bv replicated (V accumulators start per COLUMN)
  static constexpr int IMAGE = VB4 + P * 16;
  static constexpr int GW1 = IMAGE, GW2 = GW1 + F * E;               // W12G: offsets in the global image
  static constexpr int GIMAGE = W12G ? GW2 + E * F : IMAGE;          // bytes of the device image
  // ---- built / used at run time
  static constexpr int K = IMAGE;                    // int8 [12][128][16]  fragment order
  static constexpr int VT = K + S * P;               // int8 [8][192][16]   V^T, keys permuted (ita_int8_kernels.h)
  static constexpr int COLSUM = VT + P * S;          // int32 [2][192]: 128 * column sums of V, per frame parity
  static constexpr int IMG = COLSUM + 2 * P * 4;     // u8 [8 waves][9][96]: rows 2*y0-3 .. 2*y0+5 of the next frame, 3 zero columns each side
  static constexpr int IMG_WAVE = 9 * 96;
  static constexpr int TOTAL = IMG + (TOK ? 8 * IMG_WAVE : 0);
  static_assert(IMAGE % 16 == 0 && K % 16 == 0 && IMG % 16 == 0, "16-byte alignment");
  static_assert(TOTAL <= 160 * 1024, "LDS budget");
};

// ITA_ABLATE (diagnostic builds only, results are wrong): 1 no requantisation VALU, 2 no int8 MFMA, 4 no tokenizer,
// 8 no LDS operand reads, 16 no LayerNorm, 32 no softmax, 64 no barriers, 128 no h0 copy / output stores -- what remains is timed against the full kernel to see what a frame's time is made of
__device__ __forceinline__ i32x4 mfma16(i32x4 a, i32x4 b, i32x4 c) {
  if constexpr (ITA_ABLATE & 2) return c;   // (operand loads become dead code too)
  return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ i32x4 sfrag(const char* lds, int off) {
  if constexpr (ITA_ABLATE & 8) return (i32x4){off, off, off, off};
  return *(const i32x4*)(lds + off);
}
// sums / maxima over the four lanes (kq = 0..3) that share a token: lane ^ 16, lane ^ 32.  With both
// operands the same register the swap leaves {row pairs duplicated} in the two results, so the
// commutative combine needs no select.
__device__ __forceinline__ float sum16_f(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum32_f(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ int sum1632_i(int v) {
  auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
  v = (int)r[0] + (int)r[1];
  r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  return (int)r[0] + (int)r[1];
}
__device__ __forceinline__ int max1632_i(int v) {
  auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
  v = max((int)r[0], (int)r[1]);
  r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  return max((int)r[0], (int)r[1]);
}

// LayerNorm over E channels held E/4 per lane by the four lanes qi, qi+16, qi+32, qi+48, in the oracle's
// summation order: the quarter sums p_kq sequentially over consecutive channels, combined (p0+p1)+(p2+p3)
// -- layernorm_lanes<E, 4> (ita_device.h) with the lane exchange 16 / 32 apart instead of 1 / 2.
template <int E>
__device__ __forceinline__ void layernorm_q16(float (&r)[E / 4], const float* w, const float* b, int c0) {
  constexpr int EC = E / 4;
  if constexpr (ITA_ABLATE & 16) return;
  const float inv_e = 1.0f / (float)E;
  float p = 0.0f;
#pragma unroll
  for (int i = 0; i < EC; ++i) p = p + r[i];
  float tot = sum32_f(sum16_f(p));
  const float mean = tot * inv_e;
  p = 0.0f;
#pragma unroll
  for (int i = 0; i < EC; ++i) { float d = r[i] - mean; p = fmaf(d, d, p); }
  tot = sum32_f(sum16_f(p));
  const float var = tot * inv_e;
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
  for (int i = 0; i < EC; i += 4) {
    const f32x4 w4 = *(const f32x4*)(w + c0 + i), b4 = *(const f32x4*)(b + c0 + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) r[i + j] = fmaf((r[i + j] - mean) * rstd, w4[j], b4[j]);
  }
}

// Integer softmax (models/ITA/QAT/ITA_softmax.py:51-61) of the 16 rows a wave holds as packed signed 16-bit logit pairs
// -- lane (qi, kq): w[2kt + j] = {logit, logit'} of keys 16kt + 4kq + 2j, +1 of row qi -- entirely on v_pk_*_16, two keys
// per VALU op:  shift = max - x,  num = 256 >> shift,  inv = floor(255 * 2^16 / sum),  y = (num * inv) >> 16 = (inv >> 8)
// >> shift.  Out: the A.V MFMA's B fragments, pf[kb] byte 4t+i = (y - 128) of key 64kb + 16t + 4kq + i (the u8
// probabilities ride a signed MFMA as p - 128; the 128 * colsum(V) term restores them).
typedef short ita_s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short ita_u16x2 __attribute__((ext_vector_type(2)));
// The logits arrive UNCLAMPED and biased, u = rne(acc * m) + 32768 (order-preserving u16, lg8_v3); the reference's clamp to
// [-128, 127] happens here at no cost:  m' = clamp(max, -128, 127);  shift = m' - clamp(x)  =  min(max(m' - x, 0), m' + 128)
// -- the saturating unsigned subtraction is the max(., 0) (x above the clamped maximum), and the row-wise cap
// min(m' + 128, 15) covers both x below -128 and the 15 beyond which every shift gives 0.
__device__ __forceinline__ void ita_softmax_packed16(const ita_u16x2 (&w)[16], i32x4 (&pf)[2]) {
  typedef ita_u16x2 u16x2;
  u16x2 m2 = w[0];
#pragma unroll
  for (int j = 1; j < 16; ++j) m2 = __builtin_elementwise_max(m2, w[j]);
  int m = max1632_i(max((int)m2.x, (int)m2.y));
  m = min(max(m, 32768 - 128), 32768 + 127);
  const u16x2 mm = {(unsigned short)m, (unsigned short)m};
  const unsigned short capv = (unsigned short)min(m - (32768 - 128), 15);   // 256 >> s and inv_hi >> s are both 0 from s = 9 on
  const u16x2 cap = {capv, capv};
  const u16x2 one = {256, 256};
  u16x2 sh[16], sum2 = {0, 0};
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    sh[j] = __builtin_elementwise_min(__builtin_elementwise_sub_sat(mm, w[j]), cap);
    sum2 += one >> sh[j];
  }
  int sum = sum1632_i((int)sum2.x + (int)sum2.y);   // <= 16 * 256 per half: no 16-bit overflow
  sum = max(sum, 1);
  const int inv_hi = ((int)floorf((1.0f / (float)sum) * 16711680.0f)) >> 8;   // <= 255: sum >= 256
  const u16x2 iv = {(unsigned short)inv_hi, (unsigned short)inv_hi};
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int kt = 4 * kb + t;
      const unsigned p01 = __builtin_bit_cast(unsigned, (u16x2)(iv >> sh[2 * kt]));
      const unsigned p23 = __builtin_bit_cast(unsigned, (u16x2)(iv >> sh[2 * kt + 1]));
      pf[kb][t] = (int)(__builtin_amdgcn_perm(p23, p01, 0x06040200u) ^ 0x80808080u);
    }
}

// Test entry for that function alone: rows of int8 logits (R,128) -> u8 probabilities, 16 rows per wave in exactly the
// register layout the encoder kernel has them in.  Lets the parity tests feed chosen rows (all equal, one-hot, the
// shift 8 / 9 boundary, +-127 / -128) through the GPU formulation, which otherwise only ever sees what QK^T produces.
__global__ __launch_bounds__(64) void ita_softmax_rows_kernel(const int8_t* __restrict__ logits, uint8_t* __restrict__ probs, int rows) {
  const int lane = threadIdx.x, qi = lane & 15, kq = lane >> 4;
  const int row = min((int)blockIdx.x * 16 + qi, rows - 1);
  ita_u16x2 w[16];
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) {
    const int v = *(const int*)(logits + (size_t)row * 128 + 16 * kt + 4 * kq);
    w[2 * kt] = (ita_u16x2){(unsigned short)((int)(int8_t)v + 32768), (unsigned short)((int)(int8_t)(v >> 8) + 32768)};
    w[2 * kt + 1] = (ita_u16x2){(unsigned short)((int)(int8_t)(v >> 16) + 32768), (unsigned short)((int)(int8_t)(v >> 24) + 32768)};
  }
  i32x4 pf[2];
  ita_softmax_packed16(w, pf);
  if ((int)blockIdx.x * 16 + qi < rows) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        *(unsigned*)(probs + (size_t)row * 128 + 64 * kb + 16 * t + 4 * kq) = (unsigned)pf[kb][t] ^ 0x80808080u;
  }
}

// ---- software pipeline pieces.  One "group" = four 16-feature output tiles of this wave's 16 tokens.  Its LDS
// operands (weight fragments + the accumulator initialisers) are fetched one step ahead of its MFMAs, and its
// requantisation runs one step behind them, so neither the LDS latency nor the MFMA latency is ever waited for:
//     step g:  load(g+1) | mfma(g) | epilogue(g-1)
template <int N>
struct ItaFr {
  i32x4 w[N];   // weight fragments, tile-major: w[t * (N/4) + k-step]
};
// natural-k chunk-major image [E/16][ROWS][16]; the accumulators start from bias rows 16(tile0+t) + 4kq .. +3
template <int NK, int ROWS>
__device__ __forceinline__ void ld_nat(ItaFr<4 * NK>& f, i32x4 (&acc)[4], const char* img, const int* lds_bias, int tile0,
                                       int qi, int kq) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    acc[t] = *(const i32x4*)(lds_bias + (tile0 + t) * 16 + 4 * kq);
#pragma unroll
    for (int c = 0; c < NK; ++c) f.w[t * NK + c] = sfrag(img, ((NK * kq + c) * ROWS + (tile0 + t) * 16 + qi) << 4);
  }
}
// the same image read as the B operand (V projection: roles swapped, column = feature qi): bias is per column
template <int NK, int ROWS>
__device__ __forceinline__ void ld_natv(ItaFr<4 * NK>& f, i32x4 (&acc)[4], const char* img, const int* lds_bias, int tile0,
                                        int qi, int kq) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    acc[t] = *(const i32x4*)(lds_bias + ((tile0 + t) * 16 + qi) * 4);   // replicated table: one 16-byte read, no splat
#pragma unroll
    for (int c = 0; c < NK; ++c) f.w[t * NK + c] = sfrag(img, ((NK * kq + c) * ROWS + (tile0 + t) * 16 + qi) << 4);
  }
}
// one k-step (four tiles) of a fragment-order image [4*NKS][E][16] of a block output projection (out_proj, fc2):
// row 16 et + rho <-> channel (E/4)(rho>>2) + 4 et + (rho&3), so that lane (qi, kq) receives its own channels
// (E/4)kq + 4 et + i; and its MFMAs
struct ItaF4 { i32x4 w[4]; };
template <int E>
__device__ __forceinline__ void ld_frg_ks(ItaF4& f, const char* img, int et0, int ks, int qi, int kq) {
#pragma unroll
  for (int t = 0; t < 4; ++t) f.w[t] = sfrag(img, ((4 * ks + kq) * E + (et0 + t) * 16 + qi) << 4);
}
__device__ __forceinline__ void mm_ks(const ItaF4& f, const i32x4& x, i32x4 (&acc)[4]) {
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = mfma16(f.w[t], x, acc[t]);
}
template <int E>
__device__ __forceinline__ void ld_obias(i32x4 (&acc)[4], const int* lds_bias, int et0, int kq) {
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = *(const i32x4*)(lds_bias + (E / 4) * kq + 4 * (et0 + t));
}
// acc[t] += sum_k  W-fragment x activation fragment   (SWAP: the activations are the A operand)
template <int NKS, bool SWAP>
__device__ __forceinline__ void mm_group(const ItaFr<4 * NKS>& f, const i32x4 (&x)[NKS], i32x4 (&acc)[4]) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      acc[t] = SWAP ? mfma16(x[ks], f.w[t * NKS + ks], acc[t]) : mfma16(f.w[t * NKS + ks], x[ks], acc[t]);
}
// requantise four tiles and pack them: ONE B fragment of the next GEMM (byte 4t+i <-> row 4kq+i of tile t).
// fast (wave-uniform): this site's multiplier passed the load-time single-rounding proof (ita_device.h: rq_pack16_v3);
// RELU: fc1, clamp to [0, 127].  -DITA_RQ_STYLE=2 keeps the round-2 form (float clamp, 3.0 instructions per value).
template <bool RELU = false>
__device__ __forceinline__ i32x4 rq_group(const i32x4 (&acc)[4], float mult, bool fast) {
  if constexpr (ITA_ABLATE & 1) return acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
#if defined(ITA_RQ_STYLE) && ITA_RQ_STYLE == 1
  return rq_pack16_c(acc, mult, RELU ? 0.0f : -128.0f);
#elif defined(ITA_RQ_STYLE) && ITA_RQ_STYLE == 2
  return rq_pack16_b(acc, mult, RELU ? 0.0f : -128.0f);
#else
  if constexpr (RELU) return rq_pack16_v3<ITA_RQ_RELU>(acc, mult);
  if (fast) return rq_pack16_v3<ITA_RQ_FAST>(acc, mult);
  return rq_pack16_v3<ITA_RQ_EXACT>(acc, mult);
#endif
}
// block output: d[4t+i] = dequantised int8 code of channel (E/4)kq + 4(et0+t) + i
__device__ __forceinline__ void dq_group(const i32x4 (&acc)[4], float mult, float scale, float (&d)[16]) {
  if constexpr (ITA_ABLATE & 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = __int_as_float(acc[i >> 2][i & 3]);
    return;
  }
#if defined(ITA_RQ_STYLE) && ITA_RQ_STYLE == 1
  dq16_c(acc, mult, scale, d);
#else
  dq16_b(acc, mult, scale, d);
#endif
}

// ---- f32 token rows in and out.  Lane (qi, kq) owns the quarter [E/4 kq, E/4 (kq+1)) of token qi's row.  Loaded or stored
// quarter by quarter, a wave-instruction touches 64 different cache lines for 16 bytes each -- the shape one CU moves at a
// fifth of its contiguous rate (tools/microbench/cu_fill_rows.hip).  Instead the four lanes of a token move 64 CONTIGUOUS
// bytes per instruction (16 rows x 64 B per wave-instruction) and a 4 x 4 transpose of 16-byte items across those lanes
// (lanes 16 and 32 apart: two rounds of v_permlane32_swap / v_permlane16_swap, 16 instructions) sorts the quarters out.
// raw[16 jj + 4 i + c]: item i of transpose jj = floats 4 p .. 4 p + 3 of the row, p = (E/16) i + 4 jj + kq.
template <int E>
__device__ __forceinline__ void ld_tok_items(const float* row, int kq, float (&raw)[E / 4]) {
  constexpr int PQ = E / 16, NJ = E / 64;
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 v = *(const f32x4*)(row + 4 * (PQ * i + 4 * jj + kq));
      raw[16 * jj + 4 * i] = v.x; raw[16 * jj + 4 * i + 1] = v.y; raw[16 * jj + 4 * i + 2] = v.z; raw[16 * jj + 4 * i + 3] = v.w;
    }
}
// in place: items (as loaded / as they will be stored) <-> this lane's quarter in channel order; its own inverse
template <int E>
__device__ __forceinline__ void tok_items_transpose(float (&x)[E / 4]) {
  constexpr int NJ = E / 64;
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      unsigned i0 = __float_as_uint(x[16 * jj + c]), i1 = __float_as_uint(x[16 * jj + 4 + c]);
      unsigned i2 = __float_as_uint(x[16 * jj + 8 + c]), i3 = __float_as_uint(x[16 * jj + 12 + c]);
      const auto a02 = __builtin_amdgcn_permlane32_swap(i0, i2, false, false);
      const auto a13 = __builtin_amdgcn_permlane32_swap(i1, i3, false, false);
      const auto b01 = __builtin_amdgcn_permlane16_swap(a02[0], a13[0], false, false);
      const auto b23 = __builtin_amdgcn_permlane16_swap(a02[1], a13[1], false, false);
      x[16 * jj + c] = __uint_as_float(b01[0]); x[16 * jj + 4 + c] = __uint_as_float(b01[1]);
      x[16 * jj + 8 + c] = __uint_as_float(b23[0]); x[16 * jj + 12 + c] = __uint_as_float(b23[1]);
    }
}
template <int E>
__device__ __forceinline__ void st_tok_quarter(float* row, int kq, const float (&y)[E / 4]) {
  constexpr int PQ = E / 16, NJ = E / 64;
  float t[E / 4];
#pragma unroll
  for (int i = 0; i < E / 4; ++i) t[i] = y[i];
  tok_items_transpose<E>(t);
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *(f32x4*)(row + 4 * (PQ * i + 4 * jj + kq)) = (f32x4){t[16 * jj + 4 * i], t[16 * jj + 4 * i + 1], t[16 * jj + 4 * i + 2], t[16 * jj + 4 * i + 3]};
}

// (amdgpu_waves_per_eu(2, 2): the workgroup owns the CU's LDS, so two waves per SIMD is all there will ever be -- without
// it the scheduler trades instruction-level parallelism for registers it has no use for: the LDS size is dynamic)
// FAST: every requantisation site of this layer passed the load-time single-rounding proof (ita_plugin.hip: fast_site_ok).
// A compile-time switch, two instantiations: as a run-time flag per site the two forms sit side by side in the frame loop
// and the ~40 uniform branches per frame cost more than the shorter form saves (measured: 56.7 us against 56.5 for the
// round-2 form, 54.0 with the fast form compiled in, 55.3 with the exact one).
template <int E, bool FFN, int TOK, bool STAMP = false, bool IO8 = false, bool FAST = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void ita_stream_kernel(const ItaStreamArgs a) {
  static_assert(!IO8 || (!FFN && TOK == 0), "int8 I/O is the attention block's form");
  using L = ItaStreamLds<E, FFN, TOK != 0>;
  constexpr int S = 128, P = 192, F = 256, EC = E / 4, NK = E / 64, NTE = E / 16;
  static_assert(!TOK || (E == 64 && FFN), "the fused tokenizer is built for the ITAViTLSTM shape");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keep it (and what derives from it) in SGPRs
  const int qi0 = lane & 15, kq0 = lane >> 4;   // E = 64: the handful of LDS bases derived from these stay hoisted
  const int* bias = (const int*)(lds + L::BIAS);
  const int *l_bq = bias, *l_bk = bias + P, *l_bo = bias + 3 * P, *l_b1 = bias + 3 * P + E,
            *l_b2 = bias + 3 * P + E + F;
  const float* lnp = (const float*)(lds + L::LNP);
  int* colsum = (int*)(lds + L::COLSUM);
  // per-site single-rounding permission (ItaStreamArgs::fast_sites, proven at load time): wave-uniform scalars
  auto site_fast = [&](unsigned bit) {
#ifdef ITA_FORCE_SITES   // counting builds (tools/valu_budget.py): one form per site at compile time
    return ((unsigned)(ITA_FORCE_SITES) & bit) != 0;
#endif
    (void)bit;
    return FAST;
  };
#define fq site_fast(ITA_SITE_Q)
#define fk site_fast(ITA_SITE_K)
#define fv site_fast(ITA_SITE_V)
#define fl site_fast(ITA_SITE_L)
#define fc site_fast(ITA_SITE_C)
#define fo site_fast(ITA_SITE_O)

  // waves 4-7 were dispatched second and lose every VALU arbitration against their SIMD partner (MI355X guide,
  // "Two waves per SIMD", item 4): static priority evens the pair out
#ifndef ITA_NO_SETPRIO
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  int fi = 0;
#define ITA_SSTAMP(ph)                                                                                  \
  do {                                                                                                  \
    if (STAMP && a.stamps && (tid & 255) == 0 && fi < 8)                                                      \
      a.stamps[(((size_t)blockIdx.x * 8 + fi) * 2 + (wave >> 2)) * 16 + (ph)] = __builtin_amdgcn_s_memtime(); \
  } while (0)

  // ---- fused tokenizer (OverlapPatchMerging, reference models/ITA/QAT/layers.py:39-45; same arithmetic and
  // operation order as ita_tokenizer_kernel).  Conv7x7/s2 and the bilinear 30x45 -> 8x16 resize are both linear,
  // so the 7x7 patch is blended first and convolved once per token.  Wave w = token row w needs image rows
  // 2*y0-3 .. 2*y0+5 only (y0 = source row of the resize), a private 9 x 96 byte window with a zero border:
  //   fetch  : five dwords per lane of the next frame (issued a phase early, consumed by fill)
  //   fill   : funnel-shift to the window's 16-byte pieces, mask the border, one ds_write_b128 per lane
  //   compute: lane (qi, kq) blends taps 4s+kq (s = 0..12) of token qi -- exactly the B operand
  //            (column = token, k = kq) of v_mfma_f32_16x16x4_f32, on gfx950 an exact ascending-k fmaf chain;
  //            the A operand is the conv weight fragment image.  C = lane (token qi, channels 16kq+4ct+i):
  //            the layout the encoder keeps x in.  No patch, no pre-LayerNorm token ever reaches LDS.
  // Everything per-lane below is derived from an OPAQUE copy of the lane id taken once per frame: otherwise the compiler
  // hoists ~60 registers of loop-invariant geometry and LDS addresses out of the frame loop, and the kernel (which
  // lives at the 256-register limit of two waves per SIMD) spills.
  unsigned tk_d[5] = {0, 0, 0, 0, 0};
  struct TokGeo { int y0, x0, rr, pc, row, o; float h1, w1; };
  auto tok_geo = [&](int ol) {
    TokGeo g;
    int yp, xp;
    bilinear_src_dev(wave, 30.0f / 8.0f, 30, g.y0, yp, g.h1);        // y0 <= 27 < 29 and x0 <= 43 < 44: the second
    bilinear_src_dev(ol & 15, 45.0f / 16.0f, 45, g.x0, xp, g.w1);    // neighbour is always one row / column on
    g.rr = ol / 6; g.pc = ol - 6 * g.rr;                             // window piece of this lane (lane < 54)
    g.row = 2 * g.y0 - 3 + g.rr;                                     // image row (-1 for the top row of wave 0)
    g.o = g.row * 90 + 16 * g.pc - 3;                                // frame byte of the piece's first window byte
    return g;
  };
  auto tok_fetch = [&](int fb, int ol) {
    const TokGeo g = tok_geo(ol);
    const uint8_t* src = (const uint8_t*)a.img + (size_t)fb * 5400;
    const int a0 = g.o & ~3;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int aj = a0 + 4 * j;
      tk_d[j] = 0;
      if (ol < 54 && g.row >= 0 && aj >= 0 && aj < 5400) tk_d[j] = *(const unsigned*)(src + aj);
    }
  };
  auto tok_fill = [&](int ol) {
    const TokGeo g = tok_geo(ol);
    const int sh = g.o & 3;
    unsigned o4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o4[j] = __builtin_amdgcn_alignbyte(tk_d[j + 1], tk_d[j], sh);
    if (g.pc == 0) o4[0] &= 0xff000000u;        // window columns 0..2  = image columns -3..-1
    if (g.pc == 5) o4[3] &= 0x000000ffu;        // window columns 93..95 = image columns 90..92
    if (ol < 54)
      *(i32x4*)(lds + L::IMG + wave * L::IMG_WAVE + g.rr * 96 + 16 * g.pc) =
          (i32x4){(int)o4[0], (int)o4[1], (int)o4[2], (int)o4[3]};
  };
  // blend: this lane's 13 taps of its token as exact integers, packed into the two B fragments of the int8 conv
  i32x4 tk_a0, tk_a1;
  float tk_out[16];
  auto tok_blend = [&](int ol) {
    const TokGeo g = tok_geo(ol);
    const int kq = ol >> 4;
    const uint8_t* win = (const uint8_t*)(lds + L::IMG + wave * L::IMG_WAVE) + 2 * g.x0;
    const int* tap = (const int*)(lds + L::TAP);
    // The weights of this fixed resize are dyadic, h = H / 8 and w = W / 32 with 0 < H1 < 8, 0 < W1 < 32 for every
    // token, so the blend of a tap is the exact integer  256 * 255 * value = H0 W0 a + H0 W1 b + H1 W0 c + H1 W1 d
    // on the pixel CODES (each product weight <= 7 * 31 fits a byte); 1 / 65280 is folded into the conv scales
    // (oracle/ita_oracle.c ita_oracle_tokenizer_u8).  No k / 255 table, no float blend.
    const unsigned H1 = (unsigned)(8.0f * g.h1) & 7u, W1 = (unsigned)(32.0f * g.w1) & 31u, H0 = 8u - H1, W0 = 32u - W1;
    const unsigned w00 = (H0 * W0) & 255u, w01 = (H0 * W1) & 255u, w10 = (H1 * W0) & 255u, w11 = (H1 * W1) & 255u;
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the window is private to this wave
    __builtin_amdgcn_wave_barrier();
    unsigned pb[13];
#pragma unroll
    for (int s = 0; s < 13; ++s) {
      if constexpr (ITA_ABLATE & 512) { pb[s] = w00 * s; continue; }
      const int off = tap[4 * s + kq];    // ky * 96 + kx of tap 4s + kq (0 for the pad taps 49..51: their weights are 0)
      pb[s] = (unsigned)win[off] * w00 + (unsigned)win[off + 2] * w01 + (unsigned)win[off + 192] * w10 +
              (unsigned)win[off + 194] * w11;
    }
    tok_u8_fragments(pb, tk_a0, tk_a1);
  };
  // one 16-channel tile of the conv: six int8 MFMAs + the float recombination, placed between the phases of the int8
  // blocks by the caller
  auto tok_step = [&](int ct, int ol) {
    if constexpr (ITA_ABLATE & 1024) { tk_out[4 * ct] = __int_as_float(tk_a0[ct]); return; }
    float o4[4];
    tok_u8_tile<E>(lds + L::CW, ct, ol, ol >> 4, tk_a0, tk_a1, o4);
#pragma unroll
    for (int i = 0; i < 4; ++i) tk_out[4 * ct + i] = o4[i];
  };
  auto tok_finish = [&](int fb, bool store_tap, float (&xr)[EC], int ol) {
    const int qi = ol & 15, kq = ol >> 4, token = wave * 16 + qi;
#pragma unroll
    for (int i = 0; i < 16; ++i) xr[i] = tk_out[i];
    layernorm_q16<E>(xr, lnp + 4 * E, lnp + 5 * E, EC * kq);
    if (a.tok_tap && store_tap) {
      float* o = a.tok_tap + ((size_t)fb * S + token) * E + EC * kq;
#pragma unroll
      for (int i = 0; i < EC; i += 4) *(f32x4*)(o + i) = (f32x4){xr[i], xr[i + 1], xr[i + 2], xr[i + 3]};
    }
  };

  // diagnostic: s_memrealtime (100 MHz, one clock for the whole chip) at kernel entry [12], prologue end [13], exit [14]
  if (STAMP && a.stamps && (tid & 255) == 0)
    a.stamps[(((size_t)blockIdx.x * 8) * 2 + (wave >> 2)) * 16 + 12] = __builtin_amdgcn_s_memrealtime();
  // ---- once per workgroup.  Order matters (it is 10+ % of a 4-frame launch): the first frame's pixels are requested
  // first, then the tokenizer's tables (the tail of the image), then the 84 KB of weights -- which stay in flight, in
  // registers, while the first frame is tokenized, and only then go to LDS.
  float xr[EC];
  i32x4 xq_cur[NK];       // IO8: this lane's k-slots of the block input, as they stand in memory
  f32x4 h0_cur = {0.0f, 0.0f, 0.0f, 0.0f};
  int h0_row_next = 0;
  {
    int ol = lane;
    asm volatile("" : "+v"(ol));
#ifndef ITA_PROLOGUE_ORDER
#define ITA_PROLOGUE_ORDER 1     // 1: weights before pixels (round 3); 0: the round-2 order, pixels first
#endif
    if constexpr (TOK != 0) {
      if (ITA_PROLOGUE_ORDER == 0) tok_fetch(blockIdx.x, ol);
    } else if constexpr (IO8) {
      const int8_t* xrow = a.xq + ((size_t)blockIdx.x * S + wave * 16 + (ol & 15)) * E + EC * (ol >> 4);
#pragma unroll
      for (int c = 0; c < NK; ++c) xq_cur[c] = *(const i32x4*)(xrow + 16 * c);
    } else {
      ld_tok_items<E>(a.x + ((size_t)blockIdx.x * S + wave * 16 + (ol & 15)) * E, ol >> 4, xr);   // transposed where it is first used
    }
    if (a.h0_dst && tid < 32) {
      const size_t row = a.slots ? (size_t)a.slots[blockIdx.x] : (size_t)blockIdx.x;
      h0_cur = *(const f32x4*)(a.h0_src + row * 128 + 4 * tid);
    }
    constexpr int T0 = L::LNP / 16, NT = (L::IMAGE - L::LNP) / 16, NTJ = (NT + 511) / 512;   // tables: LayerNorm, conv, taps
    constexpr int NW = L::LNP / 16, NWJ = (NW + 511) / 512;                                     // weights + biases
    static_assert(L::LNP % 16 == 0, "");
    i32x4 vt[NTJ], vw[NWJ];
#pragma unroll
    for (int j = 0; j < NTJ; ++j) {
      const int p = tid + 512 * j;
      vt[j] = (i32x4){0, 0, 0, 0};
      if (p < NT) vt[j] = *(const i32x4*)(a.image + (size_t)(T0 + p) * 16);
    }
#pragma unroll
    for (int j = 0; j < NWJ; ++j) {
      const int p = tid + 512 * j;
      vw[j] = (i32x4){0, 0, 0, 0};
      if (p < NW) vw[j] = *(const i32x4*)(a.image + (size_t)p * 16);
    }
    // Loads return in order.  The image (tables, weights) is L2-resident after the first workgroups, the first frame's pixels come
    // from HBM: requested LAST, the 100 KB of the image reach LDS while the pixels are still on their way, and the tokenizer of the
    // first frame is the only thing left between their arrival and the frame loop (the round-2 order -- pixels first, weights
    // parked in registers until the tokenizer was done -- put the weights' LDS stores behind it)
    if constexpr (TOK != 0) {
      if (ITA_PROLOGUE_ORDER != 0) tok_fetch(blockIdx.x, ol);
    }
    if (tid < 2 * P) colsum[tid] = ITA_ACC_BIAS;
#pragma unroll
    for (int j = 0; j < NTJ; ++j) {
      const int p = tid + 512 * j;
      if (p < NT) *(i32x4*)(lds + (T0 + p) * 16) = vt[j];
    }
    if constexpr (TOK != 0 && ITA_PROLOGUE_ORDER != 0) {
#pragma unroll
      for (int j = 0; j < NWJ; ++j) {
        const int p = tid + 512 * j;
        if (p < NW) *(i32x4*)(lds + p * 16) = vw[j];
      }
    }
    if constexpr (TOK != 0) {
      tok_fill(ol);
      lds_barrier();   // tables (and, in the round-3 order, weights) in place
      tok_blend(ol);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) tok_step(ct, ol);
      tok_finish(blockIdx.x, true, xr, ol);
    }
    if constexpr (!(TOK != 0 && ITA_PROLOGUE_ORDER != 0)) {
#pragma unroll
      for (int j = 0; j < NWJ; ++j) {
        const int p = tid + 512 * j;
        if (p < NW) *(i32x4*)(lds + p * 16) = vw[j];
      }
      lds_barrier();   // weights, biases and column-sum bases in place
    }
    if constexpr (TOK == 0 && !IO8) tok_items_transpose<E>(xr);   // the first frame's rows were fetched as items (ld_tok_items)
  }
  if (STAMP && a.stamps && (tid & 255) == 0)
    a.stamps[(((size_t)blockIdx.x * 8) * 2 + (wave >> 2)) * 16 + 13] = __builtin_amdgcn_s_memrealtime();

  for (int b = blockIdx.x; b < a.B; b += gridDim.x, ++fi) {
    int ol = lane;
    asm volatile("" : "+v"(ol));   // see the tokenizer above: its per-lane values are re-derived each frame
    // E = 128 holds twice the activations in registers: there the LDS bases are re-derived per frame as well
    const int qi = E > 64 ? (ol & 15) : qi0, kq = E > 64 ? (ol >> 4) : kq0, token = wave * 16 + qi;
    const int nb = b + gridDim.x;
    const bool more = nb < a.B;   // (uniform) the next frame exists: its tokenizer pieces run between this frame's phases
    int* cs = colsum + (fi & 1) * P;
    ITA_SSTAMP(0);
    // ---------------- quantise: this lane's E/4 channels are its k-slots of the B fragment(s)
    i32x4 xf[NK];
#pragma unroll
    for (int c = 0; c < NK; ++c) {
      if constexpr (IO8) {
        xf[c] = xq_cur[c];
      } else {
        unsigned p4[4];
        q_pack16(&xr[16 * c], a.inv_sx, p4);
        xf[c] = (i32x4){(int)p4[0], (int)p4[1], (int)p4[2], (int)p4[3]};
      }
    }
    // side copy of the LSTM layer-0 state row (see ItaStreamArgs): the row was requested a phase ago -- a load waited
    // for here would hold wave 0, and with it the whole workgroup's next barrier, for a full memory latency
    if (!(ITA_ABLATE & 128) && a.h0_dst && tid < 32) {
      *(f32x4*)(a.h0_dst + (size_t)b * 128 + 4 * ol) = h0_cur;   // (tid < 32: tid == ol; the opaque copy keeps the address out of the loop-invariant set)
      if (nb < a.B) h0_row_next = a.slots ? a.slots[nb] : nb;
    }

    // ---------------- projections, nine groups: Q0-2 (stay in registers), K0-2 and V0-2 (this wave's 16 rows of the
    // K image / 16 columns of the V^T image)
    i32x4 qf[3];
    {
      ItaFr<4 * NK> fr[2];
      i32x4 ac[3][4];
      auto load = [&](int g, ItaFr<4 * NK>& f, i32x4 (&acc)[4]) {
        const int mat = g / 3, t0 = 4 * (g - 3 * mat);
        if (mat == 0) ld_nat<NK, P>(f, acc, lds + L::WQ, l_bq, t0, qi, kq);
        else if (mat == 1) ld_nat<NK, P>(f, acc, lds + L::WK, l_bk, t0, qi, kq);
        else ld_natv<NK, P>(f, acc, lds + L::WV, (const int*)(lds + L::VB4), t0, qi, kq);
      };
      auto epilogue = [&](int g, const i32x4 (&acc)[4]) {
        const int mat = g / 3, gg = g - 3 * mat;
        if (mat == 0) {
          qf[gg] = rq_group(acc, a.mq, fq);
        } else if (mat == 1) {
          *(i32x4*)(lds + L::K + (((4 * gg + kq) * S + token) << 4)) = rq_group(acc, a.mk, fk);
        } else {
          // V with the roles swapped (A = tokens, B = weights): lane (feature qi, kq) holds keys 16w + 4kq + i of
          // feature 16 dt + qi -- one dword of the V^T slot (key block w>>2, k-group kq), at word w&3
          const i32x4 p4 = rq_group(acc, a.mv, fv);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int d = (4 * gg + t) * 16 + qi;
            *(int*)(lds + L::VT + (((((wave >> 2) * 4 + kq) * P) + d) << 4) + 4 * (wave & 3)) = p4[t];
            // column sum of the requantised codes for the unsigned-probability offset, kept pre-multiplied by 128
            atomicAdd(&cs[d], __builtin_amdgcn_sdot4(p4[t], 0x01010101, 0, false) << 7);
          }
        }
      };
      // (three accumulator sets: group g+1's start from its bias while g computes and g-1 is requantised)
      load(0, fr[0], ac[0]);
#pragma unroll
      for (int g = 0; g < 9; ++g) {
        if (g + 1 < 9) load(g + 1, fr[(g + 1) & 1], ac[(g + 1) % 3]);
        if (g < 6) mm_group<NK, false>(fr[g & 1], xf, ac[g % 3]);
        else mm_group<NK, true>(fr[g & 1], xf, ac[g % 3]);
        if (g > 0) epilogue(g - 1, ac[(g - 1) % 3]);
        ITA_SCHED_BARRIER();   // keep the step's loads ahead of the next step's MFMAs
      }
      epilogue(8, ac[8 % 3]);
    }
    ITA_SSTAMP(1);
    if constexpr (!(ITA_ABLATE & 64)) lds_barrier();   // B1: K, V^T and the column sums of this frame are complete
    ITA_SSTAMP(2);

    // next frame's input: issued now, consumed after the attention phase
    if (!(ITA_ABLATE & 128) && a.h0_dst && tid < 32 && nb < a.B)
      h0_cur = *(const f32x4*)(a.h0_src + (size_t)h0_row_next * 128 + 4 * ol);
    float xn[EC];
    i32x4 xq_nxt[NK];
    if constexpr (TOK != 0) {
      if (nb < a.B) tok_fetch(nb, ol);
    } else if constexpr (IO8) {
      const int8_t* xnrow = a.xq + ((size_t)min(nb, a.B - 1) * S + token) * E + EC * kq;
#pragma unroll
      for (int c = 0; c < NK; ++c) xq_nxt[c] = *(const i32x4*)(xnrow + 16 * c);
    } else {
      ld_tok_items<E>(a.x + ((size_t)min(nb, a.B - 1) * S + token) * E, kq, xn);   // items; transposed at the end of the frame
    }

    // ---------------- attention for this wave's 16 queries: logits and probabilities stay in registers
    i32x4 cf[3];
    constexpr int EG = NTE / 4;   // groups of four output tiles of a block output projection
    ItaF4 ob[2];
    i32x4 oa[4];
    {
      // logits as packed signed 16-bit pairs: the integer softmax then runs on v_pk_*_16, two keys per
      // VALU op.  w[2kt + j] = {logit 4kt+2j, logit 4kt+2j+1} of keys 16kt + 4kq + ...
      typedef ita_u16x2 u16x2;
      u16x2 w[16];
      struct KFr { i32x4 w[6]; } kfr[2];   // two key tiles x three k-steps
      ItaF4 vb[2];
      i32x4 va[3][4];
      auto ldk = [&](int p, KFr& f) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) f.w[t * 3 + ks] = sfrag(lds + L::K, ((4 * ks + kq) * S + (2 * p + t) * 16 + qi) << 4);
      };
      // A.V step st = (feature group dg = st >> 1, key block kb = st & 1): four V^T fragments; the accumulators of a
      // feature group start from 128 * column sum
      auto ldv = [&](int st, ItaF4& f) {
        const int dg = st >> 1, kb = st & 1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          f.w[t] = sfrag(lds + L::VT, (((kb * 4 + kq) * P) + (4 * dg + t) * 16 + qi) << 4);
          if (kb == 0) va[dg][t] = *(const i32x4*)(cs + (4 * dg + t) * 16 + 4 * kq);
        }
      };
      i32x4 la[2][2];
      auto mmk = [&](const KFr& f, i32x4 (&acc)[2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[t] = (i32x4){ITA_ACC_BIAS, ITA_ACC_BIAS, ITA_ACC_BIAS, ITA_ACC_BIAS};
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) acc[t] = mfma16(f.w[t * 3 + ks], qf[ks], acc[t]);
        }
      };
      auto epil = [&](int p, const i32x4 (&acc)[2]) {
        // low 16 bits = rne(logit) + 32768, unclamped: the softmax clamps (ita_softmax_packed16)
        unsigned w4[4];
        if (fl) lg8_v3<ITA_RQ_FAST>(acc, a.ml, w4);
        else lg8_v3<ITA_RQ_EXACT>(acc, a.ml, w4);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int j = 0; j < 2; ++j) w[2 * (2 * p + t) + j] = __builtin_bit_cast(u16x2, w4[2 * t + j]);
      };
      ldk(0, kfr[0]);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (p + 1 < 4) ldk(p + 1, kfr[(p + 1) & 1]);
        else ldv(0, vb[0]);
        mmk(kfr[p & 1], la[p & 1]);
        if (p > 0) epil(p - 1, la[(p - 1) & 1]);
        ITA_SCHED_BARRIER();   // keep the step's loads ahead of the next step's MFMAs
      }
      epil(3, la[1]);
      ITA_SSTAMP(3);
      i32x4 pf[2];
      if constexpr (ITA_ABLATE & 32) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int t = 0; t < 4; ++t) pf[kb][t] = __builtin_bit_cast(int, w[2 * (4 * kb + t)]);
      } else {
        ita_softmax_packed16(w, pf);
      }
      ITA_SSTAMP(4);
      // A.V with uint8 probabilities on a signed MFMA: (p - 128) * v summed + 128 * colsum(v)
#pragma unroll
      for (int st = 0; st < 6; ++st) {
        if (st + 1 < 6) ldv(st + 1, vb[(st + 1) & 1]);
        mm_ks(vb[st & 1], pf[st & 1], va[st >> 1]);
        if (st == 3) cf[0] = rq_group(va[0], a.mc, fc);
        ITA_SCHED_BARRIER();   // keep the step's loads ahead of the next step's MFMAs
      }
      ITA_SSTAMP(5);
      if constexpr (!(ITA_ABLATE & 64)) lds_barrier();   // B2: every wave is done with K, V^T and this frame's column sums
      ITA_SSTAMP(6);
      if (tid < P) cs[tid] = ITA_ACC_BIAS;   // this parity is next accumulated after the NEXT frame's B2
      ld_frg_ks<E>(ob[0], lds + L::WO, 0, 0, qi, kq);
      ld_obias<E>(oa, l_bo, 0, kq);
      // the next frame's pixel window, before this frame's global stores (one in-order vmcnt for loads and stores)
      if constexpr (TOK != 0) { if (more) tok_fill(ol); }
      cf[1] = rq_group(va[1], a.mc, fc);
      ld_frg_ks<E>(ob[1], lds + L::WO, 0, 1, qi, kq);
      mm_ks(ob[0], cf[0], oa);
      cf[2] = rq_group(va[2], a.mc, fc);
    }
    if constexpr (TOK != 0 && !(ITA_ABLATE & 4) && !(ITA_TOK_PLACE == 0 && ITA_TOK_BLEND_END)) { if (more) tok_blend(ol); }
    if constexpr (TOK != 0 && ITA_TOK_PLACE == 2 && !(ITA_ABLATE & 4)) {
      if (more) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) tok_step(ct, ol);
      }
    }

    // ---------------- out_proj + residual + LayerNorm1 (k-step 0 of the first group is already in flight)
    const char* w1p = lds + L::W1;   // fc1 / fc2 weights: LDS image, or (E = 128) the global image behind it
    const char* w2p = lds + L::W2;
    if constexpr (L::W12G) { w1p = a.image + L::GW1; w2p = a.image + L::GW2; }
    float x1[EC];
    ItaFr<4 * NK> ffr[2];
    i32x4 fa[3][4];
#pragma unroll
    for (int eg = 0; eg < EG; ++eg) {
      // k-steps 3 eg + ks; fragments are fetched one k-step ahead
#pragma unroll
      for (int ks = (eg == 0 ? 1 : 0); ks < 3; ++ks) {
        const int cur = (3 * eg + ks) & 1;
        if (ks + 1 < 3) ld_frg_ks<E>(ob[cur ^ 1], lds + L::WO, 4 * eg, ks + 1, qi, kq);
        else if (eg + 1 < EG) ld_frg_ks<E>(ob[cur ^ 1], lds + L::WO, 4 * (eg + 1), 0, qi, kq);
        else if constexpr (FFN) ld_nat<NK, F>(ffr[0], fa[0], w1p, l_b1, 0, qi, kq);
        mm_ks(ob[cur], cf[ks], oa);
        ITA_SCHED_BARRIER();   // keep the step's loads ahead of the next step's MFMAs
      }
      if constexpr (TOK != 0 && ITA_TOK_MID && !(ITA_ABLATE & 4)) { if (more) tok_step(0, ol); }
      if constexpr (IO8) {   // the int8 codes themselves: 16 channels of this lane's token, one 16-byte store
        *(i32x4*)(a.yq + ((size_t)b * S + token) * E + EC * kq + 16 * eg) = rq_group(oa, a.mo, fo);
        if (eg + 1 < EG) ld_obias<E>(oa, l_bo, 4 * (eg + 1), kq);
        continue;
      }
      float d[16];
      dq_group(oa, a.mo, a.so, d);
      if (eg + 1 < EG) ld_obias<E>(oa, l_bo, 4 * (eg + 1), kq);
#pragma unroll
      for (int j = 0; j < 16; ++j) x1[16 * eg + j] = (FFN || a.fuse_ln) ? xr[16 * eg + j] + d[j] : d[j];
    }
    if constexpr (IO8) {
#pragma unroll
      for (int c = 0; c < NK; ++c) xq_cur[c] = xq_nxt[c];
      continue;
    }
    if (FFN || a.fuse_ln) layernorm_q16<E>(x1, lnp, lnp + E, EC * kq);
    if (a.x1_tap) {
      float* o = a.x1_tap + ((size_t)b * S + token) * E + EC * kq;
#pragma unroll
      for (int i = 0; i < EC; i += 4) *(f32x4*)(o + i) = (f32x4){x1[i], x1[i + 1], x1[i + 2], x1[i + 3]};
    }
    ITA_SSTAMP(7);

    float yv[EC];
    if constexpr (FFN) {
      // ---------------- FFN: fc1 + ReLU (hidden layer = four B fragments in registers) -> fc2 -> LayerNorm2.
      // fc2's k-step ks consumes hidden fragment ks: its MFMAs are issued one step behind fc1's epilogue.
      i32x4 x1f[NK];
#pragma unroll
      for (int c = 0; c < NK; ++c) {
        unsigned p4[4];
        q_pack16(&x1[16 * c], a.f_inv_sx, p4);
        x1f[c] = (i32x4){(int)p4[0], (int)p4[1], (int)p4[2], (int)p4[3]};
      }
      i32x4 hf[4], ya[4];
      ItaF4 wb[2];
      ld_obias<E>(ya, l_b2, 0, kq);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g + 1 < 4) ld_nat<NK, F>(ffr[(g + 1) & 1], fa[(g + 1) % 3], w1p, l_b1, 4 * (g + 1), qi, kq);
        if (g >= 1) ld_frg_ks<E>(wb[(g - 1) & 1], w2p, 0, g - 1, qi, kq);
        mm_group<NK, false>(ffr[g & 1], x1f, fa[g % 3]);
        if (g >= 2) mm_ks(wb[(g - 2) & 1], hf[g - 2], ya);
        if constexpr (TOK != 0 && ITA_TOK_MID && !(ITA_ABLATE & 4)) { if (g < 3 && more) tok_step(1 + g, ol); }
        if (g >= 1) hf[g - 1] = rq_group<true>(fa[(g - 1) % 3], a.m1, false);
        ITA_SCHED_BARRIER();   // keep the step's loads ahead of the next step's MFMAs
      }
      ld_frg_ks<E>(wb[1], w2p, 0, 3, qi, kq);
      mm_ks(wb[0], hf[2], ya);
      hf[3] = rq_group<true>(fa[3 % 3], a.m1, false);
      mm_ks(wb[1], hf[3], ya);
      ITA_SSTAMP(8);
      {
        float d[16];
        dq_group(ya, a.m2, a.s2, d);
#pragma unroll
        for (int j = 0; j < 16; ++j) yv[j] = x1[j] + d[j];
      }
      // E = 128: the second half of the output channels, from the same hidden fragments
#pragma unroll
      for (int eg = 1; eg < EG; ++eg) {
        ld_obias<E>(ya, l_b2, 4 * eg, kq);
        ld_frg_ks<E>(wb[0], w2p, 4 * eg, 0, qi, kq);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks + 1 < 4) ld_frg_ks<E>(wb[(ks + 1) & 1], w2p, 4 * eg, ks + 1, qi, kq);
          mm_ks(wb[ks & 1], hf[ks], ya);
          ITA_SCHED_BARRIER();
        }
        float d[16];
        dq_group(ya, a.m2, a.s2, d);
#pragma unroll
        for (int j = 0; j < 16; ++j) yv[16 * eg + j] = x1[16 * eg + j] + d[j];
      }
      layernorm_q16<E>(yv, lnp + 2 * E, lnp + 3 * E, EC * kq);
    } else {
#pragma unroll
      for (int i = 0; i < EC; ++i) yv[i] = x1[i];
    }
    {
      if (a.y) st_tok_quarter<E>(a.y + ((size_t)b * S + token) * E, kq, yv);
      if (!(ITA_ABLATE & 128) && a.y_hi) {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int i = 0; i < EC; i += 8) {
          h8 vh, vl;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const _Float16 hh = (_Float16)yv[i + j];
            vh[j] = hh;
            vl[j] = (_Float16)(yv[i + j] - (float)hh);
          }
          const size_t po = (size_t)b * a.ld_planes + token * E + EC * kq + i;
          if constexpr (ITA_ABLATE & 256) {   // values stay live, one 2-byte store instead of two 16-byte ones
            a.y_hi[po] = vh[0] + vh[1] + vh[2] + vh[3] + vh[4] + vh[5] + vh[6] + vh[7] + vl[0] + vl[1] + vl[2] + vl[3] + vl[4] + vl[5] + vl[6] + vl[7];
          } else {
            *(h8*)(a.y_hi + po) = vh;
            *(h8*)(a.y_lo + po) = vl;
          }
        }
      }
    }
    ITA_SSTAMP(11);
    // ---------------- the next frame's tokens
    if constexpr (TOK != 0 && !(ITA_ABLATE & 4)) {
      if constexpr (ITA_TOK_PLACE == 0) {
        if constexpr (ITA_TOK_BLEND_END) { if (more) tok_blend(ol); }   // (the window written after B2 is still in LDS: private to this wave)
        if (more) {
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) tok_step(ct, ol);
        }
      }
      if (more) tok_finish(nb, true, xr, ol);
    } else {
      tok_items_transpose<E>(xn);
#pragma unroll
      for (int i = 0; i < EC; ++i) xr[i] = xn[i];
    }
  }
  if (STAMP && a.stamps && (tid & 255) == 0)
    a.stamps[(((size_t)blockIdx.x * 8) * 2 + (wave >> 2)) * 16 + 14] = __builtin_amdgcn_s_memrealtime();
#undef ITA_SSTAMP
#undef fq
#undef fk
#undef fv
#undef fl
#undef fc
#undef fo
}

// ------------------------------------------------------------------ the tokenizer on its own
// OverlapPatchMerging (reference models/ITA/QAT/layers.py:39-45) with the structure of the fused tokenizer above -- per-wave
// 9-row pixel windows, the blended taps as the lane's own MFMA B operand, conv weights as v_mfma_f32_16x16x4_f32 A fragments,
// LayerNorm in registers -- for the cases the fused form does not cover: E = 128 (its encoder kernel has no LDS left for
// the 27 KB of conv fragments), f32 frames (module.main_graph's own input type: four times the window bytes) and callers of
// ita_tokenizer.  Wave w = token row w of the 8 x 16 grid; one persistent 512-thread workgroup per CU, the next frame's
// pixels requested a frame ahead.
//   U8  : the exact integer blend of the pixel codes and the conv as int8 MFMA on 23-bit fixed-point weights (ItaTokTab,
//         oracle: ita_oracle_tokenizer_u8) -- the fused tokenizer's functions, bit-identical to it;
//   !U8 : the oracle's float blend  h0 (w0 a + w1 b) + h1 (w0 c + w1 d)  of pixels already scaled by the caller
//         (ita_oracle_tokenizer), conv weights as they are, v_mfma_f32_16x16x4_f32; same results as
//         ita_tokenizer_kernel<E, false> bit for bit (same operation order), at a fraction of its time.
template <int E, bool U8>
struct ItaTokStreamLds {
  static constexpr int NCT = E / 16;                           // 16-channel output tiles
  static constexpr int LNP = 0;                                // f32: ln_w | ln_b
  static constexpr int CW = LNP + 2 * E * 4;                   // !U8: f32 [13][NCT][64] conv weights as A fragments; U8: ItaTokTab<E>
  static constexpr int CB = CW + (U8 ? ItaTokTab<E>::BYTES : 13 * NCT * 64 * 4);   // !U8: f32 [E] conv bias (U8: inside the table)
  static constexpr int TAP = CB + (U8 ? 0 : E * 4);            // int32 [52]
  static constexpr int IMAGE = TAP + 52 * 4;
  static constexpr int IMG = (IMAGE + 15) & ~15;               // [8 waves][9][96] pixels (u8 or f32), 3 zero columns each side
  static constexpr int IMG_WAVE = 9 * 96 * (U8 ? 1 : 4);
  static constexpr int TOTAL = IMG + 8 * IMG_WAVE;
};
struct ItaTokStreamArgs {
  const char* image;     // device copy of the LDS image (ItaTokStreamLds<E, U8>::IMAGE bytes)
  const void* img;       // (B,60,90) u8 or f32
  float* tokens;         // (B,128,E)
  int B;
};
template <int E, bool U8>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4))) void ita_tok_stream_kernel(const ItaTokStreamArgs a) {
  using L = ItaTokStreamLds<E, U8>;
  constexpr int S = 128, EC = E / 4, NCT = L::NCT;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* lnp = (const float*)(lds + L::LNP);
  // geometry of this lane: its token (row = wave, column = lane & 15), its piece of the 9 x 96 window
  int y0, yp, x0, xp;
  float h1, w1;
  bilinear_src_dev(wave, 30.0f / 8.0f, 30, y0, yp, h1);
  bilinear_src_dev(lane & 15, 45.0f / 16.0f, 45, x0, xp, w1);
  const int rr = lane / 6, pc = lane - 6 * rr;           // u8: window piece of this lane (lane < 54)
  const int row = 2 * y0 - 3 + rr;                        // image row (-1 for the top row of wave 0)
  const int o = row * 90 + 16 * pc - 3;                   // frame byte of the piece's first window byte
  const int kq = lane >> 4, qi = lane & 15;
  // f32 frames: window element e = lane + 64 j (j < 14) = (window row e / 96, window column e % 96)
  constexpr int NF = U8 ? 1 : 14;
  unsigned tk_d[5] = {0, 0, 0, 0, 0};
  float tk_f[NF];
  auto fetch = [&](int fb) {
    if constexpr (U8) {
      const uint8_t* src = (const uint8_t*)a.img + (size_t)fb * 5400;
      const int a0 = o & ~3;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int aj = a0 + 4 * j;
        tk_d[j] = 0;
        if (lane < 54 && row >= 0 && aj >= 0 && aj < 5400) tk_d[j] = *(const unsigned*)(src + aj);
      }
    } else {
      const float* src = (const float*)a.img + (size_t)fb * 5400;
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int e = lane + 64 * j, wr = e / 96, wc = e - 96 * wr;
        const int iy = 2 * y0 - 3 + wr, ix = wc - 3;
        tk_f[j] = 0.0f;
        if (e < 9 * 96 && iy >= 0 && iy < 60 && ix >= 0 && ix < 90) tk_f[j] = src[iy * 90 + ix];
      }
    }
  };
  if ((int)blockIdx.x < a.B) fetch(blockIdx.x);
  for (int p = tid; p < L::IMAGE / 16; p += 512) *(i32x4*)(lds + p * 16) = *(const i32x4*)(a.image + (size_t)p * 16);
  __syncthreads();
  const unsigned H1 = (unsigned)(8.0f * h1) & 7u, W1 = (unsigned)(32.0f * w1) & 31u, H0 = 8u - H1, W0 = 32u - W1;
  const unsigned w00 = (H0 * W0) & 255u, w01 = (H0 * W1) & 255u, w10 = (H1 * W0) & 255u, w11 = (H1 * W1) & 255u;
  const float fh0 = 1.0f - h1, fw0 = 1.0f - w1;
  char* wbase = lds + L::IMG + wave * L::IMG_WAVE;
  const int* tap = (const int*)(lds + L::TAP);
  const float* cw = (const float*)(lds + L::CW);
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    if constexpr (U8) {   // window fill: funnel-shift to 16-byte pieces, mask the border
      const int sh = o & 3;
      unsigned o4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o4[j] = __builtin_amdgcn_alignbyte(tk_d[j + 1], tk_d[j], sh);
      if (pc == 0) o4[0] &= 0xff000000u;
      if (pc == 5) o4[3] &= 0x000000ffu;
      if (lane < 54) *(i32x4*)(wbase + rr * 96 + 16 * pc) = (i32x4){(int)o4[0], (int)o4[1], (int)o4[2], (int)o4[3]};
    } else {
#pragma unroll
      for (int j = 0; j < NF; ++j)
        if (lane + 64 * j < 9 * 96) ((float*)wbase)[lane + 64 * j] = tk_f[j];
    }
    if (b + (int)gridDim.x < a.B) fetch(b + gridDim.x);
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the window is private to this wave
    __builtin_amdgcn_wave_barrier();
    float xr[EC];
    if constexpr (U8) {
      const uint8_t* win = (const uint8_t*)wbase + 2 * x0;
      unsigned pb[13];
#pragma unroll
      for (int s = 0; s < 13; ++s) {
        const int off = tap[4 * s + kq];
        pb[s] = (unsigned)win[off] * w00 + (unsigned)win[off + 2] * w01 + (unsigned)win[off + 192] * w10 + (unsigned)win[off + 194] * w11;
      }
      i32x4 a0, a1;
      tok_u8_fragments(pb, a0, a1);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        float o4[4];
        tok_u8_tile<E>(lds + L::CW, ct, lane, kq, a0, a1, o4);
#pragma unroll
        for (int i = 0; i < 4; ++i) xr[4 * ct + i] = o4[i];
      }
    } else {
      float tk_pt[13];
#pragma unroll
      for (int s = 0; s < 13; ++s) {
        const int off = tap[4 * s + kq];
        const float* win = (const float*)wbase + 2 * x0;
        const float va = win[off], vb = win[off + 2], vc = win[off + 192], vd = win[off + 194];
        tk_pt[s] = fh0 * (fw0 * va + w1 * vb) + h1 * (fw0 * vc + w1 * vd);   // ita_oracle_blend_patch's expression
      }
      f32x4 acc[NCT];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[ct] = *(const f32x4*)(lds + L::CB + (EC * kq + 4 * ct) * 4);
#pragma unroll
      for (int s = 0; s < 13; ++s)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(cw[(s * NCT + ct) * 64 + lane], tk_pt[s], acc[ct], 0, 0, 0);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int i = 0; i < 4; ++i) xr[4 * ct + i] = acc[ct][i];
    }
    layernorm_q16<E>(xr, lnp, lnp + E, EC * kq);
    st_tok_quarter<E>(a.tokens + ((size_t)b * S + wave * 16 + qi) * E, kq, xr);   // 64 contiguous bytes per token and store
    __builtin_amdgcn_wave_barrier();      // the window is rewritten for the next frame only after these reads were issued
  }
}
