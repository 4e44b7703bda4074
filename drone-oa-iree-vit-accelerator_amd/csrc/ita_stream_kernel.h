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
// (f32 MFMA fragments) + k/255 table 1 KB | K 24 KB | V^T 24 KB | column sums | eight private 9 x 96
// byte windows of the next frame = 159 488 B of the CU's 160 KB.
#pragma once
#include "ita_int8_kernels.h"

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
  // optional side copy for the LSTM that follows (see ita_encoder_kernel.h): h0_src row -> h0_dst[b]
  const float* h0_src;
  float* h0_dst;
  const int* slots;
  const void* img;        // (B,60,90) u8 wire frames (TOK == 1)
  float* tok_tap;         // optional (B,128,E): the tokens
};

template <int E, bool FFN, bool TOK>
struct ItaStreamLds {
  static constexpr int S = 128, P = 192, F = 256;
  // ---- image: copied verbatim from global memory once per workgroup
  static constexpr int WQ = 0;                       // int8 [E/16][192][16]  chunk-major, natural k
  static constexpr int WK = WQ + P * E;
  static constexpr int WV = WK + P * E;
  static constexpr int WO = WV + P * E;              // int8 [12][E][16]     fragment order (see build_stream_image)
  static constexpr int W1 = WO + E * P;              // int8 [E/16][256][16] natural k
  static constexpr int W2 = W1 + (FFN ? F * E : 0);  // int8 [16][E][16]     fragment order
  static constexpr int BIAS = W2 + (FFN ? E * F : 0);   // int32: bq | bk | bv | bo | b1 | b2
  static constexpr int NBIAS = 3 * P + E + (FFN ? F + E : 0);
  static constexpr int LNP = BIAS + NBIAS * 4;       // f32: n1w | n1b | n2w | n2b | tok_lnw | tok_lnb
  static constexpr int NLN = 2 * E + (FFN ? 2 * E : 0) + (TOK ? 2 * E : 0);
  static constexpr int CW = LNP + NLN * 4;           // f32 [13][4][64]: conv7x7 weights as 16x16x4 A fragments
  static constexpr int CB = CW + (TOK ? 13 * 4 * 64 * 4 : 0);   // f32 [E]: conv bias
  static constexpr int IMAGE = CB + (TOK ? E * 4 : 0);
  // ---- built / used at run time
  static constexpr int LUT = IMAGE;                  // f32 [256]: k / 255.0f
  static constexpr int K = LUT + (TOK ? 1024 : 0);   // int8 [12][128][16]  fragment order
  static constexpr int VT = K + S * P;               // int8 [8][192][16]   V^T, keys permuted (ita_int8_kernels.h)
  static constexpr int COLSUM = VT + P * S;          // int32 [2][192]: 128 * column sums of V, per frame parity
  static constexpr int IMG = COLSUM + 2 * P * 4;     // u8 [8 waves][9][96]: rows 2*y0-3 .. 2*y0+5 of the next frame, 3 zero columns each side
  static constexpr int IMG_WAVE = 9 * 96;
  static constexpr int TOTAL = IMG + (TOK ? 8 * IMG_WAVE : 0);
  static_assert(IMAGE % 16 == 0 && K % 16 == 0 && IMG % 16 == 0, "16-byte alignment");
  static_assert(TOTAL <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ i32x4 mfma16(i32x4 a, i32x4 b, i32x4 c) {
  return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
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

// Four 16-feature output tiles (features 16*tile0 .. 16*tile0+63) of  W . x^T  for this wave's 16 tokens:
// A = weight rows from a natural-k chunk-major image [E/16][ROWS][16], B = the wave's activation fragments.
// Requantised and packed: the result is ONE B fragment of the next GEMM (byte 4t+i <-> feature
// 16*(tile0+t) + 4*kq + i of token qi).
template <int NK, int ROWS>
__device__ __forceinline__ i32x4 wx_group(const char* w_img, const int* lds_bias, int tile0, const i32x4 (&xf)[NK],
                                          float mult, float lo, int qi, int kq) {
  i32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = *(const i32x4*)(lds_bias + (tile0 + t) * 16 + 4 * kq);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < NK; ++c)
      acc[t] = mfma16(lds_frag(w_img, ((NK * kq + c) * ROWS + (tile0 + t) * 16 + qi) << 4), xf[c], acc[t]);
  float f[16];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    float f4[4];
    scale_clamp<4>(acc[t], mult, lo, f4);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[4 * t + i] = f4[i];
  }
  unsigned p4[4];
  round_pack16(f, p4);
  return (i32x4){(int)p4[0], (int)p4[1], (int)p4[2], (int)p4[3]};
}

// Four output tiles et0 .. et0+3 of a block output projection (out_proj, fc2): A = weight rows from a
// fragment-order image [4*NKS][E][16] (row et*16 + rho <-> channel (E/4)(rho>>2) + 4 et + (rho&3), so that
// lane (qi, kq) receives its own channels (E/4)kq + 4 et + i), B = the NKS packed fragments of the previous
// GEMM.  Returns d[4t+i] = dequantised block output of channel (E/4)kq + 4(et0+t) + i.
template <int NKS, int E>
__device__ __forceinline__ void out_group(const char* w_img, const int* lds_bias, int et0, const i32x4 (&bf)[NKS],
                                          float mult, float scale, int qi, int kq, float (&d)[16]) {
  i32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = *(const i32x4*)(lds_bias + (E / 4) * kq + 4 * (et0 + t));
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      acc[t] = mfma16(lds_frag(w_img, ((4 * ks + kq) * E + (et0 + t) * 16 + qi) << 4), bf[ks], acc[t]);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    float f4[4];
    scale_clamp<4>(acc[t], mult, -128.0f, f4);
    // rne of the clamped value is the int8 code; its float is what dequantisation multiplies
#pragma unroll
    for (int i = 0; i < 4; ++i) d[4 * t + i] = __builtin_rintf(f4[i]) * scale;
  }
}

template <int E, bool FFN, int TOK>
__global__ __launch_bounds__(512) void ita_stream_kernel(const ItaStreamArgs a) {
  using L = ItaStreamLds<E, FFN, TOK != 0>;
  constexpr int S = 128, P = 192, F = 256, EC = E / 4, NK = E / 64, NTE = E / 16;
  static_assert(!TOK || (E == 64 && FFN), "the fused tokenizer is built for the ITAViTLSTM shape");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = lane & 15, kq = lane >> 4;
  const int token = wave * 16 + qi;
  const int* bias = (const int*)(lds + L::BIAS);
  const int *l_bq = bias, *l_bk = bias + P, *l_bv = bias + 2 * P, *l_bo = bias + 3 * P, *l_b1 = bias + 3 * P + E,
            *l_b2 = bias + 3 * P + E + F;
  const float* lnp = (const float*)(lds + L::LNP);
  int* colsum = (int*)(lds + L::COLSUM);

  // ---- once per workgroup: the weight image, every load issued before the first LDS store
  {
    constexpr int NP = L::IMAGE / 16, N = (NP + 511) / 512;
    i32x4 v[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int p = tid + 512 * j;
      v[j] = (i32x4){0, 0, 0, 0};
      if (p < NP) v[j] = *(const i32x4*)(a.image + (size_t)p * 16);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int p = tid + 512 * j;
      if (p < NP) *(i32x4*)(lds + p * 16) = v[j];
    }
    if constexpr (TOK != 0) {
      if (tid < 256) ((float*)(lds + L::LUT))[tid] = (float)tid / 255.0f;   // the reference host's float(pixel) / 255.0f (main.cpp:168-169)
    }
    if (tid < 2 * P) colsum[tid] = 0;
  }

  int fi = 0;
#define ITA_SSTAMP(ph)                                                                                  \
  do {                                                                                                  \
    if (a.stamps && (tid & 255) == 0 && fi < 8)                                                         \
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
  int tk_y0 = 0, tk_x0 = 0;
  float tk_h1 = 0.0f, tk_w1 = 0.0f;
  unsigned tk_d[5] = {0, 0, 0, 0, 0};
  if constexpr (TOK != 0) {
    int yp, xp;
    bilinear_src_dev(wave, 30.0f / 8.0f, 30, tk_y0, yp, tk_h1);    // y0 <= 27 < 29 and x0 <= 43 < 44: the second
    bilinear_src_dev(qi, 45.0f / 16.0f, 45, tk_x0, xp, tk_w1);     // neighbour is always one row / column on
  }
  const int tk_rr = lane / 6, tk_pc = lane - 6 * tk_rr;            // window piece of this lane (lane < 54)
  const int tk_row = 2 * tk_y0 - 3 + tk_rr;                        // image row (-1 for the top row of wave 0)
  const int tk_o = tk_row * 90 + 16 * tk_pc - 3;                   // frame byte of the piece's first window byte
  auto tok_fetch = [&](int fb) {
    const uint8_t* src = (const uint8_t*)a.img + (size_t)fb * 5400;
    const int a0 = tk_o & ~3;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int aj = a0 + 4 * j;
      tk_d[j] = 0;
      if (lane < 54 && tk_row >= 0 && aj >= 0 && aj < 5400) tk_d[j] = *(const unsigned*)(src + aj);
    }
  };
  auto tok_fill = [&]() {
    const int sh = tk_o & 3;
    unsigned o4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o4[j] = __builtin_amdgcn_alignbyte(tk_d[j + 1], tk_d[j], sh);
    if (tk_pc == 0) o4[0] &= 0xff000000u;        // window columns 0..2  = image columns -3..-1
    if (tk_pc == 5) o4[3] &= 0x000000ffu;        // window columns 93..95 = image columns 90..92
    if (lane < 54)
      *(i32x4*)(lds + L::IMG + wave * L::IMG_WAVE + tk_rr * 96 + 16 * tk_pc) =
          (i32x4){(int)o4[0], (int)o4[1], (int)o4[2], (int)o4[3]};
  };
  auto tok_compute = [&](int fb, float (&xr)[EC]) {
    const float* lut = (const float*)(lds + L::LUT);
    const uint8_t* win = (const uint8_t*)(lds + L::IMG + wave * L::IMG_WAVE);
    const float h1 = tk_h1, h0 = 1.0f - tk_h1, w1 = tk_w1, w0 = 1.0f - tk_w1;
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the window is private to this wave
    __builtin_amdgcn_wave_barrier();
    float pt[13];
#pragma unroll
    for (int s = 0; s < 13; ++s) {
      const int t = 4 * s + kq;
      int off = 0;                        // taps 49..51 pad K to 52: the zero corner of the window
      if (t < 49) { const int ky = t / 7; off = ky * 96 + (t - 7 * ky) + 2 * tk_x0; }
      const float va = lut[win[off]], vb = lut[win[off + 2]], vc = lut[win[off + 192]], vd = lut[win[off + 194]];
      pt[s] = h0 * (w0 * va + w1 * vb) + h1 * (w0 * vc + w1 * vd);
    }
    ITA_SSTAMP(9);
    const float* cw = (const float*)(lds + L::CW);
    f32x4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = *(const f32x4*)(lds + L::CB + (16 * kq + 4 * ct) * 4);
#pragma unroll
    for (int s = 0; s < 13; ++s)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(cw[(s * 4 + ct) * 64 + lane], pt[s], acc[ct], 0, 0, 0);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) xr[4 * ct + i] = acc[ct][i];
    layernorm_q16<E>(xr, lnp + 4 * E, lnp + 5 * E, EC * kq);
    if (a.tok_tap) {
      float* o = a.tok_tap + ((size_t)fb * S + token) * E + EC * kq;
#pragma unroll
      for (int i = 0; i < EC; i += 4) *(f32x4*)(o + i) = (f32x4){xr[i], xr[i + 1], xr[i + 2], xr[i + 3]};
    }
    ITA_SSTAMP(10);
  };

  // ---- the first frame's input
  float xr[EC];
  if constexpr (TOK != 0) {
    if ((int)blockIdx.x < a.B) tok_fetch(blockIdx.x);
  } else if ((int)blockIdx.x < a.B) {
    const float* xrow = a.x + ((size_t)blockIdx.x * S + token) * E + EC * kq;
#pragma unroll
    for (int i = 0; i < EC; i += 4) {
      const f32x4 v = *(const f32x4*)(xrow + i);
      xr[i] = v.x; xr[i + 1] = v.y; xr[i + 2] = v.z; xr[i + 3] = v.w;
    }
  }
  lds_barrier();   // image, table and zeroed column sums are in place
  if constexpr (TOK != 0) {
    if ((int)blockIdx.x < a.B) {
      tok_fill();
      tok_compute(blockIdx.x, xr);
    }
  }

  for (int b = blockIdx.x; b < a.B; b += gridDim.x, ++fi) {
    const int nb = b + gridDim.x;
    int* cs = colsum + (fi & 1) * P;
    ITA_SSTAMP(0);
    // ---------------- quantise: this lane's E/4 channels are its k-slots of the B fragment(s)
    i32x4 xf[NK];
#pragma unroll
    for (int c = 0; c < NK; ++c) {
      unsigned p4[4];
      q_pack16(&xr[16 * c], a.inv_sx, p4);
      xf[c] = (i32x4){(int)p4[0], (int)p4[1], (int)p4[2], (int)p4[3]};
    }
    if (a.h0_dst && tid < 32) {
      const size_t row = a.slots ? (size_t)a.slots[b] : (size_t)b;
      *(f32x4*)(a.h0_dst + (size_t)b * 128 + 4 * tid) = *(const f32x4*)(a.h0_src + row * 128 + 4 * tid);
    }

    // ---------------- projections.  Q stays in registers; K and V^T rows of this wave's 16 tokens go to LDS.
    i32x4 qf[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) qf[g] = wx_group<NK, P>(lds + L::WQ, l_bq, 4 * g, xf, a.mq, -128.0f, qi, kq);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const i32x4 kf = wx_group<NK, P>(lds + L::WK, l_bk, 4 * g, xf, a.mk, -128.0f, qi, kq);
      *(i32x4*)(lds + L::K + (((4 * g + kq) * S + token) << 4)) = kf;
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      // V with the roles swapped (A = tokens, B = weights): lane (feature qi, kq) gets keys 16w + 4kq + i of
      // feature 16 dt + qi -- one dword of the V^T slot (key block w>>2, k-group kq), at word w&3
      i32x4 acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int bvv = l_bv[(4 * g + t) * 16 + qi];
        acc[t] = (i32x4){bvv, bvv, bvv, bvv};
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < NK; ++c)
          acc[t] = mfma16(xf[c], lds_frag(lds + L::WV, ((NK * kq + c) * P + (4 * g + t) * 16 + qi) << 4), acc[t]);
      float f[16];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float f4[4];
        scale_clamp<4>(acc[t], a.mv, -128.0f, f4);
#pragma unroll
        for (int i = 0; i < 4; ++i) f[4 * t + i] = f4[i];
      }
      unsigned p4[4];
      round_pack16(f, p4);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int d = (4 * g + t) * 16 + qi;
        *(unsigned*)(lds + L::VT + (((((wave >> 2) * 4 + kq) * P) + d) << 4) + 4 * (wave & 3)) = p4[t];
        // column sum of the requantised codes for the unsigned-probability offset, kept pre-multiplied by 128
        atomicAdd(&cs[d], __builtin_amdgcn_sdot4((int)p4[t], 0x01010101, 0, false) << 7);
      }
    }
    ITA_SSTAMP(1);
    lds_barrier();   // B1: K, V^T and the column sums of this frame are complete
    ITA_SSTAMP(2);

    // next frame's input: issued now, consumed after the attention phase
    float xn[EC];
    if constexpr (TOK != 0) {
      if (nb < a.B) tok_fetch(nb);
    } else {
      const float* xnrow = a.x + ((size_t)min(nb, a.B - 1) * S + token) * E + EC * kq;
#pragma unroll
      for (int i = 0; i < EC; i += 4) {
        const f32x4 v = *(const f32x4*)(xnrow + i);
        xn[i] = v.x; xn[i + 1] = v.y; xn[i + 2] = v.z; xn[i + 3] = v.w;
      }
    }

    // ---------------- attention for this wave's 16 queries: logits and probabilities stay in registers
    i32x4 cf[3];
    {
      // logits as packed signed 16-bit pairs: the integer softmax then runs on v_pk_*_16, two keys per
      // VALU op.  w[2kt + j] = {logit 4kt+2j, logit 4kt+2j+1} of keys 16kt + 4kq + ...
      typedef short s16x2 __attribute__((ext_vector_type(2)));
      typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
      s16x2 w[16];
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) {
        i32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
          acc = mfma16(lds_frag(lds + L::K, ((4 * ks + kq) * S + kt * 16 + qi) << 4), qf[ks], acc);
        float lf[4];
        scale_clamp<4>(acc, a.ml, -128.0f, lf);
        unsigned bi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bi[i] = __float_as_uint(lf[i] + ITA_MAGIC_F);   // low 16 bits = rne(logit), two's complement
#pragma unroll
        for (int j = 0; j < 2; ++j)
          w[2 * kt + j] = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(bi[2 * j + 1], bi[2 * j], 0x05040100u));
      }
      ITA_SSTAMP(3);
      // integer softmax (models/ITA/QAT/ITA_softmax.py:51-61): shift = max - x, num = 256 >> shift,
      // inv = floor(255 * 2^16 / sum), y = (num * inv) >> 16 = (inv >> 8) >> shift
      s16x2 m2 = w[0];
#pragma unroll
      for (int j = 1; j < 16; ++j) m2 = __builtin_elementwise_max(m2, w[j]);
      const int m = max1632_i(max((int)m2.x, (int)m2.y));
      const s16x2 mm = {(short)m, (short)m};
      const s16x2 cap = {15, 15};                  // 256 >> s and inv_hi >> s are both 0 from s = 9 on
      const u16x2 one = {256, 256};
      u16x2 sh[16], sum2 = {0, 0};
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        sh[j] = __builtin_bit_cast(u16x2, __builtin_elementwise_min((s16x2)(mm - w[j]), cap));
        sum2 += one >> sh[j];
      }
      int sum = sum1632_i((int)sum2.x + (int)sum2.y);   // <= 16 * 256 per half: no 16-bit overflow
      sum = max(sum, 1);
      const int inv_hi = ((int)floorf((1.0f / (float)sum) * 16711680.0f)) >> 8;   // <= 255: sum >= 256
      const u16x2 iv = {(unsigned short)inv_hi, (unsigned short)inv_hi};
      i32x4 pf[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int kt = 4 * kb + t;
          const unsigned p01 = __builtin_bit_cast(unsigned, (u16x2)(iv >> sh[2 * kt]));
          const unsigned p23 = __builtin_bit_cast(unsigned, (u16x2)(iv >> sh[2 * kt + 1]));
          pf[kb][t] = (int)(__builtin_amdgcn_perm(p23, p01, 0x06040200u) ^ 0x80808080u);
        }
      ITA_SSTAMP(4);
      // A.V with uint8 probabilities on a signed MFMA: (p - 128) * v summed + 128 * colsum(v)
#pragma unroll
      for (int dg = 0; dg < 3; ++dg) {
        i32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = *(const i32x4*)(cs + (4 * dg + t) * 16 + 4 * kq);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
            acc[t] = mfma16(lds_frag(lds + L::VT, (((kb * 4 + kq) * P) + (4 * dg + t) * 16 + qi) << 4), pf[kb], acc[t]);
        float f[16];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float f4[4];
          scale_clamp<4>(acc[t], a.mc, -128.0f, f4);
#pragma unroll
          for (int i = 0; i < 4; ++i) f[4 * t + i] = f4[i];
        }
        unsigned c4[4];
        round_pack16(f, c4);
        cf[dg] = (i32x4){(int)c4[0], (int)c4[1], (int)c4[2], (int)c4[3]};
      }
    }
    ITA_SSTAMP(5);
    lds_barrier();   // B2: every wave is done with K, V^T and this frame's column sums
    ITA_SSTAMP(6);
    if (tid < P) cs[tid] = 0;   // this parity is next accumulated after the NEXT frame's B2
    if constexpr (TOK != 0) {
      if (nb < a.B) tok_fill();   // before this frame's global stores: one in-order vmcnt for loads and stores
    }

    // ---------------- out_proj + residual + LayerNorm1
    float x1[EC];
#pragma unroll
    for (int eg = 0; eg < NTE / 4; ++eg) {
      float d[16];
      out_group<3, E>(lds + L::WO, l_bo, 4 * eg, cf, a.mo, a.so, qi, kq, d);
#pragma unroll
      for (int j = 0; j < 16; ++j) x1[16 * eg + j] = (FFN || a.fuse_ln) ? xr[16 * eg + j] + d[j] : d[j];
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
      // ---------------- FFN: fc1 + ReLU (hidden layer = four B fragments in registers) -> fc2 -> LayerNorm2
      i32x4 x1f[NK];
#pragma unroll
      for (int c = 0; c < NK; ++c) {
        unsigned p4[4];
        q_pack16(&x1[16 * c], a.f_inv_sx, p4);
        x1f[c] = (i32x4){(int)p4[0], (int)p4[1], (int)p4[2], (int)p4[3]};
      }
      i32x4 hf[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) hf[g] = wx_group<NK, F>(lds + L::W1, l_b1, 4 * g, x1f, a.m1, 0.0f, qi, kq);
      ITA_SSTAMP(8);
#pragma unroll
      for (int eg = 0; eg < NTE / 4; ++eg) {
        float d[16];
        out_group<4, E>(lds + L::W2, l_b2, 4 * eg, hf, a.m2, a.s2, qi, kq, d);
#pragma unroll
        for (int j = 0; j < 16; ++j) yv[16 * eg + j] = x1[16 * eg + j] + d[j];
      }
      layernorm_q16<E>(yv, lnp + 2 * E, lnp + 3 * E, EC * kq);
    } else {
#pragma unroll
      for (int i = 0; i < EC; ++i) yv[i] = x1[i];
    }
    {
      const size_t o = ((size_t)b * S + token) * E + EC * kq;
      if (a.y) {
#pragma unroll
        for (int i = 0; i < EC; i += 4) *(f32x4*)(a.y + o + i) = (f32x4){yv[i], yv[i + 1], yv[i + 2], yv[i + 3]};
      }
      if (a.y_hi) {
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
          *(h8*)(a.y_hi + po) = vh;
          *(h8*)(a.y_lo + po) = vl;
        }
      }
    }
    ITA_SSTAMP(11);
    // ---------------- the next frame's tokens
    if constexpr (TOK != 0) {
      if (nb < a.B) tok_compute(nb, xr);
    } else {
#pragma unroll
      for (int i = 0; i < EC; ++i) xr[i] = xn[i];
    }
  }
#undef ITA_SSTAMP
}
