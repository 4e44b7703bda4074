// ita_int8_kernels.h -- the ITA int8 blocks on gfx950 int8 MFMA.
//
//   ita_mha_kernel<E>  : ITASelfAttention_QAT.forward   (reference models/ITA/QAT/layers.py:101-127)
//                        + optional fused residual add + LayerNorm
//                        (models/ITA_single_layer_upsample_shuffle/QAT/model.py:102-106)
//   ita_ffn_kernel<E>  : ITAFeedForward_QAT.forward     (layers.py:61-75) + residual + LayerNorm
//                        (QAT/model.py:109-113)
//
// One 512-thread workgroup (8 waves, 2 per SIMD) processes one frame at a time and walks the
// batch with a grid stride.  A frame's activations never leave the CU between the input
// quantiser and the output LayerNorm: x_q, Q, K, V^T, the context and the FFN hidden layer are
// chunk-major int8 images in LDS (ita_device.h); logits and softmax probabilities never leave
// registers.
//
// Roofline: int8 MFMA bound; algorithmic work per frame
//   MHA  E=64: 12.58 M MAC = 25.17 MOP     E=128: 18.87 M MAC = 37.75 MOP
//   FFN  E=64:  4.19 M MAC =  8.39 MOP     E=128:  8.39 M MAC = 16.78 MOP
// HBM bytes per frame: 2 * 128*E*4 (f32 in/out); weights (<=100 KB) stay in L2.
#pragma once
#include "ita_device.h"

struct ItaMhaArgs {
  const float* x;   // (B,128,E) f32 block input
  float* y;         // (B,128,E) f32: fuse_ln ? LayerNorm(x + attn(x)) : attn(x)
  const int8_t *wq, *wk, *wv, *wo;     // [P][E] x3, [E][P]   (reference nn.Linear [out][in])
  const int32_t *bq, *bk, *bv, *bo;    // biases in accumulator units
  float inv_sx, mq, mk, mv, ml, mc, mo, so;
  const float *ln_w, *ln_b;
  int B;
  int fuse_ln;
  // optional per-stage taps for parity tests (all may be null)
  int8_t *t_xq, *t_Q, *t_K, *t_V, *t_logits;
  uint8_t* t_probs;
  int8_t *t_ctx, *t_out;
};

struct ItaFfnArgs {
  const float* x;
  float* y;
  const int8_t *w1, *w2;   // [F][E], [E][F]
  const int32_t *b1, *b2;
  float inv_sx, m1, m2, s2;
  const float *ln_w, *ln_b;
  int B;
  int fuse_ln;
  int8_t *t_xq, *t_h, *t_out;
  // optional second form of the output: f16 hi/lo planes [B][128*E] for the split-precision tail
  // GEMM (ita_f16x3_kernels.h); y may then be null
  _Float16 *y_hi, *y_lo;
  int ld_planes;           // row stride (halves) of the planes per frame
};

template <int E>
struct ItaMhaLds {
  static constexpr int S = 128, P = 192;
  static constexpr int XQ = 0;                  // int8 [E/16][128][16]
  static constexpr int Q = XQ + S * E;          // int8 [12][128][16]   (the context reuses it)
  static constexpr int K = Q + S * P;           // int8 [12][128][16]
  static constexpr int VT = K + S * P;          // int8 [8][192][16]    V^T, keys permuted
  static constexpr int COLSUM = VT + P * S;     // int32 [192]
  static constexpr int OUTQ = COLSUM + P * 4;   // int8 [128][E] row-major
  static constexpr int TOTAL = OUTQ + S * E;
};

template <int E>
struct ItaFfnLds {
  static constexpr int S = 128, F = 256;
  static constexpr int XQ = 0;               // int8 [E/16][128][16]
  static constexpr int H = XQ + S * E;       // int8 [16][128][16]
  static constexpr int OUTQ = H + S * F;     // int8 [128][E] row-major
  static constexpr int TOTAL = OUTQ + S * E;
};

__device__ __forceinline__ i32x4 lds_frag(const char* lds, int off) { return *(const i32x4*)(lds + off); }
__device__ __forceinline__ i32x4 gl_frag(const int8_t* __restrict__ p) { return *(const i32x4*)p; }

// ---- phase 0: quantise the float block input into the chunk-major x_q image; keep the float
// values (this thread's E/4 channels of one token) in registers for the residual.
template <int E>
__device__ __forceinline__ void quantize_tokens(const float* __restrict__ xrow, float inv_sx, char* lds_xq,
                                                int token, int qtr, float (&xr)[E / 4], int8_t* tap_row) {
  constexpr int EC = E / 4;
#pragma unroll
  for (int i = 0; i < EC; i += 4) {
    const f32x4 v = *(const f32x4*)(xrow + i);
    xr[i] = v.x; xr[i + 1] = v.y; xr[i + 2] = v.z; xr[i + 3] = v.w;
  }
#pragma unroll
  for (int c = 0; c < EC; c += 16) {
    unsigned p4[4];
    q_pack16(&xr[c], inv_sx, p4);
    const i32x4 pk = {(int)p4[0], (int)p4[1], (int)p4[2], (int)p4[3]};
    *(i32x4*)(lds_xq + cm_off(token, qtr * EC + c, 128)) = pk;
    if (tap_row) *(i32x4*)(tap_row + qtr * EC + c) = pk;
  }
}

// ---- final phase: dequantise the int8 block output, optional residual + LayerNorm, store f32
template <int E>
__device__ __forceinline__ void finish_tokens(const char* lds_outq, float so, const float (&xr)[E / 4], int token,
                                              int qtr, bool fuse_ln, const float* __restrict__ ln_w,
                                              const float* __restrict__ ln_b, float* __restrict__ yrow,
                                              _Float16* __restrict__ hi_row = nullptr,
                                              _Float16* __restrict__ lo_row = nullptr) {
  constexpr int EC = E / 4;
  float r[EC];
#pragma unroll
  for (int c = 0; c < EC; c += 16) {
    const i32x4 pk = *(const i32x4*)(lds_outq + token * E + qtr * EC + c);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int v = (int)(int8_t)((unsigned)pk[j >> 2] >> (8 * (j & 3)));
      r[c + j] = (float)v * so;
    }
  }
  if (fuse_ln) {
#pragma unroll
    for (int i = 0; i < EC; ++i) r[i] = xr[i] + r[i];
    layernorm_lanes<E, 4>(r, ln_w, ln_b, qtr * EC);
  }
  if (yrow) {
#pragma unroll
    for (int i = 0; i < EC; i += 4) {
      f32x4 v = {r[i], r[i + 1], r[i + 2], r[i + 3]};
      *(f32x4*)(yrow + i) = v;
    }
  }
  if (hi_row) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#pragma unroll
    for (int i = 0; i < EC; i += 8) {
      h8 vh, vl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const _Float16 hh = (_Float16)r[i + j];
        vh[j] = hh;
        vl[j] = (_Float16)(r[i + j] - (float)hh);
      }
      *(h8*)(hi_row + i) = vh;
      *(h8*)(lo_row + i) = vl;
    }
  }
}

// 32x32 output tile, weights as the A operand (rows = output features) read from global memory,
// activations as the Bt operand (cols = tokens) read from a chunk-major LDS image.
//   acc[i] <-> feature f0 + (i&3) + 8*(i>>2) + 4*h,  token t0 + (lane&31)
template <int KB>   // K in bytes
__device__ __forceinline__ i32x16 tile_w_x(const int8_t* __restrict__ w, int f0, const int32_t* __restrict__ bias,
                                           const char* lds_act, int t0, int lane) {
  const int r = lane & 31, h = lane >> 5;
  i32x16 acc;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const i32x4 b4 = *(const i32x4*)(bias + f0 + 8 * g + 4 * h);
    acc[4 * g] = b4.x; acc[4 * g + 1] = b4.y; acc[4 * g + 2] = b4.z; acc[4 * g + 3] = b4.w;
  }
  const int8_t* wrow = w + (size_t)(f0 + r) * KB + 16 * h;
#pragma unroll
  for (int ks = 0; ks < KB / 32; ++ks) {
    const i32x4 a = gl_frag(wrow + 32 * ks);
    const i32x4 b = lds_frag(lds_act, cm_off(t0 + r, 32 * ks + 16 * h, 128));
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
  }
  return acc;
}

// requantise a 32x32 accumulator tile (features on rows) and store it
//   chunk-major into `dst` (rows = tokens, K = features)   [CM = true]
//   or row-major [token][LD]                               [CM = false]
template <bool CM>
__device__ __forceinline__ void store_tile_fx(const i32x16& acc, float mult, float lo, char* dst, int ld, int f0,
                                              int t0, int lane, int8_t* tap, int tap_ld) {
  const int r = lane & 31, h = lane >> 5;
  unsigned p4[4];
  rq_pack16(acc, mult, lo, p4);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const unsigned pk = p4[g];
    const int f = f0 + 8 * g + 4 * h;
    if constexpr (CM) *(unsigned*)(dst + cm_off(t0 + r, f, 128)) = pk;
    else *(unsigned*)(dst + (t0 + r) * ld + f) = pk;
    if (tap) *(unsigned*)(tap + (size_t)(t0 + r) * tap_ld + f) = pk;
  }
}

// ======================================================================================
template <int E>
__global__ __launch_bounds__(512) void ita_mha_kernel(const ItaMhaArgs a) {
  using L = ItaMhaLds<E>;
  constexpr int S = 128, P = 192, EC = E / 4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int token = tid >> 2, qtr = tid & 3;
  int* colsum = (int*)(lds + L::COLSUM);

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    float xr[EC];
    // ---------------- phase 0: quantise
    quantize_tokens<E>(a.x + ((size_t)b * S + token) * E + qtr * EC, a.inv_sx, lds + L::XQ, token, qtr, xr,
                       a.t_xq ? a.t_xq + ((size_t)b * S + token) * E : nullptr);
    if (tid < P) colsum[tid] = 0;
    __syncthreads();

    // ---------------- phase P: Q, K, V projections (18 feature tiles x 4 token tiles)
    {
      const int tt = wave & 3, half = wave >> 2;
      const int r = lane & 31, h = lane >> 5;
      for (int ft = 9 * half; ft < 9 * half + 9; ++ft) {
        const int mat = ft / 6, dt = ft - 6 * mat;
        if (mat < 2) {   // Q, K stored [token][d] (d contiguous): weights are the A operand
          const i32x16 acc = tile_w_x<E>(mat == 0 ? a.wq : a.wk, dt * 32, mat == 0 ? a.bq : a.bk, lds + L::XQ,
                                         tt * 32, lane);
          int8_t* tap = mat == 0 ? a.t_Q : a.t_K;
          store_tile_fx<true>(acc, mat == 0 ? a.mq : a.mk, -128.0f, lds + (mat == 0 ? L::Q : L::K), 0, dt * 32,
                              tt * 32, lane, tap ? tap + (size_t)b * S * P : nullptr, P);
        } else {         // V stored transposed [d][key]: activations are the A operand
          const int d = dt * 32 + r;
          const int bias = a.bv[d];
          i32x16 acc;
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[i] = bias;
          const int8_t* wrow = a.wv + (size_t)d * E + 16 * h;
#pragma unroll
          for (int ks = 0; ks < E / 32; ++ks) {
            const i32x4 xa = lds_frag(lds + L::XQ, cm_off(tt * 32 + r, 32 * ks + 16 * h, 128));
            const i32x4 wb = gl_frag(wrow + 32 * ks);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa, wb, acc, 0, 0, 0);
          }
          // acc[4g+i] <-> key tt*32 + 8g + 4h + i, feature d.  Keys are stored permuted inside
          // each 64-key block so that the 16x16x64 AV MFMA reads 16 contiguous bytes per lane
          // in the slot order the softmax leaves its probabilities in:
          //   key = kb*64 + 16*t + 4*kq + i   ->   byte kb*64 + 16*kq + 4*t + i
          int csum = 0;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const unsigned b0 = rq_bits(acc[4 * g], a.mv), b1 = rq_bits(acc[4 * g + 1], a.mv),
                           b2 = rq_bits(acc[4 * g + 2], a.mv), b3 = rq_bits(acc[4 * g + 3], a.mv);
            csum += bits_to_int(b0) + bits_to_int(b1) + bits_to_int(b2) + bits_to_int(b3);
            const int kb = tt >> 1, kq = 2 * (g & 1) + h, t = 2 * (tt & 1) + (g >> 1);
            *(unsigned*)(lds + L::VT + (((kb * 4 + kq) * P + d) << 4) + 4 * t) = pack4(b0, b1, b2, b3);
            if (a.t_V) {
              int8_t* tv = a.t_V + ((size_t)b * S + tt * 32 + 8 * g + 4 * h) * P + d;
              tv[0] = (int8_t)b0; tv[P] = (int8_t)b1; tv[2 * P] = (int8_t)b2; tv[3 * P] = (int8_t)b3;
            }
          }
          csum += __shfl_xor(csum, 32);
          if (h == 0) atomicAdd(&colsum[d], csum);
        }
      }
    }
    __syncthreads();

    // ---------------- phase A: per wave 16 queries: QK^T -> integer softmax -> A.V, in registers
    {
      const int q0 = wave * 16, qi = lane & 15, kq = lane >> 4;
      i32x4 qf[3];
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) qf[ks] = lds_frag(lds + L::Q, cm_off(q0 + qi, 64 * ks + 16 * kq, 128));
      // S^T tiles: rows = keys, cols = queries: v[4*kt + i] <-> key kt*16 + 4*kq + i, query q0 + qi
      int v[32];
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) {
        i32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const i32x4 kf = lds_frag(lds + L::K, cm_off(kt * 16 + qi, 64 * ks + 16 * kq, 128));
          acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(kf, qf[ks], acc, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) v[4 * kt + i] = bits_to_int(rq_bits(acc[i], a.ml));
      }
      if (a.t_logits) {
        int8_t* tl = a.t_logits + ((size_t)b * S + q0 + qi) * S + 4 * kq;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
          *(unsigned*)(tl + 16 * kt) = pack4(v[4 * kt], v[4 * kt + 1], v[4 * kt + 2], v[4 * kt + 3]);
      }
      // integer softmax (models/ITA/QAT/ITA_softmax.py:51-61): shift = max - x,
      // num = 256 >> shift, inv = floor(255*2^16 / sum), y = (num * inv) >> 16 = (inv >> 8) >> shift
      int m = v[0];
#pragma unroll
      for (int j = 1; j < 32; ++j) m = max(m, v[j]);
      m = max(m, __shfl_xor(m, 16));
      m = max(m, __shfl_xor(m, 32));
      int sum = 0;
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        v[j] = min(m - v[j], 23);
        sum += 256 >> v[j];
      }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      sum = max(sum, 1);
      const int inv = (int)floorf((1.0f / (float)sum) * 16711680.0f);
      const int inv_hi = inv >> 8;
      unsigned pk[8];
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
        pk[kt] = pack4(inv_hi >> v[4 * kt], inv_hi >> v[4 * kt + 1], inv_hi >> v[4 * kt + 2],
                       inv_hi >> v[4 * kt + 3]);
      if (a.t_probs) {
        uint8_t* tp = a.t_probs + ((size_t)b * S + q0 + qi) * S + 4 * kq;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) *(unsigned*)(tp + 16 * kt) = pk[kt];
      }
      // A.V with uint8 probabilities on a signed MFMA: (p - 128) * v summed + 128 * colsum(v)
      i32x4 pf[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int t = 0; t < 4; ++t) pf[kb][t] = (int)(pk[4 * kb + t] ^ 0x80808080u);
#pragma unroll
      for (int dt = 0; dt < 12; ++dt) {
        i32x4 acc = *(const i32x4*)(colsum + dt * 16 + 4 * kq);
        acc = acc << 7;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const i32x4 vf = lds_frag(lds + L::VT, ((((kb * 4 + kq) * P) + dt * 16 + qi) << 4));
          acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(vf, pf[kb], acc, 0, 0, 0);
        }
        // acc[i] <-> feature dt*16 + 4*kq + i, query q0 + qi; the context overwrites this wave's own Q rows
        const unsigned c4 = pack4(rq_bits(acc[0], a.mc), rq_bits(acc[1], a.mc), rq_bits(acc[2], a.mc),
                                  rq_bits(acc[3], a.mc));
        *(unsigned*)(lds + L::Q + cm_off(q0 + qi, dt * 16 + 4 * kq, 128)) = c4;
        if (a.t_ctx) *(unsigned*)(a.t_ctx + ((size_t)b * S + q0 + qi) * P + dt * 16 + 4 * kq) = c4;
      }
    }
    __syncthreads();

    // ---------------- phase O: output projection (E/32 feature tiles x 4 token tiles)
    for (int tile = wave; tile < (E / 32) * 4; tile += 8) {
      const int et = tile >> 2, tt = tile & 3;
      const i32x16 acc = tile_w_x<P>(a.wo, et * 32, a.bo, lds + L::Q, tt * 32, lane);
      store_tile_fx<false>(acc, a.mo, -128.0f, lds + L::OUTQ, E, et * 32, tt * 32, lane,
                           a.t_out ? a.t_out + (size_t)b * S * E : nullptr, E);
    }
    __syncthreads();

    // ---------------- phase L: dequantise (+ residual + LayerNorm) and store
    finish_tokens<E>(lds + L::OUTQ, a.so, xr, token, qtr, a.fuse_ln != 0, a.ln_w, a.ln_b,
                     a.y + ((size_t)b * S + token) * E + qtr * EC);
  }
}

// ======================================================================================
template <int E>
__global__ __launch_bounds__(512) void ita_ffn_kernel(const ItaFfnArgs a) {
  using L = ItaFfnLds<E>;
  constexpr int S = 128, F = 256, EC = E / 4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int token = tid >> 2, qtr = tid & 3;

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    float xr[EC];
    quantize_tokens<E>(a.x + ((size_t)b * S + token) * E + qtr * EC, a.inv_sx, lds + L::XQ, token, qtr, xr,
                       a.t_xq ? a.t_xq + ((size_t)b * S + token) * E : nullptr);
    __syncthreads();
    // fc1 + ReLU: 8 feature tiles x 4 token tiles; ReLU on the int8 code = clamp to [0,127]
    {
      const int tt = wave & 3;
      for (int ft = (wave >> 2) * 4; ft < (wave >> 2) * 4 + 4; ++ft) {
        const i32x16 acc = tile_w_x<E>(a.w1, ft * 32, a.b1, lds + L::XQ, tt * 32, lane);
        store_tile_fx<true>(acc, a.m1, 0.0f, lds + L::H, 0, ft * 32, tt * 32, lane,
                            a.t_h ? a.t_h + (size_t)b * S * F : nullptr, F);
      }
    }
    __syncthreads();
    for (int tile = wave; tile < (E / 32) * 4; tile += 8) {
      const int et = tile >> 2, tt = tile & 3;
      const i32x16 acc = tile_w_x<F>(a.w2, et * 32, a.b2, lds + L::H, tt * 32, lane);
      store_tile_fx<false>(acc, a.m2, -128.0f, lds + L::OUTQ, E, et * 32, tt * 32, lane,
                           a.t_out ? a.t_out + (size_t)b * S * E : nullptr, E);
    }
    __syncthreads();
    const size_t o = ((size_t)b * S + token) * E + qtr * EC;
    const size_t po = (size_t)b * a.ld_planes + token * E + qtr * EC;
    finish_tokens<E>(lds + L::OUTQ, a.s2, xr, token, qtr, a.fuse_ln != 0, a.ln_w, a.ln_b, a.y ? a.y + o : nullptr,
                     a.y_hi ? a.y_hi + po : nullptr, a.y_lo ? a.y_lo + po : nullptr);
  }
}
