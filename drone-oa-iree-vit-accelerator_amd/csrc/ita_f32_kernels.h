// ita_f32_kernels.h -- the float32 layers around the int8 blocks.
//
// Every kernel here uses the oracle's operation order (oracle/ita_oracle.c): one ascending-k
// fmaf chain per output started from the bias, LayerNorm sums in four blocks.  The translation
// unit is compiled with -ffp-contract=off, so the only fused multiply-adds are the explicit
// fmaf() calls and the f32 MFMA, which is itself a k-ordered fmaf chain on gfx950.
//
//   ita_tokenizer_kernel   OverlapPatchMerging: conv7x7 s2 p3 + bilinear(30x45->8x16) + LayerNorm
//                          (reference models/ITA/QAT/layers.py:39-45)
//   ita_tail_kernel        PixelShuffle(2) || Upsample(16,32,align_corners) -> cat -> conv3x3 -> 9x16x32
//                          (reference models/ITA_single_layer_upsample_shuffle/QAT/model.py:116-121)
//   ita_gemm_f32_kernel    y = x W^T + b on v_mfma_f32_16x16x4_f32 (decoder Linear 4608->512,
//                          LSTM gate pre-activations)                      (QAT/model.py:124,128)
//   ita_lstm_prep_kernel / ita_lstm_point_kernel / ita_fc_kernel   concat, LSTM cell, fc 128->3
//                          (QAT/model.py:126-130)
#pragma once
#include "ita_device.h"

typedef float f32x16v __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------ tokenizer
struct ItaTokArgs {
  const void* img;    // (B,60,90) f32, or u8 wire frames
  const float* cw;    // [50][E]: conv weights k-major, row 49 zero (prepared by ita_load_weights)
  const float* cb;    // [E]
  const float *ln_w, *ln_b;
  float* tokens;      // (B,128,E)
  int B;
  int dbg;            // diagnostic bit mask: 1 skip image fill, 2 skip blend, 4 skip MFMA, 8 skip LN/store
};

// Layout of one workgroup (256 threads = 4 waves, wave w owns tokens 32w..32w+31):
//   img  [66][96] f32   frame with zero border                                  (all waves)
//   wt   [50][E]  f32   conv weights k-major, row 49 = 0                         (all waves)
//   per wave: pb [32][51] blended 7x7 patches (col 49 = 0), later overwritten by the
//             32 x E pre-LayerNorm tokens of the same wave.
// The 49-long dot products run on v_mfma_f32_32x32x2_f32, which on gfx950 is exactly the
// chain  acc = fma(a_k1, b_k1, fma(a_k0, b_k0, acc))  in ascending k -- the oracle's order.
template <int E>
struct ItaTokLds {
  static constexpr int PH = 66, PW = 96, PBS = 51;
  static constexpr int IMG = 0;
  static constexpr int WT = IMG + PH * PW * 4;
  static constexpr int WAVE0 = WT + 50 * E * 4;
  static constexpr int WAVE_BYTES = 32 * E * 4 > 32 * PBS * 4 ? 32 * E * 4 : 32 * PBS * 4;
  static constexpr int TOTAL = WAVE0 + 4 * WAVE_BYTES;
};

template <int E, bool U8>
__global__ __launch_bounds__(256) void ita_tokenizer_kernel(const ItaTokArgs a) {
  using L = ItaTokLds<E>;
  constexpr int PW = L::PW, PBS = L::PBS, EC = E / 2;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* img = (float*)(lds + L::IMG);
  float* wt = (float*)(lds + L::WT);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* pb = (float*)(lds + L::WAVE0 + wave * L::WAVE_BYTES);   // [32][PBS], then [32][E]
  {
    constexpr int NW = (50 * E / 4 + 255) / 256;
    f32x4 w4[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int p = tid + 256 * j;
      w4[j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
      if (p < 50 * E / 4) w4[j] = *(const f32x4*)(a.cw + 4 * p);
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int p = tid + 256 * j;
      if (p < 50 * E / 4) *(f32x4*)(wt + 4 * p) = w4[j];
    }
  }
  // this thread's blend job: token (lane>>1) of this wave, taps [25*(lane&1), ...)
  const int tl = lane >> 1, tg = wave * 32 + tl;
  int y0, yp, x0, xp;
  float ly, lx;
  bilinear_src_dev(tg >> 4, 30.0f / 8.0f, 30, y0, yp, ly);
  bilinear_src_dev(tg & 15, 45.0f / 16.0f, 45, x0, xp, lx);
  // u8 wire frames: the pixel CODES are blended with the integer weights H = 8h, W = 32w (the weights of this fixed
  // resize are dyadic), every product and sum an exact integer <= 255 * 256 in f32, and 1 / 65280 is in the conv
  // weights (a.cw points at the scaled copy) -- oracle/ita_oracle.c ita_oracle_tokenizer_u8
  const float h1 = U8 ? 8.0f * ly : ly, h0 = (U8 ? 8.0f : 1.0f) - h1, w1 = U8 ? 32.0f * lx : lx, w0 = (U8 ? 32.0f : 1.0f) - w1;
  const int r = lane & 31, kk = lane >> 5;
  // per-thread constants kept in registers for the whole launch: this thread's LayerNorm affine
  // parameters (loading them per frame exposed one L2 latency per frame) ...
  float lnw[EC], lnb[EC];
#pragma unroll
  for (int c = 0; c < EC; c += 4) {
    const f32x4 w4 = *(const f32x4*)(a.ln_w + (lane & 1) * EC + c), b4 = *(const f32x4*)(a.ln_b + (lane & 1) * EC + c);
    lnw[c] = w4.x; lnw[c + 1] = w4.y; lnw[c + 2] = w4.z; lnw[c + 3] = w4.w;
    lnb[c] = b4.x; lnb[c + 1] = b4.y; lnb[c + 2] = b4.z; lnb[c + 3] = b4.w;
  }

  for (int i = tid; i < L::PH * PW; i += 256) img[i] = 0.0f;   // the zero border is written once

  // A frame is fetched with wide loads, all issued before the first LDS store, one frame ahead of
  // the frame being computed (a load per loop iteration would expose one memory latency each).
  constexpr int NV = U8 ? (5400 / 16 + 255) / 256 : (1350 + 255) / 256;
  i32x4 fv[NV];
  auto fetch = [&](int b) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int p = tid + 256 * j;
      fv[j] = (i32x4){0, 0, 0, 0};
      if constexpr (U8) {
        const uint8_t* src = (const uint8_t*)a.img + (size_t)b * 5400;
        if (p < 337) fv[j] = *(const i32x4*)(src + 16 * p);
        else if (p == 337) { fv[j].x = *(const int*)(src + 5392); fv[j].y = *(const int*)(src + 5396); }
      } else {
        const float* src = (const float*)a.img + (size_t)b * 5400;
        if (p < 1350) fv[j] = *(const i32x4*)(src + 4 * p);
      }
    }
  };
  if ((int)blockIdx.x < a.B) fetch(blockIdx.x);

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    __syncthreads();   // previous frame's img fully consumed
    if (!(a.dbg & 1))
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int p = tid + 256 * j;
      if constexpr (U8) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int idx = 16 * p + e;
          if (idx < 5400) {
            const int y = idx / 90, x = idx - 90 * y;
            img[(y + 3) * PW + x + 3] = (float)(((unsigned)fv[j][e >> 2] >> (8 * (e & 3))) & 0xffu);
          }
        }
      } else {
        if (p < 1350) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int idx = 4 * p + e, y = idx / 90, x = idx - 90 * y;
            img[(y + 3) * PW + x + 3] = __int_as_float(fv[j][e]);
          }
        }
      }
    }
    __syncthreads();
    if (b + (int)gridDim.x < a.B) fetch(b + gridDim.x);
    // blended patches (both the conv and the bilinear resize are linear): even lanes take kernel
    // rows 0..3 of their token, odd lanes rows 4..6 (+ the zero pad column)
    if (!(a.dbg & 2)) {
      const int odd = lane & 1;
      const int kyb = odd ? 4 : 0, kye = odd ? 7 : 4;
      const float* p00 = img + (2 * y0) * PW + 2 * x0;   // (-3 conv padding) + (3 border) = 0
      const int dx = 2 * xp, dy = 2 * yp * PW;
      for (int ky = kyb; ky < kye; ++ky) {
        const float* q = p00 + ky * PW;
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
          const float va = q[kx], vb = q[kx + dx], vc = q[kx + dy], vd = q[kx + dy + dx];
          pb[tl * PBS + ky * 7 + kx] = h0 * (w0 * va + w1 * vb) + h1 * (w0 * vc + w1 * vd);
        }
      }
      if (odd) pb[tl * PBS + 49] = 0.0f;
    }
    // (pb rows are private to this wave: wave-synchronous, no workgroup barrier needed)
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    // tokens[32 x E] = bias + pb[32 x 50] . wt[50 x E]
    f32x16v acc[E / 32];
#pragma unroll
    for (int ct = 0; ct < E / 32; ++ct) {
      const float bv = a.cb[ct * 32 + r];
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[ct][e] = bv;
    }
    if (!(a.dbg & 4))
#pragma unroll 5
    for (int s = 0; s < 25; ++s) {
      const float av = pb[r * PBS + 2 * s + kk];
#pragma unroll
      for (int ct = 0; ct < E / 32; ++ct)
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wt[(2 * s + kk) * E + ct * 32 + r], acc[ct], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
    // C layout: col = channel ct*32 + r, row = token (e&3) + 8*(e>>2) + 4*kk  -> pre[token][E] over pb
#pragma unroll
    for (int ct = 0; ct < E / 32; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) pb[((e & 3) + 8 * (e >> 2) + 4 * kk) * E + ct * 32 + r] = acc[ct][e];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (!(a.dbg & 8)) {
      const int half = lane & 1;
      float v[EC];
#pragma unroll
      for (int c = 0; c < EC; c += 4) {
        const f32x4 q = *(const f32x4*)(pb + tl * E + half * EC + c);
        v[c] = q.x; v[c + 1] = q.y; v[c + 2] = q.z; v[c + 3] = q.w;
      }
      layernorm_lanes<E, 2>(v, lnw, lnb, 0);
      float* out = a.tokens + ((size_t)b * 128 + tg) * E + half * EC;
#pragma unroll
      for (int c = 0; c < EC; c += 4) *(f32x4*)(out + c) = (f32x4){v[c], v[c + 1], v[c + 2], v[c + 3]};
    }
    __builtin_amdgcn_wave_barrier();
  }
}
template <int E>
constexpr int ita_tok_lds_bytes() { return ItaTokLds<E>::TOTAL; }

// ------------------------------------------------------------------ fusion tail
struct ItaTailArgs {
  const float* x2;     // (B,128,E) tokens after the encoder
  const float* wT;     // [5E/4 * 9][12]: conv3x3 weights re-laid [c][ky][kx][o (9, padded to 12)]
  const float* cb;     // [9]
  float* feat;         // (B, 9*16*32) with row stride ld_feat
  int ld_feat;
  int B;
};

template <int E>
__global__ __launch_bounds__(256) void ita_tail_kernel(const ItaTailArgs a) {
  constexpr int XS = E + 1;              // token row stride (floats) in LDS
  constexpr int FH = 18, FW = 34;        // 16x32 map with a zero border
  constexpr int CH = 16;                 // channels per staged chunk
  constexpr int NCHUNK = (E / 4 + E) / CH;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* xs = (float*)lds;               // [128][XS]
  float* fu = xs + 128 * XS;             // [CH][FH][FW]
  const int tid = threadIdx.x;
  for (int i = tid; i < CH * FH * FW; i += 256) fu[i] = 0.0f;   // borders stay zero
  const int py = tid >> 5, px = tid & 31;                        // outputs (py, px) and (py + 8, px)
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    __syncthreads();
    for (int i = tid; i < 128 * E; i += 256) xs[(i / E) * XS + (i % E)] = a.x2[(size_t)b * 128 * E + i];
    float acc0[9], acc1[9];
#pragma unroll
    for (int o = 0; o < 9; ++o) { acc0[o] = a.cb[o]; acc1[o] = a.cb[o]; }
    for (int ch = 0; ch < NCHUNK; ++ch) {
      __syncthreads();
      for (int i = tid; i < CH * 512; i += 256) {
        const int c = i >> 9, y = (i >> 5) & 15, x = i & 31;
        float v;
        if (ch * CH + c < E / 4) {           // PixelShuffle(2): out[c][2h+i][2w+j] = in[4c+2i+j][h][w]
          const int cc = ch * CH + c;
          v = xs[((y >> 1) * 16 + (x >> 1)) * XS + 4 * cc + 2 * (y & 1) + (x & 1)];
        } else {                             // bilinear x2, align_corners=True
          const int cc = ch * CH + c - E / 4;
          const float sy = (7.0f / 15.0f) * (float)y, sx = (15.0f / 31.0f) * (float)x;
          int y0 = (int)sy, x0 = (int)sx;
          if (y0 > 7) y0 = 7;
          if (x0 > 15) x0 = 15;
          const int yp = y0 < 7 ? 1 : 0, xp = x0 < 15 ? 1 : 0;
          const float h1 = sy - (float)y0, h0 = 1.0f - h1, w1 = sx - (float)x0, w0 = 1.0f - w1;
          const float v00 = xs[(y0 * 16 + x0) * XS + cc], v01 = xs[(y0 * 16 + x0 + xp) * XS + cc];
          const float v10 = xs[((y0 + yp) * 16 + x0) * XS + cc], v11 = xs[((y0 + yp) * 16 + x0 + xp) * XS + cc];
          v = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11);
        }
        fu[(c * FH + y + 1) * FW + x + 1] = v;
      }
      __syncthreads();
      for (int c = 0; c < CH; ++c) {
        const float* f = fu + c * FH * FW;
        const float* w = a.wT + (size_t)((ch * CH + c) * 9) * 12;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float v0 = f[(py + ky) * FW + px + kx], v1 = f[(py + 8 + ky) * FW + px + kx];
            const float* wk = w + (ky * 3 + kx) * 12;
#pragma unroll
            for (int o = 0; o < 9; ++o) {
              acc0[o] = fmaf(v0, wk[o], acc0[o]);
              acc1[o] = fmaf(v1, wk[o], acc1[o]);
            }
          }
      }
    }
    float* out = a.feat + (size_t)b * a.ld_feat;
#pragma unroll
    for (int o = 0; o < 9; ++o) {
      out[o * 512 + py * 32 + px] = acc0[o];
      out[o * 512 + (py + 8) * 32 + px] = acc1[o];
    }
  }
}
template <int E>
constexpr int ita_tail_lds_bytes() { return (128 * (E + 1) + 16 * 18 * 34) * 4; }

// ------------------------------------------------------------------ f32 GEMM (NT) on f32 MFMA
// C[m][n] = bias[n] + sum_k A[m][k] * W[n][k], k ascending in one chain per output.
// Workgroup = 4 waves as 2(M) x 2(N); wave tile 16(M) x 32(N) = two 16x16x4 accumulators;
// workgroup tile 32 x 64; K staged through LDS 32 columns at a time.  N % 64 == 0, K % 32 == 0.
struct ItaGemmArgs {
  const float* A; int lda;
  const float* W; int ldw;
  const float* bias;
  float* C; int ldc;
  int M, N, K;
};

__global__ __launch_bounds__(256) void ita_gemm_f32_kernel(const ItaGemmArgs g) {
  constexpr int LD = 36;
  __shared__ __attribute__((aligned(16))) float As[32 * LD];
  __shared__ __attribute__((aligned(16))) float Ws[64 * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 64;
  const int r = lane & 15, kq = lane >> 4;
  // loader mapping: one float4 of A, two of W per thread per K-tile
  const int lrow = tid >> 3, lk = (tid & 7) * 4;
  const int arow = min(m0 + lrow, g.M - 1);
  const float* ap = g.A + (size_t)arow * g.lda + lk;
  const float* wp0 = g.W + (size_t)(n0 + lrow) * g.ldw + lk;
  const float* wp1 = g.W + (size_t)(n0 + 32 + lrow) * g.ldw + lk;
  f32x4 acc0, acc1;
  {
    const float b0 = g.bias ? g.bias[n0 + wn * 32 + r] : 0.0f;
    const float b1 = g.bias ? g.bias[n0 + wn * 32 + 16 + r] : 0.0f;
    acc0 = (f32x4){b0, b0, b0, b0};
    acc1 = (f32x4){b1, b1, b1, b1};
  }
  f32x4 ra = *(const f32x4*)ap, rw0 = *(const f32x4*)wp0, rw1 = *(const f32x4*)wp1;
  for (int k0 = 0; k0 < g.K; k0 += 32) {
    __syncthreads();
    *(f32x4*)(As + lrow * LD + lk) = ra;
    *(f32x4*)(Ws + lrow * LD + lk) = rw0;
    *(f32x4*)(Ws + (32 + lrow) * LD + lk) = rw1;
    __syncthreads();
    if (k0 + 32 < g.K) {
      ra = *(const f32x4*)(ap + k0 + 32);
      rw0 = *(const f32x4*)(wp0 + k0 + 32);
      rw1 = *(const f32x4*)(wp1 + k0 + 32);
    }
    const float* as = As + (wm * 16 + r) * LD + kq;
    const float* ws0 = Ws + (wn * 32 + r) * LD + kq;
    const float* ws1 = Ws + (wn * 32 + 16 + r) * LD + kq;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float av = as[4 * s];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, ws0[4 * s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, ws1[4 * s], acc1, 0, 0, 0);
    }
  }
  // C layout: col = lane&15, row = 4*(lane>>4) + i
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 16 + 4 * kq + i;
    if (m < g.M) {
      g.C[(size_t)m * g.ldc + n0 + wn * 32 + r] = acc0[i];
      g.C[(size_t)m * g.ldc + n0 + wn * 32 + 16 + r] = acc1[i];
    }
  }
}

// ------------------------------------------------------------------ LSTM glue
// cat0[b] = [dec(512, written by the decoder GEMM) | desvel/10 | quat(4) | h_in[0][b](128) | 0-pad] (K0P wide)
// cat1[b] = [h_out[0][b] (written by layer 0) | h_in[1][b]],  cat2[b] likewise
struct ItaLstmPrepArgs {
  const float *desvel, *quat, *h_in;   // (B), (B,4), (3,B,128)
  float *cat0, *cat1, *cat2;
  int k0p;                             // padded width of cat0 (672)
  int B;
};
__global__ void ita_lstm_prep_kernel(const ItaLstmPrepArgs a) {
  const int b = blockIdx.x, t = threadIdx.x;   // 256 threads
  if (b >= a.B) return;
  float* c0 = a.cat0 + (size_t)b * a.k0p;
  if (t == 0) c0[512] = a.desvel[b] / 10.0f;
  if (t >= 1 && t < 5) c0[512 + t] = a.quat[(size_t)b * 4 + t - 1];
  if (t < 128) {
    c0[517 + t] = a.h_in[(size_t)b * 128 + t];
    a.cat1[(size_t)b * 256 + 128 + t] = a.h_in[((size_t)a.B + b) * 128 + t];
    a.cat2[(size_t)b * 256 + 128 + t] = a.h_in[((size_t)2 * a.B + b) * 128 + t];
  }
  if (t >= 128 && 517 + t < a.k0p) c0[517 + t] = 0.0f;
}

struct ItaLstmPointArgs {
  const float* gates;   // (B,512) i,f,g,o pre-activations (biases included)
  const float* c_in;    // (B,128) this layer
  float *h_out, *c_out; // (B,128) this layer
  float* next_in;       // next layer's concat buffer (row stride next_ld) or null
  int next_ld;
  int B;
};
__global__ void ita_lstm_point_kernel(const ItaLstmPointArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.B * 128) return;
  const int b = idx >> 7, j = idx & 127;
  const float* g = a.gates + (size_t)b * 512;
  const float ig = ita_sigmoid(g[j]), fg = ita_sigmoid(g[128 + j]), gg = ita_tanh(g[256 + j]),
              og = ita_sigmoid(g[384 + j]);
  const float c = fmaf(fg, a.c_in[idx], ig * gg);
  const float h = og * ita_tanh(c);
  a.c_out[idx] = c;
  a.h_out[idx] = h;
  if (a.next_in) a.next_in[(size_t)b * a.next_ld + j] = h;
}

// fc 128 -> 3
__global__ void ita_fc_kernel(const float* __restrict__ h, const float* __restrict__ w, const float* __restrict__ bias,
                              float* __restrict__ vel, int B, const int* __restrict__ slots = nullptr) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * 3) return;
  const int b = idx / 3, o = idx - 3 * b;
  const size_t hb = slots ? (size_t)slots[b] : (size_t)b;
  float acc = bias[o];
  for (int k = 0; k < 128; ++k) acc = fmaf(h[hb * 128 + k], w[o * 128 + k], acc);
  vel[idx] = acc;
}
