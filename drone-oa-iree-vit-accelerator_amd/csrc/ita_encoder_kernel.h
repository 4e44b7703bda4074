// ita_encoder_kernel.h -- one whole encoder layer per launch (E = 64, the ITAViTLSTM shape):
//
//   x1 = LayerNorm1(x + ITASelfAttention_QAT(x));  y = LayerNorm2(x1 + ITAFeedForward_QAT(x1))
//   (reference models/ITA_single_layer_upsample_shuffle/QAT/model.py:100-113,
//    models/ITA/QAT/layers.py:61-75,101-127, models/ITA/QAT/ITA_softmax.py:51-61)
//
// Same arithmetic, bit for bit, as ita_mha_kernel + ita_ffn_kernel (ita_int8_kernels.h); what
// changes is where the operands live, because those two kernels are bound by the latency of
// their global weight loads, not by MFMA or VALU issue:
//   * persistent workgroups (one per CU) keep Wq/Wk/Wv/W1/W2 and all biases in LDS as chunk-major
//     int8 images for the whole launch; every wave always computes the same out_proj output tile,
//     so its Wo fragments stay in registers (24 VGPRs);
//   * x1 never leaves the CU: LayerNorm1's result stays in the registers of the thread that
//     quantises it for the FFN;
//   * the next frame's tokens are fetched into registers while the current frame is in the
//     attention phase.
// LDS: x_q/out_q 8 KB | Q(ctx) 24 KB | K 24 KB | V^T 24 KB (K..V^T reused for the 32 KB FFN hidden
// layer) | colsum | Wq Wk Wv 36 KB | W1 16 KB | W2 16 KB | biases 3.75 KB | LN 1 KB = 157 184 B of the
// CU's 160 KB.
#pragma once
#include "ita_int8_kernels.h"

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct ItaEncArgs {
  const float* x;        // (B,128,64)
  float* y;              // (B,128,64) f32, may be null when planes are given
  _Float16 *y_hi, *y_lo; // optional f16 hi/lo planes of y, row stride ld_planes (>= 8192) per frame
  int ld_planes;
  float* x1_tap;         // optional (B,128,64): LayerNorm1 output
  const int8_t *wq, *wk, *wv, *wo, *w1, *w2;
  const int32_t *bq, *bk, *bv, *bo, *b1, *b2;
  float inv_sx, mq, mk, mv, ml, mc, mo, so;      // attention scalars (ita_weights.h)
  float f_inv_sx, m1, m2, s2;                    // FFN scalars
  const float *n1w, *n1b, *n2w, *n2b;
  int B;
  // diagnostic only (null in production): wave 0 of each workgroup stores s_memtime at the phase
  // boundaries of its first 8 frames: stamps[(block * 8 + frame) * 16 + slot] (ita_mi355x.h)
  unsigned long long* stamps;
  // optional side copy for the LSTM that follows: layer-0 hidden state of frame b (row slots[b] or b of
  // h0_src) -> h0_dst[b].  Layer 0 reads whole rows of h while other workgroups overwrite parts of the
  // same row when the state is updated in place; the staged copy removes that race.
  const float* h0_src;
  float* h0_dst;
  const int* slots;
  // fused tokenizer (ita_encoder_kernel<true>): u8 wire frames in, tokens never leave the CU
  const void* img;       // (B,60,90) u8 wire frames (TOK 1) or f32 frames (TOK 2)
  const float* tok_w;    // [53][64]: conv weights k-major (rows 0..48), rows 49..51 zero, row 52 = conv bias;
                         // channel c of an odd row k is stored at c ^ 16 (LDS bank spread for the MFMA B reads)
  const float *tok_lnw, *tok_lnb;
  float* tok_tap;        // optional (B,128,64): the tokens (LayerNorm output of the tokenizer)
};

struct ItaEncLds {
  static constexpr int S = 128, E = 64, P = 192, F = 256;
  static constexpr int XQ = 0;                    // int8 [4][128][16]; also the block outputs (out_q)
  static constexpr int Q = XQ + S * E;            // int8 [12][128][16]; context overwrites it
  static constexpr int K = Q + S * P;
  static constexpr int VT = K + S * P;            // int8 [8][192][16]
  static constexpr int H = K;                     // int8 [16][128][16] FFN hidden (K and V^T are dead by then)
  static constexpr int COLSUM = VT + P * S;       // int32 [192]
  static constexpr int WQ = COLSUM + P * 4;       // int8 [4][192][16] x3
  static constexpr int WK = WQ + P * E;
  static constexpr int WV = WK + P * E;
  static constexpr int W1 = WV + P * E;           // int8 [4][256][16]
  static constexpr int W2 = W1 + F * E;           // int8 [16][64][16]
  static constexpr int BIAS = W2 + E * F;         // int32: bq 192 | bk 192 | bv 192 | bo 64 | b1 256 | b2 64
  static constexpr int LNP = BIAS + (3 * P + E + F + E) * 4;   // f32: n1w | n1b | n2w | n2b (64 each)
  static constexpr int TLN = LNP + 4 * E * 4;      // f32: tokenizer LayerNorm w | b
  static constexpr int LUT = TLN + 2 * E * 4;      // f32 [256]: k / 255.0f
  static constexpr int TOTAL = LUT + 256 * 4;
  // tokenizer scratch, alive only between two frames, over the dead Q | K | V^T images:
  static constexpr int TK_IMG = Q;                         // f32 [66][96] frame / 255 with zero border
  static constexpr int TK_WT = TK_IMG + 66 * 96 * 4;       // f32 [53][64]
  static constexpr int TK_WAVE = TK_WT + 53 * 64 * 4;      // per wave: f32 [16][52] patches, then [16][68] pre-LN tokens
  static constexpr int TK_WAVE_BYTES = 16 * 68 * 4;
};
static_assert(ItaEncLds::TK_WAVE + 8 * ItaEncLds::TK_WAVE_BYTES <= ItaEncLds::COLSUM, "tokenizer scratch must fit in Q | K | V^T");
static_assert(ItaEncLds::H + 128 * 256 <= ItaEncLds::COLSUM, "FFN hidden layer must fit in the K + V^T region");
static_assert(ItaEncLds::TOTAL <= 160 * 1024, "LDS budget");

// weights [ROWS][KB] row-major in global -> chunk-major LDS image, in two steps so that the caller can
// issue the loads of ALL matrices before the first LDS store: the memory latency is paid once per
// workgroup, not once per matrix.
template <int KB, int ROWS>
struct ItaWeightStage {
  static constexpr int CH = KB / 16, NP = ROWS * CH, N = (NP + 511) / 512;
  i32x4 v[N];
  __device__ __forceinline__ void fetch(const int8_t* __restrict__ w, int tid) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int p = tid + 512 * j;
      v[j] = (i32x4){0, 0, 0, 0};
      if (p < NP) v[j] = *(const i32x4*)(w + (size_t)p * 16);
    }
  }
  __device__ __forceinline__ void put(char* dst, int tid) const {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int p = tid + 512 * j, row = p / CH, ch = p - row * CH;
      if (p < NP) *(i32x4*)(dst + ((ch * ROWS + row) << 4)) = v[j];
    }
  }
};

// Requantise a 32x32 accumulator tile (rows = 32 features f0.., columns = 32 tokens t0..) and store it into a
// chunk-major activation image.  Lane (r, h) holds, for token t0 + r, features f0 + 8g + 4h .. +3 (g = 0..3), i.e.
// dwords of two different 16-byte slots; store_tile_fx writes them as four 4-byte stores that are 8-way bank
// conflicted (64 lanes x 4 B at a 16-byte stride).  Two v_permlane32_swap give lane (r, 0) the whole slot of features
// f0..f0+15 and lane (r, 1) the slot f0+16..f0+31: ONE conflict-free 16-byte store per lane.
template <typename ACC>
__device__ __forceinline__ void store_tile_cm16(const ACC& acc, float mult, float lo, char* dst, int f0, int t0, int lane) {
  const int r = lane & 31, h = lane >> 5;
  unsigned p4[4];
  rq_pack16(acc, mult, lo, p4);
  const auto s02 = __builtin_amdgcn_permlane32_swap(p4[0], p4[2], false, false);   // [0]: g0@h0 | g2@h0, [1]: g0@h1 | g2@h1
  const auto s13 = __builtin_amdgcn_permlane32_swap(p4[1], p4[3], false, false);
  *(i32x4*)(dst + cm_off(t0 + r, f0 + 16 * h, 128)) = (i32x4){(int)s02[0], (int)s02[1], (int)s13[0], (int)s13[1]};
}

// 32x32 tile: A = weights from a chunk-major LDS image (WROWS rows), Bt = activations (128 rows)
template <int KB, int WROWS>
__device__ __forceinline__ i32x16 tile_wlds_x(const char* lds_w, int f0, const int* lds_bias, const char* lds_act,
                                              int t0, int lane) {
  const int r = lane & 31, h = lane >> 5;
  i32x16 acc;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const i32x4 b4 = *(const i32x4*)(lds_bias + f0 + 8 * g + 4 * h);
    acc[4 * g] = b4.x; acc[4 * g + 1] = b4.y; acc[4 * g + 2] = b4.z; acc[4 * g + 3] = b4.w;
  }
#pragma unroll
  for (int ks = 0; ks < KB / 32; ++ks) {
    const i32x4 a = lds_frag(lds_w, cm_off(f0 + r, 32 * ks + 16 * h, WROWS));
    const i32x4 b = lds_frag(lds_act, cm_off(t0 + r, 32 * ks + 16 * h, 128));
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
  }
  return acc;
}

// 32x32 tile with the weight fragments already in registers
template <int KSTEPS>
__device__ __forceinline__ i32x16 tile_wreg_x(const i32x4 (&wf)[KSTEPS], const int* lds_bias, int f0,
                                              const char* lds_act, int t0, int lane) {
  const int r = lane & 31, h = lane >> 5;
  i32x16 acc;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const i32x4 b4 = *(const i32x4*)(lds_bias + f0 + 8 * g + 4 * h);
    acc[4 * g] = b4.x; acc[4 * g + 1] = b4.y; acc[4 * g + 2] = b4.z; acc[4 * g + 3] = b4.w;
  }
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const i32x4 b = lds_frag(lds_act, cm_off(t0 + r, 32 * ks + 16 * h, 128));
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[ks], b, acc, 0, 0, 0);
  }
  return acc;
}

// TOK (0 none, 1 u8 wire frames, 2 f32 frames): the layer's input is computed in place from the frames (OverlapPatchMerging, reference
// models/ITA/QAT/layers.py:39-45; same arithmetic and operation order as ita_tokenizer_kernel) instead of
// being read from a.x: between two frames the workgroup tokenizes its next frame into the registers that
// phase 0 quantises -- no token round trip through HBM and one kernel boundary less.
template <int TOK>
__global__ __launch_bounds__(512) void ita_encoder_kernel(const ItaEncArgs a) {
  using L = ItaEncLds;
  constexpr int S = 128, E = 64, P = 192, F = 256, EC = 16;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int token = tid >> 2, qtr = tid & 3;
  const int r = lane & 31, h = lane >> 5;
  int* colsum = (int*)(lds + L::COLSUM);
  int* bias = (int*)(lds + L::BIAS);
  const int *l_bq = bias, *l_bk = bias + P, *l_bv = bias + 2 * P, *l_bo = bias + 3 * P, *l_b1 = bias + 3 * P + E,
            *l_b2 = bias + 3 * P + E + F;

  // ---- once per workgroup: weights, biases, LayerNorm parameters, this wave's out_proj fragments and
  // the first frame's tokens -- every global load is issued before the first wait
  const int et = wave >> 2, tt = wave & 3;   // out_proj / fc2 output tile of this wave: features et*32.., tokens tt*32..
  float* lnp = (float*)(lds + L::LNP);
  i32x4 wo_f[6];                             // (there is no LDS left for Wo: its fragments live in registers)
  float xr[EC];
  {
    ItaWeightStage<E, P> sq, sk, sv;
    ItaWeightStage<E, F> s1;
    ItaWeightStage<F, E> s2;
    sq.fetch(a.wq, tid); sk.fetch(a.wk, tid); sv.fetch(a.wv, tid); s1.fetch(a.w1, tid); s2.fetch(a.w2, tid);
    constexpr int NB = 3 * P + E + F + E;
    int bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = tid + 512 * j;
      int v = 0;
      if (i < P) v = a.bq[i];
      else if (i < 2 * P) v = a.bk[i - P];
      else if (i < 3 * P) v = a.bv[i - 2 * P];
      else if (i < 3 * P + E) v = a.bo[i - 3 * P];
      else if (i < 3 * P + E + F) v = a.b1[i - 3 * P - E];
      else if (i < NB) v = a.b2[i - 3 * P - E - F];
      bv[j] = v;
    }
    float lv = 0.0f;
    if (tid < 4 * E) {
      const int which = tid >> 6, c = tid & 63;
      lv = which == 0 ? a.n1w[c] : which == 1 ? a.n1b[c] : which == 2 ? a.n2w[c] : a.n2b[c];
    }
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) wo_f[ks] = gl_frag(a.wo + (size_t)(et * 32 + r) * P + 32 * ks + 16 * h);
    float tlv = 0.0f;
    if constexpr (TOK) {
      if (tid < 2 * E) tlv = tid < E ? a.tok_lnw[tid] : a.tok_lnb[tid - E];
    } else if ((int)blockIdx.x < a.B) {
      const float* xrow = a.x + ((size_t)blockIdx.x * S + token) * E + qtr * EC;
#pragma unroll
      for (int i = 0; i < EC; i += 4) {
        const f32x4 v = *(const f32x4*)(xrow + i);
        xr[i] = v.x; xr[i + 1] = v.y; xr[i + 2] = v.z; xr[i + 3] = v.w;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TOK) {
      if (tid < 2 * E) ((float*)(lds + L::TLN))[tid] = tlv;
      if (tid < 256) ((float*)(lds + L::LUT))[tid] = (float)tid / 255.0f;   // the reference host's float(pixel) / 255.0f (main.cpp:168-169)
    }
    sq.put(lds + L::WQ, tid); sk.put(lds + L::WK, tid); sv.put(lds + L::WV, tid); s1.put(lds + L::W1, tid);
    s2.put(lds + L::W2, tid);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (tid + 512 * j < NB) bias[tid + 512 * j] = bv[j];
    if (tid < 4 * E) lnp[tid] = lv;
  }

  int fi = 0;
#define ITA_STAMP(ph)                                                                                   \
  do {                                                                                                  \
    if (a.stamps && tid == 0 && fi < 8) a.stamps[((size_t)blockIdx.x * 8 + fi) * 16 + (ph)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
  // ---- fused tokenizer (TOK): fetch = global loads of one frame + the conv weights into registers,
  // tokenize = those registers -> this thread's 16 channels of its token in xr.
  constexpr int NPX = TOK == 2 ? 3 : 1;      // 16-byte pieces of a frame per thread: 338 (u8) or 1350 (f32) in all
  i32x4 tk_px[NPX], tk_w[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  auto tok_fetch = [&](int fb) {
    if constexpr (TOK == 2) {
      const float* src = (const float*)a.img + (size_t)fb * 5400;
#pragma unroll
      for (int j = 0; j < NPX; ++j) {
        tk_px[j] = (i32x4){0, 0, 0, 0};
        if (tid + 512 * j < 1350) tk_px[j] = *(const i32x4*)(src + 4 * (tid + 512 * j));
      }
    } else {
      const uint8_t* src = (const uint8_t*)a.img + (size_t)fb * 5400;
      tk_px[0] = (i32x4){0, 0, 0, 0};
      if (tid < 337) tk_px[0] = *(const i32x4*)(src + 16 * tid);
      else if (tid == 337) { tk_px[0].x = *(const int*)(src + 5392); tk_px[0].y = *(const int*)(src + 5396); }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (tid + 512 * j < 53 * 16) tk_w[j] = *(const i32x4*)(a.tok_w + 4 * (tid + 512 * j));
  };
  // stage: the prefetched registers -> LDS (frame as f32 through the k/255 table, zero border, conv
  // weights).  Runs right after the barrier that ends phase F2, BEFORE phase L2 issues its global stores:
  // consuming the prefetched loads any later would wait on those stores too (one in-order vmcnt).
  auto tok_stage = [&]() {
    float* img = (float*)(lds + L::TK_IMG);
    float* wt = (float*)(lds + L::TK_WT);
    const float* lut = (const float*)(lds + L::LUT);
    int otid = tid;
    asm volatile("" : "+v"(otid));   // see tok_compute
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (otid + 512 * j < 53 * 16) *(i32x4*)(wt + 4 * (otid + 512 * j)) = tk_w[j];
    // 936 border cells (rows 0-2, 63-65; columns 0-2, 93-95 of the 60 image rows): disjoint from the pixels
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int bc = otid + 512 * j;
      if (bc < 936) {
        int row, col;
        if (bc < 576) { const int q = bc / 96; row = q < 3 ? q : q + 60; col = bc - 96 * q; }
        else { const int e = bc - 576, q = e / 6, k = e - 6 * q; row = 3 + q; col = k < 3 ? k : 90 + k; }
        img[row * 96 + col] = 0.0f;
      }
    }
    if constexpr (TOK == 2) {
      // four consecutive f32 pixels per piece; a piece spans at most two image rows
#pragma unroll
      for (int j = 0; j < NPX; ++j) {
        const int p = otid + 512 * j;
        if (p < 1350) {
          const int y = (4 * p) / 90, x = 4 * p - 90 * y, n0 = 90 - x;
          float* dst = img + (y + 3) * 96 + x + 3;
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[e + (e >= n0 ? 6 : 0)] = __int_as_float(tk_px[j][e]);
        }
      }
    } else {
      // 16 consecutive pixels per thread; they span at most two image rows.  Chunk 337's eight pad bytes
      // (code 0 -> 0.0f) land in the bottom border.
      if (otid < 338) {
        const int y = (16 * otid) / 90, x = 16 * otid - 90 * y, n0 = 90 - x;
        float* dst = img + (y + 3) * 96 + x + 3;
#pragma unroll
        for (int e = 0; e < 16; ++e)
          dst[e + (e >= n0 ? 6 : 0)] = lut[((unsigned)tk_px[0][e >> 2] >> (8 * (e & 3))) & 0xffu];
      }
    }
  };
  auto tok_compute = [&](int fb) {
    const float* img = (const float*)(lds + L::TK_IMG);
    const float* wt = (const float*)(lds + L::TK_WT);
    float* pb = (float*)(lds + L::TK_WAVE + wave * L::TK_WAVE_BYTES);   // [16][52], then [16][68]
    // everything below is derived from an opaque copy of the thread id: otherwise the compiler hoists the
    // per-thread geometry (pixel addresses, bilinear weights) out of the frame loop and keeps ~60 registers
    // alive across the register-critical attention phase
    int otid = tid;
    asm volatile("" : "+v"(otid));
    const int token = otid >> 2, qtr = otid & 3, lane = otid & 63;
    lds_barrier();
    ITA_STAMP(9);
    // T2: blended 7x7 patches (conv and bilinear resize are both linear): thread (token, part) takes kernel
    // rows 2*part, 2*part+1 (part 3: row 6 and the zero pad columns)
    {
      const int part = qtr, tl = token & 15;
      int y0, yp, x0, xp;
      float ly, lx;
      bilinear_src_dev(token >> 4, 30.0f / 8.0f, 30, y0, yp, ly);
      bilinear_src_dev(token & 15, 45.0f / 16.0f, 45, x0, xp, lx);
      const float h1 = ly, h0 = 1.0f - ly, w1 = lx, w0 = 1.0f - lx;
      // the four bilinear neighbours of a tap are the same window shifted by (2*yp rows, 2*xp columns): four
      // 8-byte aligned row reads instead of selects; two taps per VALU op (v_pk_mul_f32 / v_pk_add_f32 round
      // exactly like the scalar ops)
      const float* wa = img + (2 * y0 + 2 * part) * 96 + 2 * x0;   // (-3 conv padding) + (3 border) = 0
      const float* wb = wa + 2 * xp;
      const float* wc = wa + 2 * yp * 96;
      const float* wd = wc + 2 * xp;
      const f32x2 h0v = {h0, h0}, h1v = {h1, h1}, w0v = {w0, w0}, w1v = {w1, w1};
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        if (jj == 0 || part < 3) {
          float* prow = pb + tl * 52 + (2 * part + jj) * 7;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const f32x2 va = *(const f32x2*)(wa + jj * 96 + 2 * c), vb = *(const f32x2*)(wb + jj * 96 + 2 * c);
            const f32x2 vc = *(const f32x2*)(wc + jj * 96 + 2 * c), vd = *(const f32x2*)(wd + jj * 96 + 2 * c);
            const f32x2 t = h0v * (w0v * va + w1v * vb) + h1v * (w0v * vc + w1v * vd);
            prow[2 * c] = t.x;
            if (c < 3) prow[2 * c + 1] = t.y;
          }
        }
      }
      if (part == 3) { pb[tl * 52 + 49] = 0.0f; pb[tl * 52 + 50] = 0.0f; pb[tl * 52 + 51] = 0.0f; }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the patch rows are private to this wave
    __builtin_amdgcn_wave_barrier();
    ITA_STAMP(10);
    // T3: tokens[16 x 64] = bias + patches[16 x 52] . wt[52 x 64] on v_mfma_f32_16x16x4_f32, on gfx950 an
    // exact ascending-k fmaf chain -- the oracle's order
    {
      const int col = lane & 15, kq = lane >> 4;
      f32x4 acc[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const float bv = wt[52 * 64 + ct * 16 + col];
        acc[ct] = (f32x4){bv, bv, bv, bv};
      }
#pragma unroll
      for (int s = 0; s < 13; ++s) {
        const float av = pb[col * 52 + 4 * s + kq];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wt[(4 * s + kq) * 64 + ((ct * 16 + col) ^ ((kq & 1) << 4))],
                                                         acc[ct], 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();
      // C layout: col = channel ct*16 + (lane&15), row = token 4*(lane>>4) + i  -> pre[token][68] over the patches
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int i = 0; i < 4; ++i) pb[(4 * kq + i) * 68 + ct * 16 + col] = acc[ct][i];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    ITA_STAMP(11);
    // T4: LayerNorm of this thread's 16 channels of its token
    {
      const float* pre = pb + (token & 15) * 68 + qtr * EC;
#pragma unroll
      for (int i = 0; i < EC; i += 4) {
        const f32x4 v = *(const f32x4*)(pre + i);
        xr[i] = v.x; xr[i + 1] = v.y; xr[i + 2] = v.z; xr[i + 3] = v.w;
      }
      const float* tln = (const float*)(lds + L::TLN);
      layernorm_lanes<E, 4>(xr, tln, tln + E, qtr * EC);
      if (a.tok_tap) {
        float* o = a.tok_tap + ((size_t)fb * S + token) * E + qtr * EC;
#pragma unroll
        for (int i = 0; i < EC; i += 4) *(f32x4*)(o + i) = (f32x4){xr[i], xr[i + 1], xr[i + 2], xr[i + 3]};
      }
    }
    ITA_STAMP(12);
  };
  if constexpr (TOK) {
    lds_barrier();   // LUT and LayerNorm parameters are in place
    if ((int)blockIdx.x < a.B) {
      tok_fetch(blockIdx.x);
      tok_stage();
      tok_compute(blockIdx.x);
    }
  }

  for (int b = blockIdx.x; b < a.B; b += gridDim.x, ++fi) {
    ITA_STAMP(0);
    // ---------------- phase 0: quantise (xr holds this thread's 16 channels of one token)
    {
      unsigned p4[4];
      q_pack16(xr, a.inv_sx, p4);
      *(i32x4*)(lds + L::XQ + cm_off(token, qtr * EC, 128)) = (i32x4){(int)p4[0], (int)p4[1], (int)p4[2], (int)p4[3]};
    }
    if (tid < P) colsum[tid] = 0;
    if (a.h0_dst && tid < 32) {
      const size_t row = a.slots ? (size_t)a.slots[b] : (size_t)b;
      *(f32x4*)(a.h0_dst + (size_t)b * 128 + 4 * tid) = *(const f32x4*)(a.h0_src + row * 128 + 4 * tid);
    }
    lds_barrier();
    ITA_STAMP(1);

    // ---------------- phase P: Q, K, V projections.  Each wave owns token tile tt and eleven or seven feature tiles
    // (waves 0-3: Q0-5, K0-4; waves 4-7: K5, V0-5), processed TWO AT A TIME: both tiles' fragment reads
    // and MFMAs are issued before either epilogue, so the VALU-heavy requantisation of one tile overlaps
    // the LDS/MFMA latency of the other (the wave has only one SIMD partner to hide latency behind).
    {
      auto qk_acc = [&](int mat, int dt) {
        return tile_wlds_x<E, P>(lds + (mat == 0 ? L::WQ : L::WK), dt * 32, mat == 0 ? l_bq : l_bk, lds + L::XQ,
                                 tt * 32, lane);
      };
      auto qk_store = [&](const i32x16& acc, int mat, int dt) {
        store_tile_cm16(acc, mat == 0 ? a.mq : a.mk, -128.0f, lds + (mat == 0 ? L::Q : L::K), dt * 32, tt * 32, lane);
      };
      auto v_acc = [&](int dt) {
        const int d = dt * 32 + r;
        const int bias_d = l_bv[d];
        i32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = bias_d;
#pragma unroll
        for (int ks = 0; ks < E / 32; ++ks) {
          const i32x4 xa = lds_frag(lds + L::XQ, cm_off(tt * 32 + r, 32 * ks + 16 * h, 128));
          const i32x4 wb = lds_frag(lds + L::WV, cm_off(d, 32 * ks + 16 * h, P));
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa, wb, acc, 0, 0, 0);
        }
        return acc;
      };
      auto v_store = [&](const i32x16& acc, int dt) {
        // acc[4g+i] <-> key tt*32 + 8g + 4h + i, feature d; keys are stored permuted inside each 64-key
        // block (ita_int8_kernels.h).  The column sum of the requantised V codes (for the
        // unsigned-probability offset) is a byte dot product of the packed result.
        const int d = dt * 32 + r;
        unsigned p4[4];
        rq_pack16(acc, a.mv, -128.0f, p4);
        // p4[g] belongs to slot kq = 2*(g&1) + h at word t = 2*(tt&1) + (g>>1): words g and g+2 are adjacent
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
          const int kb = tt >> 1, kq = 2 * gq + h;
          *(u32x2*)(lds + L::VT + (((kb * 4 + kq) * P + d) << 4) + 8 * (tt & 1)) = (u32x2){p4[gq], p4[gq + 2]};
        }
        int csum = sum_bytes16(p4);
        csum += xor32_i(csum);
        if (h == 0) atomicAdd(&colsum[d], csum << 7);   // kept pre-multiplied by the 128 offset
      };
      // A V tile costs ~1.6x a Q/K tile (scattered 4-byte V^T writes, column sums), measured with the per-wave
      // arrival stamps: waves 0-3 take eleven Q/K tiles, waves 4-7 one K tile and the six V tiles.
      if (wave < 4) {
#pragma unroll 1
        for (int ft = 0; ft < 10; ft += 2) {         // (Q0,Q1) (Q2,Q3) (Q4,Q5) (K0,K1) (K2,K3)
          const int m0 = ft / 6, d0 = ft - 6 * m0, m1 = (ft + 1) / 6, d1 = ft + 1 - 6 * m1;
          const i32x16 a0 = qk_acc(m0, d0), a1 = qk_acc(m1, d1);
          qk_store(a0, m0, d0);
          qk_store(a1, m1, d1);
        }
        qk_store(qk_acc(1, 4), 1, 4);                 // K4
      } else {
        {
          const i32x16 a0 = qk_acc(1, 5), a1 = v_acc(0);   // K5, V0
          qk_store(a0, 1, 5);
          v_store(a1, 0);
        }
#pragma unroll 1
        for (int dt = 1; dt < 5; dt += 2) {
          const i32x16 a0 = v_acc(dt), a1 = v_acc(dt + 1);
          v_store(a0, dt);
          v_store(a1, dt + 1);
        }
        v_store(v_acc(5), 5);
      }
    }
    if (a.stamps && fi < 8 && (tid == 0 || tid == 256))     // diagnostic: arrival of waves 0 and 4 at the barrier
      a.stamps[((size_t)blockIdx.x * 8 + fi) * 16 + 13 + (tid >> 8)] = __builtin_amdgcn_s_memtime();
    lds_barrier();
    ITA_STAMP(2);

    // prefetch of the next frame's tokens (consumed at the next phase 0): the four 16-byte loads are
    // spread over the logits loop below -- eight waves issuing them back to back would each sit at the
    // head of the CU's address path (64 B/clk) instead of issuing MFMAs
    float xn[EC];
    const int nb = b + gridDim.x;
    const float* xnrow = TOK ? nullptr : a.x + ((size_t)min(nb, a.B - 1) * S + token) * E + qtr * EC;

    // ---------------- phase A: 16 queries per wave, logits and probabilities stay in registers
    {
      const int q0 = wave * 16, qi = lane & 15, kq = lane >> 4;
      i32x4 qf[3];
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) qf[ks] = lds_frag(lds + L::Q, cm_off(q0 + qi, 64 * ks + 16 * kq, 128));
      // logits as packed signed 16-bit pairs: the whole integer softmax below then runs on v_pk_*_16
      // instructions, two keys per VALU op.  w[2kt + j] = {logit 4kt+2j, logit 4kt+2j+1}.
      typedef short s16x2 __attribute__((ext_vector_type(2)));
      typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
      s16x2 w[16];
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) {
        i32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const i32x4 kf = lds_frag(lds + L::K, cm_off(kt * 16 + qi, 64 * ks + 16 * kq, 128));
          acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(kf, qf[ks], acc, 0, 0, 0);
        }
        if (!TOK && (kt & 1) == 0) {
          const f32x4 v = *(const f32x4*)(xnrow + 2 * kt);
          xn[2 * kt] = v.x; xn[2 * kt + 1] = v.y; xn[2 * kt + 2] = v.z; xn[2 * kt + 3] = v.w;
          __builtin_amdgcn_sched_barrier(0);
        }
        float lf[4];
        scale_clamp<4>(acc, a.ml, -128.0f, lf);
        unsigned bi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bi[i] = __float_as_uint(lf[i] + ITA_MAGIC_F);   // low 16 bits = rne(logit), two's complement
#pragma unroll
        for (int j = 0; j < 2; ++j)
          w[2 * kt + j] = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(bi[2 * j + 1], bi[2 * j], 0x05040100u));
      }
      s16x2 m2 = w[0];
#pragma unroll
      for (int j = 1; j < 16; ++j) m2 = __builtin_elementwise_max(m2, w[j]);
      int m = max((int)m2.x, (int)m2.y);
      m = max(m, xor16_i(m));
      m = max(m, xor32_i(m));
      const s16x2 mm = {(short)m, (short)m};
      const s16x2 cap = {15, 15};                  // 256 >> s and inv_hi >> s are both 0 from s = 9 on
      const u16x2 one = {256, 256};
      u16x2 sh[16], sum2 = {0, 0};
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        sh[j] = __builtin_bit_cast(u16x2, __builtin_elementwise_min((s16x2)(mm - w[j]), cap));
        sum2 += one >> sh[j];
      }
      int sum = (int)sum2.x + (int)sum2.y;         // <= 16 * 256 per half: no 16-bit overflow
      sum += xor16_i(sum);
      sum += xor32_i(sum);
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
#pragma unroll
      for (int dg = 0; dg < 3; ++dg) {          // four 16-feature tiles at a time: 16 values per lane
        float cf[16];
#pragma unroll
        for (int di = 0; di < 4; ++di) {
          const int dt = 4 * dg + di;
          i32x4 acc = *(const i32x4*)(colsum + dt * 16 + 4 * kq);   // 128 * column sum
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            const i32x4 vf = lds_frag(lds + L::VT, ((((kb * 4 + kq) * P) + dt * 16 + qi) << 4));
            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(vf, pf[kb], acc, 0, 0, 0);
          }
          float t4[4];
          scale_clamp<4>(acc, a.mc, -128.0f, t4);
#pragma unroll
          for (int i = 0; i < 4; ++i) cf[4 * di + i] = t4[i];
        }
        unsigned c4[4];
        round_pack16(cf, c4);
        // c4[di] <-> features (4dg+di)*16 + 4*kq .. +3, query q0 + qi; the context overwrites this wave's own Q rows
#pragma unroll
        for (int di = 0; di < 4; ++di)
          *(unsigned*)(lds + L::Q + cm_off(q0 + qi, (4 * dg + di) * 16 + 4 * kq, 128)) = c4[di];
      }
    }
    lds_barrier();
    ITA_STAMP(3);

    // ---------------- phase O: out_proj, weights in registers -> out_q (chunk-major, over x_q)
    {
      const i32x16 acc = tile_wreg_x<6>(wo_f, l_bo, et * 32, lds + L::Q, tt * 32, lane);
      store_tile_cm16(acc, a.mo, -128.0f, lds + L::XQ, et * 32, tt * 32, lane);
    }
    lds_barrier();
    ITA_STAMP(4);

    // ---------------- phase L1: x1 = LN1(x + dequant(out_q)); quantise x1 for the FFN in place
    if constexpr (TOK) {
      if (nb < a.B) tok_fetch(nb);   // next frame's pixels and the conv weights: in flight until phase L2 is over
    }
    float x1[EC];
    {
      const i32x4 pk = *(const i32x4*)(lds + L::XQ + cm_off(token, qtr * EC, 128));
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int vv = (int)(int8_t)((unsigned)pk[j >> 2] >> (8 * (j & 3)));
        x1[j] = xr[j] + (float)vv * a.so;
      }
      layernorm_lanes<E, 4>(x1, lnp, lnp + E, qtr * EC);
      if (a.x1_tap) {
        float* o = a.x1_tap + ((size_t)b * S + token) * E + qtr * EC;
#pragma unroll
        for (int i = 0; i < EC; i += 4) *(f32x4*)(o + i) = (f32x4){x1[i], x1[i + 1], x1[i + 2], x1[i + 3]};
      }
      unsigned p4[4];
      q_pack16(x1, a.f_inv_sx, p4);   // written over the same 16 bytes this thread just read
      *(i32x4*)(lds + L::XQ + cm_off(token, qtr * EC, 128)) = (i32x4){(int)p4[0], (int)p4[1], (int)p4[2], (int)p4[3]};
    }
    lds_barrier();
    ITA_STAMP(5);

    // ---------------- phase F1: fc1 + ReLU -> hidden (chunk-major over the dead K / V^T images)
#pragma unroll 1
    for (int ft = (wave >> 2) * 4; ft < (wave >> 2) * 4 + 4; ft += 2) {   // two tiles in flight, as in phase P
      const i32x16 a0 = tile_wlds_x<E, F>(lds + L::W1, ft * 32, l_b1, lds + L::XQ, tt * 32, lane);
      const i32x16 a1 = tile_wlds_x<E, F>(lds + L::W1, (ft + 1) * 32, l_b1, lds + L::XQ, tt * 32, lane);
      store_tile_cm16(a0, a.m1, 0.0f, lds + L::H, ft * 32, tt * 32, lane);
      store_tile_cm16(a1, a.m1, 0.0f, lds + L::H, (ft + 1) * 32, tt * 32, lane);
    }
    lds_barrier();
    ITA_STAMP(6);

    // ---------------- phase F2: fc2 -> out_q (over the FFN's x_q)
    {
      const i32x16 acc = tile_wlds_x<F, E>(lds + L::W2, et * 32, l_b2, lds + L::H, tt * 32, lane);
      store_tile_cm16(acc, a.m2, -128.0f, lds + L::XQ, et * 32, tt * 32, lane);
    }
    lds_barrier();
    ITA_STAMP(7);
    if constexpr (TOK) {
      // the Q | K | V^T images are dead from here on: the tokenizer's scratch
      if (nb < a.B) tok_stage();
    }

    // ---------------- phase L2: y = LN2(x1 + dequant(out_q))
    {
      const i32x4 pk = *(const i32x4*)(lds + L::XQ + cm_off(token, qtr * EC, 128));
      float y[EC];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int vv = (int)(int8_t)((unsigned)pk[j >> 2] >> (8 * (j & 3)));
        y[j] = x1[j] + (float)vv * a.s2;
      }
      layernorm_lanes<E, 4>(y, lnp + 2 * E, lnp + 3 * E, qtr * EC);
      const size_t o = ((size_t)b * S + token) * E + qtr * EC;
      if (a.y) {
#pragma unroll
        for (int i = 0; i < EC; i += 4) *(f32x4*)(a.y + o + i) = (f32x4){y[i], y[i + 1], y[i + 2], y[i + 3]};
      }
      if (a.y_hi) {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int i = 0; i < EC; i += 8) {
          h8 vh, vl;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const _Float16 hh = (_Float16)y[i + j];
            vh[j] = hh;
            vl[j] = (_Float16)(y[i + j] - (float)hh);
          }
          const size_t po = (size_t)b * a.ld_planes + token * E + qtr * EC + i;
          *(h8*)(a.y_hi + po) = vh;
          *(h8*)(a.y_lo + po) = vl;
        }
      }
    }
    ITA_STAMP(8);
    if constexpr (TOK) {
      if (nb < a.B) tok_compute(nb);
    } else {
#pragma unroll
      for (int i = 0; i < EC; ++i) xr[i] = xn[i];
    }
    // phase 0 of the next frame writes x_q, which phase L2 above has just read: the read and the
    // write of a given 16-byte slot are by the same thread, so no barrier is needed here
  }
}
