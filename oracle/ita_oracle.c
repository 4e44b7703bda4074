/*
 * ita_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's ITA ViT+LSTM inference path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's library
 * (oracle/libita_oracle.so); the product (libita_mi355x.so) never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every stage below against
 * golden vectors produced by running the reference's own PyTorch int8 modules in the build
 * container (tools/gen_golden.py -> tests/golden/ *.npz).  Integer stages agree bit for bit,
 * with one documented exception: the reference computes matmul1 (QK^T) through
 * dequantize -> float sgemm -> quantize (torch.ops.quantized.matmul fallback), so ~1 element in
 * 32k whose exact product lies within 1e-4 of a rounding tie follows the sgemm's float32
 * accumulation error; this restatement accumulates in int32 (what the ITA hardware and the
 * reference's own validation harness, tests/export_and_validation_W_B.py:120-151, do).
 * The PyITA bit-true simulator the reference validates against (MAE <= 1 LSB,
 * tests/export_and_validation_W_B.py:308-334) is an absent submodule: parity to it is unpinned.
 *
 * Reference lines followed (all under /root/reference):
 *   refine_inputs / forward ........ models/ITA_single_layer_upsample_shuffle/QAT/model.py:22-31,93-132
 *   forward without the fusion tail  models/ITA/QAT/model.py:62-87 (2 layers, E = 128, decoder on the flattened tokens)
 *   OverlapPatchMerging ............ models/ITA/QAT/layers.py:25-45
 *   ITASelfAttention_QAT ........... models/ITA/QAT/layers.py:77-127
 *   ITAFeedForward_QAT ............. models/ITA/QAT/layers.py:47-75
 *   IntegerApproximatedSoftmax ..... models/ITA/QAT/ITA_softmax.py:29-35,51-61,66-73
 *   matmul2 (quint8 x qint8) ....... tests/export_and_validation_W_B.py:120-151
 *   bias -> int32 .................. tests/export_and_validation_W_B.py:233-245
 *
 * Float stages use one fixed operation order (fmaf chains in ascending k, LayerNorm sums in
 * four blocks) so that the HIP kernels, which use the same order, can be compared for
 * equality rather than within a tolerance.  Build with -ffp-contract=off.
 */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ita_weights.h"

#define TOK_H 8
#define TOK_W 16
#define IMG_H 60
#define IMG_W 90
#define CONV_H 30
#define CONV_W 45

/* ------------------------------------------------------------------ small helpers */

static inline float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

/* exp / sigmoid / tanh with a fixed arithmetic (Cephes-style range reduction + degree-5
 * polynomial, fmaf Horner).  The HIP kernels carry the identical sequence, so the LSTM is
 * comparable bit for bit; against PyTorch's libm-based LSTM the difference is < 3e-7. */
float ita_oracle_expf(float x) {
  x = clampf(x, -87.0f, 88.0f);
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
  union { uint32_t u; float f; } s;
  s.u = (uint32_t)((int32_t)n + 127) << 23;
  return p * s.f;
}
static inline float sigmoidf_(float x) { return 1.0f / (1.0f + ita_oracle_expf(-x)); }
static inline float tanhf_(float x) { return 1.0f - 2.0f / (ita_oracle_expf(2.0f * x) + 1.0f); }

/* ------------------------------------------------------------------ LayerNorm (eps 1e-5) */
/* Sums are taken in 4 blocks of E/4 consecutive channels, combined (p0+p1)+(p2+p3). */
static void layernorm_row(const float* x, int E, const float* w, const float* b, float* y) {
  const int q = E / 4;
  float p[4];
  for (int j = 0; j < 4; ++j) {
    float s = 0.0f;
    for (int e = 0; e < q; ++e) s = s + x[j * q + e];
    p[j] = s;
  }
  const float inv_e = 1.0f / (float)E;
  const float mean = ((p[0] + p[1]) + (p[2] + p[3])) * inv_e;
  for (int j = 0; j < 4; ++j) {
    float s = 0.0f;
    for (int e = 0; e < q; ++e) {
      float d = x[j * q + e] - mean;
      s = fmaf(d, d, s);
    }
    p[j] = s;
  }
  const float var = ((p[0] + p[1]) + (p[2] + p[3])) * inv_e;
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  for (int e = 0; e < E; ++e) y[e] = fmaf((x[e] - mean) * rstd, w[e], b[e]);
}

/* out = LayerNorm(x + y)    (QAT/model.py:100-113: float add, then nn.LayerNorm) */
void ita_oracle_add_ln(const float* x, const float* y, int rows, int E, const float* w, const float* b,
                       float* out) {
  float* t = (float*)malloc(sizeof(float) * (size_t)E);
  for (int r = 0; r < rows; ++r) {
    for (int e = 0; e < E; ++e) t[e] = x[(size_t)r * E + e] + y[(size_t)r * E + e];
    layernorm_row(t, E, w, b, out + (size_t)r * E);
  }
  free(t);
}

/* ------------------------------------------------------------------ tokenizer */
/* conv7x7 s2 p3 -> bilinear(30x45 -> 8x16, align_corners=False) -> LayerNorm.
 * conv and bilinear are both linear, so the four bilinear taps are blended on the 7x7 input
 * patches first (PyTorch's tap formula h0*(w0*a+w1*b)+h1*(w0*c+w1*d)) and ONE 49-long fmaf
 * chain per (token, channel) follows, starting from the conv bias. */
static void bilinear_src(int dst, int in, int out, int* i0, int* ip, float* l1) {
  const float scale = (float)in / (float)out;
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.0f) src = 0.0f;
  int i = (int)src;
  if (i > in - 1) i = in - 1;
  *i0 = i;
  *ip = (i < in - 1) ? 1 : 0;
  *l1 = src - (float)i;
}

static inline float img_at(const float* img, int y, int x) {
  return (y < 0 || y >= IMG_H || x < 0 || x >= IMG_W) ? 0.0f : img[y * IMG_W + x];
}

void ita_oracle_blend_patch(const float* img, int oy, int ox, float* pb /*49*/) {
  int y0, yp, x0, xp;
  float ly, lx;
  bilinear_src(oy, CONV_H, TOK_H, &y0, &yp, &ly);
  bilinear_src(ox, CONV_W, TOK_W, &x0, &xp, &lx);
  const float h1 = ly, h0 = 1.0f - ly, w1 = lx, w0 = 1.0f - lx;
  for (int ky = 0; ky < 7; ++ky)
    for (int kx = 0; kx < 7; ++kx) {
      const int iy = 2 * y0 - 3 + ky, ix = 2 * x0 - 3 + kx;
      const float a = img_at(img, iy, ix), b = img_at(img, iy, ix + 2 * xp);
      const float c = img_at(img, iy + 2 * yp, ix), d = img_at(img, iy + 2 * yp, ix + 2 * xp);
      pb[ky * 7 + kx] = h0 * (w0 * a + w1 * b) + h1 * (w0 * c + w1 * d);
    }
}

int ita_oracle_tokenizer(const float* img, int B, int E, const float* cw, const float* cb, const float* lnw,
                         const float* lnb, float* tokens) {
  float pb[49];
  float* pre = (float*)malloc(sizeof(float) * (size_t)E);
  for (int b = 0; b < B; ++b)
    for (int oy = 0; oy < TOK_H; ++oy)
      for (int ox = 0; ox < TOK_W; ++ox) {
        ita_oracle_blend_patch(img + (size_t)b * IMG_H * IMG_W, oy, ox, pb);
        for (int c = 0; c < E; ++c) {
          float acc = cb[c];
          for (int k = 0; k < 49; ++k) acc = fmaf(pb[k], cw[c * 49 + k], acc);
          pre[c] = acc;
        }
        layernorm_row(pre, E, lnw, lnb, tokens + ((size_t)b * 128 + oy * TOK_W + ox) * E);
      }
  free(pre);
  return 0;
}

/* The tokenizer on u8 WIRE FRAMES (what the UDP host receives, main.cpp:37,166).  The reference host divides every
 * pixel by 255.0f and the graph then convolves and resizes in f32; here the pixel codes stay integers as long as they can:
 * the bilinear weights of this fixed 30x45 -> 8x16 resize are dyadic (multiples of 1/8 vertically, 1/32 horizontally), so
 *     B256[tap] = H0*(W0*a + W1*b) + H1*(W0*c + W1*d),   H = 8h, W = 32w,   0 <= B256 <= 255 * 256
 * is an EXACT integer = 256 * 255 * (blended patch value).  Since round 3 the convolution is an exact integer too: the
 * conv weights of a channel become 23-bit fixed point with one power-of-two scale per channel,
 *     Wq[c][k] = rne(w[c][k] * 2^e_c),  e_c = 22 - exponent(max_k |w[c][k]|)   (|Wq| <= 2^22; all-zero channel: e_c = 0)
 * and the 49-tap sum runs on BYTES -- B256 = 256 a1 + a0, Wq = 65536 w2 + 256 w1 + w0 with balanced digits w0, w1 in
 * [-128, 127] -- as four int32 sums (what the GPU's int8 MFMA produces; every grouping is exact):
 *     S0 = sum a0 w0,  S1 = sum (a0 w1 + a1 w0),  S2 = sum (a0 w2 + a1 w1),  S3 = sum a1 w2
 *     L = S0 + 256 S1,  H = S2 + 256 S3                      (sum B256 * Wq = 65536 H + L, both below 2^31)
 *     pre[c] = fmaf((float)H, 65536 s_c, fmaf((float)L, s_c, bias[c])),   s_c = 2^-e_c / 65280.0f
 * Same linear map as ita_oracle_tokenizer on pixel / 255.0f, closer to the exact result than its 49-step f32 chain (three
 * f32 roundings instead of forty-nine), and (why it exists) the f32 conv -- v_mfma_f32_16x16x4_f32 runs at the f32 vector
 * rate and blocks the SIMD's VALU -- becomes 24 int8 MFMAs per wave.  The two entry points agree to ~1e-6, not bit for bit. */
void ita_oracle_tok_quant_weights(const float* cw, int E, int32_t* wq /*[E][49]*/, float* sc /*[E]*/) {
  for (int c = 0; c < E; ++c) {
    float mx = 0.0f;
    for (int k = 0; k < 49; ++k) mx = fmaxf(mx, fabsf(cw[c * 49 + k]));
    int e = 0;
    if (mx > 0.0f) {
      int ex;
      (void)frexpf(mx, &ex);      /* mx = m * 2^ex, m in [0.5, 1) */
      e = 22 - ex;                /* mx * 2^e in [2^21, 2^22) */
    }
    for (int k = 0; k < 49; ++k) wq[c * 49 + k] = (int32_t)rintf(ldexpf(cw[c * 49 + k], e));
    sc[c] = ldexpf(1.0f, -e) / 65280.0f;
  }
}

int ita_oracle_tokenizer_u8(const uint8_t* img, int B, int E, const float* cw, const float* cb, const float* lnw,
                            const float* lnb, float* tokens) {
  int32_t pb[49];
  float* pre = (float*)malloc(sizeof(float) * (size_t)E);
  int32_t* wq = (int32_t*)malloc(sizeof(int32_t) * (size_t)E * 49);
  float* sc = (float*)malloc(sizeof(float) * (size_t)E);
  ita_oracle_tok_quant_weights(cw, E, wq, sc);
  int rc = 0;
  for (int b = 0; b < B && !rc; ++b)
    for (int oy = 0; oy < TOK_H && !rc; ++oy)
      for (int ox = 0; ox < TOK_W && !rc; ++ox) {
        int y0, yp, x0, xp;
        float ly, lx;
        bilinear_src(oy, CONV_H, TOK_H, &y0, &yp, &ly);
        bilinear_src(ox, CONV_W, TOK_W, &x0, &xp, &lx);
        const float H1f = 8.0f * ly, W1f = 32.0f * lx;
        const int H1 = (int)H1f, W1 = (int)W1f, H0 = 8 - H1, W0 = 32 - W1;
        if ((float)H1 != H1f || (float)W1 != W1f) { rc = -3; break; }   /* the weights of this resize are dyadic */
        const uint8_t* im = img + (size_t)b * IMG_H * IMG_W;
        for (int ky = 0; ky < 7; ++ky)
          for (int kx = 0; kx < 7; ++kx) {
            const int iy = 2 * y0 - 3 + ky, ix = 2 * x0 - 3 + kx;
#define PX_(y, x) (((y) < 0 || (y) >= IMG_H || (x) < 0 || (x) >= IMG_W) ? 0 : (int)im[(y) * IMG_W + (x)])
            const int a = PX_(iy, ix), bb = PX_(iy, ix + 2 * xp), c = PX_(iy + 2 * yp, ix), d = PX_(iy + 2 * yp, ix + 2 * xp);
#undef PX_
            pb[ky * 7 + kx] = H0 * (W0 * a + W1 * bb) + H1 * (W0 * c + W1 * d);
          }
        for (int c = 0; c < E; ++c) {
          int32_t S0 = 0, S1 = 0, S2 = 0, S3 = 0;
          for (int k = 0; k < 49; ++k) {
            const int32_t a0 = pb[k] & 255, a1 = pb[k] >> 8;
            const int32_t W = wq[c * 49 + k];
            const int32_t w0 = ((W + 128) & 255) - 128, W1r = (W - w0) >> 8;
            const int32_t w1 = ((W1r + 128) & 255) - 128, w2 = (W1r - w1) >> 8;
            S0 += a0 * w0; S1 += a0 * w1 + a1 * w0; S2 += a0 * w2 + a1 * w1; S3 += a1 * w2;
          }
          const int32_t L = S0 + 256 * S1, Hh = S2 + 256 * S3;
          pre[c] = fmaf((float)Hh, 65536.0f * sc[c], fmaf((float)L, sc[c], cb[c]));
        }
        layernorm_row(pre, E, lnw, lnb, tokens + ((size_t)b * 128 + oy * TOK_W + ox) * E);
      }
  free(pre); free(wq); free(sc);
  return rc;
}

/* u8 wire frame -> f32, as the reference host does (samples/inference_udp_FPGA_custom_dispatch/
 * main.cpp:168-169: float(pixel) / 255.0f). */
void ita_oracle_u8_to_f32(const uint8_t* in, size_t n, float* out) {
  for (size_t i = 0; i < n; ++i) out[i] = (float)in[i] / 255.0f;
}

/* ------------------------------------------------------------------ int8 primitives */

/* torch.quantize_per_tensor(x, s, 0, qint8): rne(x * (1.0f/s)) clamped (layers.py:103, :63) */
void ita_oracle_quantize(const float* x, size_t n, float inv_scale, int8_t* q) {
  for (size_t i = 0; i < n; ++i) q[i] = (int8_t)clampf(rintf(x[i] * inv_scale), -128.0f, 127.0f);
}

static inline int8_t requant(int32_t acc, float mult) {
  return (int8_t)clampf(rintf((float)acc * mult), -128.0f, 127.0f);
}

/* nnq.Linear (qnnpack): y = clamp(rne(float(acc + bias_q) * mult)); optional ReLU on the int8
 * result (layers.py:66-68: ReLU is a separate module on the quantised tensor, zero_point 0). */
void ita_oracle_linear_q(const int8_t* x, int rows, int K, int N, const int8_t* w, const int32_t* bq, float mult,
                         int relu, int8_t* y) {
  for (int r = 0; r < rows; ++r)
    for (int n = 0; n < N; ++n) {
      int32_t acc = bq ? bq[n] : 0;
      const int8_t* xr = x + (size_t)r * K;
      const int8_t* wr = w + (size_t)n * K;
      for (int k = 0; k < K; ++k) acc += (int32_t)xr[k] * (int32_t)wr[k];
      int8_t v = requant(acc, mult);
      if (relu && v < 0) v = 0;
      y[(size_t)r * N + n] = v;
    }
}

/* matmul1: logits = requant(Q K^T)  -- NO 1/sqrt(d) (layers.py:115) */
void ita_oracle_matmul_qk(const int8_t* Q, const int8_t* K, int S, int P, float mult, int8_t* logits) {
  for (int i = 0; i < S; ++i)
    for (int j = 0; j < S; ++j) {
      int32_t acc = 0;
      for (int d = 0; d < P; ++d) acc += (int32_t)Q[(size_t)i * P + d] * (int32_t)K[(size_t)j * P + d];
      logits[(size_t)i * S + j] = requant(acc, mult);
    }
}

/* ITA integer softmax over the last dim (ITA_softmax.py:51-61): works on raw int8 codes,
 * output uint8 with scale 1/255.  eps_max = 32*8/256 = 1 => shift = diff. */
void ita_oracle_softmax(const int8_t* logits, int rows, int cols, uint8_t* probs) {
  for (int r = 0; r < rows; ++r) {
    const int8_t* x = logits + (size_t)r * cols;
    int32_t m = -128;
    for (int j = 0; j < cols; ++j) m = x[j] > m ? x[j] : m;
    int32_t sum = 0;
    for (int j = 0; j < cols; ++j) {
      const int32_t diff = m - (int32_t)x[j];
      const int32_t shift = (int32_t)floorf((float)diff * 1.0f + 0.5f);
      sum += shift > 31 ? 0 : (256 >> shift); /* torch: shifts >= 32 give 0 */
    }
    if (sum < 1) sum = 1;
    /* torch: `python_int / int32_tensor` is Tensor.__rtruediv__ = reciprocal(tensor) * int in
     * float32, then floor.  For every reachable sum (256..32768) this equals the exact integer
     * floor(255*2^16 / sum) (tests/test_oracle_golden.py::test_softmax_inverse_exhaustive). */
    const int32_t inv = (int32_t)floorf((1.0f / (float)sum) * 16711680.0f);
    for (int j = 0; j < cols; ++j) {
      const int32_t diff = m - (int32_t)x[j];
      const int32_t shift = (int32_t)floorf((float)diff * 1.0f + 0.5f);
      const int32_t num = shift > 31 ? 0 : (256 >> shift);
      probs[(size_t)r * cols + j] = (uint8_t)floorf((float)(num * inv) / 65536.0f);
    }
  }
}

int32_t ita_oracle_softmax_inverse(int32_t sum) { return (int32_t)floorf((1.0f / (float)sum) * 16711680.0f); }

/* matmul2: ctx = requant(A(u8) V(s8)) with the multiplier of the validation harness */
void ita_oracle_matmul_av(const uint8_t* A, const int8_t* V, int S, int P, float mult, int8_t* ctx) {
  for (int i = 0; i < S; ++i)
    for (int d = 0; d < P; ++d) {
      int32_t acc = 0;
      for (int j = 0; j < S; ++j) acc += (int32_t)A[(size_t)i * S + j] * (int32_t)V[(size_t)j * P + d];
      ctx[(size_t)i * P + d] = requant(acc, mult);
    }
}

/* ------------------------------------------------------------------ blocks */

/* ITASelfAttention_QAT.forward (H = 1).  x: (B,S,E) f32 -> out_f (B,S,E) f32.
 * Optional taps (may be NULL) receive the per-stage int tensors. */
int ita_oracle_mha(const float* x, int B, int S, int E, int P, const int8_t* wq, const int32_t* bq,
                   const int8_t* wk, const int32_t* bk, const int8_t* wv, const int32_t* bv, const int8_t* wo,
                   const int32_t* bo, const float* scal, float* out_f, int8_t* t_xq, int8_t* t_Q, int8_t* t_K,
                   int8_t* t_V, int8_t* t_logits, uint8_t* t_probs, int8_t* t_ctx, int8_t* t_out) {
  int8_t* xq = (int8_t*)malloc((size_t)S * E);
  int8_t* Q = (int8_t*)malloc((size_t)S * P);
  int8_t* K = (int8_t*)malloc((size_t)S * P);
  int8_t* V = (int8_t*)malloc((size_t)S * P);
  int8_t* L = (int8_t*)malloc((size_t)S * S);
  uint8_t* A = (uint8_t*)malloc((size_t)S * S);
  int8_t* C = (int8_t*)malloc((size_t)S * P);
  int8_t* O = (int8_t*)malloc((size_t)S * E);
  for (int b = 0; b < B; ++b) {
    ita_oracle_quantize(x + (size_t)b * S * E, (size_t)S * E, scal[ITA_A_INV_SX], xq);
    ita_oracle_linear_q(xq, S, E, P, wq, bq, scal[ITA_A_MQ], 0, Q);
    ita_oracle_linear_q(xq, S, E, P, wk, bk, scal[ITA_A_MK], 0, K);
    ita_oracle_linear_q(xq, S, E, P, wv, bv, scal[ITA_A_MV], 0, V);
    ita_oracle_matmul_qk(Q, K, S, P, scal[ITA_A_ML], L);
    ita_oracle_softmax(L, S, S, A);
    ita_oracle_matmul_av(A, V, S, P, scal[ITA_A_MC], C);
    ita_oracle_linear_q(C, S, P, E, wo, bo, scal[ITA_A_MO], 0, O);
    for (int i = 0; i < S * E; ++i) out_f[(size_t)b * S * E + i] = (float)O[i] * scal[ITA_A_SO];
    if (t_xq) memcpy(t_xq + (size_t)b * S * E, xq, (size_t)S * E);
    if (t_Q) memcpy(t_Q + (size_t)b * S * P, Q, (size_t)S * P);
    if (t_K) memcpy(t_K + (size_t)b * S * P, K, (size_t)S * P);
    if (t_V) memcpy(t_V + (size_t)b * S * P, V, (size_t)S * P);
    if (t_logits) memcpy(t_logits + (size_t)b * S * S, L, (size_t)S * S);
    if (t_probs) memcpy(t_probs + (size_t)b * S * S, A, (size_t)S * S);
    if (t_ctx) memcpy(t_ctx + (size_t)b * S * P, C, (size_t)S * P);
    if (t_out) memcpy(t_out + (size_t)b * S * E, O, (size_t)S * E);
  }
  free(xq); free(Q); free(K); free(V); free(L); free(A); free(C); free(O);
  return 0;
}

/* The same block at the accelerator's own boundary: int8 codes in (what attention_blocks.N.quant produces), out_proj's
 * int8 codes out (layers.py:106-123 without the QuantStub / DeQuantStub at :103,:125).  Lets a test hand ARBITRARY codes
 * to the int8-in GPU entry (ita_mha_q8) without a float round trip that may not reproduce them. */
int ita_oracle_mha_q8(const int8_t* xq, int B, int S, int E, int P, const int8_t* wq, const int32_t* bq,
                      const int8_t* wk, const int32_t* bk, const int8_t* wv, const int32_t* bv, const int8_t* wo,
                      const int32_t* bo, const float* scal, int8_t* out_q) {
  int8_t* Q = (int8_t*)malloc((size_t)S * P);
  int8_t* K = (int8_t*)malloc((size_t)S * P);
  int8_t* V = (int8_t*)malloc((size_t)S * P);
  int8_t* L = (int8_t*)malloc((size_t)S * S);
  uint8_t* A = (uint8_t*)malloc((size_t)S * S);
  int8_t* C = (int8_t*)malloc((size_t)S * P);
  for (int b = 0; b < B; ++b) {
    const int8_t* x = xq + (size_t)b * S * E;
    ita_oracle_linear_q(x, S, E, P, wq, bq, scal[ITA_A_MQ], 0, Q);
    ita_oracle_linear_q(x, S, E, P, wk, bk, scal[ITA_A_MK], 0, K);
    ita_oracle_linear_q(x, S, E, P, wv, bv, scal[ITA_A_MV], 0, V);
    ita_oracle_matmul_qk(Q, K, S, P, scal[ITA_A_ML], L);
    ita_oracle_softmax(L, S, S, A);
    ita_oracle_matmul_av(A, V, S, P, scal[ITA_A_MC], C);
    ita_oracle_linear_q(C, S, P, E, wo, bo, scal[ITA_A_MO], 0, out_q + (size_t)b * S * E);
  }
  free(Q); free(K); free(V); free(L); free(A); free(C);
  return 0;
}

/* Long sequences (BASELINE config 5 as worded: a 64x patch-token blow-up, S = 8192): the same block, but only the query rows a
 * test asks for -- K and V are projected for every token (O(S)), logits / softmax / A.V / out_proj only for `rows` (O(n S)),
 * so a sampled check of an S = 8192 frame takes a second instead of the half minute of the S x S form.  Same primitives, same
 * arithmetic: out_q[i] is row rows[i] of ita_oracle_mha_q8's result (tests/test_oracle_golden.py checks that on S = 256). */
int ita_oracle_mha_q8_rows(const int8_t* xq, int S, int E, int P, const int8_t* wq, const int32_t* bq, const int8_t* wk,
                           const int32_t* bk, const int8_t* wv, const int32_t* bv, const int8_t* wo, const int32_t* bo,
                           const float* scal, const int32_t* rows, int n, int8_t* out_q /* (n, E) */) {
  int8_t* K = (int8_t*)malloc((size_t)S * P);
  int8_t* V = (int8_t*)malloc((size_t)S * P);
  int8_t* Q = (int8_t*)malloc((size_t)P);
  int8_t* L = (int8_t*)malloc((size_t)S);
  uint8_t* A = (uint8_t*)malloc((size_t)S);
  int8_t* C = (int8_t*)malloc((size_t)P);
  ita_oracle_linear_q(xq, S, E, P, wk, bk, scal[ITA_A_MK], 0, K);
  ita_oracle_linear_q(xq, S, E, P, wv, bv, scal[ITA_A_MV], 0, V);
  for (int i = 0; i < n; ++i) {
    const int r = rows[i];
    if (r < 0 || r >= S) { free(K); free(V); free(Q); free(L); free(A); free(C); return -1; }
    ita_oracle_linear_q(xq + (size_t)r * E, 1, E, P, wq, bq, scal[ITA_A_MQ], 0, Q);
    for (int j = 0; j < S; ++j) {
      int32_t acc = 0;
      for (int d = 0; d < P; ++d) acc += (int32_t)Q[d] * (int32_t)K[(size_t)j * P + d];
      L[j] = requant(acc, scal[ITA_A_ML]);
    }
    ita_oracle_softmax(L, 1, S, A);
    for (int d = 0; d < P; ++d) {
      int32_t acc = 0;
      for (int j = 0; j < S; ++j) acc += (int32_t)A[j] * (int32_t)V[(size_t)j * P + d];
      C[d] = requant(acc, scal[ITA_A_MC]);
    }
    ita_oracle_linear_q(C, 1, P, E, wo, bo, scal[ITA_A_MO], 0, out_q + (size_t)i * E);
  }
  free(K); free(V); free(Q); free(L); free(A); free(C);
  return 0;
}

/* ITAFeedForward_QAT.forward */
int ita_oracle_ffn(const float* x, int B, int S, int E, int F, const int8_t* w1, const int32_t* b1,
                   const int8_t* w2, const int32_t* b2, const float* scal, float* out_f, int8_t* t_xq,
                   int8_t* t_h, int8_t* t_out) {
  int8_t* xq = (int8_t*)malloc((size_t)S * E);
  int8_t* Hh = (int8_t*)malloc((size_t)S * F);
  int8_t* O = (int8_t*)malloc((size_t)S * E);
  for (int b = 0; b < B; ++b) {
    ita_oracle_quantize(x + (size_t)b * S * E, (size_t)S * E, scal[ITA_F_INV_SX], xq);
    ita_oracle_linear_q(xq, S, E, F, w1, b1, scal[ITA_F_M1], 1, Hh);
    ita_oracle_linear_q(Hh, S, F, E, w2, b2, scal[ITA_F_M2], 0, O);
    for (int i = 0; i < S * E; ++i) out_f[(size_t)b * S * E + i] = (float)O[i] * scal[ITA_F_S2];
    if (t_xq) memcpy(t_xq + (size_t)b * S * E, xq, (size_t)S * E);
    if (t_h) memcpy(t_h + (size_t)b * S * F, Hh, (size_t)S * F);
    if (t_out) memcpy(t_out + (size_t)b * S * E, O, (size_t)S * E);
  }
  free(xq); free(Hh); free(O);
  return 0;
}

/* ------------------------------------------------------------------ fusion tail */
/* QAT/model.py:116-121: tokens -> (E,8,16) map; PixelShuffle(2) -> (E/4,16,32);
 * Upsample((16,32), bilinear, align_corners=True) -> (E,16,32); cat; Conv2d(5E/4 -> 9, k3, p1).
 * feat is the flattened (9,16,32) map (c*512 + y*32 + x), the decoder's input. */
static void upsample_src_ac(int dst, int in, int out, int* i0, int* ip, float* l1) {
  const float scale = (float)(in - 1) / (float)(out - 1);
  const float src = scale * (float)dst;
  int i = (int)src;
  if (i > in - 1) i = in - 1;
  *i0 = i;
  *ip = (i < in - 1) ? 1 : 0;
  *l1 = src - (float)i;
}

/* General form (BASELINE config 5): the same three layers on a tok_h x tok_w token
 * grid with out_ch conv outputs: x2 (B, tok_h*tok_w, E) -> out (B, out_ch, 2*tok_h, 2*tok_w). */
static float tail_fused_at(const float* xt, int E, int TH, int TW, int c, int y, int x) {
  const int C4 = E / 4, OH = 2 * TH, OW = 2 * TW;
  if (y < 0 || y >= OH || x < 0 || x >= OW) return 0.0f;
  if (c < C4) return xt[(size_t)((y >> 1) * TW + (x >> 1)) * E + (4 * c + 2 * (y & 1) + (x & 1))];
  c -= C4;
  int y0, yp, x0, xp; float ly, lx;
  upsample_src_ac(y, TH, OH, &y0, &yp, &ly);
  upsample_src_ac(x, TW, OW, &x0, &xp, &lx);
  const float h1 = ly, h0 = 1.0f - ly, w1 = lx, w0 = 1.0f - lx;
  const float v00 = xt[(size_t)(y0 * TW + x0) * E + c], v01 = xt[(size_t)(y0 * TW + x0 + xp) * E + c];
  const float v10 = xt[(size_t)((y0 + yp) * TW + x0) * E + c], v11 = xt[(size_t)((y0 + yp) * TW + x0 + xp) * E + c];
  return h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11);
}

void ita_oracle_tail_general(const float* x2, int B, int E, int TH, int TW, int CO, const float* cw, const float* cb,
                             float* out, float* t_fused /* optional (B, 5E/4, 2TH, 2TW) */) {
  const int C4 = E / 4, CIN = C4 + E, OH = 2 * TH, OW = 2 * TW;
  float* fused = (float*)malloc(sizeof(float) * (size_t)CIN * OH * OW);
  for (int b = 0; b < B; ++b) {
    const float* xt = x2 + (size_t)b * TH * TW * E; /* [token = h*TW+w][c] */
    for (int c = 0; c < CIN; ++c)
      for (int y = 0; y < OH; ++y)
        for (int x = 0; x < OW; ++x) fused[((size_t)c * OH + y) * OW + x] = tail_fused_at(xt, E, TH, TW, c, y, x);
    if (t_fused) memcpy(t_fused + (size_t)b * CIN * OH * OW, fused, sizeof(float) * (size_t)CIN * OH * OW);
    for (int o = 0; o < CO; ++o)
      for (int y = 0; y < OH; ++y)
        for (int x = 0; x < OW; ++x) {
          float acc = cb[o];
          for (int c = 0; c < CIN; ++c)
            for (int ky = 0; ky < 3; ++ky)
              for (int kx = 0; kx < 3; ++kx) {
                const int iy = y + ky - 1, ix = x + kx - 1;
                if (iy < 0 || iy >= OH || ix < 0 || ix >= OW) continue;
                acc = fmaf(fused[((size_t)c * OH + iy) * OW + ix], cw[(((size_t)o * CIN + c) * 3 + ky) * 3 + kx], acc);
              }
          out[(((size_t)b * CO + o) * OH + y) * OW + x] = acc;
        }
  }
  free(fused);
}

/* the same outputs at n sampled positions pts[i] = {b, o, y, x} (for checks at sizes where the full map
 * would take minutes on one core) */
void ita_oracle_tail_general_at(const float* x2, int E, int TH, int TW, int CO, const float* cw, const float* cb,
                                const int* pts, int n, float* vals) {
  const int CIN = E / 4 + E, OH = 2 * TH, OW = 2 * TW;
  (void)CO;
  for (int i = 0; i < n; ++i) {
    const int b = pts[4 * i], o = pts[4 * i + 1], y = pts[4 * i + 2], x = pts[4 * i + 3];
    const float* xt = x2 + (size_t)b * TH * TW * E;
    float acc = cb[o];
    for (int c = 0; c < CIN; ++c)
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
          const int iy = y + ky - 1, ix = x + kx - 1;
          if (iy < 0 || iy >= OH || ix < 0 || ix >= OW) continue;
          acc = fmaf(tail_fused_at(xt, E, TH, TW, c, iy, ix), cw[(((size_t)o * CIN + c) * 3 + ky) * 3 + kx], acc);
        }
    vals[i] = acc;
  }
}

void ita_oracle_tail(const float* x2, int B, int E, const float* cw, const float* cb, float* feat,
                     float* t_fused /* optional (B, 5E/4, 16, 32) */) {
  ita_oracle_tail_general(x2, B, E, TOK_H, TOK_W, 9, cw, cb, feat, t_fused);
}

/* ------------------------------------------------------------------ float linear / LSTM */

/* y = x W^T + b, one ascending-k fmaf chain per output, started from the bias */
void ita_oracle_linear_f32(const float* x, int rows, int K, int N, const float* w, const float* b, float* y) {
  for (int r = 0; r < rows; ++r)
    for (int n = 0; n < N; ++n) {
      float acc = b ? b[n] : 0.0f;
      const float* xr = x + (size_t)r * K;
      const float* wr = w + (size_t)n * K;
      for (int k = 0; k < K; ++k) acc = fmaf(xr[k], wr[k], acc);
      y[(size_t)r * N + n] = acc;
    }
}

/* one nn.LSTM layer, seq_len 1 (gate order i,f,g,o; QAT/model.py:128-129) */
void ita_oracle_lstm_cell(const float* x, int B, int In, const float* w_ih, const float* w_hh, const float* b_ih,
                          const float* b_hh, const float* h_in, const float* c_in, float* h_out, float* c_out) {
  const int Hd = 128;
  float g[512];
  for (int b = 0; b < B; ++b) {
    for (int j = 0; j < 4 * Hd; ++j) {
      float acc = b_ih[j] + b_hh[j];
      for (int k = 0; k < In; ++k) acc = fmaf(x[(size_t)b * In + k], w_ih[(size_t)j * In + k], acc);
      for (int k = 0; k < Hd; ++k) acc = fmaf(h_in[(size_t)b * Hd + k], w_hh[(size_t)j * Hd + k], acc);
      g[j] = acc;
    }
    for (int j = 0; j < Hd; ++j) {
      const float ig = sigmoidf_(g[j]), fg = sigmoidf_(g[Hd + j]), gg = tanhf_(g[2 * Hd + j]),
                  og = sigmoidf_(g[3 * Hd + j]);
      const float c = fmaf(fg, c_in[(size_t)b * Hd + j], ig * gg);
      c_out[(size_t)b * Hd + j] = c;
      h_out[(size_t)b * Hd + j] = og * tanhf_(c);
    }
  }
}

/* ------------------------------------------------------------------ full forward from a blob */

#define GET(name) ita_blob_data(blob, ita_blob_find(blob, nbytes, name))

/* module.main_graph(image, additional_data, quat_data, hidden_in_h, hidden_in_c)
 *   -> (output, hidden_out_h, hidden_out_c)        (tests/export_onnx_for_FPGA.py:71-80)
 * with a leading batch B.  image: f32 (B,60,90) or u8 wire frames.  h/c: (3,B,128).
 * Optional taps: tokens, x1, x2 (B,128,E), feat (B,4608), dec (B,512). */
int ita_oracle_forward(const void* blob, size_t nbytes, const void* image, int image_is_u8, const float* desvel,
                       const float* quat, const float* h_in, const float* c_in, int B, float* vel, float* h_out,
                       float* c_out, float* t_tokens, float* t_x1, float* t_x2, float* t_feat, float* t_dec) {
  const ita_blob_header* hd = (const ita_blob_header*)blob;
  if (nbytes < sizeof(*hd) || memcmp(hd->magic, ITA_BLOB_MAGIC, 8) != 0) return -1;
  const int E = hd->E, S = hd->S, P = hd->P, F = hd->F, L = hd->num_layers;
  if (S != 128 || hd->H != 1) return -2;
  if (!GET("dec.w") || !GET("lstm.w_ih0") || !GET("fc.w")) return -2;   /* a blocks-only blob has no whole graph */
  const int has_tail = hd->has_tail;   /* 0: models/ITA/QAT/model.py:80-81 -- the decoder reads the flattened tokens */
  const size_t tokn = (size_t)B * S * E;
  float* img = (float*)malloc(sizeof(float) * (size_t)B * IMG_H * IMG_W);
  if (!image_is_u8) memcpy(img, image, sizeof(float) * (size_t)B * IMG_H * IMG_W);
  float* x = (float*)malloc(sizeof(float) * tokn);
  float* y = (float*)malloc(sizeof(float) * tokn);
  float* feat = (float*)malloc(sizeof(float) * (size_t)B * 4608);
  float* cat = (float*)malloc(sizeof(float) * (size_t)B * 517);
  float* dec = (float*)malloc(sizeof(float) * (size_t)B * 512);
  int rc = 0;
  if (image_is_u8)
    rc = ita_oracle_tokenizer_u8((const uint8_t*)image, B, E, (const float*)GET("tok.conv_w"), (const float*)GET("tok.conv_b"),
                                 (const float*)GET("tok.ln_w"), (const float*)GET("tok.ln_b"), x);
  else
    ita_oracle_tokenizer(img, B, E, (const float*)GET("tok.conv_w"), (const float*)GET("tok.conv_b"),
                         (const float*)GET("tok.ln_w"), (const float*)GET("tok.ln_b"), x);
  if (t_tokens) memcpy(t_tokens, x, sizeof(float) * tokn);
  char nm[32];
  for (int i = 0; i < L; ++i) {
#define N_(fmt) (snprintf(nm, sizeof nm, fmt, i), GET(nm))
    ita_oracle_mha(x, B, S, E, P, (const int8_t*)N_("attn%d.wq"), (const int32_t*)N_("attn%d.bq"),
                   (const int8_t*)N_("attn%d.wk"), (const int32_t*)N_("attn%d.bk"), (const int8_t*)N_("attn%d.wv"),
                   (const int32_t*)N_("attn%d.bv"), (const int8_t*)N_("attn%d.wo"), (const int32_t*)N_("attn%d.bo"),
                   (const float*)N_("attn%d.scal"), y, 0, 0, 0, 0, 0, 0, 0, 0);
    ita_oracle_add_ln(x, y, B * S, E, (const float*)N_("norm1_%d.w"), (const float*)N_("norm1_%d.b"), x);
    if (t_x1 && i == L - 1) memcpy(t_x1, x, sizeof(float) * tokn);
    ita_oracle_ffn(x, B, S, E, F, (const int8_t*)N_("ffn%d.w1"), (const int32_t*)N_("ffn%d.b1"),
                   (const int8_t*)N_("ffn%d.w2"), (const int32_t*)N_("ffn%d.b2"), (const float*)N_("ffn%d.scal"), y,
                   0, 0, 0);
    ita_oracle_add_ln(x, y, B * S, E, (const float*)N_("norm2_%d.w"), (const float*)N_("norm2_%d.b"), x);
#undef N_
  }
  if (t_x2) memcpy(t_x2, x, sizeof(float) * tokn);
  if (has_tail) {
    ita_oracle_tail(x, B, E, (const float*)GET("tail.conv_w"), (const float*)GET("tail.conv_b"), feat, 0);
    if (t_feat) memcpy(t_feat, feat, sizeof(float) * (size_t)B * 4608);
    ita_oracle_linear_f32(feat, B, 4608, 512, (const float*)GET("dec.w"), (const float*)GET("dec.b"), dec);
  } else {   /* x.flatten(1): token-major, channel fastest -- the (B,128,E) layout as it stands */
    ita_oracle_linear_f32(x, B, S * E, 512, (const float*)GET("dec.w"), (const float*)GET("dec.b"), dec);
  }
  if (t_dec) memcpy(t_dec, dec, sizeof(float) * (size_t)B * 512);
  for (int b = 0; b < B; ++b) {
    memcpy(cat + (size_t)b * 517, dec + (size_t)b * 512, sizeof(float) * 512);
    cat[(size_t)b * 517 + 512] = desvel[b] / 10.0f;
    for (int k = 0; k < 4; ++k) cat[(size_t)b * 517 + 513 + k] = quat[(size_t)b * 4 + k];
  }
  const float* lin = cat;
  int In = 517;
  for (int l = 0; l < 3; ++l) {
    char a[32], bn[32], c[32], d[32];
    snprintf(a, sizeof a, "lstm.w_ih%d", l); snprintf(bn, sizeof bn, "lstm.w_hh%d", l);
    snprintf(c, sizeof c, "lstm.b_ih%d", l); snprintf(d, sizeof d, "lstm.b_hh%d", l);
    ita_oracle_lstm_cell(lin, B, In, (const float*)GET(a), (const float*)GET(bn), (const float*)GET(c),
                         (const float*)GET(d), h_in + (size_t)l * B * 128, c_in + (size_t)l * B * 128,
                         h_out + (size_t)l * B * 128, c_out + (size_t)l * B * 128);
    lin = h_out + (size_t)l * B * 128;
    In = 128;
  }
  ita_oracle_linear_f32(lin, B, 128, 3, (const float*)GET("fc.w"), (const float*)GET("fc.b"), vel);
  free(img); free(x); free(y); free(feat); free(cat); free(dec);
  return rc;
}

/* ------------------------------------------------------------------ UDP host (row n1) */
/* restated from samples/inference_udp_FPGA_custom_dispatch/main.cpp:320-354 (unpack_frame, incl. its
 * quaternion stride of sizeof(size_t) when ref_bug != 0) and :381-417 (calculate_final_velocity) */
static float be_f32(const unsigned char* p) {
  unsigned char b[4] = {p[3], p[2], p[1], p[0]};   /* swap_endian_4 then reinterpret on a little-endian host */
  float f;
  memcpy(&f, b, 4);
  return f;
}
int ita_oracle_unpack_packet(const unsigned char* packet, size_t nbytes, int ref_bug, float* out6) {
  if (nbytes < 5424) return -1;
  size_t offset = 5400;
  out6[0] = be_f32(packet + offset); offset += 4;
  out6[1] = be_f32(packet + offset); offset += 4;
  for (int i = 0; i < 4; ++i) {
    unsigned char q[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; ++k) if (offset + k < nbytes) q[k] = packet[offset + k];
    out6[2 + i] = be_f32(q);
    offset += ref_bug ? sizeof(size_t) : 4;
  }
  return 0;
}
void ita_oracle_final_velocity(const float* raw, float desired_vel, float pos_x, float* out) {
  float fv[3] = {raw[0], raw[1], raw[2]};
  fv[0] = fminf(fmaxf(fv[0], -1.0f), 1.0f);
  const float norm = sqrtf(fv[0] * fv[0] + fv[1] * fv[1] + fv[2] * fv[2]);
  if (norm > 0.0f) { fv[0] /= norm; fv[1] /= norm; fv[2] /= norm; }
  fv[0] *= desired_vel; fv[1] *= desired_vel; fv[2] *= desired_vel;
  if (pos_x < 2.0f) fv[0] = fmaxf(1.0f, (pos_x / 2.0f) * desired_vel);
  out[0] = fv[0]; out[1] = fv[1]; out[2] = fv[2];
}
