// ita_long_attn_kernel.h -- ITASelfAttention_QAT on LONG token sequences (S a multiple of 128, E = 128, P = 192, H = 1).
//
// BASELINE config 5 as it is worded -- a 480 x 720 input, a 64x patch-token blow-up -- means S = 8192 tokens through the
// attention block (reference models/ITA/QAT/layers.py:101-127, models/ITA/QAT/ITA_softmax.py:51-61).  A row of 8192 logits
// cannot stay in registers and the integer softmax is NOT associative the way exp-softmax is: the shift of every element
// depends on the row's GLOBAL maximum, and the probability floor((256 >> shift) * inv / 2^16) on the row's global sum, so
// neither partial maxima nor partial sums of one sweep can be repaired afterwards ("Config 5").  Three
// sweeps over the key tiles per query tile, each recomputing Q K^T on the matrix cores (the logits are never written out):
//     sweep 1: row maximum                     (of the RAW accumulators: requantisation is monotonic, one value per row is requantised)
//     sweep 2: row sum of 256 >> (max - x)      (per key tile in 16-bit pairs, accumulated in 32 bits)
//     sweep 3: probabilities -> A.V            (exactly ita_softmax_packed16's last step + the stream kernel's A.V)
// Same arithmetic as the S = 128 stream kernel and the oracle (oracle/ita_oracle.c: ita_oracle_mha_q8 / _rows): int8 codes in,
// out_proj's int8 codes out.
//
//   ita_long_proj_kernel : persistent, one 128-token tile at a time -- the stream kernel's projection phase (Q stays in
//                          registers as MFMA B fragments, K and V^T are built as LDS images) and then the images go to
//                          global memory verbatim: Q fragments, a 24 KB K image and a 24 KB V^T image + 192 column sums per tile
//   ita_long_attn_kernel : one workgroup per (query tile, frame), two workgroups per CU; wave w owns 16 queries; K tiles arrive
//                          by LDS-DMA into a two-slot ring (one barrier per key tile), V^T tiles (sweep 3) into one slot (a second
//                          barrier per step); context requantised, out_proj with its fragments from the layer image in L2,
//                          codes stored 16 bytes per lane
#pragma once
#include "ita_stream_kernel.h"

struct ItaLongArgs {
  const char* image;      // the layer's attention LDS image (ItaStreamLds<128, false, false>::IMAGE bytes)
  const int8_t* xq;       // (B, S, 128) block input codes
  int8_t* yq;             // (B, S, 128) out_proj codes
  char* qfrag;            // workspace (B, S/16, 3, 64 lanes, 16 B): Q as B fragments of the logits MFMA
  char* kimg;             // workspace (B, S/128, 24576): K images [12][128][16]
  char* vimg;             // workspace (B, S/128, 24576): V^T images [8][192][16], keys permuted as in the stream kernel
  int* csum;              // workspace (B, S/128, 192): 128 * column sums of V per key tile
  float mq, mk, mv, ml, mc, mo;
  int B, S;
};

template <bool FAST>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void ita_long_proj_kernel(const ItaLongArgs a) {
  constexpr int E = 128, S = 128, P = 192, NK = 2;
  using L = ItaStreamLds<E, false, false>;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 15, kq = lane >> 4, token = wave * 16 + qi;
  const int* bias = (const int*)(lds + L::BIAS);
  const int *l_bq = bias, *l_bk = bias + P;
  int* colsum = (int*)(lds + L::COLSUM);
  for (int p = tid; p < L::IMAGE / 16; p += 512) *(i32x4*)(lds + p * 16) = *(const i32x4*)(a.image + (size_t)p * 16);
  if (tid < P) colsum[tid] = 0;
  __syncthreads();
  const int ntile = a.B * (a.S / 128);
  for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int8_t* xrow = a.xq + ((size_t)tile * S + token) * E + 32 * kq;
    i32x4 xf[NK];
#pragma unroll
    for (int c = 0; c < NK; ++c) xf[c] = *(const i32x4*)(xrow + 16 * c);
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
          qf[gg] = rq_group(acc, a.mq, FAST);
        } else if (mat == 1) {
          *(i32x4*)(lds + L::K + (((4 * gg + kq) * S + token) << 4)) = rq_group(acc, a.mk, FAST);
        } else {
          const i32x4 p4 = rq_group(acc, a.mv, FAST);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int d = (4 * gg + t) * 16 + qi;
            *(int*)(lds + L::VT + (((((wave >> 2) * 4 + kq) * P) + d) << 4) + 4 * (wave & 3)) = p4[t];
            atomicAdd(&colsum[d], __builtin_amdgcn_sdot4(p4[t], 0x01010101, 0, false) << 7);
          }
        }
      };
      load(0, fr[0], ac[0]);
#pragma unroll
      for (int g = 0; g < 9; ++g) {
        if (g + 1 < 9) load(g + 1, fr[(g + 1) & 1], ac[(g + 1) % 3]);
        if (g < 6) mm_group<NK, false>(fr[g & 1], xf, ac[g % 3]);
        else mm_group<NK, true>(fr[g & 1], xf, ac[g % 3]);
        if (g > 0) epilogue(g - 1, ac[(g - 1) % 3]);
        ITA_SCHED_BARRIER();
      }
      epilogue(8, ac[8 % 3]);
    }
    // Q: this wave's 16 tokens as three B fragments (one per 64-feature k-step), 1 KB per wave and k-step
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) *(i32x4*)(a.qfrag + ((((size_t)tile * 8 + wave) * 3 + ks) * 64 + lane) * 16) = qf[ks];
    __syncthreads();     // K, V^T and the column sums of this tile are complete
    for (int p = tid; p < 2 * S * P / 16; p += 512) {
      const i32x4 v = *(const i32x4*)(lds + L::K + p * 16);      // (K and V^T are adjacent in the LDS layout)
      char* dst = p < S * P / 16 ? a.kimg + (size_t)tile * S * P + (size_t)p * 16 : a.vimg + (size_t)tile * S * P + (size_t)(p - S * P / 16) * 16;
      *(i32x4*)dst = v;
    }
    if (tid < P) a.csum[(size_t)tile * P + tid] = colsum[tid];
    __syncthreads();     // images copied: the next tile may overwrite them
    if (tid < P) colsum[tid] = 0;
  }
}

// 73 KB of LDS and 128 registers: two workgroups per CU, four waves per SIMD -- this kernel is a stream of dependent MFMA -> requantise
// -> softmax steps and at two waves per SIMD 30 % of the issue slots went unused.  K tiles: two-slot ring; V^T tiles (sweep 3 only): ONE
// slot, staged at the start of a step and waited for behind that step's Q K^T and softmax; out_proj's fragments come straight from the
// layer image in L2 (once per workgroup).
struct ItaLongLds {
  static constexpr int KB = 128 * 192;                 // bytes per K or V^T image
  static constexpr int K = 0, V = 2 * KB;              // K: two-slot ring; V^T: one slot
  static constexpr int BO = 3 * KB;                    // int32 [128]: bo + ITA_ACC_BIAS
  static constexpr int CS = BO + 128 * 4;              // int32 [192]: 128 * column sums of V over the whole sequence
  static constexpr int TOTAL = CS + 192 * 4;
  static_assert(2 * TOTAL <= 160 * 1024, "two workgroups per CU");
};

template <bool FAST>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void ita_long_attn_kernel(const ItaLongArgs a) {
  constexpr int E = 128, S = 128, P = 192;
  using LI = ItaStreamLds<E, false, false>;   // offsets inside the layer's image
  using L = ItaLongLds;
  typedef ita_u16x2 u16x2;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 15, kq = lane >> 4;
  const int nkt = a.S / 128, qt = blockIdx.x, b = blockIdx.y;
  const size_t tile0 = (size_t)b * nkt;              // first tile index of this frame
  const char* kimg = a.kimg + tile0 * L::KB;
  const char* vimg = a.vimg + tile0 * L::KB;

  auto stage = [&](const char* src, int ldsoff) {    // 24 KB by LDS-DMA: three 1 KB pieces per wave
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int piece = wave + 8 * q;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(lds + ldsoff + piece * 1024), 16, 0, 0);
    }
  };
  stage(kimg, L::K);
  // out_proj fragments and bias from the layer's image; column sums of V over all key tiles of the frame
  if (tid < E / 4) *(i32x4*)(lds + L::BO + tid * 16) = *(const i32x4*)(a.image + LI::BIAS + (3 * P) * 4 + (size_t)tid * 16);
  if (tid < P) {
    int s = 0;
    for (int t = 0; t < nkt; ++t) s += a.csum[(tile0 + t) * P + tid];
    ((int*)(lds + L::CS))[tid] = s;
  }
  i32x4 qf[3];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) qf[ks] = *(const i32x4*)(a.qfrag + ((((tile0 + qt) * 8 + wave) * 3 + ks) * 64 + lane) * 16);

  // logits of this wave's 16 queries against one 128-key tile, as biased u16 pairs (ita_stream_kernel's attention core)
  auto logits = [&](const char* kb, u16x2 (&w)[16]) {
    struct KFr { i32x4 w[6]; } kfr[2];
    i32x4 la[2][2];
    auto ldk = [&](int p, KFr& f) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) f.w[t * 3 + ks] = *(const i32x4*)(kb + (((4 * ks + kq) * S + (2 * p + t) * 16 + qi) << 4));
    };
    auto mmk = [&](const KFr& f, i32x4 (&acc)[2]) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[t] = (i32x4){ITA_ACC_BIAS, ITA_ACC_BIAS, ITA_ACC_BIAS, ITA_ACC_BIAS};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f.w[t * 3 + ks], qf[ks], acc[t], 0, 0, 0);
      }
    };
    auto epil = [&](int p, const i32x4 (&acc)[2]) {
      unsigned w4[4];
      if constexpr (FAST) lg8_v3<ITA_RQ_FAST>(acc, a.ml, w4);
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
      mmk(kfr[p & 1], la[p & 1]);
      if (p > 0) epil(p - 1, la[(p - 1) & 1]);
      ITA_SCHED_BARRIER();
    }
    epil(3, la[1]);
  };
  // the same products, reduced to the running maximum of this lane's accumulators (all of them belong to query qi)
  auto logits_max = [&](const char* kb, int& mx) {
    struct KFr { i32x4 w[6]; } kfr[2];
    i32x4 la[2][2];
    auto ldk = [&](int p, KFr& f) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) f.w[t * 3 + ks] = *(const i32x4*)(kb + (((4 * ks + kq) * S + (2 * p + t) * 16 + qi) << 4));
    };
    auto mmk = [&](const KFr& f, i32x4 (&acc)[2]) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[t] = (i32x4){ITA_ACC_BIAS, ITA_ACC_BIAS, ITA_ACC_BIAS, ITA_ACC_BIAS};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f.w[t * 3 + ks], qf[ks], acc[t], 0, 0, 0);
      }
    };
    auto epil = [&](const i32x4 (&acc)[2]) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        mx = max(max(mx, acc[t][0]), acc[t][1]);           // (v_max3_i32)
        mx = max(max(mx, acc[t][2]), acc[t][3]);
      }
    };
    ldk(0, kfr[0]);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (p + 1 < 4) ldk(p + 1, kfr[(p + 1) & 1]);
      mmk(kfr[p & 1], la[p & 1]);
      if (p > 0) epil(la[(p - 1) & 1]);
      ITA_SCHED_BARRIER();
    }
    epil(la[1]);
  };
  // one step of sweeps 1 and 2: steps are counted through the sweeps (step = sweep * nkt + kt), the K tile of a step sits in
  // ring slot step & 1 with its DMA waited for here; the next step's tile goes out after the barrier (the last step of a
  // sweep prefetches tile 0 for the next sweep)
  auto next_tile = [&](int step, int kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();     // this step's tile landed for every wave; every wave is done with the previous step's slot
    const int nx = kt + 1 < nkt ? kt + 1 : 0;
    stage(kimg + (size_t)nx * L::KB, L::K + ((step + 1) & 1) * L::KB);
  };

  // ---------------- sweep 1: row maximum.  Requantisation is monotonic (m > 0; multiply, round and the biased 16-bit form all keep
  // the order), so max_k rq(acc_k) = rq(max_k acc_k): the sweep keeps the maximum of the raw (biased) accumulators -- one v_max3_i32
  // per two logits instead of the 2.25 to 2.75 instructions per logit of the requantisation -- and requantises ONE value per row
  int mx = (int)0x80000000;
  for (int kt = 0; kt < nkt; ++kt) {
    next_tile(kt, kt);
    logits_max(lds + L::K + (kt & 1) * L::KB, mx);
  }
  mx = max1632_i(mx);
  int m;
  {
    const i32x4 am[2] = {(i32x4){mx, mx, mx, mx}, (i32x4){mx, mx, mx, mx}};
    unsigned w4[4];
    if constexpr (FAST) lg8_v3<ITA_RQ_FAST>(am, a.ml, w4);
    else lg8_v3<ITA_RQ_EXACT>(am, a.ml, w4);
    m = (int)(w4[0] & 0xffffu);
  }
  m = min(max(m, 32768 - 128), 32768 + 127);
  const u16x2 mm = {(unsigned short)m, (unsigned short)m};
  const unsigned short capv = (unsigned short)min(m - (32768 - 128), 15);
  const u16x2 cap = {capv, capv}, one = {256, 256};
  // ---------------- sweep 2: row sum (nkt is even or odd: the ring slot of tile kt is (sweep * nkt + kt) & 1)
  int sumlo = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const int step = nkt + kt;
    next_tile(step, kt);
    u16x2 w[16];
    logits(lds + L::K + (step & 1) * L::KB, w);
    u16x2 s2 = {0, 0};
#pragma unroll
    for (int j = 0; j < 16; ++j) s2 += one >> __builtin_elementwise_min(__builtin_elementwise_sub_sat(mm, w[j]), cap);
    sumlo += (int)s2.x + (int)s2.y;       // <= 32 * 256 per tile and lane
  }
  int sum = sum1632_i(sumlo);
  sum = max(sum, 1);
  const int inv_hi = ((int)floorf((1.0f / (float)sum) * 16711680.0f)) >> 8;
  const u16x2 iv = {(unsigned short)inv_hi, (unsigned short)inv_hi};
  // ---------------- sweep 3: probabilities and A.V; the accumulators start from 128 * column sums (u8 probabilities ride the
  // signed MFMA as p - 128, as in the stream kernel)
  i32x4 va[3][4];
#pragma unroll
  for (int dg = 0; dg < 3; ++dg)
#pragma unroll
    for (int t = 0; t < 4; ++t) va[dg][t] = *(const i32x4*)(lds + L::CS + ((4 * dg + t) * 16 + 4 * kq) * 4);
  for (int kt = 0; kt < nkt; ++kt) {
    const int step = 2 * nkt + kt;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();     // K tile kt landed; every wave is done with the previous step: the V^T slot and the other K slot are free
    stage(vimg + (size_t)kt * L::KB, L::V);
    if (kt + 1 < nkt) stage(kimg + (size_t)(kt + 1) * L::KB, L::K + ((step + 1) & 1) * L::KB);
    u16x2 w[16];
    logits(lds + L::K + (step & 1) * L::KB, w);
    i32x4 pf[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int k4 = 4 * kb + t;
        const u16x2 s0 = __builtin_elementwise_min(__builtin_elementwise_sub_sat(mm, w[2 * k4]), cap);
        const u16x2 s1 = __builtin_elementwise_min(__builtin_elementwise_sub_sat(mm, w[2 * k4 + 1]), cap);
        const unsigned p01 = __builtin_bit_cast(unsigned, (u16x2)(iv >> s0));
        const unsigned p23 = __builtin_bit_cast(unsigned, (u16x2)(iv >> s1));
        pf[kb][t] = (int)(__builtin_amdgcn_perm(p23, p01, 0x06040200u) ^ 0x80808080u);
      }
    // V^T tile kt (requested first: the three K pieces of the next step may still be on their way)
    if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const char* vb = lds + L::V;
#pragma unroll
    for (int st = 0; st < 6; ++st) {
      const int dg = st >> 1, kb = st & 1;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const i32x4 vf = *(const i32x4*)(vb + ((((kb * 4 + kq) * P) + (4 * dg + t) * 16 + qi) << 4));
        va[dg][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(vf, pf[kb], va[dg][t], 0, 0, 0);
      }
    }
  }
  // ---------------- context requantisation (plain int32 sums: |sum p v| <= 255 * 128), out_proj, codes out
  i32x4 cf[3];
#pragma unroll
  for (int dg = 0; dg < 3; ++dg) {
    float f[16];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) f[4 * t + i] = __builtin_amdgcn_fmed3f((float)va[dg][t][i] * a.mc, -128.0f, 127.0f);
    unsigned pk[4];
    round_pack16(f, pk);
    cf[dg] = (i32x4){(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]};
  }
  __syncthreads();      // (BO was written in the prologue: visible since the first barrier; keeps the waves together for the stores)
  const int token = wave * 16 + qi;
#pragma unroll
  for (int eg = 0; eg < 2; ++eg) {
    i32x4 oa[4];
    ld_obias<E>(oa, (const int*)(lds + L::BO), 4 * eg, kq);
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      ItaF4 ob;
      ld_frg_ks<E>(ob, a.image + LI::WO, 4 * eg, ks, qi, kq);      // (global: 48 KB per workgroup from L2, once)
      mm_ks(ob, cf[ks], oa);
    }
    *(i32x4*)(a.yq + (((size_t)b * a.S + (size_t)qt * 128 + token) * E) + 32 * kq + 16 * eg) = rq_group(oa, a.mo, FAST);
  }
}
