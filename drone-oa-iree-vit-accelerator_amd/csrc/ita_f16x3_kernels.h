// ita_f16x3_kernels.h -- the float tail (fusion conv + decoder, LSTM) on f16 MFMA with
// split-precision operands.
//
// The reference computes this tail in f16 (its .vmfb is compiled with
// --iree-input-demote-f32-to-f16, docs/HOW-TO-compile-onnx-mlir-model.md:38-48); the task's
// tolerance for it is 1e-4 against the f32 graph.  A single f16 product does not meet that
// (K = 4608..8192 sums), so every operand is split x = hi + lo with hi = f16(x),
// lo = f16(x - hi) and three MFMA products are accumulated in f32:
//     x.w ~= hi_x.hi_w + lo_x.hi_w + hi_x.lo_w          (dropped lo.lo term ~ 2^-22 relative)
// which keeps ~22 significant bits at 16/3 the rate of f32 MFMA.  Weights are pre-scaled by a
// power of two so that their lo parts stay in f16's normal range; consumers undo the scale.
//
// PixelShuffle / bilinear Upsample / concat / conv3x3 (QAT/model.py:116-121) and the decoder
// Linear (:124) have no non-linearity between them, so they are folded at weight-load time
// into ONE matrix  Wfold[512][8192]  (built on the device by pushing unit impulses through the
// exact f32 tail + decoder kernels) and the whole tail becomes  dec = x2 . Wfold^T + bias'.
//
//   ita_gemm_f16x3_kernel<BM,BN>   C_partial[z][m][n] = sum_{k in slice z} A[m][k] W[n][k]
//   ita_lstm0_kernel               LSTM layer 0: sums the split-K partials (the decoder is folded into
//                                  its input projection), adds the [h | desvel | quat] remainder, cell update
//   ita_lstm_layer_kernel<NK>      LSTM layers 1, 2: [x | h] GEMM fused with the cell update
#pragma once
#include <type_traits>
#include "ita_device.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void split_f16(float x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}

struct ItaGemmSplitArgs {
  const _Float16 *a_hi, *a_lo; int lda;   // activations [M][lda], k contiguous
  const _Float16 *w_hi, *w_lo; int ldw;   // weights     [N][ldw]
  float* out;                             // [nsplit][M][N] raw f32 accumulators
  int M, N, K;                            // K % (64 * nsplit) == 0, N % BN == 0
  int nsplit;
  int dbg;                                // diagnostic: 1 = stage only (no MFMA), 2 = MFMA only (stage once), 4 no barrier, 8 one buffer, 16/32/64 L2-resident staging
  const _Float16 *wf_hi, *wf_lo;          // the same weights as MFMA A... B fragments, [N / 32][K / 16][lane 64][8] (tiny kernel only)
};

// LDS image of one operand plane for one 64-deep K tile: row-major 128-byte rows, the eight
// 16-byte chunks of a row XOR-swizzled with (row>>1)&7 so that the 32x32x16 fragment reads
// (lane -> row, fixed chunk) are bank-conflict free.  Tiles are filled by LDS-DMA
// (global_load_lds_dwordx4): LDS destination lane-linear, swizzle applied to the SOURCE chunk.
template <int ROWS, int NT>
__device__ __forceinline__ void stage_plane(const _Float16* __restrict__ g, int ld, int row0, int max_row, int k0,
                                            char* lds_plane, int tid) {
  static_assert(ROWS * 8 % NT == 0, "pieces must divide evenly over the workgroup");
#pragma unroll
  for (int p = 0; p < ROWS * 8 / NT; ++p) {
    const int piece = p * NT + tid;             // 16-byte piece index = LDS slot (lane-linear per wave)
    const int r = piece >> 3, s = piece & 7;
    const int c = s ^ ((r >> 1) & 7);
    const int gr = min(row0 + r, max_row);
    const _Float16* src = g + (size_t)gr * ld + k0 + c * 8;
    char* dst = lds_plane + (p * NT + (tid & ~63)) * 16;    // wave-uniform base; HW adds lane*16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  }
}

__device__ __forceinline__ f16x8 frag_f16(const char* lds_plane, int r, int chunk) {
  return *(const f16x8*)(lds_plane + r * 128 + ((chunk ^ ((r >> 1) & 7)) << 4));
}

template <int BM, int BN>
struct ItaGemmSplitLds {
  static constexpr int A_PLANE = BM * 128, W_PLANE = BN * 128;
  static constexpr int BUF = 2 * A_PLANE + 2 * W_PLANE;
  static constexpr int TOTAL = 2 * BUF;
};

// WM x WN waves; each wave owns a (BM/WM) x (BN/WN) block of 32x32 MFMA tiles.  Two waves per SIMD
// (WM*WN = 8) let one wave's MFMAs run while its partner waits for the LDS-DMA of the next K tile.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void ita_gemm_f16x3_kernel(const ItaGemmSplitArgs g) {
  using L = ItaGemmSplitLds<BM, BN>;
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);   // 32x32 MFMA tiles per wave per dim
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // 1-D grid, XCD-aware: consecutive workgroup ids are dealt round-robin over the 8 XCDs, so
  // id % nsplit picks the K slice -> with nsplit = 8 every XCD works on ONE K slice and its private
  // L2 holds that slice of A and W (6 MB) instead of seeing all 50 MB.  Placement only affects speed.
  const int bid = blockIdx.x;
  const int zsplit = bid % g.nsplit, tile = bid / g.nsplit;
  const int ntn = g.N / BN;
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
  const int kslice = g.K / g.nsplit, kbeg = zsplit * kslice;
  const int nt = kslice / 64;
  const int r = lane & 31, h = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  auto stage = [&](int buf, int t) {
    char* b = lds + buf * L::BUF;
    const int k0 = kbeg + t * 64;
    // diagnostic masks 16 / 32 / 64: every workgroup stages M tile 0 / N tile 0 / K slice 0 (everything L2-resident)
    const int m0s = (g.dbg & 16) ? 0 : m0, n0s = (g.dbg & 32) ? 0 : n0, k0s = (g.dbg & 64) ? t * 64 : k0;
    stage_plane<BM, NT>(g.a_hi, g.lda, m0s, g.M - 1, k0s, b, tid);
    stage_plane<BM, NT>(g.a_lo, g.lda, m0s, g.M - 1, k0s, b + L::A_PLANE, tid);
    stage_plane<BN, NT>(g.w_hi, g.ldw, n0s, g.N - 1, k0s, b + 2 * L::A_PLANE, tid);
    stage_plane<BN, NT>(g.w_lo, g.ldw, n0s, g.N - 1, k0s, b + 2 * L::A_PLANE + L::W_PLANE, tid);
  };

  auto compute = [&](int buf) {
    const char* b = lds + buf * L::BUF;
    const char *ah = b, *al = b + L::A_PLANE, *wh = b + 2 * L::A_PLANE, *wl = wh + L::W_PLANE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      f16x8 fah[TM], fal[TM], fwh[TN], fwl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * (BM / WM) + i * 32 + r;
        fah[i] = frag_f16(ah, row, 2 * ks + h);
        fal[i] = frag_f16(al, row, 2 * ks + h);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * (BN / WN) + j * 32 + r;
        fwh[j] = frag_f16(wh, row, 2 * ks + h);
        fwl[j] = frag_f16(wl, row, 2 * ks + h);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[i], fwh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[i], fwl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[i], fwh[j], acc[i][j], 0, 0, 0);
        }
    }
  };
  int cur = 0;
  stage(0, 0);
  __syncthreads();   // hipcc drains the LDS-DMA (vmcnt(0)) ahead of the barrier
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt && !(g.dbg & 2)) stage(cur ^ 1, t + 1);
    if (!(g.dbg & 1)) compute(g.dbg & 8 ? 0 : cur);
    if (!(g.dbg & 4)) __syncthreads();
    cur ^= 1;
  }
  // C layout: col n = lane&31, row m = (e&3) + 8*(e>>2) + 4*h
  float* out = g.out + (size_t)zsplit * g.M * g.N;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / WN) + j * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m < g.M) out[(size_t)m * g.N + n] = acc[i][j][e];
      }
    }
}

// ---- the same GEMM for 32 < M <= 256.  ita_gemm_f16x3_kernel needs M >= 1024 to fill the chip (at 256 frames it has 64
// workgroups).  Here a workgroup of four waves owns MT (1..4) 32-row M tiles x one 32-column
// W tile x one K slice: wave w computes M tile w, ALL four waves issue the LDS-DMA of the next k-tiles (an LDS ring
// of three 64-deep k-tiles).  grid.x = N/32 * nsplit with id % nsplit = the K slice (each XCD's L2 sees one
// slice of A and of G0), grid.y = ceil(M / (32 MT)); the launcher picks MT (ita_plugin.hip: launch_gemm_split).
// What this shape answers (all measured, 128 frames unless noted): a fragment is 16 bytes per lane at ROW r, so a
// wave-load of fragments touches 32 cache lines for 32 bytes each and the load pipeline's per-instruction cost, not
// bandwidth or latency, sets the pace -- one wave streaming its own tile into a register ring took 20 us whether its
// lines were hot or cold (8 us per operand; fetching whole lines per lane pair, touching everything first or placing
// the K slices per XCD changed nothing); the same wave with an LDS-DMA ring (8 whole lines per wave-instruction) 14 us,
// because an LDS-DMA piece costs its issuing wave ~100 cycles.  So the issue is spread over four waves, and the W tile
// is staged once for the M tiles that share it.
// Per output element the arithmetic is IDENTICAL to the large kernel's -- the same v_mfma_f32_32x32x16_f16 on the
// same operand values in the same lane slots, k-steps in the same order, the same three products per step -- so a
// frame's result does not depend on which of the two kernels served its batch (tests: a 5-frame batch against
// rows of a 1024-frame batch, bit for bit).
// k-tiles in the LDS ring (4..8 where 160 KB hold them measured 1-3 % slower at every M)
constexpr int ITA_GEMM_SMALL_NSTG = 3;
constexpr int ita_gemm_small_lds(int mt) { return ITA_GEMM_SMALL_NSTG * (2 * mt + 2) * 32 * 128; }
constexpr int ita_waitcnt_vm(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }   // s_waitcnt vmcnt(n) only
template <int MT>
__global__ __launch_bounds__(256) void ita_gemm_f16x3_small_kernel(const ItaGemmSplitArgs g) {
  constexpr int NSTG = ITA_GEMM_SMALL_NSTG, PLANE = 32 * 128, APL = MT * PLANE, STG = 2 * APL + 2 * PLANE;
  constexpr int DMA = 2 * MT + 2;   // LDS-DMA wave-instructions per wave and k-tile
  static_assert(NSTG == 3 && DMA < 64, "one younger k-tile in flight at the wait; vmcnt range");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const int z = blockIdx.x % g.nsplit, n0 = (blockIdx.x / g.nsplit) * 32, m0 = blockIdx.y * 32 * MT;
  const int kslice = g.K / g.nsplit, kbeg = z * kslice, nt = kslice / 64;
  auto stage = [&](int kt, int slot) {
    char* b = lds + slot * STG;
    const int k0 = kbeg + kt * 64;
    stage_plane<32 * MT, 256>(g.a_hi, g.lda, m0, g.M - 1, k0, b, tid);
    stage_plane<32 * MT, 256>(g.a_lo, g.lda, m0, g.M - 1, k0, b + APL, tid);
    stage_plane<32, 256>(g.w_hi, g.ldw, n0, g.N - 1, k0, b + 2 * APL, tid);
    stage_plane<32, 256>(g.w_lo, g.ldw, n0, g.N - 1, k0, b + 2 * APL + PLANE, tid);
  };
#pragma unroll
  for (int s = 0; s < NSTG - 1; ++s)
    if (s < nt) stage(s, s);
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
  const bool compute = wave < MT && m0 + 32 * wave < g.M;
  for (int kt = 0; kt < nt; ++kt) {
    // The compiler does not track LDS-DMA, so the waits are explicit.  This wave's pieces of tile kt have landed when
    // at most those of the younger tiles are outstanding; the barrier then says the same of every wave's pieces
    // -- and that every wave has finished reading tile kt - 1, whose slot the next stage call refills.
    if (kt + 1 < nt) __builtin_amdgcn_s_waitcnt(ita_waitcnt_vm(DMA)); else __builtin_amdgcn_s_waitcnt(ita_waitcnt_vm(0));
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's fragment reads of tile kt - 1 have returned
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (kt + NSTG - 1 < nt) stage(kt + NSTG - 1, (kt + NSTG - 1) % NSTG);
    if (compute) {
      const char* b = lds + (kt % NSTG) * STG;
      const char *pah = b + wave * PLANE, *pal = b + APL + wave * PLANE, *pwh = b + 2 * APL, *pwl = pwh + PLANE;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const f16x8 ah = frag_f16(pah, r, 2 * ks + h), al = frag_f16(pal, r, 2 * ks + h);
        const f16x8 wh = frag_f16(pwh, r, 2 * ks + h), wl = frag_f16(pwl, r, 2 * ks + h);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, acc, 0, 0, 0);
      }
    }
  }
  if (!compute) return;
  // C layout: col n = lane&31, row m = (e&3) + 8*(e>>2) + 4*h
  float* out = g.out + (size_t)z * g.M * g.N;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int m = m0 + 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * h;
    if (m < g.M) out[(size_t)m * g.N + n0 + r] = acc[e];
  }
}

// ---- and for M <= 32 (one M tile: the single-drone deployment): one wave per 32 x 32 output tile and K slice streams
// its fragments straight from L2 into an 8-slot register ring (7 k-steps in flight); grid (N/32, 1, nsplit).  With a
// single M tile nothing is shared inside a workgroup, and the barrier per k-tile of the kernel above costs more than
// the row-strided loads it avoids (M = 1: 5 us here, 8 us there).  Same arithmetic per output element again.
__global__ __launch_bounds__(64) void ita_gemm_f16x3_tiny_kernel(const ItaGemmSplitArgs g) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32, z = blockIdx.z;
  const int kslice = g.K / g.nsplit, kbeg = z * kslice, nstep = kslice / 16;
  const size_t ao = (size_t)min(m0 + r, g.M - 1) * g.lda + kbeg + 8 * h;
  // W comes from its fragment-order copy: 1 KB contiguous per wave-load.  From the row-major matrix a fragment load
  // touches 32 cache lines for 32 bytes each, which one CU pulls 4-6x slower (tools/microbench/cu_fill_rows.hip);
  // A keeps its row-major planes -- at M = 1 all its lanes read the same row.
  const size_t wo = (((size_t)blockIdx.x * (g.K / 16) + kbeg / 16) * 64 + lane) * 8;
  constexpr int R = 8;
  f16x8 ah[R], al[R], wh[R], wl[R];
  auto fetch = [&](int s, int slot) {
    ah[slot] = *(const f16x8*)(g.a_hi + ao + 16 * s);
    al[slot] = *(const f16x8*)(g.a_lo + ao + 16 * s);
    wh[slot] = *(const f16x8*)(g.wf_hi + wo + (size_t)s * 512);
    wl[slot] = *(const f16x8*)(g.wf_lo + wo + (size_t)s * 512);
  };
#pragma unroll
  for (int s = 0; s < R; ++s) fetch(s, s);
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
  for (int s0 = 0; s0 < nstep; s0 += R) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[j], wh[j], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], wl[j], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], wh[j], acc, 0, 0, 0);
      if (s0 + R + j < nstep) fetch(s0 + R + j, j);
    }
  }
  // C layout: col n = lane&31, row m = (e&3) + 8*(e>>2) + 4*h
  float* out = g.out + (size_t)z * g.M * g.N;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int m = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
    if (m < g.M) out[(size_t)m * g.N + n0 + r] = acc[e];
  }
}

// ------------------------------------------------------------------ LSTM layer 0
// The decoder Linear feeds nothing but LSTM layer 0 (QAT/model.py:124-128), so its weights are folded
// one step further at load time:  G0 = W_ih0[:, :512] . Wfold  (512 x 8192).  The big GEMM then
// yields layer 0's gate pre-activations directly (as split-K partials, columns in the permuted
// gate order below) and this kernel only adds the small remainder
//     [h_in0 | desvel/10 | quat] . [W_hh0 | W_ih0[:, 512:517]]^T        (K = 133, padded to 144)
// sums the partials in a fixed order, and performs the cell update.
// One wave per workgroup = 32 frames x 8 units; grid (16, ceil(B/32)); every load is issued before the
// first MFMA.
// The [x | h] operand planes of LSTM layers 1, 2 (K = 256, f16 hi and lo) live in FRAGMENT order: the 32x32x16 B fragment
// of frame tile fg, k-range kw (64 k, one per wave of ita_lstm_layer_kernel) and k-step s is 64 lanes x 16 bytes
// contiguous, lane (r = frame & 31, h) holding k = 64 kw + 16 s + 8 h .. + 7.  Row-major planes made every fragment
// load touch 32 cache lines for 32 bytes each (and every epilogue store 32 lines for 2 bytes each); the load pipeline's
// per-line cost, not bandwidth, is what these small kernels wait for.  Rows are padded to whole 32-frame tiles.
__device__ __forceinline__ size_t ita_lstm_plane_index(int b, int k) {
  return ((size_t)(((b >> 5) * 4 + (k >> 6)) * 4 + ((k >> 4) & 3)) * 64 + ((k >> 3) & 1) * 32 + (b & 31)) * 8 + (k & 7);
}
// Gate non-linearities of the f16x3 path: v_exp_f32 / v_rcp_f32 forms (1 ulp each, ~6 instructions instead of ~28 for
// the oracle's fixed-arithmetic expf + IEEE division).  This path's tolerance against the f32 oracle is 2e-5 (measured
// max |vel - oracle| stays <= 4e-6); the exact-f32 path (tail mode 0) keeps ita_sigmoid / ita_tanh.
__device__ __forceinline__ float lstm_sigmoid_fast(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}
__device__ __forceinline__ float lstm_tanh_fast(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * 2.88539008177792681f) + 1.0f);
}
struct ItaLstm0Args {
  const float* part; int nsplit; float inv_fold_scale;   // [nsplit][B][512] raw accumulators of x2 . (G0 * scale)^T
  const _Float16 *w_hi, *w_lo; float inv_wscale;         // [ut 16][k-step 9][lane 64][8] A fragments of the permuted, pre-scaled [W_hh0 | w_dv | w_quat | 0]
  const float* bias;                                     // [512] gate-major: W_ih0[:, :512].bias' + b_ih0 + b_hh0
  const float *desvel, *quat;                            // (B), (B,4)
  const float* h_in;                                     // (B,128) layer-0 hidden state STAGED by frame index
  const float* c_in;                                     // layer-0 cell state rows (slot- or frame-indexed)
  float *h_out, *c_out;
  _Float16 *nx_hi, *nx_lo; const float* nx_h_in;         // layer 1 operand planes [h_out | h_in1] in fragment order (ita_lstm_plane_index)
  int B;
  const int* slots;
};
template <int NS>
__global__ __launch_bounds__(64) void ita_lstm0_kernel(const ItaLstm0Args a) {
  // Every global load below is laid out so that a wave-load covers whole cache lines (8 lines of 128 B, not 32 lines for
  // 32 bytes each): the partials and the h rows are fetched row-wise and turned to the MFMA layouts through LDS.
  __shared__ __attribute__((aligned(16))) float hs[32][132];   // h_in rows of the 32 frames
  __shared__ __attribute__((aligned(16))) float pt[32][36];    // this tile's summed split-K partials [frame][gate * 8 + unit]
  const int lane = threadIdx.x;
  const int ut = blockIdx.x, r = lane & 31, h = lane >> 5;
  const int f0 = blockIdx.y * 32;
  const int b = f0 + r;
  const int bc = min(b, a.B - 1);
  const size_t sb = a.slots ? (size_t)a.slots[bc] : (size_t)bc;
  const int u0 = ut * 8 + 4 * h;
  // split-K partials: load i covers frames 8i .. 8i+7, lane (frame 8i + lane/8, columns 4 (lane % 8) ..)
  f32x4 pz[NS][4];
#pragma unroll
  for (int z = 0; z < NS; ++z)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      pz[z][i] = *(const f32x4*)(a.part + ((size_t)z * a.B + min(f0 + 8 * i + (lane >> 3), a.B - 1)) * 512 + ut * 32 + 4 * (lane & 7));
  // h rows: load i covers frames 2i, 2i+1
  f32x4 hv[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
    hv[i] = *(const f32x4*)(a.h_in + (size_t)min(f0 + 2 * i + h, a.B - 1) * 128 + 4 * r);
  // small GEMM: A operand = weights (rows = permuted gates), B operand = this frame's [h | dv | quat]
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
  const _Float16* wfr_hi = a.w_hi + ((size_t)ut * 9 * 64 + lane) * 8;   // fragment order: 1 KB per wave-load
  const _Float16* wfr_lo = a.w_lo + ((size_t)ut * 9 * 64 + lane) * 8;
  f16x8 wh[9], wl[9];
#pragma unroll
  for (int s = 0; s < 9; ++s) {
    wh[s] = *(const f16x8*)(wfr_hi + s * 512);
    wl[s] = *(const f16x8*)(wfr_lo + s * 512);
  }
  f32x4 xlast0 = {0.0f, 0.0f, 0.0f, 0.0f};
  float xlast1 = 0.0f;
  if (h == 0) {
    const f32x4 q = *(const f32x4*)(a.quat + (size_t)bc * 4);
    xlast0 = (f32x4){a.desvel[bc] / 10.0f, q.x, q.y, q.z};
    xlast1 = q.w;
  }
  // epilogue operands too: everything this lane will ever read is in flight before the first wait
  const f32x4 ci = *(const f32x4*)(a.c_in + sb * 128 + u0);
  const f32x4 nh = *(const f32x4*)(a.nx_h_in + sb * 128 + u0);
  f32x4 bz[4];
#pragma unroll
  for (int gt = 0; gt < 4; ++gt) bz[gt] = *(const f32x4*)(a.bias + gt * 128 + u0);
  __builtin_amdgcn_sched_barrier(0);
  // partials summed in their fixed order (elementwise, so the layout does not matter), then to LDS with the h rows
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x4 ps = pz[0][i];
#pragma unroll
    for (int z = 1; z < NS; ++z) ps += pz[z][i];
    *(f32x4*)&pt[8 * i + (lane >> 3)][4 * (lane & 7)] = ps;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) *(f32x4*)&hs[2 * i + h][4 * r] = hv[i];
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 9; ++s) {
    f32x4 x0 = xlast0, x1 = {xlast1, 0.0f, 0.0f, 0.0f};
    if (s < 8) {
      x0 = *(const f32x4*)&hs[r][16 * s + 8 * h];
      x1 = *(const f32x4*)&hs[r][16 * s + 8 * h + 4];
    }
    f16x8 xh, xl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = j < 4 ? x0[j & 3] : x1[j & 3];
      const _Float16 hi = (_Float16)x;
      xh[j] = hi;
      xl[j] = (_Float16)(x - (float)hi);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[s], xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], xl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], xh, acc, 0, 0, 0);
  }
  if (b >= a.B) return;
  // acc[4*gate + q] <-> gate (i,f,g,o), unit u0 + q, frame b
  f32x4 hn, cn;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float g[4];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt)
      g[gt] = (pt[r][gt * 8 + 4 * h + q] * a.inv_fold_scale + acc[4 * gt + q] * a.inv_wscale) + bz[gt][q];
    const float ig = lstm_sigmoid_fast(g[0]), fg = lstm_sigmoid_fast(g[1]), cg = lstm_tanh_fast(g[2]), og = lstm_sigmoid_fast(g[3]);
    const float c = fmaf(fg, ci[q], ig * cg);
    cn[q] = c;
    hn[q] = og * lstm_tanh_fast(c);
  }
  *(f32x4*)(a.c_out + sb * 128 + u0) = cn;
  *(f32x4*)(a.h_out + sb * 128 + u0) = hn;
  f16x4 h_hi, h_lo, n_hi, n_lo;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    _Float16 x, y;
    split_f16(hn[q], x, y); h_hi[q] = x; h_lo[q] = y;
    split_f16(nh[q], x, y); n_hi[q] = x; n_lo[q] = y;
  }
  const size_t o0 = ita_lstm_plane_index(b, u0), o1 = ita_lstm_plane_index(b, 128 + u0);   // 32 lanes x 16 B contiguous
  *(f16x4*)(a.nx_hi + o0) = h_hi;
  *(f16x4*)(a.nx_lo + o0) = h_lo;
  *(f16x4*)(a.nx_hi + o1) = n_hi;
  *(f16x4*)(a.nx_lo + o1) = n_lo;
}

// ------------------------------------------------------------------ one LSTM layer per launch
// gates = [x | h] . [W_ih | W_hh]^T on split-precision f16 MFMA, fused with the cell update
// (nn.LSTM, seq_len 1, gate order i,f,g,o; reference QAT/model.py:84,128-129).
// The 512 weight rows are permuted at load time to  r' = ut*32 + gate*8 + u  (ut = unit tile of 8
// hidden units): one 32x32 MFMA tile then holds i,f,g,o of 8 units x 32 frames, and the C layout
// (row = (e&3) + 8*(e>>2) + 4*h) puts all four gates of a (frame, unit) pair in ONE lane:
// e>>2 is the gate, (e&3) + 4*h the unit -- the cell update needs no cross-lane traffic.
struct ItaLstmLayerArgs {
  const _Float16 *a_hi, *a_lo; int lda;   // input planes [x | h_in] in fragment order (ita_lstm_plane_index); lda unused
  const _Float16 *w_hi, *w_lo; int ldw;   // [ut 16][k-range 4][k-step 4][lane 64][8] A fragments of the permuted, pre-scaled rows; ldw unused
  float inv_wscale;
  const float* bsum;      // [512] b_ih + b_hh, original gate-major order
  const float* c_in;      // (B,128)
  float *h_out, *c_out;   // (B,128)
  _Float16 *nx_hi, *nx_lo;   // next layer's planes [B][256] = [h_out | next h_in], or null
  const float* nx_h_in;      // (B,128)
  int B, K;               // K % 64 == 0
  const int* slots;       // optional: state row (c_in, h_out, c_out, nx_h_in) of frame b, else b
};
// Workgroup = 4 waves = 32 frames x 8 units; the K dimension is split over the four waves, each
// streaming its operand fragments straight from L2 into registers (all loads issued up front: the
// LDS-staged version paid one memory latency per 64-deep K tile and was purely latency-bound); the
// four partial accumulators are combined through LDS in a fixed order.  grid (16, ceil(B/32)).
template <int NK>   // 16-deep k-steps per wave: K = 64 * NK
__global__ __launch_bounds__(256) void ita_lstm_layer_kernel(const ItaLstmLayerArgs a) {
  __shared__ float part[4][16][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ut = blockIdx.x, f0 = blockIdx.y * 32;
  static_assert(NK == 4, "fragment-order operands: four k-ranges of four k-steps (K = 256)");
  const size_t ao = ((size_t)((blockIdx.y * 4 + wave) * 4) * 64 + lane) * 8;   // 1 KB per wave-load, both operands
  const size_t wo = ((size_t)((ut * 4 + wave) * 4) * 64 + lane) * 8;
  f16x8 fah[NK], fal[NK], fwh[NK], fwl[NK];
#pragma unroll
  for (int s = 0; s < NK; ++s) {
    fwh[s] = *(const f16x8*)(a.w_hi + wo + 512 * s);
    fah[s] = *(const f16x8*)(a.a_hi + ao + 512 * s);
    fwl[s] = *(const f16x8*)(a.w_lo + wo + 512 * s);
    fal[s] = *(const f16x8*)(a.a_lo + ao + 512 * s);
  }
  // The cell update is dealt out by MEMORY layout, not by MFMA layout: thread t finishes (frame t / 8, unit t % 8), so a
  // wave reads and writes 8 frames x 8 consecutive units = 8 row segments of 32 bytes of the (3, rows, 128) state per
  // instruction (dealt by the C layout -- wave q = unit 4h + q of frame r -- it was 4 bytes in each of 64 lines), and
  // one 128-byte line of the next layer's fragment-order planes.
  const int ef = tid >> 3, eu = tid & 7;
  const int b = f0 + ef, u = ut * 8 + eu;
  const size_t sb = a.slots ? (size_t)a.slots[min(b, a.B - 1)] : (size_t)min(b, a.B - 1);
  const float c_prev = a.c_in[sb * 128 + u];
  const float nx_prev = a.nx_hi ? a.nx_h_in[sb * 128 + u] : 0.0f;
  const float b_i = a.bsum[u], b_f = a.bsum[128 + u], b_g = a.bsum[256 + u], b_o = a.bsum[384 + u];
  __builtin_amdgcn_sched_barrier(0);   // keep every load ahead of the first MFMA (hipcc otherwise sinks them)
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
  for (int s = 0; s < NK; ++s) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwl[s], fah[s], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh[s], fal[s], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh[s], fah[s], acc, 0, 0, 0);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) part[wave][e][lane] = acc[e];
  __syncthreads();
  // C layout of a tile: accumulator e of lane (r, h) = gate e / 4 of unit (e % 4) + 4 h, frame r
  if (b >= a.B) return;
  float gsum[4];
  const int pl = ef + 32 * (eu >> 2);
#pragma unroll
  for (int gte = 0; gte < 4; ++gte) {
    const int e = 4 * gte + (eu & 3);
    gsum[gte] = ((part[0][e][pl] + part[1][e][pl]) + part[2][e][pl]) + part[3][e][pl];
  }
  const float gi = gsum[0] * a.inv_wscale + b_i, gf = gsum[1] * a.inv_wscale + b_f,
              gg = gsum[2] * a.inv_wscale + b_g, go = gsum[3] * a.inv_wscale + b_o;
  const float ig = lstm_sigmoid_fast(gi), fg = lstm_sigmoid_fast(gf), cg = lstm_tanh_fast(gg), og = lstm_sigmoid_fast(go);
  const float c = fmaf(fg, c_prev, ig * cg);
  const float hn = og * lstm_tanh_fast(c);
  a.c_out[sb * 128 + u] = c;
  a.h_out[sb * 128 + u] = hn;
  if (a.nx_hi) {
    _Float16 x, y;
    const size_t o0 = ita_lstm_plane_index(b, u), o1 = ita_lstm_plane_index(b, 128 + u);
    split_f16(hn, x, y);
    a.nx_hi[o0] = x;
    a.nx_lo[o0] = y;
    split_f16(nx_prev, x, y);
    a.nx_hi[o1] = x;
    a.nx_lo[o1] = y;
  }
}

// f32 rows -> the hi / lo f16 planes the split-precision GEMM reads (the encoder kernels write them directly;
// this is for ita_vitlstm_tail, which starts from a given activation): eight values per thread
__global__ void ita_split_planes_kernel(const float* __restrict__ x, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                        int width, int ld, int rows) {
  const size_t i8 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i8 >= (size_t)rows * width) return;
  const size_t r = i8 / width, c = i8 - r * width;
  f16x8 vh, vl;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    _Float16 a, b;
    split_f16(x[i8 + j], a, b);
    vh[j] = a; vl[j] = b;
  }
  *(f16x8*)(hi + r * ld + c) = vh;
  *(f16x8*)(lo + r * ld + c) = vl;
}

// one-hot rows for the load-time folding: x[i][col0 + i] = 1
__global__ void ita_impulse_kernel(float* x, int rows, int width, int col0) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)rows * width) return;
  const int i = (int)(idx / width), c = (int)(idx % width);
  x[idx] = (c == col0 + i) ? 1.0f : 0.0f;
}

// ------------------------------------------------------------------ fusion tail on large token grids
// BASELINE config 5: the layers of QAT/model.py:116-121 on a tok_h x tok_w
// token grid with out_ch conv outputs,  x (B, tok_h*tok_w, E) f32 -> out (B, out_ch, 2 tok_h, 2 tok_w) f32:
// PixelShuffle(2) || Upsample(x2, bilinear, align_corners=True) -> cat (5E/4 channels) -> conv3x3 pad 1.
// At 64 x 128 tokens, E = 128, 48 outputs this is 2.26 G MAC per frame against 5 MB of traffic: an
// implicit GEMM on split-precision f16 MFMA (hi + lo operand planes, three products, f32 accumulate --
// the same scheme and error, ~2^-22 relative, as ita_gemm_f16x3_kernel), never materialising the
// 160-channel concatenated map:
//   workgroup = WAVES (8, or 4 when the map height is not a multiple of 16) waves = one (2 WAVES) x 32 output
//   tile, all output channels; two waves per SIMD cover each other's LDS latency; wave w owns rows 2w, 2w+1 as four
//   16-pixel M tiles x NT 16-channel N tiles of v_mfma_f32_16x16x32_f16;
//   K loop: channel chunks of 32 x 9 taps; per chunk the halo of the concatenated map is BUILT in
//   LDS from the tokens (shuffle gather / exact f32 bilinear blend, then split into f16 hi + lo,
//   pixel-major [pixel][32 ch]) next to the chunk's weights [tap][co][32 ch] (prepared at load time).
struct ItaTailBigArgs {
  const float* x;                 // (B, TH*TW, E)
  const _Float16 *w_hi, *w_lo;    // [chunks][9 taps][NT*16][32] pre-scaled, zero padded
  const float* bias;              // [NT*16]
  float inv_wscale;
  float* out;                     // (B, CO, 2TH, 2TW)
  int B, E, TH, TW, CO, nchunk;
  int accumulate;                 // 1: out += this kernel's part (no bias): the upsample channels came from ita_tail_up_kernel
};
template <int NT, int WAVES, int TPS>
struct ItaTailBigLds {
  static constexpr int HP = (2 * WAVES + 2) * 34;          // halo pixels of a (2 WAVES) x 32 tile
  static constexpr int A_PLANE = HP * 64;                  // [pixel][32 ch] f16
  static constexpr int W_PLANE = TPS * NT * 16 * 64;       // [tap (TPS of the 9 at a time)][co][32 ch] f16
  static constexpr int AH = 0, AL = A_PLANE, WH = 2 * A_PLANE, WL = WH + W_PLANE;
  static constexpr int GEO_I = WL + W_PLANE;               // per halo pixel: {upsample offset, shuffle offset, flags, -}
  static constexpr int GEO_F = GEO_I + HP * 16;            // per halo pixel: {h1, w1}
  static constexpr int TOTAL = GEO_F + HP * 8;
};

template <int NT, int WAVES, int TPS>
__global__ __launch_bounds__(64 * WAVES) void ita_tail_big_kernel(const ItaTailBigArgs a) {
  using L = ItaTailBigLds<NT, WAVES, TPS>;
  static_assert(9 % TPS == 0, "taps per weight stage must divide 9");
  constexpr int W_CHUNK_HALVES = 9 * NT * 16 * 32;   // f16 elements of one channel chunk's weights per plane in global
  constexpr int NTHR = 64 * WAVES;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int OH = 2 * a.TH, OW = 2 * a.TW, C4 = a.E / 4, CIN = C4 + a.E;
  const int tx0 = blockIdx.x * 32, ty0 = blockIdx.y * 2 * WAVES, b = blockIdx.z;
  const float* xt = a.x + (size_t)b * a.TH * a.TW * a.E;
  const float sy = (float)(a.TH - 1) / (float)(OH - 1), sx = (float)(a.TW - 1) / (float)(OW - 1);

  f32x4 acc[4][NT];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

  // ---- per halo pixel, once per tile: where its shuffle source and its four bilinear sources sit in x, the
  // two interpolation weights, validity.  (Recomputing this for every (chunk, channel quad) cost more VALU
  // time than the MFMAs of the chunk.)
  for (int p = tid; p < L::HP; p += NTHR) {
    const int hy = p / 34, hx = p - 34 * hy;
    const int y = ty0 + hy - 1, x = tx0 + hx - 1;
    const int ok = (y >= 0 && y < OH && x >= 0 && x < OW) ? 4 : 0;
    const int yc = min(max(y, 0), OH - 1), xc = min(max(x, 0), OW - 1);
    const float fy = sy * (float)yc, fx = sx * (float)xc;   // bilinear x2, align_corners=True
    int y0 = (int)fy, x0 = (int)fx;
    if (y0 > a.TH - 1) y0 = a.TH - 1;
    if (x0 > a.TW - 1) x0 = a.TW - 1;
    const int yp = y0 < a.TH - 1 ? 2 : 0, xp = x0 < a.TW - 1 ? 1 : 0;
    *(i32x4*)(lds + L::GEO_I + p * 16) = (i32x4){(y0 * a.TW + x0) * a.E,
                                                   ((yc >> 1) * a.TW + (xc >> 1)) * a.E + 2 * (yc & 1) + (xc & 1), ok | yp | xp, 0};
    *(f32x2*)(lds + L::GEO_F + p * 8) = (f32x2){fy - (float)y0, fx - (float)x0};
  }

  for (int ch = 0; ch < a.nchunk; ++ch) {
    __syncthreads();   // the previous chunk's fragments are consumed (first pass: the geometry table is complete)
    // ---- weights: already in fragment order -> a straight LDS-DMA copy (global_load_lds_dwordx4, 1 KB per wave
    // instruction, asynchronous).  TPS of the chunk's 9 taps are resident at a time: with TPS = 3 the workgroup
    // needs 70 KB of LDS and TWO of them share a CU -- one builds its halo while the other runs MFMAs.
    auto stage_w = [&](int tap0) {
      const _Float16* gh = a.w_hi + (size_t)ch * W_CHUNK_HALVES + (size_t)tap0 * NT * 16 * 32;
      const _Float16* gl = a.w_lo + (size_t)ch * W_CHUNK_HALVES + (size_t)tap0 * NT * 16 * 32;
      for (int wc = wave; wc < L::W_PLANE / 1024; wc += WAVES) {
        const int piece = wc * 64 + lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gh + piece * 8),
                                         (__attribute__((address_space(3))) void*)(lds + L::WH + wc * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gl + piece * 8),
                                         (__attribute__((address_space(3))) void*)(lds + L::WL + wc * 1024), 16, 0, 0);
      }
    };
    stage_w(0);
    // ---- halo of the concatenated map, channels 32ch .. 32ch+31: thread = (pixel, 4 channels).  Items are
    // processed G at a time with every global load of the group issued (from clamped, always valid addresses)
    // before the first use: one memory latency per group instead of one per item.
    {
      constexpr int G = 5, NITEM = L::HP * 8;
      const bool all_shuffle = ch * 32 + 31 < C4, all_up = ch * 32 >= C4;
      for (int i0 = tid; i0 < NITEM; i0 += NTHR * G) {
        f32x4 q00[G], q01[G], q10[G], q11[G];
        f32x2 hw[G];
        int item[G];
        bool ok[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int i = min(i0 + NTHR * g, NITEM - 1);
          const int p = i >> 3, cq = i & 7;
          const int c0 = ch * 32 + 4 * cq;
          const i32x4 gi = *(const i32x4*)(lds + L::GEO_I + p * 16);
          hw[g] = *(const f32x2*)(lds + L::GEO_F + p * 8);
          item[g] = i;
          ok[g] = i0 + NTHR * g < NITEM && (gi.z & 4) && c0 < CIN;
          if (all_shuffle || (!all_up && c0 < C4)) {   // PixelShuffle(2): out[c][2h+i][2w+j] = in[4c+2i+j][h][w]
            const float* t = xt + gi.y + 4 * min(c0, C4 - 4);
            q00[g] = (f32x4){t[0], t[4], t[8], t[12]};
            q01[g] = q10[g] = q11[g] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
          } else {
            const float* t00 = xt + gi.x + min(max(c0 - C4, 0), a.E - 4);
            const int dx = (gi.z & 1) ? a.E : 0, dy = (gi.z & 2) ? a.TW * a.E : 0;
            q00[g] = *(const f32x4*)t00;
            q01[g] = *(const f32x4*)(t00 + dx);
            q10[g] = *(const f32x4*)(t00 + dy);
            q11[g] = *(const f32x4*)(t00 + dy + dx);
          }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (i0 + NTHR * g < NITEM) {
            const int p = item[g] >> 3, cq = item[g] & 7;
            const bool shuf = all_shuffle || (!all_up && ch * 32 + 4 * cq < C4);
            const float h1 = hw[g].x, h0 = 1.0f - h1, w1 = hw[g].y, w0 = 1.0f - w1;
            f16x4 vh, vl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              // (ita_tail_kernel's / the oracle's expression; the shuffle branch passes its value through)
              float v = shuf ? q00[g][j] : h0 * (w0 * q00[g][j] + w1 * q01[g][j]) + h1 * (w0 * q10[g][j] + w1 * q11[g][j]);
              if (!ok[g]) v = 0.0f;
              _Float16 hq, lq;
              split_f16(v, hq, lq);
              vh[j] = hq; vl[j] = lq;
            }
            *(f16x4*)(lds + L::AH + p * 64 + cq * 8) = vh;
            *(f16x4*)(lds + L::AL + p * 64 + cq * 8) = vl;
          }
        }
      }
    }
    __syncthreads();
    // ---- 9 taps x (4 M tiles x NT N tiles) x 3 products; A lane = (pixel l&15, channels 8(l>>4)..+7)
    const int px = lane & 15, kg = lane >> 4;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      if (TPS < 9 && tap > 0 && tap % TPS == 0) {
        __syncthreads();     // the resident taps are consumed
        stage_w(tap);
        __syncthreads();     // (hipcc drains the LDS-DMA, vmcnt(0), ahead of the barrier)
      }
      const int ky = tap / 3, kx = tap - 3 * ky, tl = tap % TPS;
      f16x8 bh[NT], bl[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int off = ((tl * NT * 16 + nt * 16 + px) * 64) + kg * 16;
        bh[nt] = *(const f16x8*)(lds + L::WH + off);
        bl[nt] = *(const f16x8*)(lds + L::WL + off);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int hp = (2 * wave + (mt >> 1) + ky) * 34 + (mt & 1) * 16 + px + kx;
        const f16x8 ah = *(const f16x8*)(lds + L::AH + hp * 64 + kg * 16);
        const f16x8 al = *(const f16x8*)(lds + L::AL + hp * 64 + kg * 16);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[nt], acc[mt][nt], 0, 0, 0);
        }
      }
    }
  }
  // C layout: column n = lane&15 (output channel), rows m = 4*(lane>>4) + i (pixel)
  const int co_l = lane & 15, m0 = 4 * (lane >> 4);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int y = ty0 + 2 * wave + (mt >> 1), x = tx0 + (mt & 1) * 16 + m0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = nt * 16 + co_l;
      if (co < a.CO) {
        const float bv = a.accumulate ? 0.0f : a.bias[co];
        float* op = a.out + (((size_t)b * a.CO + co) * OH + y) * OW + x;
        f32x4 o = a.accumulate ? *(const f32x4*)op : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = o[i] + (acc[mt][nt][i] * a.inv_wscale + bv);
        *(f32x4*)op = o;
      }
    }
  }
}

// ------------------------------------------------------------------ fusion tail on large token grids: the upsample branch by linearity
// The 4E/5 upsampled channels of the concatenated map are bilinear interpolations of the TOKENS, and conv3x3 is linear, so for
// every tap t the channel contraction commutes with the interpolation:
//     sum_c W[co][c][t] * U[q][c]  =  sum_{s in 4 neighbours of q} I[q][s] * Y_t[s][co],     Y_t[s][co] = sum_c W[co][c][t] * x[s][c]
// -- Y_t is a GEMM over the LOW-RESOLUTION tokens (a quarter of the pixels: 4x fewer MACs than the implicit GEMM on the
// upsampled map, and no upsampled map, halo build or hi/lo split of 160 x 32768 values per frame), the interpolation an
// exact f32 blend of four Y rows per tap.  Zero padding of the conv = taps whose position q falls outside the map are skipped.
//   persistent workgroups (512 threads, one per CU: the kernel fills the 160 KB of LDS), tiles dealt round robin; a tile = 16 x 32
//   output pixels, all output channels; its source region is at most 10 x 18 tokens (host-checked) = 12 M tiles of 16 tokens x 3 N
//   tiles = 36 units, 5 / 4 per wave (details at the token load), the tokens' f16 hi / lo fragments resident in registers for all nine
//   taps -- and requested for the NEXT tile as soon as phase 2 no longer needs them;
//   per tap: Y_t = W_t . x^T on v_mfma_f32_16x16x32_f16 (three split-precision products, weights as the A operand so that a
//   lane holds four consecutive output channels of one token: one 16-byte LDS store) -> Y slab [192 tokens][52] f32 in LDS;
//   blend: thread = output pixel, 48 accumulators, four Y rows per tap (12 x ds_read_b128 each), scalar v_fmac_f32;
//   one barrier per tap: GEMM(t) and blend(t - 1) of a wave as one software pipeline, two Y slabs, two weight buffers (LDS-DMA).
// The E/4 pixel-shuffle channels are no interpolation: phase 2 of the same workgroup runs them as an implicit GEMM on the tile's
// halo, written to LDS from the token fragments the waves hold (no second trip to memory), and the two parts are added in LDS and
// stored once, 16 bytes per lane.  What bounds it: MFMA time + the blend's VALU issue, which add.
struct ItaTailUpArgs {
  const float* x;                  // (B, TH*TW, 128)
  const _Float16 *w_hi, *w_lo;     // [9 taps][4 k-steps][3 N tiles][64 lanes][8]: A fragments (row = output channel), pre-scaled
  const _Float16 *s_hi, *s_lo;     // pixel-shuffle channels (conv input channels 0..31): [9 taps][48][32], same scale, zero rows beyond CO
  const float* bias;               // [48]
  float inv_wscale;
  float* out;                      // (B, CO, 2TH, 2TW)
  int B, TH, TW, CO;
};
struct ItaTailUpLds {
  static constexpr int RH = 10, RW = 18, NTOK = 192;
  static constexpr int YLD = 52;                             // floats per token row (48 + 4: conflict-free 16-byte stores and loads)
  static constexpr int YB = NTOK * YLD * 4;                  // bytes per Y slab
  static constexpr int Y = 0;
  static constexpr int WPL = 4 * 3 * 1024;                   // bytes per weight plane and tap
  static constexpr int W = 2 * YB;                           // two buffers x (hi | lo)
  static constexpr int TOTAL1 = W + 2 * 2 * WPL;             // phase 1: two Y slabs + two weight buffers
  // phase 2: shuffle halo hi | lo at 0 (over the Y slabs); its weights sit at the END of the 160 KB so that they can be staged
  // while phase 1 still runs: the lo plane lies wholly behind phase 1's LDS, the hi plane overlaps weight buffer 1 only (free
  // once GEMM(7) is done)
  static constexpr int HP = 18 * 34, A_PLANE = HP * 64, W2_PLANE = 9 * 48 * 64;
  static constexpr int TOTAL = 160 * 1024;
  static constexpr int WL2 = TOTAL - W2_PLANE, WH2 = WL2 - W2_PLANE;
  static_assert(WL2 >= TOTAL1 && WH2 >= W + 2 * WPL && 2 * A_PLANE <= WH2 && 48 * 516 * 4 <= WH2, "phase-2 LDS");
};

#ifndef ITA_UP_NB
#define ITA_UP_NB 2
#endif
#ifndef ITA_UP_PKFMA
#define ITA_UP_PKFMA 0
#endif
#ifdef ITA_UP_STAMP
// diagnostic build only (tools/tail_up_stamps.py): s_memrealtime (100 MHz) of every wave at the phase boundaries (<= 2048 workgroups)
__device__ unsigned long long ita_up_stamp_buf[2048 * 8 * 12];
#define ITA_UP_ST(i) do { if (lane == 0 && tile < 2048) ita_up_stamp_buf[((size_t)tile * 8 + wave) * 12 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ITA_UP_ST(i) do { } while (0)
#endif
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void ita_tail_up_kernel(const ItaTailUpArgs a) {
  using L = ItaTailUpLds;
  constexpr int E = 128;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int tid = threadIdx.x, lane = tid & 63;              // (not const: re-derived per tile, see the tile loop)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int OH = 2 * a.TH, OW = 2 * a.TW;
  const float sy = (float)(a.TH - 1) / (float)(OH - 1), sx = (float)(a.TW - 1) / (float)(OW - 1);
  // persistent: workgroup w runs tiles w, w + gridDim.x, ... (tile = (frame, tile row, tile column), column fastest).  At 160 KB
  // of LDS a CU holds one workgroup, so with one workgroup per tile every token fetch (an HBM round trip, 1.3 - 3.7 us) and every
  // workgroup launch was exposed: now the next tile's tokens are requested during phase 2, when the fragment registers are free
  const int tiles_x = OW / 32, tiles_y = OH / 16, ntiles = tiles_x * tiles_y * a.B;
  int tx0, ty0, b, ry0, rx0;
  const float* xt;
  // source row / column of a clamped output coordinate: the oracle's upsample_src_ac (bilinear x2, align_corners=True)
  auto src = [](int q, float s, int n, int& i0, int& ip, float& l1) {
    const float f = s * (float)q;
    int i = (int)f;
    if (i > n - 1) i = n - 1;
    i0 = i; ip = i < n - 1 ? 1 : 0; l1 = f - (float)i;
  };
  auto set_tile = [&](int tile) {
    const int bx = tile % tiles_x, r_ = tile / tiles_x;
    tx0 = bx * 32; ty0 = (r_ % tiles_y) * 16; b = r_ / tiles_y;
    xt = a.x + (size_t)b * a.TH * a.TW * E;
    int dummy_i; float dummy_f;
    src(max(ty0 - 1, 0), sy, a.TH, ry0, dummy_i, dummy_f);
    src(max(tx0 - 1, 0), sx, a.TW, rx0, dummy_i, dummy_f);
  };

  // ---- weights of taps 0 and 1 on their way into LDS (LDS-DMA: 24 pieces of 1 KB per tap, three per wave)
  auto stage_w = [&](int tap) {
    const int buf = tap & 1;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int piece = wave + 8 * q;                      // 0..23: hi plane pieces 0..11, lo plane 12..23
      const _Float16* g = (piece < 12 ? a.w_hi : a.w_lo) + (size_t)tap * (L::WPL / 2) + (size_t)(piece % 12) * 512 + lane * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(lds + L::W + buf * 2 * L::WPL + piece * 1024), 16, 0, 0);
    }
  };
  auto stage_w2 = [&](const _Float16* g, int dst) {
    for (int wc = wave; wc < L::W2_PLANE / 1024; wc += 8)     // 27 pieces per plane
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (wc * 64 + lane) * 8),
                                       (__attribute__((address_space(3))) void*)(lds + dst + wc * 1024), 16, 0, 0);
  };

  // ---- this wave's token tiles as B fragments (column = token, k = the 8 channels 32 j + 8 (lane >> 4) .. + 7 of each k-step), f16 hi / lo.
  // 12 M tiles of 16 tokens x 3 N tiles = 36 units for 8 waves: wave w owns tile w (three units) and shares tile 8 + (w & 3)
  // with its SIMD partner w ^ 4 -- waves 0..3 take its N tiles 0 and 1, waves 4..7 N tile 2 (and compute N tile 1 once more
  // so that all waves run the same straight-line code): 5 / 4 units instead of 6 / 3
  const bool heavy = wave < 4;
  const int nt3 = heavy ? 0 : 2;
  f32x4 raw[2][4][2];                                         // a tile's tokens as loaded (64 registers: those of xh / xl)
  f16x8 xh[2][4], xl[2][4];
  auto request_tokens = [&]() {                               // (of the tile set_tile() has just selected)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int slot = min(16 * (m ? 8 + (wave & 3) : wave) + (lane & 15), L::RH * L::RW - 1);   // (slots 180..191: unused copies)
      const int r = slot / L::RW, c = slot - L::RW * r;
      const float* tp = xt + ((size_t)min(ry0 + r, a.TH - 1) * a.TW + min(rx0 + c, a.TW - 1)) * E + 8 * (lane >> 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { raw[m][j][0] = *(const f32x4*)(tp + 32 * j); raw[m][j][1] = *(const f32x4*)(tp + 32 * j + 4); }
    }
  };
  auto split_tokens = [&]() {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          _Float16 h, l;
          // element 2 e <-> channel offset e, element 2 e + 1 <-> offset 4 + e (the weight image uses the same order): dword e of a
          // fragment is then the channel pair (c, c + 1) of pixel-shuffle parity e -- phase 2 writes it to its halo as it is
          split_f16(raw[m][j][0][e], h, l); xh[m][j][2 * e] = h; xl[m][j][2 * e] = l;
          split_f16(raw[m][j][1][e], h, l); xh[m][j][2 * e + 1] = h; xl[m][j][2 * e + 1] = l;
        }
  };
  int tile = blockIdx.x;
  set_tile(tile);
  request_tokens();
  stage_w2(a.s_lo, L::WL2);                            // phase 2's lo weight plane lies behind everything else in the LDS: staged once

#pragma unroll 1
  for (; tile < ntiles; tile += gridDim.x) {
  // everything derived from the thread id is recomputed per tile: hoisted out of this loop (hipcc does) the ~40 lane-dependent
  // addresses and offsets stay live through phase 1 and spill
  asm volatile("" : "+v"(tid));
  lane = tid & 63;
  ITA_UP_ST(0);
  __syncthreads();                                     // the previous tile's hand-over reads are done: the LDS is free
  stage_w(0);
  split_tokens();

  // ---- this thread's output pixel (its source rows / columns and weights are recomputed per tap: 30 instructions against 96
  // packed FMAs, and 24 registers fewer across the loop)
  const int py = ty0 + (tid >> 5), px = tx0 + (tid & 31);
  f32x4 acc[12];
#pragma unroll
  for (int g = 0; g < 12; ++g) acc[g] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

  // One tap step = GEMM(t) and blend(t - 1) of this wave as ONE software pipeline over the four k-steps: the LDS reads of
  // k-step j + 1 (weight fragments) and of blend groups 3 j + 3 .. 3 j + 5 (Y rows) are in flight while the MFMAs of k-step j
  // issue and the FMAs of groups 3 j .. 3 j + 2 run; sched_barriers pin that order (left alone hipcc runs the GEMM, then the
  // blend with a wait per group: 6.9 k cycles per tap against 1.7 k of MFMA).  A wave can have 16 LDS instructions in flight.
  auto tap = [&](auto G_, auto B_, const int t) {
    constexpr bool DO_G = decltype(G_)::value, DO_B = decltype(B_)::value;
    const char* wb = lds + L::W + (t & 1) * 2 * L::WPL;
    float* ysw = (float*)(lds + L::Y + (t & 1) * L::YB);
    // blend(t - 1): the four Y rows and weights of this thread's pixel
    const int tb = t - 1, dy = tb / 3, dx = tb - 3 * dy;
    const float* p00 = nullptr; const float* p01 = nullptr; const float* p10 = nullptr; const float* p11 = nullptr;
    f32x4 q00 = {}, q01 = {}, q10 = {}, q11 = {};
    if (DO_B) {
      const float* ysr = (const float*)(lds + L::Y + (tb & 1) * L::YB);
      const int qy = py + dy - 1, qx = px + dx - 1;
      int r0, rp, c0, cp; float h1, w1;
      src(min(max(qy, 0), OH - 1), sy, a.TH, r0, rp, h1);
      src(min(max(qx, 0), OW - 1), sx, a.TW, c0, cp, w1);
      const float ok = (qy >= 0 && qy < OH && qx >= 0 && qx < OW) ? 1.0f : 0.0f;   // 0: the tap's position lies in the conv's zero padding
      const float h0 = 1.0f - h1, w0 = 1.0f - w1;
      const float k00 = ok * (h0 * w0), k01 = ok * (h0 * w1), k10 = ok * (h1 * w0), k11 = ok * (h1 * w1);
      q00 = (f32x4){k00, k00, k00, k00}; q01 = (f32x4){k01, k01, k01, k01}; q10 = (f32x4){k10, k10, k10, k10}; q11 = (f32x4){k11, k11, k11, k11};
      p00 = ysr + ((r0 - ry0) * L::RW + (c0 - rx0)) * L::YLD;
      p01 = p00 + cp * L::YLD;
      p10 = p00 + rp * L::RW * L::YLD;
      p11 = p10 + cp * L::YLD;
    }
    f32x4 c[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) c[u] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    // weight fragments of one k-step, N tiles in the order (nt3, 1, 2 - nt3): units 0..2 = tile A x them, 3 and 4 = tile B x the first two
    f16x8 wh[3], wl[3];
    constexpr int NB = ITA_UP_NB;                        // blend groups in flight
    f32x4 v[NB][4];
    auto ntq = [&](int q) { return q == 1 ? 1 : (q == 0 ? nt3 : 2 - nt3); };
    auto ldwh = [&](int j) {
#pragma unroll
      for (int q = 0; q < 3; ++q) wh[q] = *(const f16x8*)(wb + ((j * 3 + ntq(q)) * 64 + lane) * 16);
    };
    auto ldwl = [&](int j) {
#pragma unroll
      for (int q = 0; q < 3; ++q) wl[q] = *(const f16x8*)(wb + L::WPL + ((j * 3 + ntq(q)) * 64 + lane) * 16);
    };
    auto ldv = [&](int g) {
#if defined(ITA_UP_ABLATE) && ITA_UP_ABLATE == 5
      if (g >= 4) return;                                // timing experiment: a third of the blend's LDS reads, all of its FMAs (wrong results)
#endif
      v[g % NB][0] = *(const f32x4*)(p00 + 4 * g); v[g % NB][1] = *(const f32x4*)(p01 + 4 * g);
      v[g % NB][2] = *(const f32x4*)(p10 + 4 * g); v[g % NB][3] = *(const f32x4*)(p11 + 4 * g);
    };
    if (DO_G) { ldwl(0); ldwh(0); }
    if (DO_B) {
#pragma unroll
      for (int g = 0; g < NB; ++g) ldv(g);
    }
    // twelve micro-steps: five MFMAs (one split-precision product of the five units) | the FMAs of one blend group | reads
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int j = i / 3, pr = i % 3;                   // products in the order lo.hi, hi.lo, hi.hi: wl is free after the first
      if (DO_G) {
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const int m = u < 3 ? 0 : 1, q = u < 3 ? u : u - 3;
          c[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pr == 0 ? wl[q] : wh[q], pr == 1 ? xl[m][j] : xh[m][j], c[u], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (pr == 0 && j < 3) ldwl(j + 1);
        if (pr == 2 && j < 3) ldwh(j + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (DO_B) {
        // (explicit fma: the library is built with -ffp-contract=off)
#if ITA_UP_PKFMA
        acc[i] = __builtin_elementwise_fma(q00, v[i % NB][0], acc[i]);
        acc[i] = __builtin_elementwise_fma(q01, v[i % NB][1], acc[i]);
        acc[i] = __builtin_elementwise_fma(q10, v[i % NB][2], acc[i]);
        acc[i] = __builtin_elementwise_fma(q11, v[i % NB][3], acc[i]);
#else
        // scalar v_fmac_f32, not v_pk_fma_f32: beside MFMAs a packed f32 FMA costs ~20 cycles more than the two scalar ones
        // (MI355X guide, 'price of one filler beside MFMAs'); hipcc packs whatever it sees, hence the asm
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const float kq = nb == 0 ? q00[0] : nb == 1 ? q01[0] : nb == 2 ? q10[0] : q11[0];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float r = acc[i][e];
            asm("v_fmac_f32 %0, %1, %2" : "+v"(r) : "v"(kq), "v"(v[i % NB][nb][e]));
            acc[i][e] = r;
          }
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (i + NB < 12) ldv(i + NB);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (DO_G) {
      // C layout: lane = (token column lane & 15, output channels 4 (lane >> 4) .. + 3 of the N tile)
      const int rowA = (16 * wave + (lane & 15)) * L::YLD + 4 * (lane >> 4);
      const int rowB = (16 * (8 + (wave & 3)) + (lane & 15)) * L::YLD + 4 * (lane >> 4);
#pragma unroll
      for (int q = 0; q < 3; ++q) *(f32x4*)(ysw + rowA + 16 * ntq(q)) = c[q];
      *(f32x4*)(ysw + rowB + 16 * nt3) = c[3];
      *(f32x4*)(ysw + rowB + 16) = c[4];              // (both waves of a SIMD store this unit: the same bits from the same instruction sequence; a branch here would let the compiler sink the blend's FMAs behind it)
    }
  };
  using std::true_type; using std::false_type;

  ITA_UP_ST(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of W[0] have landed
  __syncthreads();
  stage_w(1);
  tap(true_type{}, false_type{}, 0);
#pragma unroll 1
  for (int t = 1; t < 9; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of W[t] have landed
    __syncthreads();                                   // W[t] complete; GEMM(t-1) done by every wave: Y[(t-1)&1] complete, W[(t+1)&1] free; blend(t-2) done: Y[t&1] free
    if (t + 1 < 9) stage_w(t + 1);
    if (t == 8) stage_w2(a.s_hi, L::WH2);            // phase 2's hi plane: into weight buffer 1, free from here on
#ifndef ITA_UP_ABLATE
    tap(true_type{}, true_type{}, t);
#elif ITA_UP_ABLATE == 1
    tap(false_type{}, true_type{}, t);                 // timing experiments only: wrong results
#elif ITA_UP_ABLATE == 2
    tap(true_type{}, false_type{}, t);
#elif ITA_UP_ABLATE == 4 || ITA_UP_ABLATE == 5
    tap(true_type{}, true_type{}, t);
#else
    (void)t;
#endif
  }
  __syncthreads();
  tap(false_type{}, true_type{}, 9);
  ITA_UP_ST(2);

  // ================= phase 2: the E/4 pixel-shuffle channels -- PixelShuffle(2): out[c][2h+i][2w+j] = in[4c+2i+j][h][w], a plain
  // regrouping of token channels, no interpolation -- as an implicit GEMM on the 18 x 34 halo of this tile, then both parts
  // meet in LDS and leave as 16-byte stores.  The halo comes from the token fragments this wave already holds (f16 hi / lo,
  // dword e of a fragment = channels (c, c + 1) of parity e): one ds_write_b32 per dword, no second trip to memory.
  // Pixel hp's 64 bytes hold its 32 channels as four 16-byte chunks, chunk q at position q ^ ((hp >> 2) & 3) (conflict-free
  // fragment reads, 2-way writes).  Phase 1's Y slabs are dead: [halo hi | halo lo] over them.
  __syncthreads();
  {
    constexpr int HP = L::HP, A_PLANE = L::A_PLANE;
    constexpr int AH = 0, AL = A_PLANE, WH = L::WH2, WL = L::WL2;
    // halo pixels outside the image (the conv's zero padding): no token writes them
    for (int i0 = tid; i0 < HP * 8; i0 += 512) {
      const int p = i0 >> 3, hy = p / 34, hx = p - 34 * hy;
      const int y = ty0 + hy - 1, x = tx0 + hx - 1;
      if (y < 0 || y >= OH || x < 0 || x >= OW) *(f32x4*)(lds + p * 64 + (i0 & 4 ? AL : AH) + (i0 & 3) * 16) = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      if (m == 0 || heavy) {                                                          // (tile B is held by both waves of a SIMD)
        const int slot = 16 * (m ? 8 + (wave & 3) : wave) + (lane & 15);
        const int r = slot / L::RW, c = slot - L::RW * r;
        const int th_ = ry0 + r, tw_ = rx0 + c;                                        // (clamped copies beyond the grid: skipped)
        const bool tok_ok = slot < L::RH * L::RW && th_ < a.TH && tw_ < a.TW;
        const int hy0 = 2 * th_ - (ty0 - 1), hx0 = 2 * tw_ - (tx0 - 1);
#pragma unroll
        for (int par = 0; par < 4; ++par) {
          const int hy = hy0 + (par >> 1), hx = hx0 + (par & 1);
          if (tok_ok && hy >= 0 && hy < 18 && hx >= 0 && hx < 34) {
            const int hp = hy * 34 + hx;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int off = hp * 64 + ((j ^ ((hp >> 2) & 3)) << 4) + 4 * (lane >> 4);
              *(int*)(lds + AH + off) = __builtin_bit_cast(i32x4, xh[m][j])[par];
              *(int*)(lds + AL + off) = __builtin_bit_cast(i32x4, xl[m][j])[par];
            }
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ITA_UP_ST(3);
    const int tx0_c = tx0, ty0_c = ty0, b_c = b;
    f32x4 acc2[4][3];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) acc2[mt][nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    const int pxl = lane & 15, kg = lane >> 4;
    // the three biases of this lane's output channels: requested here, used behind the MFMA loop (a load at the point of use
    // waits 8 us behind the other workgroups' output stores -- tools/tail_up_stamps.py)
    float bco[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) bco[nt] = a.bias[nt * 16 + pxl];
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - 3 * ky;
      f16x8 bh[3], bl[3];
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) {
        const int off = ((tap * 48 + nt * 16 + pxl) * 64) + kg * 16;
        bh[nt] = *(const f16x8*)(lds + WH + off);
        bl[nt] = *(const f16x8*)(lds + WL + off);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int hp = (2 * wave + (mt >> 1) + ky) * 34 + (mt & 1) * 16 + pxl + kx;
        const int ao = hp * 64 + ((kg ^ ((hp >> 2) & 3)) << 4);
        const f16x8 ah = *(const f16x8*)(lds + AH + ao);
        const f16x8 al = *(const f16x8*)(lds + AL + ao);
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
          acc2[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[nt], acc2[mt][nt], 0, 0, 0);
          acc2[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[nt], acc2[mt][nt], 0, 0, 0);
          acc2[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[nt], acc2[mt][nt], 0, 0, 0);
        }
      }
    }
    ITA_UP_ST(4);
    // the next tile's tokens travel during the hand-over and the stores (the fragment registers are free since the halo was written;
    // requested before the MFMA loop instead: +2 to 4 % on the launch).  Unconditional -- the last round fetches a tile again for
    // nothing: a request under a condition would make the old contents of the 64 registers live through phase 1
    set_tile(min(tile + (int)gridDim.x, ntiles - 1));
    request_tokens();
    __syncthreads();       // halo and weights consumed: the interpolated part moves into LDS as U[co][pixel] (row stride 516)
    ITA_UP_ST(8);
    float* U = (float*)lds;
#if !defined(ITA_UP_ABLATE) || ITA_UP_ABLATE != 4
#pragma unroll
    for (int g = 0; g < 12; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) U[(4 * g + i) * 516 + tid] = acc[g][i];
#else
    if (acc[0][0] == 12345.678f) U[tid] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + acc[4][0] + acc[5][0] + acc[6][0] + acc[7][0] + acc[8][0] + acc[9][0] + acc[10][0] + acc[11][0];
#endif   // (raw: a VALU temporary between the stores serialises them -- each multiply waited for the previous ds_write to have read its operand, 8 us for the 48)
    ITA_UP_ST(9);
    __syncthreads();
    ITA_UP_ST(7);
    // C layout of phase 2: lane = (output channel 16 nt + lane & 15, pixels 4 (lane >> 4) .. + 3 of M tile mt): 16-byte stores
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int row = 2 * wave + (mt >> 1), col = (mt & 1) * 16 + 4 * kg;
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) {
        const int co = nt * 16 + pxl;
        if (co < a.CO) {
          const f32x4 u = *(const f32x4*)(U + co * 516 + row * 32 + col);
          f32x4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = (u[i] * a.inv_wscale + bco[nt]) + acc2[mt][nt][i] * a.inv_wscale;
          *(f32x4*)(a.out + (((size_t)b_c * a.CO + co) * OH + ty0_c + row) * OW + tx0_c + col) = o;
        }
      }
    }
    ITA_UP_ST(5);
#ifdef ITA_UP_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ITA_UP_ST(6);
#endif
  }
  }   // tiles
}
