// Cycles per requantisation of sixteen accumulators (one rq_group call of the encoder kernel), by form, with 1 / 2 waves per
// SIMD, nothing else in the loop: round-2 float clamp (rq_pack16_b), round-3 exact (rq_pack16_v3<EXACT>), single rounding
// (<FAST>), ReLU (<RELU>), and the logits forms.  hipcc --offload-arch=gfx950 -O3 -I ../../drone-oa-iree-vit-accelerator_amd/csrc rq_seq.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "ita_device.h"
template <int OP>
__global__ void k(unsigned long long* out, int* sink, int iters, float mult) {
  i32x4 acc[4];
  for (int t = 0; t < 4; ++t) acc[t] = (i32x4){ITA_ACC_BIAS + (int)threadIdx.x * 7 + t, ITA_ACC_BIAS - (int)threadIdx.x * 3, ITA_ACC_BIAS + 100 * t, ITA_ACC_BIAS + 5};
  i32x4 r = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      i32x4 p;
      if constexpr (OP == 0) p = rq_pack16_b(acc, mult, -128.0f);
      if constexpr (OP == 1) p = rq_pack16_v3<ITA_RQ_EXACT>(acc, mult);
      if constexpr (OP == 2) p = rq_pack16_v3<ITA_RQ_FAST>(acc, mult);
      if constexpr (OP == 3) p = rq_pack16_v3<ITA_RQ_RELU>(acc, mult);
      if constexpr (OP == 4) { unsigned b[8]; i32x4 a2[2] = {acc[0], acc[1]}; lg8_b(a2, mult, b); p = (i32x4){(int)(b[0] ^ b[1]), (int)(b[2] ^ b[3]), (int)(b[4] ^ b[5]), (int)(b[6] ^ b[7])}; }
      if constexpr (OP == 5) { unsigned w[4]; i32x4 a2[2] = {acc[0], acc[1]}; lg8_v3<ITA_RQ_EXACT>(a2, mult, w); p = (i32x4){(int)w[0], (int)w[1], (int)w[2], (int)w[3]}; }
      if constexpr (OP == 6) { unsigned w[4]; i32x4 a2[2] = {acc[0], acc[1]}; lg8_v3<ITA_RQ_FAST>(a2, mult, w); p = (i32x4){(int)w[0], (int)w[1], (int)w[2], (int)w[3]}; }
      r ^= p;
      // next call's accumulators depend on this result (no hoisting), stay inside the biased range
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = (acc[t] & 0xffff00ff) | ((p & 0x0f) << 8);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = r[0] ^ r[1] ^ r[2] ^ r[3];
}
template <int OP>
void run(const char* name) {
  unsigned long long* d; int* sink;
  (void)hipMalloc(&d, 256 * 16 * 8); (void)hipMalloc(&sink, 256 * 1024 * 4);
  const int iters = 200;
  printf("%-34s", name);
  for (int threads : {256, 512, 1024}) {
    k<OP><<<256, threads>>>(d, sink, iters, 1.3e-3f);
    k<OP><<<256, threads>>>(d, sink, iters, 1.3e-3f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> v;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) v.push_back((double)h[b * 16 + w]);
    std::sort(v.begin(), v.end());
    const int wps = threads / 256;
    printf("  %dw/SIMD: %6.1f cyc/call/wave (%6.1f per SIMD)", wps, v[v.size() / 2] / (iters * 4), v[v.size() / 2] / (iters * 4) / wps);
  }
  printf("\n");
  (void)hipFree(d); (void)hipFree(sink);
}
int main() {
  run<0>("round 2: float clamp, 16 values");
  run<1>("round 3 exact, 16 values");
  run<2>("round 3 fast, 16 values");
  run<3>("round 3 relu, 16 values");
  run<4>("logits round 2, 8 values");
  run<5>("logits round 3 exact, 8 values");
  run<6>("logits round 3 fast, 8 values");
  return 0;
}
