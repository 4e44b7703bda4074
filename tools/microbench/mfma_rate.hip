// What a CU's f16 MFMA pipe delivers in wall-clock time (not cycles): bursts of the length of the folded GEMM and long runs.
// One workgroup per CU, WAVES waves of 64 lanes, each wave a loop of `iters` x 8 v_mfma_f32_32x32x16_f16 on two accumulators,
// operands in registers.  hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters, long long* clk) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * (threadIdx.x - j)); }
  f32x16 c0, c1;
  for (int e = 0; e < 16; ++e) { c0[e] = 0.0f; c1[e] = 0.0f; }
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  float s = 0.0f;
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e];
  if (s == 123.456f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  float* out; long long* clk;
  hipMalloc(&out, 4); hipMalloc(&clk, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves : {4, 8}) {
    for (int iters : {48, 480, 48000}) {   // 48 x 8 = 384 MFMAs per wave = the folded GEMM's count per wave (16 k-tiles x 24)
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop, dim3(cus), dim3(64 * waves), 0, 0, out, iters, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        const double flop = 2.0 * 32 * 32 * 16 * 8.0 * iters * waves * cus;
        printf("waves/CU %d  iters %6d  %9.3f us  %7.1f TFLOP/s   clock64 delta %lld (%.1f per MFMA per wave)\n", waves, iters,
               ms * 1e3, flop / (ms * 1e-3) / 1e12, c, (double)c / (8.0 * iters));
      }
    }
  }
  return 0;
}
