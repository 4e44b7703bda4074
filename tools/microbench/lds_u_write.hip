// The config-5 tail's U hand-over on its own: 512 threads, each stores 48 floats to U[k][516] (k = 0..47, column = thread)
// with ds_write_b32, then reads 12 x 16 bytes in the transposed assignment.  Cycles per pass from s_memtime, one workgroup per CU.
// Variants: 0 = ds_write_b32 as in the kernel; 1 = same addresses below 64 KB only (row stride 260); 2 = ds_write_b128 of a
// [pixel][52] layout.   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/lds_u_write.bin tools/microbench/lds_u_write.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int V>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(float* out, unsigned long long* cyc, int reps) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* U = (float*)lds;
  const int tid = threadIdx.x;
  float a[48];
#pragma unroll
  for (int i = 0; i < 48; ++i) a[i] = (float)(tid + i);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    if (V == 0) {
#pragma unroll
      for (int i = 0; i < 48; ++i) U[i * 516 + tid] = a[i];
    } else if (V == 1) {
#pragma unroll
      for (int i = 0; i < 48; ++i) U[(i & 15) * 516 + tid + 8192 * 0] = a[i];
    } else {
#pragma unroll
      for (int g = 0; g < 12; ++g) *(f32x4*)(U + tid * 52 + 4 * g) = (f32x4){a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 48; ++i) a[i] += 1.0f;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  float s = 0;
#pragma unroll
  for (int i = 0; i < 48; ++i) s += a[i];
  out[blockIdx.x * 512 + tid] = s + U[tid];
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  unsigned long long h[256];
  const int reps = 200;
  for (int v = 0; v < 3; ++v) {
    for (int it = 0; it < 2; ++it) {
      if (v == 0) { hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 160 * 1024, 0, out, cyc, reps); }
      if (v == 1) { hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 160 * 1024, 0, out, cyc, reps); }
      if (v == 2) { hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 160 * 1024, 0, out, cyc, reps); }
      hipDeviceSynchronize();
    }
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
    printf("variant %d: %.0f s_memtime ticks per pass (48 stores per thread + barrier), error %s\n", v, s / 256 / reps, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
