// Does the SHAPE of a wave-load matter for how fast one CU pulls L2-resident data?  Every workgroup (512 threads, one
// per CU) reads the same 1 MB region (so it is L2-hot) as 16-byte pieces, 8 loads in flight per lane, in four shapes:
//   contiguous : a wave-load = 1 KB contiguous (8 consecutive lines)
//   rows128    : a wave-load = 8 lanes per 128-byte row segment, rows LD bytes apart (the GEMM's k-tile staging)
//   rows32     : a wave-load = 2 lanes per row (32 bytes of a line), 32 rows (MFMA-fragment loads from a row-major matrix)
//   rows128dma : rows128 through LDS-DMA (global_load_lds_dwordx4) into a 64 KB LDS ring instead of registers
// hipcc --offload-arch=gfx950 -O3 -o tools/microbench/cu_fill_rows.bin tools/microbench/cu_fill_rows.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int LD = 16512;   // row pitch in bytes (8256 halves, as the x2 planes)

template <int MODE>
__global__ __launch_bounds__(512) void fill_kernel(const char* __restrict__ base, int iters, int* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  i32x4 acc = {0, 0, 0, 0};
  // the region: 64 rows x 16 KB (row pitch LD); iteration it reads k-tile it of 128 bytes per row ... x 8 waves x 8 loads
  for (int it = 0; it < iters; ++it) {
    i32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int piece = (j * 8 + wave) * 64 + lane;   // 4096 pieces of 16 B = 64 KB per iteration
      const char* src;
      if (MODE == 0) src = base + ((size_t)(it & 15) * 4096 + piece) * 16;                                  // contiguous
      else if (MODE == 1 || MODE == 3) src = base + (size_t)(piece >> 3) * LD + (it & 127) * 128 + (piece & 7) * 16;   // 512 rows x 128 B
      else src = base + (size_t)(piece >> 1) * (LD / 4) + (it & 31) * 32 + (piece & 1) * 16;              // 2048 "rows" x 32 B
      if (MODE == 3) {
        char* dst = lds + ((j * 8 + wave) * 64) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      } else {
        v[j] = *(const i32x4*)src;
      }
    }
    if (MODE == 3) { __builtin_amdgcn_s_waitcnt(0x0F70); acc.x ^= *(const int*)(lds + tid * 4); }
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc ^= v[j];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) out[blockIdx.x] = 1;
}

int main() {
  const int IT = 50, ITERS = 64;   // 64 iterations x 64 KB = 4 MB per workgroup and launch
  char* buf; int* out;
  const size_t bytes = (size_t)LD * 2048 + (1 << 20);
  CHK(hipMalloc(&buf, bytes)); CHK(hipMemset(buf, 1, bytes)); CHK(hipMalloc(&out, 4096));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  CHK(hipFuncSetAttribute((const void*)fill_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  const char* names[4] = {"contiguous", "rows128", "rows32", "rows128dma"};
  for (int nwg : {64, 256})
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0, 0));
        for (int it = 0; it < IT; ++it) {
          if (mode == 0) hipLaunchKernelGGL(fill_kernel<0>, dim3(nwg), dim3(512), 0, 0, buf, ITERS, out);
          if (mode == 1) hipLaunchKernelGGL(fill_kernel<1>, dim3(nwg), dim3(512), 0, 0, buf, ITERS, out);
          if (mode == 2) hipLaunchKernelGGL(fill_kernel<2>, dim3(nwg), dim3(512), 0, 0, buf, ITERS, out);
          if (mode == 3) hipLaunchKernelGGL(fill_kernel<3>, dim3(nwg), dim3(512), 65536, 0, buf, ITERS, out);
        }
        CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      const float us = best * 1000 / IT;
      printf("%3d workgroups, %-11s: %7.2f us per launch -> %6.1f GB/s per CU\n", nwg, names[mode], us, ITERS * 65536.0 / (us - 2.5) / 1e3);
    }
  return 0;
}
