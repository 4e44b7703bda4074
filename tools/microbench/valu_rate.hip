// Microbenchmark (GPU box): cycles per VALU wave-instruction on one SIMD, for 1, 2 and 4 waves per SIMD, per opcode.
// Answers: is the requantisation mix (v_cvt_f32_i32, v_mul_f32 / v_pk_mul_f32, v_med3_f32, v_add_f32_sdwa) issued
// at one instruction per 2 or per 4 cycles when two waves share a SIMD?
// build: hipcc --offload-arch=gfx950 -O2 tools/microbench/valu_rate.hip -o tools/microbench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void k(unsigned long long* out, float* sink, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float m = 0.999f, lo = -128.0f, hi = 127.0f, mg = 12582912.0f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (OP == 0) {   // v_fma_f32
      asm volatile(REP16("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(lo));
    } else if constexpr (OP == 1) {   // v_med3_f32
      asm volatile(REP16("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n"
                   "v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(lo), "v"(hi));
    } else if constexpr (OP == 2) {   // v_cvt_f32_i32
      asm volatile(REP16("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3\n"
                   "v_cvt_f32_i32 %4, %4\n v_cvt_f32_i32 %5, %5\n v_cvt_f32_i32 %6, %6\n v_cvt_f32_i32 %7, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (OP == 3) {   // v_add_f32_sdwa dst_sel byte
      asm volatile(REP16("v_add_f32_sdwa %0, %1, %8 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
                   "v_add_f32_sdwa %2, %3, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
                   "v_add_f32_sdwa %4, %5, %8 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
                   "v_add_f32_sdwa %6, %7, %8 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
                   "v_add_f32_sdwa %0, %3, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
                   "v_add_f32_sdwa %2, %5, %8 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
                   "v_add_f32_sdwa %4, %7, %8 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
                   "v_add_f32_sdwa %6, %1, %8 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(mg));
    } else if constexpr (OP == 4) {   // v_pk_mul_f32 (counts as ONE instruction = two multiplies)
      asm volatile(REP16("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                   "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n")
                   : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(const double*)&m));
    } else if constexpr (OP == 5) {   // v_mul_f32
      asm volatile(REP16("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                   "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
    } else if constexpr (OP == 6) {   // v_rndne_f32
      asm volatile(REP16("v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3\n"
                   "v_rndne_f32 %4, %4\n v_rndne_f32 %5, %5\n v_rndne_f32 %6, %6\n v_rndne_f32 %7, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (OP == 7) {   // v_pk_add_u16 (packed 16-bit integer)
      asm volatile(REP16("v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n"
                   "v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
    } else if constexpr (OP == 8) {   // v_perm_b32
      asm volatile(REP16("v_perm_b32 %0, %0, %1, %8\n v_perm_b32 %1, %1, %2, %8\n v_perm_b32 %2, %2, %3, %8\n v_perm_b32 %3, %3, %4, %8\n"
                   "v_perm_b32 %4, %4, %5, %8\n v_perm_b32 %5, %5, %6, %8\n v_perm_b32 %6, %6, %7, %8\n v_perm_b32 %7, %7, %0, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP>
void run(const char* name) {
  unsigned long long* d;
  float* sink;
  hipMalloc(&d, 256 * 16 * 8);
  hipMalloc(&sink, 256 * 1024 * 4);
  const int iters = 200, per_iter = 128;
  printf("%-16s", name);
  for (int threads : {256, 512, 1024}) {
    k<OP><<<256, threads>>>(d, sink, iters);
    k<OP><<<256, threads>>>(d, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> v;
    for (int b = 0; b < 256; ++b)
      for (int w = 0; w < threads / 64; ++w) v.push_back((double)h[b * 16 + w]);
    std::sort(v.begin(), v.end());
    const double med = v[v.size() / 2];
    const int wps = threads / 256;
    // cycles per wave-instruction as seen by one wave, and per SIMD (wps waves share it)
    printf("  %d w/SIMD: %.2f cyc/instr/wave = %.2f cyc/instr/SIMD", wps, med / (iters * per_iter), med / (iters * per_iter) / wps);
  }
  printf("\n");
  hipFree(d); hipFree(sink);
}

int main() {
  run<0>("v_fma_f32");
  run<5>("v_mul_f32");
  run<1>("v_med3_f32");
  run<2>("v_cvt_f32_i32");
  run<3>("v_add_f32_sdwa");
  run<4>("v_pk_mul_f32");
  run<6>("v_rndne_f32");
  run<7>("v_pk_add_u16");
  run<8>("v_perm_b32");
  return 0;
}
