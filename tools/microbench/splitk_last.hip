// Split-K "last arriver" epilogue against "partials + reduction in the next launch", on the folded GEMM's own shape:
// 1024 x 512 f32 result, 128 x 128 tiles (32 tiles) x 8 K slices = 256 workgroups of 256 threads, each holding a 64 KB partial.
//   write    : every workgroup stores its partial (what ita_gemm_f16x3_kernel's epilogue does today)
//   reduce   : a second launch of 256 workgroups, each summing 8 x 8 KB (what ita_lstm0_kernel does first: its 16.8 MB read)
//   last     : store, one release fence per workgroup, atomic ticket on the tile's counter; the workgroup that draws ticket 7
//              acquires, sums the tile's 8 partials (512 KB) and stores 64 KB
// MAP 0: K slice = workgroup id % 8 (the slices of a tile sit on eight XCDs: the GEMM's mapping, one slice of G0 per L2)
// MAP 1: tile = workgroup id % 32 ... all 8 slices of a tile on ONE XCD (id % 8 equal) -- the reader finds them in its own L2,
//        but each XCD's L2 would then have to hold all of G0 for its tiles
// A "compute" delay (s_sleep loop, ~20 us) in front of the epilogue spreads the arrivals as the real GEMM does.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/splitk_last tools/microbench/splitk_last.hip && /tmp/splitk_last
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void decode(int id, int map, int& tile, int& slice) {
  if (map == 0) { slice = id & 7; tile = id >> 3; }
  else { const int x = id & 7, j = id >> 3; tile = x + 8 * (j >> 3); slice = j & 7; }   // the 8 slices of a tile share id % 8
}
__device__ __forceinline__ void busy(int n) { for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(64); }

// MODE 0: write only.  MODE 1: write + last-arriver reduction.
template <int MODE>
__global__ __launch_bounds__(256) void gemm_epilogue(float* part, float* out, int* ctr, int map, int delay, int epoch) {
  __shared__ int last;
  int tile, slice; decode(blockIdx.x, map, tile, slice);
  busy(delay);
  const int tid = threadIdx.x;
  float4* mine = (float4*)(part + ((size_t)slice * 32 + tile) * 16384);
  const float v = (float)(slice + epoch);
#pragma unroll 4
  for (int i = tid; i < 4096; i += 256) mine[i] = make_float4(v, v + 1.0f, v + 2.0f, v + 3.0f);
  if (MODE == 1) {
    __syncthreads();                                          // (s_waitcnt vmcnt(0) of every wave precedes it)
    if (tid == 0) {
      const int t = __hip_atomic_fetch_add(ctr + tile, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      last = (t == 8 * epoch + 7);
    }
    __syncthreads();
    if (last) {
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      float4* o = (float4*)(out + (size_t)tile * 16384);
      for (int i = tid; i < 4096; i += 256) {
        float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float4 p = ((const float4*)(part + ((size_t)k * 32 + tile) * 16384))[i];
          s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
        }
        o[i] = s;
      }
    }
  }
}
__global__ __launch_bounds__(256) void reduce_kernel(const float* part, float* out) {
  const int id = blockIdx.x, tid = threadIdx.x;             // 256 workgroups x 2048 floats
  for (int i = tid; i < 512; i += 256) {
    float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float4 p = ((const float4*)(part + (size_t)k * 524288))[id * 512 + i];
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    ((float4*)out)[id * 512 + i] = s;
  }
}

int main() {
  const int IT = 100;
  float *part, *out; int* ctr;
  CHK(hipMalloc(&part, sizeof(float) * 8 * 524288)); CHK(hipMalloc(&out, sizeof(float) * 524288)); CHK(hipMalloc(&ctr, 4 * 32));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  float ms;
  for (int delay : {0, 40}) {
    for (int map : {0, 1}) {
      // write, then reduce in a second launch
      for (int i = 0; i < 10; ++i) { gemm_epilogue<0><<<256, 256>>>(part, out, ctr, map, delay, 0); reduce_kernel<<<256, 256>>>(part, out); }
      CHK(hipDeviceSynchronize());
      CHK(hipEventRecord(e0));
      for (int i = 0; i < IT; ++i) gemm_epilogue<0><<<256, 256>>>(part, out, ctr, map, delay, 0);
      CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
      const float t_w = ms * 1e3f / IT;
      CHK(hipEventRecord(e0));
      for (int i = 0; i < IT; ++i) { gemm_epilogue<0><<<256, 256>>>(part, out, ctr, map, delay, 0); reduce_kernel<<<256, 256>>>(part, out); }
      CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
      const float t_wr = ms * 1e3f / IT;
      // last arriver
      CHK(hipMemset(ctr, 0, 4 * 32));
      int epoch = 0;
      for (int i = 0; i < 10; ++i) gemm_epilogue<1><<<256, 256>>>(part, out, ctr, map, delay, epoch++);
      CHK(hipDeviceSynchronize());
      CHK(hipEventRecord(e0));
      for (int i = 0; i < IT; ++i) gemm_epilogue<1><<<256, 256>>>(part, out, ctr, map, delay, epoch++);
      CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
      const float t_l = ms * 1e3f / IT;
      float h[4]; CHK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
      const float want = 8.0f * (float)(epoch - 1) + 28.0f;
      printf("delay %2d map %d: write %.2f us | write + reduce launch %.2f us (reduce adds %.2f) | write + last-arriver %.2f us (adds %.2f)  check %s\n",
             delay, map, t_w, t_wr, t_wr - t_w, t_l, t_l - t_w, h[0] == want ? "ok" : "BAD");
    }
  }
  return 0;
}
