// Cost of a frame-group sync (G workgroups meet on a counter in global memory: release, atomic add, bounded
// spin, acquire) against a kernel boundary, on the shapes the LSTM chain has: grid (G, 32) x 512 threads, each
// workgroup writes 4 KB and, after the sync, reads the 32 KB its G partners wrote.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/fg_sync tools/microbench/fg_sync.hip && /tmp/fg_sync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: every thread fences (release before, acquire after).  MODE 1: one thread per workgroup fences; the other
// waves' stores are ordered before it by the workgroup barrier (s_waitcnt vmcnt(0) + s_barrier), and its cache
// invalidate serves the whole CU.
template <int MODE>
__device__ __forceinline__ bool fg_sync(int* ctr, int target, int* flag) {
  if (MODE == 0) __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int n = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++n < (1 << 22)) __builtin_amdgcn_s_sleep(1);
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    *flag = n < (1 << 22);
  }
  __syncthreads();
  if (MODE == 0) __threadfence();
  return *flag != 0;
}

// phases = number of write -> sync -> read rounds inside one launch
template <int MODE>
__global__ __launch_bounds__(512) void chain_kernel(float* buf, int* ctr, int G, int phases, int epoch, int* err) {
  __shared__ int flag;
  const int p = blockIdx.x, fg = blockIdx.y, tid = threadIdx.x;
  float acc = 0.0f;
  for (int ph = 0; ph < phases; ++ph) {
    float* mine = buf + ((size_t)(ph * gridDim.y + fg) * G + p) * 1024;
    for (int i = tid; i < 1024; i += 512) mine[i] = acc + (float)(p + i + epoch);
    if (!fg_sync<MODE>(ctr + fg * 4 + ph, G * (epoch + 1), &flag)) { if (tid == 0) atomicAdd(err, 1); return; }
    const float* all = buf + (size_t)(ph * gridDim.y + fg) * G * 1024;
    for (int i = tid; i < G * 1024; i += 512) acc += all[i];
  }
  if (acc == 123.456f) buf[0] = acc;
}
__global__ __launch_bounds__(512) void phase_kernel(float* buf, int G, int ph, int epoch) {
  const int p = blockIdx.x, fg = blockIdx.y, tid = threadIdx.x;
  float acc = 0.0f;
  if (ph > 0) {
    const float* all = buf + (size_t)((ph - 1) * gridDim.y + fg) * G * 1024;
    for (int i = tid; i < G * 1024; i += 512) acc += all[i];
  }
  float* mine = buf + ((size_t)(ph * gridDim.y + fg) * G + p) * 1024;
  for (int i = tid; i < 1024; i += 512) mine[i] = acc + (float)(p + i + epoch);
}
__global__ __launch_bounds__(512) void empty_kernel(float* buf) { if (buf == nullptr) buf[0] = 1.0f; }

int main() {
  const int NFG = 32, IT = 200;
  float* buf; int *ctr, *err;
  CHK(hipMalloc(&buf, sizeof(float) * 4 * NFG * 16 * 1024));
  CHK(hipMalloc(&ctr, sizeof(int) * NFG * 4)); CHK(hipMalloc(&err, 4));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipStream_t s; CHK(hipStreamCreate(&s));
  for (int G : {4, 8, 16}) {
    for (int mode : {0, 1})
    for (int phases : {0, 1, 3}) {
      CHK(hipMemset(ctr, 0, sizeof(int) * NFG * 4)); CHK(hipMemset(err, 0, 4));
      // the counters count up across launches (target G * (epoch + 1)): no reset needed in the benchmark
      for (int rep = 0; rep < 2; ++rep) {
        CHK(hipEventRecord(e0, s));
        for (int it = 0; it < IT; ++it) {
          if (mode == 0) hipLaunchKernelGGL(chain_kernel<0>, dim3(G, NFG), dim3(512), 0, s, buf, ctr, G, phases, rep * IT + it, err);
          else hipLaunchKernelGGL(chain_kernel<1>, dim3(G, NFG), dim3(512), 0, s, buf, ctr, G, phases, rep * IT + it, err);
        }
        CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1));
      }
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
      int herr; CHK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      printf("G=%2d fences by %s, %d syncs : %7.2f us per launch   (spin timeouts: %d)\n", G, mode ? "one thread " : "all threads", phases, ms * 1000 / IT, herr);
    }
    for (int rep = 0; rep < 2; ++rep) {
      CHK(hipEventRecord(e0, s));
      for (int it = 0; it < IT; ++it)
        for (int ph = 0; ph < 4; ++ph) hipLaunchKernelGGL(phase_kernel, dim3(G, NFG), dim3(512), 0, s, buf, G, ph, it);
      CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1));
    }
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("G=%2d four separate kernels   : %7.2f us per chain\n", G, ms * 1000 / IT);
  }
  CHK(hipEventRecord(e0, s));
  for (int it = 0; it < IT; ++it) hipLaunchKernelGGL(empty_kernel, dim3(8, NFG), dim3(512), 0, s, buf);
  CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  printf("empty kernel (8 x 32 x 512)   : %7.2f us per launch\n", ms * 1000 / IT);
  return 0;
}
