// Probe: cost of global atomics issued by lane 0 of many resident wavefronts (the reset queue's access pattern), by memory scope and address pattern.
// hipcc --offload-arch=gfx950 -O3 -o probe probe.hip && ./probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int SCOPE, bool RET>
__global__ void k_atomic(unsigned long long* p, int stride_words, int per_xcd, int iters, unsigned long long* sink) {
  const int lane = threadIdx.x & 63;
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u;
  size_t idx = (size_t)(per_xcd ? xcc * 32 : 0) + (size_t)blockIdx.x * stride_words;
  unsigned long long acc = 0;
  if (lane == 0) {
    for (int i = 0; i < iters; ++i) {
      if (RET) acc += __hip_atomic_fetch_add(p + idx, 1ull, __ATOMIC_RELAXED, SCOPE);
      else __hip_atomic_fetch_add(p + idx, 1ull, __ATOMIC_RELAXED, SCOPE);
    }
  }
  if (RET && acc == 0x123456789ull) sink[0] = acc;
}
template <class K>
float run(K kern, unsigned long long* p, int stride, int per_xcd, int iters, unsigned long long* sink, int grid) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, p, stride, per_xcd, iters, sink);
  hipEventRecord(a);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, p, stride, per_xcd, iters, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 10 * 1000;  // us per launch
}
int main() {
  unsigned long long *p, *sink;
  hipMalloc(&p, 64 << 20); hipMemset(p, 0, 64 << 20); hipMalloc(&sink, 64);
  const int grid = 4096, iters = 4;
  printf("4096 wavefronts x %d atomics from lane 0 each (us per launch; an empty launch of the same grid first)\n", iters);
  printf("empty                         %8.1f\n", run(k_atomic<__HIP_MEMORY_SCOPE_WORKGROUP, true>, p, 0, 0, 0, sink, grid));
#define ROW(name, SC) \
  printf("%-12s one address   ret %8.1f  noret %8.1f | per-XCD address ret %8.1f noret %8.1f | own line ret %8.1f noret %8.1f\n", name, \
         run(k_atomic<SC, true>, p, 0, 0, iters, sink, grid), run(k_atomic<SC, false>, p, 0, 0, iters, sink, grid), \
         run(k_atomic<SC, true>, p, 0, 1, iters, sink, grid), run(k_atomic<SC, false>, p, 0, 1, iters, sink, grid), \
         run(k_atomic<SC, true>, p + 1024, 16, 0, iters, sink, grid), run(k_atomic<SC, false>, p + 1024, 16, 0, iters, sink, grid));
  ROW("wavefront", __HIP_MEMORY_SCOPE_WAVEFRONT)
  ROW("workgroup", __HIP_MEMORY_SCOPE_WORKGROUP)
  ROW("agent", __HIP_MEMORY_SCOPE_AGENT)
  ROW("system", __HIP_MEMORY_SCOPE_SYSTEM)
  return 0;
}
