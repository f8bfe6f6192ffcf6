// Probe: how (and how fast) does a wavefront on another CU / XCD see a word that one wavefront updates with an agent-scope atomic?  Poll methods: plain load, agent-scope
// atomic load (sc1), system-scope atomic load (sc0 sc1), atomic fetch-add of 0.  hipcc --offload-arch=gfx950 -O3 -o visibility visibility.hip && ./visibility
#include <hip/hip_runtime.h>
#include <cstdio>
template <int M>
__device__ __forceinline__ unsigned long long poll(unsigned long long* p) {
  if (M == 0) return *(volatile unsigned long long*)p;
  if (M == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (M == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return __hip_atomic_fetch_add(p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int M, int W>
__global__ void k(unsigned long long* flag, unsigned long long* out) {  // out[b] = {xcc, t_seen - t_written (10 ns ticks), polls}
  const int b = blockIdx.x;
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u;
  if (threadIdx.x != 0) return;
  if (b == 0) {
    // touch the line first with a plain load (as a tile that cleared it earlier would have), wait, then publish
    unsigned long long x = *(volatile unsigned long long*)flag;
    for (int i = 0; i < 300; ++i) __builtin_amdgcn_s_sleep(127);
    const unsigned long long t = __builtin_amdgcn_s_memrealtime();
    if (W == 0) __hip_atomic_fetch_add(flag, t + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_exchange(flag, t + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out[3 * b] = xcc; out[3 * b + 1] = 0; out[3 * b + 2] = 0;
    return;
  }
  unsigned long long v = 0, n = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  do { v = poll<M>(flag); ++n; } while (v == 0 && __builtin_amdgcn_s_memrealtime() - t0 < 100000000ull / 100);  // give up after 10 ms
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  out[3 * b] = xcc; out[3 * b + 1] = v ? t1 - v : ~0ull; out[3 * b + 2] = n;
}
template <int M, int W>
void run(const char* name, unsigned long long* flag, unsigned long long* out) {
  hipMemset(flag, 0, 256); hipMemset(out, 0, 4096);
  hipLaunchKernelGGL((k<M, W>), dim3(32), dim3(64), 0, 0, flag, out);
  hipDeviceSynchronize();
  unsigned long long h[96]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-28s writer xcc %llu | ", name, h[0]);
  for (int b = 1; b < 32; ++b) { if (h[3 * b + 1] == ~0ull) printf("x%llu:NEVER ", h[3 * b]); else printf("x%llu:%.2fus/%llu ", h[3 * b], h[3 * b + 1] * 0.01, h[3 * b + 2]); }
  printf("\n");
}
int main() {
  unsigned long long *flag, *out;
  hipMalloc(&flag, 4096); hipMalloc(&out, 4096);
  printf("latency from the writer's atomic to the poller seeing it (us) / number of polls, per poller (x<XCC_ID>)\n");
  run<0, 0>("plain load", flag, out);
  run<1, 0>("agent atomic load (sc1)", flag, out);
  run<2, 0>("system atomic load (sc0 sc1)", flag, out);
  run<3, 0>("atomic fetch_add 0", flag, out);
  run<1, 1>("agent load, writer swaps", flag, out);
  return 0;
}
