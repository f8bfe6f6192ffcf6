// Issue cost of VALU instruction FORMS on gfx950 with 4 wavefronts per SIMD (the step kernel's occupancy): 16 independent instructions of one form per loop iteration,
// written in inline asm so that the form is exactly what is named.  The step kernel is VALU-issue-bound: this table says which forms are cheap.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/inst_probe/probe tools/inst_probe/probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define KERNEL(NAME, ASM32)                                                                                             \
  __global__ void __launch_bounds__(256) NAME(float* out, int iters, float sa, float sb) {                              \
    float r[16], q[16];                                                                                                 \
    for (int i = 0; i < 16; ++i) { r[i] = threadIdx.x * 1e-3f + i + 1.0f; q[i] = 1.0f + 1e-4f * i; }                     \
    unsigned long long m = (unsigned long long)threadIdx.x * 0x9E3779B97F4A7C15ull | 1ull;                              \
    for (int it = 0; it < iters; ++it) {                                                                                \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) { ASM32 }                                                          \
    }                                                                                                                   \
    float s = 0;                                                                                                        \
    for (int i = 0; i < 16; ++i) s += r[i] + q[i];                                                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)m;                                                          \
  }
#define J ((i + 5) & 15)
#define L ((i + 11) & 15)
KERNEL(k_fma_vvv, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(q[J]), "v"(q[L]));)
KERNEL(k_fma_vsv, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "s"(sa), "v"(q[L]));)
KERNEL(k_fma_vvc, asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(r[i]) : "v"(q[J]));)
KERNEL(k_mul_vv, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(q[J]));)
KERNEL(k_mul_sv, asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r[i]) : "s"(sa));)
KERNEL(k_mul_lit, asm volatile("v_mul_f32 %0, 0x3f8020c5, %0" : "+v"(r[i]));)
KERNEL(k_add_vv, asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(q[J]));)
KERNEL(k_fmac_vv, asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[i]) : "v"(q[J]), "v"(q[L]));)
KERNEL(k_max_vv, asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(q[J]));)
KERNEL(k_min3, asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(q[J]), "v"(q[L]));)
KERNEL(k_pk_fma, asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double*)&r[i & 14]) : "v"(*(double*)&q[J & 14]), "v"(*(double*)&q[L & 14]));)
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(q[J]));)
KERNEL(k_mov_dpp, asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r[i]) : "v"(q[J]));)
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(q[J]) : );)
KERNEL(k_cmp, asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(r[i]), "v"(q[J]) : "vcc");)
KERNEL(k_cmp_s, asm volatile("v_cmp_gt_f32 vcc, %1, %0" : : "v"(r[i]), "s"(sa) : "vcc");)
KERNEL(k_add_u32, asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(q[J]));)
KERNEL(k_add_u32_s, asm volatile("v_add_u32 %0, %1, %0" : "+v"(r[i]) : "s"(sa));)
KERNEL(k_lshl_add, asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(r[i]) : "v"(q[J]));)
KERNEL(k_and, asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(q[J]));)
KERNEL(k_bfe, asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(r[i]));)
KERNEL(k_mul_lo, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(q[J]));)
KERNEL(k_mul_hi, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[i]) : "v"(q[J]));)
KERNEL(k_mul_u24, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(q[J]));)
KERNEL(k_mad_u24, asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r[i]) : "v"(q[J]), "v"(q[L]));)
KERNEL(k_mad_u64, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(double*)&r[i & 14]) : "v"(q[J]), "v"(q[L]) : "vcc");)
KERNEL(k_lshl_add_u64, asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(*(double*)&r[i & 14]) : "v"(*(double*)&q[J & 14]));)
KERNEL(k_fma_f64, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(*(double*)&r[i & 14]) : "v"(*(double*)&q[J & 14]), "v"(*(double*)&q[L & 14]));)
KERNEL(k_fma_f64_s, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(*(double*)&r[i & 14]) : "v"(*(double*)&q[J & 14]), "s"(*(double*)&m));)
KERNEL(k_mul_f64, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(*(double*)&r[i & 14]) : "v"(*(double*)&q[J & 14]));)
KERNEL(k_add_f64, asm volatile("v_add_f64 %0, %0, %1" : "+v"(*(double*)&r[i & 14]) : "v"(*(double*)&q[J & 14]));)
KERNEL(k_cvt_f64_f32, asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(*(double*)&r[i & 14]) : "v"(q[J]));)
KERNEL(k_cvt_f32_f64, asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r[i]) : "v"(*(double*)&q[J & 14]));)
KERNEL(k_rcp, asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));)
KERNEL(k_sqrt, asm volatile("v_sqrt_f32 %0, %0" : "+v"(r[i]));)
KERNEL(k_rcp_f64, asm volatile("v_rcp_f64 %0, %0" : "+v"(*(double*)&r[i & 14]));)
KERNEL(k_readfirstlane, { int t_; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(t_) : "v"(r[i])); asm volatile("" :: "s"(t_)); })
KERNEL(k_salu, { int t_ = i; asm volatile("s_add_u32 %0, %0, 3" : "+s"(t_)); asm volatile("" :: "s"(t_)); })
KERNEL(k_fma_plus_salu, { int t_ = i; asm volatile("v_fma_f32 %0, %0, %2, %3\n s_add_u32 %1, %1, 3" : "+v"(r[i]), "+s"(t_) : "v"(q[J]), "v"(q[L])); })
template <class K>
static void run(const char* name, K kern, float* d, int per_iter) {
  const int iters = 20000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(1024), dim3(256), 0, 0, d, 16, 1.0001f, 0.5f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(1024), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double inst_per_simd = (double)per_iter * iters * 4.0;  // 4 wavefronts per SIMD
  printf("%-34s %8.3f ms  %6.2f cycles of 2.4 GHz per instruction per SIMD\n", name, ms, ms * 1e-3 * 2.4e9 / inst_per_simd);
}
int main() {
  float* d; (void)hipMalloc(&d, 1024 * 256 * 4);
#define RUN(k) run(#k, k, d, 16);
  RUN(k_fma_vvv) RUN(k_fma_vsv) RUN(k_fma_vvc) RUN(k_mul_vv) RUN(k_mul_sv) RUN(k_mul_lit) RUN(k_add_vv) RUN(k_fmac_vv) RUN(k_max_vv) RUN(k_min3) RUN(k_pk_fma)
  RUN(k_mov) RUN(k_mov_dpp) RUN(k_cndmask) RUN(k_cmp) RUN(k_cmp_s) RUN(k_add_u32) RUN(k_add_u32_s) RUN(k_lshl_add) RUN(k_and) RUN(k_bfe)
  RUN(k_mul_lo) RUN(k_mul_hi) RUN(k_mul_u24) RUN(k_mad_u24) RUN(k_mad_u64) RUN(k_lshl_add_u64)
  RUN(k_fma_f64) RUN(k_fma_f64_s) RUN(k_mul_f64) RUN(k_add_f64) RUN(k_cvt_f64_f32) RUN(k_cvt_f32_f64) RUN(k_rcp) RUN(k_sqrt) RUN(k_rcp_f64)
  RUN(k_readfirstlane) RUN(k_salu) RUN(k_fma_plus_salu)
  return 0;
}
