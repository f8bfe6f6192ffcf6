// Issue-rate probe: is one v_pk_{fma,mul,add}_f32 as cheap as one v_fma_f32 on gfx950 for a VALU-bound kernel (4 wavefronts per SIMD)?
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o tools/pk_probe/probe tools/pk_probe/probe.hip ; run on the GPU box.
// (-fno-slp-vectorize is essential: without it the SLP vectoriser turns the "scalar" modes into v_pk_* code as well -- the round-4 run of this probe compared packed with
// packed: SQ_INSTS_VALU of mode 0 was HALF of 16 x iterations x wavefronts, profiles/r05_valu_calibration.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  float r[16];
  for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 1e-3f + i;
  f2 p[8];
  for (int i = 0; i < 8; ++i) p[i] = (f2){r[2 * i], r[2 * i + 1]};
  const f2 a2 = {a, a}, b2 = {b, b};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) r[i] = __builtin_fmaf(r[i], a, b);
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], a2, b2);
    } else if (MODE == 2) {  // mul + add, scalar
#pragma unroll
      for (int i = 0; i < 16; ++i) { r[i] = r[i] * a; r[i] = r[i] + b; }
    } else if (MODE == 3) {  // mul + add, packed
#pragma unroll
      for (int i = 0; i < 8; ++i) { p[i] = p[i] * a2; p[i] = p[i] + b2; }
    } else if (MODE == 4) {  // scalar fma, three DIFFERENT vector operands (what real code looks like: no operand is a loop constant)
#pragma unroll
      for (int i = 0; i < 16; ++i) r[i] = __builtin_fmaf(r[i], r[(i + 5) & 15], r[(i + 11) & 15]);
    } else if (MODE == 5) {  // packed fma, three different register pairs
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], p[(i + 3) & 7], p[(i + 5) & 7]);
    } else if (MODE == 6) {  // scalar: one splat operand + two different vector operands (a segment's direction x a per-query value + a per-query value)
#pragma unroll
      for (int i = 0; i < 16; ++i) r[i] = __builtin_fmaf(r[i], a, r[(i + 11) & 15]);
    } else {                 // packed: one operand read through op_sel from one half of a pair, two different pairs
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], (f2){p[(i + 3) & 7].x, p[(i + 3) & 7].x}, p[(i + 5) & 7]);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += r[i];
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> float run(float* d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256 * 4, 256>>>(d, 16, 1.0001f, 0.5f);
  hipEventRecord(e0);
  k<MODE><<<256 * 4, 256>>>(d, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* d; hipMalloc(&d, 256 * 4 * 256 * 4);
  const int iters = 20000;
  const float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters), t3 = run<3>(d, iters);
  // 16 fp32 fma-equivalents per iteration and lane
  printf("scalar fma %.3f ms | packed fma %.3f ms (x%.2f) | scalar mul+add %.3f ms | packed mul+add %.3f ms (x%.2f)\n", t0, t1, t0 / t1, t2, t3, t2 / t3);
  const double waves = 256.0 * 4 * 4, inst0 = 16.0 * iters;
  printf("cycles per scalar VALU instruction per SIMD (2.4 GHz, 4 waves/SIMD): %.2f ; per packed instruction: %.2f\n",
         t0 * 1e-3 * 2.4e9 / (inst0 * waves / 1024), t1 * 1e-3 * 2.4e9 / (inst0 / 2 * waves / 1024));
  const float t4 = run<4>(d, iters), t5 = run<5>(d, iters), t6 = run<6>(d, iters), t7 = run<7>(d, iters);
  printf("three different vector operands: scalar fma %.3f ms (%.2f cycles per instruction) | packed fma %.3f ms (%.2f cycles per instruction, x%.2f)\n", t4,
         t4 * 1e-3 * 2.4e9 / (inst0 * waves / 1024), t5, t5 * 1e-3 * 2.4e9 / (inst0 / 2 * waves / 1024), t4 / t5);
  printf("one splat + two different operands: scalar fma %.3f ms (%.2f) | packed fma with op_sel %.3f ms (%.2f, x%.2f)\n", t6,
         t6 * 1e-3 * 2.4e9 / (inst0 * waves / 1024), t7, t7 * 1e-3 * 2.4e9 / (inst0 / 2 * waves / 1024), t6 / t7);
  return 0;
}
