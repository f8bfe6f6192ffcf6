// Probe for the split-fp16 matrix path (sigmaenv_mlp32.inc, "split" mode) on gfx950:
//  (1) lane maps of v_mfma_f32_32x32x16_f16: C[32x32] = A[32x16] * B[16x32]
//  (2) are fp16 SUBNORMAL A / B inputs preserved by the matrix pipe?
//  (3) error of the three-product split (hi hi + hi lo + lo hi, operands scaled by 2^8) against fp64 on K = 256 dot products
// hipcc --offload-arch=gfx950 -O2 -o /tmp/probe_f16_split tools/mfma_probe/probe_f16_split.hip && /tmp/probe_f16_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(const float* A, const float* B, float* C, float sa, float sb) {
  const int l = threadIdx.x;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (_Float16)(A[(l & 31) * 16 + 8 * (l >> 5) + j] * sa);   // A[m][k]
    b[j] = (_Float16)(B[(8 * (l >> 5) + j) * 32 + (l & 31)] * sb); // B[k][n]
  }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

// split product over K = 256: every lane owns rows of W and columns of X through the same maps, 16 k blocks
__global__ void split_dot(const float* W, const float* X, float* out, int K, float S) {
  const int l = threadIdx.x;
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  for (int kb = 0; kb < K / 16; ++kb) {
    f16x8 wh, wl, xh, xl;
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * kb + 8 * (l >> 5) + j;
      const float w = W[(l & 31) * K + k] * S, x = X[k * 32 + (l & 31)] * S;
      const _Float16 h1 = (_Float16)w, h2 = (_Float16)x;
      wh[j] = h1; wl[j] = (_Float16)(w - (float)h1);
      xh[j] = h2; xl[j] = (_Float16)(x - (float)h2);
    }
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, c, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r] / (S * S);
}

int main() {
  std::vector<float> A(32 * 16), B(16 * 32), C(1024), R(1024, 0.f);
  for (auto& v : A) v = (float)((rand() % 17) - 8) / 8.f;
  for (auto& v : B) v = (float)((rand() % 13) - 6) / 4.f;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) R[i * 32 + j] += A[i * 16 + k] * B[k * 32 + j];
  float *dA, *dB, *dC;
  hipMalloc(&dA, 256 * 32 * 4); hipMalloc(&dB, 256 * 32 * 4); hipMalloc(&dC, 4096);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dC, 1.f, 1.f);
  hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
  double e = 0; for (int i = 0; i < 1024; ++i) e = fmax(e, fabs(C[i] - R[i]));
  printf("layout: max err %g (C[0][1]=%g ref %g, C[1][0]=%g ref %g)\n", e, C[1], R[1], C[32], R[32]);
  // subnormal A: scale A by 2^-17 (|A| <= 1 -> fp16 values below 2^-14 are subnormal), B by 2^10: exact result = R * 2^-7 if preserved, 0 if flushed
  probe<<<1, 64>>>(dA, dB, dC, ldexpf(1.f, -17), ldexpf(1.f, 10));
  hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
  e = 0; double mx = 0; for (int i = 0; i < 1024; ++i) { e = fmax(e, fabs(C[i] - R[i] / 128.0)); mx = fmax(mx, fabs(C[i])); }
  printf("subnormal A: max err %g, max |C| %g (ref max %g) -> %s\n", e, mx, 0.0, mx == 0 ? "FLUSHED" : (e < 1e-6 ? "preserved" : "partially lost"));
  probe<<<1, 64>>>(dA, dB, dC, ldexpf(1.f, 10), ldexpf(1.f, -17));
  hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
  e = 0; mx = 0; for (int i = 0; i < 1024; ++i) { e = fmax(e, fabs(C[i] - R[i] / 128.0)); mx = fmax(mx, fabs(C[i])); }
  printf("subnormal B: max err %g, max |C| %g -> %s\n", e, mx, mx == 0 ? "FLUSHED" : (e < 1e-6 ? "preserved" : "partially lost"));
  // split accuracy
  const int K = 256;
  std::vector<float> W(32 * K), X(K * 32), O(1024);
  srand(7);
  for (auto& v : W) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.0625f;
  for (auto& v : X) v = tanhf(((float)rand() / RAND_MAX * 2.f - 1.f) * 2.f);
  hipMemcpy(dA, W.data(), W.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  for (float S : {1.f, 256.f}) {
    split_dot<<<1, 64>>>(dA, dB, dC, K, S);
    hipMemcpy(O.data(), dC, 4096, hipMemcpyDeviceToHost);
    double es = 0, ef = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double r = 0; float f = 0.f;
      for (int k = 0; k < K; ++k) { r += (double)W[i * K + k] * X[k * 32 + j]; f = fmaf(W[i * K + k], X[k * 32 + j], f); }
      es = fmax(es, fabs(O[i * 32 + j] - r)); ef = fmax(ef, fabs(f - r));
    }
    printf("split (scale %g): max |err| vs fp64 %.3g; fp32 fma chain %.3g\n", S, es, ef);
  }
  return 0;
}
