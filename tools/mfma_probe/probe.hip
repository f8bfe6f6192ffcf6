// Layout probe for v_mfma_f32_16x16x32_bf16 on gfx950: C[16x16] = A[16x32] * B[32x16] under the hypothesised lane maps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* A, const float* B, float* C) {
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (__bf16)A[(l % 16) * 32 + 8 * (l / 16) + j];      // A[row][k]
    b[j] = (__bf16)B[(8 * (l / 16) + j) * 16 + (l % 16)];    // B[k][col]
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[(4 * (l / 16) + r) * 16 + (l % 16)] = c[r];
}
int main() {
  std::vector<float> A(16 * 32), B(32 * 16), C(256), R(256, 0.f);
  for (auto& v : A) v = (float)((rand() % 17) - 8) / 8.f;
  for (auto& v : B) v = (float)((rand() % 13) - 6) / 4.f;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) R[i * 16 + j] += A[i * 32 + k] * B[k * 16 + j];
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dC);
  hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
  double e = 0; for (int i = 0; i < 256; ++i) e = fmax(e, fabs(C[i] - R[i]));
  printf("max err %g (C[0][1]=%g ref %g, C[1][0]=%g ref %g)\n", e, C[1], R[1], C[16], R[16]);
  return 0;
}
