#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float floatx16;

__device__ inline void split3(float a, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)a; float r = a - (float)h; m = (__bf16)r; r -= (float)m; l = (__bf16)r;
}
// C[32][32] = A[32][K] * B[K][32]; one wave
__global__ void k_probe(const float* A, const float* B, float* C, float* C6, int K) {
  const int l = threadIdx.x, i = l & 31, kq = l >> 5;
  floatx16 acc = {0}, acc6 = {0};
  for (int k0 = 0; k0 < K; k0 += 16) {
    bf16x8 ah, am, al, bh, bm, bl;
    for (int e = 0; e < 8; ++e) {
      const int k = k0 + 8 * kq + e;
      __bf16 h, m, lo;
      split3(A[i * K + k], h, m, lo); ah[e] = h; am[e] = m; al[e] = lo;
      split3(B[k * 32 + i], h, m, lo); bh[e] = h; bm[e] = m; bl[e] = lo;
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    // small terms first
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc6, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * kq;
    C[row * 32 + i] = acc[r];
    C6[row * 32 + i] = acc6[r];
  }
}
int main() {
  const int K = 2880;
  std::vector<float> A(32 * K), B(K * 32), C(1024), C6(1024);
  srand(1);
  for (auto& v : A) v = (rand() / (float)RAND_MAX - 0.5f);
  for (auto& v : B) v = (rand() / (float)RAND_MAX) * 3.f;
  float *dA, *dB, *dC, *dC6;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096); hipMalloc(&dC6, 4096);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dC6, K);
  hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
  hipMemcpy(C6.data(), dC6, 4096, hipMemcpyDeviceToHost);
  double e1 = 0, e6 = 0, ef = 0, mx = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double ref = 0; float f = 0;
    for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[k * 32 + j]; f = fmaf(A[i * K + k], B[k * 32 + j], f); }
    e1 = fmax(e1, fabs(C[i * 32 + j] - ref)); e6 = fmax(e6, fabs(C6[i * 32 + j] - ref)); ef = fmax(ef, fabs(f - ref)); mx = fmax(mx, fabs(ref));
  }
  printf("max|ref| %.4g  err bf16x1 %.3g  err bf16x3(6 prod) %.3g  err fp32 fma chain %.3g\n", mx, e1, e6, ef);
  return 0;
}
