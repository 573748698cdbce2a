// Layout probe for v_mfma_f32_16x16x32_bf16 on gfx950: D (16 x 16) = A (16 x 32) * B (32 x 16).
// Assumed: lane l holds A[row = l & 15][k = 8 (l >> 4) .. + 7], B[k = 8 (l >> 4) .. + 7][col = l & 15], D[row = 4 (l >> 4) + j][col = l & 15], j = 0..3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D) {
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)A[(l & 15) * 32 + 8 * (l >> 4) + i];
    b[i] = (__bf16)B[(8 * (l >> 4) + i) * 16 + (l & 15)];
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int j = 0; j < 4; ++j) D[(4 * (l >> 4) + j) * 16 + (l & 15)] = c[j];
}
int main() {
  float hA[16 * 32], hB[32 * 16], hD[256], ref[256];
  for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
  for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) { float s = 0; for (int kk = 0; kk < 32; ++kk) s += hA[r * 32 + kk] * hB[kk * 16 + c]; ref[r * 16 + c] = s; }
  float *dA, *dB, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256; ++i) if (hD[i] != ref[i]) ++bad;
  printf("mfma_f32_16x16x32_bf16 layout probe: %d mismatches of 256\n", bad);
  return bad != 0;
}
