// Probe: do VALU instructions overlap v_mfma_f32_16x16x32_bf16 on one SIMD?  512 threads (8 waves, 2 per SIMD), one workgroup per CU.
// Each wave: a loop of 24 independent-accumulator MFMAs with K dependent-free VALU ops (v_fma_f32 on private registers) after each.
// hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int K, bool MFMA>
__global__ __launch_bounds__(512, 1) void probe(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
  floatx4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 1.0f + threadIdx.x * 1e-6f * i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 24; ++m) {
      if (MFMA) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 7], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(m * K + k) & 7]) : "v"(v[(m + 3) & 7]));
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3] + v[i];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int K, bool MFMA>
float run(int iters) {
  float* out; hipMalloc(&out, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<K, MFMA>), dim3(256), dim3(512), 0, 0, out, 16);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<K, MFMA>), dim3(256), dim3(512), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  return ms;
}

int main() {
  const int iters = 20000;
  const double mf = 256.0 * 8 * 24.0 * iters;   // MFMAs
  printf("K = VALU (v_fma_f32) per MFMA; 8 waves / CU, 24 MFMAs per loop body, %d iterations\n", iters);
#define ROW(K) { float t1 = run<K, true>(iters), t0 = run<K, false>(iters); \
    printf("K=%d  MFMA+VALU %8.3f ms (%.0f TFLOP/s issue)   VALU only %8.3f ms   MFMA only see K=0\n", K, t1, mf * 16384.0 / (t1 * 1e-3) / 1e12, t0); }
  ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(6) ROW(8)
  return 0;
}
