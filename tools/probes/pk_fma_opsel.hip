// Stand-alone probe for profiles/r02_pk_fma_op_sel.md: does v_pk_fma_f32 with op_sel:[0,1,0] (low lane <- HIGH half of src1) return
// wrong low lanes when the SIMD's other wave runs MFMAs?  Workgroups of 4 waves, LDS sized so that exactly `wgs_per_cu` fit a CU.
// Every wave alternates blocks of packed FMAs (checked against scalar FMAs on the same inputs) and of v_mfma_f32_32x32x16_bf16.
//   usage: pk_fma_opsel [wgs_per_cu = 2] [workgroups = 2048] [iterations = 4000] [form: 0 = op_sel:[0,1,0], 1 = op_sel_hi:[1,0,1]]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int FORM>
__global__ __launch_bounds__(256) void probe(int iters, unsigned* bad, float* sink) {
  extern __shared__ char lds[];
  const int tid = threadIdx.x, wave = tid >> 6;
  // every wave alternates between a block of MFMAs and a block of packed FMAs; the phase depends on the wave and on the workgroup, so
  // whatever the placement, the two waves of a SIMD spend about half of the time in opposite roles
  const int phase = (wave + (int)(blockIdx.x >> 3) + (int)blockIdx.x) & 1;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (tid + i)); b[i] = (__bf16)(0.002f * (tid - i)); }
  f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
  f2 acc[8], ref[8], w[8], x[8];
  for (int k = 0; k < 8; ++k) {
    w[k] = f2{0.5f + 0.001f * (tid % 97 + k), -0.25f + 0.002f * (tid % 89 + k)};
    x[k] = f2{0.75f - 0.001f * (tid % 83 + k), 0.3f + 0.003f * (tid % 79 + k)};
    acc[k] = ref[k] = f2{0.f, 0.f};
  }
  unsigned wrong = 0;
  for (int blk = 0; blk < iters / 32; ++blk) {
    if ((blk + phase) & 1) {
      for (int it = 0; it < 32; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
      }
      continue;
    }
    for (int it = 0; it < 32; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (FORM == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[k]) : "v"(w[k]), "v"(x[k]));
        else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[k]) : "v"(w[k]), "v"(x[k]));
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float r0, r1;
        if (FORM == 0) { r0 = __builtin_fmaf(w[k][0], x[k][1], ref[k][0]); r1 = __builtin_fmaf(w[k][1], x[k][1], ref[k][1]); }
        else { r0 = __builtin_fmaf(w[k][0], x[k][0], ref[k][0]); r1 = __builtin_fmaf(w[k][1], x[k][0], ref[k][1]); }
        asm volatile("" : "+v"(r0), "+v"(r1));             // keep the reference scalar
        if (acc[k][0] != r0) wrong += 1u;
        if (acc[k][1] != r1) wrong += 0x10000u;
        ref[k] = f2{r0 * 0.5f, r1 * 0.5f};                 // decay keeps the values finite; an error does not propagate
        acc[k] = ref[k];
      }
    }
  }
  if (wrong) atomicAdd(bad, wrong);
  if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) sink[tid] = c0[0];
  if (lds[0] == 77 && tid == 999) sink[0] = 1.f;
}

int main(int argc, char** argv) {
  const int per_cu = argc > 1 ? atoi(argv[1]) : 2, wgs = argc > 2 ? atoi(argv[2]) : 2048, iters = argc > 3 ? atoi(argv[3]) : 4000;
  const int form = argc > 4 ? atoi(argv[4]) : 0;
  unsigned* bad; float* sink;
  hipMalloc(&bad, 4); hipMalloc(&sink, 4096); hipMemset(bad, 0, 4);
  const size_t ldsb = per_cu >= 2 ? 70 * 1024 : 100 * 1024;            // 160 KB per CU: 2 x 70 KB fit, 2 x 100 KB do not
  if (form == 0) hipLaunchKernelGGL(probe<0>, dim3(wgs), dim3(256), ldsb, 0, iters, bad, sink);
  else hipLaunchKernelGGL(probe<1>, dim3(wgs), dim3(256), ldsb, 0, iters, bad, sink);
  hipDeviceSynchronize();
  unsigned h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("form %s, %d workgroups per CU, %d workgroups x %d iterations x 8 chains: low-lane mismatches %u, high-lane mismatches %u\n",
         form == 0 ? "op_sel:[0,1,0]" : "op_sel_hi:[1,0,1]", per_cu, wgs, iters, h & 0xffffu, h >> 16);
  return 0;
}
