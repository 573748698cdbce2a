// Probe: semantics of ds_read_b64_tr_b16 (builtin __builtin_amdgcn_ds_read_tr16_b64_v4i16) on gfx950.
// LDS image: rows of 64 shorts, value = row * 100 + col.  Each 16-lane group reads the 4 x 16 block at
// (row0 = 4 * group, col0 = 16 * (group & 1)).  Prints what every lane received.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)((i / 64) * 100 + (i % 64));
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, gl = lane & 15, q = gl >> 2, p = gl & 3;
  const int row = 4 * g + q, col = 16 * (g & 1) + 4 * p;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + row * 64 + col));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
  short* d; short h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  return 0;
}
