// Peak-mask precision / recall counts for MI355X (gfx950), testing/metrics.py:10-192 of the
// reference.  The reference walks torch.nonzero(mask) in Python and multiplies a clipped 3x3
// window by a centre-only kernel sliced [:2] on the low borders, so a peak (i, j) is looked up in
// the other mask at (i + [i == 0], j + [j == 0]).  Here every cell is tested in one coalesced
// sweep; a workgroup owns a clip and emits integer counts [hit_p, n_p, hit_r, n_r] (HBM-bound:
// 2 bytes read per cell, the shifted look-ups hit L1/L2).
#include "mfpa_common.h"

namespace {

__global__ __launch_bounds__(256) void peak_metrics_kernel(const uint8_t* __restrict__ pred,
                                                           const uint8_t* __restrict__ gt, int N1, int N2,
                                                           int64_t* __restrict__ counts) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const size_t n = (size_t)N1 * N2;
  const uint8_t* P = pred + b * n;
  const uint8_t* G = gt + b * n;
  int hp = 0, np_ = 0, hr = 0, nr = 0;
  for (size_t e = tid; e < n; e += 256) {
    const uint8_t p = P[e], g = G[e];
    if (p | g) {
      const int i = (int)(e / N2), j = (int)(e % N2);
      const size_t tap = (size_t)(i + (i == 0)) * N2 + (j + (j == 0));
      if (p) {
        ++np_;
        hp += G[tap] != 0;
      }
      if (g) {
        ++nr;
        hr += P[tap] != 0;
      }
    }
  }
  __shared__ int red[4][4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    hp += __shfl_xor(hp, o);
    np_ += __shfl_xor(np_, o);
    hr += __shfl_xor(hr, o);
    nr += __shfl_xor(nr, o);
  }
  if ((tid & 63) == 0) {
    red[tid >> 6][0] = hp;
    red[tid >> 6][1] = np_;
    red[tid >> 6][2] = hr;
    red[tid >> 6][3] = nr;
  }
  __syncthreads();
  if (tid < 4) counts[(size_t)b * 4 + tid] = (int64_t)red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// Per-clip statistics for PSNR(pred, target): out[b] = {sum (pred-target)^2, min(target), max(target)} in float64.
// psnr = 10 log10((max-min)^2 / (sse / n)) -- torchmetrics' PeakSignalNoiseRatio with data_range taken from the target
// (testing/metrics.py:7; the package is not vendored in the reference tree: parity unpinned).
template <typename TP>
__global__ __launch_bounds__(256) void psnr_stats_kernel(const TP* __restrict__ pred, const double* __restrict__ target,
                                                         long long n, double* __restrict__ out) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const TP* P = pred + (size_t)b * n;
  const double* G = target + (size_t)b * n;
  double sse = 0, mn = INFINITY, mx = -INFINITY;
  for (long long i = tid; i < n; i += 256) {
    const double g = G[i], d = (double)P[i] - g;
    sse += d * d;
    mn = g < mn ? g : mn;
    mx = g > mx ? g : mx;
  }
  __shared__ double sh[3][256];
  sh[0][tid] = sse; sh[1][tid] = mn; sh[2][tid] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      sh[0][tid] += sh[0][tid + o];
      sh[1][tid] = fmin(sh[1][tid], sh[1][tid + o]);
      sh[2][tid] = fmax(sh[2][tid], sh[2][tid + o]);
    }
    __syncthreads();
  }
  if (tid == 0) { out[3 * b] = sh[0][0]; out[3 * b + 1] = sh[1][0]; out[3 * b + 2] = sh[2][0]; }
}

}  // namespace

extern "C" int mfpa_psnr_stats(const void* pred, int pred_dtype, const double* target, int B, long long n, double* out,
                               void* stream) {
  if (B == 0) return MFPA_OK;
  if (!pred || !target || !out || B < 0 || n < 1 || (pred_dtype != MFPA_F32 && pred_dtype != MFPA_F64)) return MFPA_EINVAL;
  if (pred_dtype == MFPA_F32)
    hipLaunchKernelGGL(psnr_stats_kernel<float>, dim3(B), dim3(256), 0, mfpa_stream(stream), (const float*)pred, target, n, out);
  else
    hipLaunchKernelGGL(psnr_stats_kernel<double>, dim3(B), dim3(256), 0, mfpa_stream(stream), (const double*)pred, target, n, out);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

extern "C" int mfpa_peak_metrics(const uint8_t* predicted, const uint8_t* gt, int B, int N1, int N2, int64_t* counts,
                                 void* stream) {
  if (B == 0) return MFPA_OK;
  if (!predicted || !gt || !counts || B < 0 || N1 < 2 || N2 < 2) return MFPA_EINVAL;
  hipLaunchKernelGGL(peak_metrics_kernel, dim3(B), dim3(256), 0, mfpa_stream(stream), predicted, gt, N1, N2, counts);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}
