// Waveform-domain spectral losses of the Demucs branch (reference training/loss.py:10-186) for MI355X (gfx950).
//
// The three STFT resolutions (fft 1024 / 2048 / 512, hop 120 / 240 / 50, hann 600 / 1200 / 240 zero-padded to the FFT size)
// have windows of only 0.47 .. 0.59 of the frame and hops that are not powers of two, so the transform is evaluated as
// the product  frames (rows = strided windows of the reflect-padded signal, K = window length) x windowed DFT matrix
// on the fp32 matrix cores through mfpa_gemm_mfma -- exact fp32 products, fp32 accumulate, the accuracy class of the
// reference's fp32 FFT.  This file holds the kernels around that GEMM: reflect padding, magnitude, and the loss sums.
#include "mfpa_common.h"

namespace {

// out[b][i] = x[b][reflect(i + shift - pad)] for i + shift < T + 2 pad, else 0   (torch.stft center=True, pad_mode="reflect")
__global__ __launch_bounds__(256) void reflect_pad_kernel(const float* __restrict__ x, int T, int pad, int shift, int Lout,
                                                          float* __restrict__ out) {
  const int b = blockIdx.y;
  const float* xb = x + (size_t)b * T;
  float* ob = out + (size_t)b * Lout;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < Lout; i += gridDim.x * 256) {
    const int p = i + shift;
    float v = 0.f;
    if (p < T + 2 * pad) {
      int s = p - pad;
      if (s < 0) s = -s;                       // reflect without repeating the edge sample
      if (s >= T) s = 2 * (T - 1) - s;
      v = xb[s];
    }
    ob[i] = v;
  }
}

__device__ __forceinline__ float dft_mag(const float* __restrict__ row, int k, int im_off) {
  const float re = row[k], im = row[im_off + k];
  return sqrtf(fmaxf(re * re + im * im, 1e-7f));      // loss.py:38-41: clamp before the square root
}

// mag[row][k] = sqrt(clamp(re^2 + im^2, 1e-7)) from the GEMM output rows [re(0..bins) ... | im at im_off ...]
__global__ __launch_bounds__(256) void dft_mag_kernel(const float* __restrict__ c, long long rows, int bins, long long ldc,
                                                      int im_off, float* __restrict__ mag) {
  const long long total = rows * bins;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long r = e / bins;
    const int k = (int)(e % bins);
    mag[e] = dft_mag(c + r * ldc, k, im_off);
  }
}

// partial[block] = [sum (ym - xm)^2, sum ym^2, sum |log ym - log xm|] in float64 (per-element arithmetic in float32 like the
// reference); a second launch adds the partials in block order (deterministic).
__global__ __launch_bounds__(256) void stft_loss_partial_kernel(const float* __restrict__ cx, const float* __restrict__ cy,
                                                                long long rows, int bins, long long ldc, int im_off,
                                                                double* __restrict__ partial) {
  const long long total = rows * bins;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long r = e / bins;
    const int k = (int)(e % bins);
    const float xm = dft_mag(cx + r * ldc, k, im_off), ym = dft_mag(cy + r * ldc, k, im_off);
    const float d = ym - xm;
    s0 += (double)(d * d);
    s1 += (double)(ym * ym);
    s2 += (double)fabsf(logf(ym) - logf(xm));
  }
  __shared__ double sh[3][256];
  sh[0][threadIdx.x] = s0; sh[1][threadIdx.x] = s1; sh[2][threadIdx.x] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o)
#pragma unroll
      for (int q = 0; q < 3; ++q) sh[q][threadIdx.x] += sh[q][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 3) partial[(size_t)blockIdx.x * 3 + threadIdx.x] = sh[threadIdx.x][0];
}

__global__ __launch_bounds__(64) void stft_loss_finish_kernel(const double* __restrict__ partial, int nblk, double* __restrict__ out3) {
  if (threadIdx.x < 3) {
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += partial[(size_t)i * 3 + threadIdx.x];
    out3[threadIdx.x] = s;
  }
}

// Backward of one resolution with respect to the PREDICTED signal x.  In place on the GEMM output of x:
//   c[row][re k], c[row][im k]  <-  dL/d re, dL/d im,   L = w_sc * ||Y|-|X||_F / ||Y||_F + w_mag * mean |log|Y| - log|X||
// (loss.py:44-83; w_* carry the factors and the 1/#resolutions of MultiResolutionSTFTLoss).  sums = the forward's
// [sum (|Y|-|X|)^2, sum |Y|^2, .] (device).  The clamp at 1e-7 has zero gradient below it.
__global__ __launch_bounds__(256) void stft_loss_grad_kernel(float* __restrict__ cx, const float* __restrict__ cy, long long rows,
                                                             int bins, long long ldc, int im_off, const double* __restrict__ sums,
                                                             double w_sc, double w_mag) {
  const long long total = rows * bins;
  const double n_diff = sqrt(sums[0]), n_y = sqrt(sums[1]);
  const float k_sc = (n_diff > 0.0 && n_y > 0.0) ? (float)(w_sc / (n_diff * n_y)) : 0.f;      // d sc / d|X| = -(|Y|-|X|) * k_sc
  const float k_mag = (float)(w_mag / (double)total);
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long r = e / bins;
    const int k = (int)(e % bins);
    float* row = cx + r * ldc;
    const float re = row[k], im = row[im_off + k];
    const float p = re * re + im * im;
    const float xm = sqrtf(fmaxf(p, 1e-7f)), ym = dft_mag(cy + r * ldc, k, im_off);
    const float dl = logf(ym) - logf(xm);
    float g = -(ym - xm) * k_sc;                                           // spectral convergence
    g -= (dl > 0.f ? 1.f : (dl < 0.f ? -1.f : 0.f)) * k_mag / xm;          // log-magnitude L1
    const float s = p > 1e-7f ? g / xm : 0.f;                              // d|X| / d(re, im) = (re, im) / |X| above the clamp
    row[k] = s * re;
    row[im_off + k] = s * im;
  }
}

// Adjoint of the framing: dxp[b][i] = sum over frames t of dframes[b][t][i - t*hop - off]  (0 <= . < win).
__global__ __launch_bounds__(256) void frames_adjoint_kernel(const float* __restrict__ df, int frames, long long ldf, int win,
                                                             int hop, int off, int L, float* __restrict__ dxp) {
  const int b = blockIdx.y;
  const float* dfb = df + (size_t)b * frames * ldf;
  float* ob = dxp + (size_t)b * L;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < L; i += gridDim.x * 256) {
    const int u = i - off;                                   // u = t*hop + j, 0 <= j < win
    float acc = 0.f;
    if (u >= 0) {
      int t_hi = u / hop; if (t_hi > frames - 1) t_hi = frames - 1;
      for (int t = t_hi; t >= 0; --t) {
        const int j = u - t * hop;
        if (j >= win) break;
        acc += dfb[(size_t)t * ldf + j];
      }
    }
    ob[i] = acc;
  }
}

// Adjoint of the reflect padding: dx[s] = dxp[s + pad] + dxp[pad - s] (1 <= s <= pad) + dxp[2(T-1) - s + pad] (T-1-pad <= s <= T-2);
// accumulate != 0 adds to dx (the three resolutions and the L1 term share one gradient buffer).
__global__ __launch_bounds__(256) void reflect_pad_adjoint_kernel(const float* __restrict__ dxp, int T, int pad, int L, int accumulate,
                                                                  float* __restrict__ dx) {
  const int b = blockIdx.y;
  const float* ib = dxp + (size_t)b * L;
  float* ob = dx + (size_t)b * T;
  for (int s = blockIdx.x * 256 + threadIdx.x; s < T; s += gridDim.x * 256) {
    float v = ib[s + pad];
    if (s >= 1 && s <= pad) v += ib[pad - s];
    if (s >= T - 1 - pad && s <= T - 2) v += ib[2 * (T - 1) - s + pad];
    ob[s] = accumulate ? ob[s] + v : v;
  }
}

constexpr int LOSS_BLOCKS = 1024;

}  // namespace

extern "C" {

int mfpa_loss_blocks(void) { return LOSS_BLOCKS; }

int mfpa_reflect_pad(const float* x, int B, int T, int pad, int shift, int Lout, float* out, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!x || !out || B < 0 || B > 65535 || T < 2 || pad < 0 || pad >= T || shift < 0 || Lout < 1) return MFPA_EINVAL;
  int gx = (Lout + 255) / 256; if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(reflect_pad_kernel, dim3(gx, B), dim3(256), 0, mfpa_stream(stream), x, T, pad, shift, Lout, out);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_dft_mag(const float* c, long long rows, int bins, long long ldc, int im_off, float* mag, void* stream) {
  if (rows == 0) return MFPA_OK;
  if (!c || !mag || rows < 0 || bins < 1 || im_off < bins || ldc < im_off + bins) return MFPA_EINVAL;
  long long blocks = (rows * bins + 255) / 256; if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dft_mag_kernel, dim3((unsigned)blocks), dim3(256), 0, mfpa_stream(stream), c, rows, bins, ldc, im_off, mag);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_stft_loss_sums(const float* cx, const float* cy, long long rows, int bins, long long ldc, int im_off, double* out3,
                        double* workspace, void* stream) {
  if (!cx || !cy || !out3 || !workspace || rows < 1 || bins < 1 || im_off < bins || ldc < im_off + bins) return MFPA_EINVAL;
  long long blocks = (rows * bins + 256 * 8 - 1) / (256 * 8);
  if (blocks > LOSS_BLOCKS) blocks = LOSS_BLOCKS;
  if (blocks < 1) blocks = 1;
  hipStream_t s = mfpa_stream(stream);
  hipLaunchKernelGGL(stft_loss_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, s, cx, cy, rows, bins, ldc, im_off, workspace);
  MFPA_CHECK_LAUNCH();
  hipLaunchKernelGGL(stft_loss_finish_kernel, dim3(1), dim3(64), 0, s, workspace, (int)blocks, out3);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_stft_loss_grad(float* cx, const float* cy, long long rows, int bins, long long ldc, int im_off, const double* sums,
                        double w_sc, double w_mag, void* stream) {
  if (rows == 0) return MFPA_OK;
  if (!cx || !cy || !sums || rows < 0 || bins < 1 || im_off < bins || ldc < im_off + bins) return MFPA_EINVAL;
  long long blocks = (rows * bins + 255) / 256; if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(stft_loss_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, mfpa_stream(stream), cx, cy, rows, bins, ldc, im_off,
                     sums, w_sc, w_mag);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_frames_adjoint(const float* dframes, int B, int frames, long long ldf, int win, int hop, int off, int L, float* dxp,
                        void* stream) {
  if (B == 0) return MFPA_OK;
  if (!dframes || !dxp || B < 0 || B > 65535 || frames < 1 || win < 1 || hop < 1 || off < 0 || L < 1 || ldf < win) return MFPA_EINVAL;
  int gx = (L + 255) / 256; if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(frames_adjoint_kernel, dim3(gx, B), dim3(256), 0, mfpa_stream(stream), dframes, frames, ldf, win, hop, off, L, dxp);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_reflect_pad_adjoint(const float* dxp, int B, int T, int pad, int L, int accumulate, float* dx, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!dxp || !dx || B < 0 || B > 65535 || T < 2 || pad < 0 || pad >= T || L < T + 2 * pad) return MFPA_EINVAL;
  int gx = (T + 255) / 256; if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(reflect_pad_adjoint_kernel, dim3(gx, B), dim3(256), 0, mfpa_stream(stream), dxp, T, pad, L, accumulate, dx);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

}  // extern "C"
