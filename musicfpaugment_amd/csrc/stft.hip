// Fused window + 512-point real FFT + magnitude (or one-sided PSD) for MI355X (gfx950).
//
// Replaces torch.stft/abs (training/visualisation.py:20-28), np.abs(stft.stft) (afp/audfprint/
// stft.py:15-62 via peak_extractor.py:259) and mlab.specgram (afp/dejavu/fingerprint.py:60-66).
//
// Mapping: one 256-thread workgroup = 16 consecutive frames of one clip; 16 lanes per frame.
// The 512-point real FFT is a 256-point complex FFT (z[n] = x[2n] + i x[2n+1]) done as
// 16 x 16: each lane runs a 16-point DFT in registers, the 16 lanes of a frame exchange once
// through a padded LDS tile (17-slot rows: conflict-free for both the row writes and the
// column reads), run the second 16-point DFT, and the real-FFT split pairs bins k / 256-k.
// Magnitudes are staged in LDS as [bin][16 frames] so the global store writes 16-frame
// segments of each bin row.  All arithmetic is float64 (the reference's window is float64,
// which promotes the whole STFT): HBM traffic is 514 KB per 8 s clip, the FFT ~4 MFLOP.
#include "mfpa_common.h"

#include <cmath>

namespace {

struct cd {
  double re, im;
};
__device__ __forceinline__ cd operator+(cd a, cd b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cd operator-(cd a, cd b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cd cmul(cd a, cd b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }

__device__ __forceinline__ void dft4(cd& a, cd& b, cd& c, cd& d) {
  cd s0 = a + c, s1 = a - c, s2 = b + d, s3 = b - d;
  a = s0 + s2;
  c = s0 - s2;
  b = {s1.re + s3.im, s1.im - s3.re};  // s1 - i s3
  d = {s1.re - s3.im, s1.im + s3.re};  // s1 + i s3
}

// In-place 16-point DFT.  On return X[k] sits at v[4*(k&3) + (k>>2)].
__device__ __forceinline__ void dft16(cd (&v)[16]) {
  constexpr double C1 = 0.92387953251128673848, S1 = 0.38268343236508978178, R2 = 0.70710678118654752440;
#pragma unroll
  for (int j2 = 0; j2 < 4; ++j2) dft4(v[j2], v[4 + j2], v[8 + j2], v[12 + j2]);
  // v[4*k1 + j2] *= W16^(j2*k1)
  v[4 * 1 + 1] = cmul(v[4 * 1 + 1], cd{C1, -S1});
  v[4 * 1 + 2] = cmul(v[4 * 1 + 2], cd{R2, -R2});
  v[4 * 1 + 3] = cmul(v[4 * 1 + 3], cd{S1, -C1});
  v[4 * 2 + 1] = cmul(v[4 * 2 + 1], cd{R2, -R2});
  v[4 * 2 + 2] = cd{v[4 * 2 + 2].im, -v[4 * 2 + 2].re};  // * -i
  v[4 * 2 + 3] = cmul(v[4 * 2 + 3], cd{-R2, -R2});
  v[4 * 3 + 1] = cmul(v[4 * 3 + 1], cd{S1, -C1});
  v[4 * 3 + 2] = cmul(v[4 * 3 + 2], cd{-R2, -R2});
  v[4 * 3 + 3] = cmul(v[4 * 3 + 3], cd{-C1, S1});
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) dft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}

constexpr int FRAMES_PER_BLOCK = 16;
constexpr int SLOTS = 16 * 17;  // padded 16x16 tile of doubles per frame (>= 257 + 1 slots of the [bin][16 frames] output tile per frame)

// MODE 0: centre/reflect magnitude (hypot).  MODE 1: no padding, one-sided PSD (|X|^2, interior bins x2).
template <int MODE, typename OutT>
__global__ __launch_bounds__(256, 4) void stft_kernel(const float* __restrict__ wav, int T_w, int nF,
                                                   const double* __restrict__ tables, OutT* __restrict__ out,
                                                   double* __restrict__ clip_max, double scale_in) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* buf = reinterpret_cast<double*>(smem);
  const int tid = threadIdx.x, l = tid & 15, fs = tid >> 4;
  const int b = blockIdx.y, f0 = blockIdx.x * FRAMES_PER_BLOCK, f = f0 + fs;
  const bool active = f < nF;
  const float* x = wav + (size_t)b * T_w;
  const double* win = tables;
  const double* tw256 = tables + 512;
  const double* tw512 = tables + 1024;
  double* mybuf = buf + fs * SLOTS;   // this frame's 16 x 17 slots: real parts, then imaginary parts

  cd a[16];
  const int start = (MODE == 0) ? 256 * f - 256 : 256 * f;
  const bool vec_ok = ((T_w & 1) == 0);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int i0 = 2 * (l + 16 * j);
    int s0 = start + i0, s1 = s0 + 1;
    float v0 = 0.f, v1 = 0.f;
    if (active) {
      if (MODE == 0 && (s0 < 0 || s1 >= T_w)) {
        if (s0 < 0) s0 = -s0;
        if (s1 < 0) s1 = -s1;
        if (s0 >= T_w) s0 = 2 * (T_w - 1) - s0;
        if (s1 >= T_w) s1 = 2 * (T_w - 1) - s1;
        v0 = x[s0];
        v1 = x[s1];
      } else if (vec_ok) {
        const float2 p = *reinterpret_cast<const float2*>(x + s0);
        v0 = p.x;
        v1 = p.y;
      } else {
        v0 = x[s0];
        v1 = x[s1];
      }
    }
    const double2 w = *reinterpret_cast<const double2*>(win + i0);
    if (MODE == 1) {
      a[j] = {(double)v0 * scale_in * w.x, (double)v1 * scale_in * w.y};
    } else {
      a[j] = {(double)v0 * w.x, (double)v1 * w.y};
    }
  }
  dft16(a);  // A[k1] at a[4*(k1&3) + (k1>>2)]
  // The 16 lanes of a frame sit in ONE wavefront (4 frames per wave), so the two exchanges need no workgroup barrier: a wave's LDS
  // operations execute in order, wave_sync() only keeps the compiler from moving them across each other.  Real and imaginary parts
  // go through the same 8-byte slots one after the other: 34 KB of LDS per workgroup instead of 68 KB -- four resident
  // workgroups per CU instead of two (the kernel is latency-bound: profiles/r03_prune_sq.md, 1.8 waves per SIMD, 38 % issuing).
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  cd c[16];
  {
    double tim[16];
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
      const int m = (l * k1) & 255;
      const double2 t = *reinterpret_cast<const double2*>(tw256 + 2 * m);
      const cd v = cmul(a[4 * (k1 & 3) + (k1 >> 2)], cd{t.x, t.y});
      mybuf[k1 * 17 + l] = v.re;
      tim[k1] = v.im;
    }
    wave_sync();
#pragma unroll
    for (int q = 0; q < 16; ++q) c[q].re = mybuf[l * 17 + q];
    wave_sync();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) mybuf[k1 * 17 + l] = tim[k1];
    wave_sync();
#pragma unroll
    for (int q = 0; q < 16; ++q) c[q].im = mybuf[l * 17 + q];
    wave_sync();
  }
  dft16(c);  // Z[l + 16*k2] at c[4*(k2&3) + (k2>>2)]

  // real-FFT split: bins k and 256-k from Z[k], Z[256-k]; slot(k) = k + (k >> 4)
  double zkr[9], znr[9], zki[9], zni[9];
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) mybuf[l + 17 * k2] = c[4 * (k2 & 3) + (k2 >> 2)].re;
  wave_sync();
#pragma unroll
  for (int m = 0; m < 9; ++m) {
    const int k = (m < 8) ? l + 16 * m : 128;
    const int kn = (256 - k) & 255;
    zkr[m] = mybuf[k + (k >> 4)];
    znr[m] = mybuf[kn + (kn >> 4)];
  }
  wave_sync();
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) mybuf[l + 17 * k2] = c[4 * (k2 & 3) + (k2 >> 2)].im;
  wave_sync();
#pragma unroll
  for (int m = 0; m < 9; ++m) {
    const int k = (m < 8) ? l + 16 * m : 128;
    const int kn = (256 - k) & 255;
    zki[m] = mybuf[k + (k >> 4)];
    zni[m] = mybuf[kn + (kn >> 4)];
  }
  double lo[9], hi[9];
#pragma unroll
  for (int m = 0; m < 9; ++m) {
    const int k = (m < 8) ? l + 16 * m : 128;
    const cd zk = {zkr[m], zki[m]}, zn = {znr[m], zni[m]};
    const cd e = {0.5 * (zk.re + zn.re), 0.5 * (zk.im - zn.im)};
    const cd o = {0.5 * (zk.im + zn.im), -0.5 * (zk.re - zn.re)};  // -i (zk - conj(zn)) / 2
    const double2 t = *reinterpret_cast<const double2*>(tw512 + 2 * k);
    const cd wo = cmul(cd{t.x, t.y}, o);
    const cd xk = e + wo, xn = e - wo;
    if (MODE == 0) {
      lo[m] = hypot(xk.re, xk.im);          // (sqrt(fma(re, re, im * im)) measured the same time and the same bits on the synthetic clips: kept hypot, numpy's abs)
      hi[m] = hypot(xn.re, xn.im);
    } else {
      lo[m] = (xk.re * xk.re + xk.im * xk.im) * ((k >= 1) ? 2.0 : 1.0);
      hi[m] = (xn.re * xn.re + xn.im * xn.im) * ((k >= 1) ? 2.0 : 1.0);
    }
  }
  __syncthreads();
  double* tile = reinterpret_cast<double*>(smem);  // [257][16]
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int k = l + 16 * m;
    tile[k * 16 + fs] = lo[m];
    if (k > 0) tile[(256 - k) * 16 + fs] = hi[m];
    if (k == 0) tile[256 * 16 + fs] = hi[m];  // Nyquist
  }
  if (l == 0) tile[128 * 16 + fs] = lo[8];
  __syncthreads();

  double vmax = 0.0;
  const int nvalid = min(FRAMES_PER_BLOCK, nF - f0);
  OutT* obase = out + (size_t)b * MFPA_N_BINS * nF + f0;
  for (int idx = tid; idx < MFPA_N_BINS * FRAMES_PER_BLOCK; idx += 256) {
    const int bin = idx >> 4, fr = idx & 15;
    if (fr < nvalid) {
      const double v = tile[idx];
      vmax = fmax(vmax, v);
      obase[(size_t)bin * nF + fr] = (OutT)v;
    }
  }
  if (clip_max != nullptr) {
    vmax = mfpa_wave_max(vmax);
    double* wmax = tile + MFPA_N_BINS * FRAMES_PER_BLOCK;  // past the tile, still inside the FFT buffer
    if ((tid & 63) == 0) wmax[tid >> 6] = vmax;
    __syncthreads();
    if (tid == 0) mfpa_atomic_max_nonneg(clip_max + b, fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3])));
  }
}

template <typename T>
__global__ void normalize_kernel(T* __restrict__ data, long long n, const double* __restrict__ clip_max, int B,
                                 int per_clip) {
  const int b = blockIdx.y;
  double d;
  if (per_clip) {
    d = clip_max[b];
  } else {
    __shared__ double gmax;
    if (threadIdx.x < MFPA_WAVE) {
      double m = 0.0;
      for (int i = threadIdx.x; i < B; i += MFPA_WAVE) m = fmax(m, clip_max[i]);
      m = mfpa_wave_max(m);
      if (threadIdx.x == 0) gmax = m;
    }
    __syncthreads();
    d = gmax;
  }
  T* p = data + (size_t)b * n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    p[i] = (T)((double)p[i] / d);
}

__global__ void f64_to_f32_kernel(const double* __restrict__ in, float* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = (float)in[i];
}

}  // namespace

extern "C" {

int mfpa_version(void) { return 38; }

int mfpa_stft_frames(int T_w) { return 1 + T_w / MFPA_N_HOP; }

int mfpa_stft_tables(const double* window512, double* out) {
  if (!window512 || !out) return MFPA_EINVAL;
  const long double pi = 3.14159265358979323846264338327950288L;
  for (int i = 0; i < 512; ++i) out[i] = window512[i];
  for (int m = 0; m < 256; ++m) {
    out[512 + 2 * m] = (double)cosl(2.0L * pi * m / 256.0L);
    out[512 + 2 * m + 1] = (double)(-sinl(2.0L * pi * m / 256.0L));
  }
  for (int k = 0; k <= 128; ++k) {
    out[1024 + 2 * k] = (double)cosl(2.0L * pi * k / 512.0L);
    out[1024 + 2 * k + 1] = (double)(-sinl(2.0L * pi * k / 512.0L));
  }
  for (int i = 1024 + 258; i < MFPA_STFT_TABLE_LEN; ++i) out[i] = 0.0;
  return MFPA_OK;
}

int mfpa_stft_mag(const float* wav, int B, int T_w, const double* tables, void* mag, int out_dtype,
                  double* clip_max, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!wav || !tables || !mag || B < 0 || T_w <= 256) return MFPA_EINVAL;
  if (out_dtype != MFPA_F32 && out_dtype != MFPA_F64) return MFPA_EINVAL;
  hipStream_t s = mfpa_stream(stream);
  const int nF = mfpa_stft_frames(T_w);
  if (clip_max) MFPA_HIP(hipMemsetAsync(clip_max, 0, sizeof(double) * B, s));
  dim3 grid((nF + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK, B);
  const size_t lds = sizeof(double) * SLOTS * FRAMES_PER_BLOCK;
  if (out_dtype == MFPA_F64)
    hipLaunchKernelGGL((stft_kernel<0, double>), grid, dim3(256), lds, s, wav, T_w, nF, tables, (double*)mag, clip_max, 1.0);
  else
    hipLaunchKernelGGL((stft_kernel<0, float>), grid, dim3(256), lds, s, wav, T_w, nF, tables, (float*)mag, clip_max, 1.0);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_specgram_frames(int T_w) { return (T_w - 256) / 256; }

int mfpa_specgram_psd(const float* wav, int B, int T_w, double scale_in, const double* tables, double* psd,
                      double* clip_max, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!wav || !tables || !psd || B < 0 || T_w < 512) return MFPA_EINVAL;
  hipStream_t s = mfpa_stream(stream);
  const int nF = mfpa_specgram_frames(T_w);
  if (clip_max) MFPA_HIP(hipMemsetAsync(clip_max, 0, sizeof(double) * B, s));
  dim3 grid((nF + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK, B);
  const size_t lds = sizeof(double) * SLOTS * FRAMES_PER_BLOCK;
  hipLaunchKernelGGL((stft_kernel<1, double>), grid, dim3(256), lds, s, wav, T_w, nF, tables, psd, clip_max, scale_in);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_normalize(void* data, int dtype, int B, long long n, const double* clip_max, int per_clip, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!data || !clip_max || B < 0 || n < 0) return MFPA_EINVAL;
  if (dtype != MFPA_F32 && dtype != MFPA_F64) return MFPA_EINVAL;
  if (B == 0 || n == 0) return MFPA_OK;
  hipStream_t s = mfpa_stream(stream);
  const int gx = (int)((n + 256 * 8 - 1) / (256 * 8));
  dim3 grid(gx < 1 ? 1 : (gx > 64 ? 64 : gx), B);
  if (dtype == MFPA_F64)
    hipLaunchKernelGGL(normalize_kernel<double>, grid, dim3(256), 0, s, (double*)data, n, clip_max, B, per_clip);
  else
    hipLaunchKernelGGL(normalize_kernel<float>, grid, dim3(256), 0, s, (float*)data, n, clip_max, B, per_clip);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

// out[b][i] = (float)(in[b][i] / denom[b]): the reference's `spectrogram / max` followed by `.float()` (training/train.py:264-272) as one pass
__global__ __launch_bounds__(256) void normalize_f32_kernel(const double* __restrict__ in, long long n, const double* __restrict__ denom,
                                                            float* __restrict__ out) {
  const int b = blockIdx.y;
  const double d = denom[b];
  const double* ib = in + (size_t)b * n;
  float* ob = out + (size_t)b * n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) ob[i] = (float)(ib[i] / d);
}

int mfpa_normalize_f32(const double* in, int B, long long n, const double* denom, float* out, void* stream) {
  if (B == 0 || n == 0) return MFPA_OK;
  if (!in || !denom || !out || B < 0 || B > 65535 || n < 0) return MFPA_EINVAL;
  const long long gx = (n + 255) / 256;
  hipLaunchKernelGGL(normalize_f32_kernel, dim3((unsigned)(gx > 256 ? 256 : gx), (unsigned)B), dim3(256), 0, mfpa_stream(stream), in, n, denom, out);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_f64_to_f32(const double* in, float* out, long long n, void* stream) {
  if (!in || !out || n < 0) return MFPA_EINVAL;
  if (n == 0) return MFPA_OK;
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(f64_to_f32_kernel, dim3((int)blocks), dim3(256), 0, mfpa_stream(stream), in, out, n);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

}  // extern "C"
