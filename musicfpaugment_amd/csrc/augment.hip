// AugmentFP signal chain for MI355X (gfx950) -- next-tier row SURVEY.md §8f-3.
// Reference: augmentation/__init__.py:46-93 and augmentation/transformations/*.py of deezer/musicFPaugment:
//   HighPass -> impulse response -> background noise @SNR -> gain -> clipping -> LowPass -> HighPass -> peak normalise,
// each transform Bernoulli-gated per example (`apply[b]`; a skipped example is copied through).
//
// The random draws (gates, cut-offs, SNRs, gains, percentiles, which IR / noise) are made on the host exactly like the
// reference does with torch.distributions; these kernels do the per-sample arithmetic:
//   fir_kernel      one tiled FIR engine for the windowed-sinc low/high-pass (julius-style, replicate padding) and the
//                   impulse-response convolution (direct form, == the reference's FFT convolution up to rounding);
//                   4 outputs per thread, input window + taps staged in LDS, 16 FMAs per ds_read_b128
//   mix_kernel      x + rms(x)/10^(snr/20) * noise, then peak normalisation
//   clip_kernel     per-example torch.quantile (linear interpolation) by a 3-pass radix select in LDS, then clamp
//   scale / peak    gain, peak normalisation
#include "mfpa_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FIR_OUT = 1024;   // outputs per workgroup (256 threads x 4)
constexpr int FIR_KCH = 512;    // taps per LDS chunk

// Windowed-sinc low-pass taps (julius.LowPassFilters 0.2.7, zeros = 8): taps[k] = 2c hann(k) sinc(2c pi (k - half)),
// k = 0..2*half, normalised to unit sum.  One workgroup per example; example b owns taps[tap_off[b] .. + 2*half[b]+1).
__global__ __launch_bounds__(256) void lowpass_taps_kernel(const float* __restrict__ cutoff, const int* __restrict__ half,
                                                           const long long* __restrict__ tap_off, float* __restrict__ taps) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int h = half[b], n = 2 * h + 1;
  const float c = cutoff[b];
  float* T = taps + tap_off[b];
  double s = 0.0;
  for (int k = tid; k < n; k += 256) {
    const float win = 0.5f - 0.5f * cosf(6.283185307179586f * (float)k / (float)(n - 1));   // hann_window(n, periodic=False)
    const float t = (float)(k - h);
    const float arg = 2.f * c * 3.141592653589793f * t;
    const float sinc = (k == h) ? 1.f : sinf(arg) / arg;
    const float v = 2.f * c * win * sinc;
    T[k] = v;
    s += (double)v;
  }
  __shared__ double sh[256];
  sh[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) sh[tid] += sh[tid + o];
    __syncthreads();
  }
  const float inv = (float)(1.0 / sh[0]);
  for (int k = tid; k < n; k += 256) T[k] *= inv;
}

// y[t] = sum_k taps[k] * xpad[t + k - off[b]],  k < ntaps[b];  xpad = x with replicate (pad_mode 0) or zero (1) padding.
// out_mode 0: y;  1: x - y (high-pass);  2: store t < T only and track max |y| over all t < Tout (impulse response).
// A filter may be far longer than the clip (a 0.5 Hz high-pass at 8 kHz has 128 001 taps): a workgroup only multiplies the
// taps that meet a real sample for at least one of its outputs, [klo, khi); the taps before / after that range see only
// the replicated first / last sample (or zeros), so they enter as two tap sums.  Work per clip <= T * (T + 1023) MACs.
__global__ MFPA_NO_PK_F32 __launch_bounds__(256) void fir_kernel(const float* __restrict__ x, int T, int Tout, const float* __restrict__ taps,
                                                  const long long* __restrict__ tap_off, const int* __restrict__ ntaps,
                                                  const int* __restrict__ off, const uint8_t* __restrict__ apply, int pad_mode,
                                                  int out_mode, float* __restrict__ y, float* __restrict__ peak) {
  __shared__ __attribute__((aligned(16))) float win[FIR_OUT + FIR_KCH + 8];
  __shared__ __attribute__((aligned(16))) float tp[FIR_KCH];
  __shared__ double esum[2][4];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int t0 = blockIdx.x * FIR_OUT;
  const float* xb = x + (size_t)b * T;
  float* yb = y + (size_t)b * T;
  if (!apply[b]) {                                        // gate off: copy through
    for (int i = tid; i < FIR_OUT; i += 256)
      if (t0 + i < T) yb[t0 + i] = xb[t0 + i];
    return;
  }
  const int n = ntaps[b], of = off[b];
  const float* tb = taps + tap_off[b];
  // taps k in [klo, khi) read a sample index in [0, T) for some output of this workgroup
  const long long klo_l = (long long)of - (t0 + FIR_OUT - 1), khi_l = (long long)of - t0 + T;
  const int klo = klo_l < 0 ? 0 : (klo_l > n ? n : (int)klo_l);
  const int khi = khi_l > n ? n : (khi_l < klo ? klo : (int)khi_l);
  float edge = 0.f;
  if (pad_mode == 0 && (klo > 0 || khi < n)) {            // uniform per workgroup
    double lo = 0.0, hi = 0.0;
    for (int k = tid; k < klo; k += 256) lo += (double)tb[k];
    for (int k = khi + tid; k < n; k += 256) hi += (double)tb[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo += __shfl_xor(lo, o);
      hi += __shfl_xor(hi, o);
    }
    if ((tid & 63) == 0) { esum[0][tid >> 6] = lo; esum[1][tid >> 6] = hi; }
    __syncthreads();
    lo = esum[0][0] + esum[0][1] + esum[0][2] + esum[0][3];
    hi = esum[1][0] + esum[1][1] + esum[1][2] + esum[1][3];
    edge = (float)(lo * (double)xb[0] + hi * (double)xb[T - 1]);
  }
  f32x4 acc = {edge, edge, edge, edge};
  for (int k0 = klo; k0 < khi; k0 += FIR_KCH) {
    __syncthreads();
    for (int i = tid; i < FIR_OUT + FIR_KCH + 4; i += 256) {       // window: xpad[t0 + k0 - of + i]
      long long s = (long long)t0 + k0 - of + i;
      float v;
      if (pad_mode == 0) {
        s = s < 0 ? 0 : (s >= T ? T - 1 : s);
        v = xb[s];
      } else {
        v = (s >= 0 && s < T) ? xb[s] : 0.f;
      }
      win[i] = v;
    }
    for (int i = tid; i < FIR_KCH; i += 256) tp[i] = (k0 + i < khi) ? tb[k0 + i] : 0.f;
    __syncthreads();
    const float* w = win + 4 * tid;
    f32x4 lo = *reinterpret_cast<const f32x4*>(w);
#pragma unroll 4
    for (int kk = 0; kk < FIR_KCH; kk += 4) {
      const f32x4 hi = *reinterpret_cast<const f32x4*>(w + kk + 4);
      const f32x4 c = *reinterpret_cast<const f32x4*>(tp + kk);
      // outputs j = 0..3 use window[kk + j + i], i = 0..3
      acc[0] += c[0] * lo[0] + c[1] * lo[1] + c[2] * lo[2] + c[3] * lo[3];
      acc[1] += c[0] * lo[1] + c[1] * lo[2] + c[2] * lo[3] + c[3] * hi[0];
      acc[2] += c[0] * lo[2] + c[1] * lo[3] + c[2] * hi[0] + c[3] * hi[1];
      acc[3] += c[0] * lo[3] + c[1] * hi[0] + c[2] * hi[1] + c[3] * hi[2];
      lo = hi;
    }
  }
  float pk = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = t0 + 4 * tid + j;
    if (t < Tout) {
      pk = fmaxf(pk, fabsf(acc[j]));
      if (t < T) yb[t] = (out_mode == 1) ? xb[t] - acc[j] : acc[j];
    }
  }
  if (out_mode == 2) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pk = fmaxf(pk, __shfl_xor(pk, o));
    if ((tid & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(peak + b), __float_as_uint(pk));
  }
}

// y[b] = x[b] * (apply[b] ? factor[b] : 1)   (gain; division by a peak when invert != 0)
__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ x, int T, const float* __restrict__ factor,
                                                         const uint8_t* __restrict__ apply, int invert, float* __restrict__ y) {
  const int b = blockIdx.y;
  const bool on = apply ? apply[b] != 0 : true;
  const float f = factor[b];
  const float* xb = x + (size_t)b * T;
  float* yb = y + (size_t)b * T;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < T; i += gridDim.x * 256) {
    const float v = xb[i];
    yb[i] = !on ? v : (invert ? v / f : v * f);
  }
}

__device__ __forceinline__ float block_reduce(float v, float* sh, bool is_max) {  // 1024 threads
  const int tid = threadIdx.x;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float u = __shfl_xor(v, o);
    v = is_max ? fmaxf(v, u) : v + u;
  }
  __syncthreads();
  if ((tid & 63) == 0) sh[tid >> 6] = v;
  __syncthreads();
  float r = sh[0];
  for (int w = 1; w < 16; ++w) r = is_max ? fmaxf(r, sh[w]) : r + sh[w];
  return r;
}

// AddBackgroundNoise.random_background (background_noise.py:64-141, non-mixup branch): example b is the concatenation of
// up to P slices bank[src .. src+len) of the resident noise bank, each RMS-normalised on its own, the whole RMS-normalised
// again (x / (rms + 1e-8), utils.py:190-205).  One workgroup per example.
__global__ __launch_bounds__(1024) void gather_background_kernel(const float* __restrict__ bank, const long long* __restrict__ src,
                                                                 const int* __restrict__ len, int P, int T,
                                                                 float* __restrict__ out) {
  __shared__ float sh[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  float* ob = out + (size_t)b * T;
  int pos = 0;
  for (int p = 0; p < P && pos < T; ++p) {
    const int L = len[b * P + p];
    if (L <= 0) break;
    const float* sp = bank + src[b * P + p];
    float s = 0.f;
    for (int i = tid; i < L; i += 1024) s += sp[i] * sp[i];
    s = block_reduce(s, sh, false);
    const float r = sqrtf(s / (float)L) + 1e-8f;
    for (int i = tid; i < L; i += 1024) ob[pos + i] = sp[i] / r;
    pos += L;
  }
  __syncthreads();
  float s = 0.f;
  for (int i = tid; i < T; i += 1024) s += ob[i] * ob[i];
  s = block_reduce(s, sh, false);
  const float r = sqrtf(s / (float)T) + 1e-8f;
  for (int i = tid; i < T; i += 1024) ob[i] = ob[i] / r;
}

// AddBackgroundNoise.apply_transform (background_noise.py:183-215): y = x + rms(x) / 10^(snr/20) * noise; y /= max|y|.
// PeakNormalization (peak_normalization.py:38-67) is the same kernel with noise == nullptr: y = x / max|x| when max > 0.
__global__ MFPA_NO_PK_F32 __launch_bounds__(1024) void mix_kernel(const float* __restrict__ x, int T, const float* __restrict__ noise,
                                                   const float* __restrict__ snr_db, const uint8_t* __restrict__ apply,
                                                   float* __restrict__ y) {
  __shared__ float sh[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* xb = x + (size_t)b * T;
  float* yb = y + (size_t)b * T;
  if (apply && !apply[b]) {
    for (int i = tid; i < T; i += 1024) yb[i] = xb[i];
    return;
  }
  float scale = 0.f;
  if (noise) {
    float s = 0.f;
    for (int i = tid; i < T; i += 1024) s += xb[i] * xb[i];
    s = block_reduce(s, sh, false);
    const float rms = sqrtf(s / (float)T);
    scale = rms / powf(10.f, snr_db[b] / 20.f);
  }
  const float* nb = noise ? noise + (size_t)b * T : nullptr;
  float pk = 0.f;
  for (int i = tid; i < T; i += 1024) {
    const float v = noise ? xb[i] + scale * nb[i] : xb[i];
    yb[i] = v;
    pk = fmaxf(pk, fabsf(v));
  }
  pk = block_reduce(pk, sh, true);
  if (!noise && !(pk > 0.f)) return;            // PeakNormalization leaves silent clips alone
  for (int i = tid; i < T; i += 1024) yb[i] = yb[i] / pk;
}

// Order-preserving float -> uint32 key.
__device__ __forceinline__ unsigned f2key(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// Clipping.apply_transform for ONE example per workgroup (clipping.py:67-100 as AugmentFP.__call__ uses it):
// lo = quantile(x, p/2), hi = quantile(x, 1 - p/2) with torch.quantile's linear interpolation, y = clamp(x, lo, hi).
// Four order statistics (floor/ceil ranks of both quantiles) are found together by a 3-pass (11/11/10 bit) radix select.
// FLAT: the quantiles of example b are taken over ALL selected examples flattened together -- what the reference's
// batch_augment computes, because its torch.quantile call has no dim argument (clipping.py:77-93); every workgroup then
// walks the whole selected sub-batch (B' * T elements, at most torch.quantile's 16 000 000).
template <bool FLAT>
__global__ __launch_bounds__(1024) void clip_kernel(const float* __restrict__ x, int B, int T, const float* __restrict__ pct,
                                                    const uint8_t* __restrict__ apply, float* __restrict__ y) {
  __shared__ unsigned hist[4][2048];
  __shared__ unsigned prefix[4], want[4];
  __shared__ int uniq[4];                     // uniq[r] = the first rank with the same prefix: ranks that share a prefix share ONE histogram
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* xb = x + (size_t)b * T;
  float* yb = y + (size_t)b * T;
  if (!apply[b]) {
    for (int i = tid; i < T; i += 1024) yb[i] = xb[i];
    return;
  }
  int nsel = 1;
  if (FLAT) {
    nsel = 0;
    for (int r = 0; r < B; ++r) nsel += apply[r] != 0;
  }
  const long long n = (long long)nsel * T;
  const float qlo = pct[b] / 2.f, qhi = 1.f - qlo;
  const float rlo = qlo * (float)(n - 1), rhi = qhi * (float)(n - 1);      // torch.quantile: rank = q * (n - 1), in float32
  const int ranks[4] = {(int)floorf(rlo), (int)ceilf(rlo), (int)floorf(rhi), (int)ceilf(rhi)};
  if (tid < 4) { prefix[tid] = 0; want[tid] = (unsigned)ranks[tid]; }
  const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
  for (int pass = 0; pass < 3; ++pass) {
    const int nb = 1 << bits[pass];
    for (int i = tid; i < 4 * 2048; i += 1024) (&hist[0][0])[i] = 0;
    if (tid < 4) {
      // all four ranks share the (empty) prefix in pass 0, and floor / ceil of one rank usually share it later: counting an element once per
      // DISTINCT prefix instead of once per rank cuts the LDS atomics of pass 0 -- where most samples of a clip fall into a few exponent
      // bins and serialise -- by four (352 -> ... us per 64 clips)
      int u = tid;
      for (int r = tid - 1; r >= 0; --r) if (prefix[r] == prefix[tid]) u = r;
      uniq[tid] = u;
    }
    __syncthreads();
    const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shifts[pass] + bits[pass]));
    const bool own0 = true, own1 = uniq[1] == 1, own2 = uniq[2] == 2, own3 = uniq[3] == 3;
    const unsigned p0 = prefix[0], p1 = prefix[1], p2 = prefix[2], p3 = prefix[3];
    for (int row = FLAT ? 0 : b; row < (FLAT ? B : b + 1); ++row) {
      if (FLAT && !apply[row]) continue;
      const float* xr = x + (size_t)row * T;
      for (int i = tid; i < T; i += 1024) {
        const unsigned k = f2key(xr[i]);
        const unsigned digit = (k >> shifts[pass]) & (nb - 1);
        const unsigned kh = k & himask;
        if (own0 && kh == p0) atomicAdd(&hist[0][digit], 1u);
        if (own1 && kh == p1) atomicAdd(&hist[1][digit], 1u);
        if (own2 && kh == p2) atomicAdd(&hist[2][digit], 1u);
        if (own3 && kh == p3) atomicAdd(&hist[3][digit], 1u);
      }
    }
    __syncthreads();
    if (tid < 4) {                                   // walk the histogram to the bin holding the wanted rank
      unsigned acc = 0, w = want[tid];
      int d = 0;
      for (; d < nb; ++d) {
        const unsigned c = hist[uniq[tid]][d];
        if (acc + c > w) break;
        acc += c;
      }
      want[tid] = w - acc;
      prefix[tid] |= (unsigned)d << shifts[pass];
    }
    __syncthreads();
  }
  const float v0 = key2f(prefix[0]), v1 = key2f(prefix[1]), v2 = key2f(prefix[2]), v3 = key2f(prefix[3]);
  auto lerp = [](float a, float c, float w) { return w < 0.5f ? a + w * (c - a) : c - (c - a) * (1.f - w); };
  const float lo = lerp(v0, v1, rlo - floorf(rlo)), hi = lerp(v2, v3, rhi - floorf(rhi));
  for (int i = tid; i < T; i += 1024) {
    const float v = xb[i];
    yb[i] = fminf(fmaxf(v, lo), hi);
  }
}

}  // namespace

extern "C" {

int mfpa_lowpass_taps(const float* cutoff, const int* half, const long long* tap_off, int B, float* taps, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!cutoff || !half || !tap_off || !taps || B < 0) return MFPA_EINVAL;
  hipLaunchKernelGGL(lowpass_taps_kernel, dim3(B), dim3(256), 0, mfpa_stream(stream), cutoff, half, tap_off, taps);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_fir(const float* x, int B, int T, int Tout, const float* taps, const long long* tap_off, const int* ntaps, const int* off,
             const uint8_t* apply, int pad_mode, int out_mode, float* y, float* peak, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!x || !taps || !tap_off || !ntaps || !off || !apply || !y || B < 0 || B > 65535 || T < 1 || Tout < T) return MFPA_EINVAL;
  if (pad_mode < 0 || pad_mode > 1 || out_mode < 0 || out_mode > 2 || (out_mode == 2 && !peak)) return MFPA_EINVAL;
  if ((long long)Tout + FIR_OUT > 0x7fffffffLL) return MFPA_EINVAL;
  hipStream_t s = mfpa_stream(stream);
  if (out_mode == 2) MFPA_HIP(hipMemsetAsync(peak, 0, sizeof(float) * B, s));
  dim3 grid((Tout + FIR_OUT - 1) / FIR_OUT, B);
  hipLaunchKernelGGL(fir_kernel, grid, dim3(256), 0, s, x, T, Tout, taps, tap_off, ntaps, off, apply, pad_mode, out_mode, y, peak);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_scale_rows(const float* x, int B, int T, const float* factor, const uint8_t* apply, int invert, float* y, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!x || !factor || !y || B < 0 || B > 65535 || T < 1) return MFPA_EINVAL;
  int gx = (T + 255) / 256; if (gx > 256) gx = 256;
  hipLaunchKernelGGL(scale_rows_kernel, dim3(gx, B), dim3(256), 0, mfpa_stream(stream), x, T, factor, apply, invert, y);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_gather_background(const float* bank, const long long* src, const int* len, int B, int P, int T, float* out,
                           void* stream) {
  if (B == 0) return MFPA_OK;
  if (!bank || !src || !len || !out || B < 0 || P < 1 || T < 1) return MFPA_EINVAL;
  hipLaunchKernelGGL(gather_background_kernel, dim3(B), dim3(1024), 0, mfpa_stream(stream), bank, src, len, P, T, out);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_mix_background(const float* x, int B, int T, const float* noise, const float* snr_db, const uint8_t* apply, float* y,
                        void* stream) {
  if (B == 0) return MFPA_OK;
  if (!x || !y || B < 0 || T < 1 || (noise && !snr_db)) return MFPA_EINVAL;
  hipLaunchKernelGGL(mix_kernel, dim3(B), dim3(1024), 0, mfpa_stream(stream), x, T, noise, snr_db, apply, y);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_clip_quantile(const float* x, int B, int T, const float* pct, const uint8_t* apply, float* y, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!x || !pct || !apply || !y || B < 0 || T < 2) return MFPA_EINVAL;
  hipLaunchKernelGGL(clip_kernel<false>, dim3(B), dim3(1024), 0, mfpa_stream(stream), x, B, T, pct, apply, y);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_clip_quantile_flat(const float* x, int B, int T, const float* pct, const uint8_t* apply, int n_apply, float* y,
                            void* stream) {
  if (B == 0) return MFPA_OK;
  if (!x || !pct || !apply || !y || B < 0 || T < 2 || n_apply < 0 || n_apply > B) return MFPA_EINVAL;
  if ((long long)n_apply * T > 16000000LL) return MFPA_EINVAL;       // torch.quantile's own input limit
  hipLaunchKernelGGL(clip_kernel<true>, dim3(B), dim3(1024), 0, mfpa_stream(stream), x, B, T, pct, apply, y);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

}  // extern "C"
