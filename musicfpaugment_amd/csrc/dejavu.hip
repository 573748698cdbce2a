// Dejavu 2-D local-maximum peak picker for MI355X (gfx950).
//
//   prepare   : arr = scale * ln(max(a, max(a)/1e6)) - mean, a = psd / denom
//               (afp/dejavu/fingerprint.py:68,78-79); np.mean's summation order reproduced.
//   localmax2d: get_2D_peaks (fingerprint.py:94-171): (2r+1)^2 maximum filter with scipy
//               'reflect' borders, (filtered == value), XOR with the erosion of the exact-zero
//               background (border_value 1), amplitude > amp_min.  Only comparisons: bit-exact
//               by construction for a given arr.
//
// localmax2d tiles the (F, T) plane: a workgroup stages its tile plus a radius-wide halo in LDS
// (values with reflected indices for the maximum, a background flag with the constant-1 border
// for the erosion), then runs the separable row pass and column pass out of LDS.
// Compiled with -ffp-contract=off (the pre-processing feeds exact comparisons).
#include "mfpa_common.h"
#include "mfpa_fastlog.h"
#include "mfpa_npsum.h"
#include "mfpa_prepsum.h"

namespace {

using namespace mfpa_np;
using namespace mfpa_prepsum;
constexpr int PREP_THREADS = 1024;   // one workgroup per clip: as many waves as a workgroup can have (latency-bound streaming passes)

// TIn = double: the un-denoised path (psd / max, float64 throughout).  TIn = float: the UNet path (fingerprint.py:70-79) --
// the network's float32 output is squared and every later step (max, floor max / 1e6, 10 * log, mean, subtraction) stays in
// float32 like numpy on a float32 array; the result is widened to float64 for the peak picker (comparisons are unchanged).
template <typename TIn>
__global__ __launch_bounds__(PREP_THREADS) void dejavu_prepare_kernel(const TIn* __restrict__ psd, int F, int T,
                                                                      const double* __restrict__ denom, double scale,
                                                                      int mean_order, int square, double* __restrict__ arr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TIn* heap = reinterpret_cast<TIn*>(smem);
  __shared__ double red[PREP_THREADS / 64];
  __shared__ double bcast[2];
  __shared__ double logtab[128][3];                  // the log table in LDS (three dependent-address global loads per logarithm otherwise)
  for (int i = threadIdx.x; i < 128 * 3; i += PREP_THREADS) (&logtab[0][0])[i] = (&mfpa_log_tab[0][0])[i];
  __syncthreads();
  const int tid = threadIdx.x, b = blockIdx.x;
  const int N = F * T;
  const TIn* x = psd + (size_t)b * N;
  double* L = arr + (size_t)b * N;
  const bool has_den = denom != nullptr;
  const double den = has_den ? denom[b] : 1.0;
  auto value = [&](TIn v) __attribute__((always_inline)) -> TIn {
    if (has_den) return (TIn)((double)v / den);
    return square ? v * v : v;
  };

  // 8 independent loads in flight per thread in every streaming pass (one workgroup per clip: memory-latency-bound otherwise)
  double m = -INFINITY;
  for (int i0 = tid; i0 < N; i0 += 8 * PREP_THREADS) {
    TIn v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x[min(i0 + u * PREP_THREADS, N - 1)];      // unconditional (clamped) loads: masked ones are waited for one by one
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u * PREP_THREADS < N) {
        const double s = (double)value(v[u]);
        m = s > m ? s : m;
      }
  }
  m = mfpa_wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    double mm = red[0];
    for (int w = 1; w < PREP_THREADS / 64; ++w) mm = fmax(mm, red[w]);
    bcast[0] = mm;
  }
  __syncthreads();
  const TIn floor_v = (TIn)bcast[0] / (TIn)1e6;
  const TIn sc = (TIn)scale;
  for (int i0 = tid; i0 < N; i0 += 8 * PREP_THREADS) {
    TIn v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x[min(i0 + u * PREP_THREADS, N - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u * PREP_THREADS < N) {
        TIn s = value(v[u]);
        s = s > floor_v ? s : floor_v;
        L[i0 + u * PREP_THREADS] = (double)(sc * (TIn)mfpa_log_t((double)s, logtab));   // float32: the float64 log rounded once (as audfprint.hip)
      }
  }
  __syncthreads();
  const TIn total = block_numpy_sum<TIn>(L, N, F, T, mean_order, heap, &bcast[1], tid, PREP_THREADS);
  const TIn mean = total / (TIn)N;
  for (int i0 = tid; i0 < N; i0 += 8 * PREP_THREADS) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = L[min(i0 + u * PREP_THREADS, N - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u * PREP_THREADS < N) L[i0 + u * PREP_THREADS] = (double)((TIn)v[u] - mean);
  }
}

// 1024 threads for a 32 x 64 tile: the 69 KB of LDS allow two workgroups per CU whatever their size, and the kernel is a chain of
// dependent round trips (halo loads, two LDS passes) -- 32 resident waves hide them better than 8 (measured per 256 clips, with all
// halo loads of a thread in one batch: 256 threads 203 us, 512: 173 us, 1024: 158 us).  LM_U = halo elements per thread (one batch).
constexpr int TH = 32, TW = 64, LM_THREADS = 1024, LM_U = 5;

__device__ __forceinline__ int reflect_index(int i, int n) {  // scipy.ndimage mode='reflect'
  if (i >= 0 && i < n) return i;                  // the common case without an integer division
  if (i < 0 && i >= -n) return -1 - i;
  if (i >= n && i < 2 * n) return 2 * n - 1 - i;
  const int period = 2 * n;
  i %= period;
  if (i < 0) i += period;
  return i >= n ? period - 1 - i : i;
}

// RR: the radius as a compile-time constant (10 = Dejavu's PEAK_NEIGHBORHOOD_SIZE: the index divisions by the halo width become
// multiplications and the window loops unroll), 0 = the run-time value `r_`.
// node_sums != nullptr (mfpa_dejavu_pick): `arr` holds the values BEFORE the mean is subtracted and node_sums the sums of the nodes of
// np.mean's tree (prep_sum_kernel); every workgroup forms the clip's mean from them exactly as dejavu_prepare_kernel does (acc = 0;
// acc += pairwise(chunk) for every chunk; / N) and subtracts it while the halo tile is loaded -- the mean-subtracted array never exists.
template <int RR>
__global__ __launch_bounds__(LM_THREADS) void localmax2d_kernel(const double* __restrict__ arr, int F, int T, int r_,
                                                                double amp_min, uint8_t* __restrict__ mask,
                                                                int32_t* __restrict__ npeaks, const double* __restrict__ node_sums) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int r = RR ? RR : r_;
  const int HH = TH + 2 * r, HW = TW + 2 * r;
  double* A = reinterpret_cast<double*>(smem);     // [HH][HW] values, reflected
  double* Hm = A + HH * HW;                        // [HH][TW] row-pass maxima
  uint8_t* Bg = reinterpret_cast<uint8_t*>(Hm + HH * TW);  // [HH][HW] background flag (out of bounds = 1)
  uint8_t* Hb = Bg + HH * HW;                      // [HH][TW] row-pass AND

  const int tid = threadIdx.x, b = blockIdx.z;
  const int i0 = blockIdx.y * TH, j0 = blockIdx.x * TW;
  const double* X = arr + (size_t)b * F * T;
  double mean = 0.0;
  const bool sub_mean = node_sums != nullptr;
  if (sub_mean) {
    const int N = F * T, nchunks = (N + NPY_BUFSIZE - 1) / NPY_BUFSIZE;
    double total = 0.0;
    for (int c = 0; c < nchunks; ++c) {
      const int cn = min(NPY_BUFSIZE, N - c * NPY_BUFSIZE);
      const double* h = node_sums + ((size_t)b * MAX_CHUNKS + c) * 2;
      total = total + (cn > PW_BLOCK ? h[0] + h[1] : h[0]);
    }
    mean = total / (double)N;
  }

  // halo tile: all of a thread's loads in flight at once (one per trip left every element a full memory round trip: 17 in a row with
  // 256 threads); slots past the end re-read the last element and are not stored
  for (int e0 = tid; e0 < HH * HW; e0 += LM_U * LM_THREADS) {
    double v[LM_U];
    int gi[LM_U], gj[LM_U];
#pragma unroll
    for (int u = 0; u < LM_U; ++u) {
      const int e = min(e0 + u * LM_THREADS, HH * HW - 1);
      const int hi = e / HW, hj = e - hi * HW;
      gi[u] = i0 - r + hi; gj[u] = j0 - r + hj;
      v[u] = X[(size_t)reflect_index(gi[u], F) * T + reflect_index(gj[u], T)];
    }
    if (sub_mean) {
#pragma unroll
      for (int u = 0; u < LM_U; ++u) v[u] = v[u] - mean;
    }
#pragma unroll
    for (int u = 0; u < LM_U; ++u) {
      const int e = e0 + u * LM_THREADS;
      if (e < HH * HW) {
        A[e] = v[u];
        const bool inside = gi[u] >= 0 && gi[u] < F && gj[u] >= 0 && gj[u] < T;
        Bg[e] = inside ? (uint8_t)(v[u] == 0.0) : (uint8_t)1;
      }
    }
  }
  __syncthreads();
  // Both passes give every thread FOUR adjacent outputs: the 2r + 4 inputs they share are read from LDS once (6 reads
  // per output instead of 2r + 1) and the window maxima are formed in registers: the 2r - 2 interior values are common
  // to all four windows.
  for (int e = tid; e < HH * (TW / 4); e += LM_THREADS) {
    const int hi = e / (TW / 4), j = 4 * (e % (TW / 4));
    const double* row = A + hi * HW + j;
    const uint8_t* brow = Bg + hi * HW + j;
    double m = row[3];                         // common part: offsets 3 .. 2r
    uint8_t bg = brow[3];
    for (int d = 4; d <= 2 * r; ++d) {
      m = row[d] > m ? row[d] : m;
      bg &= brow[d];
    }
    double lo[3], hi3[3];
    uint8_t blo[3], bhi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { lo[k] = row[k]; hi3[k] = row[2 * r + 1 + k]; blo[k] = brow[k]; bhi[k] = brow[2 * r + 1 + k]; }
    // output o covers offsets o .. o + 2r: the common part plus lo[o..2] and hi3[0..o-1]
    double mo[4];
    uint8_t bo[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      double v = m;
      uint8_t g = bg;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (k >= o) { v = lo[k] > v ? lo[k] : v; g &= blo[k]; }
        if (k < o) { v = hi3[k] > v ? hi3[k] : v; g &= bhi[k]; }
      }
      mo[o] = v; bo[o] = g;
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) { Hm[hi * TW + j + o] = mo[o]; Hb[hi * TW + j + o] = bo[o]; }
  }
  __syncthreads();
  int count = 0;
  for (int e = tid; e < (TH / 4) * TW; e += LM_THREADS) {
    const int i = 4 * (e / TW), j = e % TW;
    const double* col = Hm + i * TW + j;
    const uint8_t* bcol = Hb + i * TW + j;
    double m = col[3 * TW];
    uint8_t bg = bcol[3 * TW];
    for (int d = 4; d <= 2 * r; ++d) {
      const double v = col[d * TW];
      m = v > m ? v : m;
      bg &= bcol[d * TW];
    }
    double lo[3], hi3[3];
    uint8_t blo[3], bhi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      lo[k] = col[k * TW]; hi3[k] = col[(2 * r + 1 + k) * TW];
      blo[k] = bcol[k * TW]; bhi[k] = bcol[(2 * r + 1 + k) * TW];
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      double mm = m;
      uint8_t g = bg;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (k >= o) { mm = lo[k] > mm ? lo[k] : mm; g &= blo[k]; }
        if (k < o) { mm = hi3[k] > mm ? hi3[k] : mm; g &= bhi[k]; }
      }
      const int gi = i0 + i + o, gj = j0 + j;
      if (gi < F && gj < T) {
        const double val = A[(i + o + r) * HW + j + r];
        const bool local_max = (mm == val);
        const bool detected = local_max != (g != 0);
        const bool keep = detected && (val > amp_min);
        mask[((size_t)b * F + gi) * T + gj] = keep ? 1 : 0;
        count += keep ? 1 : 0;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) count += __shfl_xor(count, o);
  if ((tid & 63) == 0 && count) atomicAdd(npeaks + b, count);
}

}  // namespace

extern "C" {

int mfpa_dejavu_prepare(const double* psd, int B, int F, int T, const double* denom, double scale, int mean_order,
                        double* arr, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!psd || !arr || B < 0 || F < 1 || T < 1) return MFPA_EINVAL;
  const long long N = (long long)F * T;
  const long long nchunks = (N + NPY_BUFSIZE - 1) / NPY_BUFSIZE;
  if (nchunks > MAX_CHUNKS) return MFPA_EINVAL;
  hipLaunchKernelGGL(dejavu_prepare_kernel<double>, dim3(B), dim3(PREP_THREADS), sizeof(double) * nchunks * HEAP,
                     mfpa_stream(stream), psd, F, T, denom, scale, mean_order, 0, arr);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_dejavu_prepare_f32(const float* x, int B, int F, int T, int square, double scale, int mean_order, double* arr,
                            void* stream) {
  if (B == 0) return MFPA_OK;
  if (!x || !arr || B < 0 || F < 1 || T < 1) return MFPA_EINVAL;
  const long long N = (long long)F * T;
  const long long nchunks = (N + NPY_BUFSIZE - 1) / NPY_BUFSIZE;
  if (nchunks > MAX_CHUNKS) return MFPA_EINVAL;
  hipLaunchKernelGGL(dejavu_prepare_kernel<float>, dim3(B), dim3(PREP_THREADS), sizeof(double) * nchunks * HEAP,
                     mfpa_stream(stream), x, F, T, (const double*)nullptr, scale, mean_order, square, arr);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_localmax2d(const double* arr, int B, int F, int T, int radius, double amp_min, uint8_t* mask,
                    int32_t* npeaks, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!arr || !mask || !npeaks || B < 0 || F < 1 || T < 1 || radius < 2 || radius > 16) return MFPA_EINVAL;   // the 4-outputs-per-thread passes need 2r >= 3
  if (B > 65535) return MFPA_EINVAL;
  hipStream_t s = mfpa_stream(stream);
  MFPA_HIP(hipMemsetAsync(npeaks, 0, sizeof(int32_t) * B, s));
  const int HH = TH + 2 * radius, HW = TW + 2 * radius;
  const size_t lds = sizeof(double) * ((size_t)HH * HW + (size_t)HH * TW) + (size_t)HH * HW + (size_t)HH * TW;
  dim3 grid((T + TW - 1) / TW, (F + TH - 1) / TH, B);
  if (radius == 10) hipLaunchKernelGGL(localmax2d_kernel<10>, grid, dim3(LM_THREADS), lds, s, arr, F, T, radius, amp_min, mask, npeaks, (const double*)nullptr);
  else hipLaunchKernelGGL(localmax2d_kernel<0>, grid, dim3(LM_THREADS), lds, s, arr, F, T, radius, amp_min, mask, npeaks, (const double*)nullptr);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_dejavu_pick_work_doubles(int F, int T, long long* per_clip) {
  if (!per_clip || F < 1 || T < 1) return MFPA_EINVAL;
  *per_clip = (long long)F * T + 2 * MAX_CHUNKS;
  return MFPA_OK;
}

int mfpa_dejavu_pick(const double* psd, const double* clip_max, int B, int F, int T, double scale, int mean_order, int radius,
                     double amp_min, double* work, uint8_t* mask, int32_t* npeaks, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!psd || !clip_max || !work || !mask || !npeaks || B < 0 || B > 65535 || radius < 2 || radius > 16) return MFPA_EINVAL;
  if (F < 141 || F > 257 || T < 1 || T > SPLIT_MAX_T || (mean_order != 0 && mean_order != 1)) return MFPA_EINVAL;   // (a half-chunk node spans <= 32 frames)
  const long long N = (long long)F * T;
  const int nchunks = (int)((N + NPY_BUFSIZE - 1) / NPY_BUFSIZE);
  if (nchunks > MAX_CHUNKS) return MFPA_EINVAL;
  hipStream_t s = mfpa_stream(stream);
  MFPA_HIP(hipMemsetAsync(npeaks, 0, sizeof(int32_t) * B, s));
  double* sums = work + (size_t)B * N;
  const size_t lds1 = sizeof(double) * (NPY_BUFSIZE / 2 + 8 + HEAP + 128 * 3);
  hipLaunchKernelGGL(prep_sum_kernel, dim3(B, 2 * nchunks), dim3(SPLIT_THREADS), lds1, s, psd, F, T, clip_max, mean_order, work, 0, sums,
                     (long long)(2 * MAX_CHUNKS), scale, F);
  MFPA_CHECK_LAUNCH();
  const int HH = TH + 2 * radius, HW = TW + 2 * radius;
  const size_t lds = sizeof(double) * ((size_t)HH * HW + (size_t)HH * TW) + (size_t)HH * HW + (size_t)HH * TW;
  dim3 grid((T + TW - 1) / TW, (F + TH - 1) / TH, B);
  if (radius == 10) hipLaunchKernelGGL(localmax2d_kernel<10>, grid, dim3(LM_THREADS), lds, s, (const double*)work, F, T, radius, amp_min, mask, npeaks, (const double*)sums);
  else hipLaunchKernelGGL(localmax2d_kernel<0>, grid, dim3(LM_THREADS), lds, s, (const double*)work, F, T, radius, amp_min, mask, npeaks, (const double*)sums);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

}  // extern "C"
