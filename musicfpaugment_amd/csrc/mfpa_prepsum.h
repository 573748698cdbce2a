// Log values + the node sums of np.mean's pairwise tree, many workgroups per clip: the first launch of the fused pickers
// (mfpa_audfprint_pick in audfprint.hip, mfpa_dejavu_pick in dejavu.hip; both compiled with -ffp-contract=off).
#pragma once
#include "mfpa_common.h"
#include "mfpa_fastlog.h"
#include "mfpa_npsum.h"

namespace mfpa_prepsum {
namespace {                      // every translation unit that includes this gets its own copy of the kernel

using namespace mfpa_np;

// First launch of the fused picker for the path whose per-clip maximum is already known (config 2's STFT -> peak-pick chain); same
// arithmetic as prepare_kernel<double>.  Workgroup (clip, chunk, half) = one child of the root of one 8192-element chunk of numpy's
// pairwise tree (memory order = mean_order): the raw values of its <= 4104 elements are gathered into LDS in that order (read from
// `spec` along the frames, whatever the order), turned into log values there and summed in numpy's order; the log values go out
// frame-major (fm = 1: coalesced) or bin-major (fm = 0), the node sum beside them.  16 workgroups per clip instead of one: the stage is
// instruction-bound (a float64 division and a float64 log per element), so what counts is that the whole chip works on it.
// (A 16-bin-per-workgroup filter kernel completed this into a two-launch replacement of prepare_kernel; it measured 75 us per 256 clips,
//  the pair 190-230 us against the single kernel's 176 -- the pruner now filters the frames itself, mfpa_audfprint_pick.)
// 512 threads: the node's 38 KB of LDS allow four workgroups per CU -- 32 resident waves with 512 threads, 16 with 256 (measured:
// pick stage 365 -> 354 us per 256 clips; the kernel needs 56 registers, inside the 64 that eight waves per SIMD leave each)
#ifndef MFPA_PREP_FASTDIV
#define MFPA_PREP_FASTDIV 1       // 0: the IEEE division per element (A/B builds)
#endif
constexpr int SPLIT_THREADS = 512, SPLIT_RPP = SPLIT_THREADS / 16;   // threads; rows per pass of the 16-lane-per-row mappings
constexpr int SPLIT_MAX_T = 512;

// v / den for 0 <= v <= den with y = RN(1 / den) (one true division per workgroup), CORRECTLY ROUNDED like the division it replaces:
// q0 = RN(v y); two residual / correction steps r = v - den q (exact in an fma), q <- RN(q + r y).  After the first step q is a/b + O(2^-2p)
// rounded, i.e. within one ulp (faithful); Markstein's theorem (1990; Muller et al., Handbook of Floating-Point Arithmetic, "division
// by a correctly rounded reciprocal") then makes the second step's result RN(v / den), absent underflow -- the caller takes this path only
// for 1e-200 < den < 1e300, where every residual of a quotient above the floor (1e-6) is a normal number; quotients below the floor are
// replaced by it, so their last bit is immaterial.  Five multiply-adds instead of the ~11-instruction division sequence with its
// quarter-rate v_rcp_f64 (tests/test_gpu_peaks.py compares it with IEEE division on 2^24 pairs, near-ties included).
__device__ __forceinline__ double prep_div_fast(double v, double den, double y) {
  const double q0 = v * y;
  const double q1 = __builtin_fma(__builtin_fma(-q0, den, v), y, q0);
  return __builtin_fma(__builtin_fma(-q1, den, v), y, q1);
}

__device__ __forceinline__ double prep_log_value(double v, double den, double rden, bool do_log, double floor_v, const double (*tab)[3]) {
  double s = rden != 0.0 ? prep_div_fast(v, den, rden) : v / den;      // (rden = 0: the caller asks for the true division; uniform)
  if (do_log) {
    s = s > floor_v ? s : floor_v;
    s = mfpa_log_t(s, tab);
  }
  return s;
}

// numpy pairwise sum of n <= 8192 doubles held in LDS (vals[0 .. n)); all threads call; heap: HEAP doubles of LDS
__device__ __forceinline__ double lds_pairwise_sum(const double* vals, int n, double* heap, int tid, int nthreads) {
  const int lane8 = tid & 7, grp = tid >> 3;
  for (int id = 1 + grp; id < HEAP; id += nthreads / 8) {
    const NodeInfo nd = pw_node(n, id);
    if (!nd.exists || nd.n > PW_BLOCK) continue;
    const double* a = vals + nd.off;
    double res;
    if (nd.n < 8) {
      res = 0;
      for (int i = 0; i < nd.n; ++i) res = res + a[i];
    } else {
      const int n8 = nd.n - (nd.n % 8);
      double acc = a[lane8];
      for (int j = 1; 8 * j < n8; ++j) acc = acc + a[8 * j + lane8];
      res = group8_sum(acc);
      for (int i = n8; i < nd.n; ++i) res = res + a[i];
    }
    if (lane8 == 0) heap[id] = res;
  }
  __syncthreads();
  for (int d = 6; d >= 0; --d) {
    for (int k = tid; k < (1 << d); k += nthreads) {
      const int id = (1 << d) + k;
      const NodeInfo nd = pw_node(n, id);
      if (nd.exists && nd.n > PW_BLOCK) heap[id] = heap[2 * id] + heap[2 * id + 1];
    }
    __syncthreads();
  }
  return heap[1];
}

// fm = 0: log values to Lout bin-major (b, F, T) (row F - 1 not written); fm = 1 (mean_order 1 only): frame-major (b, T, F), i.e. in
// numpy's memory order, written coalesced from LDS.  Node sums to sums_out[b * sum_stride + 2 * chunk + half].
// `scale`: the log values are multiplied by it before they are stored and summed (Dejavu's 10 * np.log; 1.0 = no multiplication).
// `out_rows`: bin-major output (fm = 0) keeps rows [0, out_rows) -- F - 1 for Audfprint (the Nyquist bin is dropped), F for Dejavu.
__global__ __launch_bounds__(SPLIT_THREADS) void prep_sum_kernel(const double* __restrict__ spec, int F, int T,
                                                                 const double* __restrict__ denom, int mean_order,
                                                                 double* __restrict__ Lout, int fm, double* __restrict__ sums_out,
                                                                 long long sum_stride, double scale, int out_rows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* vals = reinterpret_cast<double*>(smem);              // [<= 4096 + 8]
  double* heap = vals + NPY_BUFSIZE / 2 + 8;                   // [HEAP]
  const int tid = threadIdx.x, b = blockIdx.x, c = blockIdx.y >> 1, half = blockIdx.y & 1;
  const int N = F * T;
  const int cn = min(NPY_BUFSIZE, N - c * NPY_BUFSIZE);
  double* L = Lout + (size_t)b * N;
  double* out = sums_out + (size_t)b * sum_stride + 2 * c + half;
  // this workgroup's node of the chunk's tree: the chunk itself when it is a single leaf (then `half` 1 has nothing to do)
  int off = 0, n = cn;
  if (cn > PW_BLOCK) {
    int n2 = cn / 2;
    n2 -= n2 % 8;
    off = half ? n2 : 0;
    n = half ? cn - n2 : n2;
  } else if (half) {
    if (tid == 0) *out = 0.0;
    return;
  }
  const double den = denom[b];
  const double smax = den > 0.0 ? 1.0 : (double)NAN;          // prepare_kernel: denom[b] is this clip's own maximum
  const bool do_log = smax > 0.0;
  const double floor_v = smax / 1e6;
  // the reciprocal for prep_div_fast: only where its premises hold (a positive, finite, not tiny maximum; the values are magnitudes <= it)
  const double rden = (MFPA_PREP_FASTDIV && do_log && den > 1e-200 && den < 1e300) ? 1.0 / den : 0.0;
  const double* x = spec + (size_t)b * N;
  const int e0 = c * NPY_BUFSIZE + off;                        // first element (memory order) of the node
  // the log table into LDS (three dependent-address global loads per logarithm would put a memory round trip into every call); its loads
  // are issued here and stored after the gather's, so that the two memory round trips overlap
  double (*tab)[3] = reinterpret_cast<double (*)[3]>(heap + HEAP);
  constexpr int TABN = 128 * 3;
  double tabv[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) tabv[u] = (&mfpa_log_tab[0][0])[min(tid + u * SPLIT_THREADS, TABN - 1)];
  if (mean_order == 0) {
    // memory order = the layout of `spec`: the node is one contiguous piece; eight independent loads in flight per thread
    for (int i0 = tid; i0 < n; i0 += 8 * SPLIT_THREADS) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = x[e0 + min(i0 + u * SPLIT_THREADS, n - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * SPLIT_THREADS < n) vals[i0 + u * SPLIT_THREADS] = v[u];
    }
  } else {
    // memory order e = t * F + f: the node covers frames t0 .. t1 (the first and the last partly, nt <= 32 of them).  Gather along the
    // frames of `spec` into the node's own order in LDS.  Every load is unconditional (clamped row / frame; only the LDS store is
    // predicated): loads inside exec-masked branches are waited for one by one.  Three pieces, so that (almost) every lane carries
    // an element: (a) whole blocks of 16 frames x the rows below Fm = F - F % 16, a 16-lane group per row, 16 rows per pass, eight
    // passes in flight; (b) the nt % 16 frames left over, a lane per row; (c) the F % 16 rows left over (the Nyquist row at F = 257).
    const int t0 = e0 / F, t1 = (e0 + n - 1) / F, nt = t1 - t0 + 1;
    if (nt > 32) return;                                       // (cannot happen: the launcher takes F >= 141 only)
    const int Fm = F & ~15, nb = nt >> 4, rem = nt & 15;
    const int tt = tid & 15, fr = tid >> 4;
    auto put = [&](int t, int f, double v) {
      const int e = t * F + f - e0;
      if (e >= 0 && e < n) vals[e] = v;
    };
    for (int tb = 0; tb < nb; ++tb) {                          // (a)
      const int t = t0 + 16 * tb + tt;                         // < T: a whole block lies inside the node's frames
      const double* src = x + t;
      for (int f0 = fr; f0 < Fm; f0 += 8 * SPLIT_RPP) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)min(f0 + SPLIT_RPP * u, Fm - 1) * T];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (f0 + SPLIT_RPP * u < Fm) put(t, f0 + SPLIT_RPP * u, v[u]);
      }
    }
    for (int r = 0; r < rem; ++r) {                            // (b)
      const int t = t0 + 16 * nb + r;
      for (int f0 = tid; f0 < Fm; f0 += 2 * SPLIT_THREADS) {
        double v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) v[u] = x[(size_t)min(f0 + u * SPLIT_THREADS, Fm - 1) * T + t];
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (f0 + u * SPLIT_THREADS < Fm) put(t, f0 + u * SPLIT_THREADS, v[u]);
      }
    }
    for (int i = tid; i < (F - Fm) * nt; i += SPLIT_THREADS) { // (c)
      const int f = Fm + i / nt, t = t0 + i % nt;
      put(t, f, x[(size_t)f * T + t]);
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (tid + u * SPLIT_THREADS < TABN) (&tab[0][0])[tid + u * SPLIT_THREADS] = tabv[u];
  __syncthreads();                                             // raw values in memory order, the table in LDS
  // s = x / max, floor, log: in the node's memory order (what the pairwise sum reads).  Written out frame-major as the contiguous piece
  // of (T, F) the node is (fm = 1: coalesced), or bin-major (fm = 0): directly when that is the memory order, else from LDS after
  // the loop, transposed -- 16 lanes along the frames of one bin, so that a bin's 16 log values leave as one 128-byte piece
  // (an element at a time in memory order every lane wrote into a different row of L).
  const bool transposed_store = !fm && mean_order != 0;
  for (int i = tid; i < n; i += SPLIT_THREADS) {
    double lv = prep_log_value(vals[i], den, rden, do_log, floor_v, tab);
    if (scale != 1.0) lv = scale * lv;                         // (uniform; x * 1.0 would be exact as well)
    vals[i] = lv;
    const int e = e0 + i;
    if (fm) {
      L[e] = lv;
    } else if (!transposed_store) {
      if (e < out_rows * T) L[e] = lv;                         // bin-major like `spec`
    }
  }
  __syncthreads();
  if (transposed_store) {
    const int t0 = e0 / F, t1 = (e0 + n - 1) / F;
    const int tt = tid & 15, fr = tid >> 4;
    for (int tb = t0; tb <= t1; tb += 16) {
      const int t = tb + tt;
      if (t1 - tb >= 4) {                                      // a block of frames: a 16-lane group per bin
        for (int f = fr; f < out_rows; f += SPLIT_THREADS / 16) {
          const int e = t * F + f - e0;
          if (t <= t1 && e >= 0 && e < n) L[(size_t)f * T + t] = vals[e];
        }
      } else {                                                 // the one to four frames left over: a lane per bin
        for (int tl = tb; tl <= t1; ++tl)
          for (int f = tid; f < out_rows; f += SPLIT_THREADS) {
            const int e = tl * F + f - e0;
            if (e >= 0 && e < n) L[(size_t)f * T + tl] = vals[e];
          }
      }
    }
  }
  const double r = lds_pairwise_sum(vals, n, heap, tid, SPLIT_THREADS);
  if (tid == 0) *out = r;
}

}  // namespace
}  // namespace mfpa_prepsum
