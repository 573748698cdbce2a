// Audfprint spectral-peak picker for MI355X (gfx950): everything of Audfprint_peaks.find_peaks
// after the optional UNet (afp/audfprint/peak_extractor.py:271-311 of the reference).
//
// Compiled with -ffp-contract=off: every float64 value that takes part in a comparison is
// produced by the same un-fused IEEE add / multiply sequence numpy and scipy execute, so for a
// given log-spectrogram the peak set is bit-identical to the reference's by construction.
//
//   prepare : log(max(s, max/1e6)) - mean   (np.mean's pairwise-summation tree reproduced
//             in the array's memory order), scipy lfilter([1,-1],[1,-0.98]) per bin, output
//             written FRAME-major so the pruner reads whole spectrum columns coalesced.
//   prune   : forward decaying-threshold pass + backward pruning.  The scan over frames is
//             inherently sequential (each column's threshold depends on every earlier peak), so
//             one 64-lane wavefront owns a clip: 4 bins per lane, threshold in registers,
//             per-frame top-k by wavefront arg-max (value desc, bin desc), columns prefetched
//             4 frames ahead.  Latency-bound by design; parallelism comes from clips.
#include "mfpa_common.h"
#include "mfpa_fastlog.h"
#include "mfpa_npsum.h"
#include "mfpa_prepsum.h"

namespace {

using namespace mfpa_np;
using namespace mfpa_prepsum;
constexpr int PREP_THREADS = 512;
constexpr int TT = 16;  // frames per transpose tile

// TIn: dtype of the spectrogram (and of the log / mean arithmetic, as numpy keeps it).
template <typename TIn>
__global__ __launch_bounds__(PREP_THREADS) void prepare_kernel(const TIn* __restrict__ spec, int F, int T,
                                                               const double* __restrict__ denom, int mean_order,
                                                               int log_input, double pole,
                                                               double* __restrict__ filtered,
                                                               double* __restrict__ scratch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int R = F - 1;
  double* tile_in = reinterpret_cast<double*>(smem);   // [R][TT+1]
  double* tile_out = tile_in + R * (TT + 1);           // [TT][R]
  TIn* heap = reinterpret_cast<TIn*>(tile_out + TT * R);  // [chunks][HEAP]
  __shared__ double red[PREP_THREADS / 64];
  __shared__ double bcast[2];
  __shared__ double logtab[128][3];                  // the log table in LDS (three dependent-address global loads per logarithm otherwise)
  for (int i = threadIdx.x; i < 128 * 3; i += PREP_THREADS) (&logtab[0][0])[i] = (&mfpa_log_tab[0][0])[i];
  __syncthreads();

  const int tid = threadIdx.x, b = blockIdx.x;
  const int N = F * T;
  const TIn* x = spec + (size_t)b * N;
  double* L = scratch + (size_t)b * N;
  const bool has_den = denom != nullptr;
  const double den = has_den ? denom[b] : 1.0;

  bool do_log = false;
  TIn floor_v = 0;
  const bool den_is_max = (log_input & 2) != 0 && has_den && sizeof(TIn) == sizeof(double);
  log_input &= 1;
  if (!log_input && den_is_max) {
    // denom[b] IS the maximum of this float64 clip (the caller says so: it came from mfpa_stft_mag with this spectrogram), so the
    // maximum of spec / denom is denom / denom = 1 exactly -- or NaN for an all-zero clip, as numpy's 0 / 0 gives -- and the
    // max pass over the clip is not needed
    const TIn smax = den > 0.0 ? (TIn)1 : (TIn)NAN;
    do_log = smax > (TIn)0;
    floor_v = smax / (TIn)1e6;
  } else if (!log_input) {
    // pass A: max of the (normalised) spectrogram
    // One workgroup streams its clip: keep 8 independent loads in flight per thread (a single dependent load per
    // iteration leaves the CU waiting on memory latency: 8 waves x 1 load = 4 KB in flight).
    TIn m = -INFINITY;
    constexpr int UN = 8;
    for (int i0 = tid; i0 < N; i0 += UN * PREP_THREADS) {
      TIn v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = i0 + u * PREP_THREADS;
        v[u] = i < N ? x[i] : (TIn)0;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (i0 + u * PREP_THREADS < N) {
          const TIn s = has_den ? (TIn)((double)v[u] / den) : v[u];
          m = s > m ? s : m;
        }
      }
    }
    double md = mfpa_wave_max((double)m);
    if ((tid & 63) == 0) red[tid >> 6] = md;
    __syncthreads();
    if (tid == 0) {
      double mm = red[0];
      for (int w = 1; w < PREP_THREADS / 64; ++w) mm = fmax(mm, red[w]);
      bcast[0] = mm;
    }
    __syncthreads();
    const TIn smax = (TIn)bcast[0];
    do_log = smax > (TIn)0;                  // peak_extractor.py:274
    floor_v = smax / (TIn)1e6;               // np.max(sgram) / 1e6 in the array's dtype
  }
  // pass B: log values (kept in the spectrogram's dtype, stored widened) in the input's bin-major layout
  for (int i0 = tid; i0 < N; i0 += 8 * PREP_THREADS) {
    TIn v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * PREP_THREADS;
      v[u] = i < N ? x[i] : (TIn)1;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * PREP_THREADS;
      if (i < N) {
        TIn s = has_den ? (TIn)((double)v[u] / den) : v[u];
        if (do_log) {
          s = s > floor_v ? s : floor_v;
          s = (TIn)mfpa_log_t((double)s, logtab);
        }
        L[i] = (double)s;
      }
    }
  }
  __syncthreads();

  TIn mean = 0;
  if (do_log || log_input) {
    // numpy mean: chunked pairwise sum in the array's memory order, one divide in the array's dtype
        const TIn total = block_numpy_sum<TIn>(L, N, F, T, mean_order, heap, &bcast[1], tid, PREP_THREADS);
    mean = (TIn)(total / (TIn)N);
  }

  // pass C: x - mean, then y[n] = x[n] + z; z = -x[n] - (-pole) * y[n]  (scipy lfilter, DF-II transposed)
  double z = 0.0;
  const double npole = -pole;
  double* outp = filtered + (size_t)b * T * R;
  // Tiles of TT frames; the next tile's global loads are issued (into registers) before the sequential filter of the
  // current one runs, so the 16 dependent steps per bin hide the memory latency.
  constexpr int TL = 8;                                    // tile elements per thread: R * TT / PREP_THREADS for R = 256
  double pre[TL];
  auto load_tile = [&](int t0) __attribute__((always_inline)) {
    const int nt = min(TT, T - t0);
#pragma unroll
    for (int u = 0; u < TL; ++u) {
      const int e = tid + u * PREP_THREADS;
      const int r = e / TT, tt = e % TT;
      pre[u] = (e < R * TT && tt < nt) ? L[(size_t)r * T + t0 + tt] : 0.0;
    }
  };
  const bool fits = R * TT <= TL * PREP_THREADS;           // otherwise fall back to direct loads below
  if (fits) load_tile(0);
  for (int t0 = 0; t0 < T; t0 += TT) {
    const int nt = min(TT, T - t0);
    if (fits) {
#pragma unroll
      for (int u = 0; u < TL; ++u) {
        const int e = tid + u * PREP_THREADS;
        if (e < R * TT) tile_in[(e / TT) * (TT + 1) + e % TT] = pre[u];
      }
    } else {
      for (int e = tid; e < R * TT; e += PREP_THREADS) {
        const int r = e / TT, tt = e % TT;
        if (tt < nt) tile_in[r * (TT + 1) + tt] = L[(size_t)r * T + t0 + tt];
      }
    }
    __syncthreads();
    if (fits && t0 + TT < T) load_tile(t0 + TT);
    if (tid < R) {
      for (int tt = 0; tt < nt; ++tt) {
        const TIn lv = (TIn)tile_in[tid * (TT + 1) + tt];
        const double xn = (double)(TIn)(lv - mean);
        const double yn = xn + z;
        z = -xn - npole * yn;
        tile_out[tt * R + tid] = yn;
      }
    }
    __syncthreads();
    for (int e = tid; e < R * nt; e += PREP_THREADS) outp[(size_t)t0 * R + e] = tile_out[e];
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------- prune
constexpr int MAXP = 8;

struct Best {
  double v;
  int p;
};

// lexicographic max on (value, bin); p < 0 means "none"
__device__ __forceinline__ Best best_of(Best a, Best b) {
  const bool take_b = (a.p < 0) || (b.p >= 0 && (b.v > a.v || (b.v == a.v && b.p > a.p)));
  return Best{take_b ? b.v : a.v, take_b ? b.p : a.p};      // field by field: a struct select goes through scratch memory
}

// v of lane `src` (wave-uniform index) through v_readlane: a scalar-path broadcast, ~10x the speed of the LDS-crossbar
// shuffle the generic __shfl lowers to
__device__ __forceinline__ double readlane_f64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// One-lane wavefront shifts as DPP moves (GFX9 wave_shr:1 / wave_shl:1): no LDS crossbar round trip.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false),
                          __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false));
}

// Wavefront arg-max on (value, bin).  A frame has only a handful of candidate lanes (0 or 1 in most frames), so the
// candidates are visited one by one from the ballot mask with scalar broadcasts instead of a 6-level butterfly of
// 3 cross-lane shuffles each.  The result is wave-uniform.
__device__ __forceinline__ Best wave_best(Best x) {
  unsigned long long m = __ballot(x.p >= 0);
  Best best{0.0, -1};
  while (m) {
    const int src = __ffsll((long long)m) - 1;
    m &= m - 1;
    best = best_of(best, Best{readlane_f64(x.v, src), __builtin_amdgcn_readlane(x.p, src)});
  }
  return best;
}

// One wavefront per clip (the scan over frames is sequential: every threshold depends on every earlier peak; parallelism exists
// across the 256 bins -- 4 per lane -- and across clips).  At BASELINE's 256 clips that is ONE wave per CU, so what counts is the
// latency of a frame step, not throughput (profiles/r03_prune_sq.md: 55 % of the wave's cycles were waits on dependent LDS round
// trips -- per-frame peak lists, the Gaussian row -- and 154 vector instructions went into a frame pair):
//   * forward: a frame in which no bin exceeds its threshold (35-45 % of the frames) costs four compares and the decay; the
//     neighbour exchange of locmax runs only otherwise; a frame with ONE candidate (most of the rest) takes it straight from the
//     ballot mask -- no arg-max loop;
//   * peaks go into ONE compact list (value, frame << 8 | bin) plus a frame -> first-entry table, 12 bytes per peak instead of
//     [T][8] slots (LDS 25 -> 20 KB per clip: 8 instead of 6 resident clips per CU at large batches);
//   * backward: the entries of frame c-1 and the table cells of frame c-2 are read from LDS while frame c is processed (a
//     two-deep software pipeline: no LDS round trip on the critical path except the Gaussian row of a surviving peak); a frame's
//     entries sit in lanes 0..7 and are visited with scalar broadcasts; the "delete the following peak in the same bin" rule
//     compares against the previous frame's bins held in registers.
// FUSED: `filtered` holds the LOG values frame-major with pitch R + 1 (prep_sum_kernel, fm = 1) and the kernel applies
// "minus mean, 1-pole high-pass along the frames" itself while it walks the frames forward (state in registers, exactly prepare's
// arithmetic) -- the filtered spectrogram never exists in memory.
template <bool FUSED>
__global__ __launch_bounds__(64) void prune_kernel(const double* __restrict__ filtered, int R, int T,
                                                   const double* __restrict__ gauss, double a_dec, int maxpks,
                                                   uint8_t* __restrict__ mask, int32_t* __restrict__ npeaks,
                                                   const double* __restrict__ node_sums, const double* __restrict__ denom, double pole) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int cap = T * maxpks;                                  // most peaks a clip can record
  double* G = reinterpret_cast<double*>(smem);                 // [2R+2]
  double* ev = G + (2 * R + 2);                                // [cap] peak values, in recording order (frame, then rank)
  int* ep = reinterpret_cast<int*>(ev + cap);                  // [cap] frame << 8 | bin; sign bit set = pruned
  short* fs = reinterpret_cast<short*>(ep + cap);              // [T + 2] first entry of every frame, fs[T] = number of entries

  const int lane = threadIdx.x, b = blockIdx.x;
  const int P = FUSED ? R + 1 : R;                             // pitch of a frame
  const double* S = filtered + (size_t)b * T * P;
  const int k0 = 4 * lane;
  double mean = 0.0, zf[4] = {0.0, 0.0, 0.0, 0.0}, lastcol[4] = {0.0, 0.0, 0.0, 0.0};
  const double npole = -pole;
  if (FUSED && denom[b] > 0.0) {                               // np.mean: acc = 0; acc += pairwise(chunk) for every chunk; / N
    const int N = (R + 1) * T, nchunks = (N + NPY_BUFSIZE - 1) / NPY_BUFSIZE;
    double total = 0.0;
    for (int c = 0; c < nchunks; ++c) {
      const int cn = min(NPY_BUFSIZE, N - c * NPY_BUFSIZE);
      const double* h = node_sums + ((size_t)b * MAX_CHUNKS + c) * 2;
      total = total + (cn > PW_BLOCK ? h[0] + h[1] : h[0]);
    }
    mean = total / (double)N;
  }
  // y[n] = x[n] + z; z = -x[n] - (-pole) * y[n] on this lane's four bins (scipy lfilter, DF-II transposed), x = log value - mean
  auto filt = [&](double (&v)[4], double (&z)[4]) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const double xn = v[q4] - mean;
      const double yn = xn + z[q4];
      z[q4] = -xn - npole * yn;
      v[q4] = yn;
    }
  };
  const bool own = k0 < R;  // R % 4 == 0: a lane owns 4 bins or none

  for (int i = lane; i < 2 * R + 1; i += 64) G[i] = gauss[i];
  if (FUSED) {
    // the clip's mask is zeroed here (the stores drain while the frames are scanned) instead of by a memset launch in front of the kernel;
    // the launcher guarantees 16-byte pieces (R * T % 16 == 0, aligned base)
    uint4* Z = reinterpret_cast<uint4*>(mask + (size_t)b * R * T);
    for (int i = lane; i < R * T / 16; i += 64) Z[i] = uint4{0u, 0u, 0u, 0u};
  }
  __syncthreads();

  // Unconditional loads from clamped addresses (frames past the end re-read frame T - 1, lanes without bins read bins 0..3; neither is
  // ever used): with the loads inside exec-masked branches hipcc could not count them across the loop and put s_waitcnt vmcnt(0)
  // right behind the NEXT group's loads, in front of the first use of the current one -- every group of four frames paid a full
  // memory round trip (a third of the forward pass at one wave per CU).
  const double* Sk = S + (own ? k0 : 0);
  auto load_col = [&](int c, double (&v)[4]) {
    const int cc = c < T ? c : T - 1;
    if (FUSED) {                                               // pitch R + 1 doubles: 8-byte aligned only
      const double* q = Sk + (size_t)cc * P;
      v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
    } else {
      const double2 a = *reinterpret_cast<const double2*>(Sk + (size_t)cc * R);
      const double2 d = *reinterpret_cast<const double2*>(Sk + (size_t)cc * R + 2);
      v[0] = a.x; v[1] = a.y; v[2] = d.x; v[3] = d.y;
    }
  };
  // locmax flags of a column held 4 bins per lane (peak_extractor.py:61-73)
  auto locmax4 = [&](const double (&v)[4], bool (&pk)[4]) {
    const double left = dpp_f64<0x138>(v[3]);   // wave_shr:1 -> lane - 1's value: bin k0-1 (lane 0 keeps its own, unused)
    const double right = dpp_f64<0x130>(v[0]);  // wave_shl:1 -> lane + 1's value: bin k0+4 (lane 63: unused)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = k0 + s;
      const double lft = (s == 0) ? left : v[s - 1];
      const double rgt = (s == 3) ? right : v[s + 1];
      const bool ge_prev = (k == 0) || (v[s] >= lft);
      const bool nxt_ge = (k < R - 1) && (rgt >= v[s]);
      pk[s] = own && ge_prev && !nxt_ge;
    }
  };
  double th[4];
  // th[k] = max(th[k], val * G[k - p + R]) for this lane's bins
  auto raise = [&](double val, int p) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (own) {
        const double g = val * G[R - p + k0 + s];
        th[s] = g > th[s] ? g : th[s];
      }
    }
  };
  // spreadpeaksinvector(vec, f_sd): zeros raised by every local maximum of vec
  auto spread_init = [&](const double (&v)[4]) {
    bool pk[4];
    locmax4(v, pk);
#pragma unroll
    for (int s = 0; s < 4; ++s) th[s] = 0.0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      unsigned long long m = __ballot(pk[s]);
      while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const double val = readlane_f64(v[s], src);
        raise(val, 4 * src + s);
      }
    }
  };
  int ne = 0;                                                  // entries recorded so far (wave-uniform)
  auto record = [&](int c, double val, int p) {
    ev[ne] = val;                                              // every lane stores the same (wave-uniform) value to the same address:
    ep[ne] = (c << 8) | p;                                     // cheaper than switching the exec mask to one lane and back
    ++ne;
  };

  // ---- forward pass (peak_extractor.py:173-204)
  {
    double v10[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    const int n10 = T < 10 ? T : 10;
    double z10[4] = {0.0, 0.0, 0.0, 0.0};                      // FUSED: the filter over the first frames, run again from zero by the main loop
    for (int c = 0; c < n10; ++c) {
      double v[4];
      load_col(c, v);
      if (FUSED) filt(v, z10);
#pragma unroll
      for (int s = 0; s < 4; ++s) v10[s] = v[s] > v10[s] ? v[s] : v10[s];
    }
    spread_init(v10);
  }
  double cur[4][4], nxt[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) load_col(q, cur[q]);
  for (int c0 = 0; c0 < T; c0 += 4) {
#pragma unroll
    for (int q = 0; q < 4; ++q) load_col(c0 + 4 + q, nxt[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + q;
      if (c < T) {
        if (FUSED) {
          filt(cur[q], zf);
          if (c == T - 1) {
#pragma unroll
            for (int s = 0; s < 4; ++s) lastcol[s] = cur[q][s];
          }
        }
        fs[c] = (short)ne;                                       // (uniform store, as in record)
        bool ex[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) ex[s] = own && (cur[q][s] > th[s]);    // against the pre-update threshold
        if (__ballot(ex[0] || ex[1] || ex[2] || ex[3]) != 0ull) {           // else: no candidate in this frame, whatever locmax says
          bool cand[4];
          locmax4(cur[q], cand);
          unsigned long long m[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            cand[s] = cand[s] && ex[s];
            m[s] = __ballot(cand[s]);
          }
          const int total = __popcll(m[0]) + __popcll(m[1]) + __popcll(m[2]) + __popcll(m[3]);
          if (total == 1) {                                                  // the common case: take it from the masks
            const int s1 = m[0] ? 0 : (m[1] ? 1 : (m[2] ? 2 : 3));
            const unsigned long long mm = m[0] | m[1] | m[2] | m[3];
            const int src = __ffsll((long long)mm) - 1;
            const double sel = s1 == 0 ? cur[q][0] : (s1 == 1 ? cur[q][1] : (s1 == 2 ? cur[q][2] : cur[q][3]));
            const double val = readlane_f64(sel, src);
            const int p = 4 * src + s1;
            raise(val, p);
            record(c, val, p);
          } else if (total > 1) {
            int cnt = 0;
            while (cnt < maxpks) {
              Best mine{0.0, -1};
#pragma unroll
              for (int s = 0; s < 4; ++s)
                if (cand[s]) mine = best_of(mine, Best{cur[q][s], k0 + s});
              if (__ballot(mine.p >= 0) == 0ull) break;
              const Best w = wave_best(mine);
              raise(w.v, w.p);
#pragma unroll
              for (int s = 0; s < 4; ++s)
                if (k0 + s == w.p) cand[s] = false;
              record(c, w.v, w.p);
              ++cnt;
            }
          }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) th[s] = th[s] * a_dec;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int s = 0; s < 4; ++s) cur[q][s] = nxt[q][s];
  }
  fs[T] = (short)ne;
  fs[T + 1] = (short)ne;
  __syncthreads();

  // ---- backward pass (peak_extractor.py:206-234)
  {
    double v[4];
    if (FUSED) {
#pragma unroll
      for (int s = 0; s < 4; ++s) v[s] = lastcol[s];
    } else {
      load_col(T - 1, v);
    }
    spread_init(v);
  }
  // entries of a frame in lanes 0..7: (value, frame << 8 | bin, entry index); -1 bin = none / pruned
  auto fetch = [&](int start, int n, double& fv, int& fp) {
    fv = 0.0; fp = -1;
    if (lane < n) { fv = ev[start + lane]; fp = ep[start + lane]; }
  };
  auto ufs = [&](int i) { return __builtin_amdgcn_readfirstlane((int)fs[i]); };      // wave-uniform table cell
  int st_c = ufs(T - 1), st_c1 = ufs(T);                       // frame T-1: entries [st_c, st_c1)
  double cv; int cp;
  fetch(st_c, st_c1 - st_c, cv, cp);
  int st_n = T >= 2 ? ufs(T - 2) : 0;                          // frame T-2 starts here (its end is st_c)
  int prev_p = -1, prev_idx = 0;                               // frame c+1's entries in lanes 0..7 (bin, entry index), -1 = none
  for (int c = T - 1; c >= 0; --c) {
    const int n = st_c1 - st_c;
    // requests for the next iterations go out first: entries of frame c-1, the table cell of frame c-2
    double nv = 0.0; int np = -1;
    if (c >= 1) fetch(st_n, st_c - st_n, nv, np);
    const short st_nn_raw = c >= 2 ? fs[c - 2] : (short)0;     // consumed (made uniform) at the end of the iteration
    int my_p = (lane < n) ? (cp & 255) : -1;                   // this frame's bins (lane i = rank i), -1 once pruned
    for (int i = 0; i < n; ++i) {
      const double val = readlane_f64(cv, i);
      const int p = __builtin_amdgcn_readlane(cp, i) & 255;    // wave-uniform
      const int s_sel = p & 3;
      const double mine = s_sel == 0 ? th[0] : (s_sel == 1 ? th[1] : (s_sel == 2 ? th[2] : th[3]));
      const double thp = readlane_f64(mine, p >> 2);
      if (val >= thp) {
        raise(val, p);
        if (prev_p == p) {                                     // delete any following peak in the same bin (frame c+1)
          ep[prev_idx] |= (int)0x80000000;
          prev_p = -1;
        }
      } else if (lane == i) {
        ep[st_c + i] = cp | (int)0x80000000;
        my_p = -1;
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) th[s] = a_dec * th[s];
    prev_p = my_p; prev_idx = st_c + lane;
    st_c1 = st_c; st_c = st_n; st_n = __builtin_amdgcn_readfirstlane((int)st_nn_raw);
    cv = nv; cp = np;
  }
  __syncthreads();

  // ---- emit: mask (R, T) uint8 was zeroed by the host-side memset on the same stream (FUSED: by this wave, above: fence first)
  if (FUSED) __threadfence();
  uint8_t* M = mask + (size_t)b * R * T;
  int count = 0;
  for (int e = lane; e < ne; e += 64) {
    const int pe = ep[e];
    if (pe >= 0) {
      M[(size_t)(pe & 255) * T + (pe >> 8)] = 1;
      ++count;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) count += __shfl_xor(count, o);
  if (lane == 0) npeaks[b] = count;
}

}  // namespace

// test hook for prep_div_fast (csrc/mfpa_prepsum.h): out[i] = v[i] / den[i] by the pick kernels' reciprocal + two-correction sequence
__global__ __launch_bounds__(256) void div_fast_kernel(const double* __restrict__ v, const double* __restrict__ den, long long n,
                                                       double* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const double d = den[i];
    out[i] = mfpa_prepsum::prep_div_fast(v[i], d, 1.0 / d);
  }
}

extern "C" {

int mfpa_div_by_reciprocal(const double* v, const double* den, long long n, double* out, void* stream) {
  if (n == 0) return MFPA_OK;
  if (!v || !den || !out || n < 0) return MFPA_EINVAL;
  const long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(div_fast_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, mfpa_stream(stream), v, den, n, out);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}


int mfpa_audfprint_prepare(const void* spec, int dtype, int B, int F, int T, const double* denom, int mean_order,
                           int log_input, double pole, double* filtered, double* scratch, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!spec || !filtered || !scratch || B < 0 || F < 2 || T < 1) return MFPA_EINVAL;
  if (dtype != MFPA_F32 && dtype != MFPA_F64) return MFPA_EINVAL;
  if (F - 1 > 256) return MFPA_EINVAL;
  const long long N = (long long)F * T;
  if ((N + NPY_BUFSIZE - 1) / NPY_BUFSIZE > MAX_CHUNKS) return MFPA_EINVAL;
  const int R = F - 1;
  const int nchunks = (int)((N + NPY_BUFSIZE - 1) / NPY_BUFSIZE);
  const size_t lds = sizeof(double) * ((size_t)R * (TT + 1) + (size_t)TT * R) + sizeof(double) * (size_t)nchunks * HEAP;
  hipStream_t s = mfpa_stream(stream);
  if (dtype == MFPA_F64)
    hipLaunchKernelGGL(prepare_kernel<double>, dim3(B), dim3(PREP_THREADS), lds, s, (const double*)spec, F, T, denom,
                       mean_order, log_input, pole, filtered, scratch);
  else
    hipLaunchKernelGGL(prepare_kernel<float>, dim3(B), dim3(PREP_THREADS), lds, s, (const float*)spec, F, T, denom,
                       mean_order, log_input, pole, filtered, scratch);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_audfprint_prune(const double* filtered, int B, int R, int T, const double* gauss, double a_dec, int maxpks,
                         uint8_t* mask, int32_t* npeaks, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!filtered || !gauss || !mask || !npeaks || B < 0) return MFPA_EINVAL;
  if (R < 4 || R > 256 || (R % 4) != 0 || T < 1 || T > 1500 || maxpks < 1 || maxpks > MAXP) return MFPA_EINVAL;
  hipStream_t s = mfpa_stream(stream);
  MFPA_HIP(hipMemsetAsync(mask, 0, (size_t)B * R * T, s));
  const size_t lds = sizeof(double) * (2 * R + 2) + (size_t)T * maxpks * (sizeof(double) + sizeof(int)) + sizeof(short) * (T + 2);
  hipLaunchKernelGGL(prune_kernel<false>, dim3(B), dim3(64), lds, s, filtered, R, T, gauss, a_dec, maxpks, mask, npeaks, (const double*)nullptr,
                     (const double*)nullptr, 0.0);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_audfprint_pick(const double* spec, const double* clip_max, int B, int F, int T, double pole, const double* gauss, double a_dec,
                        int maxpks, double* work, uint8_t* mask, int32_t* npeaks, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!spec || !clip_max || !gauss || !work || !mask || !npeaks || B < 0) return MFPA_EINVAL;
  const int R = F - 1;
  if (F < 141 || F > 257 || (R % 4) != 0 || T < 1 || T > SPLIT_MAX_T || maxpks < 1 || maxpks > MAXP) return MFPA_EINVAL;   // (a half-chunk node spans <= 32 frames)
  const long long N = (long long)F * T;
  const int nchunks = (int)((N + NPY_BUFSIZE - 1) / NPY_BUFSIZE);
  if (nchunks > MAX_CHUNKS) return MFPA_EINVAL;
  hipStream_t s = mfpa_stream(stream);
  if (((size_t)R * T) % 16 != 0 || (reinterpret_cast<uintptr_t>(mask) & 15) != 0) return MFPA_EINVAL;   // the pruner zeroes the mask in 16-byte pieces
  double* sums = work + (size_t)B * N;
  const size_t lds1 = sizeof(double) * (NPY_BUFSIZE / 2 + 8 + HEAP + 128 * 3);
  hipLaunchKernelGGL(prep_sum_kernel, dim3(B, 2 * nchunks), dim3(SPLIT_THREADS), lds1, s, spec, F, T, clip_max, 1, work, 1, sums, (long long)(2 * MAX_CHUNKS),
                     1.0, F - 1);
  MFPA_CHECK_LAUNCH();
  const size_t lds = sizeof(double) * (2 * R + 2) + (size_t)T * maxpks * (sizeof(double) + sizeof(int)) + sizeof(short) * (T + 2);
  hipLaunchKernelGGL(prune_kernel<true>, dim3(B), dim3(64), lds, s, (const double*)work, R, T, gauss, a_dec, maxpks, mask, npeaks, (const double*)sums, clip_max, pole);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

}  // extern "C"
