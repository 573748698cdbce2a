// Demucs training step, backward-pass kernels for MI355X (gfx950) -- SURVEY.md §8f-2 ("+ MRSTFT loss train step").
// Reference: training/train.py:275-312 (input_type == "audio": L1 + MultiResolutionSTFTLoss, loss.backward(), Adam) over
// training/model.py:163-326.  torch autograd is not used: every layer's adjoint is written out.
//
// Activations are time-major (B, L, C) like the forward (csrc/demucs.hip).  Input gradients of every Conv1d /
// ConvTranspose1d / 1x1 / LSTM projection are again strided-window GEMMs served by mfpa_gemm_mfma (a Conv1d's input
// gradient is a ConvTranspose1d of the output gradient and vice versa); its epilogue applies the ReLU mask of the layer
// below (mode 3).  This file holds what that kernel cannot express:
//   gemm_tn_kernel        weight gradients  dW[m][n] += sum_rows dY[row][m] * Xwin[row][n]  (K = every time step of the batch)
//   glu_bwd_kernel        GLU backward on the packed [32 values | 32 gates] pre-activation tiles the forward saved
//   lstm_step_bwd_kernel  one backward time step: dh_rec = dgates[t+1] W_hh, then the cell backward, in one launch
//   downsample2 adjoint, the two 1-channel convolutions' weight gradients, column sums (bias gradients)
#include "mfpa_common.h"
#include <cstdlib>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 t_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 t_bf16x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------- weight-gradient GEMM ("TN")
// C[m][n] += sum_{b < batch} sum_{r < R} A[b*strideA + r*lda + m] * Bm[b*strideB + r*ldb + n]
// Both operands are K-major in memory (a row = one time step), which is exactly the fragment order of
// v_mfma_f32_32x32x2_f32: lane (i, k) of the A operand holds A[k][i], so the tiles are staged as they lie in HBM
// ([k][m] rows, 32-float pad so the two k rows a wave reads fall on disjoint banks) and read with ds_read_b32.
// K is split over blockIdx.y (row ranges that never straddle a clip); partial tiles are added with float atomics.
struct TnArgs {
  const float* A; long long lda, strideA;
  const float* Bm; long long ldb, strideB;
  float* C; long long ldc;
  float* colsum;               // optional: colsum[m] += sum over all rows of A[.][m] (the bias gradient), by the n-tile-0 workgroups
  int R, M, N;
  int rs, spb;                 // rows per split, splits per clip
  int tiles, nsplit, xcd;      // 1-D launch: tiles x nsplit workgroups; xcd = 1: XCD-aware order
};

// (tile, split) of a workgroup.  Every tile of one K split reads the same rows of A and Bm; consecutive workgroup ids go round-robin
// over the 8 XCDs (one L2 each), so in the plain order each XCD fetched every split's rows.  Here XCD k owns a contiguous range
// of the (split, tile) order, tiles fastest: the tiles of a split run back to back on ONE XCD.
__device__ __forceinline__ bool tn_tile(const TnArgs& a, int& tile, int& split) {
  const unsigned total = (unsigned)a.tiles * a.nsplit;
  unsigned lin = blockIdx.x;
  if (a.xcd) {
    const unsigned per = (total + 7) / 8;
    lin = (blockIdx.x % 8) * per + blockIdx.x / 8;
  }
  if (lin >= total) return false;
  tile = lin % a.tiles;
  split = lin / a.tiles;
  return true;
}

// column sums of the A tiles a workgroup staged (every thread keeps the same column quad for all of its loads): combine the row
// lanes through LDS, one atomic per column
template <int QA>
__device__ __forceinline__ void tn_colsum_flush(float* red, f32x4 csum, int tid, int m0, int M, float* __restrict__ out) {
  __syncthreads();                                   // the operand buffers are free
  const int q = tid % QA, rl = tid / QA;
  *reinterpret_cast<f32x4*>(red + (rl * QA + q) * 4) = csum;
  __syncthreads();
  if (tid < QA * 4) {
    const int qq = tid >> 2, k = tid & 3;
    float sacc = 0.f;
#pragma unroll
    for (int r = 0; r < 256 / QA; ++r) sacc += red[(r * QA + qq) * 4 + k];
    const int m = m0 + 4 * qq + k;
    if (m < M) unsafeAtomicAdd(out + m, sacc);
  }
}

constexpr int TKC = 16;

template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_tn_kernel(TnArgs a) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int LDA_ = BM + 32, LDB_ = BN + 32;
  constexpr int QA = BM / 4, QB = BN / 4;                // float4 per staged row
  constexpr int FA = TKC * QA / 256, FB = TKC * QB / 256;  // float4 per thread per chunk (TM, TN)
  __shared__ __attribute__((aligned(16))) float As[2][TKC * LDA_];
  __shared__ __attribute__((aligned(16))) float Bs[2][TKC * LDB_];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  const int ntn = (a.N + BN - 1) / BN;
  int tile, split;
  if (!tn_tile(a, tile, split)) return;              // uniform, before any barrier
  const int n0 = (tile % ntn) * BN, m0 = (tile / ntn) * BM;
  const int b = split / a.spb;
  const int rbeg = (split % a.spb) * a.rs;
  const int rend = rbeg + a.rs < a.R ? rbeg + a.rs : a.R;
  const float* Ab = a.A + (size_t)b * a.strideA;
  const float* Bb = a.Bm + (size_t)b * a.strideB;
  const int nk = (rend - rbeg + TKC - 1) / TKC;

  f32x4 ar[FA], br[FB];
  const bool do_colsum = a.colsum != nullptr && n0 == 0;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  auto load = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < FA; ++i) {
      const int idx = tid + 256 * i, row = idx / QA, q = idx % QA;
      const int r = rbeg + kc * TKC + row, m = m0 + 4 * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < rend && m < a.M) v = *reinterpret_cast<const f32x4*>(Ab + (size_t)r * a.lda + m);
      ar[i] = v;
    }
#pragma unroll
    for (int i = 0; i < FB; ++i) {
      const int idx = tid + 256 * i, row = idx / QB, q = idx % QB;
      const int r = rbeg + kc * TKC + row, n = n0 + 4 * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < rend && n < a.N) v = *reinterpret_cast<const f32x4*>(Bb + (size_t)r * a.ldb + n);
      br[i] = v;
    }
  };
  auto store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < FA; ++i) {
      const int idx = tid + 256 * i;
      *reinterpret_cast<f32x4*>(&As[buf][(idx / QA) * LDA_ + 4 * (idx % QA)]) = ar[i];
      if (do_colsum) csum += ar[i];
    }
#pragma unroll
    for (int i = 0; i < FB; ++i) {
      const int idx = tid + 256 * i;
      *reinterpret_cast<f32x4*>(&Bs[buf][(idx / QB) * LDB_ + 4 * (idx % QB)]) = br[i];
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nk > 0) {
    load(0);
    store(0);
    if (nk > 1) load(1);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
      const int buf = kc & 1;
      if (kc + 1 < nk) store(buf ^ 1);
      if (kc + 2 < nk) load(kc + 2);
      const float* Ap = &As[buf][lh * LDA_ + wm * 32 * TM + li];
      const float* Bp = &Bs[buf][lh * LDB_ + wn * 32 * TN + li];
#pragma unroll
      for (int s = 0; s < TKC / 2; ++s) {
        float af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = Ap[2 * s * LDA_ + 32 * i];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = Bp[2 * s * LDB_ + 32 * j];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
  }
  if (do_colsum) tn_colsum_flush<QA>(&As[0][0], csum, tid, m0, a.M, a.colsum);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * 32 * TN + 32 * j + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 * TM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < a.M && n < a.N) unsafeAtomicAdd(a.C + (size_t)m * a.ldc + n, acc[i][j][r]);
      }
    }
}

// The same product on the bf16 matrix cores (precision 1: bf16x3, precision 2: plain bf16).  K is the ROW index and rows are
// what is contiguous in HBM, so the 8-consecutive-k fragments of v_mfma_f32_32x32x16_bf16 come from gfx950's transposing LDS
// read ds_read_b64_tr_b16 (lane i of a 16-lane group receives column i of a 4-row x 16-column block): the tiles stay in their
// natural [k][m] order.  LDS row = [BM bf16 hi | BM bf16 lo (bf16x3 only) | 64 B pad], which puts the four rows of a block on
// bank offsets 0 / 64 / 128 / 192.  Plain bf16 is the default for training: a weight gradient sums over every time step of the
// batch, so the 2^-9 product rounding averages out, and only the optimiser consumes the result (as for the UNet).
typedef short t_s16x4 __attribute__((ext_vector_type(4)));
constexpr int HKC_T = 32;

__device__ __forceinline__ t_bf16x8 tn_tr_frag(const char* p0, const char* p1) {
  typedef t_s16x4 __attribute__((address_space(3))) * lds_ptr;
  const t_s16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0));
  const t_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p1));
  union { short s[8]; t_bf16x8 b; } r;
#pragma unroll
  for (int j = 0; j < 4; ++j) { r.s[j] = u[j]; r.s[4 + j] = v[j]; }
  return r.b;
}

template <int TM, int TN, bool PLAIN>
__global__ __launch_bounds__(256) void gemm_tn_bf16_kernel(TnArgs a) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int ROWA = (PLAIN ? 2 : 4) * BM + 64, ROWB = (PLAIN ? 2 : 4) * BN + 64;     // bytes
  constexpr int QA = BM / 4, QB = BN / 4;
  constexpr int FA = HKC_T * QA / 256, FB = HKC_T * QB / 256;
  extern __shared__ __attribute__((aligned(16))) char tsm[];
  char* As = tsm;                                  // [2][HKC_T][ROWA]
  char* Bs = tsm + 2 * HKC_T * ROWA;               // [2][HKC_T][ROWB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  const int ntn = (a.N + BN - 1) / BN;
  int tile, split;
  if (!tn_tile(a, tile, split)) return;              // uniform, before any barrier
  const int n0 = (tile % ntn) * BN, m0 = (tile / ntn) * BM;
  const int b = split / a.spb;
  const int rbeg = (split % a.spb) * a.rs;
  const int rend = rbeg + a.rs < a.R ? rbeg + a.rs : a.R;
  const float* Ab = a.A + (size_t)b * a.strideA;
  const float* Bb = a.Bm + (size_t)b * a.strideB;
  const int nk = (rend - rbeg + HKC_T - 1) / HKC_T;
  // transposing read: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of its block
  const int gl = lane & 15, tq = gl >> 2, tp = gl & 3, gsel = (lane >> 4) & 1;
  const int a_off = (8 * lh + tq) * ROWA + (wm * 32 * TM + 16 * gsel + 4 * tp) * 2;
  const int b_off = (8 * lh + tq) * ROWB + (wn * 32 * TN + 16 * gsel + 4 * tp) * 2;

  f32x4 ar[FA], br[FB];
  const bool do_colsum = a.colsum != nullptr && n0 == 0;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  auto load = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < FA; ++i) {
      const int idx = tid + 256 * i, row = idx / QA, q = idx % QA;
      const int r = rbeg + kc * HKC_T + row, m = m0 + 4 * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < rend && m < a.M) v = *reinterpret_cast<const f32x4*>(Ab + (size_t)r * a.lda + m);
      ar[i] = v;
    }
#pragma unroll
    for (int i = 0; i < FB; ++i) {
      const int idx = tid + 256 * i, row = idx / QB, q = idx % QB;
      const int r = rbeg + kc * HKC_T + row, n = n0 + 4 * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < rend && n < a.N) v = *reinterpret_cast<const f32x4*>(Bb + (size_t)r * a.ldb + n);
      br[i] = v;
    }
  };
  auto split_store = [&](char* row, int q, int lo_off, f32x4 v) __attribute__((always_inline)) {
    t_bf16x4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hi[k] = (__bf16)v[k];
      lo[k] = (__bf16)(v[k] - (float)hi[k]);
    }
    *reinterpret_cast<t_bf16x4*>(row + 8 * q) = hi;
    if (!PLAIN) *reinterpret_cast<t_bf16x4*>(row + lo_off + 8 * q) = lo;
  };
  auto store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < FA; ++i) {
      const int idx = tid + 256 * i;
      split_store(As + (buf * HKC_T + idx / QA) * ROWA, idx % QA, 2 * BM, ar[i]);
      if (do_colsum) csum += ar[i];
    }
#pragma unroll
    for (int i = 0; i < FB; ++i) {
      const int idx = tid + 256 * i;
      split_store(Bs + (buf * HKC_T + idx / QB) * ROWB, idx % QB, 2 * BN, br[i]);
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nk > 0) {
    load(0);
    store(0);
    if (nk > 1) load(1);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
      const int buf = kc & 1;
      if (kc + 1 < nk) store(buf ^ 1);
      if (kc + 2 < nk) load(kc + 2);
      const char* Ap = As + buf * HKC_T * ROWA + a_off;
      const char* Bp = Bs + buf * HKC_T * ROWB + b_off;
#pragma unroll
      for (int ks = 0; ks < HKC_T / 16; ++ks) {
        t_bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const char* p = Ap + 16 * ks * ROWA + 64 * i;
          ah[i] = tn_tr_frag(p, p + 4 * ROWA);
          if (!PLAIN) al[i] = tn_tr_frag(p + 2 * BM, p + 4 * ROWA + 2 * BM);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const char* p = Bp + 16 * ks * ROWB + 64 * j;
          bh[j] = tn_tr_frag(p, p + 4 * ROWB);
          if (!PLAIN) bl[j] = tn_tr_frag(p + 2 * BN, p + 4 * ROWB + 2 * BN);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if (!PLAIN) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
      }
      __syncthreads();
    }
  }
  if (do_colsum) tn_colsum_flush<QA>(reinterpret_cast<float*>(tsm), csum, tid, m0, a.M, a.colsum);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * 32 * TN + 32 * j + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 * TM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < a.M && n < a.N) unsafeAtomicAdd(a.C + (size_t)m * a.ldc + n, acc[i][j][r]);
      }
    }
}

// ---------------------------------------------------------------------------------- GLU backward
// u (rows, npad): packed pre-activations, tile t = [32 values | 32 gates] of output columns 32 t .. 32 t + 31 (< N).
// In place: u <- d(loss)/du given dg (rows, N columns, row pitch ldg).
__global__ __launch_bounds__(256) void glu_bwd_kernel(float* __restrict__ u, long long rows, int npad, int N,
                                                      const float* __restrict__ dg, long long ldg) {
  const int tiles = npad / 64;
  const long long total = rows * tiles * 8;              // one thread = 4 consecutive outputs of a tile
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int q = (int)(e & 7);
    const long long rt = e >> 3;
    const int t = (int)(rt % tiles);
    const long long row = rt / tiles;
    float* up = u + row * npad + t * 64 + 4 * q;
    const f32x4 v = *reinterpret_cast<const f32x4*>(up), s = *reinterpret_cast<const f32x4*>(up + 32);
    const int n = t * 32 + 4 * q;
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    if (n < N) d = *reinterpret_cast<const f32x4*>(dg + row * ldg + n);   // N is a multiple of 4
    f32x4 dv, ds;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float sg = 1.f / (1.f + __expf(-s[k]));
      dv[k] = d[k] * sg;
      ds[k] = d[k] * v[k] * sg * (1.f - sg);
    }
    *reinterpret_cast<f32x4*>(up) = dv;
    *reinterpret_cast<f32x4*>(up + 32) = ds;
  }
}

// ---------------------------------------------------------------------------------- column sums (bias gradients)
// out[c] += sum_r x[r*ld + c], any C that is a multiple of 4 (48, 96, ... 3072).  A thread keeps float4 register
// accumulators for its column quad(s) and walks the rows of its block's range (four independent loads in flight); the
// row lanes are combined once per block through LDS, then one global atomic per column and block.
template <int NQ>
__global__ __launch_bounds__(256) void colsum_any_kernel(const float* __restrict__ x, long long rows, int C, long long ld,
                                                         float* __restrict__ out, long long rows_per_block) {
  extern __shared__ float accs[];
  const int tid = threadIdx.x;
  for (int c = tid; c < C; c += 256) accs[c] = 0.f;
  __syncthreads();
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  const int Q = C / 4;
  const int QT = Q < 256 ? Q : 256;
  const int lanes = Q <= 256 ? 256 / Q : 1;
  const int q0 = tid % QT, rl = tid / QT;
  f32x4 acc[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (rl < lanes) {
    long long r = r0 + rl;
    for (; r + 3LL * lanes < r1; r += 4LL * lanes) {
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int q = q0 + 256 * i;
        if (q < Q) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + r * ld + 4 * q);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(x + (r + lanes) * ld + 4 * q);
          const f32x4 v2 = *reinterpret_cast<const f32x4*>(x + (r + 2LL * lanes) * ld + 4 * q);
          const f32x4 v3 = *reinterpret_cast<const f32x4*>(x + (r + 3LL * lanes) * ld + 4 * q);
          acc[i] += (v0 + v1) + (v2 + v3);
        }
      }
    }
    for (; r < r1; r += lanes) {
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int q = q0 + 256 * i;
        if (q < Q) acc[i] += *reinterpret_cast<const f32x4*>(x + r * ld + 4 * q);
      }
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int q = q0 + 256 * i;
      if (q < Q) {
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(&accs[4 * q + k], acc[i][k]);
      }
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) unsafeAtomicAdd(out + c, accs[c]);
}

// ---------------------------------------------------------------------------------- 1-channel convolutions' weight gradients
// dw[j][c] += sum_{b,t} x[b*ldx + 4t + j] * g[b*strideG + t*ldg + c],  j < 8, t < L   (w (8, C) tap-major)
// serves encoder.0.0 (x = the upsampled input, g = the masked gradient of its ReLU output) and the last
// ConvTranspose1d (x = the gradient of its output, g = the GLU output it consumed).
__global__ __launch_bounds__(256) void c1_wgrad_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ g,
                                                       long long ldg, long long strideG, int L, int C, float* __restrict__ dw,
                                                       int rows_per_block) {
  __shared__ float accs[8 * 256];                        // C <= 256
  const int tid = threadIdx.x, b = blockIdx.y;
  for (int i = tid; i < 8 * C; i += 256) accs[i] = 0.f;
  __syncthreads();
  const int Q = C / 4, lanes = 256 / Q;                  // row lanes per pass (Q = 12 for C = 48 -> 21 row lanes)
  const int q = tid % Q, rl = tid / Q;
  const int t0 = blockIdx.x * rows_per_block;
  const int t1 = t0 + rows_per_block < L ? t0 + rows_per_block : L;
  mfpa_f32x2 acc[8][2];                                  // [tap][channel pair]: packed FMAs without operand selection (mfpa_common.h)
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j][0] = acc[j][1] = mfpa_f32x2{0.f, 0.f};
  if (rl < lanes) {
    const float* xb = x + (size_t)b * ldx;
    const float* gb = g + (size_t)b * strideG;
    for (int t = t0 + rl; t < t1; t += lanes) {
      const f32x4 gv = *reinterpret_cast<const f32x4*>(gb + (size_t)t * ldg + 4 * q);
      const mfpa_f32x2 g01 = {gv[0], gv[1]}, g23 = {gv[2], gv[3]};
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(xb + 4 * (size_t)t), x1 = *reinterpret_cast<const f32x4*>(xb + 4 * (size_t)t + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const mfpa_f32x2 u = mfpa_bcast2(x0[j]), v = mfpa_bcast2(x1[j]);
        acc[j][0] += u * g01; acc[j][1] += u * g23;
        acc[4 + j][0] += v * g01; acc[4 + j][1] += v * g23;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(&accs[j * C + 4 * q + k], acc[j][k >> 1][k & 1]);
  }
  __syncthreads();
  for (int i = tid; i < 8 * C; i += 256) unsafeAtomicAdd(dw + i, accs[i]);
}

// ---------------------------------------------------------------------------------- downsample2 adjoint
// forward (model.py:69-88): out[i] = sc * 0.5 * (x[2i] + sum_k xodd[i + k - 56] ker[k]), i < nout, xodd[j] = x[2j+1].
// adjoint: dx[2i] = g[i], dx[2j+1] = sum_k g[j + 56 - k] ker[k] with g[i] = 0.5 * sc * dy[i] (0 for i >= nout).
__global__ __launch_bounds__(256) void downsample2_adj_kernel(const float* __restrict__ dy, int ldy, int nout, const float* __restrict__ ker,
                                                              const float* __restrict__ scale, int T, float* __restrict__ dx) {
  __shared__ float kk[112];
  const int b = blockIdx.y, tid = threadIdx.x;
  if (tid < 112) kk[tid] = ker[tid];
  __syncthreads();
  const float sc = 0.5f * (scale ? scale[b] : 1.f);
  const float* dyb = dy + (size_t)b * ldy;
  float* dxb = dx + (size_t)b * T;
  for (int p = blockIdx.x * 256 + tid; p < T; p += gridDim.x * 256) {
    float v;
    if ((p & 1) == 0) {
      const int i = p >> 1;
      v = i < nout ? dyb[i] : 0.f;
    } else {
      const int j = p >> 1;
      float s = 0.f;
      for (int k = 0; k < 112; ++k) {
        const int i = j + 56 - k;
        if (i >= 0 && i < nout) s += dyb[i] * kk[k];
      }
      v = s;
    }
    dxb[p] = sc * v;
  }
}

// ---------------------------------------------------------------------------------- LSTM backward time step
// One launch per step t (t = T-1 .. 0):
//   dh    = dhout[t] + dgates[t+1] W_hh                      (GEMM, K = 4H; skipped at t = T-1)
//   do = dh tanh(c_t) o (1-o);  dc = dc_next + dh o (1 - tanh^2 c_t);  di = dc g i (1-i);  df = dc c_{t-1} f (1-f);
//   dg = dc i (1-g^2);  dc_next <- dc f;   dgates[t] overwrites the saved gate activations [i | f | g | o] of step t.
// A workgroup owns 64 clips x 32 hidden units: B operand = rows u of W_hh^T (H, 4H).  8 waves = (clip half) x (K quarter of
// every 128-wide chunk); bf16x3 products like the forward step; workgroup id -> (XCD, slot) so that an XCD's three unit
// groups keep their 1.2 MB of W_hh^T in that XCD's L2 for all steps.
constexpr int BKC = 128;
constexpr int BROW = 4 * BKC + 16;       // LDS row bytes [128 hi | 128 lo | pad]
constexpr int BU = 32;
constexpr int BTHREADS = 512;
#ifndef MFPA_LSTM_BPF
#define MFPA_LSTM_BPF 6
#endif
constexpr int BPF = MFPA_LSTM_BPF;       // chunks of global loads in flight per thread

// MT: 32-clip tiles per workgroup: 2 = (clip half) x (K quarter), 1 = K eighths (small batches: twice the workgroups).
// BUT: hidden units per workgroup, 32 or 16 (16: the matrix tile is half empty, but the step is bound by the bytes a workgroup
// streams -- 32 x 3072 gate gradients + BUT x 3072 weights -- and twice as many CUs pull them).
template <int MT, int BUT>
__global__ __launch_bounds__(BTHREADS, 1) void lstm_step_bwd_kernel(const float* __restrict__ dgnext, long long ldgn,
                                                                    const float* __restrict__ whhT, float* gs, long long ldgs,
                                                                    const float* __restrict__ ct, long long ldct,
                                                                    const float* __restrict__ cprev, long long ldcp,
                                                                    const float* __restrict__ dhout, long long lddh,
                                                                    float* __restrict__ dcstate, int B, int H, int mtiles) {
  constexpr int BBM = 32 * MT;               // clips per workgroup
  constexpr int WK = 8 / MT;                 // k-step groups
  constexpr int KS = 8 / WK;                 // k-steps of 16 per wave and chunk
  constexpr int FA = BBM * 32 / BTHREADS;    // float4 per thread per chunk for the dgates rows (2 MT)
  extern __shared__ __attribute__((aligned(16))) char lsm[];
  char* As = lsm;                            // [2][BBM][BROW]
  char* Bs = lsm + 2 * BBM * BROW;           // [2][32][BROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave % MT, wk = wave / MT;  // wk: k-steps KS wk .. of each chunk
  const int ngroups = H / BUT;
  int grp, mt;
  {
    const int id = blockIdx.x, total = ngroups * mtiles;
    const int per_xcd = (total + 7) / 8;
    const int lin = (id % 8) * per_xcd + id / 8;
    if (lin >= total) return;                // uniform per workgroup, before any barrier
    grp = lin / mtiles; mt = lin % mtiles;
  }
  const int m0 = mt * BBM;
  const int K = 4 * H;
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  if (dgnext != nullptr) {
    const float* Wg = whhT + (size_t)grp * BUT * K;
    const int nk = K / BKC;
    constexpr int FB = BUT / 16;                       // float4 per thread per chunk for the weight rows
    if (BUT < 32) {                                    // rows BUT .. 31 of both weight buffers are never staged: keep them zero
      for (int i = tid; i < 2 * (32 - BUT) * (BROW / 16); i += BTHREADS) {
        const int buf = i / ((32 - BUT) * (BROW / 16)), rem = i % ((32 - BUT) * (BROW / 16));
        *reinterpret_cast<f32x4*>(Bs + (buf * 32 + BUT + rem / (BROW / 16)) * BROW + 16 * (rem % (BROW / 16))) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    // register ring of BPF chunks of global loads (the step is latency-bound: dgates[t+1] was written by the previous launch)
    f32x4 ar[BPF][FA], br[BPF][FB];
    const int q = tid & 31, r0 = tid >> 5;             // column quad, first row; rows r0 + 16 i
    auto load = [&](int kc, f32x4 (&a4)[FA], f32x4 (&b2)[FB]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < FA; ++i) {
        const int m = m0 + r0 + 16 * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < B) v = *reinterpret_cast<const f32x4*>(dgnext + (size_t)m * ldgn + kc * BKC + 4 * q);
        a4[i] = v;
      }
#pragma unroll
      for (int i = 0; i < FB; ++i) b2[i] = *reinterpret_cast<const f32x4*>(Wg + (size_t)(r0 + 16 * i) * K + kc * BKC + 4 * q);
    };
    auto split_store = [&](char* row, f32x4 v) __attribute__((always_inline)) {
      t_bf16x4 hi, lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        hi[k] = (__bf16)v[k];
        lo[k] = (__bf16)(v[k] - (float)hi[k]);
      }
      *reinterpret_cast<t_bf16x4*>(row + 8 * q) = hi;
      *reinterpret_cast<t_bf16x4*>(row + 2 * BKC + 8 * q) = lo;
    };
    auto store = [&](int buf, f32x4 (&a4)[FA], f32x4 (&b2)[FB]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < FA; ++i) split_store(As + (buf * BBM + r0 + 16 * i) * BROW, a4[i]);
#pragma unroll
      for (int i = 0; i < FB; ++i) split_store(Bs + (buf * 32 + r0 + 16 * i) * BROW, b2[i]);
    };
#pragma unroll
    for (int j = 0; j < BPF; ++j)
      if (j < nk) load(j, ar[j], br[j]);
    for (int base = 0; base < nk; base += BPF) {
#pragma unroll
      for (int j = 0; j < BPF; ++j) {
        const int kc = base + j;
        if (kc < nk) {                                   // uniform over the workgroup
          const int buf = kc & 1;
          store(buf, ar[j], br[j]);                      // buffer (kc & 1) was last read for chunk kc - 2, before the previous barrier
          if (kc + BPF < nk) load(kc + BPF, ar[j], br[j]);
          __syncthreads();
          const char* Ap = As + (buf * BBM + wm * 32 + li) * BROW + 16 * lh;
          const char* Bp = Bs + (buf * 32 + li) * BROW + 16 * lh;
#pragma unroll
          for (int s = KS * wk; s < KS * wk + KS; ++s) {
            const t_bf16x8 ah = *reinterpret_cast<const t_bf16x8*>(Ap + 32 * s);
            const t_bf16x8 al = *reinterpret_cast<const t_bf16x8*>(Ap + 2 * BKC + 32 * s);
            const t_bf16x8 bh = *reinterpret_cast<const t_bf16x8*>(Bp + 32 * s);
            const t_bf16x8 bl = *reinterpret_cast<const t_bf16x8*>(Bp + 2 * BKC + 32 * s);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();                                     // the slabs below reuse the operand buffers
  }
  // the WK partial tiles -> LDS slabs [WK][BBM clips][36], summed by the cell threads
  float* G = reinterpret_cast<float*>(lsm);
  constexpr int GLDW = 36;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    G[(wk * BBM + m) * GLDW + li] = acc[r];
  }
  __syncthreads();
  constexpr int UQ = BUT / 4;                  // unit quads per clip
  const int clip = tid / UQ, uq = tid % UQ;
  const int m = m0 + clip;
  if (clip < BBM && m < B) {
    const int u0 = grp * BUT + 4 * uq;
    f32x4 dh = *reinterpret_cast<const f32x4*>(dhout + (size_t)m * lddh + u0);
#pragma unroll
    for (int w = 0; w < WK; ++w) dh += *reinterpret_cast<const f32x4*>(G + (w * BBM + clip) * GLDW + 4 * uq);
    float* gr = gs + (size_t)m * ldgs;
    const f32x4 vi = *reinterpret_cast<const f32x4*>(gr + u0), vf = *reinterpret_cast<const f32x4*>(gr + H + u0);
    const f32x4 vg = *reinterpret_cast<const f32x4*>(gr + 2 * H + u0), vo = *reinterpret_cast<const f32x4*>(gr + 3 * H + u0);
    const f32x4 c = *reinterpret_cast<const f32x4*>(ct + (size_t)m * ldct + u0);
    const f32x4 cp = cprev ? *reinterpret_cast<const f32x4*>(cprev + (size_t)m * ldcp + u0) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 dcs = *reinterpret_cast<const f32x4*>(dcstate + (size_t)m * H + u0);
    f32x4 di, df, dg, dO;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float tc = tanhf(c[k]);
      dO[k] = dh[k] * tc * vo[k] * (1.f - vo[k]);
      const float dc = dcs[k] + dh[k] * vo[k] * (1.f - tc * tc);
      di[k] = dc * vg[k] * vi[k] * (1.f - vi[k]);
      df[k] = dc * cp[k] * vf[k] * (1.f - vf[k]);
      dg[k] = dc * vi[k] * (1.f - vg[k] * vg[k]);
      dcs[k] = dc * vf[k];
    }
    *reinterpret_cast<f32x4*>(gr + u0) = di;
    *reinterpret_cast<f32x4*>(gr + H + u0) = df;
    *reinterpret_cast<f32x4*>(gr + 2 * H + u0) = dg;
    *reinterpret_cast<f32x4*>(gr + 3 * H + u0) = dO;
    *reinterpret_cast<f32x4*>(dcstate + (size_t)m * H + u0) = dcs;
  }
}


// ---------------------------------------------------------------------------------- persistent LSTM backward layer
// The backward recurrence of a layer for steps t1-1 .. t0 in ONE launch: lstm_seq_kernel's scheme (demucs.hip) turned round.
//   dh[t] = dhout[t] + dgates[t+1] W_hh  is K = 4H long and 16 hidden units wide per workgroup, so the weights only fit the
//   registers with the 16 x 16 x 32 MFMA: wave w of 8 holds the B-fragments of W_hh^T rows u0 .. u0+15, K range [w 4H/8, ..):
//   KS = H / 64 k-steps x (hi, lo) x 4 VGPRs (96 at H = 768), read once per launch instead of once per step;
//   dgates[t+1] (64 clips x 4H, already split [32 hi | 32 lo] by the cells that produced it) is exchanged through a ping-pong
//   buffer with agent-scope stores / buffer loads, A-fragments read straight from it; the 8 partial 64 x 16 tiles meet in LDS;
//   a cell thread owns (clip, 2 units): dc lives in its registers, the saved gate activations / cell states / dhout of step t are
//   fetched before the wait; it writes dgates[t] over the activations (the weight-gradient GEMMs read them later) and into the
//   exchange buffer.  Slab counter, bounded waits and the error word as in lstm_seq_kernel.
constexpr int QB_W = 8;
constexpr int QB_GLD = 17;
constexpr unsigned QB_SPIN_LIMIT = 1u << 22;
constexpr int QB_SYNC_WORDS = 1024, QB_ERR_WORD = 512;

struct LstmBwdSeqArgs {
  const float* whhT;      // (H, 4H)
  float* gates;           // (B, Tn, 4H): activations [i | f | g | o] in, pre-activation gradients out
  const float* cseq;      // (B, Tn, H)
  const float* dhout;     // (B, Tn, H)
  float* dcstate;         // (B, H): read when t1 < Tn, written at the end
  unsigned* sync;
  char* gsplit;           // [2][B][4H * 4 bytes]
  int B, Tn, H, t0, t1, nslab, ngroups;
};

// MTB = 16-clip MFMA row tiles per workgroup (slab = 16 MTB clips).  A workgroup reads its slab's WHOLE dgates[t+1] row block every
// step (K = 4H: 12 KB per clip), four times the forward's bytes, and a CU pulls ~60 GB/s of such loads: small batches therefore use
// small slabs, so that more CUs share the reading (64 clips: 192 workgroups of 16 clips instead of 48 of 64).
template <int KS, int MTB>          // k-steps of 32 per wave: 4H = 256 KS
__global__ __launch_bounds__(64 * QB_W, 1) void lstm_bwd_seq_kernel(LstmBwdSeqArgs a) {
  constexpr int SLAB = 16 * MTB;
  constexpr int UPT = MTB == 4 ? 2 : 1;                     // hidden units per cell thread
  constexpr int TPC = 16 / UPT;                             // cell threads per clip
  __shared__ __attribute__((aligned(16))) float G[QB_W * SLAB * QB_GLD];
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 15, kg = lane >> 4;
  const int H = a.H, K = 4 * a.H;
  int slab, grp;
  {
    const int id = blockIdx.x, total = a.nslab * a.ngroups;
    const int per_xcd = (total + 7) / 8;
    const int lin = (id % 8) * per_xcd + id / 8;
    if (lin >= total) return;                               // padding workgroups are not counted at the barrier
    slab = lin / a.ngroups; grp = lin % a.ngroups;
  }
  const int m0 = slab * SLAB, u0 = grp * 16;
  unsigned* cnt = a.sync + 16 * slab;
  unsigned* err = a.sync + QB_ERR_WORD;
  const unsigned members = (unsigned)a.ngroups;
  const size_t rowb = (size_t)K * 4, bufb = (size_t)a.B * rowb;

  // ---- W_hh^T fragments of this workgroup's 16 units, split once
  t_bf16x8 wh[KS], wl[KS];
  {
    const float* Wr = a.whhT + (size_t)(u0 + ln) * K + (size_t)wave * KS * 32 + 8 * kg;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(Wr + 32 * s), v1 = *reinterpret_cast<const f32x4*>(Wr + 32 * s + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const __bf16 h0 = (__bf16)v0[k], h1 = (__bf16)v1[k];
        wh[s][k] = h0; wh[s][4 + k] = h1;
        wl[s][k] = (__bf16)(v0[k] - (float)h0); wl[s][4 + k] = (__bf16)(v1[k] - (float)h1);
      }
    }
  }
  // ---- cell threads: clip tid / TPC, hidden units u .. u + UPT - 1
  const int clip = tid / TPC, ui = tid % TPC;
  const int m = m0 + clip;
  const bool live = clip < SLAB && m < a.B;
  const int u = u0 + UPT * ui;
  const size_t ldg = (size_t)a.Tn * K, ldh = (size_t)a.Tn * H;
  auto put_split = [&](char* buf, int q, const float (&v)[UPT]) __attribute__((always_inline)) {   // gate q of this thread's units -> the exchange rows
    const int col = q * H + u;
    char* p = buf + (size_t)(live ? m : 0) * rowb + (size_t)(col >> 5) * 128 + (size_t)(col & 31) * 2;
    if (UPT == 2) {
      const __bf16 h0 = (__bf16)v[0], h1 = (__bf16)v[UPT - 1];
      const __bf16 l0 = (__bf16)(v[0] - (float)h0), l1 = (__bf16)(v[UPT - 1] - (float)h1);
      const unsigned hi = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
      const unsigned lo = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
      __hip_atomic_store(reinterpret_cast<unsigned*>(p), hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(reinterpret_cast<unsigned*>(p + 64), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const __bf16 h0 = (__bf16)v[0];
      const __bf16 l0 = (__bf16)(v[0] - (float)h0);
      __hip_atomic_store(reinterpret_cast<unsigned short*>(p), __builtin_bit_cast(unsigned short, h0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(reinterpret_cast<unsigned short*>(p + 64), __builtin_bit_cast(unsigned short, l0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  float dc[UPT];
#pragma unroll
  for (int k = 0; k < UPT; ++k) dc[k] = 0.f;
  if (live) {
    char* buf = a.gsplit + (size_t)(a.t1 & 1) * bufb;                // dgates[t] live in buffer t & 1
    if (a.t1 < a.Tn) {                                               // a later range has run: its dgates[t1] and dc
      const float* gr = a.gates + (size_t)m * ldg + (size_t)a.t1 * K + u;
#pragma unroll
      for (int k = 0; k < UPT; ++k) dc[k] = a.dcstate[(size_t)m * H + u + k];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[UPT];
#pragma unroll
        for (int k = 0; k < UPT; ++k) v[k] = gr[q * H + k];
        put_split(buf, q, v);
      }
    } else {
      float z[UPT];
#pragma unroll
      for (int k = 0; k < UPT; ++k) z[k] = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) put_split(buf, q, z);
    }
  }
  bool dead = false;
  auto arrive = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto wait = [&](unsigned round) __attribute__((always_inline)) {
    if (tid == 0 && !dead) {
      const unsigned target = round * members;
      unsigned n = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if ((++n & 63u) == 0u && (n > QB_SPIN_LIMIT || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
          __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          dead = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
  };
  arrive();

  unsigned arow[MTB];
#pragma unroll
  for (int mt = 0; mt < MTB; ++mt) {
    int r = m0 + mt * 16 + ln;
    r = r < a.B ? r : a.B - 1;
    arow[mt] = (unsigned)((size_t)r * rowb + (size_t)wave * KS * 128 + 16 * kg);          // k-step s = chunk wave KS + s of the row
  }
  const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(a.gsplit, 0, (int)(2 * bufb), 0x00020000);
  constexpr int PF = (MTB == 4) ? 2 : (KS < 4 ? KS : 4);    // k-steps of A loads in flight (8 MTB VGPRs each)
  for (int t = a.t1 - 1; t >= a.t0; --t) {
    // what the cell needs of step t does not depend on the recurrence: fetch it before the wait
    float gv[4][UPT], ct[UPT], cp[UPT], dho[UPT];
#pragma unroll
    for (int k = 0; k < UPT; ++k) { gv[0][k] = gv[1][k] = gv[2][k] = gv[3][k] = 0.f; ct[k] = cp[k] = dho[k] = 0.f; }
    if (live) {
      const float* gr = a.gates + (size_t)m * ldg + (size_t)t * K + u;
      const size_t o = (size_t)m * ldh + (size_t)t * H + u;
#pragma unroll
      for (int k = 0; k < UPT; ++k) {
#pragma unroll
        for (int q = 0; q < 4; ++q) gv[q][k] = gr[q * H + k];
        ct[k] = a.cseq[o + k];
        if (t > 0) cp[k] = a.cseq[o + k - H];
        dho[k] = a.dhout[o + k];
      }
    }
    wait((unsigned)(a.t1 - t));
    const unsigned pb = (unsigned)(((t + 1) & 1) * bufb);
    f32x4 acc[MTB];
#pragma unroll
    for (int mt = 0; mt < MTB; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    t_bf16x8 fa[PF][MTB][2];
    auto issue = [&](int s, t_bf16x8 (&f)[MTB][2]) __attribute__((always_inline)) {
#pragma unroll
      for (int mt = 0; mt < MTB; ++mt) {
        const unsigned o = pb + arow[mt] + (unsigned)s * 128u;
        f[mt][0] = __builtin_bit_cast(t_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(grsrc, o, 0, 16));
        f[mt][1] = __builtin_bit_cast(t_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(grsrc, o + 64, 0, 16));
      }
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) issue(s, fa[s]);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int mt = 0; mt < MTB; ++mt) {
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[s % PF][mt][1], wh[s], acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[s % PF][mt][0], wl[s], acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[s % PF][mt][0], wh[s], acc[mt], 0, 0, 0);
      }
      if (s + PF < KS) issue(s + PF, fa[s % PF]);
    }
    // partial tiles -> LDS: D[row = 4 kg + j][col = ln] of m-tile mt
#pragma unroll
    for (int mt = 0; mt < MTB; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) G[(wave * SLAB + mt * 16 + 4 * kg + j) * QB_GLD + ln] = acc[mt][j];
    __syncthreads();
    if (live) {
      float dg_[4][UPT];
#pragma unroll
      for (int k = 0; k < UPT; ++k) {
        float dh = dho[k];
#pragma unroll
        for (int w = 0; w < QB_W; ++w) dh += G[(w * SLAB + clip) * QB_GLD + UPT * ui + k];
        const float vi = gv[0][k], vf = gv[1][k], vg = gv[2][k], vo = gv[3][k];
        const float tc = tanhf(ct[k]);
        dg_[3][k] = dh * tc * vo * (1.f - vo);
        const float dcv = dc[k] + dh * vo * (1.f - tc * tc);
        dg_[0][k] = dcv * vg * vi * (1.f - vi);
        dg_[1][k] = dcv * cp[k] * vf * (1.f - vf);
        dg_[2][k] = dcv * vi * (1.f - vg * vg);
        dc[k] = dcv * vf;
      }
      char* buf = a.gsplit + (size_t)(t & 1) * bufb;
      float* gr = a.gates + (size_t)m * ldg + (size_t)t * K + u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        put_split(buf, q, dg_[q]);
#pragma unroll
        for (int k = 0; k < UPT; ++k) gr[q * H + k] = dg_[q][k];
      }
    }
    if (t > a.t0) arrive();                                  // (its __syncthreads also frees the partial-tile slabs)
  }
  if (live) {
#pragma unroll
    for (int k = 0; k < UPT; ++k) a.dcstate[(size_t)m * H + u + k] = dc[k];
  }
}

static int lstm_bwd_seq_cus() { return mfpa_current_device_cus(); }

// slab size (mtb x 16 clips) and resident workgroups of the persistent backward launch; 0 = the per-step path
static int lstm_bwd_seq_plan(int B, int H, int wg_budget, int* mtb_out) {
  const int ks = (H % 64 == 0) ? H / 64 : 0, ngroups = H / 16;
  static const int force_mtb = MFPA_EXP_ENV("MFPA_LSTM_BWD_MTB", 0);
  const int cus = lstm_bwd_seq_cus();
  const int budget = (wg_budget > 0 && wg_budget < cus) ? wg_budget : cus;
  int mtb = 0;
  for (int c = 1; c <= 4 && !mtb; c *= 2)
    if ((long long)((B + 16 * c - 1) / (16 * c)) * ngroups <= budget) mtb = c;
  if (force_mtb == 1 || force_mtb == 2 || force_mtb == 4) mtb = force_mtb;
  const int nslab = mtb ? (B + 16 * mtb - 1) / (16 * mtb) : 0;
  if (mtb_out) *mtb_out = mtb;
  if (!mtb || !(ks == 4 || ks == 8 || ks == 12) || nslab > 32 || (long long)B * H * 32 > 0x7fffffffLL || (long long)nslab * ngroups > budget)
    return 0;
  return nslab * ngroups;
}

}  // namespace

extern "C" {

int mfpa_gemm_tn(const mfpa_gemm_tn_desc* d, void* stream) {
  if (!d) return MFPA_EINVAL;
  if (d->batch == 0 || d->R == 0) return MFPA_OK;
  if (!d->A || !d->Bm || !d->C || d->batch < 0 || d->R < 0 || d->M < 4 || d->N < 4 || d->M % 4 || d->N % 4) return MFPA_EINVAL;
  if (d->lda % 4 || d->ldb % 4 || d->strideA % 4 || d->strideB % 4 || d->ldc < d->N) return MFPA_EINVAL;   // float4 row loads
  if (d->precision < 0 || d->precision > 2) return MFPA_EINVAL;
  const bool big = d->M > 64 && d->N > 64;
  const int BM = big ? 128 : 64, BN = big ? 128 : 64;
  const long long tiles = (long long)((d->M + BM - 1) / BM) * ((d->N + BN - 1) / BN);
  // rows per K split: about 2048 workgroups in all, at least 64 rows each, never across clips
  long long want = 2048 / tiles; if (want < 1) want = 1;
  long long rs = ((long long)d->batch * d->R + want - 1) / want;
  if (rs < 64) rs = 64;
  rs = (rs + HKC_T - 1) / HKC_T * HKC_T;
  long long spb = (d->R + rs - 1) / rs;
  if (tiles > 0x7fffffffLL) return MFPA_EINVAL;
  TnArgs a{};
  a.A = d->A; a.lda = d->lda; a.strideA = d->strideA; a.Bm = d->Bm; a.ldb = d->ldb; a.strideB = d->strideB;
  a.C = d->C; a.ldc = d->ldc; a.colsum = d->colsum; a.R = d->R; a.M = d->M; a.N = d->N; a.rs = (int)rs; a.spb = (int)spb;
  static const int xcd_env = MFPA_EXP_ENV("MFPA_GEMM_XCD", 1);
  a.tiles = (int)tiles; a.nsplit = (int)(spb * d->batch); a.xcd = xcd_env;
  if (tiles * a.nsplit > 0x3fffffffLL) return MFPA_EINVAL;
  dim3 grid((unsigned)(((tiles * a.nsplit + 7) / 8) * 8));
  hipStream_t st = mfpa_stream(stream);
  if (d->precision == 0) {
    if (big) hipLaunchKernelGGL((gemm_tn_kernel<2, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm_tn_kernel<1, 1>), grid, dim3(256), 0, st, a);
  } else {
    const bool plain = d->precision == 2;
    const size_t lds = (size_t)2 * HKC_T * 2 * ((plain ? 2 : 4) * BM + 64);      // BM == BN
    if (big && plain) hipLaunchKernelGGL((gemm_tn_bf16_kernel<2, 2, true>), grid, dim3(256), lds, st, a);
    else if (big) hipLaunchKernelGGL((gemm_tn_bf16_kernel<2, 2, false>), grid, dim3(256), lds, st, a);
    else if (plain) hipLaunchKernelGGL((gemm_tn_bf16_kernel<1, 1, true>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((gemm_tn_bf16_kernel<1, 1, false>), grid, dim3(256), lds, st, a);
  }
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_glu_bwd(float* u, long long rows, int npad, int N, const float* dg, long long ldg, void* stream) {
  if (rows == 0) return MFPA_OK;
  if (!u || !dg || rows < 0 || npad < 64 || npad % 64 || N < 4 || N % 4 || N > npad / 2 || ldg < N || ldg % 4) return MFPA_EINVAL;
  const long long total = rows * (npad / 64) * 8;
  long long blocks = (total + 255) / 256; if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(glu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, mfpa_stream(stream), u, rows, npad, N, dg, ldg);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_colsum_any(const float* x, long long rows, int C, long long ld, float* out, void* stream) {
  if (rows == 0) return MFPA_OK;
  if (!x || !out || rows < 0 || C < 4 || C % 4 || C > 4096 || ld < C || ld % 4) return MFPA_EINVAL;
  long long blocks = (rows * (C / 4) + 256 * 32 - 1) / (256 * 32);      // about 32 float4 per thread
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  const long long rpb = (rows + blocks - 1) / blocks;
  blocks = (rows + rpb - 1) / rpb;
  if (C <= 1024)
    hipLaunchKernelGGL(colsum_any_kernel<1>, dim3((unsigned)blocks), dim3(256), (size_t)C * sizeof(float), mfpa_stream(stream), x, rows,
                       C, ld, out, rpb);
  else
    hipLaunchKernelGGL(colsum_any_kernel<4>, dim3((unsigned)blocks), dim3(256), (size_t)C * sizeof(float), mfpa_stream(stream), x, rows,
                       C, ld, out, rpb);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_c1_wgrad(const float* x, long long ldx, const float* g, long long ldg, long long strideG, int B, int L, int C, float* dw,
                  void* stream) {
  if (B == 0 || L == 0) return MFPA_OK;
  if (!x || !g || !dw || B < 0 || B > 65535 || L < 0 || C < 4 || C % 4 || C > 256 || ldx % 4 || ldg % 4 || strideG % 4 ||
      ldx < 4 * ((long long)L - 1) + 8) return MFPA_EINVAL;
  int gx = (L + 2047) / 2048; if (gx > 64) gx = 64;
  const int rpb = (L + gx - 1) / gx;
  gx = (L + rpb - 1) / rpb;
  hipLaunchKernelGGL(c1_wgrad_kernel, dim3(gx, B), dim3(256), 0, mfpa_stream(stream), x, ldx, g, ldg, strideG, L, C, dw, rpb);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_downsample2_adjoint(const float* dy, int B, int ldy, int nout, const float* kernel112, const float* scale, int T, float* dx,
                             void* stream) {
  if (B == 0) return MFPA_OK;
  if (!dy || !kernel112 || !dx || B < 0 || B > 65535 || T < 2 || nout < 1 || nout > (T + 1) / 2 || ldy < nout) return MFPA_EINVAL;
  int gx = (T + 255) / 256; if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(downsample2_adj_kernel, dim3(gx, B), dim3(256), 0, mfpa_stream(stream), dy, ldy, nout, kernel112, scale, T, dx);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_lstm_step_bwd(const float* dgnext, long long ldgn, const float* whhT, float* gates, long long ldg, const float* ct,
                       long long ldct, const float* cprev, long long ldcp, const float* dhout, long long lddh, float* dcstate, int B,
                       int H, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!whhT || !gates || !ct || !dhout || !dcstate || B < 0 || H < BKC || H % BKC) return MFPA_EINVAL;
  if (ldgn % 4 || ldg % 4 || ldct % 4 || ldcp % 4 || lddh % 4) return MFPA_EINVAL;
  static const int force = MFPA_EXP_ENV("MFPA_LSTM_MT", 0);
  static const int force_bu = MFPA_EXP_ENV("MFPA_LSTM_BU", 0);
  int MT = ((long long)(H / BU) * ((B + 31) / 32) <= 256) ? 1 : 2;   // 32-clip tiles while they leave the chip under-filled
  if (force == 1 || force == 2) MT = force;
  int but = (MT == 1 && (long long)(H / 16) * ((B + 31) / 32) <= 128) ? 16 : 32;   // 16-unit groups while even those leave half the chip free
  if (force_bu == 16 || force_bu == 32) but = (MT == 1) ? force_bu : 32;
  const int mtiles = (B + 32 * MT - 1) / (32 * MT);
  const long long total = (long long)(H / but) * mtiles;
  if (total > 0x7fffff) return MFPA_EINVAL;
  const unsigned grid = (unsigned)(((total + 7) / 8) * 8);
  const size_t lds = (size_t)2 * (32 * MT + 32) * BROW;
  if (MT == 1 && but == 16)
    hipLaunchKernelGGL((lstm_step_bwd_kernel<1, 16>), dim3(grid), dim3(BTHREADS), lds, mfpa_stream(stream), dgnext, ldgn, whhT, gates, ldg,
                       ct, ldct, cprev, ldcp, dhout, lddh, dcstate, B, H, mtiles);
  else if (MT == 1)
    hipLaunchKernelGGL((lstm_step_bwd_kernel<1, 32>), dim3(grid), dim3(BTHREADS), lds, mfpa_stream(stream), dgnext, ldgn, whhT, gates, ldg,
                       ct, ldct, cprev, ldcp, dhout, lddh, dcstate, B, H, mtiles);
  else
    hipLaunchKernelGGL((lstm_step_bwd_kernel<2, 32>), dim3(grid), dim3(BTHREADS), lds, mfpa_stream(stream), dgnext, ldgn, whhT, gates, ldg,
                       ct, ldct, cprev, ldcp, dhout, lddh, dcstate, B, H, mtiles);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

/* The backward recurrence of a whole LSTM layer: mfpa_lstm_step_bwd for t = Tn-1 .. 0 from one host loop.  gates / cseq / dhout
 * are (B, Tn, .) as mfpa_lstm_layer(train = 1) left them; dcstate (B, H) scratch (zeroed here). */
int mfpa_lstm_layer_bwd_range(const float* whhT, float* gates, const float* cseq, const float* dhout, float* dcstate, int B, int Tn,
                              int H, int t0, int t1, void* stream) {
  if (B == 0 || Tn == 0 || t1 <= t0) return MFPA_OK;
  if (!whhT || !gates || !cseq || !dhout || !dcstate || B < 0 || Tn < 0 || t0 < 0 || t1 > Tn) return MFPA_EINVAL;
  if (t1 == Tn) MFPA_HIP(hipMemsetAsync(dcstate, 0, (size_t)B * H * sizeof(float), mfpa_stream(stream)));
  const long long ldg = (long long)Tn * 4 * H, ldh = (long long)Tn * H;
  for (int t = t1 - 1; t >= t0; --t) {
    const int rc = mfpa_lstm_step_bwd(t + 1 < Tn ? gates + (size_t)(t + 1) * 4 * H : nullptr, ldg, whhT, gates + (size_t)t * 4 * H, ldg,
                                      cseq + (size_t)t * H, ldh, t ? cseq + (size_t)(t - 1) * H : nullptr, ldh, dhout + (size_t)t * H, ldh,
                                      dcstate, B, H, stream);
    if (rc != MFPA_OK) return rc;
  }
  return MFPA_OK;
}

int mfpa_lstm_layer_bwd(const float* whhT, float* gates, const float* cseq, const float* dhout, float* dcstate, int B, int Tn, int H,
                        void* stream) {
  return mfpa_lstm_layer_bwd_range(whhT, gates, cseq, dhout, dcstate, B, Tn, H, 0, Tn, stream);
}

/* mfpa_lstm_layer_bwd_range as ONE persistent launch (lstm_bwd_seq_kernel): same arguments and results; `work` = device scratch of the
 * size mfpa_lstm_bwd_seq_work_bytes reports, owned by this layer while the call runs, zeroed once before its first use; the error
 * word (bounded waits, as for mfpa_lstm_layer_seq) sits at the same byte offset.  Shapes outside the persistent kernel's range
 * (H / 64 not in {4, 8, 12}, more workgroups than `wg_budget` even with 64-clip slabs) take the per-step path inside the same call.
 * wg_budget: how many workgroups this launch may keep resident (0 = one per CU); a caller running two such launches at once passes half. */
int mfpa_lstm_bwd_seq_work_bytes(int B, int H, long long* bytes) {
  if (!bytes || B < 0 || H < 0) return MFPA_EINVAL;
  *bytes = (long long)QB_SYNC_WORDS * 4 + 2LL * B * 4 * H * 4;
  return MFPA_OK;
}

int mfpa_lstm_bwd_seq_workgroups(int B, int H, int wg_budget, int* workgroups) {
  if (!workgroups || B < 0 || H < 64) return MFPA_EINVAL;
  static const int persistent = MFPA_EXP_ENV("MFPA_LSTM_BWD_SEQ", 1);
  *workgroups = (persistent && B > 0) ? lstm_bwd_seq_plan(B, H, wg_budget, nullptr) : 0;
  return MFPA_OK;
}

int mfpa_lstm_layer_bwd_seq(const float* whhT, float* gates, const float* cseq, const float* dhout, float* dcstate, int B, int Tn, int H,
                            int t0, int t1, int wg_budget, void* work, void* stream) {
  if (B == 0 || Tn == 0 || t1 <= t0) return MFPA_OK;
  if (!whhT || !gates || !cseq || !dhout || !dcstate || !work || B < 0 || Tn < 0 || t0 < 0 || t1 > Tn || H < 64) return MFPA_EINVAL;
  const int ks = (H % 64 == 0) ? H / 64 : 0, ngroups = H / 16;
  static const int persistent = MFPA_EXP_ENV("MFPA_LSTM_BWD_SEQ", 1);
  // the smallest slab (16, 32 or 64 clips) whose workgroups still fit the chip: more CUs share the reading of dgates[t+1]
  // (every workgroup of a launch must be resident at once; a caller that runs two such launches side by side -- the chunked
  // two-stream pipeline -- passes half the CUs as wg_budget, 0 = all of them)
  int mtb = 0;
  const int wgs = persistent ? lstm_bwd_seq_plan(B, H, wg_budget, &mtb) : 0;
  if (wgs == 0)
    return mfpa_lstm_layer_bwd_range(whhT, gates, cseq, dhout, dcstate, B, Tn, H, t0, t1, stream);
  const int nslab = (B + 16 * mtb - 1) / (16 * mtb);
  LstmBwdSeqArgs a;
  a.whhT = whhT; a.gates = gates; a.cseq = cseq; a.dhout = dhout; a.dcstate = dcstate;
  a.sync = reinterpret_cast<unsigned*>(work);
  a.gsplit = reinterpret_cast<char*>(work) + (size_t)QB_SYNC_WORDS * 4;
  a.B = B; a.Tn = Tn; a.H = H; a.t0 = t0; a.t1 = t1; a.nslab = nslab; a.ngroups = ngroups;
  hipStream_t st = mfpa_stream(stream);
  MFPA_HIP(hipMemsetAsync(work, 0, (size_t)QB_ERR_WORD * 4, st));           // the slab counters; the error word stays
  const unsigned grid = (unsigned)(((nslab * ngroups + 7) / 8) * 8);
#define QB_LAUNCH(KS_)                                                                                         \
  if (mtb == 1) hipLaunchKernelGGL((lstm_bwd_seq_kernel<KS_, 1>), dim3(grid), dim3(64 * QB_W), 0, st, a);      \
  else if (mtb == 2) hipLaunchKernelGGL((lstm_bwd_seq_kernel<KS_, 2>), dim3(grid), dim3(64 * QB_W), 0, st, a); \
  else hipLaunchKernelGGL((lstm_bwd_seq_kernel<KS_, 4>), dim3(grid), dim3(64 * QB_W), 0, st, a)
  switch (ks) {
    case 4: QB_LAUNCH(4); break;
    case 8: QB_LAUNCH(8); break;
    default: QB_LAUNCH(12); break;
  }
#undef QB_LAUNCH
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

}  // extern "C"
