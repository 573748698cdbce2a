// Demucs causal waveform denoiser, forward, for MI355X (gfx950) -- next-tier row SURVEY.md §8f-2.
// Reference: training/model.py:22-110 (sinc resampling, LSTM) and :163-326 (Demucs) of deezer/musicFPaugment.
//
// Activations are time-major "NLC": (B, L, C) float32, channels contiguous.  In that layout every layer is a
// plain GEMM over overlapping row windows, all served by ONE batched strided fp32-MFMA kernel:
//   Conv1d(k=8, s=4)        : row t = x[4t : 4t+8] flattened (8C contiguous floats), row stride 4C, K = 8C
//   Conv1d(k=1) + GLU       : K = C, weights packed so a wave's 64-column tile is [32 values | their 32 gates]
//   ConvTranspose1d(k8, s4) : row t = [g[t-1] | g[t]] (2C contiguous floats of a buffer with one zero row at each
//                             end), N = 4*Cout: output row t is positions 4t..4t+3 -- the overlap-add of the
//                             transposed convolution becomes part of K; the skip connection of the next decoder
//                             layer is added in the epilogue
//   LSTM                    : input projection of all steps as one GEMM; per step gates = h[t-1] W_hh^T + X[t]
//                             (the same kernel, rows = clips) followed by the cell kernel
// The 1-channel ends (first Conv1d, last ConvTranspose1d), the sinc x2 resamplers and the std normalisation are
// small VALU kernels.
#include "mfpa_common.h"
#include <type_traits>
#include <cstdlib>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GKC = 16;            // K chunk
constexpr int GLD = GKC + 4;       // padded LDS row (floats): 80 B, conflict-free ds_read_b128 fragments
constexpr int GBM = 128, GBN = 64; // workgroup tile; 4 waves stacked along M, each 32 x 64

struct GemmArgs {
  const float* A; long long lda, strideA;      // A[b][m][k] = A[b*strideA + m*lda + k]
  const float* W;                              // [Npad][K], K contiguous, Npad multiple of 64 (zero rows beyond N)
  const float* bias;                           // [Npad] or null
  const float* addend; long long ldadd, strideAdd;   // mode 2: y += addend[b*strideAdd + m*ldadd + n]
  float* C; long long ldc, strideC;
  int exp;                                     // experiments only (MFPA_EXP_FLAG): bit 0 / 1 = every K chunk re-reads chunk 0 of A / W
  int nx, ny, nz, xcd;                         // tile grid (n tiles, m tiles, clips) of the 1-D launch; xcd = 1: XCD-aware tile order
  float* C2; long long ldc2, strideC2;         // training: second output (mode 1: the packed GLU pre-activations; relu 2: relu(y) before the addition)
  int M, N, K, mode, relu;                     // mode 0: bias(+relu); 1: GLU (N = output columns = Npad_total/2 pairs); 2: + addend;
                                               // 3: * (addend > 0) -- the ReLU backward mask of the layer that produced addend
  // fp32 kernel only: A is not read but COMPUTED while it is staged -- the first encoder layer Conv1d(1 -> K, k8, s4) + ReLU
  // (conv1d_c1_kernel) of the waveform: A[b][m][c] = relu(c1_b[c] + sum_j c1_w[j][c] * c1_x[b][4 m + j])
  const float* c1_x; long long c1_lin;         // (batch, c1_lin) samples
  const float* c1_w; const float* c1_b;        // (8, K), (K)
};

// Tile of a workgroup in the 1-D launch.  Consecutive workgroup ids are dealt round-robin over the 8 XCDs, each with its own L2:
// with the plain order the n tiles that share an A tile land on 8 different XCDs and every one of them fetches that A tile from
// memory (PMC: the mid-level launches fetched 2.7x what they wrote).  Here XCD k owns a contiguous range of the (clip, m tile,
// n tile) order, n fastest, so the workgroups sharing an A tile run back to back on ONE XCD.
__device__ __forceinline__ bool gemm_tile_at(const GemmArgs& a, unsigned vb, int& bx, int& by, int& bz) {
  const unsigned total = (unsigned)a.nx * a.ny * a.nz;
  unsigned lin = vb;
  if (a.xcd) {
    const unsigned per = (total + 7) / 8;
    if (vb / 8 >= per) return false;           // past the last virtual id (a persistent workgroup stepping by gridDim.x): NOT the next XCD's range
    lin = (vb % 8) * per + vb / 8;
  }
  if (lin >= total) return false;              // uniform over the workgroup, before any barrier
  bx = lin % a.nx;
  const unsigned rest = lin / a.nx;
  by = rest % a.ny;
  bz = rest / a.ny;
  return true;
}
__device__ __forceinline__ bool gemm_tile(const GemmArgs& a, int& bx, int& by, int& bz) { return gemm_tile_at(a, blockIdx.x, bx, by, bz); }

// epilogue shared by all GEMM kernels: D[row = m][col = n]; a lane holds column li of both 32-wide n tiles.
// The short-K launches are bound by vector-instruction issue, and the first form of this epilogue was most of it (64-bit
// address arithmetic and an exec-mask branch per element, an IEEE division in the sigmoid): here every offset is a 32-bit byte
// offset from a per-clip scalar base (the host checks that a clip's tile fits 4 GB), full row tiles skip the row bounds checks
// (FULL), column predicates are hoisted out of the row loop, and the sigmoid uses v_rcp_f32.
__device__ __forceinline__ float gemm_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ void st_f32(char* base, unsigned boff, float v) { *reinterpret_cast<float*>(base + boff) = v; }
__device__ __forceinline__ float ld_f32(const char* base, unsigned boff) { return *reinterpret_cast<const float*>(base + boff); }

template <bool FULL>
__device__ __forceinline__ void gemm_epilogue_t(const GemmArgs& a, floatx16 (&acc)[2], int m0, int n0, int b, int wave, int li, int lh) {
  const float bias0 = a.bias ? a.bias[n0 + li] : 0.f;
  const float bias1 = a.bias ? a.bias[n0 + 32 + li] : 0.f;
  char* Cb = reinterpret_cast<char*>(a.C + (size_t)b * a.strideC);
  char* C2b = a.C2 ? reinterpret_cast<char*>(a.C2 + (size_t)b * a.strideC2) : nullptr;
  const unsigned ldc = (unsigned)a.ldc, ldc2 = (unsigned)a.ldc2;
  const int mbase = m0 + wave * 32 + 4 * lh;
  if (a.mode == 1) {                           // GLU: value * sigmoid(gate); output column = (tile index) * 32 + li
    const int n = (n0 / 64) * 32 + li;
    const bool ok = n < a.N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mbase + (r & 3) + 8 * (r >> 2);
      if (!FULL && m >= a.M) continue;
      const float v0 = acc[0][r] + bias0, v1 = acc[1][r] + bias1;
      if (C2b) {                               // the pre-activations in the packed [32 values | 32 gates] tile order
        const unsigned o = ((unsigned)m * ldc2 + (unsigned)(n0 + li)) * 4u;
        st_f32(C2b, o, v0);
        st_f32(C2b, o + 128u, v1);
      }
      if (ok) st_f32(Cb, ((unsigned)m * ldc + (unsigned)n) * 4u, v0 * gemm_sigmoid(v1));
    }
    return;
  }
  const int n = n0 + li;
  const bool ok0 = n < a.N, ok1 = n + 32 < a.N;
  const unsigned ldadd = (unsigned)a.ldadd;
  // modes 2 / 3: ALL of this lane's addend values are loaded before the first store -- the addend may alias C as far as the
  // compiler knows, so loads interleaved with the stores were serialised into 16 exposed memory round trips per wave (PMC: the
  // transposed-convolution launches spent 57 .. 70 % of their wave cycles in s_waitcnt)
  float ad0[16], ad1[16];
  if (a.mode >= 2) {
    const char* Adb = reinterpret_cast<const char*>(a.addend + (size_t)b * a.strideAdd);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mbase + (r & 3) + 8 * (r >> 2);
      const unsigned oa = ((unsigned)m * ldadd + (unsigned)n) * 4u;
      const bool rowok = FULL || m < a.M;
      ad0[r] = (rowok && ok0) ? ld_f32(Adb, oa) : 0.f;
      ad1[r] = (rowok && ok1) ? ld_f32(Adb, oa + 128u) : 0.f;
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = mbase + (r & 3) + 8 * (r >> 2);
    if (!FULL && m >= a.M) continue;
    float v0 = acc[0][r] + bias0, v1 = acc[1][r] + bias1;
    const unsigned o = ((unsigned)m * ldc + (unsigned)n) * 4u;
    if (a.relu == 2) {                         // ReLU before the skip addition
      v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f);
      if (C2b) {
        const unsigned o2 = ((unsigned)m * ldc2 + (unsigned)n) * 4u;
        if (ok0) st_f32(C2b, o2, v0);
        if (ok1) st_f32(C2b, o2 + 128u, v1);
      }
    }
    if (a.mode == 2) {
      v0 += ad0[r]; v1 += ad1[r];
    } else if (a.mode == 3) {
      if (C2b) {                               // the unmasked gradient as well (it is also the skip connection's gradient)
        const unsigned o2 = ((unsigned)m * ldc2 + (unsigned)n) * 4u;
        if (ok0) st_f32(C2b, o2, v0);
        if (ok1) st_f32(C2b, o2 + 128u, v1);
      }
      v0 = ad0[r] > 0.f ? v0 : 0.f;
      v1 = ad1[r] > 0.f ? v1 : 0.f;
    }
    if (a.relu == 1) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
    if (ok0) st_f32(Cb, o, v0);
    if (ok1) st_f32(Cb, o + 128u, v1);
  }
}

__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, floatx16 (&acc)[2], int m0, int n0, int b, int wave, int li, int lh) {
  if (m0 + GBM <= a.M) gemm_epilogue_t<true>(a, acc, m0, n0, b, wave, li, lh);
  else gemm_epilogue_t<false>(a, acc, m0, n0, b, wave, li, lh);
}

template <bool C1SRC>
__global__ __launch_bounds__(256, 2) void gemm_mfma_kernel(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float As[2][GBM * GLD];
  __shared__ __attribute__((aligned(16))) float Bs[2][GBN * GLD];
  __shared__ __attribute__((aligned(16))) float W1s[C1SRC ? 9 * 256 : 4];      // C1SRC: [8][K] taps then [K] bias, K <= 256
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  int bx, by, b;
  if (!gemm_tile(a, bx, by, b)) return;
  const int n0 = bx * GBN, m0 = by * GBM;
  const float* Ab = a.A + (size_t)b * a.strideA;
  const int nk = a.K / GKC;

  // staging: A tile 128 rows x 4 float4, B tile 64 rows x 4 float4
  f32x4 ar[2], br;
  const int arow0 = tid >> 2, aq = tid & 3;             // rows arow0 and arow0 + 64
  const int brow = tid >> 2, bq = tid & 3;              // 64 rows
  f32x4 xr[2][2];                              // C1SRC: the 8 samples under each of this thread's two rows, loaded once
  if (C1SRC) {
    const float* xb = a.c1_x + (size_t)b * a.c1_lin;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + arow0 + 64 * i;
      xr[i][0] = xr[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (m < a.M) {
        xr[i][0] = *reinterpret_cast<const f32x4*>(xb + 4 * (size_t)m);
        xr[i][1] = *reinterpret_cast<const f32x4*>(xb + 4 * (size_t)m + 4);
      }
    }
    for (int i = tid; i < 8 * a.K; i += 256) W1s[i] = a.c1_w[i];
    for (int i = tid; i < a.K; i += 256) W1s[8 * a.K + i] = a.c1_b[i];
    __syncthreads();
  }
  auto load = [&](int kc) __attribute__((always_inline)) {
    if (C1SRC) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {             // conv1d_c1_kernel's arithmetic, same order: bias, then taps 0..7
        const int c = kc * GKC + 4 * aq;
        f32x4 v = *reinterpret_cast<const f32x4*>(&W1s[8 * a.K + c]);
#pragma unroll
        for (int j = 0; j < 8; ++j) v += xr[i][j >> 2][j & 3] * *reinterpret_cast<const f32x4*>(&W1s[j * a.K + c]);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
        ar[i] = (m0 + arow0 + 64 * i < a.M) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = m0 + arow0 + 64 * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < a.M) v = *reinterpret_cast<const f32x4*>(Ab + (size_t)m * a.lda + kc * GKC + 4 * aq);
        ar[i] = v;
      }
    }
    br = *reinterpret_cast<const f32x4*>(a.W + (size_t)(n0 + brow) * a.K + kc * GKC + 4 * bq);
  };
  auto store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(&As[buf][(arow0 + 64 * i) * GLD + 4 * aq]) = ar[i];
    *reinterpret_cast<f32x4*>(&Bs[buf][brow * GLD + 4 * bq]) = br;
  };

  floatx16 acc[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

  load(0);
  store(0);
  if (nk > 1) load(1);
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < nk) store(buf ^ 1);          // B/A(kc+1) registers -> the other LDS buffer (free since the last barrier)
    if (kc + 2 < nk) load(kc + 2);
    const float* Ap = &As[buf][(wave * 32 + li) * GLD + 4 * lh];
    const float* Bp = &Bs[buf][li * GLD + 4 * lh];
#pragma unroll
    for (int s = 0; s < GKC / 8; ++s) {
      const f32x4 af = *reinterpret_cast<const f32x4*>(Ap + 8 * s);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bp + 8 * s);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(Bp + 32 * GLD + 8 * s);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[k], b0[k], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[k], b1[k], acc[1], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  gemm_epilogue(a, acc, m0, n0, b, wave, li, lh);
}

// Short-K form (the 48-channel 1x1 + GLU layers at the full-rate ends of the network: 16.4 M rows x 48 x 96, as much HBM time
// as MFMA time).  The whole K extent of both tiles is staged at once -- every global load of the workgroup is in flight
// together, one barrier, no chunk loop -- which is what a three-chunk pipeline cannot do: it exposed one memory latency
// per chunk.
template <bool C1SRC, int K>
__global__ __launch_bounds__(256, 3) void gemm_smallk_kernel(GemmArgs a) {
  constexpr int LD = K + 4, QPR = K / 4;                // LDS row (floats): 208 B for K = 48, conflict-free ds_read_b128
  constexpr int A_F4 = (GBM * QPR + 255) / 256, B_F4 = (GBN * QPR + 255) / 256;
  __shared__ __attribute__((aligned(16))) float As[GBM * LD];
  __shared__ __attribute__((aligned(16))) float Bs[GBN * LD];
  __shared__ __attribute__((aligned(16))) float W1s[C1SRC ? 9 * K : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  int bx, by, b;
  if (!gemm_tile(a, bx, by, b)) return;
  const int n0 = bx * GBN, m0 = by * GBM;
  f32x4 ar[A_F4], br[B_F4], x0[C1SRC ? A_F4 : 1], x1[C1SRC ? A_F4 : 1];
#pragma unroll
  for (int i = 0; i < B_F4; ++i) {
    const int idx = tid + 256 * i;
    br[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (idx < GBN * QPR) br[i] = *reinterpret_cast<const f32x4*>(a.W + (size_t)(n0 + idx / QPR) * K + 4 * (idx % QPR));
  }
  if (C1SRC) {
    const float* xb = a.c1_x + (size_t)b * a.c1_lin;
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const int idx = tid + 256 * i, m = m0 + idx / QPR;
      x0[i] = x1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (idx < GBM * QPR && m < a.M) {
        x0[i] = *reinterpret_cast<const f32x4*>(xb + 4 * (size_t)m);
        x1[i] = *reinterpret_cast<const f32x4*>(xb + 4 * (size_t)m + 4);
      }
    }
    for (int i = tid; i < 8 * K; i += 256) W1s[i] = a.c1_w[i];
    for (int i = tid; i < K; i += 256) W1s[8 * K + i] = a.c1_b[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {              // conv1d_c1_kernel's arithmetic, same order: bias, then taps 0..7
      const int idx = tid + 256 * i, c = 4 * (idx % QPR);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < GBM * QPR && m0 + idx / QPR < a.M) {
        v = *reinterpret_cast<const f32x4*>(&W1s[8 * K + c]);
#pragma unroll
        for (int j = 0; j < 4; ++j) v += x0[i][j] * *reinterpret_cast<const f32x4*>(&W1s[j * K + c]);
#pragma unroll
        for (int j = 0; j < 4; ++j) v += x1[i][j] * *reinterpret_cast<const f32x4*>(&W1s[(4 + j) * K + c]);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
      }
      ar[i] = v;
    }
  } else {
    const float* Ab = a.A + (size_t)b * a.strideA;
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const int idx = tid + 256 * i, m = m0 + idx / QPR;
      ar[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (idx < GBM * QPR && m < a.M) ar[i] = *reinterpret_cast<const f32x4*>(Ab + (size_t)m * a.lda + 4 * (idx % QPR));
    }
  }
#pragma unroll
  for (int i = 0; i < A_F4; ++i) {
    const int idx = tid + 256 * i;
    if (idx < GBM * QPR) *reinterpret_cast<f32x4*>(&As[(idx / QPR) * LD + 4 * (idx % QPR)]) = ar[i];
  }
#pragma unroll
  for (int i = 0; i < B_F4; ++i) {
    const int idx = tid + 256 * i;
    if (idx < GBN * QPR) *reinterpret_cast<f32x4*>(&Bs[(idx / QPR) * LD + 4 * (idx % QPR)]) = br[i];
  }
  __syncthreads();
  floatx16 acc[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
  const float* Ap = &As[(wave * 32 + li) * LD + 4 * lh];
  const float* Bp = &Bs[li * LD + 4 * lh];
#pragma unroll
  for (int s = 0; s < K / 8; ++s) {
    const f32x4 af = *reinterpret_cast<const f32x4*>(Ap + 8 * s);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bp + 8 * s);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(Bp + 32 * LD + 8 * s);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[k], b0[k], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[k], b1[k], acc[1], 0, 0, 0);
    }
  }
  gemm_epilogue(a, acc, m0, n0, b, wave, li, lh);
}

// The same GEMM with bf16x3 products (precision 1): K chunks of 32, LDS rows [32 hi | 32 lo | pad] = 144 B (the layout of
// conv_mfma_kernel<PREC 1>), operands split while they are staged, 3 x v_mfma_f32_32x32x16_bf16 per fp32 product.
typedef __bf16 g_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 g_bf16x4 __attribute__((ext_vector_type(4)));
constexpr int HKC = 32, HROW = 144;

__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) char As[2][GBM * HROW];
  __shared__ __attribute__((aligned(16))) char Bs[2][GBN * HROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  int bx, by, b;
  if (!gemm_tile(a, bx, by, b)) return;
  const int n0 = bx * GBN, m0 = by * GBM;
  const float* Ab = a.A + (size_t)b * a.strideA;
  const int nk = a.K / HKC;

  // staging: A tile 128 rows x 8 float4 (4 per thread), B tile 64 rows x 8 float4 (2 per thread)
  f32x4 ar[4], br[2];
  const int q = tid & 7, r0 = tid >> 3;                  // rows r0 + 32 i
  auto load = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + r0 + 32 * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < a.M) v = *reinterpret_cast<const f32x4*>(Ab + (size_t)m * a.lda + kc * HKC + 4 * q);
      ar[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      br[i] = *reinterpret_cast<const f32x4*>(a.W + (size_t)(n0 + r0 + 32 * i) * a.K + kc * HKC + 4 * q);
  };
  auto split_store = [&](char* row, f32x4 v) __attribute__((always_inline)) {
    g_bf16x4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hi[k] = (__bf16)v[k];
      lo[k] = (__bf16)(v[k] - (float)hi[k]);
    }
    *reinterpret_cast<g_bf16x4*>(row + 8 * q) = hi;
    *reinterpret_cast<g_bf16x4*>(row + 64 + 8 * q) = lo;
  };
  auto store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) split_store(&As[buf][(r0 + 32 * i) * HROW], ar[i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) split_store(&Bs[buf][(r0 + 32 * i) * HROW], br[i]);
  };

  floatx16 acc[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

  load(0);
  store(0);
  if (nk > 1) load(1);
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < nk) store(buf ^ 1);
    if (kc + 2 < nk) load(kc + 2);
    const char* Ap = &As[buf][(wave * 32 + li) * HROW + 16 * lh];
    const char* Bp = &Bs[buf][li * HROW + 16 * lh];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const g_bf16x8 ah = *reinterpret_cast<const g_bf16x8*>(Ap + 32 * s);
      const g_bf16x8 al = *reinterpret_cast<const g_bf16x8*>(Ap + 64 + 32 * s);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const g_bf16x8 bh = *reinterpret_cast<const g_bf16x8*>(Bp + nt * 32 * HROW + 32 * s);
        const g_bf16x8 bl = *reinterpret_cast<const g_bf16x8*>(Bp + nt * 32 * HROW + 64 + 32 * s);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[nt], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  gemm_epilogue(a, acc, m0, n0, b, wave, li, lh);
}

// The same with a 128 x 128 workgroup tile of four 64 x 64 wave tiles (2 x 2 MFMA tiles per wave): a wave reads 8 fragments per
// 12 MFMAs instead of 6 per 6 -- the 32 x 64 wave tile above keeps the LDS pipe as busy as the matrix pipe (PMC: 30 % of the wave
// cycles issue-stalled on the mid-level launches).  Used when the padded N is a multiple of 128.  73.7 KB of LDS, two workgroups
// per CU.
constexpr int WBN = 128;

// WSPLIT: W arrives already split -- every 32-element chunk of a row as [32 bf16 hi | 32 bf16 lo] (ops_demucs.split_rows, for weights
// that do not change between calls): staging it is a 16-byte copy instead of 12 vector instructions per quad in every workgroup.
template <bool WSPLIT>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_wide_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char wsm[];
  char* As = wsm;                                    // [2][GBM][HROW]
  char* Bs = wsm + 2 * GBM * HROW;                   // [2][WBN][HROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  int bx, by, b;
  if (!gemm_tile(a, bx, by, b)) return;
  const int n0 = bx * WBN, m0 = by * GBM;
  const float* Ab = a.A + (size_t)b * a.strideA;
  const int nk = a.K / HKC;

  f32x4 ar[4], br[4];
  const int q = tid & 7, r0 = tid >> 3;                  // rows r0 + 32 i
  auto load = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + r0 + 32 * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < a.M) v = *reinterpret_cast<const f32x4*>(Ab + (size_t)m * a.lda + kc * HKC + 4 * q);
      ar[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      br[i] = *reinterpret_cast<const f32x4*>(a.W + (size_t)(n0 + r0 + 32 * i) * a.K + kc * HKC + 4 * q);
  };
  auto split_store = [&](char* row, f32x4 v) __attribute__((always_inline)) {
    g_bf16x4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hi[k] = (__bf16)v[k];
      lo[k] = (__bf16)(v[k] - (float)hi[k]);
    }
    *reinterpret_cast<g_bf16x4*>(row + 8 * q) = hi;
    *reinterpret_cast<g_bf16x4*>(row + 64 + 8 * q) = lo;
  };
  auto store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) split_store(As + (buf * GBM + r0 + 32 * i) * HROW, ar[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (WSPLIT) *reinterpret_cast<f32x4*>(Bs + (buf * WBN + r0 + 32 * i) * HROW + 16 * q) = br[i];
      else split_store(Bs + (buf * WBN + r0 + 32 * i) * HROW, br[i]);
    }
  };

  floatx16 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  load(0);
  store(0);
  if (nk > 1) load(1);
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < nk) store(buf ^ 1);
    if (kc + 2 < nk) load(kc + 2);
    const char* Ap = As + (buf * GBM + wm * 64 + li) * HROW + 16 * lh;
    const char* Bp = Bs + (buf * WBN + wn * 64 + li) * HROW + 16 * lh;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      g_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        ah[t] = *reinterpret_cast<const g_bf16x8*>(Ap + t * 32 * HROW + 32 * s);
        al[t] = *reinterpret_cast<const g_bf16x8*>(Ap + t * 32 * HROW + 64 + 32 * s);
        bh[t] = *reinterpret_cast<const g_bf16x8*>(Bp + t * 32 * HROW + 32 * s);
        bl[t] = *reinterpret_cast<const g_bf16x8*>(Bp + t * 32 * HROW + 64 + 32 * s);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  // the shared epilogue takes a 32-row x 64-column wave tile at row (wave index) * 32: two calls per wave
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) gemm_epilogue(a, acc[mt], m0, n0 + wn * 64, b, wm * 2 + mt, li, lh);
}

// The software-pipelined form of the wide kernel (the schedule of conv_mfma_kernel<PREC 1>'s tap body, unet.hip): 8 waves own a
// 256 x 128 tile (4 x 2 waves of 64 x 64), one workgroup per CU.  A K chunk of 32 is two k-steps; the fragments of a k-step are
// read from LDS one step AHEAD of the MFMAs that use them (two register sets of fragments), the barrier sits between the two
// k-steps of a chunk, and the global loads run two chunks ahead of their LDS stores through two register sets (straight-line
// code per chunk parity, so hipcc's vmcnt bookkeeping is exact: no s_waitcnt vmcnt(0) in the steady state).  Rows past M are
// clamped to row M - 1 when loaded (never stored), so every load is unconditional.  K must be a multiple of 64, at least 128.
constexpr int PBM = 256;
template <int SLOTS, int LEFT, int I = 0>
__device__ __forceinline__ void g_pin_reads() {          // "one MFMA, then k LDS reads", LEFT reads spread over SLOTS MFMAs
  if constexpr (I < SLOTS && LEFT > 0) {
    constexpr int k = (LEFT + (SLOTS - I) - 1) / (SLOTS - I);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, k, 0);
    g_pin_reads<SLOTS, LEFT - k, I + 1>();
  }
}

template <bool WSPLIT>
__global__ __launch_bounds__(512, 1) void gemm_bf16x3_pipe_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char psm[];
  constexpr int ASZ = PBM * HROW, BSZ = WBN * HROW;
  char* As = psm;                                    // [2][PBM][HROW]
  char* Bs = psm + 2 * ASZ;                          // [2][WBN][HROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave & 3, wn = wave >> 2;
  const int nk = a.K / HKC;
  const int q = tid & 7, r0 = tid >> 3;              // staging: column quad, rows r0 + 64 i
  // PERSISTENT over tiles: workgroup w takes the virtual workgroup ids w, w + gridDim.x, ... (gridDim.x is a multiple of 8, so they all map
  // into the same XCD's range of gemm_tile_at).  One workgroup owns a CU (111 KB of LDS), so as one tile per workgroup nothing overlapped a
  // tile's first loads (a memory round trip before the first MFMA) and its epilogue (stores, addend loads) with anything: on the K = 192 .. 384
  // layers those two were as long as the main loop.  Here the first two chunks of the NEXT tile are requested before the epilogue of the
  // current one (their staging registers are free by then), so the round trip runs under the epilogue.
  struct TileP { const float* ap[4]; const float* bp[2]; int m0, n0, b; };
  auto tile_at = [&](unsigned vb, TileP& t) __attribute__((always_inline)) -> bool {
    int bx, by, b;
    if (!gemm_tile_at(a, vb, bx, by, b)) return false;
    t.n0 = bx * WBN; t.m0 = by * PBM; t.b = b;
    const float* Ab = a.A + (size_t)b * a.strideA;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m = t.m0 + r0 + 64 * i;
      m = m < a.M ? m : a.M - 1;
      t.ap[i] = Ab + (size_t)m * a.lda + 4 * q;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) t.bp[i] = a.W + (size_t)(t.n0 + r0 + 64 * i) * a.K + 4 * q;
    return true;
  };
  struct Stage { f32x4 a[4], b[2]; };
  Stage st0, st1;
  auto load = [&](const TileP& t, int kc, Stage& st) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) st.a[i] = *reinterpret_cast<const f32x4*>(t.ap[i] + (MFPA_EXP_FLAG(a.exp, 1) ? 0 : kc) * HKC);
#pragma unroll
    for (int i = 0; i < 2; ++i) st.b[i] = *reinterpret_cast<const f32x4*>(t.bp[i] + (MFPA_EXP_FLAG(a.exp, 2) ? 0 : kc) * HKC);
  };
  auto split_store = [&](char* row, f32x4 v) __attribute__((always_inline)) {
    g_bf16x4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hi[k] = (__bf16)v[k];
      lo[k] = (__bf16)(v[k] - (float)hi[k]);
    }
    *reinterpret_cast<g_bf16x4*>(row + 8 * q) = hi;
    *reinterpret_cast<g_bf16x4*>(row + 64 + 8 * q) = lo;
  };
  auto store = [&](int buf, const Stage& st) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#ifdef MFPA_GEMM_ACOPY   // timing-only variant (wrong results by design): the activation tile is COPIED into LDS as if it arrived already split
      *reinterpret_cast<f32x4*>(As + buf * ASZ + (r0 + 64 * i) * HROW + 16 * q) = st.a[i];
#else
      split_store(As + buf * ASZ + (r0 + 64 * i) * HROW, st.a[i]);
#endif
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (WSPLIT) *reinterpret_cast<f32x4*>(Bs + buf * BSZ + (r0 + 64 * i) * HROW + 16 * q) = st.b[i];
      else split_store(Bs + buf * BSZ + (r0 + 64 * i) * HROW, st.b[i]);
    }
  };
  struct Frags { g_bf16x8 ah[2], al[2], bh[2], bl[2]; };
  Frags fr0, fr1;
  const char* Ap = As + (wm * 64 + li) * HROW + 16 * lh;
  const char* Bp = Bs + (wn * 64 + li) * HROW + 16 * lh;
  auto read_frags = [&](Frags& f, int buf, int s) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f.ah[t] = *reinterpret_cast<const g_bf16x8*>(Ap + buf * ASZ + t * 32 * HROW + 32 * s);
      f.al[t] = *reinterpret_cast<const g_bf16x8*>(Ap + buf * ASZ + t * 32 * HROW + 64 + 32 * s);
      f.bh[t] = *reinterpret_cast<const g_bf16x8*>(Bp + buf * BSZ + t * 32 * HROW + 32 * s);
      f.bl[t] = *reinterpret_cast<const g_bf16x8*>(Bp + buf * BSZ + t * 32 * HROW + 64 + 32 * s);
    }
  };
  floatx16 acc[2][2];
  auto mfma12 = [&](const Frags& f) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[mt], f.bh[nt], acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mt], f.bl[nt], acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mt], f.bh[nt], acc[mt][nt], 0, 0, 0);
      }
  };
  constexpr int N_DSW = WSPLIT ? 10 : 12;            // LDS stores of one chunk per thread
  TileP cur;
  // one K chunk; SET = chunk parity = LDS buffer it is computed from; STORE: chunk kc + 1 (register set SET ^ 1) goes to the other
  // buffer; LOAD: that register set is refilled with chunk kc + 3
  auto body = [&](auto SET_, auto STORE_, auto LOAD_, int kc) __attribute__((always_inline)) {
    constexpr int SET = decltype(SET_)::value;
    constexpr bool STORE = decltype(STORE_)::value, LOAD = decltype(LOAD_)::value;
    Stage& other = SET ? st0 : st1;
    read_frags(fr1, SET, 1);
    mfma12(fr0);
    if constexpr (STORE) store(SET ^ 1, other);
    if constexpr (LOAD) load(cur, kc + 3, other);
    g_pin_reads<8, 8>();
    if constexpr (STORE) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, N_DSW / 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, N_DSW - N_DSW / 2, 0);
      if constexpr (LOAD) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      }
    } else {
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (STORE) read_frags(fr0, SET ^ 1, 0);
    mfma12(fr1);
    if constexpr (STORE) g_pin_reads<12, 8>();
    __builtin_amdgcn_sched_barrier(0);
  };
  using T = std::true_type;
  using F = std::false_type;
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  unsigned vb = blockIdx.x;
  if (!tile_at(vb, cur)) return;                         // uniform over the workgroup, before any barrier
  // (the scheduling barriers keep the prologue's loads in the order the loop issues them, so the vmcnt state merged at the loop
  // header is the steady state's and the first stores of a trip do not wait for the newest loads)
  load(cur, 0, st0);
  __builtin_amdgcn_sched_barrier(0);
  load(cur, 1, st1);
  __builtin_amdgcn_sched_barrier(0);
  for (;;) {
    store(0, st0);
    __builtin_amdgcn_sched_barrier(0);
    load(cur, 2, st0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    read_frags(fr0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    int kc = 0;
    for (; kc < nk - 4; kc += 2) {
      body(S0{}, T{}, T{}, kc);
      body(S1{}, T{}, T{}, kc + 1);
    }
    body(S0{}, T{}, T{}, kc);                               // kc = nk - 4: chunk nk - 1 is the last load
    body(S1{}, T{}, F{}, kc + 1);
    body(S0{}, T{}, F{}, kc + 2);
    body(S1{}, F{}, F{}, kc + 3);
    // the next tile's first two chunks go out now: both staging sets are free, and nobody reads LDS after the barrier inside the last body
    const int m0 = cur.m0, n0 = cur.n0, b = cur.b;
    vb += gridDim.x;
    const bool more = tile_at(vb, cur);
    if (more) {
      load(cur, 0, st0);
      __builtin_amdgcn_sched_barrier(0);
      load(cur, 1, st1);
      __builtin_amdgcn_sched_barrier(0);
    }
    const bool full = m0 + PBM <= a.M;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      if (full) gemm_epilogue_t<true>(a, acc[mt], m0, n0 + wn * 64, b, wm * 2 + mt, li, lh);
      else gemm_epilogue_t<false>(a, acc[mt], m0, n0 + wn * 64, b, wm * 2 + mt, li, lh);
    }
    if (!more) break;
  }
}

// Short-K layers with bf16x3 products (K = 48 and 96: the full-rate outer levels of the network; measured on 256 clips: forward
// 50.1 -> 46.8 ms with these two; the same form for K = 128 / 192 was slower than the fp32 kernel and is not instantiated).  With exact fp32 products
// these launches are bound by the fp32 MFMA rate (64 cycles per 32x32x2), not by HBM: e.g. the 1x1 + GLU of level 0 is 151 GFLOP
// per 256 clips = 1.2 ms of fp32 MFMA against 0.8 ms of HBM traffic.  Here EVERY global load of the workgroup (all NCH chunks
// of both tiles, up to 36 float4 per thread) is issued before anything is consumed; the chunks are then split to bf16 hi / lo and
// staged one at a time through a single small LDS buffer (40 .. 77 KB, two workgroups per CU), 3 x v_mfma_f32_32x32x16_bf16 per
// product.  Same tile (128 x 64, four 32 x 64 wave tiles) and epilogue as the other GEMM kernels.
template <bool C1SRC, int K, int KCW>
__global__ __launch_bounds__(256, 2) void gemm_shortk_bf16x3_kernel(GemmArgs a) {
  static_assert(K % KCW == 0 && KCW % 16 == 0 && (!C1SRC || K == KCW), "chunking");
  constexpr int NCH = K / KCW, QPR = KCW / 4, ROW = 4 * KCW + 16;   // LDS row bytes: [KCW hi | KCW lo | pad], conflict-free b128 reads
  constexpr int A_F4 = (GBM * QPR + 255) / 256, B_F4 = (GBN * QPR + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char sksm[];
  char* As = sksm;                                                      // [GBM][ROW]
  char* Bs = sksm + GBM * ROW;                                          // [GBN][ROW]
  float* W1s = reinterpret_cast<float*>(sksm + (GBM + GBN) * ROW);      // C1SRC: [8][K] taps then [K] bias
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  int bx, by, b;
  if (!gemm_tile(a, bx, by, b)) return;
  const int n0 = bx * GBN, m0 = by * GBM;
  f32x4 ar[NCH][A_F4], br[NCH][B_F4];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      const int idx = tid + 256 * i;
      br[c][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (idx < GBN * QPR) br[c][i] = *reinterpret_cast<const f32x4*>(a.W + (size_t)(n0 + idx / QPR) * K + c * KCW + 4 * (idx % QPR));
    }
  if (C1SRC) {
    f32x4 x0[A_F4], x1[A_F4];
    const float* xb = a.c1_x + (size_t)b * a.c1_lin;
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const int idx = tid + 256 * i, m = m0 + idx / QPR;
      x0[i] = x1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (idx < GBM * QPR && m < a.M) {
        x0[i] = *reinterpret_cast<const f32x4*>(xb + 4 * (size_t)m);
        x1[i] = *reinterpret_cast<const f32x4*>(xb + 4 * (size_t)m + 4);
      }
    }
    for (int i = tid; i < 8 * K; i += 256) W1s[i] = a.c1_w[i];
    for (int i = tid; i < K; i += 256) W1s[8 * K + i] = a.c1_b[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {              // conv1d_c1_kernel's arithmetic, same order: bias, then taps 0..7
      const int idx = tid + 256 * i, c = 4 * (idx % QPR);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < GBM * QPR && m0 + idx / QPR < a.M) {
        v = *reinterpret_cast<const f32x4*>(&W1s[8 * K + c]);
#pragma unroll
        for (int j = 0; j < 4; ++j) v += x0[i][j] * *reinterpret_cast<const f32x4*>(&W1s[j * K + c]);
#pragma unroll
        for (int j = 0; j < 4; ++j) v += x1[i][j] * *reinterpret_cast<const f32x4*>(&W1s[(4 + j) * K + c]);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
      }
      ar[0][i] = v;
    }
  } else {
    const float* Ab = a.A + (size_t)b * a.strideA;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < A_F4; ++i) {
        const int idx = tid + 256 * i, m = m0 + idx / QPR;
        ar[c][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (idx < GBM * QPR && m < a.M) ar[c][i] = *reinterpret_cast<const f32x4*>(Ab + (size_t)m * a.lda + c * KCW + 4 * (idx % QPR));
      }
  }
  auto split_store = [&](char* row, int q, f32x4 v) __attribute__((always_inline)) {
    g_bf16x4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hi[k] = (__bf16)v[k];
      lo[k] = (__bf16)(v[k] - (float)hi[k]);
    }
    *reinterpret_cast<g_bf16x4*>(row + 8 * q) = hi;
    *reinterpret_cast<g_bf16x4*>(row + 2 * KCW + 8 * q) = lo;
  };
  floatx16 acc[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
  const char* Ap = As + (wave * 32 + li) * ROW + 16 * lh;
  const char* Bp = Bs + li * ROW + 16 * lh;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c > 0) __syncthreads();                  // the previous chunk's fragment reads are done
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const int idx = tid + 256 * i;
      if (idx < GBM * QPR) split_store(As + (idx / QPR) * ROW, idx % QPR, ar[c][i]);
    }
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      const int idx = tid + 256 * i;
      if (idx < GBN * QPR) split_store(Bs + (idx / QPR) * ROW, idx % QPR, br[c][i]);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KCW / 16; ++s) {
      const g_bf16x8 ah = *reinterpret_cast<const g_bf16x8*>(Ap + 32 * s);
      const g_bf16x8 al = *reinterpret_cast<const g_bf16x8*>(Ap + 2 * KCW + 32 * s);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const g_bf16x8 bh = *reinterpret_cast<const g_bf16x8*>(Bp + nt * 32 * ROW + 32 * s);
        const g_bf16x8 bl = *reinterpret_cast<const g_bf16x8*>(Bp + nt * 32 * ROW + 2 * KCW + 32 * s);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[nt], 0, 0, 0);
      }
    }
  }
  gemm_epilogue(a, acc, m0, n0, b, wave, li, lh);
}

// ---------------------------------------------------------------------------------- last decoder level in one launch
// Conv1d(1x1, C -> 2C) + GLU + ConvTranspose1d(C -> 1, k8, s4) of the last decoder level (model.py:80-88,316-318) without the
// (B, L, C) intermediate: as two launches that tensor (3.15 GB per 256 clips at C = 48) is written once and read twice.
// A workgroup owns 127 output groups of a clip (group t = samples 4t .. 4t+3 = taps 0..3 of g[t] + taps 4..7 of g[t-1]) and
// computes the 128 GLU rows g[127 by - 1 .. 127 by + 126] they need (one row of overlap instead of an exchange):
//   * the 1x1 + GLU is the short-K bf16x3 MFMA GEMM of gemm_shortk_bf16x3_kernel (K = C = 48 in one chunk, all 128 packed columns
//     in one workgroup, so a row's 48 GLU outputs meet in one place); W is staged once per workgroup and reused for TPW tiles;
//   * g goes to LDS in fp32 (over the A rows of the same wave: no extra barrier), a thread pair per row forms the 8 tap sums
//     p[t][j] = sum_c g[t][c] w[j][c] with fp32 FMAs, and y[4t + j] = bias + p[t][j] + p[t-1][j+4] is written coalesced.
constexpr int TT_K = 48, TT_ROW = 4 * TT_K + 16, TT_OUT = 127, TT_TPW = 8;

__global__ MFPA_NO_PK_F32 __launch_bounds__(256, 2) void glu_convT_c1_kernel(const float* __restrict__ x, int L, const float* __restrict__ gw,
                                                              const float* __restrict__ gb, const float* __restrict__ wl, float bias,
                                                              float* __restrict__ y, int tiles_per_clip, int groups) {
  constexpr int K = TT_K, ROW = TT_ROW, QPR = K / 4, NQ = 128 * QPR / 256;      // 12 quads per row, 6 per thread
  __shared__ __attribute__((aligned(16))) char As[128 * ROW];                   // A rows (bf16 hi | lo), then g rows (fp32) of the same wave
  __shared__ __attribute__((aligned(16))) char Ws[128 * ROW];
  __shared__ __attribute__((aligned(16))) float wls[8 * K];
  __shared__ __attribute__((aligned(16))) float Ps[128 * 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y;
  const int tile0 = blockIdx.x * TT_TPW;
  const float* xb = x + (size_t)b * L * K;
  float* yb = y + (size_t)b * 4 * (L + 1);
  auto split_store = [&](char* row, int q, f32x4 v) __attribute__((always_inline)) {
    g_bf16x4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hi[k] = (__bf16)v[k];
      lo[k] = (__bf16)(v[k] - (float)hi[k]);
    }
    *reinterpret_cast<g_bf16x4*>(row + 8 * q) = hi;
    *reinterpret_cast<g_bf16x4*>(row + 2 * K + 8 * q) = lo;
  };
  f32x4 ar[NQ];
  auto load_a = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int idx = tid + 256 * i;
      int t = tile * TT_OUT - 1 + idx / QPR;               // rows outside [0, L) are masked when g is written
      t = t < 0 ? 0 : (t < L ? t : L - 1);
      ar[i] = *reinterpret_cast<const f32x4*>(xb + (size_t)t * K + 4 * (idx % QPR));
    }
  };
  load_a(tile0);
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int idx = tid + 256 * i;
    split_store(Ws + (idx / QPR) * ROW, idx % QPR, *reinterpret_cast<const f32x4*>(gw + (size_t)(idx / QPR) * K + 4 * (idx % QPR)));
  }
  for (int i = tid; i < 8 * K; i += 256) wls[i] = wl[i];
  float gbias[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) gbias[nt] = gb[nt * 32 + li];
  const char* Ap = As + (wave * 32 + li) * ROW + 16 * lh;
  const char* Bp = Ws + li * ROW + 16 * lh;
  const int ntile = tiles_per_clip - tile0 < TT_TPW ? tiles_per_clip - tile0 : TT_TPW;
  for (int it = 0; it < ntile; ++it) {
    const int tile = tile0 + it;
    const int tbase = tile * TT_OUT - 1;                   // clip row of tile row 0
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int idx = tid + 256 * i;
      split_store(As + (idx / QPR) * ROW, idx % QPR, ar[i]);
    }
    if (it + 1 < ntile) load_a(tile + 1);
    __syncthreads();                                       // A (and, the first time, W / wl) staged
    floatx16 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
      const g_bf16x8 ah = *reinterpret_cast<const g_bf16x8*>(Ap + 32 * s);
      const g_bf16x8 al = *reinterpret_cast<const g_bf16x8*>(Ap + 2 * K + 32 * s);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const g_bf16x8 bh = *reinterpret_cast<const g_bf16x8*>(Bp + nt * 32 * ROW + 32 * s);
        const g_bf16x8 bl = *reinterpret_cast<const g_bf16x8*>(Bp + nt * 32 * ROW + 2 * K + 32 * s);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[nt], 0, 0, 0);
      }
    }
    // GLU -> g rows in LDS (fp32, same 208-byte rows; this wave's own 32 rows, whose fragments it has just read)
    float* Gs = reinterpret_cast<float*>(As);
    constexpr int GLDF = ROW / 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int t = tbase + row;
      const bool in = t >= 0 && t < L;
      const float g0 = (acc[0][r] + gbias[0]) * gemm_sigmoid(acc[1][r] + gbias[1]);
      Gs[row * GLDF + li] = in ? g0 : 0.f;
      if (li < K - 32) {
        const float g1 = (acc[2][r] + gbias[2]) * gemm_sigmoid(acc[3][r] + gbias[3]);
        Gs[row * GLDF + 32 + li] = in ? g1 : 0.f;
      }
    }
    // (rows tid / 2 of a wave's 64 threads are that wave's own 32 rows: only the wave's own LDS stores are needed here)
    {
      const int row = tid >> 1, half = tid & 1;
      const float* g = Gs + row * GLDF;
      const float* w = wls + half * 4 * K;
      f32x4 p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
      for (int c = 0; c < K; c += 4) {
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(w + j * K + c);
          p[j] += gv[0] * wv[0] + gv[1] * wv[1] + gv[2] * wv[2] + gv[3] * wv[3];
        }
      }
      *reinterpret_cast<f32x4*>(Ps + row * 8 + 4 * half) = p;
    }
    __syncthreads();                                       // p of all 128 rows
    if (tid < 2 * TT_OUT) {
      const int i = 1 + (tid >> 1), j = (tid & 1) * 2;     // tile row of the group, first of its two samples
      const int t = tbase + i;                             // group index: 127 tile .. 127 tile + 126
      if (t < groups) {
        float2 o;
        o.x = bias + Ps[i * 8 + j] + Ps[(i - 1) * 8 + 4 + j];
        o.y = bias + Ps[i * 8 + j + 1] + Ps[(i - 1) * 8 + 5 + j];
        *reinterpret_cast<float2*>(yb + 4 * (size_t)t + j) = o;
      }
    }
    __syncthreads();                                       // p and g are free: the next tile's A may be staged
  }
}

// ---------------------------------------------------------------------------------- first encoder level in one launch
// Conv1d(1 -> C, k8, s4) + ReLU + Conv1d(C -> 2C, 1) + GLU of the first encoder level (model.py:66-75,303-307): x (B, Lin) -> h
// (B, Lout, C).  gemm_shortk_bf16x3_kernel<true, 48, 48> does the same with 128 x 64 tiles, i.e. TWO workgroups (the value /
// gate column pairs 0..31 and 32..47) each evaluate the first convolution for the same 128 rows; here a workgroup owns all 128
// packed columns: the first convolution (fp32 FMAs, bias then taps 0..7 -- conv1d_c1_kernel's order) and the bf16 split run once
// per row, W is staged once per workgroup and reused for TT_TPW tiles.
__global__ __launch_bounds__(256, 2) void c1_glu_kernel(const float* __restrict__ x, int Lin, int Lout, const float* __restrict__ w1,
                                                        const float* __restrict__ b1, const float* __restrict__ gw,
                                                        const float* __restrict__ gb, float* __restrict__ y, int tiles_per_clip) {
  constexpr int K = TT_K, ROW = TT_ROW, QPR = K / 4, NQ = 128 * QPR / 256;
  constexpr int NSLOT = 7;                                                      // rows per first-convolution thread: cr + 21 p
  constexpr bool ALLT = false;
  __shared__ __attribute__((aligned(16))) char As[128 * ROW];
  __shared__ __attribute__((aligned(16))) char Ws[128 * ROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y;
  const int tile0 = blockIdx.x * TT_TPW;
  const float* xb = x + (size_t)b * Lin;
  char* yb = reinterpret_cast<char*>(y + (size_t)b * Lout * K);
  auto split_store = [&](char* row, int q, f32x4 v) __attribute__((always_inline)) {
    g_bf16x4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hi[k] = (__bf16)v[k];
      lo[k] = (__bf16)(v[k] - (float)hi[k]);
    }
    *reinterpret_cast<g_bf16x4*>(row + 8 * q) = hi;
    *reinterpret_cast<g_bf16x4*>(row + 2 * K + 8 * q) = lo;
  };
  // first-convolution slot of a thread: channel quad cq, rows cr + 21 p of every tile (252 of the 256 threads); the 8 samples under
  // a row come straight from memory (two aligned float4, L1 hits for the 11 other quads of the row), one tile ahead
  const int cq = tid % QPR, cr = tid / QPR;
  f32x4 xr[NSLOT][2];
  auto load_x = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < NSLOT; ++p) {
      int m = tile * 128 + cr + 21 * p;
      m = m < Lout ? m : Lout - 1;                         // rows past Lout (and slots past row 127) are never used
      const float* px = xb + 4 * (size_t)m;
      xr[p][0] = *reinterpret_cast<const f32x4*>(px);
      xr[p][1] = *reinterpret_cast<const f32x4*>(px + 4);
    }
  };
  if (ALLT || tid < 252) load_x(tile0);
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int idx = tid + 256 * i;
    split_store(Ws + (idx / QPR) * ROW, idx % QPR, *reinterpret_cast<const f32x4*>(gw + (size_t)(idx / QPR) * K + 4 * (idx % QPR)));
  }
  f32x4 wq[9];                                               // the quad's 8 taps and bias
#pragma unroll
  for (int j = 0; j < 8; ++j) wq[j] = *reinterpret_cast<const f32x4*>(w1 + j * K + 4 * cq);
  wq[8] = *reinterpret_cast<const f32x4*>(b1 + 4 * cq);
  float gbias[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) gbias[nt] = gb[nt * 32 + li];
  const char* Ap = As + (wave * 32 + li) * ROW + 16 * lh;
  const char* Bp = Ws + li * ROW + 16 * lh;
  const int ntile = tiles_per_clip - tile0 < TT_TPW ? tiles_per_clip - tile0 : TT_TPW;
  for (int it = 0; it < ntile; ++it) {
    const int tile = tile0 + it;
    // A = relu(b1 + sum_j w1[j] * x[4 row + j]), bias then taps 0..7 (conv1d_c1_kernel's order), split into LDS
    if (ALLT || tid < 252) {
#pragma unroll
      for (int p = 0; p < NSLOT; ++p) {
        const int row = cr + 21 * p;
        if (row < 128) {
          f32x4 v = wq[8];
#ifndef MFPA_HEAD_PACKED_FMA
          // One v_fma_f32 per element, kept from being paired (the empty asm).  Written as `v += x * wq[j]` hipcc emits
          // v_pk_fma_f32 with op_sel:[0,1,0] (the LOW lane takes the HIGH half of the sample pair) for the odd samples, and with
          // TWO workgroups per CU -- i.e. while the SIMD's other wave runs MFMAs -- the low lane of those instructions sporadically
          // came out wrong: wrong rows of A, different from run to run, only at > 256 workgroups; bit-exact with one workgroup per
          // CU, with this form, and in the 128 x 64-tile GEMM form, whose loader only uses the op_sel_hi:[1,0,1] broadcast
          // (profiles/r02_pk_fma_op_sel.md; -DMFPA_HEAD_PACKED_FMA + tools/probes/head_kernel_two_wg_per_cu.py reproduce it).
#pragma unroll
          for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              float t = __builtin_fmaf(xr[p][j >> 2][j & 3], wq[j][k], v[k]);
              asm volatile("" : "+v"(t));
              v[k] = t;
            }
#elif MFPA_HEAD_PACKED_FMA == 2
          // packed FMAs WITHOUT operand selection: every sample is first materialised as an {x, x} pair (experiment)
          {
            typedef float hf2 __attribute__((ext_vector_type(2)));
            hf2 v01 = {v[0], v[1]}, v23 = {v[2], v[3]};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              hf2 xx = {xr[p][j >> 2][j & 3], xr[p][j >> 2][j & 3]};
              asm volatile("" : "+v"(xx));
              v01 += xx * hf2{wq[j][0], wq[j][1]};
              v23 += xx * hf2{wq[j][2], wq[j][3]};
            }
            v = f32x4{v01[0], v01[1], v23[0], v23[1]};
          }
#else
#pragma unroll
          for (int j = 0; j < 8; ++j) v += xr[p][j >> 2][j & 3] * wq[j];
#endif
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
          split_store(As + row * ROW, cq, v);
        }
      }
      if (it + 1 < ntile) load_x(tile + 1);
    }
    __syncthreads();                                       // A (and, the first time, W) staged
    floatx16 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
      const g_bf16x8 ah = *reinterpret_cast<const g_bf16x8*>(Ap + 32 * s);
      const g_bf16x8 al = *reinterpret_cast<const g_bf16x8*>(Ap + 2 * K + 32 * s);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const g_bf16x8 bh = *reinterpret_cast<const g_bf16x8*>(Bp + nt * 32 * ROW + 32 * s);
        const g_bf16x8 bl = *reinterpret_cast<const g_bf16x8*>(Bp + nt * 32 * ROW + 2 * K + 32 * s);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[nt], 0, 0, 0);
      }
    }
    const int mbase = tile * 128 + wave * 32 + 4 * lh;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mbase + (r & 3) + 8 * (r >> 2);
      if (m < Lout) {
        const unsigned o = ((unsigned)m * K + (unsigned)li) * 4u;
        st_f32(yb, o, (acc[0][r] + gbias[0]) * gemm_sigmoid(acc[1][r] + gbias[1]));
        if (li < K - 32) st_f32(yb, o + 128u, (acc[2][r] + gbias[2]) * gemm_sigmoid(acc[3][r] + gbias[3]));
      }
    }
    __syncthreads();                                       // every wave has read its A fragments: the next tile may be staged
  }
}
// ---------------------------------------------------------------------------------- small kernels
// mix / (floor + std), zero-padded to VL samples; std = unbiased std over time (model.py:293-301).
__global__ __launch_bounds__(256) void demucs_prep_kernel(const float* __restrict__ wav, int T, int VL, float floor_,
                                                          float* __restrict__ out, float* __restrict__ stdv) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* x = wav + (size_t)b * T;
  double s = 0, ss = 0;
  for (int i = tid; i < T; i += 256) { const double v = x[i]; s += v; ss += v * v; }
  __shared__ double sh[2][256];
  sh[0][tid] = s; sh[1][tid] = ss;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { sh[0][tid] += sh[0][tid + o]; sh[1][tid] += sh[1][tid + o]; }
    __syncthreads();
  }
  const double mean = sh[0][0] / T;
  double var = (sh[1][0] - T * mean * mean) / (T - 1);
  if (var < 0) var = 0;
  const float sd = (float)sqrt(var);
  if (tid == 0) stdv[b] = sd;
  const float inv = sd + floor_;
  float* o = out + (size_t)b * VL;
  for (int i = tid; i < VL; i += 256) o[i] = i < T ? x[i] / inv : 0.f;
}

// Sinc x2 resamplers: 112-tap FIRs.  A workgroup produces RS_OUT consecutive filter outputs from a window of the input
// staged once in LDS; a thread owns 4 adjacent outputs and slides a float4 window over the taps (7 LDS float4 reads per
// 4 taps x 4 outputs instead of 16 scalar loads).
constexpr int RS_OUT = 1024, RS_TAPS = 112;

__device__ __forceinline__ void rs_fir4(const float* __restrict__ w, const float* __restrict__ kk, f32x4& acc) {
  f32x4 lo = *reinterpret_cast<const f32x4*>(w);
#pragma unroll 4
  for (int k = 0; k < RS_TAPS; k += 4) {
    const f32x4 hi = *reinterpret_cast<const f32x4*>(w + k + 4);
    const f32x4 c = *reinterpret_cast<const f32x4*>(kk + k);
    acc[0] += c[0] * lo[0] + c[1] * lo[1] + c[2] * lo[2] + c[3] * lo[3];
    acc[1] += c[0] * lo[1] + c[1] * lo[2] + c[2] * lo[3] + c[3] * hi[0];
    acc[2] += c[0] * lo[2] + c[1] * lo[3] + c[2] * hi[0] + c[3] * hi[1];
    acc[3] += c[0] * lo[3] + c[1] * hi[0] + c[2] * hi[1] + c[3] * hi[2];
    lo = hi;
  }
}

// upsample2 (model.py:41-53): y[2i] = x[i], y[2i+1] = sum_k x[i + k - 55] ker[k], k < 112 (zero beyond the ends).
__global__ MFPA_NO_PK_F32 __launch_bounds__(256) void upsample2_kernel(const float* __restrict__ x, int T, const float* __restrict__ ker,
                                                        float* __restrict__ y) {
  __shared__ __attribute__((aligned(16))) float win[RS_OUT + RS_TAPS + 8];
  __shared__ __attribute__((aligned(16))) float kk[RS_TAPS];
  const int b = blockIdx.y, tid = threadIdx.x, i0 = blockIdx.x * RS_OUT;
  const float* xb = x + (size_t)b * T;
  float* yb = y + (size_t)b * 2 * T;
  if (tid < RS_TAPS) kk[tid] = ker[tid];
  for (int j = tid; j < RS_OUT + RS_TAPS + 4; j += 256) {        // win[j] = x[i0 - 55 + j]
    const int s = i0 - 55 + j;
    win[j] = (s >= 0 && s < T) ? xb[s] : 0.f;
  }
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  rs_fir4(win + 4 * tid, kk, acc);
  const int i = i0 + 4 * tid;
  if (i + 3 < T) {
    const f32x4 xv = {win[4 * tid + 55], win[4 * tid + 56], win[4 * tid + 57], win[4 * tid + 58]};
    *reinterpret_cast<f32x4*>(yb + 2 * (size_t)i) = f32x4{xv[0], acc[0], xv[1], acc[1]};
    *reinterpret_cast<f32x4*>(yb + 2 * (size_t)i + 4) = f32x4{xv[2], acc[2], xv[3], acc[3]};
  } else {
#pragma unroll
    for (int o = 0; o < 4; ++o)
      if (i + o < T) { yb[2 * (size_t)(i + o)] = win[4 * tid + o + 55]; yb[2 * (size_t)(i + o) + 1] = acc[o]; }
  }
}

// downsample2 (model.py:69-88): out[i] = 0.5 * (x[2i] + sum_k xodd[i + k - 56] ker[k]), xodd[j] = x[2j+1] (0 beyond the end).
__global__ MFPA_NO_PK_F32 __launch_bounds__(256) void downsample2_kernel(const float* __restrict__ x, int T, const float* __restrict__ ker,
                                                          float* __restrict__ y, int To, const float* __restrict__ scale,
                                                          int Tkeep) {
  __shared__ __attribute__((aligned(16))) float wodd[RS_OUT + RS_TAPS + 8];
  __shared__ __attribute__((aligned(16))) float kk[RS_TAPS];
  const int b = blockIdx.y, tid = threadIdx.x, i0 = blockIdx.x * RS_OUT;
  const float* xb = x + (size_t)b * T;
  const int Th = (T + 1) / 2;                       // length of xeven / xodd after the odd-length zero pad
  const float sc = scale ? scale[b] : 1.f;
  const int nout = Tkeep > 0 ? Tkeep : Th;
  float* yb = y + (size_t)b * To;
  if (tid < RS_TAPS) kk[tid] = ker[tid];
  for (int j = tid; j < RS_OUT + RS_TAPS + 4; j += 256) {        // wodd[j] = xodd[i0 - 56 + j]
    const int s = i0 - 56 + j;
    wodd[j] = (s >= 0 && s < Th && 2 * s + 1 < T) ? xb[2 * s + 1] : 0.f;
  }
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  rs_fir4(wodd + 4 * tid, kk, acc);
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const int i = i0 + 4 * tid + o;
    if (i < nout) yb[i] = sc * (0.5f * (xb[2 * (size_t)i] + acc[o]));
  }
}

// First encoder conv: Conv1d(1 -> C, k=8, s=4) + ReLU on (B, Lin) -> (B, Lout, C).  w [8][C] (tap-major), bias [C].
__global__ __launch_bounds__(256) void conv1d_c1_kernel(const float* __restrict__ x, int Lin, int Lout, int C,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ y, int relu) {
  const int b = blockIdx.y, C4 = C / 4;
  const float* xb = x + (size_t)b * Lin;
  float* yb = y + (size_t)b * Lout * C;
  const long long total = (long long)Lout * C4;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int cq = (int)(e % C4);
    const int t = (int)(e / C4);
    const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + 4 * cq) : f32x4{0.f, 0.f, 0.f, 0.f};
    mfpa_f32x2 a01 = {bv[0], bv[1]}, a23 = {bv[2], bv[3]};       // bias, then taps 0..7: one FMA chain per channel
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const mfpa_f32x2 v = mfpa_bcast2(xb[4 * t + j]);            // packed FMAs without operand selection (mfpa_common.h)
      const f32x4 ww = *reinterpret_cast<const f32x4*>(w + j * C + 4 * cq);
      a01 += v * mfpa_f32x2{ww[0], ww[1]};
      a23 += v * mfpa_f32x2{ww[2], ww[3]};
    }
    f32x4 acc = {a01[0], a01[1], a23[0], a23[1]};
    if (relu) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = acc[k] > 0.f ? acc[k] : 0.f;
    }
    *reinterpret_cast<f32x4*>(yb + (size_t)t * C + 4 * cq) = acc;
  }
}

// Last decoder layer: ConvTranspose1d(C -> 1, k=8, s=4) on the zero-padded GLU output P (B, L+2, C):
// out[4t + j] = bias + sum_c P[t+1][c] w[c][j] + P[t][c] w[c][j+4],  t = 0..L, j = 0..3.   w [8][C] (tap-major).
__global__ __launch_bounds__(256) void convT1d_c1_kernel(const float* __restrict__ P, int L, int C,
                                                         const float* __restrict__ w, float bias, const float* __restrict__ biasp,
                                                         float* __restrict__ y) {
  const int b = blockIdx.y;
  if (biasp) bias = biasp[0];
  const float* Pb = P + (size_t)b * (L + 2) * C;
  float* yb = y + (size_t)b * 4 * (L + 1);
  const int total = 4 * (L + 1);
  for (int p = blockIdx.x * 256 + threadIdx.x; p < total; p += gridDim.x * 256) {
    const int t = p >> 2, j = p & 3;
    const float* cur = Pb + (size_t)(t + 1) * C;
    const float* prev = Pb + (size_t)t * C;
    float acc = bias;
    for (int c = 0; c < C; c += 4) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(cur + c), a1 = *reinterpret_cast<const f32x4*>(prev + c);
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + j * C + c), w1 = *reinterpret_cast<const f32x4*>(w + (j + 4) * C + c);
      acc += a0[0] * w0[0] + a0[1] * w0[1] + a0[2] * w0[2] + a0[3] * w0[3];
      acc += a1[0] * w1[0] + a1[1] * w1[1] + a1[2] * w1[2] + a1[3] * w1[3];
    }
    yb[p] = acc;
  }
}

// LSTM cell (gate order i, f, g, o): c = sig(f) c + sig(i) tanh(g); h = sig(o) tanh(c).
// gates (B, 4H); c (B, H) in/out; h written to hseq[b*ldh + n] (+ optional addend for the decoder's first skip).
__global__ __launch_bounds__(256) void lstm_cell_kernel(const float* __restrict__ gates, long long ldg, float* __restrict__ c, int B, int H,
                                                        float* __restrict__ hout, long long ldh,
                                                        float* __restrict__ hsum, const float* __restrict__ addend, long long ldadd) {
  const int total = B * H;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int b = e / H, n = e % H;
    const float* g = gates + (size_t)b * ldg;
    const float gi = g[n], gf = g[H + n], gg = g[2 * H + n], go = g[3 * H + n];
    const float si = 1.f / (1.f + expf(-gi)), sf = 1.f / (1.f + expf(-gf)), so = 1.f / (1.f + expf(-go));
    const float cn = sf * c[e] + si * tanhf(gg);
    c[e] = cn;
    const float h = so * tanhf(cn);
    hout[(size_t)b * ldh + n] = h;
    if (hsum) hsum[(size_t)b * ldh + n] = h + addend[(size_t)b * ldadd + n];
  }
}

// ---------------------------------------------------------------------------------- fused LSTM time step
// One launch per time step: gates = h[t-1] W_hh^T (+ xp[t], the input projection incl. both biases) AND the cell update,
// so the recurrence costs one short kernel per step instead of a GEMM + a cell kernel.
//   * a workgroup owns 64 clips x 16 hidden units = 64 x 64 gate columns [i16 | f16 | g16 | o16] (W_hh rows regrouped
//     on the host), K = H walked in chunks of 128 with the operands double-buffered in LDS and the next chunk's global
//     loads in flight during the MFMA block;
//   * bf16x3 products (operands split hi + lo while they are staged), fp32 accumulate, like conv_mfma_kernel<PREC 1>;
//   * workgroup id -> (XCD, slot): the 6 unit groups an XCD owns keep their W_hh slices (1.2 MB) in that XCD's L2 for
//     all 248 steps.
typedef __bf16 l_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 l_bf16x4 __attribute__((ext_vector_type(4)));
constexpr int LKC = 128;                 // K chunk
constexpr int LROW = 4 * LKC + 16;       // LDS row bytes: [128 hi | 128 lo | pad]
constexpr int LU = 16;                   // hidden units per workgroup (64 gate columns)

constexpr int LTHREADS = 512;   // 8 waves: (clip half) x (gate-column half) x (k-step half of every chunk)
#ifndef MFPA_LSTM_PF
#define MFPA_LSTM_PF 3
#endif
constexpr int LPF = MFPA_LSTM_PF;  // chunks of global loads in flight per thread (register ring)

// MT = 32-clip tiles per workgroup: 2 (64 clips; 8 waves = 2 clip halves x 2 gate-column halves x 2 k-step halves) or 1 (32
// clips; 2 gate-column halves x 4 k-step quarters).  The step streams h[t-1] and its W_hh slice from memory every launch and is
// bound by what the CUs that own it can pull, so small batches use the 32-clip form: twice the workgroups, half of h[t-1] each.
template <int MT>
__global__ __launch_bounds__(LTHREADS, 1) void lstm_step_kernel(const float* __restrict__ hprev, long long ldhp,
                                                           const float* __restrict__ whh, const float* xp,
                                                           long long ldxp, const float* cin, long long ldci, float* cout,
                                                           long long ldco, int B, int H,
                                                           float* __restrict__ hout, long long ldh, float* __restrict__ hsum,
                                                           const float* __restrict__ addend, long long ldadd, int mtiles,
                                                           float* gsave, long long ldgs) {
  constexpr int BMT = 32 * MT;               // clips per workgroup
  constexpr int WK = 4 / MT;                 // k-step groups
  constexpr int KS = 8 / WK;                 // k-steps of 16 per wave and 128-wide chunk
  extern __shared__ __attribute__((aligned(16))) char lsm[];
  char* As = lsm;                            // [2][BMT][LROW]
  char* Bs = lsm + 2 * BMT * LROW;           // [2][64][LROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave % MT, wn = (wave / MT) & 1, wk = wave / (2 * MT);
  // XCD-aware decode: consecutive workgroup ids go round-robin over the 8 XCDs
  const int ngroups = H / LU;
  int grp, mt;
  {
    const int id = blockIdx.x, total = ngroups * mtiles;
    const int per_xcd = (total + 7) / 8;
    const int lin = (id % 8) * per_xcd + id / 8;        // position in (group-major, m-tile-minor) order
    if (lin >= total) return;                           // uniform per workgroup (before any barrier)
    grp = lin / mtiles; mt = lin % mtiles;
  }
  const int m0 = mt * BMT;
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  if (hprev != nullptr) {
    const float* Wg = whh + (size_t)grp * 64 * H;
    const int nk = H / LKC;
    constexpr int FA = BMT * (LKC / 4) / LTHREADS;       // float4 per thread per chunk: h rows (2 MT)
    constexpr int FB = 64 * (LKC / 4) / LTHREADS;        // ... and W_hh rows (4)
    constexpr int RSTEP = LTHREADS / (LKC / 4);          // 16 rows per pass
    // register ring of LPF chunks of global loads
    f32x4 ar[LPF][FA], br[LPF][FB];
    const int q = tid % (LKC / 4), r0 = tid / (LKC / 4); // column quad, first row; rows r0 + 16 i
    auto load = [&](int kc, f32x4 (&a4)[FA], f32x4 (&b4)[FB]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < FA; ++i) {
        const int m = m0 + r0 + RSTEP * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < B) v = *reinterpret_cast<const f32x4*>(hprev + (size_t)m * ldhp + kc * LKC + 4 * q);
        a4[i] = v;
      }
#pragma unroll
      for (int i = 0; i < FB; ++i) b4[i] = *reinterpret_cast<const f32x4*>(Wg + (size_t)(r0 + RSTEP * i) * H + kc * LKC + 4 * q);
    };
    auto split_store = [&](char* row, f32x4 v) __attribute__((always_inline)) {
      l_bf16x4 hi, lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        hi[k] = (__bf16)v[k];
        lo[k] = (__bf16)(v[k] - (float)hi[k]);
      }
      *reinterpret_cast<l_bf16x4*>(row + 8 * q) = hi;
      *reinterpret_cast<l_bf16x4*>(row + 2 * LKC + 8 * q) = lo;
    };
    auto store = [&](int buf, f32x4 (&a4)[FA], f32x4 (&b4)[FB]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < FA; ++i) split_store(As + (buf * BMT + r0 + RSTEP * i) * LROW, a4[i]);
#pragma unroll
      for (int i = 0; i < FB; ++i) split_store(Bs + (buf * 64 + r0 + RSTEP * i) * LROW, b4[i]);
    };
#pragma unroll
    for (int j = 0; j < LPF; ++j)
      if (j < nk) load(j, ar[j], br[j]);
    for (int base = 0; base < nk; base += LPF) {
#pragma unroll
      for (int j = 0; j < LPF; ++j) {
        const int kc = base + j;
        if (kc < nk) {                                   // uniform over the workgroup
          const int buf = kc & 1;
          store(buf, ar[j], br[j]);                      // buffer (kc & 1) was last read for chunk kc - 2, before the previous barrier
          if (kc + LPF < nk) load(kc + LPF, ar[j], br[j]);
          __syncthreads();
          const char* Ap = As + (buf * BMT + wm * 32 + li) * LROW + 16 * lh;
          const char* Bp = Bs + (buf * 64 + wn * 32 + li) * LROW + 16 * lh;
#pragma unroll
          for (int s = KS * wk; s < KS * wk + KS; ++s) {
            const l_bf16x8 ah = *reinterpret_cast<const l_bf16x8*>(Ap + 32 * s);
            const l_bf16x8 al = *reinterpret_cast<const l_bf16x8*>(Ap + 2 * LKC + 32 * s);
            const l_bf16x8 bh = *reinterpret_cast<const l_bf16x8*>(Bp + 32 * s);
            const l_bf16x8 bl = *reinterpret_cast<const l_bf16x8*>(Bp + 2 * LKC + 32 * s);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();                                     // the gate slabs below reuse the operand buffers
  }
  // the WK partial gate tiles -> LDS slabs [WK][BMT clips][64 + 4], summed by the cell threads
  float* G = reinterpret_cast<float*>(lsm);
  constexpr int GLDW = 68;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    G[(wk * BMT + m) * GLDW + wn * 32 + li] = acc[r];
  }
  __syncthreads();
  const int clip = tid >> 2, uq = tid & 3;
  const int m = m0 + clip;
  if (clip < BMT && m < B) {
    const int u0 = grp * LU + 4 * uq;
    const float* xr = xp + (size_t)m * ldxp;
    const f32x4 xi = *reinterpret_cast<const f32x4*>(xr + u0), xf = *reinterpret_cast<const f32x4*>(xr + H + u0);
    const f32x4 xg = *reinterpret_cast<const f32x4*>(xr + 2 * H + u0), xo = *reinterpret_cast<const f32x4*>(xr + 3 * H + u0);
    const f32x4 cp = cin ? *reinterpret_cast<const f32x4*>(cin + (size_t)m * ldci + u0) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 gi4 = xi, gf4 = xf, gg4 = xg, go4 = xo;
#pragma unroll
    for (int w = 0; w < WK; ++w) {
      const float* g = G + (w * BMT + clip) * GLDW + 4 * uq;
      gi4 += *reinterpret_cast<const f32x4*>(g);
      gf4 += *reinterpret_cast<const f32x4*>(g + 16);
      gg4 += *reinterpret_cast<const f32x4*>(g + 32);
      go4 += *reinterpret_cast<const f32x4*>(g + 48);
    }
    f32x4 cn, hn, vi, vf, vg, vo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float si = 1.f / (1.f + expf(-gi4[k])), sf = 1.f / (1.f + expf(-gf4[k])), so = 1.f / (1.f + expf(-go4[k]));
      const float tg = tanhf(gg4[k]);
      cn[k] = sf * cp[k] + si * tg;
      hn[k] = so * tanhf(cn[k]);
      vi[k] = si; vf[k] = sf; vg[k] = tg; vo[k] = so;
    }
    if (gsave) {                               // training: the gate activations the backward step needs (may alias xp)
      float* gr = gsave + (size_t)m * ldgs;
      *reinterpret_cast<f32x4*>(gr + u0) = vi;
      *reinterpret_cast<f32x4*>(gr + H + u0) = vf;
      *reinterpret_cast<f32x4*>(gr + 2 * H + u0) = vg;
      *reinterpret_cast<f32x4*>(gr + 3 * H + u0) = vo;
    }
    *reinterpret_cast<f32x4*>(cout + (size_t)m * ldco + u0) = cn;
    *reinterpret_cast<f32x4*>(hout + (size_t)m * ldh + u0) = hn;
    if (hsum) {
      const f32x4 ad = *reinterpret_cast<const f32x4*>(addend + (size_t)m * ldadd + u0);
      *reinterpret_cast<f32x4*>(hsum + (size_t)m * ldh + u0) = hn + ad;
    }
  }
}

// 32-clip tiles while they still leave the chip under-filled (<= 256 workgroups), 64-clip tiles for large batches
static int lstm_launch(const float* hprev, long long ldhp, const float* whh_grouped, const float* xp, long long ldxp, const float* cin,
                       long long ldci, float* cout, long long ldco, int B, int H, float* hout, long long ldh, float* hsum,
                       const float* addend, long long ldadd, float* gsave, long long ldgs, void* stream) {
  static const int force = MFPA_EXP_ENV("MFPA_LSTM_MT", 0);
  const int groups = H / LU;
  int MT = ((long long)groups * ((B + 31) / 32) <= 256) ? 1 : 2;
  if (force == 1 || force == 2) MT = force;
  const int mtiles = (B + 32 * MT - 1) / (32 * MT);
  const long long total = (long long)groups * mtiles;
  if (total > 0x7fffff) return MFPA_EINVAL;
  const unsigned grid = (unsigned)(((total + 7) / 8) * 8);
  const size_t lds = (size_t)2 * (32 * MT + 64) * LROW;
  if (MT == 1)
    hipLaunchKernelGGL(lstm_step_kernel<1>, dim3(grid), dim3(LTHREADS), lds, mfpa_stream(stream), hprev, ldhp, whh_grouped, xp, ldxp, cin,
                       ldci, cout, ldco, B, H, hout, ldh, hsum, addend, ldadd, mtiles, gsave, ldgs);
  else
    hipLaunchKernelGGL(lstm_step_kernel<2>, dim3(grid), dim3(LTHREADS), lds, mfpa_stream(stream), hprev, ldhp, whh_grouped, xp, ldxp, cin,
                       ldci, cout, ldco, B, H, hout, ldh, hsum, addend, ldadd, mtiles, gsave, ldgs);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}


// ---------------------------------------------------------------------------------- persistent LSTM layer
// The whole time range of one layer in ONE launch.  The per-step kernel above re-reads its W_hh slice (and splits it into
// bf16 hi / lo) in every one of the 248 steps and pays a launch per step; here
//   * a workgroup owns 64 clips x 16 hidden units (64 gate columns) for all steps.  Its W_hh slice lives in REGISTERS, already
//     split: wave w holds the MFMA B-fragments of K range [w H/8, (w+1) H/8) (H = 768: 2 column tiles x 6 k-steps x (hi, lo)
//     = 96 VGPRs), so W_hh is read from memory once per launch;
//   * h[t-1] is exchanged between workgroups through a ping-pong buffer that already holds the split form
//     ([32 bf16 hi | 32 bf16 lo] per 32 units, written once by the cell that produced it, not by each of its 48 readers);
//     a wave reads its A-fragments of it straight from global memory (L2) in MFMA layout: no LDS staging of operands;
//   * the 8 partial 64 x 64 gate tiles (one per K range) are summed through LDS by the cell threads, which keep c in registers;
//   * the workgroups of one 64-clip slab meet at a counter in device memory after every step (release / acquire at agent
//     scope: the other XCDs' L2s see the new h).  Every wait is BOUNDED: after LSTM_SPIN_LIMIT polls a workgroup raises the
//     error word and from then on nobody waits, so the grid always drains; the host reads the word later (mfpa_lstm_seq_error).
// The grid must be co-resident (one workgroup per CU: 136 KB of LDS): the host checks slabs x groups <= CUs, else the
// per-step kernels run.
constexpr int QW = 8;                        // waves = K ranges
constexpr int QGLD = 68;                     // floats per clip row of a partial gate slab (64 + pad)
constexpr unsigned LSTM_SPIN_LIMIT = 1u << 22;
constexpr int LSTM_SYNC_WORDS = 1024;        // head of the work buffer: counter of slab s at word 16 s, error word at 512
constexpr int LSTM_ERR_WORD = 512;

struct LstmSeqArgs {
  const float* whh;       // grouped W_hh (4H, H)
  float* xp;              // (B, Tn, 4H) projections (+ biases); training: overwritten with the gate activations
  float* hseq;            // (B, Tn, H)
  float* cseq;            // training: (B, Tn, H)
  float* cstate;          // inference: (B, H), read at t0 > 0, written at the end
  float* xsum;            // optional (B, Tn, H): h + skip
  const float* skip;
  unsigned* sync;         // LSTM_SYNC_WORDS words
  char* hsplit;           // [2][B][H * 4 bytes]
  int B, Tn, H, t0, t1, train, nslab, ngroups;
  int dbg;                // -DMFPA_EXPERIMENTS builds only (MFPA_LSTM_DBG, timing experiments: results are wrong): 1 no MFMAs, 2 no loads of h, 4 no waits, 8 no s_sleep in the poll
};

// COH 0: the waiting thread invalidates L1 / L2 once per step and h is read with ordinary (cached) loads; 1: no invalidate, h is read
// with agent-scope (sc1) buffer loads that do not trust the local caches.
// MS = 32-clip MFMA row tiles per workgroup (slab = 32 MS clips): 2 for large batches; 1 while that still leaves half the chip free --
// twice the workgroups, each reading half as much of h per step (the step is bound by what a CU can pull, see the timing experiments).
template <int N> struct LFV { float v[N]; __device__ __forceinline__ float& operator[](int i) { return v[i]; } __device__ __forceinline__ const float& operator[](int i) const { return v[i]; } };
template <int N> __device__ __forceinline__ LFV<N> lfv_load(const float* p) { LFV<N> r;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = p[i];
  return r; }
template <int N> __device__ __forceinline__ void lfv_store(float* p, const LFV<N>& x) {
#pragma unroll
  for (int i = 0; i < N; ++i) p[i] = x.v[i]; }

#ifndef MFPA_LSTM_WIDE_PUT
#define MFPA_LSTM_WIDE_PUT 1      // round 6: a step's new h leaves the workgroup as 16-byte write-through stores (gathered through LDS) instead of one
#endif                            // 4-byte (2-byte) agent-scope store per cell thread and half: narrow sc1 stores are one fabric write each (A/B builds: 0)
#ifndef MFPA_LSTM_FAST_CELL
#define MFPA_LSTM_FAST_CELL 1     // round 6: the cell's sigmoid / tanh on v_exp_f32 + v_rcp_f32 (absolute error ~1e-7) instead of the library expf / tanhf (A/B builds: 0)
#endif
__device__ __forceinline__ float lstm_sig(float x) {
#if MFPA_LSTM_FAST_CELL
  return __builtin_amdgcn_rcpf(1.f + __expf(-x));
#else
  return 1.f / (1.f + expf(-x));
#endif
}
__device__ __forceinline__ float lstm_tanh(float x) {
#if MFPA_LSTM_FAST_CELL
  return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x));       // e^{2x} -> inf: 1; -> 0: -1
#else
  return tanhf(x);
#endif
}

template <int KS, int COH, int MS>          // k-steps of 16 per wave: H = 128 KS
__global__ __launch_bounds__(64 * QW, 1) void lstm_seq_kernel(LstmSeqArgs a) {
  constexpr int SLAB = 32 * MS, UPT = MS, TPC = 16 / UPT;   // clips per workgroup; hidden units per cell thread; cell threads per clip
  typedef LFV<UPT> fv;
  extern __shared__ __attribute__((aligned(16))) char lsm[];
  float* G = reinterpret_cast<float*>(lsm);                 // [QW][SLAB][QGLD]
#if MFPA_LSTM_WIDE_PUT
  char* const PS = lsm + (size_t)QW * SLAB * QGLD * sizeof(float);   // [SLAB][64 B]: the step's new h, split, on its way out
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int H = a.H;
  int slab, grp;
  {
    const int id = blockIdx.x, total = a.nslab * a.ngroups;
    const int per_xcd = (total + 7) / 8;
    const int lin = (id % 8) * per_xcd + id / 8;            // slab-major: the workgroups of a slab sit in as few XCDs as possible
    if (lin >= total) return;                               // padding workgroups: they are not counted at the barrier
    slab = lin / a.ngroups; grp = lin % a.ngroups;
  }
  const int m0 = slab * SLAB;
  unsigned* cnt = a.sync + 16 * slab;
  unsigned* err = a.sync + LSTM_ERR_WORD;
  const unsigned members = (unsigned)a.ngroups;
  const size_t rowb = (size_t)H * 4;                        // bytes per clip row of the split exchange buffer
  const size_t bufb = (size_t)a.B * rowb;

  // ---- W_hh fragments, split once
  l_bf16x8 wh[2][KS], wl[2][KS];
  {
    const float* Wg = a.whh + (size_t)grp * 64 * H;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const float* p = Wg + (size_t)(nt * 32 + li) * H + (wave * KS + s) * 16 + 8 * lh;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const __bf16 h0 = (__bf16)v0[k], h1 = (__bf16)v1[k];
          wh[nt][s][k] = h0; wh[nt][s][4 + k] = h1;
          wl[nt][s][k] = (__bf16)(v0[k] - (float)h0); wl[nt][s][4 + k] = (__bf16)(v1[k] - (float)h1);
        }
      }
  }
  // ---- cell threads: clip tid / TPC, hidden units u0 .. u0 + UPT - 1
  const int clip = tid / TPC, up = tid % TPC;
  const int m = m0 + clip;
  const bool live = m < a.B;
  const int u0 = grp * LU + UPT * up;
  const size_t ldh = (size_t)a.Tn * H, ldx = (size_t)a.Tn * 4 * H;
  const size_t split_off = (size_t)(live ? m : 0) * rowb + (size_t)(u0 >> 5) * 128 + (size_t)(u0 & 31) * 2;
  auto put_split = [&](char* buf, const fv& h) __attribute__((always_inline)) {
    // agent-scope stores (sc1: written through to memory), so the release below needs no L2 write-back
    if (UPT == 2) {
      const __bf16 h0 = (__bf16)h[0], h1 = (__bf16)h[UPT - 1];
      const __bf16 l0 = (__bf16)(h[0] - (float)h0), l1 = (__bf16)(h[UPT - 1] - (float)h1);
      const unsigned hi = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
      const unsigned lo = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
      __hip_atomic_store(reinterpret_cast<unsigned*>(buf + split_off), hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(reinterpret_cast<unsigned*>(buf + split_off + 64), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const __bf16 h0 = (__bf16)h[0];
      const __bf16 l0 = (__bf16)(h[0] - (float)h0);
      __hip_atomic_store(reinterpret_cast<unsigned short*>(buf + split_off), __builtin_bit_cast(unsigned short, h0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(reinterpret_cast<unsigned short*>(buf + split_off + 64), __builtin_bit_cast(unsigned short, l0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  fv c;
#pragma unroll
  for (int k = 0; k < UPT; ++k) c[k] = 0.f;
  if (live) {
    fv hp;
#pragma unroll
    for (int k = 0; k < UPT; ++k) hp[k] = 0.f;
    if (a.t0 > 0) {
      hp = lfv_load<UPT>(a.hseq + (size_t)m * ldh + (size_t)(a.t0 - 1) * H + u0);
      c = a.train ? lfv_load<UPT>(a.cseq + (size_t)m * ldh + (size_t)(a.t0 - 1) * H + u0) : lfv_load<UPT>(a.cstate + (size_t)m * H + u0);
    }
    put_split(a.hsplit + (size_t)((a.t0 + 1) & 1) * bufb, hp);          // h[t] lives in buffer t & 1
  }
  // ---- slab barrier: arrive after the stores above, wait until all `members` workgroups of the slab have arrived `round` times
  bool dead = false;
  auto arrive = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's write-through stores of h have been acknowledged
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto wait = [&](unsigned round) __attribute__((always_inline)) {
    if (tid == 0) {
      if (!dead) {
        const unsigned target = round * members;
        unsigned n = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          if ((++n & 63u) == 0u &&
              (n > LSTM_SPIN_LIMIT || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dead = true;
            break;
          }
          if (!MFPA_EXP_FLAG(a.dbg, 8)) __builtin_amdgcn_s_sleep(1);
        }
      }
      if (!COH) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // one invalidate per step: the loads of h below go through L1 / L2
    }
    __syncthreads();
  };
  arrive();

  // A-fragment rows of this lane (clamped: rows past B compute garbage that is never stored)
  size_t arow[MS];
#pragma unroll
  for (int mt = 0; mt < MS; ++mt) {
    int r = m0 + mt * 32 + li;
    r = r < a.B ? r : a.B - 1;
    arow[mt] = (size_t)r * rowb + (size_t)wave * KS * 64 + 16 * lh;      // k = (wave KS + s) 16 + 8 lh -> chunk k / 32, 2 (k % 32)
  }
  constexpr int PF = KS < 3 ? KS : 3;                      // k-steps of A loads in flight
  const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(a.hsplit, 0, (int)(2 * bufb), 0x00020000);
  for (int t = a.t0; t < a.t1; ++t) {
    // the projections do not depend on h: fetch them before the wait
    fv xg[4], ad;
#pragma unroll
    for (int k = 0; k < UPT; ++k) { xg[0][k] = xg[1][k] = xg[2][k] = xg[3][k] = 0.f; ad[k] = 0.f; }
    if (live) {
      const float* xr = a.xp + (size_t)m * ldx + (size_t)t * 4 * H + u0;
#pragma unroll
      for (int g = 0; g < 4; ++g) xg[g] = lfv_load<UPT>(xr + g * H);
      if (a.xsum) ad = lfv_load<UPT>(a.skip + (size_t)m * ldh + (size_t)t * H + u0);
    }
    if (!MFPA_EXP_FLAG(a.dbg, 4)) wait((unsigned)(t - a.t0 + 1));
    const char* hp = a.hsplit + (size_t)((t + 1) & 1) * bufb;
    floatx16 acc[MS][2];
#pragma unroll
    for (int i = 0; i < 2 * MS; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i >> 1][i & 1][r] = 0.f;
    l_bf16x8 fa[PF][MS][2];
    auto issue = [&](int s, l_bf16x8 (&f)[MS][2]) __attribute__((always_inline)) {
#pragma unroll
      for (int mt = 0; mt < MS; ++mt) {
        const size_t off = arow[mt] + (size_t)(s >> 1) * 128 + (size_t)(s & 1) * 32;
        if (MFPA_EXP_FLAG(a.dbg, 2)) {
          f[mt][0] = wh[0][0]; f[mt][1] = wl[0][0];
        } else if (COH) {
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          const unsigned o = (unsigned)(((t + 1) & 1) * bufb + off);
          f[mt][0] = __builtin_bit_cast(l_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(hrsrc, o, 0, 16));
          f[mt][1] = __builtin_bit_cast(l_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(hrsrc, o + 64, 0, 16));
        } else {
          f[mt][0] = *reinterpret_cast<const l_bf16x8*>(hp + off);
          f[mt][1] = *reinterpret_cast<const l_bf16x8*>(hp + off + 64);
        }
      }
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) issue(s, fa[s]);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int mt = 0; mt < MS; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          if (MFPA_EXP_FLAG(a.dbg, 1)) { acc[mt][nt][0] += (float)fa[s % PF][mt][0][0] + (float)fa[s % PF][mt][1][0]; continue; }
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s % PF][mt][1], wh[nt][s], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s % PF][mt][0], wl[nt][s], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s % PF][mt][0], wh[nt][s], acc[mt][nt], 0, 0, 0);
        }
      if (s + PF < KS) issue(s + PF, fa[s % PF]);
    }
    // partial gate tiles -> LDS
#pragma unroll
    for (int mt = 0; mt < MS; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          G[(wave * SLAB + row) * QGLD + nt * 32 + li] = acc[mt][nt][r];
        }
    __syncthreads();
    if (live) {
      fv gs[4] = {xg[0], xg[1], xg[2], xg[3]};
#pragma unroll
      for (int w = 0; w < QW; ++w) {
        const float* g = G + (w * SLAB + clip) * QGLD + UPT * up;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int k = 0; k < UPT; ++k) gs[q][k] += g[16 * q + k];
      }
      fv hn, vi, vf, vg, vo, hs;
#pragma unroll
      for (int k = 0; k < UPT; ++k) {
        const float si = lstm_sig(gs[0][k]), sf = lstm_sig(gs[1][k]), so = lstm_sig(gs[3][k]);
        const float tg = lstm_tanh(gs[2][k]);
        c[k] = sf * c[k] + si * tg;
        hn[k] = so * lstm_tanh(c[k]);
        hs[k] = hn[k] + ad[k];
        vi[k] = si; vf[k] = sf; vg[k] = tg; vo[k] = so;
      }
#if MFPA_LSTM_WIDE_PUT
      {   // this thread's hi / lo halves into the workgroup's staging rows: [clip][16 units x bf16 hi | 16 units x bf16 lo]
        char* ps = PS + clip * 64 + UPT * up * 2;
        if (UPT == 2) {
          const __bf16 h0 = (__bf16)hn[0], h1 = (__bf16)hn[UPT - 1];
          const __bf16 l0 = (__bf16)(hn[0] - (float)h0), l1 = (__bf16)(hn[UPT - 1] - (float)h1);
          *reinterpret_cast<unsigned*>(ps) = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
          *reinterpret_cast<unsigned*>(ps + 32) = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
        } else {
          const __bf16 h0 = (__bf16)hn[0];
          *reinterpret_cast<unsigned short*>(ps) = __builtin_bit_cast(unsigned short, h0);
          *reinterpret_cast<unsigned short*>(ps + 32) = __builtin_bit_cast(unsigned short, (__bf16)(hn[0] - (float)h0));
        }
      }
#else
      put_split(a.hsplit + (size_t)(t & 1) * bufb, hn);
#endif
      const size_t o = (size_t)m * ldh + (size_t)t * H + u0;
      lfv_store<UPT>(a.hseq + o, hn);
      if (a.xsum) lfv_store<UPT>(a.xsum + o, hs);
      if (a.train) {
        float* gr = a.xp + (size_t)m * ldx + (size_t)t * 4 * H + u0;
        lfv_store<UPT>(gr, vi);
        lfv_store<UPT>(gr + H, vf);
        lfv_store<UPT>(gr + 2 * H, vg);
        lfv_store<UPT>(gr + 3 * H, vo);
        lfv_store<UPT>(a.cseq + o, c);
      }
    }
#if MFPA_LSTM_WIDE_PUT
    if (t + 1 < a.t1) {
      // the slab's new h as 16-byte write-through (sc1) stores: thread j takes piece j & 3 (hi 0..7, hi 8..15, lo 0..7, lo 8..15) of clip j >> 2
      __syncthreads();
      if (tid < 4 * SLAB) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int sc = tid >> 2, piece = tid & 3;
        const int sm = m0 + sc;
        if (sm < a.B) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(PS + sc * 64 + piece * 16);
          const int ug = grp * LU;
          const unsigned off = (unsigned)((size_t)(t & 1) * bufb + (size_t)sm * rowb + (size_t)(ug >> 5) * 128 + (size_t)(ug & 31) * 2 + (size_t)(piece >> 1) * 64 +
                                          (size_t)(piece & 1) * 16);
          __builtin_amdgcn_raw_buffer_store_b128(v, hrsrc, off, 0, 16);
        }
      }
      arrive();
    }
#else
    if (t + 1 < a.t1) arrive();                            // (its __syncthreads also frees the gate slabs for the next step)
#endif
  }
  if (live && !a.train) lfv_store<UPT>(a.cstate + (size_t)m * H + u0, c);
}

static int lstm_seq_cus() { return mfpa_current_device_cus(); }

// Slab size (ms x 32 clips) and resident workgroups of the persistent forward launch for (B, H) under `wg_budget` (0 = one per CU):
// 32-clip slabs while their workgroups leave half the chip free, else 64-clip slabs; 0 workgroups = the per-step path.
static int lstm_seq_plan(int B, int H, int wg_budget, int* ms_out) {
  const int cus = lstm_seq_cus();
  const int budget = (wg_budget > 0 && wg_budget < cus) ? wg_budget : cus;
  const int ks = H / 128, ngroups = H / LU;
  static const int force_ms = MFPA_EXP_ENV("MFPA_LSTM_MS", 0);
  int ms = ((long long)((B + 31) / 32) * ngroups <= (budget < cus / 2 ? budget : cus / 2)) ? 1 : 2;
  if (force_ms == 1 || force_ms == 2) ms = force_ms;
  const int nslab = (B + 32 * ms - 1) / (32 * ms);
  if (ms_out) *ms_out = ms;
  if (H % 128 || !(ks == 2 || ks == 4 || ks == 6 || ks == 8) || nslab > 32 || (long long)B * H * 8 > 0x7fffffffLL ||
      (long long)nslab * ngroups > budget)
    return 0;
  return nslab * ngroups;
}

}  // namespace

extern "C" {

#define SK_LDS(KCW) ((size_t)(GBM + GBN) * (4 * (KCW) + 16))
int mfpa_gemm_mfma(const mfpa_gemm_desc* d, void* stream) {
  if (!d) return MFPA_EINVAL;
  if (d->batch == 0 || d->M == 0) return MFPA_OK;
  if ((!d->A && !d->c1_x) || !d->W || !d->C || d->batch < 0 || d->M < 0 || d->N < 1 || d->K < GKC || d->K % GKC) return MFPA_EINVAL;
  if (d->c1_x && (!d->c1_w || !d->c1_b || d->K > 256 || (d->precision == 1 && d->K >= 128) || d->c1_lin < 4 * ((long long)d->M - 1) + 8 ||
                  d->c1_lin % 4)) return MFPA_EINVAL;
  if (d->npad < 64 || d->npad % 64 || d->mode < 0 || d->mode > 3 || (d->mode >= 2 && !d->addend)) return MFPA_EINVAL;
  if (d->lda % 4 || d->strideA % 4) return MFPA_EINVAL;   // float4 row loads
  {   // the epilogue addresses a clip's outputs with 32-bit byte offsets
    const long long lim = 0x3fffffffLL;      // floats
    if ((long long)d->M * d->ldc + d->npad > lim || (d->C2 && (long long)d->M * d->ldc2 + d->npad > lim) ||
        (d->mode >= 2 && (long long)d->M * d->ldadd + d->npad > lim)) return MFPA_EINVAL;
  }
  if (d->mode == 1 ? (d->N > d->npad / 2) : (d->N > d->npad)) return MFPA_EINVAL;
  GemmArgs a{};
  a.A = d->A; a.lda = d->lda; a.strideA = d->strideA; a.W = d->W; a.bias = d->bias;
  a.addend = d->addend; a.ldadd = d->ldadd; a.strideAdd = d->strideAdd;
  a.C = d->C; a.ldc = d->ldc; a.strideC = d->strideC;
  a.C2 = d->C2; a.ldc2 = d->ldc2; a.strideC2 = d->strideC2;
  a.M = d->M; a.N = d->N; a.K = d->K; a.mode = d->mode; a.relu = d->relu;
  a.c1_x = d->c1_x; a.c1_lin = d->c1_lin; a.c1_w = d->c1_w; a.c1_b = d->c1_b;
  static const int xcd_env = MFPA_EXP_ENV("MFPA_GEMM_XCD", 1);   // 0: plain tile order (experiments)
  a.ny = (d->M + GBM - 1) / GBM; a.nz = d->batch; a.nx = d->npad / GBN; a.xcd = xcd_env;
  a.exp = MFPA_EXP_ENV("MFPA_GEMM_EXP", 0);
  if ((long long)a.nx * a.ny * a.nz > 0x3fffffffLL) return MFPA_EINVAL;
  auto grid1d = [&](int nx) { a.nx = nx; return dim3((unsigned)((((long long)nx * a.ny * a.nz + 7) / 8) * 8)); };
  dim3 grid = grid1d(d->npad / GBN);
  if (d->precision < 0 || d->precision > 2) return MFPA_EINVAL;
  if (d->precision == 2 && !(d->K % HKC == 0 && d->K >= 128 && d->npad % WBN == 0 && !d->c1_x)) return MFPA_EINVAL;   // pre-split W: the wide kernel only
  // K >= 128: the chunked bf16x3 kernel.  (At first the K = 128 / 192 levels ran faster on the fp32 kernel; that was the
  // epilogue's serialised addend loads and 64-bit addressing, not the arithmetic: with those fixed the fp32 MFMA rate is what
  // bounds them -- PMC: 2.2 of 4.0 ms MFMA-busy on the K = 192 transposed convolution -- and bf16x3 is 10 % faster end to end.)
  static const int shortk = MFPA_EXP_ENV("MFPA_SHORTK", 1);   // 0: the fp32-MFMA kernels for K < 256 (experiments)
  hipStream_t st = mfpa_stream(stream);
  static const int wide = MFPA_EXP_ENV("MFPA_GEMM_WIDE", 1);   // 0: always the 128 x 64 tile (experiments)
  static const int pipe = MFPA_EXP_ENV("MFPA_GEMM_PIPE", 1);   // 0: the 128 x 128 kernel without the software pipeline (experiments)
  const bool wide_ok = d->K % HKC == 0 && d->K >= 128 && d->npad % WBN == 0 && (d->precision == 2 || (d->precision == 1 && wide));
  if (wide_ok && pipe && d->K % (2 * HKC) == 0 && d->M >= MFPA_EXP_ENV("MFPA_GEMM_PIPE_MINM", 192)) {   // (rows past M are clamped when loaded: a 249-row clip fills 97 % of a 256-row tile)
    a.ny = (d->M + PBM - 1) / PBM;
    dim3 gw = grid1d(d->npad / WBN);
    static const int persist = MFPA_EXP_ENV("MFPA_GEMM_PERSIST", 1);   // 0: one tile per workgroup (experiments)
    const unsigned cus8 = (unsigned)((mfpa_current_device_cus() + 7) / 8 * 8);
    if (persist && cus8 >= 8 && gw.x > cus8) gw.x = cus8;    // persistent: one workgroup per CU walks the tiles (a multiple of 8: XCD ranges)
    const size_t lds = (size_t)2 * (PBM + WBN) * HROW;
    if (d->precision == 2) hipLaunchKernelGGL(gemm_bf16x3_pipe_kernel<true>, gw, dim3(512), lds, st, a);
    else hipLaunchKernelGGL(gemm_bf16x3_pipe_kernel<false>, gw, dim3(512), lds, st, a);
  } else if (d->precision == 2) {
    dim3 gw = grid1d(d->npad / WBN);
    hipLaunchKernelGGL(gemm_bf16x3_wide_kernel<true>, gw, dim3(256), (size_t)2 * (GBM + WBN) * HROW, mfpa_stream(stream), a);
  } else if (d->precision == 1 && d->K % HKC == 0 && d->K >= 128 && wide && d->npad % WBN == 0) {
    dim3 gw = grid1d(d->npad / WBN);
    hipLaunchKernelGGL(gemm_bf16x3_wide_kernel<false>, gw, dim3(256), (size_t)2 * (GBM + WBN) * HROW, mfpa_stream(stream), a);
  } else if (d->precision == 1 && d->K % HKC == 0 && d->K >= 128) {
    hipLaunchKernelGGL(gemm_bf16x3_kernel, grid, dim3(256), 0, mfpa_stream(stream), a);
  } else if (d->precision == 1 && shortk && d->K == 48 && d->c1_x) {
    hipLaunchKernelGGL((gemm_shortk_bf16x3_kernel<true, 48, 48>), grid, dim3(256), SK_LDS(48) + 9 * 48 * 4, st, a);
  } else if (d->precision == 1 && shortk && d->K == 48) {
    hipLaunchKernelGGL((gemm_shortk_bf16x3_kernel<false, 48, 48>), grid, dim3(256), SK_LDS(48), st, a);
  } else if (d->precision == 1 && shortk && d->K == 96 && !d->c1_x) {
    hipLaunchKernelGGL((gemm_shortk_bf16x3_kernel<false, 96, 48>), grid, dim3(256), SK_LDS(48), st, a);
  } else if (d->K == 48 && d->c1_x) {
    hipLaunchKernelGGL((gemm_smallk_kernel<true, 48>), grid, dim3(256), 0, mfpa_stream(stream), a);
  } else if (d->K == 48) {
    hipLaunchKernelGGL((gemm_smallk_kernel<false, 48>), grid, dim3(256), 0, mfpa_stream(stream), a);

  } else if (d->c1_x) {
    hipLaunchKernelGGL(gemm_mfma_kernel<true>, grid, dim3(256), 0, mfpa_stream(stream), a);
  } else {
    hipLaunchKernelGGL(gemm_mfma_kernel<false>, grid, dim3(256), 0, mfpa_stream(stream), a);
  }
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_demucs_prep(const float* wav, int B, int T, int VL, float floor_, float* out, float* stdv, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!wav || !out || !stdv || B < 0 || T < 2 || VL < T) return MFPA_EINVAL;
  hipLaunchKernelGGL(demucs_prep_kernel, dim3(B), dim3(256), 0, mfpa_stream(stream), wav, T, VL, floor_, out, stdv);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_upsample2(const float* x, int B, int T, const float* kernel112, float* y, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!x || !kernel112 || !y || B < 0 || B > 65535 || T < 1) return MFPA_EINVAL;
  const int gx = (T + RS_OUT - 1) / RS_OUT;
  hipLaunchKernelGGL(upsample2_kernel, dim3(gx, B), dim3(256), 0, mfpa_stream(stream), x, T, kernel112, y);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_downsample2(const float* x, int B, int T, const float* kernel112, float* y, int To, const float* scale, int Tkeep,
                     void* stream) {
  if (B == 0) return MFPA_OK;
  if (!x || !kernel112 || !y || B < 0 || B > 65535 || T < 2) return MFPA_EINVAL;
  const int nout = Tkeep > 0 ? Tkeep : (T + 1) / 2;
  if (nout > (T + 1) / 2 || To < nout) return MFPA_EINVAL;
  const int gx = (nout + RS_OUT - 1) / RS_OUT;
  hipLaunchKernelGGL(downsample2_kernel, dim3(gx, B), dim3(256), 0, mfpa_stream(stream), x, T, kernel112, y, To, scale, Tkeep);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_conv1d_c1(const float* x, int B, int Lin, int Lout, int C, const float* w, const float* bias, int relu, float* y,
                   void* stream) {
  if (B == 0) return MFPA_OK;
  if (!x || !w || !y || B < 0 || B > 65535 || C < 4 || C % 4 || Lout < 1 || Lin < 4 * (Lout - 1) + 8) return MFPA_EINVAL;
  long long blocks = ((long long)Lout * (C / 4) + 255) / 256; if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(conv1d_c1_kernel, dim3((unsigned)blocks, B), dim3(256), 0, mfpa_stream(stream), x, Lin, Lout, C, w, bias, y, relu);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_conv1d_c1_relu(const float* x, int B, int Lin, int Lout, int C, const float* w, const float* bias, float* y,
                        void* stream) {
  if (B != 0 && !bias) return MFPA_EINVAL;
  return mfpa_conv1d_c1(x, B, Lin, Lout, C, w, bias, 1, y, stream);
}

int mfpa_convT1d_c1(const float* P, int B, int L, int C, const float* w, float bias, float* y, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!P || !w || !y || B < 0 || B > 65535 || L < 1 || C < 4 || C % 4) return MFPA_EINVAL;
  int gx = (4 * (L + 1) + 255) / 256; if (gx > 2048) gx = 2048;
  hipLaunchKernelGGL(convT1d_c1_kernel, dim3(gx, B), dim3(256), 0, mfpa_stream(stream), P, L, C, w, bias, (const float*)nullptr, y);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_convT1d_c1_dev(const float* P, int B, int L, int C, const float* w, const float* bias_dev, float* y, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!P || !w || !bias_dev || !y || B < 0 || B > 65535 || L < 1 || C < 4 || C % 4) return MFPA_EINVAL;
  int gx = (4 * (L + 1) + 255) / 256; if (gx > 2048) gx = 2048;
  hipLaunchKernelGGL(convT1d_c1_kernel, dim3(gx, B), dim3(256), 0, mfpa_stream(stream), P, L, C, w, 0.f, bias_dev, y);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_lstm_step(const float* hprev, long long ldhp, const float* whh_grouped, const float* xp, long long ldxp, float* c,
                   int B, int H, float* hout, long long ldh, float* hsum, const float* addend, long long ldadd, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!whh_grouped || !xp || !c || !hout || B < 0 || H < LKC || H % LKC) return MFPA_EINVAL;
  if (ldhp % 4 || ldxp % 4 || ldh % 4 || ldadd % 4 || (hsum && !addend)) return MFPA_EINVAL;   // float4 rows
  return lstm_launch(hprev, ldhp, whh_grouped, xp, ldxp, c, (long long)H, c, (long long)H, B, H, hout, ldh, hsum, addend, ldadd,
                     nullptr, 0LL, stream);
}

int mfpa_lstm_step_train(const float* hprev, long long ldhp, const float* whh_grouped, const float* xp, long long ldxp,
                         const float* cprev, long long ldcp, float* cout, long long ldco, int B, int H, float* hout, long long ldh,
                         float* hsum, const float* addend, long long ldadd, float* gsave, long long ldgs, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!whh_grouped || !xp || !cout || !hout || !gsave || B < 0 || H < LKC || H % LKC) return MFPA_EINVAL;
  if (ldhp % 4 || ldxp % 4 || ldh % 4 || ldadd % 4 || ldcp % 4 || ldco % 4 || ldgs % 4 || (hsum && !addend)) return MFPA_EINVAL;
  return lstm_launch(hprev, ldhp, whh_grouped, xp, ldxp, cprev, ldcp, cout, ldco, B, H, hout, ldh, hsum, addend, ldadd, gsave, ldgs,
                     stream);
}

/* A whole LSTM layer: the Tn time steps of mfpa_lstm_step / mfpa_lstm_step_train launched from one host loop (one call across
 * the ABI instead of Tn: the Python-side cost of 2 x 248 launches was a third of a 16-clip training step). */
int mfpa_lstm_layer_range(const float* whh_grouped, float* xp, float* hseq, float* cseq, float* cstate, int B, int Tn, int H,
                          float* xsum, const float* skip, int train, int t0, int t1, void* stream) {
  if (B == 0 || Tn == 0 || t1 <= t0) return MFPA_OK;
  if (!whh_grouped || !xp || !hseq || B < 0 || Tn < 0 || H < LKC || H % LKC || (xsum && !skip) || t0 < 0 || t1 > Tn) return MFPA_EINVAL;
  if (train ? !cseq : !cstate) return MFPA_EINVAL;
  const long long ldh = (long long)Tn * H, ldx = (long long)Tn * 4 * H;
  if (!train && t0 == 0) MFPA_HIP(hipMemsetAsync(cstate, 0, (size_t)B * H * sizeof(float), mfpa_stream(stream)));
  for (int t = t0; t < t1; ++t) {
    const float* hprev = t ? hseq + (size_t)(t - 1) * H : nullptr;
    float* xt = xp + (size_t)t * 4 * H;
    int rc;
    if (train)
      rc = lstm_launch(hprev, ldh, whh_grouped, xt, ldx, t ? cseq + (size_t)(t - 1) * H : nullptr, ldh, cseq + (size_t)t * H, ldh, B, H,
                       hseq + (size_t)t * H, ldh, xsum ? xsum + (size_t)t * H : nullptr, skip ? skip + (size_t)t * H : nullptr, ldh, xt, ldx,
                       stream);
    else
      rc = lstm_launch(hprev, ldh, whh_grouped, xt, ldx, cstate, (long long)H, cstate, (long long)H, B, H, hseq + (size_t)t * H, ldh,
                       xsum ? xsum + (size_t)t * H : nullptr, skip ? skip + (size_t)t * H : nullptr, ldh, nullptr, 0LL, stream);
    if (rc != MFPA_OK) return rc;
  }
  return MFPA_OK;
}

int mfpa_lstm_layer(const float* whh_grouped, float* xp, float* hseq, float* cseq, float* cstate, int B, int Tn, int H, float* xsum,
                    const float* skip, int train, void* stream) {
  return mfpa_lstm_layer_range(whh_grouped, xp, hseq, cseq, cstate, B, Tn, H, xsum, skip, train, 0, Tn, stream);
}

/* The persistent form of mfpa_lstm_layer_range (lstm_seq_kernel): one launch for steps [t0, t1).  `work` = device scratch of
 * mfpa_lstm_seq_work_bytes(B, H) bytes, private to this layer while the call is in flight; its error word (mfpa_lstm_seq_error)
 * must be zero before the first use (hipMemset the buffer once).  Shapes the persistent kernel does not take (H not 128 KS for
 * KS in {2, 4, 6, 8}, more 64-clip slabs x H / 16 groups than CUs) run the per-step kernels: the result is the same either way. */
int mfpa_lstm_seq_work_bytes(int B, int H, long long* bytes) {
  if (!bytes || B < 0 || H < 0) return MFPA_EINVAL;
  *bytes = (long long)LSTM_SYNC_WORDS * 4 + 2LL * B * H * 4;
  return MFPA_OK;
}

int mfpa_lstm_seq_error_offset(void) { return LSTM_ERR_WORD * 4; }

int mfpa_lstm_seq_workgroups(int B, int H, int wg_budget, int* workgroups) {
  if (!workgroups || B < 0 || H < LKC || H % LKC) return MFPA_EINVAL;
  static const int persistent = MFPA_EXP_ENV("MFPA_LSTM_SEQ", 1);
  *workgroups = (persistent && B > 0) ? lstm_seq_plan(B, H, wg_budget, nullptr) : 0;
  return MFPA_OK;
}

int mfpa_lstm_layer_seq(const float* whh_grouped, float* xp, float* hseq, float* cseq, float* cstate, int B, int Tn, int H, float* xsum,
                        const float* skip, int train, int t0, int t1, int wg_budget, void* work, void* stream) {
  if (B == 0 || Tn == 0 || t1 <= t0) return MFPA_OK;
  if (!whh_grouped || !xp || !hseq || !work || B < 0 || Tn < 0 || H < LKC || H % LKC || (xsum && !skip) || t0 < 0 || t1 > Tn) return MFPA_EINVAL;
  if (train ? !cseq : !cstate) return MFPA_EINVAL;
  const int ks = H / 128, ngroups = H / LU;
  static const int persistent = MFPA_EXP_ENV("MFPA_LSTM_SEQ", 1);
  // every workgroup of the launch must be resident at once: the plan keeps them within `wg_budget` (0 = one per CU of the current
  // device; a caller running two such launches side by side -- the chunked two-stream pipeline -- passes half the CU count)
  int ms = 2;
  const int wgs = persistent ? lstm_seq_plan(B, H, wg_budget, &ms) : 0;
  if (wgs == 0)
    return mfpa_lstm_layer_range(whh_grouped, xp, hseq, cseq, cstate, B, Tn, H, xsum, skip, train, t0, t1, stream);
  const int nslab = (B + 32 * ms - 1) / (32 * ms);
  LstmSeqArgs a;
  a.whh = whh_grouped; a.xp = xp; a.hseq = hseq; a.cseq = cseq; a.cstate = cstate; a.xsum = xsum; a.skip = skip;
  a.sync = reinterpret_cast<unsigned*>(work);
  a.hsplit = reinterpret_cast<char*>(work) + (size_t)LSTM_SYNC_WORDS * 4;
  a.B = B; a.Tn = Tn; a.H = H; a.t0 = t0; a.t1 = t1; a.train = train; a.nslab = nslab; a.ngroups = ngroups;
  a.dbg = MFPA_EXP_ENV("MFPA_LSTM_DBG", 0);
  hipStream_t st = mfpa_stream(stream);
  MFPA_HIP(hipMemsetAsync(work, 0, (size_t)LSTM_ERR_WORD * 4, st));          // the slab counters; the error word stays
  const unsigned grid = (unsigned)(((nslab * ngroups + 7) / 8) * 8);
  const size_t lds = (size_t)QW * 32 * ms * QGLD * sizeof(float) + (MFPA_LSTM_WIDE_PUT ? (size_t)32 * ms * 64 : 0);
  static const int coh = MFPA_EXP_ENV("MFPA_LSTM_COH", 1);   // 0: one L1 / L2 invalidate per step + cached loads (7.83 vs 7.58 ms for both layers of 256 clips)
#define SEQ_LAUNCH(KS_)                                                                                                   \
  if (ms == 1) hipLaunchKernelGGL((lstm_seq_kernel<KS_, 1, 1>), dim3(grid), dim3(64 * QW), lds, st, a);                   \
  else if (coh) hipLaunchKernelGGL((lstm_seq_kernel<KS_, 1, 2>), dim3(grid), dim3(64 * QW), lds, st, a);                  \
  else hipLaunchKernelGGL((lstm_seq_kernel<KS_, 0, 2>), dim3(grid), dim3(64 * QW), lds, st, a)
  switch (ks) {
    case 2: SEQ_LAUNCH(2); break;
    case 4: SEQ_LAUNCH(4); break;
    case 6: SEQ_LAUNCH(6); break;
    default: SEQ_LAUNCH(8); break;
  }
#undef SEQ_LAUNCH
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

/* Last decoder level of Demucs in one launch (model.py:80-88: Conv1d(C, 2C, 1) + GLU + ConvTranspose1d(C, 1, 8, 4), no ReLU):
 * x (B, L, C) -> y (B, 4 (L + 1)).  gw / gb = the 1x1 weights and bias in the packed GLU tile order of mfpa_gemm_mfma mode 1
 * (128 rows x C for C = 48), wl (8, C) tap-major = weight[c][0][j], bias = the ConvTranspose1d bias.  C must be 48. */
int mfpa_glu_convT1d_c1(const float* x, int B, int L, int C, const float* gw, const float* gb, const float* wl, float bias, float* y,
                        void* stream) {
  if (B == 0) return MFPA_OK;
  if (!x || !gw || !gb || !wl || !y || B < 0 || L < 1 || C != TT_K || B > 65535) return MFPA_EINVAL;
  const int groups = L + 1;
  const int tiles = (groups + TT_OUT - 1) / TT_OUT;
  hipLaunchKernelGGL(glu_convT_c1_kernel, dim3((tiles + TT_TPW - 1) / TT_TPW, B), dim3(256), 0, mfpa_stream(stream), x, L, gw, gb, wl, bias, y,
                     tiles, groups);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

/* First encoder level of Demucs in one launch (model.py:66-75: Conv1d(1, C, 8, 4) + ReLU + Conv1d(C, 2C, 1) + GLU):
 * x (B, Lin) -> y (B, Lout, C), Lout = (Lin - 8) / 4 + 1.  w1 (8, C) tap-major and b1 (C) as for mfpa_conv1d_c1; gw (128, C) / gb
 * (128) in the packed GLU tile order of mfpa_gemm_mfma mode 1.  C must be 48. */
int mfpa_conv1d_c1_glu(const float* x, int B, int Lin, int Lout, int C, const float* w1, const float* b1, const float* gw,
                       const float* gb, float* y, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!x || !w1 || !b1 || !gw || !gb || !y || B < 0 || B > 65535 || C != TT_K || Lout < 1 || Lin < 4 * ((long long)Lout - 1) + 8 ||
      (long long)Lout * C * 4 > 0xffffffffLL || Lin % 4 != 0 || ((size_t)x & 15) != 0)        // the rows' samples are read as aligned float4
    return MFPA_EINVAL;
  const int tiles = (Lout + 127) / 128;
  hipLaunchKernelGGL(c1_glu_kernel, dim3((tiles + TT_TPW - 1) / TT_TPW, B), dim3(256), (size_t)MFPA_EXP_ENV("MFPA_HEAD_LDS", 0), mfpa_stream(stream), x, Lin, Lout, w1, b1, gw, gb, y,
                     tiles);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}
int mfpa_lstm_cell(const float* gates, long long ldg, float* c, int B, int H, float* hout, long long ldh, float* hsum,
                   const float* addend, long long ldadd, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!gates || !c || !hout || B < 0 || H < 1 || (hsum && !addend)) return MFPA_EINVAL;
  int gx = (B * H + 255) / 256; if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(lstm_cell_kernel, dim3(gx), dim3(256), 0, mfpa_stream(stream), gates, ldg, c, B, H, hout, ldh, hsum, addend, ldadd);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

}  // extern "C"
