// A decoder level's FIRST convolution with the transposed convolution folded into it (training/unet.py:41-65, inference; bf16x3 products on
// v_mfma_f32_16x16x32_bf16, or exact fp32 products on v_mfma_f32_16x16x4_f32: template parameter PREC).  Round 6.
//
// The reference computes   up = ConvTranspose2d(k 2, s 2)(low) + bt;  pad up to the skip's size;  y = relu(bn(conv3x3(cat([skip, up])))).
// Nothing non-linear sits between the transposed convolution and the 3x3 convolution, so the `up` half of that convolution is, per output
// PHASE (py, px) = (Y & 1, X & 1), a 2 x 2 convolution of the LOW-resolution tensor with composite weights
//     Wc[py, px][ty, tx][co][ci] = sum over the 3x3 taps (a, b) whose up-sampled pixel (Y + a, X + b) lies in low-resolution pixel
//                                  (Y / 2 + ty - 1 + py, X / 2 + tx - 1 + px) of   sum_cu W3[a, b][co][Cs + cu] * Wt[(py + a) & 1, (px + b) & 1][cu][ci]
// (mfpa_upconv_pack below; algebra checked in float64 by tools/exp_phase_composite.py) plus a bias that depends only on which of the nine
// taps fall inside the up-sampled extent (a 4 x 4 table of per-channel vectors: first / interior / last row of the extent / the padding row
// of an odd size, likewise for columns).  Zero padding of the low-resolution tensor reproduces the zero padding of the up-sampled one.
// Per output and output channel: 4 x C_low products instead of 9 x C_up + C_low (-27 %), the up-sampled tensor never exists in HBM and the
// write-bound transposed-convolution launch is gone.
//
// Kernel = conv_ws64_kernel's design (csrc/unet_ws.hip: 4 compute waves of 128 px x 32 ch, 4 loader waves that request / split the next
// chunk's halo and store the finished tile, one s_barrier per 32-channel chunk, persistent tile loop) with a different pixel order: the 16
// pixels of an MFMA's B operand must share a phase, so a 8 x 32 tile's pixel groups are (row, column parity) -- 16 pixels two columns apart --
// and compute wave `wm` owns the four rows of ONE row parity.  Per tile the K loop runs the skip half as before (nine taps on a 10 x 34 halo
// patch of the skip, whose columns the loaders stage DE-INTERLEAVED: even halo columns, then odd ones, so that a group's fragment is still
// 16 consecutive 16-byte pieces of a plane) and then the low-resolution half: per 32-channel chunk a 6 x 18 patch of `low` and four
// (ty, tx) "pairs" -- the two column phases of a pair are the two halves of what a tap is in the skip half, each half with its own
// weight fragments (the wave's row phase and the half's column phase pick the composite), a ring of two pairs of register sets.
#include "mfpa_common.h"
#include "mfpa_unet_args.h"

#include <type_traits>

namespace mfpa_unet {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int KC = 32;                                                 // channels per K chunk
constexpr int PH = 8, PW = 32, HPW = PW + 2, HPH = PH + 2, HP = HPW * HPH;   // the skip's halo patch: 10 x 34
constexpr int HALF = HPW / 2;                                          // a staged halo row: 17 even halo columns, then the 17 odd ones
constexpr int LPH = PH / 2 + 2, LPW = PW / 2 + 2, LP = LPH * LPW;      // the low-resolution patch: 6 x 18
constexpr int THREADS = 512, LTHREADS = 256;
constexpr int SPP = KC / 4;                                            // staging slots (16 B = 4 fp32 channels) per pixel and chunk
constexpr int PPI = LTHREADS / SPP;                                    // pixels per loader pass
constexpr int A_F4 = (HP + PPI - 1) / PPI;                             // staging slots per loader thread and skip chunk (11)
constexpr int L_F4 = (LP + PPI - 1) / PPI;                             // ... and low-resolution chunk (4)
constexpr int HPS = A_F4 * PPI;
constexpr int PLANE = ((HPS * 16 + 255) / 256) * 256;                  // bytes of one (hi | lo, k-group) plane
constexpr int HLS = 4 * PLANE + 256;                                   // hi -> lo distance (planes 2, 3 sit 128 B further)
constexpr int STAGE = 2 * HLS;
constexpr int PT = 8;                                                  // 16-pixel groups per compute wave: index 4 * px + j (column parity, row pair)
constexpr int EPI_FLOATS = 64 + 16 * 64;                               // [shift 64 | bias table 4 x 4 x 64]
constexpr int OUTBUF = PH * PW * 64 * 4;                                // the epilogue's LDS tile (see conv_ws64_kernel)

__device__ __forceinline__ constexpr int plane_off(int hl, int kg) { return hl * HLS + kg * PLANE + (kg >> 1) * 128; }

template <int SLOTS, int LEFT, int I = 0>
__device__ __forceinline__ void pin_reads() {
  if constexpr (I < SLOTS && LEFT > 0) {
    constexpr int k = (LEFT + (SLOTS - I) - 1) / (SLOTS - I);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, k, 0);
    pin_reads<SLOTS, LEFT - k, I + 1>();
  }
}
constexpr int pin_read_slots(int slots, int left) {
  int used = 0;
  for (int i = 0; i < slots && left > 0; ++i) {
    left -= (left + (slots - i) - 1) / (slots - i);
    ++used;
  }
  return used;
}

__device__ __forceinline__ int tile_of(int b, int i, int G, int ngrp) {
  // one channel group: walkers b and b + 8 share an XCD, and the 32 walkers of an XCD take consecutive tiles (their shared halo rows meet in its L2);
  // several channel groups: walker b's XCD is fixed by its workgroup id as a whole -- plain order
  if (ngrp == 1 && (G & 7) == 0) return i * G + (b & 7) * (G >> 3) + (b >> 3);
  return i * G + b;
}

// bias class of an output row / column: which of the three taps along that axis fall inside the up-sampled extent [0, 2 n)
__device__ __forceinline__ int bias_class(int v, int n2) { return v == 0 ? 0 : (v < n2 - 1 ? 1 : (v == n2 - 1 ? 2 : 3)); }

// PREC 1: bf16x3 products (v_mfma_f32_16x16x32_bf16; operands split hi | lo while staged: planes [hi | lo][k-group]).
// PREC 0: exact fp32 products (v_mfma_f32_16x16x4_f32, the reference's arithmetic): the stage's eight planes hold the eight 16-byte pieces
// of a pixel's 32-channel chunk as they are (the loaders only copy), lane (p, g) takes channels 8 g .. 8 g + 7 (pieces 2 g, 2 g + 1) and the
// eight k-steps of a chunk pair channel 8 g + s of the pixel with the same channel of the weight fragment.
template <int PREC>
__global__ __launch_bounds__(THREADS, 1) void conv_up_kernel(UpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nsk = a.Cs / KC, nup = a.Cl / KC, nchunks = nsk + nup;
  const int ntiles = a.tiles_x * a.tiles_y * a.B;
  // one-dimensional grid, workgroup id = tile walker * (Cout / 64) + channel group: workgroup ids go to the XCDs round-robin, so with 8 (4, 2)
  // channel groups every XCD serves ONE (two, four) of them and its 32 CUs stream the same slice of the weight images through that XCD's L2 --
  // up1's images are 43 MB, eight slices of 5.4 MB (with the two-dimensional grid every XCD streamed all of them)
  const int ngrp = a.Cout / 64;
  const int G = (int)gridDim.x / ngrp;
  const int wgx = (int)blockIdx.x / ngrp;
  const int n0 = ((int)blockIdx.x % ngrp) * 64;
  const int first_tile = tile_of(wgx, 0, G, ngrp);
  const int owned = first_tile < ntiles ? (ntiles - first_tile + G - 1) / G : 0;

  // LDS: [stage 0 | stage 1 | shift 64, bias table 16 x 64 | the epilogue's output tile 64 KB]
  float* const epi = reinterpret_cast<float*>(smem + 2 * STAGE);
  char* const outbuf = smem + 2 * STAGE + EPI_FLOATS * sizeof(float);
  for (int i = tid; i < 64; i += THREADS) epi[i] = a.shift[n0 + i];
  for (int i = tid; i < 16 * 64; i += THREADS) epi[64 + i] = a.bias_tab[(i >> 6) * a.Cout + n0 + (i & 63)];

  if (wave >= 4) {
    // =================================================================================================== LOADER waves
    const int lt = tid - LTHREADS;
    const int aq = lt % SPP, pl = lt / SPP;
    struct Tile { int b, y0, x0; unsigned ain; unsigned t0, t1; bool int0, int1; };   // ain: bits 0..10 skip slots, 16..19 low slots
    auto make_tile = [&](int t) __attribute__((always_inline)) {
      Tile T;
      int bx = __builtin_amdgcn_readfirstlane(t);
      const int tx = bx % a.tiles_x; bx /= a.tiles_x;
      const int ty = bx % a.tiles_y; bx /= a.tiles_y;
      T.b = bx; T.y0 = ty * PH; T.x0 = tx * PW;
      const int ly0 = T.y0 / 2 - 1, lx0 = T.x0 / 2 - 1;                  // low-resolution pixel of patch position (0, 0)
      T.t0 = (unsigned)(((T.y0 - 1) * a.W + (T.x0 - 1)) * a.Cs) * 4u;    // halo origins: may wrap (rows above the clip)
      T.t1 = (unsigned)((ly0 * a.Wl + lx0) * a.Cl) * 4u;
      T.int0 = T.y0 >= 1 && T.y0 + PH + 1 <= a.H && T.x0 >= 1 && T.x0 + PW + 1 <= a.W;
      T.int1 = ly0 >= 0 && ly0 + LPH <= a.Hl && lx0 >= 0 && lx0 + LPW <= a.Wl;
      T.ain = 0;
      if (!T.int0) {
#pragma unroll
        for (int it = 0; it < A_F4; ++it) {
          const int pix = pl + it * PPI;
          const int pos = pix % HPW, hx = pos < HALF ? 2 * pos : 2 * (pos - HALF) + 1;
          const int gy = T.y0 + pix / HPW - 1, gx = T.x0 + hx - 1;
          T.ain |= ((pix < HP && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? 1u : 0u) << it;
        }
      }
      if (!T.int1) {
#pragma unroll
        for (int it = 0; it < L_F4; ++it) {
          const int pix = pl + it * PPI;
          const int ly = ly0 + pix / LPW, lx = lx0 + pix % LPW;
          T.ain |= ((pix < LP && ly >= 0 && ly < a.Hl && lx >= 0 && lx < a.Wl) ? 1u : 0u) << (16 + it);
        }
      }
      return T;
    };

    // the epilogue's memory work for tile `t` out of the LDS tile the compute waves filled: linear pixel n = 32 row + column of the tile sits at
    // LDS position m = (row & 1) * 128 + ((column & 1) * 4 + (row >> 1)) * 16 + (column >> 1) (the compute waves' group order); 16 lanes = the 16
    // channel quads of a pixel, so a wave instruction stores 4 consecutive pixels' 64-channel rows (1 KB)
    auto duty = [&](int t) __attribute__((always_inline)) {
      int bx = __builtin_amdgcn_readfirstlane(t);
      const int tx = bx % a.tiles_x; bx /= a.tiles_x;
      const int ty = bx % a.tiles_y; bx /= a.tiles_y;
      const int ey0 = ty * PH, ex0 = tx * PW;
      char* yb = reinterpret_cast<char*>(a.y + (size_t)bx * a.H * a.W * a.Cout + n0);
      const int q = lt & 15;
      for (int pass = 0; pass < 16; pass += 4) {                        // four LDS reads in flight, then their four stores
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int n = (pass + u) * 16 + (lt >> 4);
          const int r = n >> 5, c = n & 31;
          const int m = (r & 1) * 128 + ((c & 1) * 4 + (r >> 1)) * 16 + (c >> 1);
          v[u] = *reinterpret_cast<const f32x4*>(outbuf + m * 256 + ((q ^ (m & 15)) << 4));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int n = (pass + u) * 16 + (lt >> 4);
          const int gy = ey0 + (n >> 5), gx = ex0 + (n & 31);
          if (gy < a.H && gx < a.W) *reinterpret_cast<f32x4*>(yb + (((unsigned)gy * (unsigned)a.W + (unsigned)gx) * (unsigned)a.Cout + 4u * (unsigned)q) * 4u) = v[u];
        }
      }
    };

    // per-slot byte offsets relative to a tile's halo origin (tile-independent).  Skip: staged position `pos` of a halo row holds halo column
    // hx = 2 pos (pos < 17) or 2 (pos - 17) + 1.  A padding slot's vector offset lies beyond any clip (it is never inside).
    unsigned off0[A_F4], off1[L_F4];
#pragma unroll
    for (int it = 0; it < A_F4; ++it) {
      const int pix = pl + it * PPI;
      const int pos = pix % HPW, hx = pos < HALF ? 2 * pos : 2 * (pos - HALF) + 1;
      off0[it] = pix < HP ? (unsigned)(((pix / HPW) * a.W + hx) * a.Cs + 4 * aq) * 4u : 0xfffffff0u;
    }
#pragma unroll
    for (int it = 0; it < L_F4; ++it) {
      const int pix = pl + it * PPI;
      off1[it] = pix < LP ? (unsigned)(((pix / LPW) * a.Wl + pix % LPW) * a.Cl + 4 * aq) * 4u : 0xfffffff0u;
    }
    const unsigned clip0 = (unsigned)a.H * (unsigned)a.W * (unsigned)a.Cs * 4u, clip1 = (unsigned)a.Hl * (unsigned)a.Wl * (unsigned)a.Cl * 4u;
    f32x4 areg[2][A_F4] = {};
    struct Src { __amdgpu_buffer_rsrc_t rs; unsigned toff; bool from0, interior; int ainsh; };
    auto make_src = [&](const Tile& T, int chunk) __attribute__((always_inline)) {
      Src S;
      S.from0 = chunk < nsk;
      const char* pb = reinterpret_cast<const char*>(S.from0 ? a.skip : a.low) + (size_t)T.b * (S.from0 ? clip0 : clip1);
      S.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pb), 0, (int)(S.from0 ? clip0 : clip1), 0x00020000);
      S.toff = S.from0 ? T.t0 + (unsigned)(chunk * KC) * 4u : T.t1 + (unsigned)((chunk - nsk) * KC) * 4u;
      S.interior = S.from0 ? T.int0 : T.int1;
      S.ainsh = S.from0 ? 0 : 16;
      return S;
    };
    auto issue_slot = [&](const Src& S, unsigned ain, unsigned tab, auto INTERIOR, auto SET, auto IT) __attribute__((always_inline)) {
      constexpr int it = decltype(IT)::value, set = decltype(SET)::value;
      if constexpr (decltype(INTERIOR)::value) {   // table entry = vector offset, tile / channel offset = the instruction's scalar offset: no vector instruction per request
        areg[set][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(S.rs, (int)tab, (int)S.toff, 0));
      } else {                // a slot outside the image is requested beyond the clip: zeros come back, nothing is masked when it is split
        const bool inside = (ain >> (S.ainsh + it)) & 1u;
        areg[set][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(S.rs, (int)(inside ? tab + S.toff : 0xfffffff0u), 0, 0));
      }
    };
    char* const wbase = smem + plane_off(0, aq >> 1) + pl * 16 + 8 * (aq & 1);
    char* const wbase32 = smem + plane_off(aq >> 2, aq & 3) + pl * 16;       // PREC 0: piece aq of the pixel, as it is
    auto split_slot = [&](auto SET, auto IT, int stage_off) __attribute__((always_inline)) {
      constexpr int it = decltype(IT)::value, set = decltype(SET)::value;
      const f32x4 v = areg[set][it];
      if constexpr (PREC == 0) {
        *reinterpret_cast<f32x4*>(wbase32 + stage_off + it * PPI * 16) = v;
        return;
      }
      unsigned hi[2], lo[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x2 x = {v[2 * h], v[2 * h + 1]};
        hi[h] = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
        const f32x2 r = {x[0] - __uint_as_float(hi[h] << 16), x[1] - __uint_as_float(hi[h] & 0xffff0000u)};
        lo[h] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
      }
      char* at = wbase + stage_off + it * PPI * 16;
      *reinterpret_cast<uint2*>(at) = uint2{hi[0], hi[1]};
      *reinterpret_cast<uint2*>(at + HLS) = uint2{lo[0], lo[1]};
    };
    static_assert(A_F4 == 11 && L_F4 == 4, "eleven / four staging slots per loader thread");
#define MFPA_UP_EACH4(M) M(0) M(1) M(2) M(3)
#define MFPA_UP_EACH7(M) M(4) M(5) M(6) M(7) M(8) M(9) M(10)
    auto issue_all = [&](const Src& S, unsigned ain, auto SET) __attribute__((always_inline)) {
      if (S.from0) {
        if (S.interior) {
#define MFPA_UP_ISSUE(I) issue_slot(S, ain, off0[I], std::true_type{}, SET, std::integral_constant<int, I>{});
          MFPA_UP_EACH4(MFPA_UP_ISSUE) MFPA_UP_EACH7(MFPA_UP_ISSUE)
#undef MFPA_UP_ISSUE
        } else {
#define MFPA_UP_ISSUE(I) issue_slot(S, ain, off0[I], std::false_type{}, SET, std::integral_constant<int, I>{});
          MFPA_UP_EACH4(MFPA_UP_ISSUE) MFPA_UP_EACH7(MFPA_UP_ISSUE)
#undef MFPA_UP_ISSUE
        }
      } else {
        if (S.interior) {
#define MFPA_UP_ISSUE(I) issue_slot(S, ain, off1[I], std::true_type{}, SET, std::integral_constant<int, I>{});
          MFPA_UP_EACH4(MFPA_UP_ISSUE)
#undef MFPA_UP_ISSUE
        } else {
#define MFPA_UP_ISSUE(I) issue_slot(S, ain, off1[I], std::false_type{}, SET, std::integral_constant<int, I>{});
          MFPA_UP_EACH4(MFPA_UP_ISSUE)
#undef MFPA_UP_ISSUE
        }
      }
    };
    auto split_all = [&](auto SET, int stage_off, bool from0) __attribute__((always_inline)) {
#define MFPA_UP_SPLIT(I) split_slot(SET, std::integral_constant<int, I>{}, stage_off);
      MFPA_UP_EACH4(MFPA_UP_SPLIT)
      if (from0) { MFPA_UP_EACH7(MFPA_UP_SPLIT) }
#undef MFPA_UP_SPLIT
    };
    // the chunk sequence of this workgroup: (tile i, chunk c), c fastest: the skip's chunks, then the low-resolution tensor's
    int qi = 0, qc = 0;
    Tile TQ = make_tile(tile_of(wgx, 0, G, ngrp));
    auto issue_next = [&](auto SET) __attribute__((always_inline)) {
      const Src S = make_src(TQ, qc);
      issue_all(S, TQ.ain, SET);
      if (++qc == nchunks) {
        qc = 0; ++qi;
        if (qi < owned) TQ = make_tile(tile_of(wgx, qi, G, ngrp));
      }
    };
    using SET0 = std::integral_constant<int, 0>;
    using SET1 = std::integral_constant<int, 1>;
    issue_next(SET0{});
    int par = 0;
    const int total = owned * nchunks;
    // iteration k = -1 (prologue), then k = 0 .. total - 1: while the compute waves work on chunk k the loaders request chunk k + 2 and split
    // chunk k + 1 into the stage the compute waves read next; tile i's output (complete behind its last chunk) is stored two iterations later
    // (conv_ws64_kernel's schedule, unchanged)
    auto iteration = [&](int k, auto ISSUE_SET, auto SPLIT_SET) __attribute__((always_inline)) {
      if (k + 2 < total) issue_next(ISSUE_SET);
      __builtin_amdgcn_sched_barrier(0);
      if (k + 1 < total) split_all(SPLIT_SET, par * STAGE, ((k + 1) % nchunks) < nsk);
      if (k >= nchunks + 1 && (k - 1) % nchunks == 0) duty(tile_of(wgx, (k - 1) / nchunks - 1, G, ngrp));
      __syncthreads();
      par ^= 1;
    };
    for (int k = -1; k < total; k += 2) {
      iteration(k, SET1{}, SET0{});
      if (k + 1 < total) iteration(k + 1, SET0{}, SET1{});
    }
    __syncthreads();                                                   // the compute waves' final barrier: the last tile's output is in LDS
    duty(tile_of(wgx, owned - 1, G, ngrp));
    return;
  }

  // ===================================================================================================== COMPUTE waves
  const int wm = wave & 1, wn = wave >> 1;                             // row parity (= the row phase py), channel half
  const int p = lane & 15, g = lane >> 4;

  typedef typename std::conditional<PREC == 1, bf16x8, f32x4>::type frag_t;   // 16 bytes either way
  frag_t wq[4][2][2];                                                  // weight fragments: [set][16-channel tile][hi, lo | piece 0, 1]; the skip half rings sets 0..2
                                                                       // (tap t in set t % 3, two taps ahead), the low half pairs (0, 1) / (2, 3) (one pair ahead)
  // weight fragments through raw buffer loads: descriptor in scalar registers, ONE vector offset (lane * 16) for every load, the (tap, chunk,
  // channel-tile) block offset in the instruction's scalar offset -- as 64-bit global pointers hipcc kept a register pair per prefetch
  // target and spilled them (a scratch reload in front of a weight load drains the whole prefetch queue)
  const unsigned wrow = (unsigned)(a.Cout / 16), wcol = (unsigned)(n0 / 16 + 2 * wn);
  const __amdgpu_buffer_rsrc_t rs_skip = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w_skip), 0, (int)(9u * (unsigned)a.Cout * (unsigned)a.Cs * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_up = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w_up), 0, (int)(16u * (unsigned)a.Cout * (unsigned)a.Cl * 4u), 0x00020000);
  const int wlane = lane * 16;
  auto load_w_at = [&](__amdgpu_buffer_rsrc_t rs, unsigned blk, auto SLOT) __attribute__((always_inline)) {
    constexpr int slot = decltype(SLOT)::value;
    const int so = (int)__builtin_amdgcn_readfirstlane((int)(blk << 11));
    wq[slot][0][0] = __builtin_bit_cast(frag_t, __builtin_amdgcn_raw_buffer_load_b128(rs, wlane, so, 0));
    wq[slot][0][1] = __builtin_bit_cast(frag_t, __builtin_amdgcn_raw_buffer_load_b128(rs, wlane + 1024, so, 0));
    wq[slot][1][0] = __builtin_bit_cast(frag_t, __builtin_amdgcn_raw_buffer_load_b128(rs, wlane + 2048, so, 0));
    wq[slot][1][1] = __builtin_bit_cast(frag_t, __builtin_amdgcn_raw_buffer_load_b128(rs, wlane + 3072, so, 0));
  };
#ifdef MFPA_UP_WHOT            // timing-only variant (tools/: wrong results by design): every weight fragment comes from ONE block per wave, i.e. from L1
#define MFPA_UP_WBLK(x) (wcol)
#else
#define MFPA_UP_WBLK(x) (x)
#endif
  auto load_w_skip = [&](int chunk, int tap, auto SLOT) __attribute__((always_inline)) {
    load_w_at(rs_skip, MFPA_UP_WBLK(((unsigned)tap * (unsigned)nsk + (unsigned)chunk) * wrow + wcol), SLOT);
  };
  // composite image: "tap" index ((py * 2 + px) * 2 + ty) * 2 + tx, py = this wave's row parity
  auto load_w_up = [&](int chunk, int pair, int px, auto SLOT) __attribute__((always_inline)) {
    const unsigned t16 = (unsigned)((wm * 2 + px) * 4 + pair);
    load_w_at(rs_up, MFPA_UP_WBLK((t16 * (unsigned)nup + (unsigned)chunk) * wrow + wcol), SLOT);
  };
  struct XFrags { frag_t h[4], l[4]; };                                 // PREC 0: h = piece 2 g, l = piece 2 g + 1
  XFrags fx0, fx1;
  // fragment of group (j, px) -- tile rows 2 j + wm, columns 2 p + px:
  //   skip, tap (ta, tb): halo row 2 j + wm + ta, halo column 2 p + s with s = px + tb -> staged position (s & 1) * 17 + (s >> 1) + p
  //   low, pair (ty, tx): patch row j + ty + wm, patch column p + tx + px
  // PREC 1: plane (hi, g) and, HLS further, (lo, g); PREC 0: pieces 2 g and 2 g + 1 = planes (g >> 1, 2 (g & 1)) and the next one
  const int pl0 = PREC == 1 ? plane_off(0, g) : plane_off(g >> 1, 2 * (g & 1));
  constexpr int SECOND = PREC == 1 ? HLS : PLANE;                      // (planes 2 k and 2 k + 1 are PLANE apart: plane_off)
  const int xb_skip = (wm * HPW + p) * 16 + pl0;
  const int xb_up = (wm * LPW + p) * 16 + pl0;
  auto read_at = [&](XFrags& f, const char* base, auto STRIDE) __attribute__((always_inline)) {
    constexpr int stride = decltype(STRIDE)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const char* r = base + j * stride;
      f.l[j] = *reinterpret_cast<const frag_t*>(r + SECOND);
      f.h[j] = *reinterpret_cast<const frag_t*>(r);
    }
  };
  auto read_skip = [&](XFrags& f, const char* stage, auto TAP, auto PX) __attribute__((always_inline)) {
    constexpr int tap = decltype(TAP)::value, s = decltype(PX)::value + tap % 3;
    read_at(f, stage + xb_skip + ((tap / 3) * HPW + (s & 1) * HALF + (s >> 1)) * 16, std::integral_constant<int, 2 * HPW * 16>{});
  };
  auto read_up = [&](XFrags& f, const char* stage, auto PAIR, auto PX) __attribute__((always_inline)) {
    constexpr int pair = decltype(PAIR)::value;
    read_at(f, stage + xb_up + ((pair >> 1) * LPW + (pair & 1) + decltype(PX)::value) * 16, std::integral_constant<int, LPW * 16>{});
  };
  floatx4 acc[2][PT];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = floatx4{0.f, 0.f, 0.f, 0.f};
  floatx4 initv[2];                                                    // the accumulators' start values: the output shift (the scale is in the weights)
  auto mfma_half = [&](const XFrags& f, const frag_t (&w)[2][2], auto HALFI, auto FIRST) __attribute__((always_inline)) {
    constexpr int half = decltype(HALFI)::value;
    if constexpr (PREC == 0) {
      // eight k-steps of four channels: step s = element s & 3 of piece s >> 2; an accumulator is touched every eighth instruction
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float wv = (s8 < 4 ? w[ct][0] : w[ct][1])[s8 & 3], xv = (s8 < 4 ? f.h[i] : f.l[i])[s8 & 3];
            acc[ct][4 * half + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xv, (decltype(FIRST)::value && s8 == 0) ? initv[ct] : acc[ct][4 * half + i], 0, 0, 0);
          }
      return;
    } else {
    // term-major (lo x hi, hi x lo, hi x hi), as in conv_ws64_kernel
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[ct][4 * half + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ct][1], f.h[i], decltype(FIRST)::value ? initv[ct] : acc[ct][4 * half + i], 0, 0, 0);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[ct][4 * half + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ct][0], f.l[i], acc[ct][4 * half + i], 0, 0, 0);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[ct][4 * half + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ct][0], f.h[i], acc[ct][4 * half + i], 0, 0, 0);
    }
  };
  constexpr int N_R = 8, N_M = PREC == 1 ? 24 : 64, N_W = 4;                            // fragment reads / MFMAs of one half, weight loads of one set
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  auto pin_half_with_loads = [&]() __attribute__((always_inline)) {   // MFMAs interleaved with the half's 8 fragment reads, then its 4 weight loads
    pin_reads<N_M - 1, N_R>();
    constexpr int used = pin_read_slots(N_M - 1, N_R);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, N_W, 0);
    if constexpr (N_M - used - 1 > 0) __builtin_amdgcn_sched_group_barrier(0x008, N_M - used - 1, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto pin_half = [&]() __attribute__((always_inline)) {
    pin_reads<N_M, N_R>();
    if constexpr (N_M - pin_read_slots(N_M, N_R) > 0) __builtin_amdgcn_sched_group_barrier(0x008, N_M - pin_read_slots(N_M, N_R), 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  int kpar = 0;                                                        // parity of the stage the current chunk is read from

  // One skip tap.  Half A: MFMA(column-even groups) || read the column-odd groups' fragments, request the weights two taps ahead.  Half B:
  // MFMA(column-odd groups) || read the column-even fragments of the next tap.  The chunk's barrier sits between the halves of tap 8.
  // NEXT_UP (tap 7 / 8 only): the next chunk is the first low-resolution chunk (its pair 0 goes into sets 0 / 1), else a skip chunk.
  auto skip_tap = [&](auto TAP, int chunk, auto FIRST, auto NEXT_UP) __attribute__((always_inline)) {
    constexpr int tap = decltype(TAP)::value;
    constexpr bool next_up = decltype(NEXT_UP)::value;
    const char* cur = smem + kpar * STAGE;
    read_skip(fx1, cur, TAP, I1{});
    mfma_half(fx0, wq[tap % 3], I0{}, FIRST);
    if constexpr (tap + 2 < 9) load_w_skip(chunk, tap + 2, std::integral_constant<int, (tap + 2) % 3>{});
    else if constexpr (next_up) load_w_up(0, 0, tap - 7, std::integral_constant<int, tap - 7>{});
    else load_w_skip(chunk + 1, tap - 7, std::integral_constant<int, tap - 7>{});
    pin_half_with_loads();
    if constexpr (tap == 8) {
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      const char* nxt = smem + (kpar ^ 1) * STAGE;
      if constexpr (next_up) read_up(fx0, nxt, I0{}, I0{});
      else read_skip(fx0, nxt, I0{}, I0{});
    } else {
      read_skip(fx0, cur, std::integral_constant<int, tap + 1>{}, I0{});
    }
    mfma_half(fx1, wq[tap % 3], I1{}, FIRST);
    pin_half();
  };
  // One low-resolution pair (ty, tx) = two halves with their own weights (sets 2 (pair & 1), + 1); each half requests its counterpart of the
  // next pair into the other pair of sets.  LAST (pair 3 only): the next chunk is the next tile's first skip chunk (taps 0 / 1 into sets 0 / 1).
  auto up_pair = [&](auto PAIR, int chunk, auto LAST) __attribute__((always_inline)) {
    constexpr int pair = decltype(PAIR)::value;
    constexpr bool last = decltype(LAST)::value;
    constexpr int s0 = 2 * (pair & 1), n0s = 2 * ((pair + 1) & 1);
    const char* cur = smem + kpar * STAGE;
    read_up(fx1, cur, PAIR, I1{});
    mfma_half(fx0, wq[s0], I0{}, std::false_type{});
    if constexpr (pair < 3) load_w_up(chunk, pair + 1, 0, std::integral_constant<int, n0s>{});
    else if constexpr (last) load_w_skip(0, 0, std::integral_constant<int, n0s>{});
    else load_w_up(chunk + 1, 0, 0, std::integral_constant<int, n0s>{});
    pin_half_with_loads();
    if constexpr (pair == 3) {
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      const char* nxt = smem + (kpar ^ 1) * STAGE;
      if constexpr (last) read_skip(fx0, nxt, I0{}, I0{});
      else read_up(fx0, nxt, I0{}, I0{});
    } else {
      read_up(fx0, cur, std::integral_constant<int, pair + 1>{}, I0{});
    }
    mfma_half(fx1, wq[s0 + 1], I1{}, std::false_type{});
    if constexpr (pair < 3) load_w_up(chunk, pair + 1, 1, std::integral_constant<int, n0s + 1>{});
    else if constexpr (last) load_w_skip(0, 1, std::integral_constant<int, n0s + 1>{});
    else load_w_up(chunk + 1, 0, 1, std::integral_constant<int, n0s + 1>{});
    pin_half_with_loads();
  };
  auto skip_chunk = [&](int chunk, auto FIRST, auto NEXT_UP) __attribute__((always_inline)) {
    skip_tap(std::integral_constant<int, 0>{}, chunk, FIRST, std::false_type{});
    skip_tap(std::integral_constant<int, 1>{}, chunk, std::false_type{}, std::false_type{});
    skip_tap(std::integral_constant<int, 2>{}, chunk, std::false_type{}, std::false_type{});
    skip_tap(std::integral_constant<int, 3>{}, chunk, std::false_type{}, std::false_type{});
    skip_tap(std::integral_constant<int, 4>{}, chunk, std::false_type{}, std::false_type{});
    skip_tap(std::integral_constant<int, 5>{}, chunk, std::false_type{}, std::false_type{});
    skip_tap(std::integral_constant<int, 6>{}, chunk, std::false_type{}, std::false_type{});
    skip_tap(std::integral_constant<int, 7>{}, chunk, std::false_type{}, NEXT_UP);
    skip_tap(std::integral_constant<int, 8>{}, chunk, std::false_type{}, NEXT_UP);
    kpar ^= 1;
  };
  auto up_chunk = [&](int chunk, auto LAST) __attribute__((always_inline)) {
    up_pair(std::integral_constant<int, 0>{}, chunk, std::false_type{});
    up_pair(std::integral_constant<int, 1>{}, chunk, std::false_type{});
    up_pair(std::integral_constant<int, 2>{}, chunk, std::false_type{});
    up_pair(std::integral_constant<int, 3>{}, chunk, LAST);
    kpar ^= 1;
  };

  load_w_skip(0, 0, std::integral_constant<int, 0>{});
  load_w_skip(0, 1, std::integral_constant<int, 1>{});
  __syncthreads();                                                     // stage 0 holds chunk 0 of the first tile; `epi` is written
  read_skip(fx0, smem, I0{}, I0{});
  for (int ti = 0; ti < owned; ++ti) {
    int bx = __builtin_amdgcn_readfirstlane(tile_of(wgx, ti, G, ngrp));
    const int tx = bx % a.tiles_x; bx /= a.tiles_x;
    const int ty = bx % a.tiles_y;
    const int ey0 = ty * PH, ex0 = tx * PW;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const f32x4 iv = *reinterpret_cast<const f32x4*>(epi + wn * 32 + ct * 16 + 4 * g);
      initv[ct] = floatx4{iv[0], iv[1], iv[2], iv[3]};
    }
    skip_chunk(0, std::true_type{}, std::false_type{});                // (nsk >= 2: the first skip chunk is never the last one)
    for (int chunk = 1; chunk < nsk - 1; ++chunk) skip_chunk(chunk, std::false_type{}, std::false_type{});
    skip_chunk(nsk - 1, std::false_type{}, std::true_type{});
    for (int chunk = 0; chunk < nup - 1; ++chunk) up_chunk(chunk, std::false_type{});
    up_chunk(nup - 1, std::true_type{});
    // ---- epilogue: D[channel 4 g + j of tile ct][pixel p of group pt]: out = relu(acc + bias[row class][column class][channel]) (the BatchNorm
    // scale is in the weights, its shift was the accumulators' start value), then the tile goes to LDS for the loaders to store
    const int ccl0 = bias_class(ex0 + 2 * p, 2 * a.Wl), ccl1 = bias_class(ex0 + 2 * p + 1, 2 * a.Wl);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int rcl = bias_class(ey0 + 2 * (pt & 3) + wm, 2 * a.Hl);
      const float* bt = epi + 64 + (rcl * 4 + ((pt >> 2) ? ccl1 : ccl0)) * 64 + wn * 32 + 4 * g;
      const int m = wm * 128 + pt * 16 + p;
      char* row = outbuf + m * 256;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bt + ct * 16);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = acc[ct][pt][j] + bv[j];
          o[j] = a.relu ? fmaxf(v, 0.f) : v;
        }
        *reinterpret_cast<f32x4*>(row + (((wn * 8 + ct * 4 + g) ^ p) << 4)) = o;
      }
    }
  }
  __syncthreads();                                                     // the last tile's output is in LDS: the loaders store it
}

// ---- one-time weight preparation (mfpa_upconv_pack): the composite weights and the bias table, float64 accumulation
// w3 [9][Cout][Cs + Cu] (tap = 3 ky + kx, input channels contiguous), wt [4][Cu][Cl] (tap = 2 dy + dx), scale (Cout) or null
__global__ void up_composite_kernel(const float* __restrict__ w3, const float* __restrict__ wt, const float* __restrict__ scale, int Cout, int Cs, int Cu, int Cl,
                                    float* __restrict__ wc) {
  const int ci = blockIdx.x * blockDim.x + threadIdx.x, co = blockIdx.y, t16 = blockIdx.z;
  if (ci >= Cl) return;
  const int py = t16 >> 3, px = (t16 >> 2) & 1, ty = (t16 >> 1) & 1, tx = t16 & 1;
  double acc = 0.0;
  for (int ka = 0; ka < 3; ++ka) {
    const int ya = py + ka - 1;                                        // up-sampled row relative to the even row of this low-resolution row pair
    const int ry = (ya + 2) / 2 - 1;                                   // floor(ya / 2) for ya in -1 .. 2
    if (ry != ty - 1 + py) continue;
    for (int kb = 0; kb < 3; ++kb) {
      const int xa = px + kb - 1, rx = (xa + 2) / 2 - 1;
      if (rx != tx - 1 + px) continue;
      const float* wrow = w3 + ((size_t)(ka * 3 + kb) * Cout + co) * (Cs + Cu) + Cs;
      const float* wcol = wt + (size_t)(((ya & 1) * 2 + (xa & 1)) * Cu) * Cl + ci;
      for (int cu = 0; cu < Cu; ++cu) acc = fma((double)wrow[cu], (double)wcol[(size_t)cu * Cl], acc);
    }
  }
  wc[((size_t)t16 * Cout + co) * Cl + ci] = (float)(scale ? acc * (double)scale[co] : acc);
}

// bias_tab [4 row classes][4 column classes][Cout]: sum over the taps inside the up-sampled extent of W3u[tap][co][:] . bt
__global__ void up_bias_kernel(const float* __restrict__ w3, const float* __restrict__ bt, const float* __restrict__ scale, int Cout, int Cs, int Cu, float* __restrict__ tab) {
  const int co = blockIdx.x * blockDim.x + threadIdx.x, cls = blockIdx.y;
  if (co >= Cout) return;
  const int rc = cls >> 2, cc = cls & 3;
  // class 0: taps {0, +1} (index 1, 2); 1: all; 2: {-1, 0} (index 0, 1); 3: {-1} (index 0)
  const int lo_r = rc == 0 ? 1 : 0, hi_r = rc == 0 || rc == 1 ? 2 : (rc == 2 ? 1 : 0);
  const int lo_c = cc == 0 ? 1 : 0, hi_c = cc == 0 || cc == 1 ? 2 : (cc == 2 ? 1 : 0);
  double acc = 0.0;
  for (int ka = lo_r; ka <= hi_r; ++ka)
    for (int kb = lo_c; kb <= hi_c; ++kb) {
      const float* wrow = w3 + ((size_t)(ka * 3 + kb) * Cout + co) * (Cs + Cu) + Cs;
      for (int cu = 0; cu < Cu; ++cu) acc = fma((double)wrow[cu], (double)bt[cu], acc);
    }
  tab[(size_t)cls * Cout + co] = (float)(scale ? acc * (double)scale[co] : acc);
}

}  // namespace

bool conv_up_serves(int H, int W, int Hl, int Wl, int Cs, int Cl, int Cout) {
  if (Cout < 64 || Cout % 64 || Cs < 64 || Cs % KC || Cl < KC || Cl % KC) return false;
  if (W <= 16 || H < 8 || Hl < 2 || Wl < 2) return false;
  if (H - 2 * Hl < 0 || H - 2 * Hl > 1 || W - 2 * Wl < 0 || W - 2 * Wl > 1) return false;      // the reference's padding puts diff / 2 = 0 rows / columns in front
  // 32-bit byte offsets inside a clip; a slot outside the image is requested at 0xfffffff0 and an edge tile's wrapped halo origin (up to W + 1
  // pixels in front of the clip) plus a halo row's span must still lie beyond the clip
  if (4ull * H * W * Cs + 8ull * (W + 2) * (unsigned long long)Cs >= 0xfffffff0ull) return false;
  if (4ull * Hl * Wl * Cl + 8ull * (Wl + 2) * (unsigned long long)Cl >= 0xfffffff0ull) return false;
  if (4ull * H * W * Cout > 0xffffffffull) return false;
  if (64ull * Cout * Cl >= 0x7fffffffull || 36ull * Cout * Cs >= 0x7fffffffull) return false;   // the weight images' buffer descriptors and 32-bit block offsets
  return true;
}

int launch_conv_up(UpArgs& a, hipStream_t s) {
  if (!conv_up_serves(a.H, a.W, a.Hl, a.Wl, a.Cs, a.Cl, a.Cout)) return MFPA_EINVAL;
  a.tiles_x = (a.W + PW - 1) / PW;
  a.tiles_y = (a.H + PH - 1) / PH;
  const long long ntiles = (long long)a.tiles_x * a.tiles_y * a.B;
  if (ntiles > 0x7fffffffLL / 2) return MFPA_EINVAL;
  const size_t lds = 2 * (size_t)STAGE + EPI_FLOATS * sizeof(float) + (size_t)OUTBUF;
  static_assert(2 * STAGE + EPI_FLOATS * 4 + OUTBUF <= 160 * 1024, "LDS of a CU");
  const int cus = mfpa_current_device_cus();
  const unsigned gy = (unsigned)(a.Cout / 64);
  unsigned gx = (unsigned)(cus > 0 ? cus : 256) / gy;
  if (gx < 1) gx = 1;
  if ((long long)gx > ntiles) gx = (unsigned)ntiles;
  if (a.precision == 0) hipLaunchKernelGGL(conv_up_kernel<0>, dim3(gx * gy), dim3(THREADS), lds, s, a);
  else hipLaunchKernelGGL(conv_up_kernel<1>, dim3(gx * gy), dim3(THREADS), lds, s, a);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int launch_up_pack(const float* w3, const float* wt, const float* bt, const float* scale, int Cout, int Cs, int Cu, int Cl, float* wc16, float* bias_tab, hipStream_t s) {
  hipLaunchKernelGGL(up_composite_kernel, dim3((unsigned)((Cl + 255) / 256), (unsigned)Cout, 16u), dim3(256), 0, s, w3, wt, scale, Cout, Cs, Cu, Cl, wc16);
  MFPA_CHECK_LAUNCH();
  hipLaunchKernelGGL(up_bias_kernel, dim3((unsigned)((Cout + 63) / 64), 16u), dim3(64), 0, s, w3, bt, scale, Cout, Cs, Cu, bias_tab);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

}  // namespace mfpa_unet

// ------------------------------------------------------------------------------------------------------------------ C ABI (include/mfpa.h)
extern "C" int mfpa_upconv_serves(int H, int W, int Hl, int Wl, int Cs, int Cl, int Cout) {
  if (H < 1 || W < 1 || Hl < 1 || Wl < 1 || Cs < 1 || Cl < 1 || Cout < 1) return MFPA_EINVAL;
  return mfpa_unet::conv_up_serves(H, W, Hl, Wl, Cs, Cl, Cout) ? 1 : 0;
}

extern "C" int mfpa_upconv_pack(const float* w3, const float* wt, const float* bt, const float* scale, int Cout, int Cs, int Cu, int Cl, float* wc16,
                                float* bias_tab, void* stream) {
  if (!w3 || !wt || !bt || !wc16 || !bias_tab || Cout < 1 || Cs < 0 || Cu < 1 || Cl < 1 || Cout > 65535) return MFPA_EINVAL;
  return mfpa_unet::launch_up_pack(w3, wt, bt, scale, Cout, Cs, Cu, Cl, wc16, bias_tab, mfpa_stream(stream));
}

extern "C" int mfpa_upconv_fused(const mfpa_upconv_desc* d, void* stream) {
  if (!d) return MFPA_EINVAL;
  if (d->B == 0) return MFPA_OK;
  if (!d->skip || !d->low || !d->w_skip || !d->w_up || !d->shift || !d->bias_tab || !d->y || d->B < 0) return MFPA_EINVAL;
  mfpa_unet::UpArgs a{};
  a.skip = d->skip; a.low = d->low; a.w_skip = d->w_skip; a.w_up = d->w_up; a.shift = d->shift; a.bias_tab = d->bias_tab; a.y = d->y;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cs = d->Cs; a.Hl = d->Hl; a.Wl = d->Wl; a.Cl = d->Cl; a.Cout = d->Cout; a.relu = d->relu ? 1 : 0;
  if (d->precision != 0 && d->precision != 1) return MFPA_EINVAL;
  a.precision = d->precision;
  return mfpa_unet::launch_conv_up(a, mfpa_stream(stream));
}
