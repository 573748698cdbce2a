// UNet denoiser kernels for MI355X (gfx950), training/unet.py:8-108 of the reference.
//
// Activations are NHWC float32 (H = frequency bins, W = frames).  The 3x3 convolutions and the
// 2x2 transposed convolutions are implicit GEMMs on the matrix cores with float32-input MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate -- the reference is fp32):
//
//   M = output pixels of a PHxPW patch (128 per workgroup), N = output channels (64/128 per
//   workgroup), K = taps x input channels, walked in chunks of 32 channels.
//
// LDS im2col staging: per 32-channel chunk the workgroup stages the patch PLUS its one-pixel
// halo once ((PH+2)x(PW+2) pixels x 32 channels) and all nine taps read their shifted A
// fragments out of that tile, so the input is fetched 1.4-1.6x instead of 9x.  Both operands
// sit in LDS K-contiguous with rows padded to 36 floats: a lane's fragment for four consecutive
// MFMA k-steps is one conflict-free ds_read_b128.  Zero padding of the convolution, ragged
// patch edges and the decoder's pad+concat (Up.forward, unet.py:56-63) are all folded into
// the halo loader; the folded BatchNorm affine + ReLU run in the epilogue on the accumulators.
// Global loads of the next tap's weights (and next chunk's halo) are issued before the MFMA
// block of the current tap and written to LDS after it, so they overlap the matrix work.
#include "mfpa_common.h"
#include "mfpa_unet_args.h"

#include <cstdlib>
#include <type_traits>

namespace {

#ifdef MFPA_EXPERIMENTS
// In-kernel timeline (experiments build only; tools/exp_conv_timeline.py): every workgroup's wave 0 stamps s_memrealtime (100 MHz) at five points
// into a buffer whose address the tool stores in this device symbol.  No output value depends on a stamp.
__device__ unsigned long long* mfpa_conv_stamps = nullptr;
#define MFPA_STAMP(slot)                                                                                                   \
  do {                                                                                                                     \
    if (mfpa_conv_stamps && threadIdx.x == 0)                                                                              \
      mfpa_conv_stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime();   \
  } while (0)
#else
#define MFPA_STAMP(slot) do { } while (0)
#endif

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef MFPA_WD16_PERSIST_PLAIN
#define MFPA_WD16_PERSIST_PLAIN 1  // the same for the plain-bf16 (training) instantiations, whose MFMA time is a third: prologue / epilogue weigh three times more
#endif
#ifndef MFPA_WD16_PERSIST_ROWS
#define MFPA_WD16_PERSIST_ROWS 0   // the same for the ROWS inference form (A/B builds)
#endif
#ifndef MFPA_WD16_PERSIST2
#define MFPA_WD16_PERSIST2 1  // conv_wd16_kernel<.., WMW = 2>, tap-by-tap inference form: persistent tile loop like the 64-channel form (0: one tile per workgroup, A/B builds)
#endif
#ifndef MFPA_HALO_SPREAD
#define MFPA_HALO_SPREAD 0    // conv_wd16_kernel<.., WMW = 4>: staging slot k of the next chunk's halo requested at tap k (1) instead of all at tap 0 (A/B builds)
#endif
#ifndef MFPA_EPI_LDS
#define MFPA_EPI_LDS 1        // conv_wd16_kernel: the epilogue's per-channel constants from an LDS copy (0: global loads inside the epilogue, A/B builds)
#endif
#ifndef MFPA_CONV_PIPE
#define MFPA_CONV_PIPE 1      // 0: the round-1 main loop for the bf16x3 3x3 convolution too (A/B builds of tools/)
#endif
#ifndef MFPA_CONV_BIG_MIN_CIN
#define MFPA_CONV_BIG_MIN_CIN 64    // 128-channel tiles: the 8-wave 256-pixel shape from this many input channels on (256: round 1's choice)
#endif
#ifndef MFPA_CONV_PIPE4
#define MFPA_CONV_PIPE4 0     // 1: the pipelined loop also for the 4-wave 256 x 64 shape, one wave per SIMD (A/B builds)
#endif
#ifndef MFPA_CONV_PIPE_COND_A
#define MFPA_CONV_PIPE_COND_A 0   // 1: the pipelined loop skips halo slots outside the image too (A/B builds)
#endif
#ifndef MFPA_CONV_WN64
#define MFPA_CONV_WN64 1
#endif
#ifndef MFPA_CONV_STATIC_TAPS
#define MFPA_CONV_STATIC_TAPS 1     // 3x3 plain loop with compile-time taps (0: one runtime (chunk, tap) iteration, round 1's form)
#endif
#ifndef MFPA_CONV_MT4
#define MFPA_CONV_MT4 0             // 1: the 256 x 128 tile on FOUR waves of 128 px x 64 ch (8 x 32 patches); 2: also the 16 x 16 patches of the bottleneck
                                    // (measured per layer: 3-17 % SLOWER than the 8-wave shape -- a quarter less LDS traffic does not pay for one wave per SIMD)
#endif
#ifndef MFPA_CONV_LDS_EPI
#define MFPA_CONV_LDS_EPI 0         // 1: 64-channel tiles of the plain loop write their output (and the fused max-pool) through LDS as 16-byte pieces.
                                    // Correct (all GPU tests pass with it) and the in-kernel timeline shows the epilogue shrink 7.8 -> 5.6-6.3 us per
                                    // workgroup, but three same-call A/B pairs read 4082 / 4088 / 4091 vs 4091 / 4090 / 4093 clips/s: the stores of one
                                    // workgroup already overlap the partner workgroup's loop -- off.
#endif
#ifndef MFPA_CONV_WD16
#define MFPA_CONV_WD16 1            // >= 128-channel weights-direct layers on v_mfma_f32_16x16x32_bf16 (0: the 32 x 32 x 16 BDIR form of round 3's first half)
#endif
#ifndef MFPA_CONV_BDIR64
#define MFPA_CONV_BDIR64 0          // 1: weights-direct form also for 64-channel output tiles (8 waves of 64 px x 32 ch, one workgroup per CU): correct
                                    // (tests/test_gpu_unet.py runs it when enabled), +1.7 % per layer stand-alone but -2.8 % on the headline (3990 vs 4107 clips/s, two A/B pairs)
#endif
#ifndef MFPA_BDIR_SPREAD_SPLIT
#define MFPA_BDIR_SPREAD_SPLIT 1    // weights-direct kernels: the halo split one staging slot per tap inside the MFMA phases (0: one block at tap 2)
#endif
#ifndef MFPA_CONV_BOTTLENECK8
#define MFPA_CONV_BOTTLENECK8 1     // the 16x15 level on the 8-wave shape with 16x16-pixel patches (0: round 1's 4-wave 8x16 shape)
#endif

// PREC 1 weight image ("w3", built by ops_unet.split_bf16x3 / mfpa_pack_conv_weights): [tap][chunk][row][128 B], one row =
// the 32 channels of a chunk as 8 slots of 16 B, logical slots 0-3 = 32 bf16 hi, 4-7 = 32 bf16 lo, stored at PHYSICAL slot
// (logical ^ ((row >> 1) & 7)).  A (tap, chunk, 128-row) tile is 16 KB contiguous and is copied verbatim into LDS by LDS-DMA
// in the pipelined kernels (the XOR keeps a wave's ds_read_b128 fragment reads of 128-byte rows bank-conflict-free); the other
// kernels undo the XOR while they stage a tile through registers into their padded rows.
__device__ __forceinline__ int w3_swz(int row) { return (row >> 1) & 7; }

constexpr int KC = 32;        // channels per K chunk
constexpr int LDK = KC + 4;   // padded LDS row (floats): 144 B -> conflict-free b128 fragment reads

__device__ __forceinline__ bool v_never(float v) { return v != 12345.678f; }  // keeps the accumulators live in the 'skip stores' experiment

// sched_group_barrier pattern "one MFMA, then k LDS reads" with LEFT reads spread evenly over SLOTS MFMAs
template <int SLOTS, int LEFT, int I = 0>
__device__ __forceinline__ void pin_reads() {
  if constexpr (I < SLOTS && LEFT > 0) {
    constexpr int k = (LEFT + (SLOTS - I) - 1) / (SLOTS - I);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, k, 0);
    pin_reads<SLOTS, LEFT - k, I + 1>();
  }
}
constexpr int pin_read_slots(int slots, int left) {      // how many MFMAs pin_reads placed
  int used = 0;
  for (int i = 0; i < slots && left > 0; ++i) {
    left -= (left + (slots - i) - 1) / (slots - i);
    ++used;
  }
  return used;
}

// does this instantiation run the software-pipelined main loop (one workgroup per CU, two halo stages)?
constexpr bool conv_is_pipe(int BN, int PH, int PW, int WM, int WN, int MODE, int PREC, int MT = 2) {
  return (MFPA_CONV_PIPE != 0) && MODE == 0 && PREC == 1 &&
         (WM * WN * MT == 16 || (MFPA_CONV_PIPE4 != 0 && WM * WN == 4 && BN == 64 && PH * PW == 256));
}

using mfpa_unet::ConvArgs;     // csrc/mfpa_unet_args.h (shared with csrc/unet_ws.hip)

// MODE 0: 3x3 conv, pad 1 (9 taps, halo 1).
// MODE 1: 2x2 stride-2 transposed conv forward: one tap per workgroup column
//         (blockIdx.y = tap * (Cout/BN) + n-tile), output scattered to (2y+dy, 2x+dx).
// MODE 2: transposed-conv input gradient: 4 taps, A gathered from the (2H,2W) tensor at (2y+dy, 2x+dx).
// PREC 0: v_mfma_f32_32x32x2_f32 (exact fp32 products).
// PREC 1: "bf16x3" -- every fp32 operand is split x = hi + lo into two bf16 values and the product is
//         hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (fp32 accumulate): 3 matrix instructions at 16x
//         the fp32-MFMA rate, relative error ~2^-17 per product (UNet output: relative L1 ~2e-5, tolerance 1e-4).
//         An LDS row is [32 hi bf16 | 32 lo bf16 | pad] = the same 144 bytes as 32 floats + pad; the weights are
//         pre-split into that row format on the host, activations are split while they are staged.
// WM x WN waves, each 64 pixels x (BN/WN) channels: the workgroup tile is (64*WM pixels) x BN channels.
//
// Pipeline per (chunk, tap) iteration `it`, with the weight tile double-buffered in LDS and two register sets:
//     write B(it+1) registers -> Bs[(it+1)&1]   (loaded from L2/HBM during iteration it-1)
//     issue global loads of B(it+2) -> the other register set; at tap 0 also of the NEXT chunk's halo tile
//     MFMA block of iteration it from As / Bs[it&1]
//     one barrier                                 (+ barrier, halo store, at a chunk's last tap)
// so weight loads have two MFMA blocks to land, and the only exposed cost per iteration is the wave skew.
// MT_ = 32-pixel MFMA tiles per wave: 2 (a wave owns 64 pixels), or 4 -- 128 pixels x (BN / WN) channels per wave, FOUR waves for the
// 256 x 128 tile, one per SIMD with the whole register file (256 VGPRs + 206 AGPRs, no scratch): 12 fragment reads per 24 MFMAs
// instead of 16, a quarter less LDS traffic.  Measured 3-17 % slower per layer than two 64-pixel waves per SIMD (MFPA_CONV_MT4).
// BDIR ("weights direct"): the pipelined loop with NO weight tile in LDS.  8 waves = WM 2 x WN 4, a wave owns 128 pixels x 32
// channels (MT 4, NT 1): its B operand -- [2 substeps][hi, lo] fragments of ONE 32-channel column tile per (tap, chunk) -- comes
// straight from L1 / L2 into the MFMA operand registers out of a FRAGMENT-ORDERED weight image ("wf": [tap][chunk][Cout / 32]
// [substep][hi | lo][lane][16 B], built by ops_unet.split_bf16x3_frag; a wave-load is 1 KB contiguous), two iterations ahead
// through a ring of three register sets.  What that buys: no weight staging (2 global loads + 2 ds_write_b128 per thread and
// tap), no B-fragment LDS reads (the A side reads 8 ds_read_b128 per substep, as many as A + B did), and above all no barrier
// per tap -- the only LDS hand-off left is the halo tile, one barrier per 32-channel chunk (nine taps).  The column tile is
// fetched by the two waves that share it (wm 0 / 1): 32 KB per iteration through the CU's L1 instead of 16 KB.
template <int BN, int PH, int PW, int WM, int WN, int MODE, int PREC, bool C1SRC = false, int MT_ = 2, bool BDIR = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 8 || conv_is_pipe(BN, PH, PW, WM, WN, MODE, PREC, MT_)) ? 1 : 2) void conv_mfma_kernel(ConvArgs a) {
  constexpr int THREADS = 64 * WM * WN;
  constexpr int HALO = (MODE == 0) ? 1 : 0;
  constexpr int TAPS = (MODE == 0) ? 9 : (MODE == 2 ? 4 : 1);
  constexpr bool A_PER_TAP = (MODE == 2);               // A tile changes with the tap
  constexpr int HPW = PW + 2 * HALO, HPH = PH + 2 * HALO;
  constexpr int HP = HPW * HPH;                       // halo-tile pixels
  constexpr int BM = PH * PW;
  constexpr int MT = MT_;
  constexpr int WPX = 32 * MT;                        // pixels per wave
  static_assert(BM == WPX * WM, "workgroup tile is 32*MT*WM pixels");
  constexpr int NT = BN / (32 * WN);                  // 32-wide n tiles per wave
  constexpr int A_F4 = (HP * (KC / 4) + THREADS - 1) / THREADS;
  constexpr int B_F4 = (BN * (KC / 4) + THREADS - 1) / THREADS;
  constexpr bool B_EXACT = (BN * (KC / 4)) % THREADS == 0;

  // PIPE (the bf16x3 3x3 convolution on the 8-wave shapes, one workgroup per CU): software-pipelined main loop with the halo
  // tile double-buffered in LDS, see step_pipe below.  The 4-wave shapes keep the plain loop: with 256 threads the staging
  // registers are twice as many per thread and the second fragment set spills (measured: 2x slower).
  static_assert(!BDIR || (MODE == 0 && PREC == 1 && WM * WN == 8 && (MT_ == 4 || MT_ == 2) && BN == 32 * WN && !C1SRC),
                "BDIR: 8 waves of (32 MT) px x 32 ch, bf16x3 3x3 convolution");
  constexpr bool PIPE = BDIR || conv_is_pipe(BN, PH, PW, WM, WN, MODE, PREC, MT);
  constexpr int A_STAGES = PIPE ? 2 : 1;
  // PIPE: a halo stage has a row for every staging slot (A_F4 * THREADS / 8 >= HP), so the split / store pass needs no tail
  // predicate: every halo load is consumed on every path and hipcc keeps no "maybe pending" state across iterations
  constexpr int HPS = PIPE ? ((HP * (KC / 4) + THREADS - 1) / THREADS) * (THREADS / (KC / 4)) : HP;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* As = reinterpret_cast<float*>(smem);         // [A_STAGES][HPS][LDK]
  constexpr int B_STAGES = BDIR ? 0 : 2, B_STAGE = BN * LDK;     // weight-tile stages: [2][BN][LDK] padded rows, written through registers (BDIR: none)
  float* Bs0 = As + A_STAGES * HPS * LDK;
  constexpr int SW = PW + 4, SH = PH + 4;             // C1SRC: spectrogram patch with a 2-pixel halo, then the (9, 64) weights
  float* Sp = Bs0 + B_STAGES * B_STAGE;               // [SH][SW]
  float* W1s = Sp + SH * SW;                          // [9][64]

  MFPA_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int li = lane & 31, lh = lane >> 5;

  int bx = blockIdx.x;
  const int tx = bx % a.tiles_x; bx /= a.tiles_x;
  const int ty = bx % a.tiles_y; bx /= a.tiles_y;
  const int b = bx;
  const int n_tiles = a.Cout / BN;
  const int n_tile = (MODE == 1) ? blockIdx.y % n_tiles : blockIdx.y;
  const int ct_tap = (MODE == 1) ? blockIdx.y / n_tiles : 0;
  const int n0 = n_tile * BN;
  const int y0 = ty * PH, x0p = tx * PW;
  const int Cin = a.C0 + a.C1;
  const int nchunks = Cin / KC;
  const int nit = nchunks * TAPS;

  f32x4 areg[A_F4];
  f32x4 breg[3][B_F4];   // register sets for the weight tile (the pipelined loop uses three), always indexed with compile-time constants

  // The halo pixel of each of a thread's A_F4 staging slots never changes (idx = tid + it * THREADS -> pixel idx / 8, channel quad
  // idx % 8): its image coordinates are computed ONCE, packed (gy << 16 | gx), -1 = outside the image or past the tile.  The
  // per-chunk loader then needs two integer multiply-adds per load instead of the divisions / 64-bit index chains (~40
  // instructions per load, the largest block of vector work in the 64-channel layers).
  static_assert(THREADS % (KC / 4) == 0, "a thread keeps one channel quad for all of its halo pixels");
  const int aq = tid % (KC / 4);
  // packed as: bit 31 = outside the image (or past the tile), bits 30..16 / 15..0 = row / column CLAMPED into the image -- a slot
  // outside still loads (unconditionally, see load_a), from its nearest image pixel: a line its neighbours fetch anyway (every
  // outside slot reading ONE fixed address made that line a hot spot: +4-8 % on the 8-wave fp32 kernel)
  int apix[A_F4];
#pragma unroll
  for (int it = 0; it < A_F4; ++it) {
    const int pix = tid / (KC / 4) + it * (THREADS / (KC / 4));
    const int gy = y0 + pix / HPW - HALO, gx = x0p + pix % HPW - HALO;
    const bool in = pix < HP && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1);
    apix[it] = (in ? 0 : (int)0x80000000) | (cy << 16) | cx;
  }
  // 32-bit byte offsets from per-clip scalar bases (the host checks that one clip's input image fits 2 GB)
  const char* xb0 = reinterpret_cast<const char*>(a.x0) + (size_t)b * (MODE == 2 ? 4 : 1) * a.H * a.W * a.C0 * sizeof(float);
  const char* xb1 = reinterpret_cast<const char*>(a.x1) + (size_t)b * a.H1 * a.W1 * a.C1 * sizeof(float);
  // load_a only ISSUES the global loads of a halo tile (nothing in it reads a loaded value, so no wait lands between the
  // loads); the on-load affine + ReLU + dropout and the bf16 split happen in store_a, a whole chunk later.  Every slot loads
  // UNCONDITIONALLY -- a pixel outside the image reads its nearest image pixel and store_a zeroes it: no exec-mask branches, and
  // a wave issues exactly A_F4 loads per tile, which the pipelined loop's counted vmcnt waits rely on.
  auto src1_inside = [&](int p) __attribute__((always_inline)) {          // inside the (smaller, zero-padded) second source?
    const int y1 = ((p >> 16) & 0x7fff) - a.oy1, x1 = (p & 0xffff) - a.ox1;
    return p >= 0 && y1 >= 0 && y1 < a.H1 && x1 >= 0 && x1 < a.W1;
  };
  auto load_a = [&](int chunk, int tap) __attribute__((always_inline)) {
    if (C1SRC) return;                                 // the tile is computed from the LDS-resident spectrogram patch in store_a
    const int c0 = chunk * KC;
    const bool from0 = c0 < a.C0;
#pragma unroll
    for (int it = 0; it < A_F4; ++it) {
      const int gy = (apix[it] >> 16) & 0x7fff, gx = apix[it] & 0xffff;           // clamped into the image
      // PIPE: every slot loads, so that a wave issues exactly A_F4 loads per tile and hipcc's vmcnt bookkeeping stays exact; the
      // plain loop skips slots outside the image (measured: unconditional loads cost its fp32 8-wave form 4-8 %)
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (from0) {
        unsigned off;
        if (MODE == 2) off = ((unsigned)((2 * gy + (tap >> 1)) * (2 * a.W) + 2 * gx + (tap & 1)) * (unsigned)a.C0 + (unsigned)(c0 + 4 * aq)) * 4u;
        else off = ((unsigned)(gy * a.W + gx) * (unsigned)a.C0 + (unsigned)(c0 + 4 * aq)) * 4u;
        if ((PIPE && !MFPA_CONV_PIPE_COND_A) || apix[it] >= 0) v = *reinterpret_cast<const f32x4*>(xb0 + off);
      } else {
        const int y1 = min(max(gy - a.oy1, 0), a.H1 - 1), x1 = min(max(gx - a.ox1, 0), a.W1 - 1);
        const unsigned off = ((unsigned)(y1 * a.W1 + x1) * (unsigned)a.C1 + (unsigned)(c0 - a.C0 + 4 * aq)) * 4u;
        if ((PIPE && !MFPA_CONV_PIPE_COND_A) || src1_inside(apix[it])) v = *reinterpret_cast<const f32x4*>(xb1 + off);
      }
      areg[it] = v;
    }
  };
  auto store_a = [&](int chunk, float* As) __attribute__((always_inline)) {      // `As`: the halo stage to fill
    const int c0 = chunk * KC;
    const bool affine = a.in_scale0 != nullptr && c0 < a.C0;
    f32x4 a_sc = {1.f, 1.f, 1.f, 1.f}, a_sh = {0.f, 0.f, 0.f, 0.f};
    if (affine) {
      a_sc = *reinterpret_cast<const f32x4*>(a.in_scale0 + c0 + 4 * aq);
      a_sh = *reinterpret_cast<const f32x4*>(a.in_shift0 + c0 + 4 * aq);
    }
    // C1SRC: a thread keeps ONE channel quad for all of its halo pixels, so the first layer's nine weight quads and its folded
    // BatchNorm are read once per chunk, not once per staging slot (99 -> 9 LDS reads per chunk and thread)
    f32x4 c1w[C1SRC ? 9 : 1], c1s = {1.f, 1.f, 1.f, 1.f}, c1h = {0.f, 0.f, 0.f, 0.f};
    if constexpr (C1SRC) {
#pragma unroll
      for (int t = 0; t < 9; ++t) c1w[t] = *reinterpret_cast<const f32x4*>(W1s + t * 64 + c0 + 4 * aq);
      c1s = *reinterpret_cast<const f32x4*>(a.c1_scale + c0 + 4 * aq);
      c1h = *reinterpret_cast<const f32x4*>(a.c1_shift + c0 + 4 * aq);
    }
#pragma unroll
    for (int it = 0; it < A_F4; ++it) {
      const int pix = tid / (KC / 4) + it * (THREADS / (KC / 4)), q = aq;
      if (PIPE || pix < HP) {
        const bool inside = apix[it] >= 0;
        f32x4 v = areg[it];
        if (!C1SRC && !(c0 < a.C0 ? inside : src1_inside(apix[it]))) v = f32x4{0.f, 0.f, 0.f, 0.f};      // zero padding
        const int gy = (apix[it] >> 16) & 0x7fff, gx = apix[it] & 0xffff;
        if (C1SRC) {
          v = f32x4{0.f, 0.f, 0.f, 0.f};
          if (inside) {                                                    // conv2's zero padding stays exactly zero
            const float* sp = Sp + (pix / HPW) * SW + (pix % HPW);        // 3x3 window of the first layer around (gy, gx)
            // Packed FMAs on a sample broadcast out of the LOW half of a pair (mfpa_bcast2): written as `v += sv * wv` hipcc keeps two
            // samples in one register pair and selects the odd one with v_pk_fma_f32 ... op_sel:[1,0,0], the operand-selection form
            // the library does not ship (mfpa_common.h; tests/test_isa_scan.py).  One v_mov per sample; scalar FMAs in its place made
            // this kernel 18 % slower (7.0 instead of 5.8 vector instructions per MFMA).
            f32x2 v01 = {0.f, 0.f}, v23 = {0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
              const float sv = sp[(t / 3) * SW + (t % 3)];
              const f32x2 xx = mfpa_bcast2(sv);
              const f32x4 wv = c1w[t];
              v01 += xx * f32x2{wv.x, wv.y};
              v23 += xx * f32x2{wv.z, wv.w};
            }
            v = f32x4{v01[0], v01[1], v23[0], v23[1]};
            v = v * c1s + c1h;
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          }
        }
        if (!C1SRC && affine && inside) {      // padding stays exactly zero
          v = v * a_sc + a_sh;
          v.x = v.x > 0.f ? v.x : 0.f;
          v.y = v.y > 0.f ? v.y : 0.f;
          v.z = v.z > 0.f ? v.z : 0.f;
          v.w = v.w > 0.f ? v.w : 0.f;
          if (a.drop_thresh) {
            const unsigned long long e0 = (((unsigned long long)b * a.H + gy) * a.W + gx) * a.C0 + c0 + 4 * q;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = mfpa_keep(a.drop_seed, a.drop_thresh, e0 + k) ? v[k] * a.drop_scale : 0.f;
          }
        }
        if (PREC == 0) {
          *reinterpret_cast<f32x4*>(As + pix * LDK + 4 * q) = v;
        } else {
          bf16x4 hi, lo;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            hi[k] = (__bf16)v[k];
            lo[k] = (__bf16)(v[k] - (float)hi[k]);
          }
          char* row = reinterpret_cast<char*>(As + pix * LDK);
          *reinterpret_cast<bf16x4*>(row + 8 * q) = hi;
          *reinterpret_cast<bf16x4*>(row + 64 + 8 * q) = lo;
        }
      }
    }
  };
  // a thread's (row, quad) inside a weight tile never changes: 32-bit byte offsets computed once, added to a scalar tile base
  unsigned b_off[B_F4];
#pragma unroll
  for (int it = 0; it < B_F4; ++it) {
    const int idx = tid + it * THREADS;
    const int n = idx / (KC / 4), q = idx % (KC / 4);
    b_off[it] = (PREC == 0) ? (unsigned)(n * Cin + 4 * q) * 4u : (unsigned)(n * KC + 4 * (q ^ w3_swz(n))) * 4u;
  }
  auto load_b = [&](int it_flat, auto SET) __attribute__((always_inline)) {
    constexpr int set = decltype(SET)::value;
    const int chunk = it_flat / TAPS, tap = it_flat % TAPS;
    const int wt = (MODE == 1) ? ct_tap : tap;
    // PREC 0: [tap][Cout][Cin] floats; PREC 1: the chunk-major swizzled image (header of this file)
    const char* wbase = reinterpret_cast<const char*>((PREC == 0) ? a.w + ((size_t)wt * a.Cout + n0) * Cin + chunk * KC
                                                                 : a.w + (((size_t)wt * nchunks + chunk) * a.Cout + n0) * KC);
#pragma unroll
    for (int it = 0; it < B_F4; ++it)
      if (B_EXACT || tid + it * THREADS < BN * (KC / 4)) breg[set][it] = *reinterpret_cast<const f32x4*>(wbase + b_off[it]);
  };
  auto load_b_at = [&](int chunk, int tap, auto SET) __attribute__((always_inline)) {     // load_b with (chunk, tap) given: no division
    constexpr int set = decltype(SET)::value;
    const char* wbase = reinterpret_cast<const char*>((PREC == 0) ? a.w + ((size_t)tap * a.Cout + n0) * Cin + chunk * KC
                                                                 : a.w + (((size_t)tap * nchunks + chunk) * a.Cout + n0) * KC);
#pragma unroll
    for (int it = 0; it < B_F4; ++it)
      if (B_EXACT || tid + it * THREADS < BN * (KC / 4)) breg[set][it] = *reinterpret_cast<const f32x4*>(wbase + b_off[it]);
  };
  auto load_b_ct = [&](int chunk, int tap, auto SET) __attribute__((always_inline)) {     // PREC 1 image, (chunk, tap) given
    constexpr int set = decltype(SET)::value;
    const char* wbase = reinterpret_cast<const char*>(a.w + (((size_t)tap * nchunks + chunk) * a.Cout + n0) * KC);
#pragma unroll
    for (int it = 0; it < B_F4; ++it) breg[set][it] = *reinterpret_cast<const f32x4*>(wbase + b_off[it]);
  };
  auto store_b = [&](auto SET, float* Bs) __attribute__((always_inline)) {
    constexpr int set = decltype(SET)::value;
#pragma unroll
    for (int it = 0; it < B_F4; ++it) {
      const int idx = tid + it * THREADS;
      if (B_EXACT || idx < BN * (KC / 4)) {
        const int n = idx / (KC / 4), q = idx % (KC / 4);
        *reinterpret_cast<f32x4*>(Bs + n * LDK + 4 * q) = breg[set][it];
      }
    }
  };

  floatx16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  int a_base[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = wm * WPX + mt * 32 + li;
    a_base[mt] = ((m / PW) * HPW + (m % PW)) * LDK + 4 * lh;
  }
  int b_base[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) b_base[nt] = (wn * (NT * 32) + nt * 32 + li) * LDK + 4 * lh;

  // round 6: the plain-bf16 training step (mfpa_conv_desc.precision 2) runs the transposed convolution's INPUT GRADIENT (MODE 2) with one
  // bf16 MFMA per product too -- the hi halves of the same staged operands; wave-uniform
  const bool plain_hi = MODE == 2 && PREC == 1 && a.plain != 0;
  auto compute = [&](int tap_off, const float* Bs) __attribute__((always_inline)) {       // tap_off: LDS offset (floats) of the tap's shifted A fragments
    if (PREC == 1) {
      if (MODE == 2 && plain_hi) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8 ah[MT], bh[NT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) ah[mt] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(As + a_base[mt] + tap_off) + 32 * s);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bh[nt] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(Bs + b_base[nt]) + 32 * s);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
        }
        return;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const char* r = reinterpret_cast<const char*>(As + a_base[mt] + tap_off) + 32 * s;
          ah[mt] = *reinterpret_cast<const bf16x8*>(r);
          al[mt] = *reinterpret_cast<const bf16x8*>(r + 64);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const char* r = reinterpret_cast<const char*>(Bs + b_base[nt]) + 32 * s;
          bh[nt] = *reinterpret_cast<const bf16x8*>(r);
          bl[nt] = *reinterpret_cast<const bf16x8*>(r + 64);
        }
        // term-major order: consecutive matrix instructions write different accumulators (dependent distance MT*NT)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < KC / 8; ++s) {
        f32x4 af[MT], bf[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const f32x4*>(As + a_base[mt] + tap_off + 8 * s);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = *reinterpret_cast<const f32x4*>(Bs + b_base[nt] + 8 * s);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt].x, bf[nt].x, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt].y, bf[nt].y, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt].z, bf[nt].z, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt].w, bf[nt].w, acc[mt][nt], 0, 0, 0);
          }
      }
    }
  };

  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  // one pipeline iteration; register set CUR holds B(it+1), the other set receives B(it+2)
  auto step = [&](int it, auto CUR) __attribute__((always_inline)) {
    constexpr int cur = decltype(CUR)::value;
    const int tap = it % TAPS;
    const bool chunk_end = (tap == TAPS - 1);
    if (!MFPA_EXP_FLAG(a.dbg, 1)) {
      if (it + 1 < nit) store_b(CUR, Bs0 + ((it + 1) & 1) * (BN * LDK));
      if (it + 2 < nit) load_b(it + 2, std::integral_constant<int, 1 - cur>{});
    }
    if (!A_PER_TAP && tap == 0 && it + TAPS < nit) load_a(it / TAPS + 1, 0);   // next chunk's halo, a whole chunk ahead
    if (A_PER_TAP && it + 1 < nit) load_a((it + 1) / TAPS, (it + 1) % TAPS);
    if (!MFPA_EXP_FLAG(a.dbg, 8)) compute((MODE == 0) ? ((tap / 3) * HPW + (tap % 3)) * LDK : 0, Bs0 + (it & 1) * (BN * LDK));
    if ((chunk_end || A_PER_TAP) && it + 1 < nit) {
      if (!MFPA_EXP_FLAG(a.dbg, 2)) __syncthreads();            // every wave is done reading As
      store_a((it + 1) / TAPS, As);
    }
    if (!MFPA_EXP_FLAG(a.dbg, 2)) __syncthreads();
  };

  // The same iteration with the tap (and the parity of `it`, i.e. the weight register set and LDS stage) as compile-time constants:
  // the 3x3 convolution's plain loop then runs a chunk as nine straight-line taps -- no it / 9 and it % 9, no runtime tap offset
  // added to every fragment address (PMC on the 64-channel layers: 3.8 scalar and 4.8 vector instructions per MFMA with the
  // runtime form, their K is only 18 .. 36 iterations long)
  auto step_s = [&](auto TAP_, auto PAR_, int chunk) __attribute__((always_inline)) {
    constexpr int tap = decltype(TAP_)::value, par = decltype(PAR_)::value;
    constexpr int tap_off = ((tap / 3) * HPW + (tap % 3)) * LDK;
    const bool more = chunk + 1 < nchunks;                               // a chunk follows this one
    if (!MFPA_EXP_FLAG(a.dbg, 1)) {
      if (tap + 1 < TAPS || more) store_b(std::integral_constant<int, par>{}, Bs0 + (par ^ 1) * (BN * LDK));
      if (tap + 2 < TAPS) load_b_at(chunk, tap + 2, std::integral_constant<int, 1 - par>{});
      else if (more) load_b_at(chunk + 1, tap + 2 - TAPS, std::integral_constant<int, 1 - par>{});
    }
    if (tap == 0 && more) load_a(chunk + 1, 0);
    if (!MFPA_EXP_FLAG(a.dbg, 8)) compute(tap_off, Bs0 + par * (BN * LDK));
    if (tap == TAPS - 1 && more) {
      if (!MFPA_EXP_FLAG(a.dbg, 2)) __syncthreads();
      store_a(chunk + 1, As);
    }
    if (!MFPA_EXP_FLAG(a.dbg, 2)) __syncthreads();
  };
  auto chunk_s = [&](auto PAR0_, int chunk) __attribute__((always_inline)) {      // nine taps; PAR0 = parity of the chunk's first iteration
    constexpr int p0 = decltype(PAR0_)::value;
    step_s(std::integral_constant<int, 0>{}, std::integral_constant<int, p0>{}, chunk);
    step_s(std::integral_constant<int, 1>{}, std::integral_constant<int, p0 ^ 1>{}, chunk);
    step_s(std::integral_constant<int, 2>{}, std::integral_constant<int, p0>{}, chunk);
    step_s(std::integral_constant<int, 3>{}, std::integral_constant<int, p0 ^ 1>{}, chunk);
    step_s(std::integral_constant<int, 4>{}, std::integral_constant<int, p0>{}, chunk);
    step_s(std::integral_constant<int, 5>{}, std::integral_constant<int, p0 ^ 1>{}, chunk);
    step_s(std::integral_constant<int, 6>{}, std::integral_constant<int, p0>{}, chunk);
    step_s(std::integral_constant<int, 7>{}, std::integral_constant<int, p0 ^ 1>{}, chunk);
    step_s(std::integral_constant<int, 8>{}, std::integral_constant<int, p0>{}, chunk);
  };

  // ---- PIPE: the software-pipelined main loop of the bf16x3 3x3 convolution -------------------------------------------------
  // One (chunk, tap) iteration = two k-substeps of 16 channels, 12 MFMAs each.  The fragments of a substep (8 ds_read_b128 per
  // wave) are read one substep AHEAD into a second register set while the matrix pipe works on the current one, so no MFMA waits
  // on an LDS read it has just issued.  The iteration's one barrier sits BETWEEN its two substeps:
  //     phase A: read frags(it, s=1) -> F1  ||  MFMA(F0); behind the reads: request weight tile it+3, Bs[(it+1) & 1] <- tile it+1
  //     barrier  (tile it+1 visible; every fragment read of tile it has completed: its stage is rewritten in the next phase A)
  //     phase B: (tap 0: issue the loads of the next chunk's halo)  read frags(it+1, s=0) -> F0  ||  MFMA(F1)
  //              (tap 2: split that halo into the OTHER halo stage)
  // Each phase is one basic block whose MFMA : LDS : VMEM interleave is pinned with sched_group_barrier.
  struct Frags { bf16x8 ah[MT], al[MT], bh[NT], bl[NT]; };
  Frags fr0, fr1;
  auto read_frags = [&](Frags& f, const float* Asb, const float* Bsb, int tap_off, int sub) __attribute__((always_inline)) {
    // in the order the MFMAs consume them: (al, bh) terms first, then (ah, bl), then (ah, bh)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      f.al[mt] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(Asb + a_base[mt] + tap_off) + 32 * sub + 64);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      f.bh[nt] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(Bsb + b_base[nt]) + 32 * sub);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      f.ah[mt] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(Asb + a_base[mt] + tap_off) + 32 * sub);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      f.bl[nt] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(Bsb + b_base[nt]) + 32 * sub + 64);
  };
  auto mfma_lo = [&](const Frags& f) __attribute__((always_inline)) {      // the two correction terms: MT*NT*2 MFMAs
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[mt], f.bh[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mt], f.bl[nt], acc[mt][nt], 0, 0, 0);
  };
  auto mfma_hi = [&](const Frags& f) __attribute__((always_inline)) {      // the main term: MT*NT MFMAs
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mt], f.bh[nt], acc[mt][nt], 0, 0, 0);
  };
  constexpr int N_FR = 2 * (MT + NT), N_MFMA = 3 * MT * NT;     // fragment reads / MFMAs of one phase
  static_assert(N_MFMA >= 3, "a phase needs an MFMA in front of the weight-tile loads and one in front of its LDS stores");
  // the pinned interleave of a phase: the fragment reads spread evenly behind the first MFMAs; (phase A) the loads of the weight
  // tile three iterations ahead behind the next MFMA, the LDS stores of the next tile behind the one after; the remaining MFMAs last
  auto pin_phase = [&](auto WITH_B) __attribute__((always_inline)) {
    constexpr bool with_b = decltype(WITH_B)::value;
    constexpr int slots = with_b ? N_MFMA - 2 : N_MFMA;
    pin_reads<slots, N_FR>();
    constexpr int used = pin_read_slots(slots, N_FR) + (with_b ? 2 : 0);
    if constexpr (with_b) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, B_F4, 0);      // the loads of tile it+3 first: they touch no register of the stores
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, B_F4, 0);
    }
    if constexpr (N_MFMA - used > 0) __builtin_amdgcn_sched_group_barrier(0x008, N_MFMA - used, 0);
  };
  // One tap of the pipelined loop; TAP is a compile-time constant (the chunk loop calls it nine times), so every tap-dependent
  // choice -- LDS offsets of the shifted fragments, which of the three weight register sets is stored / refilled, where the halo
  // of the next chunk is loaded and split -- is static, a chunk is straight-line code, and hipcc's own s_waitcnt bookkeeping is
  // EXACT: the LDS stores of tile it+1 wait with vmcnt(B_F4 [+ A_F4]) and leave the loads of tile it+2 in flight.  (With a
  // runtime tap the conditional halo loads made it merge paths and wait vmcnt(0), i.e. also for the tile requested one
  // iteration ago: every iteration stalled on an L2 round trip; skipping just those loads returned 11-13 % on the deep layers.
  // Hiding the loads from hipcc instead -- an inline-asm register ring, and an LDS-DMA ring -- measured 7-15 % SLOWER.)
  static_assert(!PIPE || (B_EXACT && TAPS == 9), "tile it lives in register set it % 3 = tap % 3");
  auto tap_body = [&](auto TAP, int chunk) __attribute__((always_inline)) {
    constexpr int tap = decltype(TAP)::value;
    constexpr int ntap = (tap + 1) % TAPS;
    constexpr int tap_off = ((tap / 3) * HPW + (tap % 3)) * LDK, ntap_off = ((ntap / 3) * HPW + (ntap % 3)) * LDK;
    const int it = chunk * TAPS + tap;
    const int chunk_n = chunk + 1 < nchunks ? chunk + 1 : chunk;             // past the last chunk: the same halo again, into the stage nobody reads
    const float* Asb = As + (chunk & 1) * (HPS * LDK);
    const float* Asn = (tap == TAPS - 1) ? As + ((chunk + 1) & 1) * (HPS * LDK) : Asb;
    const float* Bsb = Bs0 + (it & 1) * B_STAGE;
    float* Bsn = Bs0 + ((it + 1) & 1) * B_STAGE;
    // ---- phase A: read frags(it, s=1) -> F1 || MFMA(F0); Bs[(it+1) & 1] <- tile it+1 (set (tap+1) % 3, requested two iterations
    //      ago; that stage's tile it-1 was last read before the previous barrier); request tile it+3 into set tap % 3
    read_frags(fr1, Asb, Bsb, tap_off, 1);
    mfma_lo(fr0);
    mfma_hi(fr0);
    if (!MFPA_EXP_FLAG(a.dbg, 1)) {
      store_b(std::integral_constant<int, (tap + 1) % 3>{}, Bsn);
      constexpr int t3 = (tap + 3) % TAPS;
      const int c3 = (tap + 3 >= TAPS) ? chunk_n : chunk;
      load_b_ct(c3, t3, std::integral_constant<int, tap % 3>{});
    }
    pin_phase(std::true_type{});
    __builtin_amdgcn_sched_barrier(0);
    if (!MFPA_EXP_FLAG(a.dbg, 2)) __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase B: (tap 0: issue the loads of the next chunk's halo)  read frags(it+1, s=0) -> F0 || MFMA(F1)
    //      (tap 2: split that halo into the OTHER halo stage)
    if (tap == 0 && !MFPA_EXP_FLAG(a.dbg, 16)) load_a(chunk_n, 0);
    read_frags(fr0, Asn, Bsn, ntap_off, 0);          // past the end: a harmless read of valid LDS
    mfma_lo(fr1);
    mfma_hi(fr1);
    pin_phase(std::false_type{});
    __builtin_amdgcn_sched_barrier(0);
    if (tap == 2 && !MFPA_EXP_FLAG(a.dbg, 16)) store_a(chunk_n, As + ((chunk + 1) & 1) * (HPS * LDK));
  };

  // ---- BDIR: weights straight from L1 / L2 into the operand registers (see the template's header comment) -------------------
  struct AFrags { bf16x8 ah[MT], al[MT]; };
  AFrags fa0, fa1;
  bf16x8 bq[3][2][2];                                  // [ring slot = tap % 3][substep][hi, lo]
  const int wn_s = __builtin_amdgcn_readfirstlane(wn);
  auto load_bq = [&](int chunk, int tap, auto SLOT) __attribute__((always_inline)) {
    constexpr int slot = decltype(SLOT)::value;
    // 4 KB per (tap, chunk, 32-channel column tile): [substep][hi | lo][lane][16 B]
    const char* wb = reinterpret_cast<const char*>(a.w) + ((((size_t)tap * nchunks + chunk) * (size_t)(a.Cout / 32) + (size_t)(n0 / 32 + wn_s)) << 12) + lane * 16;
    bq[slot][0][0] = *reinterpret_cast<const bf16x8*>(wb);
    bq[slot][0][1] = *reinterpret_cast<const bf16x8*>(wb + 1024);
    bq[slot][1][0] = *reinterpret_cast<const bf16x8*>(wb + 2048);
    bq[slot][1][1] = *reinterpret_cast<const bf16x8*>(wb + 3072);
  };
  auto read_afrags = [&](AFrags& f, const float* Asb, int tap_off, int sub) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      f.al[mt] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(Asb + a_base[mt] + tap_off) + 32 * sub + 64);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      f.ah[mt] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(Asb + a_base[mt] + tap_off) + 32 * sub);
  };
  auto mfma_d = [&](const AFrags& f, const bf16x8 bh, const bf16x8 bl) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[mt], bh, acc[mt][0], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mt], bl, acc[mt][0], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mt], bh, acc[mt][0], 0, 0, 0);
  };
  // one halo staging slot of chunk `chunk`: zero padding, the on-load affine + ReLU + dropout of the training forward, bf16 hi / lo
  // split, two 8-byte LDS stores (store_a for a single slot)
  auto split_slot = [&](auto IT, int chunk, float* Asn) __attribute__((always_inline)) {
    constexpr int it = decltype(IT)::value;
    const int pix = tid / (KC / 4) + it * (THREADS / (KC / 4));
    const int c0 = chunk * KC;
    const bool inside = (c0 < a.C0) ? (apix[it] >= 0) : src1_inside(apix[it]);
    f32x4 v = areg[it];
    if (!inside) v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.in_scale0 != nullptr && c0 < a.C0 && inside) {           // training: the producer's BatchNorm + ReLU (+ dropout) on load; padding stays 0
      const f32x4 sc = *reinterpret_cast<const f32x4*>(a.in_scale0 + c0 + 4 * aq);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(a.in_shift0 + c0 + 4 * aq);
      v = v * sc + sh;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
      if (a.drop_thresh) {
        const int gy = (apix[it] >> 16) & 0x7fff, gx = apix[it] & 0xffff;
        const unsigned long long e0 = (((unsigned long long)b * a.H + gy) * a.W + gx) * a.C0 + c0 + 4 * aq;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = mfpa_keep(a.drop_seed, a.drop_thresh, e0 + k) ? v[k] * a.drop_scale : 0.f;
      }
    }
    bf16x4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hi[k] = (__bf16)v[k];
      lo[k] = (__bf16)(v[k] - (float)hi[k]);
    }
    char* row = reinterpret_cast<char*>(Asn + pix * LDK);
    *reinterpret_cast<bf16x4*>(row + 8 * aq) = hi;
    *reinterpret_cast<bf16x4*>(row + 64 + 8 * aq) = lo;
  };
  // One tap: phase A = read A frags(it, s=1) || MFMA(s=0), request the weights of tile it+2 (slot (tap+2) % 3, whose tile it-1 was
  // consumed by the previous phase B); phase B = read A frags(it+1, s=0) || MFMA(s=1).  The chunk's ONE barrier sits between the
  // phases of tap 8: before it every wave has read the last fragments of this chunk's halo stage (rewritten at tap 2 of the next
  // chunk), behind it the next chunk's stage -- split at tap 2 of this chunk by every wave -- is complete.
  auto tap_body_d = [&](auto TAP, int chunk) __attribute__((always_inline)) {
    constexpr int tap = decltype(TAP)::value;
    constexpr int ntap = (tap + 1) % TAPS;
    constexpr int tap_off = ((tap / 3) * HPW + (tap % 3)) * LDK, ntap_off = ((ntap / 3) * HPW + (ntap % 3)) * LDK;
    constexpr int N_AR = 2 * MT, N_M = 3 * MT;
    const int chunk_n = chunk + 1 < nchunks ? chunk + 1 : chunk;
    const float* Asb = As + (chunk & 1) * (HPS * LDK);
    const float* Asn = (tap == TAPS - 1) ? As + ((chunk + 1) & 1) * (HPS * LDK) : Asb;
    read_afrags(fa1, Asb, tap_off, 1);
    mfma_d(fa0, bq[tap % 3][0][0], bq[tap % 3][0][1]);
    {
      constexpr int t2 = (tap + 2) % TAPS;
      load_bq((tap + 2 >= TAPS) ? chunk_n : chunk, t2, std::integral_constant<int, (tap + 2) % 3>{});
    }
    // pinned interleave: one fragment read behind each of the first MFMAs, the four weight loads behind the next, the rest bare
    pin_reads<N_M - 1, N_AR>();
    constexpr int used_a = pin_read_slots(N_M - 1, N_AR);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
    if constexpr (N_M - used_a - 1 > 0) __builtin_amdgcn_sched_group_barrier(0x008, N_M - used_a - 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (tap == TAPS - 1) {
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (tap == 0) load_a(chunk_n, 0);
    read_afrags(fa0, Asn, ntap_off, 0);
    mfma_d(fa1, bq[tap % 3][1][0], bq[tap % 3][1][1]);
#if MFPA_BDIR_SPREAD_SPLIT
    // The next chunk's halo (requested at tap 0) is split and stored ONE staging slot per tap, taps 2 .. 2 + A_F4 - 1, inside this
    // phase's scheduling region: its ~20 vector instructions and two LDS stores ride in the MFMA gaps instead of forming a block
    // of vector work that both waves of a SIMD reach together (they run the same program almost in lockstep) with the matrix
    // pipe idle.  All of it lies before tap 8's barrier.
    static_assert(A_F4 <= TAPS - 3, "one halo staging slot per tap, taps 2 .. 7");
    if constexpr (tap >= 2 && tap - 2 < A_F4) {
      split_slot(std::integral_constant<int, tap - 2>{}, chunk_n, As + ((chunk + 1) & 1) * (HPS * LDK));
      // MFMA, fragment read, two vector instructions ... then the stores behind two more MFMAs
#pragma unroll
      for (int i = 0; i < N_AR; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      if constexpr (N_M - N_AR - 2 > 0) __builtin_amdgcn_sched_group_barrier(0x008, N_M - N_AR - 2, 0);
    } else
#endif
    {
      pin_reads<N_M, N_AR>();
      if constexpr (N_M - pin_read_slots(N_M, N_AR) > 0) __builtin_amdgcn_sched_group_barrier(0x008, N_M - pin_read_slots(N_M, N_AR), 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#if !MFPA_BDIR_SPREAD_SPLIT
    if (tap == 2) store_a(chunk_n, As + ((chunk + 1) & 1) * (HPS * LDK));
#endif
  };

  if (C1SRC) {
    const double den = a.c1_denom ? a.c1_denom[b] : 1.0;
    for (int i = tid; i < SH * SW; i += THREADS) {
      const int gy = y0 - 2 + i / SW, gx = x0p - 2 + i % SW;
      float v = 0.f;                                   // the first layer's own zero padding
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
        const size_t o = ((size_t)b * a.H + gy) * a.W + gx;
        v = a.c1_spec64 ? (float)(a.c1_spec64[o] / den) : a.c1_x32[o];
      }
      Sp[i] = v;
    }
    for (int i = tid; i < 9 * 64; i += THREADS) W1s[i] = a.c1_w[i];
    __syncthreads();
  }
  MFPA_STAMP(1);                                       // C1SRC: the spectrogram patch and the first layer's weights are staged
  load_a(0, 0);
  if constexpr (BDIR) {
    load_bq(0, 0, Set0{});
    load_bq(0, 1, Set1{});
    store_a(0, As);
    __syncthreads();
    MFPA_STAMP(2);
    read_afrags(fa0, As, 0, 0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      tap_body_d(std::integral_constant<int, 0>{}, chunk);
      tap_body_d(std::integral_constant<int, 1>{}, chunk);
      tap_body_d(std::integral_constant<int, 2>{}, chunk);
      tap_body_d(std::integral_constant<int, 3>{}, chunk);
      tap_body_d(std::integral_constant<int, 4>{}, chunk);
      tap_body_d(std::integral_constant<int, 5>{}, chunk);
      tap_body_d(std::integral_constant<int, 6>{}, chunk);
      tap_body_d(std::integral_constant<int, 7>{}, chunk);
      tap_body_d(std::integral_constant<int, 8>{}, chunk);
    }
  } else if constexpr (PIPE) {
    using Set2 = std::integral_constant<int, 2>;
    load_b_ct(0, 0, Set0{});
    store_a(0, As);
    store_b(Set0{}, Bs0);
    load_b_ct(0, 1, Set1{});                           // tiles 1 and 2: sets 1 and 2 (tile it lives in set it % 3)
    load_b_ct(0, 2, Set2{});
    __syncthreads();
    read_frags(fr0, As, Bs0, 0, 0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      tap_body(std::integral_constant<int, 0>{}, chunk);
      tap_body(std::integral_constant<int, 1>{}, chunk);
      tap_body(std::integral_constant<int, 2>{}, chunk);
      tap_body(std::integral_constant<int, 3>{}, chunk);
      tap_body(std::integral_constant<int, 4>{}, chunk);
      tap_body(std::integral_constant<int, 5>{}, chunk);
      tap_body(std::integral_constant<int, 6>{}, chunk);
      tap_body(std::integral_constant<int, 7>{}, chunk);
      tap_body(std::integral_constant<int, 8>{}, chunk);
    }
  } else {
    load_b(0, Set0{});
    store_a(0, As);
    store_b(Set0{}, Bs0);
    if (nit > 1) load_b(1, Set0{});
    __syncthreads();
    MFPA_STAMP(2);                                     // first halo tile (C1SRC: the first layer on it) and weight tile in LDS
    if constexpr (MODE == 0 && MFPA_CONV_STATIC_TAPS != 0) {
      // nine is odd: the parity of a chunk's first iteration alternates from chunk to chunk
      int chunk = 0;
      for (; chunk + 1 < nchunks; chunk += 2) {
        chunk_s(Set0{}, chunk);
        chunk_s(Set1{}, chunk + 1);
      }
      if (chunk < nchunks) chunk_s(Set0{}, chunk);
    } else {
      for (int it = 0; it < nit; it += 2) {
        step(it, Set0{});
        if (it + 1 < nit) step(it + 1, Set1{});
      }
    }
  }

  MFPA_STAMP(3);                                       // main loop done
  // epilogue: out = relu(acc * scale[n] + shift[n]); D[row = pixel][col = channel]
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + wn * (NT * 32) + nt * 32 + li;
    const float sc = a.scale ? a.scale[n] : 1.f;
    const float sh = a.shift ? a.shift[n] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[mt][nt][r] * sc + sh;
        if (a.relu) v = v > 0.f ? v : 0.f;
        acc[mt][nt][r] = v;
      }
  }
  // 64-channel tiles of the plain loop (the full-resolution layers: 64 KB of output + 16 KB of pooled output per workgroup): the tile goes
  // through LDS -- free once every wave has left the main loop -- and out as 16-byte pieces, a pixel's 64 channels by 16 adjacent lanes:
  // 16 dwordx4 stores per thread instead of 64 scalar ones (the in-kernel timeline put the epilogue at 7.8 us of a 34-56 us workgroup
  // lifetime), and the 2x2 max-pool reads its windows from the same tile.
  constexpr bool LDS_EPI = MFPA_CONV_LDS_EPI && (MODE == 0 && !PIPE && BN == 64 && BM == 256 && PW == 32);
  if constexpr (LDS_EPI) {
    if (a.y != nullptr || a.y_pool != nullptr) {
      float* T = reinterpret_cast<float*>(smem);                       // [BM][BN]
      __syncthreads();
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = wm * WPX + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) T[m * BN + wn * (NT * 32) + nt * 32 + li] = acc[mt][nt][r];
        }
      __syncthreads();
      constexpr int Q = BN / 4;                                        // 16-byte pieces per pixel
      if (a.y != nullptr && !MFPA_EXP_FLAG(a.dbg, 4)) {
        char* yb = reinterpret_cast<char*>(a.y + (size_t)b * a.yH * a.yW * a.Cout);
#pragma unroll
        for (int it = 0; it < BM * Q / THREADS; ++it) {
          const int idx = tid + it * THREADS, p = idx / Q, q = idx % Q;
          const int gy = y0 + p / PW, gx = x0p + p % PW;
          if (gy < a.yH && gx < a.yW)
            *reinterpret_cast<f32x4*>(yb + (((unsigned)gy * (unsigned)a.yW + (unsigned)gx) * (unsigned)a.Cout + (unsigned)(n0 + 4 * q)) * 4u) =
                *reinterpret_cast<const f32x4*>(T + p * BN + 4 * q);
        }
      }
      if (a.y_pool != nullptr) {
        const int Ho = a.H / 2, Wo = a.W / 2;
#pragma unroll
        for (int it = 0; it < (BM / 4) * Q / THREADS; ++it) {
          const int idx = tid + it * THREADS, pp = idx / Q, q = idx % Q;
          const int ly = pp / (PW / 2), lx = pp % (PW / 2);
          const int py = y0 / 2 + ly, px = x0p / 2 + lx;
          if (py < Ho && px < Wo) {
            const float* t0 = T + ((2 * ly) * PW + 2 * lx) * BN + 4 * q;
            const f32x4 v00 = *reinterpret_cast<const f32x4*>(t0), v01 = *reinterpret_cast<const f32x4*>(t0 + BN);
            const f32x4 v10 = *reinterpret_cast<const f32x4*>(t0 + PW * BN), v11 = *reinterpret_cast<const f32x4*>(t0 + PW * BN + BN);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = fmaxf(fmaxf(v00[k], v01[k]), fmaxf(v10[k], v11[k]));
            *reinterpret_cast<f32x4*>(a.y_pool + (((size_t)b * Ho + py) * Wo + px) * a.Cout + n0 + 4 * q) = o;
          }
        }
      }
    }
  }
  if (!LDS_EPI && a.y != nullptr) {
    // 32-bit byte offsets from a scalar per-clip base (the host checks that one clip's output fits 4 GB), the pixel offset
    // computed once for all of a lane's channels, and no bounds checks on interior tiles: the first form of this loop (64-bit
    // index arithmetic and an exec-mask branch per element) was up to 13 % of the 64-channel layers
    const unsigned oW = (MODE == 1) ? 2u * (unsigned)a.W : (unsigned)a.yW;
    const unsigned oH = (MODE == 1) ? 2u * (unsigned)a.H : (unsigned)a.yH;
    char* yb = reinterpret_cast<char*>(a.y + (size_t)b * oH * oW * a.Cout);
    const unsigned cout = (unsigned)a.Cout;
    const unsigned nb = (unsigned)(n0 + wn * (NT * 32) + li) * 4u;
    const bool interior = (y0 + PH <= a.yH) && (x0p + PW <= a.yW);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = wm * WPX + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int gy = y0 + m / PW, gx = x0p + m % PW;
        if (interior || (gy < a.yH && gx < a.yW)) {
          unsigned pix;
          if (MODE != 1) pix = (unsigned)gy * oW + (unsigned)gx;
          else pix = (unsigned)(2 * gy + (ct_tap >> 1)) * oW + (unsigned)(2 * gx + (ct_tap & 1));
          char* yp = yb + (pix * cout * 4u + nb);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            if (!(MFPA_EXP_FLAG(a.dbg, 4) && v_never(acc[mt][nt][r]))) *reinterpret_cast<float*>(yp + nt * 128) = acc[mt][nt][r];
        }
      }
    }
  }
  if (!LDS_EPI && MODE == 0 && a.y_pool != nullptr) {
    // MaxPool2d(2) (floor): every 2x2 window lives in ONE lane's accumulators (the two rows of a window are the
    // wave's two 32-pixel MFMA tiles for 32-wide patches, registers r / r+8 for 16-wide ones; the two columns are
    // registers r / r+1), so pooling needs no cross-lane traffic.
    const int Ho = a.H / 2, Wo = a.W / 2;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + wn * (NT * 32) + nt * 32 + li;
      if (PW == 32) {
#pragma unroll
        for (int mp = 0; mp < MT; mp += 2)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const int col = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int py = (y0 + wm * MT + mp) / 2, px = (x0p + col) / 2;
            if (py < Ho && px < Wo) {
              const float v = fmaxf(fmaxf(acc[mp][nt][r], acc[mp][nt][r + 1]), fmaxf(acc[mp + 1][nt][r], acc[mp + 1][nt][r + 1]));
              a.y_pool[(((size_t)b * Ho + py) * Wo + px) * a.Cout + n] = v;
            }
          }
      } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 8; r += 2) {
            const int q = (r & 3) + 8 * (r >> 2) + 4 * lh;       // pixel in the 2x16 tile, row 0
            const int py = (y0 + 2 * (wm * MT + mt)) / 2, px = (x0p + (q % 16)) / 2;
            if (py < Ho && px < Wo) {
              const float v = fmaxf(fmaxf(acc[mt][nt][r], acc[mt][nt][r + 1]), fmaxf(acc[mt][nt][r + 8], acc[mt][nt][r + 9]));
              a.y_pool[(((size_t)b * Ho + py) * Wo + px) * a.Cout + n] = v;
            }
          }
      }
    }
  }
  if (MODE == 0 && WN <= 2 && a.w1x1 != nullptr) {
    // OutConv 1x1 to one class: a wave holds BN / WN channels of its pixels, one channel per lane of a 32-lane half.  The channel
    // sum is a DPP reduction (row_shr 1 / 2 / 4 / 8, then row_bcast:15 into the odd rows: five vector adds, no LDS crossbar -- the
    // ds_bpermute butterfly of __shfl_xor cost 126 waits per workgroup here), the totals of lanes 31 / 63 meet in LDS (the staging
    // buffers are free once every wave has left the main loop; with WN == 2 two waves contribute to a pixel) and are written out
    // one pixel per thread, coalesced along the patch rows.
    float wv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wv[nt] = a.w1x1[n0 + wn * (NT * 32) + nt * 32 + li];
    float* red = As;                                     // [WN][BM] channel sums
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) p += acc[mt][nt][r] * wv[nt];
        p += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p), 0x111, 0xf, 0xf, false));      // row_shr:1
        p += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p), 0x112, 0xf, 0xf, false));      // row_shr:2
        p += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p), 0x114, 0xf, 0xf, false));      // row_shr:4
        p += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p), 0x118, 0xf, 0xf, false));      // row_shr:8: lane 15 of a row = its total
        p += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p), 0x142, 0xa, 0xf, false));      // row_bcast:15 into rows 1 and 3
        const int m = wm * WPX + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (li == 31) red[wn * BM + m] = p;
      }
    __syncthreads();
    for (int m = tid; m < BM; m += THREADS) {
      const int gy = y0 + m / PW, gx = x0p + m % PW;
      const float p = red[m] + (WN == 2 ? red[BM + m] : 0.f);
      if (gy < a.H && gx < a.W) a.y1x1[((size_t)b * a.H + gy) * a.W + gx] = p + a.b1x1;
    }
  }
  MFPA_STAMP(4);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Weights-direct 3x3 convolution on v_mfma_f32_16x16x32_bf16 ("WD16", mfpa_conv_desc.w_layout 2).  Same structure as the BDIR form of
// conv_mfma_kernel -- 8 waves = 2 pixel halves x 4 column tiles, a wave owns 128 pixels x 32 channels, the weight operand straight from
// L1 / L2 out of a fragment-ordered image two taps ahead through a ring of three register sets, the halo tile double-buffered in LDS,
// ONE barrier per 32-channel chunk, the next chunk's halo split one staging slot per tap inside the MFMA phases -- but the matrix
// instruction is the 16 x 16 x 32 one: a whole 32-channel chunk is ONE k-step, and a quarter of the accumulator rows per instruction.
// The weights-direct loop is clock (power) limited, and under the same loop with the same operand traffic the chip holds a higher clock
// on this shape: a timing experiment with the 32 x 32 x 16 instructions of the BDIR kernel replaced one for two (wrong results) ran the
// eleven >= 128-channel layers in 11.07-11.10 ms instead of 11.93-12.00 ms (64 clips), which is what this kernel is built on.
//   roles: A operand = weights (16 output channels x 32 k), B operand = pixels (32 k x 16 pixels), so D[channel][pixel]: a lane holds FOUR
//          CONSECUTIVE CHANNELS of one pixel -- the epilogue stores 16-byte pieces (16 stores per wave instead of 64 scalar ones) and the
//          2 x 2 max-pool needs one DPP swap of adjacent lanes;
//   LDS:   per halo stage eight planes [hi | lo][k-group 0..3] of (pixel x 16 B): a fragment read is 16 consecutive pixels of one plane per
//          k-group -- conflict-free for ds_read_b128's lane groups exactly when the k-group planes are a multiple of 256 B apart; planes 2, 3
//          sit another 128 B further so that the split's 8-byte stores conflict 2-way instead of 4-way.
//   image: [tap][chunk = Cin / 32][Cout / 16][hi | lo][lane 64][16 B], lane (g = l >> 4, c = l & 15) = channel 16 t + c, k 32 chunk + 8 g .. + 7
//          (ops_unet.split_bf16x3_frag(w, 2), mfpa_pack_conv_weights(precision 3)).
// Not bit-identical to the 32 x 32 x 16 kernels (a k-step sums 32 products inside the instruction); same products, fp32 accumulate.
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dpp_row_add(float v) {               // v + (v of the lane CTRL names; 0 where there is none)
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

template <int PH, int PW, bool ROWS, int WMW = 2, bool SIDE = false, bool PLAIN = false, bool IN16 = false, bool AFF16 = false>
__global__ __launch_bounds__(512, 1) void conv_wd16_kernel(ConvArgs a) {
  // AFF16 (with IN16 and SIDE): the bfloat16 source 0 carries an on-load affine + ReLU (+ dropout) and the output may leave as bfloat16 only --
  // the training FORWARD with its activations kept as bfloat16; the input-gradient launches (IN16 without it) carry none of that code
  // PLAIN: plain bf16 products -- one MFMA per product on the hi halves only (the lo planes, their fragment reads, the lo weight
  // fragments and two of the three MFMA terms are gone): the training step's "bf16 MFMA" arithmetic (BASELINE config 4), relative
  // error ~2^-9 per product instead of bf16x3's 2^-17.  Never used by the inference chain (its 1e-4 gate needs bf16x3).
  // IN16 (with PLAIN, one source, no on-load affine): source 0 is a bfloat16 tensor -- the bf16 copy of dz the BatchNorm backward writes --
  // so a staging slot is 8 channels, goes into its (hi, k-group) plane as one 16-byte store without any split arithmetic, and the
  // loader moves half the bytes (the fp32 dz is then never written: mfpa_bn_relu_bwd(write_f32 = 0)).
  static_assert(!IN16 || PLAIN, "a bf16 source feeds the plain-bf16 products");
  constexpr int SPP = IN16 ? KC / 8 : KC / 4;                          // staging slots per pixel and 32-channel chunk
  constexpr int ESZ = IN16 ? 2 : 4;                                    // bytes per source element
  // (PLAIN halves the MFMA work per fragment read to a third: the tap-by-tap loop's 8 ds_read_b128 per 16 MFMAs saturate the CU's LDS
  //  pipe exactly -- the ROWS form, which reads every halo row once per column offset, is the one that suits it at every depth)
  // SIDE: the training step's side outputs (x0_bf16 / x1_bf16 / y_bf16 / stats_part) -- their own instantiations, so that the inference
  // kernels carry none of their code (it cost the 128-channel form 9 registers and 18 spills)
  // WMW = 2: 2 x 4 waves of 128 px x 32 ch (128-channel output tiles); WMW = 4: 4 x 2 waves of 64 px x 32 ch (the 64-channel layers)
  constexpr int THREADS = 512, WNW = 8 / WMW, BN = 32 * WNW, WPXW = 256 / WMW, TAPS = 9;
  static_assert((WMW == 2 || WMW == 4) && (!ROWS || WMW == 2), "wave grid");
  constexpr int HPW = PW + 2, HPH = PH + 2, HP = HPW * HPH, BM = PH * PW;
  static_assert(BM == 256 && (PW == 32 || PW == 16), "two waves of 128 pixels: eight 16-pixel tiles each");
  constexpr int A_F4 = (HP * SPP + THREADS - 1) / THREADS;
  constexpr int HPS = A_F4 * (THREADS / SPP);                     // staged pixels (>= HP): every staging slot has a row
  constexpr int PLANE = ((HPS * 16 + 255) / 256) * 256;                // bytes of one (hi|lo, k-group) plane, a multiple of 256
  constexpr int HLS = 4 * PLANE + 256;                                 // hi -> lo distance (planes 2, 3 sit 128 B further: room for that)
  constexpr int STAGE = 2 * HLS;
  constexpr int PT = WPXW / 16;                                        // 16-pixel tiles per wave
  static_assert(A_F4 <= TAPS - 3, "one halo staging slot per tap, taps 2 .. 7");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WMW, wn = __builtin_amdgcn_readfirstlane(wave / WMW);
  const int p = lane & 15, g = lane >> 4;
  auto plane_off = [](int hl, int kg) { return hl * HLS + kg * PLANE + (kg >> 1) * 128; };

  // PERSIST (WMW = 4): a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...; the halo of the next tile's first chunk is
  // requested and split under the last chunk of the current one, so a tile's prologue (a global round trip) and most of its epilogue
  // disappear behind the neighbours' MFMAs -- with 2 .. 4 chunks per tile they were a third of a workgroup's life.
  constexpr bool PERSIST = (WMW == 4) || (MFPA_WD16_PERSIST2 != 0 && !ROWS && !SIDE && !PLAIN) || (MFPA_WD16_PERSIST_ROWS != 0 && ROWS && !SIDE && !PLAIN) ||
                           (MFPA_WD16_PERSIST_PLAIN != 0 && PLAIN);
  const int n0 = blockIdx.y * BN;
  const int Cin = a.C0 + a.C1;
  const int nchunks = Cin / KC;
  const int ntiles = a.tiles_x * a.tiles_y * a.B;

  // ---- halo loader.  A thread's staging slots map to fixed halo pixels (pix = tid / 8 + 64 it).  Their byte offsets RELATIVE TO THE
  // TILE'S ORIGIN do not depend on the tile: one table [source][slot][thread] in LDS, built once per kernel (as registers the 12
  // offsets were spilled to scratch, and a scratch reload in front of a load drains every outstanding weight load); a tile adds its
  // scalar origin offset.  The loads are raw BUFFER loads through a per-clip descriptor (base = the clip, num_records = its bytes):
  // halo pixels above the first / below the last image row fall outside the clip and return zero without any clamping, pixels left /
  // right of the image read a neighbouring row's valid bytes; either way the slot is zeroed when it is split (`ain`: inside flags,
  // recomputed per tile from the slot's (row, column) and the tile's uniform bounds -- a few compares, no table).  Round 3 rebuilt a
  // clamped absolute table per tile (12 x (two divisions, four clamps, an LDS store) per thread): 3.7 k cycles per tile in front of the
  // first tap of every tile of the persistent form (profiles/r04_c64_timeline.txt).
  const int aq = tid % SPP;
  constexpr int TBL = 2 * A_F4 * THREADS;                              // the offset table: [2][A_F4][THREADS]
  unsigned* const aoffs0 = reinterpret_cast<unsigned*>(smem + 2 * STAGE);
#pragma unroll
  for (int it = 0; it < A_F4; ++it) {
    const int pix = tid / SPP + it * (THREADS / SPP);
    const int py = pix / HPW - 1, px = pix % HPW - 1;
    aoffs0[it * THREADS + tid] = (unsigned)((py * a.W + px) * a.C0 + (KC / SPP) * aq) * (unsigned)ESZ;    // may be "negative": wraps, see above
    aoffs0[(A_F4 + it) * THREADS + tid] = (unsigned)(((py - a.oy1) * a.W1 + (px - a.ox1)) * a.C1 + (KC / SPP) * aq) * (unsigned)ESZ;
  }
  typedef int i32x4_t __attribute__((ext_vector_type(4)));
  const unsigned clip0 = (unsigned)a.H * (unsigned)a.W * (unsigned)a.C0 * (unsigned)ESZ, clip1 = (unsigned)a.H1 * (unsigned)a.W1 * (unsigned)a.C1 * (unsigned)ESZ;
  struct Tile { int b, y0, x0p; unsigned ain; unsigned t0, t1; };      // t0 / t1: byte offset of the tile's origin pixel in source 0 / 1
  // decode tile t (workgroup-uniform scalars) and its inside flags (ain bit it: slot inside source 0's image; bit 8 + it: source 1)
  auto make_tile = [&](int t) __attribute__((always_inline)) {
    Tile T;
    int bx = __builtin_amdgcn_readfirstlane(t);
    const int tx = bx % a.tiles_x; bx /= a.tiles_x;
    const int ty = bx % a.tiles_y; bx /= a.tiles_y;
    T.b = bx; T.y0 = ty * PH; T.x0p = tx * PW; T.ain = 0;
    T.t0 = (unsigned)((T.y0 * a.W + T.x0p) * a.C0) * (unsigned)ESZ;
    T.t1 = (unsigned)((T.y0 * a.W1 + T.x0p) * a.C1) * (unsigned)ESZ;
#pragma unroll
    for (int it = 0; it < A_F4; ++it) {
      const int pix = tid / SPP + it * (THREADS / SPP);
      const int gy = T.y0 + pix / HPW - 1, gx = T.x0p + pix % HPW - 1;
      const bool in = pix < HP && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      const int y1 = gy - a.oy1, x1 = gx - a.ox1;
      const bool in1 = in && y1 >= 0 && y1 < a.H1 && x1 >= 0 && x1 < a.W1;
      T.ain |= (in ? 1u : 0u) << it | (in1 ? 1u : 0u) << (8 + it);
    }
    return T;
  };
  auto clip_rsrc = [&](const float* base, int b, unsigned clip_bytes) __attribute__((always_inline)) {
    const char* pb = reinterpret_cast<const char*>(base) + (size_t)b * clip_bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pb), 0, base != nullptr ? (int)clip_bytes : 0, 0x00020000);
  };
  int tile = blockIdx.x;
#ifdef MFPA_EXPERIMENTS
  // tap-level timeline (tools/exp_c64_timeline.py): wave 0 of workgroup (17, 0) stamps s_memtime at every tap start / epilogue start / end of
  // its first tiles into LDS (tag in the low 8 bits), dumped to the stamp buffer when the kernel ends
  unsigned long long* const tsbuf = reinterpret_cast<unsigned long long*>(smem + a.dbg_lds_stamps);
  int stamp_n = 0;
  const bool stamping = PERSIST && mfpa_conv_stamps != nullptr && a.dbg_lds_stamps != 0 && blockIdx.x == 17 && blockIdx.y == 0 && tid == 0;
  auto stamp = [&](int tag) __attribute__((always_inline)) {
    if (stamping && stamp_n < 500) tsbuf[stamp_n++] = (__builtin_amdgcn_s_memtime() & ~0xffull) | (unsigned)tag;
  };
#else
  auto stamp = [](int) {};
#endif
#ifdef MFPA_EXPERIMENTS
  if (PERSIST && a.dbg_stagger > 0) {                                  // experiment: de-synchronise the CUs' tile periods
    const int units = (int)(blockIdx.x & 7u) * a.dbg_stagger;
    for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(64);      // 64 x 64 cycles per unit
  }
#endif
  Tile S = make_tile(tile);                                            // the tile whose halo is being requested / split
  int eb = S.b, ey0 = S.y0, ex0p = S.x0p;                              // the tile being computed (epilogue coordinates)
  // training forward: the producer's per-channel (scale, shift) of source 0, copied to LDS once ([scale C0 | shift C0])
  float* aff = reinterpret_cast<float*>(smem + 2 * STAGE + TBL * sizeof(unsigned));
  if (a.in_scale0 != nullptr) {
    for (int i = tid; i < a.C0; i += THREADS) {
      aff[i] = a.in_scale0[i];
      aff[a.C0 + i] = a.in_shift0[i];
    }
    __syncthreads();
  }
  // the epilogue's per-channel constants (output affine of this workgroup's BN channels, the fused OutConv's weights) into LDS, once per
  // kernel: as global loads inside the epilogue they were followed by s_waitcnt vmcnt(0) -- which also waits for every halo and weight
  // load already in flight for the NEXT tile (persistent form) and for the epilogue's own stores of the previous one
  float* const epi = aff + (a.in_scale0 ? 2 * a.C0 : 0) + (a.w1x1 ? 2 * 256 : 0);      // [scale BN | shift BN | w1x1 64]
  for (int i = tid; i < BN; i += THREADS) {
    epi[i] = a.scale ? a.scale[n0 + i] : 1.f;
    epi[BN + i] = a.shift ? a.shift[n0 + i] : 0.f;
  }
  if (a.w1x1 != nullptr && tid < 64) epi[2 * BN + tid] = a.w1x1[tid];
  // (first read in the first epilogue, behind at least one of the main loop's barriers)
  // ROWS: the staging slots are requested in two halves (slots 0..2 in period 0, 3..5 in period 1) that share three registers
  constexpr int AHALF = (A_F4 + 1) / 2, AREGS = ROWS ? AHALF : A_F4;
  f32x4 areg[AREGS];
  auto load_a_range = [&](int chunk, auto FIRST, auto COUNT) __attribute__((always_inline)) {
    constexpr int first = decltype(FIRST)::value, count = decltype(COUNT)::value;
    const int c0 = chunk * KC;
    const bool from0 = c0 < a.C0;                                      // workgroup-uniform: scalar selects, no branch
    const auto rs = clip_rsrc(from0 ? a.x0 : a.x1, S.b, from0 ? clip0 : clip1);
    const unsigned toff = from0 ? S.t0 + (unsigned)c0 * (unsigned)ESZ : S.t1 + (unsigned)(c0 - a.C0) * (unsigned)ESZ;
    const unsigned* ao = aoffs0 + (from0 ? 0 : A_F4 * THREADS) + tid;
#pragma unroll
    for (int it = first; it < first + count && it < A_F4; ++it)
      areg[it % AREGS] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(ao[it * THREADS] + toff), 0, 0));
  };
  auto load_a = [&](int chunk) __attribute__((always_inline)) {
    if (MFPA_EXP_FLAG(a.dbg, 64)) return;
    load_a_range(chunk, std::integral_constant<int, 0>{}, std::integral_constant<int, AREGS>{});
  };
  // tap-by-tap forms: the slot offsets of the chunk requested at the next tap 0 are read from the LDS table one tap EARLIER (tap 8, when
  // the staging registers are dead) into component 0 of the staging registers themselves, so tap 0 issues its six loads without first
  // waiting for six LDS reads (the wait sat in front of tap 0's MFMAs: tap 0 took twice a steady-state tap, profiles/r04_c64_timeline.txt)
  auto preload_offsets = [&](int chunk) __attribute__((always_inline)) {
    const unsigned* ao = aoffs0 + (chunk * KC < a.C0 ? 0 : A_F4 * THREADS) + tid;
#pragma unroll
    for (int it = 0; it < A_F4; ++it) areg[it % AREGS][0] = __uint_as_float(ao[it * THREADS]);
  };
  // one staging slot of load_a_pre (MFPA_HALO_SPREAD: slot k is requested at tap k instead of all six at tap 0)
  auto load_a_one = [&](auto IT, int chunk) __attribute__((always_inline)) {
    constexpr int it = decltype(IT)::value;
    if (MFPA_EXP_FLAG(a.dbg, 64)) return;
    const int c0 = chunk * KC;
    const bool from0 = c0 < a.C0;
    const auto rs = clip_rsrc(from0 ? a.x0 : a.x1, S.b, from0 ? clip0 : clip1);
    const unsigned toff = from0 ? S.t0 + (unsigned)c0 * (unsigned)ESZ : S.t1 + (unsigned)(c0 - a.C0) * (unsigned)ESZ;
    areg[it % AREGS] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(__float_as_uint(areg[it % AREGS][0]) + toff), 0, 0));
  };
  auto load_a_pre = [&](int chunk) __attribute__((always_inline)) {
    if (MFPA_EXP_FLAG(a.dbg, 64)) return;
    const int c0 = chunk * KC;
    const bool from0 = c0 < a.C0;
    const auto rs = clip_rsrc(from0 ? a.x0 : a.x1, S.b, from0 ? clip0 : clip1);
    const unsigned toff = from0 ? S.t0 + (unsigned)c0 * (unsigned)ESZ : S.t1 + (unsigned)(c0 - a.C0) * (unsigned)ESZ;
#pragma unroll
    for (int it = 0; it < A_F4; ++it)
      areg[it % AREGS] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(__float_as_uint(areg[it % AREGS][0]) + toff), 0, 0));
  };
  // one staging slot: zero padding, the training forward's on-load affine + ReLU + dropout, bf16 hi / lo split, two 8-byte stores
  // into the (hi, k-group) and (lo, k-group) planes (a thread's channel quad is half of k-group aq >> 1)
  auto split_slot = [&](auto IT, int chunk, char* stage) __attribute__((always_inline)) {
    constexpr int it = decltype(IT)::value;
    const int pix = tid / SPP + it * (THREADS / SPP);
    const int c0 = chunk * KC;
    const bool inside = (S.ain >> ((c0 < a.C0 ? 0 : 8) + it)) & 1u;
    f32x4 v = areg[it % AREGS];
    if (!inside) v = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (IN16) {                                              // eight bf16 channels = the lane's whole (hi, k-group aq) piece
      if (AFF16 && a.in_scale0 != nullptr && c0 < a.C0 && inside) {
        // bf16 z (the training step's activations kept as bfloat16 in HBM): widen, the producer's BatchNorm affine + ReLU (+ dropout) in
        // float32 exactly as the float32 path applies them, round to bf16 once -- the MFMA operand
        float f[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned u = __float_as_uint(v[k]);
          f[2 * k] = __uint_as_float(u << 16);
          f[2 * k + 1] = __uint_as_float(u & 0xffff0000u);
        }
        const float* scp = aff + c0 + 8 * aq;
        const float* shp = aff + a.C0 + c0 + 8 * aq;
        const f32x4 sc0 = *reinterpret_cast<const f32x4*>(scp), sc1 = *reinterpret_cast<const f32x4*>(scp + 4);
        const f32x4 sh0 = *reinterpret_cast<const f32x4*>(shp), sh1 = *reinterpret_cast<const f32x4*>(shp + 4);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float y = f[k] * (k < 4 ? sc0[k & 3] : sc1[k & 3]) + (k < 4 ? sh0[k & 3] : sh1[k & 3]);
          f[k] = y > 0.f ? y : 0.f;
        }
        if (a.drop_thresh) {
          const int gy = S.y0 + pix / HPW - 1, gx = S.x0p + pix % HPW - 1;   // inside the image here
          const unsigned long long e0 = (((unsigned long long)S.b * a.H + gy) * a.W + gx) * a.C0 + c0 + 8 * aq;
#pragma unroll
          for (int k = 0; k < 8; ++k) f[k] = mfpa_keep(a.drop_seed, a.drop_thresh, e0 + k) ? f[k] * a.drop_scale : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          bf16x2 h2;
          h2[0] = (__bf16)f[2 * k];
          h2[1] = (__bf16)f[2 * k + 1];
          v[k] = __builtin_bit_cast(float, h2);
        }
      }
      *reinterpret_cast<f32x4*>(stage + plane_off(0, aq) + pix * 16) = v;
      if constexpr (SIDE && AFF16) {                                   // the activated source 0 as the weight gradient reads it (see below)
        const int py = pix / HPW, px = pix % HPW;
        if (a.x0_bf16 != nullptr && c0 < a.C0 && inside && py >= 1 && py <= PH && px >= 1 && px <= PW && blockIdx.y == 0) {
          const size_t e = (size_t)S.b * a.H * a.W * a.C0 + ((aoffs0[it * THREADS + tid] + S.t0) >> 1) + c0;
          *reinterpret_cast<f32x4*>(a.x0_bf16 + e) = v;
        }
      }
      return;
    }
    if (a.in_scale0 != nullptr && c0 < a.C0 && inside) {
      // from the LDS copy: a global load here is followed by s_waitcnt vmcnt(0), which also waits for every weight load in flight
      const f32x4 sc = *reinterpret_cast<const f32x4*>(aff + c0 + 4 * aq);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(aff + a.C0 + c0 + 4 * aq);
      v = v * sc + sh;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
      if (a.drop_thresh) {
        const int gy = S.y0 + pix / HPW - 1, gx = S.x0p + pix % HPW - 1;   // inside the image here
        const unsigned long long e0 = (((unsigned long long)S.b * a.H + gy) * a.W + gx) * a.C0 + c0 + 4 * aq;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = mfpa_keep(a.drop_seed, a.drop_thresh, e0 + k) ? v[k] * a.drop_scale : 0.f;
      }
    }
    bf16x4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hi[k] = (__bf16)v[k];
      lo[k] = (__bf16)(v[k] - (float)hi[k]);
    }
    char* at = stage + plane_off(0, aq >> 1) + pix * 16 + 8 * (aq & 1);
    *reinterpret_cast<bf16x4*>(at) = hi;
    if constexpr (!PLAIN) *reinterpret_cast<bf16x4*>(at + HLS) = lo;
    // training forward: the bf16 copy of the activated source 0 the weight gradient reads -- the hi half is exactly that.  Every pixel
    // is interior (not halo) to one tile; the first output-channel tile writes it.
    if constexpr (SIDE) {
      // (the pixel's element offset inside its clip is the loader's byte offset / 4, read back from the LDS table: as registers the six
      // per-slot offsets would be loop invariants the compiler keeps -- and spills)
      const int py = pix / HPW, px = pix % HPW;
      const bool interior = inside && py >= 1 && py <= PH && px >= 1 && px <= PW && blockIdx.y == 0;
      if (a.x0_bf16 != nullptr && c0 < a.C0 && interior) {
        const size_t e = (size_t)S.b * a.H * a.W * a.C0 + ((aoffs0[it * THREADS + tid] + S.t0) >> 2) + c0;
        *reinterpret_cast<bf16x4*>(a.x0_bf16 + e) = hi;
      }
      if (a.x1_bf16 != nullptr && c0 >= a.C0 && interior) {            // source 1: its own (smaller, offset) geometry
        const size_t e = (size_t)S.b * a.H1 * a.W1 * a.C1 + ((aoffs0[(A_F4 + it) * THREADS + tid] + S.t1) >> 2) + (c0 - a.C0);
        *reinterpret_cast<bf16x4*>(a.x1_bf16 + e) = hi;
      }
    }
  };

  // ---- weight fragments: ring of three sets, [slot][16-channel tile][hi, lo]
  bf16x8 wq[3][2][2];
  auto load_w = [&](int chunk, int tap, auto SLOT) __attribute__((always_inline)) {
    constexpr int slot = decltype(SLOT)::value;
    const char* wb = reinterpret_cast<const char*>(a.w) +
                     ((((size_t)tap * nchunks + chunk) * (size_t)(a.Cout / 16) + (size_t)(n0 / 16 + 2 * wn)) << 11) + lane * 16;
    wq[slot][0][0] = *reinterpret_cast<const bf16x8*>(wb);
    if constexpr (!PLAIN) wq[slot][0][1] = *reinterpret_cast<const bf16x8*>(wb + 1024);
    wq[slot][1][0] = *reinterpret_cast<const bf16x8*>(wb + 2048);
    if constexpr (!PLAIN) wq[slot][1][1] = *reinterpret_cast<const bf16x8*>(wb + 3072);
  };

  // ---- pixel fragments: two sets of four 16-pixel tiles (hi, lo)
  struct XFrags { bf16x8 h[4], l[4]; };
  XFrags fx0, fx1;
  // byte offset of the lane's row of pixel tile 0 in plane (hi, g) at tap (0, 0); the other tiles are compile-time displacements of it
  // (a 16-pixel tile is half a patch row of the 32-wide patches, a whole row of the 16-wide ones)
  const int xbase = (((wm * WPXW + p) / PW) * HPW + ((wm * WPXW + p) % PW)) * 16 + plane_off(0, g);
  auto tile_disp = [](int pt) { return (PW == 32) ? ((pt >> 1) * HPW + (pt & 1) * 16) * 16 : pt * HPW * 16; };
  auto read_x = [&](XFrags& f, const char* stage, int tap_off, int half) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* r = stage + xbase + tile_disp(4 * half + i) + tap_off;
      if constexpr (!PLAIN) f.l[i] = *reinterpret_cast<const bf16x8*>(r + HLS);
      f.h[i] = *reinterpret_cast<const bf16x8*>(r);
    }
  };
  floatx4 acc[2][PT];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = floatx4{0.f, 0.f, 0.f, 0.f};
  auto mfma_half = [&](const XFrags& f, const bf16x8 (&w)[2][2], int half) __attribute__((always_inline)) {
    // term-major: an accumulator is touched every eighth instruction
    if constexpr (!PLAIN) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[ct][4 * half + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ct][1], f.h[i], acc[ct][4 * half + i], 0, 0, 0);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[ct][4 * half + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ct][0], f.l[i], acc[ct][4 * half + i], 0, 0, 0);
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[ct][4 * half + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ct][0], f.h[i], acc[ct][4 * half + i], 0, 0, 0);
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  constexpr int N_R = PLAIN ? 4 : 8, N_M = PLAIN ? 8 : 24;             // fragment reads / MFMAs of one phase
  constexpr int N_W = PLAIN ? 2 : 4;                                   // weight-fragment loads of one tap
  // One tap.  Phase A: MFMA(pixel tiles 0..3 of tap t) || read tiles 4..7 of tap t, request the weights of tap t + 2.  Phase B: MFMA(tiles
  // 4..7) || read tiles 0..3 of tap t + 1, (tap 0) request the next chunk's halo, (taps 2..7) split one staging slot of it.  The chunk's
  // one barrier sits between the phases of tap 8 (see BDIR).
  auto tap_body = [&](auto TAP, int chunk) __attribute__((always_inline)) {
    constexpr int tap = decltype(TAP)::value;
    constexpr int ntap = (tap + 1) % TAPS;
    constexpr int tap_off = ((tap / 3) * HPW + (tap % 3)) * 16, ntap_off = ((ntap / 3) * HPW + (ntap % 3)) * 16;
    const int chunk_n = chunk + 1 < nchunks ? chunk + 1 : (PERSIST ? 0 : chunk);
    const char* cur = smem + (chunk & 1) * STAGE;
    const char* nxt = (tap == TAPS - 1) ? smem + ((chunk + 1) & 1) * STAGE : cur;
    read_x(fx1, cur, tap_off, 1);
    mfma_half(fx0, wq[tap % 3], 0);
    load_w((tap + 2 >= TAPS) ? chunk_n : chunk, (tap + 2) % TAPS, std::integral_constant<int, (tap + 2) % 3>{});
    pin_reads<N_M - 1, N_R>();
    constexpr int used_a = pin_read_slots(N_M - 1, N_R);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, N_W, 0);
    if constexpr (N_M - used_a - 1 > 0) __builtin_amdgcn_sched_group_barrier(0x008, N_M - used_a - 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (tap == TAPS - 1) {
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (tap == 0) load_a_pre(chunk_n);
    if (tap == TAPS - 1) preload_offsets(PERSIST ? (chunk + 2 < nchunks ? chunk + 2 : chunk + 1 < nchunks ? 0 : 1 < nchunks ? 1 : 0)
                                                 : (chunk + 2 < nchunks ? chunk + 2 : chunk + 1 < nchunks ? chunk + 1 : chunk));   // what the next tap 0 requests
    read_x(fx0, nxt, ntap_off, 0);
    mfma_half(fx1, wq[tap % 3], 1);
    if constexpr (tap >= 2 && tap - 2 < A_F4) {
      split_slot(std::integral_constant<int, tap - 2>{}, chunk_n, smem + ((chunk + 1) & 1) * STAGE);
#pragma unroll
      for (int i = 0; i < N_R; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      if constexpr (N_M - N_R - 4 > 0) __builtin_amdgcn_sched_group_barrier(0x008, N_M - N_R - 4, 0);
    } else {
      pin_reads<N_M, N_R>();
      if constexpr (N_M - pin_read_slots(N_M, N_R) > 0) __builtin_amdgcn_sched_group_barrier(0x008, N_M - pin_read_slots(N_M, N_R), 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // WMW = 4 (a wave owns four pixel tiles): ONE phase per tap -- MFMA(tiles 0..3 of tap t, fragment set t & 1) || read tiles 0..3 of tap
  // t + 1 into the other set, request the weights of tap t + 2, (tap 0) request the next chunk's halo, (taps 2..7) split one staging
  // slot.  Nine taps per chunk: the set parity flips with the chunk, so the loop body is two chunks (PAR = parity of tap 0's set).
  auto tap_body4 = [&](auto TAP, auto PAR, int chunk) __attribute__((always_inline)) {
    constexpr int tap = decltype(TAP)::value, par = (decltype(PAR)::value + tap) & 1;
    constexpr int ntap = (tap + 1) % TAPS;
    constexpr int ntap_off = ((ntap / 3) * HPW + (ntap % 3)) * 16;
    const int chunk_n = chunk + 1 < nchunks ? chunk + 1 : 0;          // past the tile's last chunk: chunk 0 of the next tile (S is that tile by then)
    const char* cur = smem + (chunk & 1) * STAGE;
    const char* nxt = (tap == TAPS - 1) ? smem + ((chunk + 1) & 1) * STAGE : cur;
    stamp(tap);
    if (tap == TAPS - 1) {
      __syncthreads();                                                 // the next stage is complete, this one is read out
      __builtin_amdgcn_sched_barrier(0);
      stamp(9);
    }
#if MFPA_HALO_SPREAD
    if constexpr (tap < A_F4) load_a_one(std::integral_constant<int, tap < A_F4 ? tap : 0>{}, chunk_n);
#else
    if (tap == 0) load_a_pre(chunk_n);
#endif
    // what the next tap 0 requests: chunk + 2; from the tile's last-but-one chunk on, the next tile's chunk 0, then its chunk 1
    if (tap == TAPS - 1) preload_offsets(chunk + 2 < nchunks ? chunk + 2 : chunk + 1 < nchunks ? 0 : 1 < nchunks ? 1 : 0);
    read_x(par ? fx0 : fx1, nxt, ntap_off, 0);
    if (!MFPA_EXP_FLAG(a.dbg, 8)) mfma_half(par ? fx1 : fx0, wq[tap % 3], 0);
    if (!MFPA_EXP_FLAG(a.dbg, 128)) load_w((tap + 2 >= TAPS) ? chunk_n : chunk, (tap + 2) % TAPS, std::integral_constant<int, (tap + 2) % 3>{});
    if constexpr (tap >= 2 && tap - 2 < A_F4) {
      if (!MFPA_EXP_FLAG(a.dbg, 256)) split_slot(std::integral_constant<int, tap - 2>{}, chunk_n, smem + ((chunk + 1) & 1) * STAGE);
#pragma unroll
      for (int i = 0; i < N_R; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
#pragma unroll
      for (int i = 0; i < N_W; ++i) {
        if constexpr (!PLAIN) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      if constexpr (N_M - N_R - 8 > 0) __builtin_amdgcn_sched_group_barrier(0x008, N_M - N_R - 8, 0);
    } else {
#pragma unroll
      for (int i = 0; i < N_R; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < N_W; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      if constexpr (N_M - N_R - N_W > 0) __builtin_amdgcn_sched_group_barrier(0x008, N_M - N_R - N_W, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- ROWS form of the main loop: the three vertical taps of a column offset dx share their pixel fragments.  A wave's 128 pixels are
  // R patch rows (4 of 32 pixels, or 8 of 16); tap (dy, dx) of row r reads halo row r + dy, so for one dx the R + 2 halo rows are read
  // ONCE each (hi and lo) and row h feeds the accumulators of rows h, h - 1, h - 2 with the weights of taps (0, dx), (1, dx), (2, dx):
  // 72 (60) fragment reads per 32-channel chunk instead of 144 -- the half-reads timing experiment on the tap-by-tap loop returned
  // 4-7 % on the >= 128-channel layers (the loop is power limited, LDS traffic is part of the power).  A "period" = one dx: 144 MFMAs,
  // the weights of its three taps in registers, the next period's three taps requested at its start into the other half of a
  // six-set ring (static indices: the loop body is two chunks = six periods), the next chunk's halo requested in period 0 and split
  // in periods 1 and 2, the chunk's one barrier in front of the last row step of period 2 (whose prefetch reads the next stage).
  constexpr int R = BM / 2 / PW, HV = PW / 16;
  bf16x8 wr[2][3][2][2];                                               // [ring half][dy][16-channel tile][hi, lo]
  auto load_wr = [&](int chunk, auto DX, auto PAR) __attribute__((always_inline)) {
    constexpr int dx = decltype(DX)::value, par = decltype(PAR)::value;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const char* wb = reinterpret_cast<const char*>(a.w) +
                       ((((size_t)(dy * 3 + dx) * nchunks + chunk) * (size_t)(a.Cout / 16) + (size_t)(n0 / 16 + 2 * wn)) << 11) + lane * 16;
      wr[par][dy][0][0] = *reinterpret_cast<const bf16x8*>(wb);
      if constexpr (!PLAIN) wr[par][dy][0][1] = *reinterpret_cast<const bf16x8*>(wb + 1024);
      wr[par][dy][1][0] = *reinterpret_cast<const bf16x8*>(wb + 2048);
      if constexpr (!PLAIN) wr[par][dy][1][1] = *reinterpret_cast<const bf16x8*>(wb + 3072);
    }
  };
  // pixel fragments: one "unit" = 16 pixels of one halo row (hi and lo); a ring of NB units, read NB - 1 units ahead of their MFMAs (a
  // unit carries only 6 .. 18 MFMAs: one unit of distance is shorter than the LDS latency).  NU units per period, NB divides NU.
  constexpr int NU = (R + 2) * HV, NB = (PW == 32) ? 4 : 5;
  static_assert(NU % NB == 0, "static ring indices");
  auto load_wr1 = [&](int chunk, auto DX, auto PAR, auto DY) __attribute__((always_inline)) {
    constexpr int dx = decltype(DX)::value, par = decltype(PAR)::value, dy = decltype(DY)::value;
    const char* wb = reinterpret_cast<const char*>(a.w) +
                     ((((size_t)(dy * 3 + dx) * nchunks + chunk) * (size_t)(a.Cout / 16) + (size_t)(n0 / 16 + 2 * wn)) << 11) + lane * 16;
    wr[par][dy][0][0] = *reinterpret_cast<const bf16x8*>(wb);
    if constexpr (!PLAIN) wr[par][dy][0][1] = *reinterpret_cast<const bf16x8*>(wb + 1024);
    wr[par][dy][1][0] = *reinterpret_cast<const bf16x8*>(wb + 2048);
    if constexpr (!PLAIN) wr[par][dy][1][1] = *reinterpret_cast<const bf16x8*>(wb + 3072);
  };
  struct XUnit { bf16x8 h, l; };
  XUnit xu[NB];
  auto read_unit = [&](XUnit& f, const char* stage, int u, int dx) __attribute__((always_inline)) {
    const char* r = stage + xbase + ((u / HV) * HPW + (u % HV) * 16 + dx) * 16;
    if constexpr (!PLAIN) f.l = *reinterpret_cast<const bf16x8*>(r + HLS);
    f.h = *reinterpret_cast<const bf16x8*>(r);
  };
  // unit u = (halo row h, half hv): every (dy, r = h - dy) pair it serves, term-major (an accumulator is touched once per term)
  auto mfma_unit = [&](auto U, const XUnit& f, auto PAR) __attribute__((always_inline)) {
    constexpr int h = decltype(U)::value / HV, hv = decltype(U)::value % HV, par = decltype(PAR)::value;
#pragma unroll
    for (int term = PLAIN ? 2 : 0; term < 3; ++term)                    // PLAIN: the hi x hi term only
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        if (h - dy < 0 || h - dy >= R) continue;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          floatx4& c = acc[ct][(h - dy) * HV + hv];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[par][dy][ct][term == 0 ? 1 : 0], term == 1 ? f.l : f.h, c, 0, 0, 0);
        }
      }
  };
  auto unit_mfmas = [](int u) { const int h = u / HV; int n = 0; for (int dy = 0; dy < 3; ++dy) n += (h - dy >= 0 && h - dy < R) ? 1 : 0; return n * (PLAIN ? 2 : 6); };
  constexpr int BARU = NU - NB + 1;                                    // first unit whose prefetch reads the next period
  constexpr int SA = NU / 2, SB = BARU - AHALF;                        // first split units of periods 1 and 2
  static_assert(SA + AHALF < NU && SB >= 0, "staging schedule");
  auto unit_step = [&](auto U, auto DX, auto PAR, int chunk) __attribute__((always_inline)) {
    constexpr int u = decltype(U)::value, dx = decltype(DX)::value, par = decltype(PAR)::value;
    constexpr int n_m = unit_mfmas(u);
    const int chunk_n = chunk + 1 < nchunks ? chunk + 1 : (PERSIST ? 0 : chunk);
    const char* cur = smem + (chunk & 1) * STAGE;
    char* nxs = smem + ((chunk + 1) & 1) * STAGE;
    if constexpr (u == BARU && dx == 2) {
      __syncthreads();                                                 // the next stage is complete, this one is read out
      __builtin_amdgcn_sched_barrier(0);
    }
    constexpr int pu = u + NB - 1;                                     // the unit requested now
    if constexpr (pu < NU) read_unit(xu[pu % NB], cur, pu, dx);
    else read_unit(xu[pu % NB], dx == 2 ? nxs : cur, pu - NU, (dx + 1) % 3);
    mfma_unit(U, xu[u % NB], PAR);
    // the period's other work: the next period's weights (one tap per unit, units 1 .. 3), the next chunk's halo (first half requested
    // in period 0, split in period 1; second half requested in period 1 behind that, split in period 2 in front of the barrier)
    constexpr int split_it = dx == 1 ? u - SA : dx == 2 ? AHALF + u - SB : -1;
    constexpr bool do_split = (dx == 1 && u >= SA && u < SA + AHALF) || (dx == 2 && u >= SB && u < SB + AHALF && split_it < A_F4);
    if constexpr (u == 0 && dx == 0) load_a_range(chunk_n, std::integral_constant<int, 0>{}, std::integral_constant<int, AHALF>{});
    if constexpr (u == SA + AHALF && dx == 1) load_a_range(chunk_n, std::integral_constant<int, AHALF>{}, std::integral_constant<int, AHALF>{});
    if constexpr (u >= 1 && u <= 3) {
      if constexpr (dx < 2) load_wr1(chunk, std::integral_constant<int, (dx + 1) % 3>{}, std::integral_constant<int, par ^ 1>{}, std::integral_constant<int, (u >= 1 && u <= 3) ? u - 1 : 0>{});
      else load_wr1(chunk_n, std::integral_constant<int, 0>{}, std::integral_constant<int, par ^ 1>{}, std::integral_constant<int, (u >= 1 && u <= 3) ? u - 1 : 0>{});
    }
    if constexpr (do_split) split_slot(std::integral_constant<int, do_split ? split_it : 0>{}, chunk_n, nxs);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, PLAIN ? 1 : 2, 0);
    if constexpr (do_split && PLAIN) {
      // a unit carries 2 .. 6 MFMAs here: the split's vector work goes between them in equal parts
#pragma unroll
      for (int i = 1; i < n_m; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x002, (24 + n_m - 2) / (n_m - 1), 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    } else if constexpr (PLAIN && u >= 1 && u <= 3) {
      __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
      if constexpr (n_m - 1 > 0) __builtin_amdgcn_sched_group_barrier(0x008, n_m - 1, 0);
    } else if constexpr (do_split) {
      __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
      if constexpr (n_m - 5 > 0) __builtin_amdgcn_sched_group_barrier(0x008, n_m - 5, 0);
    } else if constexpr (u >= 1 && u <= 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      if constexpr (n_m - 5 > 0) __builtin_amdgcn_sched_group_barrier(0x008, n_m - 5, 0);
    } else {
      if constexpr (n_m - 1 > 0) __builtin_amdgcn_sched_group_barrier(0x008, n_m - 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto period = [&](auto DX, auto PAR, int chunk) __attribute__((always_inline)) {
    unit_step(std::integral_constant<int, 0>{}, DX, PAR, chunk);
    unit_step(std::integral_constant<int, 1>{}, DX, PAR, chunk);
    unit_step(std::integral_constant<int, 2>{}, DX, PAR, chunk);
    unit_step(std::integral_constant<int, 3>{}, DX, PAR, chunk);
    unit_step(std::integral_constant<int, 4>{}, DX, PAR, chunk);
    unit_step(std::integral_constant<int, 5>{}, DX, PAR, chunk);
    unit_step(std::integral_constant<int, 6>{}, DX, PAR, chunk);
    unit_step(std::integral_constant<int, 7>{}, DX, PAR, chunk);
    unit_step(std::integral_constant<int, 8>{}, DX, PAR, chunk);
    unit_step(std::integral_constant<int, 9>{}, DX, PAR, chunk);
    if constexpr (NU == 12) {
      unit_step(std::integral_constant<int, NU == 12 ? 10 : 0>{}, DX, PAR, chunk);
      unit_step(std::integral_constant<int, NU == 12 ? 11 : 0>{}, DX, PAR, chunk);
    }
  };

  if constexpr (ROWS) {
    // chunk 0: all the staging slots in ONE round trip (the second half through temporaries: the fragment ring is not live yet)
    f32x4 keep[AREGS];
    load_a_range(0, std::integral_constant<int, AHALF>{}, std::integral_constant<int, AHALF>{});
#pragma unroll
    for (int i = 0; i < AREGS; ++i) keep[i] = areg[i];
    load_a_range(0, std::integral_constant<int, 0>{}, std::integral_constant<int, AHALF>{});
    load_wr(0, S0{}, S0{});
    // slots [0, AHALF) sit in the staging registers, slots [AHALF, A_F4) in `keep` (AHALF = 3 for float32 sources, 2 for bf16 ones)
    split_slot(std::integral_constant<int, 0>{}, 0, smem);
    if constexpr (AHALF > 1) split_slot(std::integral_constant<int, 1>{}, 0, smem);
    if constexpr (AHALF > 2) split_slot(std::integral_constant<int, 2>{}, 0, smem);
#pragma unroll
    for (int i = 0; i < AREGS; ++i) areg[i] = keep[i];
    if constexpr (A_F4 > AHALF) split_slot(std::integral_constant<int, AHALF>{}, 0, smem);
    if constexpr (A_F4 > AHALF + 1) split_slot(std::integral_constant<int, AHALF + 1 < A_F4 ? AHALF + 1 : 0>{}, 0, smem);
    if constexpr (A_F4 > AHALF + 2) split_slot(std::integral_constant<int, AHALF + 2 < A_F4 ? AHALF + 2 : 0>{}, 0, smem);
    static_assert(A_F4 <= 2 * AHALF && AHALF <= 3, "two halves of at most three staging slots");
  } else {
    load_a(0);
    load_w(0, 0, S0{});
    load_w(0, 1, S1{});
    using I0 = std::integral_constant<int, 0>;
    split_slot(I0{}, 0, smem);
    if constexpr (A_F4 > 1) split_slot(std::integral_constant<int, 1>{}, 0, smem);
    if constexpr (A_F4 > 2) split_slot(std::integral_constant<int, 2>{}, 0, smem);
    if constexpr (A_F4 > 3) split_slot(std::integral_constant<int, 3>{}, 0, smem);
    if constexpr (A_F4 > 4) split_slot(std::integral_constant<int, 4>{}, 0, smem);
    if constexpr (A_F4 > 5) split_slot(std::integral_constant<int, 5>{}, 0, smem);
    preload_offsets(PERSIST ? (1 < nchunks ? 1 : 0) : (1 < nchunks ? 1 : 0));     // chunk 0's tap 0 requests chunk 1
  }
  auto epilogue = [&]() __attribute__((always_inline)) {
  // ---- epilogue: D[channel 4 g + j of tile ct][pixel p of tile pt]: out = relu(acc * scale + shift), 16-byte stores
  #pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int chl = wn * 32 + ct * 16 + 4 * g;                       // channel inside the workgroup's BN
#if MFPA_EPI_LDS
      const f32x4 sc = *reinterpret_cast<const f32x4*>(epi + chl);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(epi + BN + chl);
#else
      const f32x4 sc = a.scale ? *reinterpret_cast<const f32x4*>(a.scale + n0 + chl) : f32x4{1.f, 1.f, 1.f, 1.f};
      const f32x4 sh = a.shift ? *reinterpret_cast<const f32x4*>(a.shift + n0 + chl) : f32x4{0.f, 0.f, 0.f, 0.f};
#endif
  #pragma unroll
      for (int pt = 0; pt < PT; ++pt)
  #pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = acc[ct][pt][j] * sc[j] + sh[j];
          if (a.relu) v = v > 0.f ? v : 0.f;
          acc[ct][pt][j] = v;
        }
    }
    if (SIDE && a.stats_part != nullptr) {
      // training forward: the BatchNorm statistics of this output, one partial row per wave -- (sum, sum of squares) over the wave's
      // stored pixels for each of its 32 channels; rows are summed in float64 by mfpa_conv_stats_reduce (fixed order: deterministic)
      float vm[PT];
  #pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int m = wm * WPXW + pt * 16 + p;
        vm[pt] = (ey0 + m / PW < a.yH && ex0p + m % PW < a.yW) ? 1.f : 0.f;
      }
      float* row = a.stats_part + ((size_t)tile * WMW + wm) * 2 * a.Cout + n0 + wn * 32 + 4 * g;
  #pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        f32x4 sv = {0.f, 0.f, 0.f, 0.f}, qv = {0.f, 0.f, 0.f, 0.f};
        if (!AFF16 && a.bz != nullptr) {                                 // (never in the training forward's AFF16 form)
          // the output is dy of a BatchNorm + ReLU whose input bz has this tensor's shape: (sum g, sum g * xhat) instead (see ConvArgs)
          const int ch = n0 + wn * 32 + ct * 16 + 4 * g;
          const f32x4 bsc = *reinterpret_cast<const f32x4*>(a.bz_scale + ch), bsh = *reinterpret_cast<const f32x4*>(a.bz_shift + ch);
          const f32x4 bmu = *reinterpret_cast<const f32x4*>(a.bz_mean + ch), bis = *reinterpret_cast<const f32x4*>(a.bz_invstd + ch);
          // (the dtype branch sits OUTSIDE the pixel loop: inside it, every iteration was branch -> load -> wait, eight dependent round trips)
          f32x4 zzs[PT];
          auto zoff = [&](int pt) __attribute__((always_inline)) {
            const int m = wm * WPXW + pt * 16 + p;
            const int gy = min(ey0 + m / PW, a.yH - 1), gx = min(ex0p + m % PW, a.yW - 1);      // clamped; masked by vm
            return (((size_t)eb * a.yH + gy) * a.yW + gx) * (size_t)a.Cout + ch;
          };
          if (a.bz16) {                                                  // bz kept as bfloat16 (mfpa_conv_desc.bwd_z_is_bf16)
            f32x2 raw[PT];
  #pragma unroll
            for (int pt = 0; pt < PT; ++pt) raw[pt] = *reinterpret_cast<const f32x2*>(reinterpret_cast<const __bf16*>(a.bz) + zoff(pt));
  #pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
              const unsigned u0 = __float_as_uint(raw[pt][0]), u1 = __float_as_uint(raw[pt][1]);
              zzs[pt] = f32x4{__uint_as_float(u0 << 16), __uint_as_float(u0 & 0xffff0000u), __uint_as_float(u1 << 16), __uint_as_float(u1 & 0xffff0000u)};
            }
          } else {
  #pragma unroll
            for (int pt = 0; pt < PT; ++pt) zzs[pt] = *reinterpret_cast<const f32x4*>(a.bz + zoff(pt));
          }
  #pragma unroll
          for (int pt = 0; pt < PT; ++pt) {
            const f32x4 zz = zzs[pt];
  #pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float gg = (zz[j] * bsc[j] + bsh[j] > 0.f) ? acc[ct][pt][j] * vm[pt] : 0.f;
              sv[j] += gg;
              qv[j] += gg * ((zz[j] - bmu[j]) * bis[j]);
            }
          }
        } else {
  #pragma unroll
        for (int pt = 0; pt < PT; ++pt)
  #pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float v = acc[ct][pt][j] * vm[pt];
            sv[j] += v;
            qv[j] += v * v;
          }
        }
  #pragma unroll
        for (int j = 0; j < 4; ++j) {
          // row_shr 1, 2, 4, 8: lane 15 of a 16-lane row ends with its total
          sv[j] = dpp_row_add<0x118>(dpp_row_add<0x114>(dpp_row_add<0x112>(dpp_row_add<0x111>(sv[j]))));
          qv[j] = dpp_row_add<0x118>(dpp_row_add<0x114>(dpp_row_add<0x112>(dpp_row_add<0x111>(qv[j]))));
        }
        if (p == 15) {
          *reinterpret_cast<f32x4*>(row + ct * 16) = sv;
          *reinterpret_cast<f32x4*>(row + a.Cout + ct * 16) = qv;
        }
      }
    }
    if (a.y != nullptr || (SIDE && IN16 && a.y_bf16 != nullptr)) {      // (y null with y_bf16: the output exists as bfloat16 only)
      char* yb = reinterpret_cast<char*>(a.y + (size_t)eb * a.yH * a.yW * a.Cout);
  #pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int m = wm * WPXW + pt * 16 + p;
        const int gy = ey0 + m / PW, gx = ex0p + m % PW;
        if (gy < a.yH && gx < a.yW) {
          char* yp = yb + (((unsigned)gy * (unsigned)a.yW + (unsigned)gx) * (unsigned)a.Cout + (unsigned)(n0 + wn * 32 + 4 * g)) * 4u;
          if (a.y != nullptr) {
  #pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            f32x4 o;
  #pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = acc[ct][pt][j];
            if (MFPA_EXP_FLAG(a.dbg, 512)) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(yp + ct * 64));
            else if (!(MFPA_EXP_FLAG(a.dbg, 4) && v_never(o[0] + o[3]))) *reinterpret_cast<f32x4*>(yp + ct * 64) = o;
          }
          }
          if (SIDE && a.y_bf16 != nullptr) {
            __bf16* hp = a.y_bf16 + (((size_t)eb * a.yH + gy) * a.yW + gx) * (size_t)a.Cout + n0 + wn * 32 + 4 * g;
  #pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
              bf16x4 h;
  #pragma unroll
              for (int j = 0; j < 4; ++j) h[j] = (__bf16)acc[ct][pt][j];
              *reinterpret_cast<bf16x4*>(hp + ct * 16) = h;
            }
          }
        }
      }
    }
    if (a.y_pool != nullptr) {
      // MaxPool2d(2) (floor): the window's two rows are two of the wave's pixel tiles, its two columns adjacent lanes (one DPP swap)
      const int Ho = a.H / 2, Wo = a.W / 2;
      constexpr int ROWSTEP = (PW == 32) ? 2 : 1;                        // pixel tiles per patch row
  #pragma unroll
      for (int ct = 0; ct < 2; ++ct)
  #pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          const int row = pt / ROWSTEP;                                  // patch row inside the wave's block
          if (row & 1) continue;
          f32x4 v;
  #pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float t = fmaxf(acc[ct][pt][j], acc[ct][pt + ROWSTEP][j]);
            const float o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0xB1, 0xf, 0xf, true));      // quad_perm [1,0,3,2]
            v[j] = fmaxf(t, o);
          }
          const int m = wm * WPXW + pt * 16 + p;
          const int py = (ey0 + m / PW) / 2, px = (ex0p + m % PW) / 2;
          if (!(p & 1) && py < Ho && px < Wo)
            *reinterpret_cast<f32x4*>(a.y_pool + (((size_t)eb * Ho + py) * Wo + px) * a.Cout + n0 + wn * 32 + ct * 16 + 4 * g) = v;
        }
    }
  
    if constexpr (WMW == 4) {
      if (a.w1x1 != nullptr) {
        // fused OutConv 1x1 to one class (the whole C_out = 64 is in this workgroup): a lane's eight channels times their weights,
        // the four channel groups of a wave through two ds_bpermute butterflies, the two channel-tile waves through LDS; then one
        // pixel per thread, stored coalesced
        float* red = reinterpret_cast<float*>(smem + 2 * STAGE + TBL * sizeof(unsigned)) + (a.in_scale0 ? 2 * a.C0 : 0);   // [2][256]
#if MFPA_EPI_LDS
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(epi + 2 * BN + wn * 32 + 4 * g);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(epi + 2 * BN + wn * 32 + 16 + 4 * g);
#else
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(a.w1x1 + wn * 32 + 4 * g);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(a.w1x1 + wn * 32 + 16 + 4 * g);
#endif
  #pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          float v = 0.f;
  #pragma unroll
          for (int j = 0; j < 4; ++j) v += acc[0][pt][j] * w0[j] + acc[1][pt][j] * w1[j];
          v += __shfl_xor(v, 16);
          v += __shfl_xor(v, 32);
          if (g == 0) red[wn * 256 + wm * WPXW + pt * 16 + p] = v;
        }
        __syncthreads();
        if (tid < 256) {
          const int gy = ey0 + tid / PW, gx = ex0p + tid % PW;
          if (gy < a.H && gx < a.W) a.y1x1[((size_t)eb * a.H + gy) * a.W + gx] = red[tid] + red[256 + tid] + a.b1x1;
        }
        __syncthreads();                                               // red is reused by the next tile
      }
    }
  };

  __syncthreads();
  if constexpr (ROWS) {
    using D0 = std::integral_constant<int, 0>;
    using D1 = std::integral_constant<int, 1>;
    using D2 = std::integral_constant<int, 2>;
#pragma unroll
    for (int u = 0; u < NB - 1; ++u) read_unit(xu[u], smem, u, 0);
    if constexpr (PERSIST) {
      for (;;) {
        const int tile_n = tile + (int)gridDim.x;
        const bool has_next = tile_n < ntiles;
        const Tile N = make_tile(has_next ? tile_n : tile);
        for (int chunk = 0; chunk < nchunks; chunk += 2) {
          period(D0{}, S0{}, chunk);
          period(D1{}, S1{}, chunk);
          period(D2{}, S0{}, chunk);
          if (chunk + 2 >= nchunks) S = N;                               // the tile's last chunk stages the next tile's first
          period(D0{}, S1{}, chunk + 1);
          period(D1{}, S0{}, chunk + 1);
          period(D2{}, S1{}, chunk + 1);
        }
        epilogue();
        if (!has_next) break;
        tile = tile_n; eb = S.b; ey0 = S.y0; ex0p = S.x0p;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = floatx4{0.f, 0.f, 0.f, 0.f};
      }
    } else {
    for (int chunk = 0; chunk < nchunks; chunk += 2) {                 // nchunks is even (C_in % 64 == 0, checked by the dispatcher)
      period(D0{}, S0{}, chunk);
      period(D1{}, S1{}, chunk);
      period(D2{}, S0{}, chunk);
      period(D0{}, S1{}, chunk + 1);
      period(D1{}, S0{}, chunk + 1);
      period(D2{}, S1{}, chunk + 1);
    }
    epilogue();
    }
  } else if constexpr (WMW == 4) {
    read_x(fx0, smem, 0, 0);
    for (;;) {
      // the next tile of this workgroup (past the end: this tile again -- its first chunk is requested once more and never used)
      const int tile_n = tile + (int)gridDim.x;
      const bool has_next = tile_n < ntiles;
      const Tile N = make_tile(has_next ? tile_n : tile);
      for (int chunk = 0; chunk < nchunks; chunk += 2) {                 // nchunks is even (C_in % 64 == 0, checked by the dispatcher)
        tap_body4(std::integral_constant<int, 0>{}, S0{}, chunk);
        tap_body4(std::integral_constant<int, 1>{}, S0{}, chunk);
        tap_body4(std::integral_constant<int, 2>{}, S0{}, chunk);
        tap_body4(std::integral_constant<int, 3>{}, S0{}, chunk);
        tap_body4(std::integral_constant<int, 4>{}, S0{}, chunk);
        tap_body4(std::integral_constant<int, 5>{}, S0{}, chunk);
        tap_body4(std::integral_constant<int, 6>{}, S0{}, chunk);
        tap_body4(std::integral_constant<int, 7>{}, S0{}, chunk);
        tap_body4(std::integral_constant<int, 8>{}, S0{}, chunk);
        if (chunk + 2 >= nchunks) S = N;                                 // the tile's last chunk stages the next tile's first
        tap_body4(std::integral_constant<int, 0>{}, S1{}, chunk + 1);
        tap_body4(std::integral_constant<int, 1>{}, S1{}, chunk + 1);
        tap_body4(std::integral_constant<int, 2>{}, S1{}, chunk + 1);
        tap_body4(std::integral_constant<int, 3>{}, S1{}, chunk + 1);
        tap_body4(std::integral_constant<int, 4>{}, S1{}, chunk + 1);
        tap_body4(std::integral_constant<int, 5>{}, S1{}, chunk + 1);
        tap_body4(std::integral_constant<int, 6>{}, S1{}, chunk + 1);
        tap_body4(std::integral_constant<int, 7>{}, S1{}, chunk + 1);
        tap_body4(std::integral_constant<int, 8>{}, S1{}, chunk + 1);
      }
      stamp(10);
      if (!(MFPA_EXP_FLAG(a.dbg, 32) && v_never(acc[0][0][0] + acc[1][PT - 1][3]))) epilogue();
      stamp(11);
      if (!has_next) break;
      tile = tile_n; eb = S.b; ey0 = S.y0; ex0p = S.x0p;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
#ifdef MFPA_EXPERIMENTS
    if (stamping) {
      for (int i = 0; i < stamp_n; ++i) mfpa_conv_stamps[1 + i] = tsbuf[i];
      mfpa_conv_stamps[0] = stamp_n;
    }
#endif
  } else if constexpr (PERSIST) {                                    // WMW = 2, tap-by-tap, persistent (MFPA_WD16_PERSIST2)
    read_x(fx0, smem, 0, 0);
    for (;;) {
      const int tile_n = tile + (int)gridDim.x;
      const bool has_next = tile_n < ntiles;
      const Tile N = make_tile(has_next ? tile_n : tile);
      for (int chunk = 0; chunk < nchunks; ++chunk) {
        if (chunk + 1 >= nchunks) S = N;                                 // the tile's last chunk stages the next tile's first
        tap_body(std::integral_constant<int, 0>{}, chunk);
        tap_body(std::integral_constant<int, 1>{}, chunk);
        tap_body(std::integral_constant<int, 2>{}, chunk);
        tap_body(std::integral_constant<int, 3>{}, chunk);
        tap_body(std::integral_constant<int, 4>{}, chunk);
        tap_body(std::integral_constant<int, 5>{}, chunk);
        tap_body(std::integral_constant<int, 6>{}, chunk);
        tap_body(std::integral_constant<int, 7>{}, chunk);
        tap_body(std::integral_constant<int, 8>{}, chunk);
      }
      epilogue();
      if (!has_next) break;
      tile = tile_n; eb = S.b; ey0 = S.y0; ex0p = S.x0p;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
  } else {
  read_x(fx0, smem, 0, 0);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    tap_body(std::integral_constant<int, 0>{}, chunk);
    tap_body(std::integral_constant<int, 1>{}, chunk);
    tap_body(std::integral_constant<int, 2>{}, chunk);
    tap_body(std::integral_constant<int, 3>{}, chunk);
    tap_body(std::integral_constant<int, 4>{}, chunk);
    tap_body(std::integral_constant<int, 5>{}, chunk);
    tap_body(std::integral_constant<int, 6>{}, chunk);
    tap_body(std::integral_constant<int, 7>{}, chunk);
    tap_body(std::integral_constant<int, 8>{}, chunk);
  }
  epilogue();
  }
}

#ifndef MFPA_CONV_WS64
#define MFPA_CONV_WS64 1             // 64-channel inference launches on conv_ws64_kernel (csrc/unet_ws.hip); 0: conv_wd16_kernel<.., WMW = 4> (A/B builds)
#endif
#ifndef MFPA_CONV_WS_C1
#define MFPA_CONV_WS_C1 1            // the UNet's first two layers (1 -> 64 computed in the loader waves on the matrix cores, 64 -> 64) on conv_ws64_kernel<C1SRC>
#endif
#ifndef MFPA_CONV_WS_ALL
#define MFPA_CONV_WS_ALL 128         // conv_ws64_kernel also for outputs of 128 channels and more with at most this many input channels (0 = never): per layer,
                                     // 64 clips: 64 -> 128 @ 128 x 125 431 -> 380 us, 128 -> 128 725 -> 704, 128 -> 256 @ 64 x 62 360 -> 345; from 256 input channels on it loses
#endif
#ifndef MFPA_CONV_WD16_64
#define MFPA_CONV_WD16_64 64         // conv_wd16_kernel<.., WMW = 4> for 64-channel layers with at least this many input channels (no fused first layer / OutConv); 0 = off
#endif
#ifndef MFPA_CONV_WD16_ROWS
#define MFPA_CONV_WD16_ROWS 512
#endif
template <int PH, int PW, int WMW = 2>
int launch_wd16(ConvArgs& a, hipStream_t s) {
  a.tiles_x = (a.W + PW - 1) / PW;
  a.tiles_y = (a.H + PH - 1) / PH;
  static const int dbg_env = MFPA_EXP_ENV("MFPA_CONV_DBG", 0);         // experiments builds: 8 skip MFMAs, 32 skip the epilogue, 64 skip halo loads,
  a.dbg = dbg_env;                                                     //   128 weight loads in the prologue only, 256 skip the halo split
  static const int stagger_env = MFPA_EXP_ENV("MFPA_CONV_STAGGER", 0);
  a.dbg_stagger = stagger_env;
  if ((long long)a.tiles_x * a.tiles_y * a.B > 0x7fffffffLL) return MFPA_EINVAL;
  constexpr int HP = (PW + 2) * (PH + 2);
  constexpr int A_F4 = (HP * (KC / 4) + 511) / 512;
  constexpr int HPS = A_F4 * 64;
  constexpr int PLANE = ((HPS * 16 + 255) / 256) * 256;
  const size_t lds = 2 * (size_t)(2 * (4 * PLANE + 256)) + (size_t)2 * A_F4 * 512 * sizeof(unsigned) +     // two halo stages + the slot offsets
                     (a.in_scale0 ? (size_t)2 * a.C0 * sizeof(float) : 0) +                                                       // + the on-load affine
                     (a.w1x1 ? (size_t)2 * 256 * sizeof(float) : 0) +                                                             // + the fused OutConv's partial sums
                     (size_t)(2 * 32 * (8 / WMW) + 64) * sizeof(float);                                                           // + the epilogue's constants
  dim3 grid((unsigned)((long long)a.tiles_x * a.tiles_y * a.B), (unsigned)(a.Cout / (32 * (8 / WMW))));
#ifdef MFPA_EXPERIMENTS
  a.dbg_lds_stamps = (int)lds;
  const_cast<size_t&>(lds) += 4096;
#endif
  static const int persist_env = MFPA_EXP_ENV("MFPA_CONV_WD16_PERSIST", 1);      // experiments: 0 = one workgroup per tile
  const bool side_ = a.x0_bf16 || a.x1_bf16 || a.y_bf16 || a.stats_part;
  const int cin_ = a.C0 + a.C1;
  const bool rows_ = WMW == 2 && MFPA_CONV_WD16_ROWS > 0 && cin_ % 64 == 0 && cin_ >= MFPA_CONV_WD16_ROWS;
  const bool persist2 = WMW == 2 && ((!side_ && !a.plain && ((MFPA_WD16_PERSIST2 != 0 && !rows_) || (MFPA_WD16_PERSIST_ROWS != 0 && rows_))) ||
                                     (MFPA_WD16_PERSIST_PLAIN != 0 && a.plain));
  if ((WMW == 4 && persist_env) || persist2) {                         // persistent: one workgroup per CU (and output-channel tile) walks the tiles
    const int cus = mfpa_current_device_cus();
    const unsigned per = (unsigned)((cus > 0 ? cus : 256) / (int)grid.y);
    if (per >= 1 && grid.x > per) grid.x = per;
  }
  // the ROWS loop form from 512 input channels up (MFPA_CONV_WD16_ROWS = that threshold; 0 = never): same-call pairs on the UNet's layers,
  // 64 clips: +2 .. +4 % at 512 / 1024 input channels, -1 .. -4 % at 64 .. 256 (its longer pipeline fill costs more than the halved
  // fragment reads return when a tile has only 2 .. 8 chunks)
  static const int rows_min = MFPA_EXP_ENV("MFPA_CONV_WD16_ROWS", MFPA_CONV_WD16_ROWS);
  const int cin = a.C0 + a.C1;
  const bool side = a.x0_bf16 || a.x1_bf16 || a.y_bf16 || a.stats_part;
  const bool rows = WMW == 2 && rows_min > 0 && cin % 64 == 0 && cin >= rows_min;
  if (a.in16) {                                                        // bf16 source: the plain-bf16 input-gradient convolutions
    if (!a.plain || a.x1_bf16 || cin % 64) return MFPA_EINVAL;         // (both sources bfloat16; source 1 IS its own bf16 copy)
    const bool fwd16 = a.in_scale0 != nullptr || a.x0_bf16 != nullptr || (a.y == nullptr && a.bz == nullptr);   // the training forward's form (AFF16)
    if constexpr (WMW == 4) {
      if (fwd16) hipLaunchKernelGGL((conv_wd16_kernel<PH, PW, false, 4, true, true, true, true>), grid, dim3(512), lds, s, a);
      else if (side) hipLaunchKernelGGL((conv_wd16_kernel<PH, PW, false, 4, true, true, true>), grid, dim3(512), lds, s, a);
      else hipLaunchKernelGGL((conv_wd16_kernel<PH, PW, false, 4, false, true, true>), grid, dim3(512), lds, s, a);
    } else {
      if (fwd16) hipLaunchKernelGGL((conv_wd16_kernel<PH, PW, true, 2, true, true, true, true>), grid, dim3(512), lds, s, a);
      else if (side) hipLaunchKernelGGL((conv_wd16_kernel<PH, PW, true, 2, true, true, true>), grid, dim3(512), lds, s, a);
      else hipLaunchKernelGGL((conv_wd16_kernel<PH, PW, true, 2, false, true, true>), grid, dim3(512), lds, s, a);
    }
    MFPA_CHECK_LAUNCH();
    return MFPA_OK;
  }
  if constexpr (WMW == 4) {
    if (cin % 64) return MFPA_EINVAL;
    if (a.plain) {
      if (side) hipLaunchKernelGGL((conv_wd16_kernel<PH, PW, false, 4, true, true>), grid, dim3(512), lds, s, a);
      else hipLaunchKernelGGL((conv_wd16_kernel<PH, PW, false, 4, false, true>), grid, dim3(512), lds, s, a);
    } else if (side) hipLaunchKernelGGL((conv_wd16_kernel<PH, PW, false, 4, true>), grid, dim3(512), lds, s, a);
    else hipLaunchKernelGGL((conv_wd16_kernel<PH, PW, false, 4, false>), grid, dim3(512), lds, s, a);
  } else if (a.plain) {
    static const int plain_rows = MFPA_EXP_ENV("MFPA_CONV_PLAIN_ROWS", 1);      // experiments: 0 = the tap-by-tap loop
    if (plain_rows && cin % 64 == 0) {
      if (side) hipLaunchKernelGGL((conv_wd16_kernel<PH, PW, true, 2, true, true>), grid, dim3(512), lds, s, a);
      else hipLaunchKernelGGL((conv_wd16_kernel<PH, PW, true, 2, false, true>), grid, dim3(512), lds, s, a);
    } else if (side) hipLaunchKernelGGL((conv_wd16_kernel<PH, PW, false, 2, true, true>), grid, dim3(512), lds, s, a);
    else hipLaunchKernelGGL((conv_wd16_kernel<PH, PW, false, 2, false, true>), grid, dim3(512), lds, s, a);
  } else if (rows) {
    if (side) hipLaunchKernelGGL((conv_wd16_kernel<PH, PW, true, 2, true>), grid, dim3(512), lds, s, a);
    else hipLaunchKernelGGL((conv_wd16_kernel<PH, PW, true, 2, false>), grid, dim3(512), lds, s, a);
  } else {
    if (side) hipLaunchKernelGGL((conv_wd16_kernel<PH, PW, false, 2, true>), grid, dim3(512), lds, s, a);
    else hipLaunchKernelGGL((conv_wd16_kernel<PH, PW, false, 2, false>), grid, dim3(512), lds, s, a);
  }
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

// ConvTranspose2d(k = 2, s = 2) forward: out[2y+dy, 2x+dx][co] = sum_ci x[y,x][ci] * w[dy,dx][co][ci] + bias[co].
// K = C_in only, so the generic kernel's one-tap-per-workgroup form re-staged the same input tile four times around a
// 4..32-iteration loop.  Here a workgroup keeps FOUR accumulator sets (one per tap) for 128 input pixels x 64 output
// channels: the input chunk is staged once per 32 channels and its fragments are reused by all four taps; a wave owns
// 32 pixels x 64 channels x 4 taps (128 accumulator VGPRs).  Two workgroups per CU overlap each other's staging.
template <int PH, int PW, int PREC, bool IO16 = false, bool PLAIN = false>
__global__ __launch_bounds__(256, 2) void convT_mfma_kernel(ConvArgs a) {
  static_assert(!PLAIN || (IO16 && PREC == 1), "the plain-bf16 form belongs to the training step's bf16-I/O instantiation");
  // IO16 (the plain-bf16 training step with its activations kept as bfloat16): the source and / or the output are bfloat16 tensors -- its
  // own instantiation, the inference kernels carry none of it
  constexpr int BM = PH * PW, BN = 64, NT = 2;
  static_assert(BM == 128, "four waves of 32 pixels");
  constexpr int A_F4 = BM * (KC / 4) / 256;            // 4
  constexpr int B_F4 = 4 * BN * (KC / 4) / 256;        // 8
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* As = reinterpret_cast<float*>(smem);          // [128][LDK]
  float* Bs = As + BM * LDK;                           // [4 taps][64][LDK]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  int bx = blockIdx.x;
  const int tx = bx % a.tiles_x; bx /= a.tiles_x;
  const int ty = bx % a.tiles_y; bx /= a.tiles_y;
  const int b = bx;
  const int n0 = blockIdx.y * BN;
  const int y0 = ty * PH, x0p = tx * PW;
  const int nchunks = a.C0 / KC;
  const int q = tid % (KC / 4);
  const bool affine = a.in_scale0 != nullptr;
  const bool in16 = IO16 && a.in16 != 0;                               // source kept as bfloat16 (the training step's activations): widened on load
  const char* xb = reinterpret_cast<const char*>(a.x0) + (size_t)b * a.H * a.W * a.C0 * (in16 ? 2 : 4);   // 32-bit offsets from the clip's base

  f32x4 areg[A_F4], breg[B_F4];
  auto load = [&](int chunk) __attribute__((always_inline)) {
    const int c0 = chunk * KC;
#pragma unroll
    for (int it = 0; it < A_F4; ++it) {
      const int pix = (tid + it * 256) / (KC / 4);
      const int gy = y0 + pix / PW, gx = x0p + pix % PW;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gy < a.H && gx < a.W) {
        const unsigned e = (unsigned)(gy * a.W + gx) * (unsigned)a.C0 + (unsigned)(c0 + 4 * q);
        if (in16) {
          const f32x2 r = *reinterpret_cast<const f32x2*>(xb + e * 2u);
          const unsigned u0 = __float_as_uint(r[0]), u1 = __float_as_uint(r[1]);
          v = f32x4{__uint_as_float(u0 << 16), __uint_as_float(u0 & 0xffff0000u), __uint_as_float(u1 << 16), __uint_as_float(u1 & 0xffff0000u)};
        } else v = *reinterpret_cast<const f32x4*>(xb + e * 4u);
      }
      areg[it] = v;
    }
#pragma unroll
    for (int it = 0; it < B_F4; ++it) {
      const int row = (tid + it * 256) / (KC / 4);     // tap * 64 + n
      const int tap = row / BN, n = row % BN;
      if (PREC == 0) breg[it] = *reinterpret_cast<const f32x4*>(a.w + ((size_t)tap * a.Cout + n0 + n) * a.C0 + c0 + 4 * q);
      else breg[it] = *reinterpret_cast<const f32x4*>(a.w + (((size_t)tap * nchunks + chunk) * a.Cout + n0 + n) * KC + 4 * (q ^ w3_swz(n)));
    }
  };
  auto store = [&](int chunk) __attribute__((always_inline)) {
    const int c0 = chunk * KC;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (affine) {
      sc = *reinterpret_cast<const f32x4*>(a.in_scale0 + c0 + 4 * q);
      sh = *reinterpret_cast<const f32x4*>(a.in_shift0 + c0 + 4 * q);
    }
#pragma unroll
    for (int it = 0; it < A_F4; ++it) {
      const int pix = (tid + it * 256) / (KC / 4);
      f32x4 v = areg[it];
      if (affine) {
        v = v * sc + sh;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
        if (a.drop_thresh) {
          const int gy = y0 + pix / PW, gx = x0p + pix % PW;
          const unsigned long long e0 = (((unsigned long long)b * a.H + gy) * a.W + gx) * a.C0 + c0 + 4 * q;
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = mfpa_keep(a.drop_seed, a.drop_thresh, e0 + k) ? v[k] * a.drop_scale : 0.f;
        }
      }
      if (PREC == 0) {
        *reinterpret_cast<f32x4*>(As + pix * LDK + 4 * q) = v;
      } else {
        bf16x4 hi, lo;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          hi[k] = (__bf16)v[k];
          lo[k] = (__bf16)(v[k] - (float)hi[k]);
        }
        char* row = reinterpret_cast<char*>(As + pix * LDK);
        *reinterpret_cast<bf16x4*>(row + 8 * q) = hi;
        if constexpr (!PLAIN) *reinterpret_cast<bf16x4*>(row + 64 + 8 * q) = lo;
        if (a.x0_bf16 != nullptr && blockIdx.y == 0) {                 // training forward: the activated input's bf16 copy (its weight gradient's operand)
          const int gy = y0 + pix / PW, gx = x0p + pix % PW;
          if (gy < a.H && gx < a.W)
            *reinterpret_cast<bf16x4*>(a.x0_bf16 + (((size_t)b * a.H + gy) * a.W + gx) * (size_t)a.C0 + c0 + 4 * q) = hi;
        }
      }
    }
#pragma unroll
    for (int it = 0; it < B_F4; ++it) {
      const int row = (tid + it * 256) / (KC / 4);
      *reinterpret_cast<f32x4*>(Bs + row * LDK + 4 * q) = breg[it];      // PREC 1: rows are pre-split [32 hi | 32 lo]
    }
  };

  floatx16 acc[4][NT];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][nt][r] = 0.f;

  const int a_off = (wave * 32 + li) * LDK + 4 * lh;
  const int b_off = li * LDK + 4 * lh;

  load(0);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    __syncthreads();                                   // the previous chunk's fragment reads are done
    store(chunk);
    __syncthreads();
    if (chunk + 1 < nchunks) load(chunk + 1);          // lands during the MFMA block
    if constexpr (PLAIN) {
      // round 6, the plain-bf16 training step (mfpa_conv_desc.precision 2): one bf16 MFMA per product on the hi halves, like its 3x3 convolutions
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(As + a_off) + 32 * s);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(Bs + (t * BN + nt * 32) * LDK + b_off) + 32 * s);
            acc[t][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t][nt], 0, 0, 0);
          }
      }
    } else if constexpr (PREC == 1) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const char* ar = reinterpret_cast<const char*>(As + a_off) + 32 * s;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ar);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(ar + 64);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const char* br = reinterpret_cast<const char*>(Bs + (t * BN + nt * 32) * LDK + b_off) + 32 * s;
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(br);
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(br + 64);
            acc[t][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t][nt], 0, 0, 0);
            acc[t][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t][nt], 0, 0, 0);
            acc[t][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t][nt], 0, 0, 0);
          }
      }
    } else {
#pragma unroll
      for (int s = 0; s < KC / 8; ++s) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(As + a_off + 8 * s);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const f32x4 bf = *reinterpret_cast<const f32x4*>(Bs + (t * BN + nt * 32) * LDK + b_off + 8 * s);
            acc[t][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc[t][nt], 0, 0, 0);
            acc[t][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc[t][nt], 0, 0, 0);
            acc[t][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc[t][nt], 0, 0, 0);
            acc[t][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc[t][nt], 0, 0, 0);
          }
      }
    }
  }
  // epilogue: D[row = pixel][col = channel]; tap t writes output pixel (2 gy + (t >> 1), 2 gx + (t & 1)).  32-bit byte offsets from
  // a scalar per-clip base (the host checks that one clip's output fits 4 GB): one offset per pixel, the four taps and the two
  // channel tiles are wave-uniform displacements of it (the first form computed a 64-bit index per stored element: 128 of them
  // against 48 MFMAs per 32-channel chunk, the largest block of vector work in this kernel)
  float sc[NT], sh[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + nt * 32 + li;
    sc[nt] = a.scale ? a.scale[n] : 1.f;
    sh[nt] = a.shift ? a.shift[n] : 0.f;
  }
  const bool y16 = IO16 && a.y == nullptr;                             // the output as bfloat16 only (a.y_bf16; checked by mfpa_conv_mfma)
  const unsigned esz = y16 ? 2u : 4u;
  char* yb = (y16 ? reinterpret_cast<char*>(a.y_bf16) : reinterpret_cast<char*>(a.y)) + (size_t)b * (2 * a.H) * (2 * a.W) * a.Cout * esz;
  const unsigned cout4 = (unsigned)a.Cout * esz, row4 = 2u * (unsigned)a.W * cout4;
  const unsigned nb = (unsigned)(n0 + li) * esz;
  if (y16) {
    // bfloat16 output: a lane pair (channels 2k, 2k + 1) swaps one value per two accumulator rows (pixels m, m + 1 -- neighbours in the
    // same patch row), so that the even lane stores both channels of pixel m and the odd lane both channels of pixel m + 1 as one 4-byte
    // piece each (2-byte stores per lane made this write-bound kernel 20 % slower than its float32 form)
    const int odd = li & 1;
    const unsigned nb2 = (unsigned)(n0 + (li & ~1)) * 2u;
#pragma unroll
    for (int rp = 0; rp < 16; rp += 2) {
      const int m = wave * 32 + (rp & 3) + 8 * (rp >> 2) + 4 * lh + odd;
      const int gy = y0 + m / PW, gx = x0p + m % PW;
      const bool live = gy < a.H && gx < a.W;
      char* yp = yb + ((unsigned)(2 * gy) * row4 + (unsigned)(2 * gx) * cout4 + nb2);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          float v0 = acc[t][nt][rp] * sc[nt] + sh[nt], v1 = acc[t][nt][rp + 1] * sc[nt] + sh[nt];
          if (a.relu) { v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f; }
          const float send = odd ? v0 : v1;
          const float recv = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send), 0xB1, 0xF, 0xF, true));   // quad_perm [1, 0, 3, 2]
          bf16x2 h;
          h[0] = (__bf16)(odd ? recv : v0);
          h[1] = (__bf16)(odd ? v1 : recv);
          if (live) *reinterpret_cast<bf16x2*>(yp + ((t >> 1) * row4 + (t & 1) * cout4 + nt * 64u)) = h;
        }
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    const int gy = y0 + m / PW, gx = x0p + m % PW;
    if (gy < a.H && gx < a.W) {
      char* yp = yb + ((unsigned)(2 * gy) * row4 + (unsigned)(2 * gx) * cout4 + nb);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          float v = acc[t][nt][r] * sc[nt] + sh[nt];
          if (a.relu) v = v > 0.f ? v : 0.f;
          *reinterpret_cast<float*>(yp + ((t >> 1) * row4 + (t & 1) * cout4 + nt * 128u)) = v;
        }
    }
  }
}

template <int PH, int PW, int PREC>
int launch_convT(ConvArgs& a, hipStream_t s) {
  a.tiles_x = (a.W + PW - 1) / PW;
  a.tiles_y = (a.H + PH - 1) / PH;
  if ((long long)a.tiles_x * a.tiles_y * a.B > 0x7fffffffLL) return MFPA_EINVAL;
  const size_t lds = sizeof(float) * ((size_t)PH * PW * LDK + 4 * (size_t)64 * LDK);
  dim3 grid((unsigned)((long long)a.tiles_x * a.tiles_y * a.B), (unsigned)(a.Cout / 64));
  if (a.in16 || a.y == nullptr) {
    if constexpr (PREC == 1) {
      if (a.plain) hipLaunchKernelGGL((convT_mfma_kernel<PH, PW, PREC, true, true>), grid, dim3(256), lds, s, a);     // round 6: the plain-bf16 training step
      else hipLaunchKernelGGL((convT_mfma_kernel<PH, PW, PREC, true>), grid, dim3(256), lds, s, a);
    } else return MFPA_EINVAL;
  } else hipLaunchKernelGGL((convT_mfma_kernel<PH, PW, PREC>), grid, dim3(256), lds, s, a);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

// First layer: 1 input channel -> Cout (multiple of 4), fused spectrogram normalisation.
// 16 lanes per pixel x 4 channels per lane... generalised: Cout/4 lanes per pixel.
__global__ __launch_bounds__(256) void conv3x3_c1_kernel(const float* __restrict__ x32, const double* __restrict__ spec64,
                                                         const double* __restrict__ denom, int per_clip, int B, int H,
                                                         int W, const float* __restrict__ w, int Cout,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         int relu, float* __restrict__ y, int y16, float* __restrict__ stats_part) {
  // stats_part (training forward): one row [2][Cout] per workgroup = (sum, sum of squares) of this image row's outputs per channel, float64
  // inside the workgroup -- mfpa_conv_stats_reduce / mfpa_conv_stats_bn_finish turn the B * H rows into the BatchNorm statistics without
  // the 200 us pass over the output
  const int lanes_per_pix = Cout / 4;
  double st_s[4] = {0., 0., 0., 0.}, st_q[4] = {0., 0., 0., 0.};
  const int pix_per_block = 256 / lanes_per_pix;
  const int sub = threadIdx.x % lanes_per_pix, pl = threadIdx.x / lanes_per_pix;
  float4 wt[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const float4*>(w + (size_t)t * Cout + 4 * sub);
  const float4 sc = scale ? *reinterpret_cast<const float4*>(scale + 4 * sub) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 sh = shift ? *reinterpret_cast<const float4*>(shift + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
  double gden = 1.0;
  if (spec64 && denom && !per_clip) {
    gden = 0.0;
    for (int i = 0; i < B; ++i) gden = fmax(gden, denom[i]);
  }
  // one workgroup per (clip, bin row): no 64-bit index divisions in the pixel loop
  const int b = blockIdx.x / H, gy = blockIdx.x % H;
  const double den = (spec64 && denom) ? (per_clip ? denom[b] : gden) : 1.0;
  const int iters = (W + pix_per_block - 1) / pix_per_block;
  for (int itr = 0; itr < iters; ++itr) {              // uniform trip count: every lane takes part in the shuffles
    const int gx_raw = itr * pix_per_block + pl;
    const bool live = gx_raw < W;
    const int gx = live ? gx_raw : W - 1;
    const size_t p = ((size_t)b * H + gy) * W + gx;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lanes_per_pix >= 9) {
      // the lanes of a pixel share its 3x3 input window: lane `sub` < 9 loads (and normalises) tap `sub`, the others
      // receive it by shuffle -- one float64 division per lane instead of nine
      float mine = 0.f;
      if (sub < 9) {
        const int yy = gy + sub / 3 - 1, xx = gx + sub % 3 - 1;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
          const size_t o = ((size_t)b * H + yy) * W + xx;
          mine = spec64 ? (float)(spec64[o] / den) : x32[o];
        }
      }
      const int base = (threadIdx.x & 63) - sub;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float v = __shfl(mine, base + t);
        acc.x += v * wt[t].x;
        acc.y += v * wt[t].y;
        acc.z += v * wt[t].z;
        acc.w += v * wt[t].w;
      }
    } else {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = gy + t / 3 - 1, xx = gx + t % 3 - 1;
        float v = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
          const size_t o = ((size_t)b * H + yy) * W + xx;
          v = spec64 ? (float)(spec64[o] / den) : x32[o];
        }
        acc.x += v * wt[t].x;
        acc.y += v * wt[t].y;
        acc.z += v * wt[t].z;
        acc.w += v * wt[t].w;
      }
    }
    float4 o4;
    o4.x = acc.x * sc.x + sh.x;
    o4.y = acc.y * sc.y + sh.y;
    o4.z = acc.z * sc.z + sh.z;
    o4.w = acc.w * sc.w + sh.w;
    if (relu) {
      o4.x = fmaxf(o4.x, 0.f);
      o4.y = fmaxf(o4.y, 0.f);
      o4.z = fmaxf(o4.z, 0.f);
      o4.w = fmaxf(o4.w, 0.f);
    }
    if (live && stats_part != nullptr) {
      st_s[0] += (double)o4.x; st_s[1] += (double)o4.y; st_s[2] += (double)o4.z; st_s[3] += (double)o4.w;
      st_q[0] += (double)o4.x * (double)o4.x; st_q[1] += (double)o4.y * (double)o4.y;
      st_q[2] += (double)o4.z * (double)o4.z; st_q[3] += (double)o4.w * (double)o4.w;
    }
    if (live) {
      if (y16) {                                                         // the output kept as bfloat16 (the plain-bf16 training step's activations)
        bf16x4 h;
        h[0] = (__bf16)o4.x; h[1] = (__bf16)o4.y; h[2] = (__bf16)o4.z; h[3] = (__bf16)o4.w;
        *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(y) + (size_t)p * Cout + 4 * sub) = h;
      } else *reinterpret_cast<float4*>(y + (size_t)p * Cout + 4 * sub) = o4;
    }
  }
  if (stats_part != nullptr) {                                           // (uniform: a kernel argument)
    __shared__ double red[256 * 8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[threadIdx.x * 8 + k] = st_s[k]; red[threadIdx.x * 8 + 4 + k] = st_q[k]; }
    __syncthreads();
    if (pl == 0) {                                                       // fixed order over the workgroup's pixel slots: deterministic
      for (int r = 1; r < pix_per_block; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          st_s[k] += red[(r * lanes_per_pix + sub) * 8 + k];
          st_q[k] += red[(r * lanes_per_pix + sub) * 8 + 4 + k];
        }
      float* row = stats_part + (size_t)blockIdx.x * 2 * Cout;
      *reinterpret_cast<float4*>(row + 4 * sub) = make_float4((float)st_s[0], (float)st_s[1], (float)st_s[2], (float)st_s[3]);
      *reinterpret_cast<float4*>(row + Cout + 4 * sub) = make_float4((float)st_q[0], (float)st_q[1], (float)st_q[2], (float)st_q[3]);
    }
  }
}

__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ x, int B, int H, int W, int C,
                                                       float* __restrict__ y) {
  const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
  const long long total = (long long)B * Ho * Wo * C4;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int c4 = (int)(e % C4);
    long long p = e / C4;
    const int xo = (int)(p % Wo); p /= Wo;
    const int yo = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const float* base = x + (((size_t)b * H + 2 * yo) * W + 2 * xo) * C + 4 * c4;
    const float4 v00 = *reinterpret_cast<const float4*>(base);
    const float4 v01 = *reinterpret_cast<const float4*>(base + C);
    const float4 v10 = *reinterpret_cast<const float4*>(base + (size_t)W * C);
    const float4 v11 = *reinterpret_cast<const float4*>(base + (size_t)W * C + C);
    float4 o;
    o.x = fmaxf(fmaxf(v00.x, v01.x), fmaxf(v10.x, v11.x));
    o.y = fmaxf(fmaxf(v00.y, v01.y), fmaxf(v10.y, v11.y));
    o.z = fmaxf(fmaxf(v00.z, v01.z), fmaxf(v10.z, v11.z));
    o.w = fmaxf(fmaxf(v00.w, v01.w), fmaxf(v10.w, v11.w));
    *reinterpret_cast<float4*>(y + (size_t)e * 4) = o;
  }
}

// OutConv 1x1 to one class: C/4 lanes per pixel, float4 per lane, shuffle reduce.
__global__ __launch_bounds__(256) void conv1x1_out_kernel(const float* __restrict__ x, long long npix, int C,
                                                          const float* __restrict__ w, float bias, float* __restrict__ y) {
  const int lpp = C / 4;  // lanes per pixel (power of two <= 64)
  const int sub = threadIdx.x % lpp, pl = threadIdx.x / lpp, ppb = 256 / lpp;
  const float4 wv = *reinterpret_cast<const float4*>(w + 4 * sub);
  const long long iters = (npix + (long long)gridDim.x * ppb - 1) / ((long long)gridDim.x * ppb);
  for (long long it = 0; it < iters; ++it) {
    const long long p = (it * gridDim.x + blockIdx.x) * ppb + pl;
    float s = 0.f;
    if (p < npix) {
      const float4 v = *reinterpret_cast<const float4*>(x + (size_t)p * C + 4 * sub);
      s = v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
    }
    for (int o = lpp >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (sub == 0 && p < npix) y[p] = s + bias;
  }
}

template <int BN, int PH, int PW, int WM, int WN, int MODE, int PREC, bool C1SRC = false, int MT = 2, bool BDIR = false>
int launch_conv(ConvArgs& a, int taps_y, hipStream_t s) {
  constexpr int HALO = (MODE == 0) ? 1 : 0;
  constexpr int HP = (PW + 2 * HALO) * (PH + 2 * HALO);
  a.tiles_x = (a.W + PW - 1) / PW;
  a.tiles_y = (a.H + PH - 1) / PH;
  static const int dbg_env = MFPA_EXP_ENV("MFPA_CONV_DBG", 0);
  a.dbg = dbg_env;
  if ((long long)a.tiles_x * a.tiles_y * a.B > 0x7fffffffLL) return MFPA_EINVAL;
  constexpr bool ADB = BDIR || conv_is_pipe(BN, PH, PW, WM, WN, MODE, PREC, MT);    // PIPE of the kernel: two padded halo stages
  constexpr int THREADS = 64 * WM * WN;
  constexpr int HPS = ADB ? ((HP * (KC / 4) + THREADS - 1) / THREADS) * (THREADS / (KC / 4)) : HP;
  const size_t lds = sizeof(float) * ((size_t)(ADB ? 2 : 1) * HPS * LDK + (BDIR ? 0 : 2) * (size_t)BN * LDK + (C1SRC ? (PH + 4) * (PW + 4) + 9 * 64 : 0));
  dim3 grid((unsigned)((long long)a.tiles_x * a.tiles_y * a.B), (unsigned)(taps_y * (a.Cout / BN)));
  hipLaunchKernelGGL((conv_mfma_kernel<BN, PH, PW, WM, WN, MODE, PREC, C1SRC, MT, BDIR>), grid, dim3(64 * WM * WN), lds, s, a);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

// Which bf16x3 weight image does the fastest kernel for this shape read?  2 = the fragment-ordered image of the 16 x 16 x 32
// weights-direct kernel (conv_wd16_kernel: 3x3 convolution, 128-channel output tiles, >= 64 input channels, the 8 x 32 patches of the
// wide levels or the 16 x 16 patches of the 16 x 15 level), 1 = the image of the 32 x 32 x 16 weights-direct form (BDIR; with
// MFPA_CONV_WD16 = 0, and the 64-channel tiles with MFPA_CONV_BDIR64), 0 = the row image.
static int conv_weight_layout(int H, int W, int Cin, int Cout, int mode, int precision) {
#ifdef MFPA_CONV_NO_BDIR
  return 0;
#endif
  if (mode != 0 || precision != 1 || Cin < MFPA_CONV_BIG_MIN_CIN) return 0;
  if (Cout % 128) {                                                    // 64-channel output tiles
    static const int wd64 = MFPA_EXP_ENV("MFPA_CONV_WD16_64", MFPA_CONV_WD16_64);
    // (wd64 = the smallest C_in that takes it.  Before its tile loop was persistent, two-chunk tiles ran better on the plain-loop
    // kernel, whose two co-resident workgroups hide each other's prologue and epilogue; with the persistent loop: 64 -> 64 @ 257 x 251,
    // 64 clips, 1366 -> 1008 us)
    if (MFPA_CONV_WD16 && wd64 > 0 && Cin >= wd64 && Cout % 64 == 0 && Cin % 64 == 0 && W > 16 && H >= 8) return 2;
    return (MFPA_CONV_BDIR64 && Cout % 64 == 0 && W > 16 && H >= 8) ? 1 : 0;
  }
  if (W > 16 && H >= 8) return MFPA_CONV_WD16 ? 2 : 1;
  if (W <= 16 && H >= 16 && MFPA_CONV_BOTTLENECK8) return MFPA_CONV_WD16 ? 2 : 1;
  return 0;
}

// Tile choice.  Waves always own 64 pixels x 64 channels.
//   Cout % 128 == 0: 128-channel tiles; 8 waves on 8x32-pixel patches (256 x 128: half the weight traffic and
//                    barriers per MFMA) when K = 9*Cin is long enough to amortise the prologue/epilogue of a
//                    one-workgroup-per-CU kernel, else 4 waves on 4x32 patches (two workgroups per CU overlap).
//   otherwise      : 64-channel tiles, 4 waves stacked along M on 8x32 patches (256 x 64).
//   W <= 16 (the 16x15 bottleneck): 8x16 patches.
template <int MODE, int PREC>
int dispatch_conv_p(ConvArgs& a, hipStream_t s) {
  static const int ct_old = MFPA_EXP_ENV("MFPA_CONVT_OLD", 0);   // experiments: generic kernel
  if (MODE == 1 && !ct_old) return a.W > 16 ? launch_convT<4, 32, PREC>(a, s) : launch_convT<8, 16, PREC>(a, s);
  const int taps_y = (MODE == 1) ? 4 : 1;
  const bool bn128 = (a.Cout % 128 == 0);
  if constexpr (MODE == 0 && PREC == 1) {
    if (a.w_frag == 2) {   // the 16 x 16 x 32 weights-direct kernel and its image
      if (conv_weight_layout(a.H, a.W, a.C0 + a.C1, a.Cout, 0, 1) != 2 || (a.w1x1 && bn128)) return MFPA_EINVAL;
      if (a.c1_x32 || a.c1_spec64) {      // fused first layer + fragment image: conv_ws64_kernel<C1SRC> only (mfpa_conv_c1_layout() says where)
        return (MFPA_CONV_WS64 && MFPA_CONV_WS_C1 && mfpa_unet::conv_ws64_serves(a)) ? mfpa_unet::launch_conv_ws64(a, s) : MFPA_EINVAL;
      }
      if (!bn128) {
        // round 5: the wave-specialised kernel (csrc/unet_ws.hip: 4 compute waves of 128 px x 32 ch + 4 loader waves) takes the inference launches
        static const int ws64 = MFPA_EXP_ENV("MFPA_CONV_WS64", MFPA_CONV_WS64);
        if (ws64 && mfpa_unet::conv_ws64_serves(a)) return mfpa_unet::launch_conv_ws64(a, s);
        if (a.x0_split || a.x1_split || a.y_split || a.y_pool_split) return MFPA_EINVAL;   // (a run-time A/B switch sent a split-layout launch here: only conv_ws64_kernel knows that layout)
        return launch_wd16<8, 32, 4>(a, s);                             // 64-channel output tiles: 4 x 2 waves of 64 px x 32 ch
      }
      {
        // the same kernel for 128-channel-multiple outputs (two or more workgroup rows of 64 channels): per-layer A/B, see NOTES.md R5
        static const int ws_all = MFPA_EXP_ENV("MFPA_CONV_WS_ALL", MFPA_CONV_WS_ALL);
        const int cin_ = a.C0 + a.C1;
        if (ws_all && cin_ <= ws_all && mfpa_unet::conv_ws64_serves(a)) return mfpa_unet::launch_conv_ws64(a, s);
      }
      if (a.x0_split || a.x1_split || a.y_split || a.y_pool_split) return MFPA_EINVAL;     // as above
      if (a.W > 16) return launch_wd16<8, 32>(a, s);
      return launch_wd16<16, 16>(a, s);
    }
    if (a.w_frag) {        // the caller packed the fragment-ordered image: only the BDIR kernels read it (conv_weight_layout() said so)
      if (!conv_weight_layout(a.H, a.W, a.C0 + a.C1, a.Cout, 0, 1) || a.c1_x32 || a.c1_spec64) return MFPA_EINVAL;
      if (!bn128) return launch_conv<64, 8, 32, 4, 2, 0, 1, false, 2, true>(a, 1, s);     // 64-channel layers: 8 waves of 64 px x 32 ch
      if (a.W > 16) return launch_conv<128, 8, 32, 2, 4, 0, 1, false, 4, true>(a, 1, s);
      return launch_conv<128, 16, 16, 2, 4, 0, 1, false, 4, true>(a, 1, s);
    }
  }
  static const int wm_env = MFPA_EXP_ENV("MFPA_CONV_WM", 0);   // experiments
  const int cin = a.C0 + a.C1;
  const bool big = (wm_env == 4) || (wm_env == 0 && cin >= ((PREC == 1 && MODE == 0) ? MFPA_CONV_BIG_MIN_CIN : 256));   // the plain loop keeps round 1's measured threshold
  constexpr int WN64 = MFPA_CONV_WN64;   // 2: the pipelined 8-wave shape with waves of 64 px x 32 ch (measured 5-10 % slower than the 4-wave shape)
  if (MODE == 0 && (a.c1_x32 || a.c1_spec64)) {        // checked by the caller: C0 == 64, C1 == 0, Cout == 64, W > 16, H >= 8
    return launch_conv<64, 8, 32, 4, WN64, 0, PREC, true>(a, 1, s);
  }
  if (a.W > 16 && a.H >= 8) {
    if (!bn128) return launch_conv<64, 8, 32, 4, WN64, MODE, PREC>(a, taps_y, s);
#if MFPA_CONV_MT4 || defined(MFPA_EXPERIMENTS)
    if constexpr (MODE == 0 && PREC == 1) {
      static const int mt4 = MFPA_EXP_ENV("MFPA_CONV_MT4", MFPA_CONV_MT4);
      if (big && mt4) return launch_conv<128, 8, 32, 2, 2, MODE, PREC, false, 4>(a, taps_y, s);
    }
#endif
    if (big) return launch_conv<128, 8, 32, 4, 2, MODE, PREC>(a, taps_y, s);
    return launch_conv<128, 4, 32, 2, 2, MODE, PREC>(a, taps_y, s);
  }
  if (a.W > 16) {
    return bn128 ? launch_conv<128, 4, 32, 2, 2, MODE, PREC>(a, taps_y, s) : launch_conv<64, 4, 32, 2, 1, MODE, PREC>(a, taps_y, s);
  }
#if MFPA_CONV_BOTTLENECK8
#if MFPA_CONV_MT4 || defined(MFPA_EXPERIMENTS)
  if constexpr (MODE == 0 && PREC == 1) {
    static const int mt4b = MFPA_EXP_ENV("MFPA_CONV_MT4", MFPA_CONV_MT4);
    if (bn128 && a.H >= 16 && mt4b >= 2) return launch_conv<128, 16, 16, 2, 2, MODE, PREC, false, 4>(a, taps_y, s);
  }
#endif
  if (bn128 && MODE == 0 && PREC == 1 && a.H >= 16) return launch_conv<128, 16, 16, 4, 2, MODE, PREC>(a, taps_y, s);
#endif
  return bn128 ? launch_conv<128, 8, 16, 2, 2, MODE, PREC>(a, taps_y, s) : launch_conv<64, 8, 16, 2, 1, MODE, PREC>(a, taps_y, s);
}

template <int MODE>
int dispatch_conv(ConvArgs& a, hipStream_t s, int precision = 0) {
  // the halo loader packs pixel coordinates into 16 bits each and addresses one clip's input with 32-bit byte offsets
  if (a.H > 32767 || a.W > 32767) return MFPA_EINVAL;
  if ((MODE == 2 ? 4LL : 1LL) * a.H * a.W * a.C0 * 4 > 0xffffffffLL || 1LL * a.H1 * a.W1 * a.C1 * 4 > 0xffffffffLL) return MFPA_EINVAL;
  return precision ? dispatch_conv_p<MODE, 1>(a, s) : dispatch_conv_p<MODE, 0>(a, s);
}

}  // namespace

extern "C" {

int mfpa_conv3x3_bn_relu(const float* x0, int C0, const float* x1, int C1, int H1, int W1, int B, int H, int W,
                         const float* w, int Cout, const float* scale, const float* shift, int relu, int precision,
                         float* y, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!x0 || !w || !y || B < 0 || H < 1 || W < 1) return MFPA_EINVAL;
  if (C0 < KC || C0 % KC || C1 < 0 || C1 % KC || Cout < 64 || Cout % 64) return MFPA_EINVAL;
  if (C1 > 0 && (!x1 || H1 < 1 || W1 < 1 || H1 > H || W1 > W)) return MFPA_EINVAL;
  if (precision != 0 && precision != 1) return MFPA_EINVAL;
  ConvArgs a{};
  a.x0 = x0; a.x1 = C1 ? x1 : nullptr; a.w = w; a.scale = scale; a.shift = shift; a.y = y;
  a.C0 = C0; a.C1 = C1; a.H1 = C1 ? H1 : 0; a.W1 = C1 ? W1 : 0;
  a.oy1 = C1 ? (H - H1) / 2 : 0;  // F.pad(x1, [dx//2, dx-dx//2, dy//2, dy-dy//2]), unet.py:59-62
  a.ox1 = C1 ? (W - W1) / 2 : 0;
  a.B = B; a.H = H; a.W = W; a.Cout = Cout; a.relu = relu; a.yH = H; a.yW = W;
  return dispatch_conv<0>(a, mfpa_stream(stream), precision);
}

int mfpa_convT2x2(const float* x, int B, int H, int W, int Cin, const float* w, const float* bias, int Cout,
                  int precision, float* y, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!x || !w || !y || B < 0 || H < 1 || W < 1) return MFPA_EINVAL;
  if (Cin < KC || Cin % KC || Cout < 64 || Cout % 64 || (precision != 0 && precision != 1)) return MFPA_EINVAL;
  if (4LL * H * W * Cout * 4 > 0xffffffffLL || 1LL * H * W * Cin * 4 > 0xffffffffLL) return MFPA_EINVAL;   // 32-bit byte offsets inside one clip
  ConvArgs a{};
  a.x0 = x; a.w = w; a.scale = nullptr; a.shift = bias; a.y = y;
  a.C0 = Cin; a.B = B; a.H = H; a.W = W; a.Cout = Cout; a.relu = 0; a.yH = H; a.yW = W;
  return dispatch_conv<1>(a, mfpa_stream(stream), precision);
}

int mfpa_conv_mfma(const mfpa_conv_desc* d, void* stream) {
  if (!d) return MFPA_EINVAL;
  if (d->B == 0) return MFPA_OK;
  if ((!d->x0 && !d->c1_x32 && !d->c1_spec64) || !d->w || (!d->y && !d->w1x1 && !d->y_bf16) || d->B < 0 || d->H < 1 || d->W < 1) return MFPA_EINVAL;
  if (d->C0 < KC || d->C0 % KC || d->C1 < 0 || d->C1 % KC || d->Cout < 64 || d->Cout % 64) return MFPA_EINVAL;
  if (d->mode < 0 || d->mode > 2) return MFPA_EINVAL;
  if (d->C1 > 0 && (d->mode != 0 || !d->x1 || d->H1 < 1 || d->W1 < 1 || d->H1 > d->H || d->W1 > d->W)) return MFPA_EINVAL;
  if ((d->in_scale0 == nullptr) != (d->in_shift0 == nullptr)) return MFPA_EINVAL;
  if (4LL * d->H * d->W * d->Cout * 4 > 0xffffffffLL) return MFPA_EINVAL;   // the epilogue addresses one clip's output with 32-bit byte offsets
  ConvArgs a{};
  a.x0 = d->x0; a.in_scale0 = d->in_scale0; a.in_shift0 = d->in_shift0;
  a.x1 = d->C1 ? d->x1 : nullptr; a.w = d->w; a.scale = d->out_scale; a.shift = d->out_shift; a.y = d->y;
  a.C0 = d->C0; a.C1 = d->C1; a.H1 = d->C1 ? d->H1 : 0; a.W1 = d->C1 ? d->W1 : 0;
  a.oy1 = d->C1 ? (d->H - d->H1) / 2 : 0;
  a.ox1 = d->C1 ? (d->W - d->W1) / 2 : 0;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cout = d->Cout; a.relu = d->relu;
  a.yH = (d->mode != 1 && d->yH > 0) ? d->yH : d->H;
  a.yW = (d->mode != 1 && d->yW > 0) ? d->yW : d->W;
  if (a.yH > d->H || a.yW > d->W) return MFPA_EINVAL;
  if (d->drop_thresh && (!d->in_scale0 || d->mode == 2)) return MFPA_EINVAL;
  a.drop_seed = d->drop_seed; a.drop_thresh = d->drop_thresh; a.drop_scale = d->drop_scale;
  if ((d->y_pool || d->w1x1) && d->mode != 0) return MFPA_EINVAL;
  if (d->y_pool && (d->H < 2 || d->W < 2 || a.yH != d->H || a.yW != d->W)) return MFPA_EINVAL;
  if (d->w1x1 && (!d->y1x1 || d->Cout != 64)) return MFPA_EINVAL;      // the 64-channel tile holds every channel
  a.y_pool = d->y_pool; a.w1x1 = d->w1x1; a.b1x1 = d->b1x1; a.y1x1 = d->y1x1;
  if (d->c1_x32 || d->c1_spec64) {
    if (d->mode != 0 || d->C0 != 64 || d->C1 != 0 || d->Cout != 64 || d->W <= 16 || d->H < 8) return MFPA_EINVAL;
    if (!d->c1_w || !d->c1_scale || !d->c1_shift || d->in_scale0 || (d->c1_x32 && d->c1_spec64)) return MFPA_EINVAL;
    a.c1_x32 = d->c1_x32; a.c1_spec64 = d->c1_spec64; a.c1_denom = d->c1_spec64 ? d->c1_denom : nullptr;
    a.c1_w = d->c1_w; a.c1_scale = d->c1_scale; a.c1_shift = d->c1_shift;
  }
  if (d->precision < 0 || d->precision > 2) return MFPA_EINVAL;
  if (d->w_layout < 0 || d->w_layout > 2) return MFPA_EINVAL;
  if (d->w_layout != 0 && (d->mode != 0 || d->precision < 1)) return MFPA_EINVAL;
  // plain bf16: conv_wd16_kernel (mode 0: it reads the hi halves of the fragment image), or -- round 6 -- the transposed convolution of the training
  // step and its input gradient (modes 1 / 2 on the row image: the hi halves of the staged operands)
  if (d->precision == 2 && d->mode == 0 && d->w_layout != 2) return MFPA_EINVAL;
  if (d->precision == 2 && d->mode == 1 && !(d->x0_is_bf16 || !d->y)) return MFPA_EINVAL;      // (its bf16-I/O instantiation carries the plain form)
  a.w_frag = d->w_layout;
  a.plain = d->precision == 2;
  // bfloat16 sources (both of them): the plain-bf16 conv_wd16_kernel (any on-load affine / dropout is applied in float32 and rounded once), or
  // the transposed convolution (mode 1)
  if (d->x0_is_bf16 && !((d->precision == 2 && d->mode == 0 && ((d->C0 + d->C1) % 64) == 0 && !d->x1_bf16) || (d->mode == 1 && d->precision >= 1)))
    return MFPA_EINVAL;
  if (d->x0_is_bf16 && (d->c1_x32 || d->c1_spec64)) return MFPA_EINVAL;
  a.in16 = d->x0_is_bf16 ? 1 : 0;
  a.x0_split = d->x0_split ? 1 : 0; a.x1_split = d->x1_split ? 1 : 0; a.y_split = d->y_split ? 1 : 0; a.y_pool_split = d->y_pool_split ? 1 : 0;
  const bool any_split = a.x0_split || a.x1_split || a.y_split || a.y_pool_split;
  if (any_split && (d->mode != 0 || d->precision != 1 || d->w_layout != 2 || (a.x1_split && d->C1 < 1) || (a.y_split && !d->y) || (a.y_pool_split && !d->y_pool) ||
                    (a.x0_split && (d->c1_x32 || d->c1_spec64)))) return MFPA_EINVAL;
  if (d->x0_bf16 != nullptr && !((d->w_layout == 2 || (d->mode == 1 && d->precision >= 1)) && d->x0)) return MFPA_EINVAL;   // conv_wd16_kernel's loader, or the bf16x3 transposed convolution's
  if (d->x1_bf16 != nullptr && (d->w_layout != 2 || !d->x1 || d->C1 < 1)) return MFPA_EINVAL;
  // y_bf16 with y: a bf16 copy beside the float32 output; y_bf16 WITHOUT y: the output exists as bfloat16 only (conv_wd16_kernel, or the
  // bf16x3 transposed convolution)
  if (d->y_bf16 != nullptr && !(d->w_layout == 2 || (d->mode == 1 && d->precision >= 1 && !d->y))) return MFPA_EINVAL;
  if (!d->y && d->y_bf16 && (d->y_pool || d->w1x1 || d->precision == 0)) return MFPA_EINVAL;
  a.x0_bf16 = reinterpret_cast<__bf16*>(d->x0_bf16);
  a.x1_bf16 = reinterpret_cast<__bf16*>(d->x1_bf16);
  a.y_bf16 = reinterpret_cast<__bf16*>(d->y_bf16);
  if (d->stats_part != nullptr && (d->w_layout != 2 || (!d->y && !d->y_bf16))) return MFPA_EINVAL;       // only conv_wd16_kernel's epilogue writes them
  a.stats_part = d->stats_part;
  if (d->bwd_z != nullptr && (!d->stats_part || !d->bwd_scale || !d->bwd_shift || !d->bwd_mean || !d->bwd_invstd)) return MFPA_EINVAL;
  if (d->bwd_z != nullptr && d->x0_is_bf16 && (d->in_scale0 || d->x0_bf16)) return MFPA_EINVAL;   // (the training forward's form of the kernel carries no bwd_z code)
  a.bz16 = (d->bwd_z != nullptr && d->bwd_z_is_bf16) ? 1 : 0;
  a.bz = d->bwd_z; a.bz_scale = d->bwd_scale; a.bz_shift = d->bwd_shift; a.bz_mean = d->bwd_mean; a.bz_invstd = d->bwd_invstd;
  if (any_split) {                                                       // only conv_ws64_kernel reads / writes the split layout
    if (!(MFPA_CONV_WS64 && mfpa_unet::conv_ws64_serves(a)) || (a.Cout % 128 == 0 && !(MFPA_CONV_WS_ALL > 0 && d->C0 + d->C1 <= MFPA_CONV_WS_ALL))) return MFPA_EINVAL;
  }
  hipStream_t s = mfpa_stream(stream);
  const int prec = d->precision ? 1 : 0;                                 // kernel family: fp32 MFMA or the bf16 matrix cores
  if (d->mode == 0) return dispatch_conv<0>(a, s, prec);
  if (d->mode == 1) return dispatch_conv<1>(a, s, prec);
  return dispatch_conv<2>(a, s, prec);
}

#ifdef MFPA_EXPERIMENTS
int mfpa_exp_conv_stamps(unsigned long long* buf) {       // experiments build only (not in include/mfpa.h): timeline buffer, 8 slots per workgroup
  return hipMemcpyToSymbol(HIP_SYMBOL(mfpa_conv_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

int mfpa_conv_c1_layout(int H, int W) {
  if (H < 1 || W < 1) return MFPA_EINVAL;
  // the fused first-layer launch (mfpa_conv_desc.c1_*, 64 -> 64) reads the fragment image 2 exactly where conv_ws64_kernel<C1SRC> takes it
  return (MFPA_CONV_WS64 && MFPA_CONV_WS_C1 && conv_weight_layout(H, W, 64, 64, 0, 1) == 2 && W > 16 && H >= 8) ? 2 : 0;
}

int mfpa_conv_scale_folds(int H, int W, int Cin, int Cout) {
  if (H < 1 || W < 1 || Cin < 1 || Cout < 1) return MFPA_EINVAL;
  if (conv_weight_layout(H, W, Cin, Cout, 0, 1) != 2 || Cout % 64 || Cin % KC || W <= 16 || H < 8) return 0;
  // the dispatcher's own routing (dispatch_conv_p): 64-channel-multiple outputs that are not 128-multiples always, the others up to MFPA_CONV_WS_ALL input channels
  if (Cout % 128) return MFPA_CONV_WS64 ? 1 : 0;
  return (MFPA_CONV_WS_ALL > 0 && Cin <= MFPA_CONV_WS_ALL) ? 1 : 0;
}

int mfpa_conv_weight_layout(int H, int W, int Cin, int Cout, int mode, int precision) {
  if (H < 1 || W < 1 || Cin < 1 || Cout < 1) return MFPA_EINVAL;
  return conv_weight_layout(H, W, Cin, Cout, mode, precision);
}

int mfpa_conv_stats_rows(int B, int H, int W, int Cin, int Cout) {
  if (B < 0 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return MFPA_EINVAL;
  if (conv_weight_layout(H, W, Cin, Cout, 0, 1) != 2) return 0;       // no kernel that writes the partials for this shape
  const int pw = W > 16 ? 32 : 16, ph = 256 / pw;
  const long long rows = (long long)((W + pw - 1) / pw) * ((H + ph - 1) / ph) * B * (Cout % 128 == 0 ? 2 : 4);
  return rows > 0x7fffffffLL ? MFPA_EINVAL : (int)rows;
}

int mfpa_conv3x3_c1_bn_relu(const float* x32, const double* spec64, const double* denom, int per_clip, int B, int H,
                            int W, const float* w, int Cout, const float* scale, const float* shift, int relu, float* y,
                            int y_is_bf16, float* stats_part, void* stream) {
  if (B == 0) return MFPA_OK;
  if ((!x32 && !spec64) || !w || !y || B < 0 || H < 1 || W < 1) return MFPA_EINVAL;
  if (Cout % 4 || Cout < 4 || Cout > 1024 || (256 % (Cout / 4)) != 0) return MFPA_EINVAL;
  if ((long long)B * H > 0x7fffffffLL) return MFPA_EINVAL;
  const long long blocks = (long long)B * H;
  hipLaunchKernelGGL(conv3x3_c1_kernel, dim3((unsigned)blocks), dim3(256), 0, mfpa_stream(stream), x32, spec64, denom,
                     per_clip, B, H, W, w, Cout, scale, shift, relu, y, y_is_bf16, stats_part);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_maxpool2(const float* x, int B, int H, int W, int C, float* y, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!x || !y || B < 0 || H < 2 || W < 2 || C < 4 || C % 4) return MFPA_EINVAL;
  const long long total = (long long)B * (H / 2) * (W / 2) * (C / 4);
  long long blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(maxpool2_kernel, dim3((unsigned)blocks), dim3(256), 0, mfpa_stream(stream), x, B, H, W, C, y);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_conv1x1_out(const float* x, long long npix, int C, const float* w, float bias, float* y, void* stream) {
  if (!x || !w || !y || npix < 0 || C < 4 || C > 256 || (C & (C - 1)) != 0) return MFPA_EINVAL;
  if (npix == 0) return MFPA_OK;
  const int ppb = 256 / (C / 4);
  long long blocks = (npix + ppb - 1) / ppb;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(conv1x1_out_kernel, dim3((unsigned)blocks), dim3(256), 0, mfpa_stream(stream), x, npix, C, w, bias, y);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

}  // extern "C"
