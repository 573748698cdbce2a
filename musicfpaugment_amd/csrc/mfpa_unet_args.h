// Kernel-side argument block of the UNet convolution kernels (csrc/unet.hip, csrc/unet_ws.hip); filled from mfpa_conv_desc (include/mfpa.h)
// by mfpa_conv_mfma and the other entry points of unet.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace mfpa_unet {

struct ConvArgs {
  const float* x0;         // source 0: (B,H,W,C0) [mode 2: (B,2H,2W,C0)]
  const float* in_scale0;  // optional per-channel affine + ReLU applied to source 0 ON LOAD (training: the
  const float* in_shift0;  //   previous layer's BatchNorm+ReLU is never materialised); null = plain
  const float* x1;         // source 1: (B,H1,W1,C1) or null, zero-padded to (H,W) at offset (oy1,ox1)
  const float* w;          // [taps][Cout][Cin], Cin contiguous
  const float* scale;      // epilogue per-output-channel affine (null = identity)
  const float* shift;
  float* y;
  int C0, C1, H1, W1, oy1, ox1;
  int B, H, W, Cout, relu;
  int yH, yW;              // output extent (crop): pixels with gy >= yH or gx >= yW are not stored
  int tiles_x, tiles_y;
  unsigned drop_seed, drop_thresh;   // dropout on source 0 after the affine+ReLU (thresh 0 = off)
  float drop_scale;
  float* y_pool;                     // optional fused MaxPool2d(2) of the (affine+ReLU) output: (B,H/2,W/2,Cout)
  const float* w1x1;                 // optional fused OutConv 1x1 to one class (needs the whole Cout in one workgroup):
  float b1x1;                        //   y1x1[pixel] = sum_c out[pixel][c] * w1x1[c] + b1x1
  float* y1x1;
  int w_frag;                        // 1: `w` is the fragment-ordered bf16x3 image of the BDIR kernels (mfpa_conv_desc.w_layout)
  int in16;                          // conv_wd16_kernel (plain) / convT_mfma_kernel: the sources are bfloat16 tensors (mfpa_conv_desc.x0_is_bf16)
  int x0_split, x1_split, y_split, y_pool_split;   // conv_ws64_kernel: tensors in the split layout ([32 bf16 hi | 32 bf16 lo] per 32-channel chunk; mfpa_conv_desc)
  int plain;                         // conv_wd16_kernel: plain bf16 products (hi halves only: mfpa_conv_desc.precision 2, the training step)
  __bf16* x0_bf16;                   // conv_wd16_kernel: optional bf16 copy of the activated source 0, (B,H,W,C0) (mfpa_conv_desc.x0_bf16)
  __bf16* x1_bf16;                   // ... of source 1, (B,H1,W1,C1)
  __bf16* y_bf16;                    // ... of the stored output, (B,yH,yW,Cout)
  const float* bz;                   // conv_wd16_kernel + stats_part: the output is a gradient dy w.r.t. relu(bn(bz)), bz (B,yH,yW,Cout): the partials are
  int bz16;                          //   (bz is a bfloat16 tensor: mfpa_conv_desc.bwd_z_is_bf16)
  const float* bz_scale;             //   (sum g, sum g * xhat), g = dy where bz * bz_scale + bz_shift > 0 else 0, xhat = (bz - bz_mean) * bz_invstd --
  const float* bz_shift;             //   the two reductions of the BatchNorm backward (mfpa_conv_desc.bwd_z ...)
  const float* bz_mean;
  const float* bz_invstd;
  float* stats_part;                 // conv_wd16_kernel: optional per-wave partial (sum, sum of squares) of the stored output per channel:
                                     //   [tile * WMW + wm][2][Cout] (mfpa_conv_desc.stats_part; rows = mfpa_conv_stats_rows())
  int dbg_stagger;                   // -DMFPA_EXPERIMENTS builds only: start delay of persistent workgroup k = (k & 7) x this x 4096 cycles
  int dbg_lds_stamps;                // -DMFPA_EXPERIMENTS builds only: LDS byte offset of the tap-timeline stamps (0 = none)
  int dbg;                           // -DMFPA_EXPERIMENTS builds only (MFPA_CONV_DBG): 1 skip B staging, 2 skip barriers, 4 skip stores, 8 skip MFMA, 16 skip halo staging
  // C1SRC: source 0 is not read but COMPUTED while it is staged -- the UNet's first layer (1 -> 64 channels, folded BN,
  // ReLU) applied to the normalised spectrogram, so its 64-channel output never exists in HBM
  const float* c1_x32;               // (B,H,W) float32, or
  const double* c1_spec64;           // (B,H,W) float64 divided by c1_denom[b] (the fused spectrogram normalisation)
  const double* c1_denom;
  const float* c1_w;                 // (9, 64)
  const float* c1_scale;             // (64) folded BatchNorm of the first layer
  const float* c1_shift;
};

// csrc/unet_up.hip: a decoder level's first convolution with the transposed convolution folded in (mfpa_upconv_desc, include/mfpa.h)
struct UpArgs {
  const float* skip;       // (B,H,W,Cs)
  const float* low;        // (B,Hl,Wl,Cl): the transposed convolution's input
  const float* w_skip;     // fragment image (w_layout 2) of [9][Cout][Cs], output scale folded in
  const float* w_up;       // fragment image of the composite weights [16][Cout][Cl]
  const float* shift;      // (Cout)
  const float* bias_tab;   // (4, 4, Cout)
  float* y;                // (B,H,W,Cout)
  int B, H, W, Cs, Hl, Wl, Cl, Cout, relu;
  int precision;           // 1: bf16x3 fragment images; 0: fp32 fragment images (exact fp32 products)
  int tiles_x, tiles_y;
};
__attribute__((visibility("hidden"))) int launch_conv_up(UpArgs& a, hipStream_t s);
__attribute__((visibility("hidden"))) bool conv_up_serves(int H, int W, int Hl, int Wl, int Cs, int Cl, int Cout);

// csrc/unet_ws.hip: the wave-specialised 64-channel 3x3 convolution (inference, bf16x3).  Returns MFPA_OK / a negative code.
__attribute__((visibility("hidden"))) int launch_conv_ws64(ConvArgs& a, hipStream_t s);
// does conv_ws64_kernel serve this ConvArgs (checked by the dispatcher before it routes a launch there)?
__attribute__((visibility("hidden"))) bool conv_ws64_serves(const ConvArgs& a);

}  // namespace mfpa_unet
