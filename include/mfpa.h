/*
 * mfpa.h -- C ABI of libmfpa.so, the MI355X (gfx950) implementation of the
 * musicFPaugment hot path:
 *
 *   waveform -> STFT magnitude -> [UNet denoiser] -> spectral-peak picking -> peak-mask metrics
 *
 * The reference (deezer/musicFPaugment) is pure Python and has no FFI of its own;
 * each entry point below names the reference function (file:line under the
 * reference tree) whose arithmetic it replaces.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add at those call sites.
 *
 * Conventions
 *  - every function returns 0 on success, MFPA_EINVAL for a shape/argument error,
 *    or MFPA_EHIP - hipError_t for a HIP runtime failure; nothing aborts or throws;
 *  - all pointers are DEVICE pointers owned by the caller (e.g. torch allocations),
 *    all arrays dense row-major; the library allocates nothing;
 *  - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default
 *    stream) with no host synchronisation, so calls can be captured in a hipGraph;
 *  - functions are re-entrant; one process drives one GPU (multi-GPU = one process
 *    per GPU, sharding clips across ranks).
 */
#ifndef MFPA_H
#define MFPA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFPA_OK 0
#define MFPA_EINVAL (-22)
#define MFPA_EHIP (-1000) /* result is MFPA_EHIP - hipError_t */

#define MFPA_F32 0
#define MFPA_F64 1

#define MFPA_N_FFT 512
#define MFPA_N_HOP 256
#define MFPA_N_BINS 257

/* ABI version: bumps when a signature changes. */
int mfpa_version(void);

/* ---------------------------------------------------------------------------------------
 * STFT tables (HOST function, no GPU work): fills out[MFPA_STFT_TABLE_LEN] doubles on the host
 * from a 512-point window -- [0,512) the window, then the FFT twiddles.  The caller uploads the
 * array once and passes the device pointer as `tables` below (the window is an argument of the
 * reference's stft too, afp/audfprint/stft.py:15-19; the library itself keeps no state).
 */
#define MFPA_STFT_TABLE_LEN 1288
int mfpa_stft_tables(const double* window512, double* out);

/* STFT magnitude.  Replaces torch.stft + abs in training/visualisation.py:20-28 and
 * np.abs(stft.stft(...)) in afp/audfprint/peak_extractor.py:259-261 (afp/audfprint/stft.py:15-62).
 *   wav      (B, T_w) float32, T_w > 256
 *   tables   device copy of mfpa_stft_tables(np.hanning(514)[1:-1]) for both reference call sites
 *   mag      (B, 257, n_frames) float32 or float64 (out_dtype), n_frames = 1 + T_w / 256;
 *            centre / reflect padding, one-sided, unnormalised, computed in float64
 *   clip_max (B) float64, may be NULL: max float64 magnitude of each clip
 */
int mfpa_stft_mag(const float* wav, int B, int T_w, const double* tables,
                  void* mag, int out_dtype, double* clip_max, void* stream);

/* Number of STFT frames for T_w samples (1 + T_w / 256). */
int mfpa_stft_frames(int T_w);

/* mlab.specgram power spectrogram, afp/dejavu/fingerprint.py:60-66: no padding, frames at
 * multiples of 256 while a full frame fits ((T_w - 256) / 256 frames), one-sided PSD with bins
 * 1..255 doubled; samples are multiplied by scale_in first (dejavu.py:109 feeds x * 32767).
 * The common factor 1/(Fs*sum(w^2)) is NOT applied: the reference divides by the maximum
 * immediately (fingerprint.py:68).
 *   tables   device copy of mfpa_stft_tables(np.hanning(512))
 *   psd (B, 257, n_frames) float64, clip_max (B) float64 may be NULL. */
int mfpa_specgram_psd(const float* wav, int B, int T_w, double scale_in, const double* tables,
                      double* psd, double* clip_max, void* stream);
int mfpa_specgram_frames(int T_w);

/* In-place division by a maximum.  per_clip = 0: one max over the whole batch
 * (torch.max(specgram), training/visualisation.py:29); per_clip = 1: each clip by its own
 * (sgram /= np.max(sgram), peak_extractor.py:263; arr2D /= arr2D.max(), fingerprint.py:68).
 *   data (B, n) dtype, clip_max (B) float64 as produced by mfpa_stft_mag. */
int mfpa_normalize(void* data, int dtype, int B, long long n, const double* clip_max,
                   int per_clip, void* stream);

/* data (B, n) float64 -> out (B, n) float32 (the `.float()` of training/train.py:272). */
/* out[b][i] = (float)(in[b][i] / denom[b]), in (B, n) float64: the reference's `spectrogram / max` + `.float()` (training/train.py:264-272)
 * as one pass -- the same float64 quotient rounded to float32 that mfpa_conv3x3_c1_bn_relu / mfpa_wgrad_c1 form per tap from (spec64, denom). */
int mfpa_normalize_f32(const double* in, int B, long long n, const double* denom, float* out, void* stream);
int mfpa_f64_to_f32(const double* in, float* out, long long n, void* stream);

/* ---------------------------------------------------------------------------------------
 * Audfprint peak picker = everything of Audfprint_peaks.find_peaks after the optional UNet,
 * afp/audfprint/peak_extractor.py:271-311.
 *
 * Stage 1, mfpa_audfprint_prepare: log(max(s, max/1e6)) - mean, per-bin 1-pole high-pass
 * (scipy lfilter([1,-1],[1,-0.98])), Nyquist bin dropped  (peak_extractor.py:272-290).
 *   spec       (B, F, T) float32 or float64 (dtype), F = bins + 1 (257)
 *   denom      (B) float64 or NULL: if given, s = spec / denom[b] first (fuses the
 *              per-clip normalisation of peak_extractor.py:263)
 *   mean_order 0: numpy sums the spectrogram bin-major (C order: the UNet output),
 *              1: frame-major (the un-denoised |stft| array, which is a transposed view)
 *              -- the pairwise summation tree of np.mean is reproduced in that order
 *   log_input  bit 0: `spec` already holds log(max(s, max/1e6)) computed by the caller
 *              (strict mode: everything downstream is IEEE add/mul/compare -> bit-exact);
 *              bit 1 (value 2): denom[b] is the maximum of the float64 spec[b] itself, as
 *              mfpa_stft_mag returned it with this spectrogram: max(spec / denom) is then 1
 *              exactly (NaN for an all-zero clip, like numpy) and the max pass is skipped
 *   filtered   (B, T, F-1) float64, FRAME-major workspace consumed by mfpa_audfprint_prune
 *   scratch    (B, F*T) float64 workspace
 */
int mfpa_audfprint_prepare(const void* spec, int dtype, int B, int F, int T, const double* denom,
                           int mean_order, int log_input, double pole,
                           double* filtered, double* scratch, void* stream);

/* Stage 2, mfpa_audfprint_prune: forward decaying-threshold pass then backward pruning
 * (_decaying_threshold_fwd_prune / _bwd_prune_peaks, peak_extractor.py:173-234).
 *   filtered (B, T, R) float64 frame-major, R = 256 bins (R % 4 == 0, R <= 256)
 *   gauss    (2R+1) float64: exp(-0.5*((j-R)/f_sd)^2), the table of peak_extractor.py:163-165
 *   a_dec    threshold decay per frame (peak_extractor.py:295)
 *   maxpks   peaks kept per frame (<= 8; reference 5)
 *   mask     (B, R, T) uint8 in {0,1} -- peaks_mask of find_peaks, bin-major like the reference
 *   npeaks   (B) int32
 */
int mfpa_audfprint_prune(const double* filtered, int B, int R, int T, const double* gauss,
                         double a_dec, int maxpks, uint8_t* mask, int32_t* npeaks, void* stream);

/* Stages 1 + 2 in one call for the un-denoised path, Audfprint_peaks.find_peaks(d) without a denoiser
 * (afp/audfprint/peak_extractor.py:253-311): raw |STFT| + its per-clip maxima as mfpa_stft_mag returned them -> peak mask.
 * Same arithmetic as mfpa_audfprint_prepare(denom = clip_max, mean_order = 1, log_input = 2) followed by
 * mfpa_audfprint_prune, as two launches: the log values (frame-major) + the pairwise-tree node sums of np.mean, then a pruner
 * that applies "- mean, lfilter" to the frames as it walks them -- the filtered spectrogram is never written.
 *   spec (B, F, T) float64, 141 <= F = R + 1 <= 257, R % 4 == 0, T <= 512, R * T % 16 == 0 (the pruner zeroes the 16-byte aligned mask itself);  clip_max (B) float64
 *   work (B * F * T + B * 128) float64 workspace;  gauss / a_dec / maxpks / mask / npeaks as for mfpa_audfprint_prune */
/* out[i] = v[i] / den[i] by the sequence the pick kernels use per spectrogram cell (one reciprocal per clip there): q0 = v * RN(1 / den), two
 * residual / correction multiply-adds -- correctly rounded for normal operands with normal residuals (Markstein), i.e. bit-identical to the IEEE
 * division; exported so that the tests can compare it with the division on arbitrary pairs. */
int mfpa_div_by_reciprocal(const double* v, const double* den, long long n, double* out, void* stream);
int mfpa_audfprint_pick(const double* spec, const double* clip_max, int B, int F, int T, double pole,
                        const double* gauss, double a_dec, int maxpks, double* work, uint8_t* mask,
                        int32_t* npeaks, void* stream);

/* ---------------------------------------------------------------------------------------
 * Dejavu picker.  mfpa_dejavu_prepare: arr = scale*ln(max(a, max/1e6)) - mean with
 * a = psd / denom (fingerprint.py:68,78-79; scale = 10; mean_order as for mfpa_audfprint_prepare:
 * mlab.specgram hands back a frame-major array, so the reference sums with mean_order = 1).  mfpa_localmax2d: get_2D_peaks,
 * afp/dejavu/fingerprint.py:94-171: (2r+1)^2 maximum filter with scipy 'reflect' borders,
 * equality test, XOR with the eroded exact-zero background (border_value 1), amp > amp_min.
 *   arr  (B, F, T) float64;  mask (B, F, T) uint8;  npeaks (B) int32;  2 <= radius <= 16 (the reference uses 10)
 */
int mfpa_dejavu_prepare(const double* psd, int B, int F, int T, const double* denom, double scale,
                        int mean_order, double* arr, void* stream);
/* The denoised branch (fingerprint.py:70-79): x is the UNet's float32 output (B, F, T) for the normalised spectrogram,
 * a = x*x when square != 0, and max / floor / log / mean / subtraction are float32 operations as numpy performs them on a
 * float32 array (the network output is C-contiguous: mean_order = 0); arr is that float32 result widened to float64. */
int mfpa_dejavu_prepare_f32(const float* x, int B, int F, int T, int square, double scale, int mean_order, double* arr,
                            void* stream);
int mfpa_localmax2d(const double* arr, int B, int F, int T, int radius, double amp_min,
                    uint8_t* mask, int32_t* npeaks, void* stream);

/* The un-denoised Dejavu chain after the spectrogram in one call -- fingerprint.py:68,78-79 + get_2D_peaks (:94-171): the PSD and
 * its per-clip maxima as mfpa_specgram_psd returned them -> peak mask.  Same arithmetic as mfpa_dejavu_prepare(denom = clip_max)
 * followed by mfpa_localmax2d, as two launches: scale * ln(max(a, 1e-6)) + the node sums of np.mean's pairwise tree (many
 * workgroups per clip), then the local-maximum kernel, which forms the mean from the node sums and subtracts it while it loads
 * its tiles -- the mean-subtracted array is never written.  clip_max[b] MUST be the maximum of psd[b] (max(a) = 1 is assumed).
 *   psd (B, F, T) float64, 141 <= F <= 257, T <= 512;  work: B * mfpa_dejavu_pick_work_doubles(F, T) float64;  mask / npeaks as for mfpa_localmax2d */
int mfpa_dejavu_pick_work_doubles(int F, int T, long long* per_clip);
int mfpa_dejavu_pick(const double* psd, const double* clip_max, int B, int F, int T, double scale, int mean_order, int radius,
                     double amp_min, double* work, uint8_t* mask, int32_t* npeaks, void* stream);

/* ---------------------------------------------------------------------------------------
 * Peak-mask metrics, testing/metrics.py:10-192.  For 0/1 masks (B, N1, N2), N1,N2 >= 2:
 *   counts (B, 4) int64 = [hits_precision, n_predicted, hits_recall, n_ground_truth] per clip,
 * where a peak (i, j) of one mask is looked up in the other at (i + [i==0], j + [j==0])
 * (the reference's low-border quirk).  Precision = sum hits_p / sum n_p, etc.
 */
int mfpa_peak_metrics(const uint8_t* predicted, const uint8_t* gt, int B, int N1, int N2,
                      int64_t* counts, void* stream);

/* Per-clip PSNR statistics (testing/metrics.py:7, torchmetrics PeakSignalNoiseRatio; used by
 * testing/audfprint_exps.py:136-137): pred (B,n) float32|float64, target (B,n) float64,
 * out (B,3) float64 = [sum (pred-target)^2, min(target), max(target)]. */
int mfpa_psnr_stats(const void* pred, int pred_dtype, const double* target, int B, long long n, double* out,
                    void* stream);

/* ---------------------------------------------------------------------------------------
 * Landmark pairing + hashing, the step after the pickers (next-tier row SURVEY.md §8f-1).  Integer only.
 *
 * mfpa_audfprint_landmarks: peaks2landmarks + landmarks2hashes + the duplicate removal of wavfile2hashes
 * (afp/audfprint/peak_extractor.py:313-346, :40-58, :443-460) from the peak mask of mfpa_audfprint_prune.
 *   mask (B,R,T) uint8 (<= 8 peaks per frame); cap = row capacity per clip (<= 8192)
 *   landmarks (B,cap,4) int32 (col, bin1, bin2, dcol) in the reference's list order
 *   hashes    (B,cap,2) int32 (time, hash) in the same order
 *   uniq      (B,cap,2) int32 unique rows sorted by (time << 32) + hash
 *   counts    (B,2) int32 = [n_landmarks, n_unique], or [-1,-1] when a clip overflows cap / 8 peaks per frame
 *   reference constants: mindt 2, targetdt 63, targetdf 31, maxpairs 3 (peak_extractor.py:99-107)
 *
 * mfpa_dejavu_hashes: generate_hashes (afp/dejavu/fingerprint.py:174-213) from the mask of mfpa_localmax2d:
 * peaks in (time, freq) order, each with its next fan-1 peaks, min_dt <= dt <= max_dt,
 * SHA-1("f1|f2|dt") truncated to 10 bytes (= 20 hex digits).
 *   digests (B,cap,10) uint8, t1 (B,cap) int32, counts (B) int32 (-1: more than peak_cap peaks or cap hashes)
 */
int mfpa_audfprint_landmarks(const uint8_t* mask, int B, int R, int T, int cap, int mindt, int targetdt,
                             int targetdf, int maxpairs, int32_t* landmarks, int32_t* hashes, int32_t* uniq,
                             int32_t* counts, void* stream);
int mfpa_dejavu_hashes(const uint8_t* mask, int B, int F, int T, int cap, int peak_cap, int fan, int min_dt,
                       int max_dt, uint8_t* digests, int32_t* t1, int32_t* counts, void* stream);

/* ---------------------------------------------------------------------------------------
 * UNet denoiser building blocks, training/unet.py:8-108.  Activations are NHWC float32
 * ("pixels x channels": H = frequency bins, W = frames); with C = 1 at both ends of the
 * network this is byte-identical to the reference's NCHW (B,1,257,T) tensors.
 */

/* 3x3 convolution, padding 1, no bias, fused per-channel affine (folded eval BatchNorm) + ReLU
 * (DoubleConv halves, unet.py:16-21).  The input may be the channel-concatenation of two
 * tensors [x0 | x1] where x1 has its own extent (H1, W1) <= (H, W) and is zero-padded at the
 * bottom/right (Up.forward, unet.py:56-63); pass x1 = NULL, C1 = 0 for a plain conv.
 *   x0 (B,H,W,C0)  x1 (B,H1,W1,C1)  w (9, Cout, C0+C1): [tap = ky*3+kx][Cout][Cin]  scale/shift (Cout)
 *   y  (B,H,W,Cout);  relu: 0/1;  scale/shift may be NULL (identity).
 * C0, C1 multiples of 32; Cout a multiple of 64.  precision: 0 = float32 MFMA (exact fp32 products). */
int mfpa_conv3x3_bn_relu(const float* x0, int C0, const float* x1, int C1, int H1, int W1,
                         int B, int H, int W, const float* w, int Cout,
                         const float* scale, const float* shift, int relu, int precision,
                         float* y, void* stream);

/* General form of the MFMA implicit-GEMM convolution (same kernel as the two entry points above
 * and below), used by the training step: forward with the previous layer's BatchNorm + ReLU applied
 * to source 0 ON LOAD, input gradients (a 3x3 conv of dz with transposed, flipped weights; mode 2 for
 * the transposed conv), cropped outputs.  training/unet.py:8-65, training/train.py:314-316 (backward).
 *   mode 0: 3x3 conv pad 1;  1: ConvTranspose2d(k2,s2) forward (y is (B,2H,2W,Cout));
 *        2: ConvTranspose2d input gradient (x0 is (B,2H,2W,C0), y is (B,H,W,Cout), w (4,Cout,C0))
 *   in_scale0/in_shift0 (C0) or NULL: x0 <- relu(x0*scale+shift) per channel when loaded
 *   x1 (B,H1,W1,C1) zero-padded to (H,W) like mfpa_conv3x3_bn_relu (mode 0 only)
 *   yH,yW: output extent (0 = H,W): pixels beyond are not stored, y is (B,yH,yW,Cout)  (modes 0, 2)
 */
typedef struct mfpa_conv_desc {
  const float* x0; const float* in_scale0; const float* in_shift0;
  const float* x1;
  const float* w; const float* out_scale; const float* out_shift;
  float* y;
  int C0, C1, H1, W1;
  int B, H, W, Cout, relu;
  int yH, yW, mode;
  unsigned drop_seed, drop_thresh; /* training Dropout on source 0 after the affine+ReLU: keep element idx iff */
  float drop_scale;                /* hash(seed, idx) >= thresh (= rate * 2^32), scaled by 1/(1-rate); 0 = off  */
  int precision;                   /* 0 = fp32 MFMA; 1 = bf16x3 (w must then be in the pre-split row format);
                                    * 2 = plain bf16 (mode 0: w_layout 2 only -- the same fragment-ordered image, of which only the hi halves are
                                    *     read; one MFMA per product, relative error ~2^-9: the training step's "bf16 MFMA" arithmetic.  Round 6:
                                    *     also mode 1 with bfloat16 I/O (x0_is_bf16 or y == NULL) and mode 2, on the ROW image of precision 1:
                                    *     the transposed convolution of the training step and its input gradient use the hi halves only) */
  float* y_pool;                   /* mode 0, optional: MaxPool2d(2) of the output fused in the epilogue, (B,H/2,W/2,Cout) */
  const float* w1x1; float b1x1;   /* mode 0, Cout == 64, optional: OutConv 1x1 to one class fused in the epilogue:        */
  float* y1x1;                     /*   y1x1 (B,H,W) = sum_c out[..][c]*w1x1[c] + b1x1; y may then be NULL (not stored)    */
  /* mode 0, C0 == Cout == 64, C1 == 0, optional: source 0 is COMPUTED while it is staged (x0 may be NULL) as the UNet's
   * first layer of mfpa_conv3x3_c1_bn_relu -- relu((conv3x3 of the 1-channel input) * c1_scale + c1_shift), input =
   * c1_x32 (B,H,W) or (float)(c1_spec64 / c1_denom[b]) -- so inc.double_conv's 64-channel intermediate never exists
   * in HBM (unet.py:8-24, 86). */
  const float* c1_x32; const double* c1_spec64; const double* c1_denom;
  const float* c1_w; const float* c1_scale; const float* c1_shift;
  /* precision 1, mode 0: which bf16x3 image `w` holds.  0 = the row image ([tap][Cin / 32][Cout][128 B], staged through LDS);
   * 1, 2 = FRAGMENT-ORDERED images of the weights-direct kernels (a wave reads the MFMA weight operand of its column tile straight
   * from L1 / L2: no weight tile in LDS, one barrier per 32-channel chunk instead of one per tap): 2 = [tap][Cin / 32][Cout / 16]
   * [hi | lo][lane 64][16 B] for v_mfma_f32_16x16x32_bf16 (conv_wd16_kernel), 1 = [tap][Cin / 32][Cout / 32][substep 2][hi | lo]
   * [lane 64][16 B] for v_mfma_f32_32x32x16_bf16 -- valid exactly where mfpa_conv_weight_layout() returns that number,
   * MFPA_EINVAL otherwise. */
  int w_layout;
  /* w_layout 2, mode 0, optional (training forward): a bf16 copy of source 0 AS THE CONVOLUTION SAW IT (in_scale0 / in_shift0 / ReLU /
   * dropout applied), (B,H,W,C0), written by the halo loader of the first output-channel tile -- the operand mfpa_wgrad_mfma(precision 3)
   * reads in the backward pass, without a cast pass of its own.  Also written by the bf16x3 transposed convolution (mode 1, precision 1).
   * MFPA_EINVAL with any other kernel. */
  void* x0_bf16;
  /* the same for the zero-padded source 1, (B,H1,W1,C1) (plain cast: source 1 carries no on-load transform), and for the OUTPUT,
   * (B,yH,yW,Cout) -- when the output is itself an operand of a later mfpa_wgrad_mfma(precision 3) (the transposed convolution's). */
  void* x1_bf16;
  void* y_bf16;
  /* w_layout 2, mode 0, optional (training forward): per-wave partial BatchNorm statistics of the stored output, [rows][2][Cout] floats
   * with rows = mfpa_conv_stats_rows(B, H, W, C0 + C1, Cout): (sum, sum of squares) per channel over each wave's pixels.
   * mfpa_conv_stats_reduce sums the rows in float64 (fixed order) into the `sums` of mfpa_bn_stats_finish -- the statistics without a
   * pass over the output.  MFPA_EINVAL with any other kernel. */
  float* stats_part;
  /* with stats_part, optional (training backward): the output is the gradient dy w.r.t. relu(bn(bwd_z)) -- bwd_z (B,yH,yW,Cout) the
   * BatchNorm's input, bwd_scale / bwd_shift / bwd_mean / bwd_invstd (Cout) its statistics, no dropout behind it -- and the partials are
   * (sum g, sum g * xhat) with g = dy where bwd_z * scale + shift > 0 else 0: after mfpa_conv_stats_reduce the `local_sums` of
   * mfpa_bn_relu_bwd_finish, without the reduction pass over (dy, z). */
  const float* bwd_z; const float* bwd_scale; const float* bwd_shift; const float* bwd_mean; const float* bwd_invstd;
  /* x0 (and x1, when given) are BFLOAT16 tensors.  precision 2, mode 0, (C0 + C1) % 64 == 0 (conv_wd16_kernel): the bf16 copy of dz
   * written by mfpa_bn_relu_bwd -- staged without any split arithmetic, half the bytes -- or, round 5, the training step's activations
   * kept as bfloat16 in HBM: an on-load affine / ReLU / dropout is applied in float32 to the widened values and rounded to bf16 once
   * (x0_bf16 still receives that copy; x1 is its own copy: x1_bf16 must be null).  mode 1, precision 1: the transposed convolution's
   * source.  Such a launch may also leave its OUTPUT as bfloat16 only: y null, y_bf16 the (B,yH,yW,Cout) [mode 1: (B,2H,2W,Cout)]
   * destination (stats_part still describes the float32 accumulators). */
  int x0_is_bf16;
  /* round 5, inference launches that conv_ws64_kernel serves only (mfpa_conv_scale_folds() == 1; otherwise MFPA_EINVAL): tensors in the
   * SPLIT layout -- same shape and byte count as the float32 NHWC tensor, but every 32-channel chunk of a pixel holds [32 bf16 hi | 32
   * bf16 lo] (x = hi + lo to 2^-17) instead of 32 floats: exactly what a bf16x3 convolution's loader makes of the float32 values, made
   * ONCE by the producer.  x0_split / x1_split: source 0 / 1 arrives in it (the loader waves copy 16-byte pieces, no arithmetic);
   * y_split / y_pool_split: `y` / `y_pool` leave in it.  A split tensor can only feed another such launch. */
  int x0_split, x1_split, y_split, y_pool_split;
  /* bwd_z is a BFLOAT16 tensor (the activations of the line above). */
  int bwd_z_is_bf16;
} mfpa_conv_desc;
int mfpa_conv_mfma(const mfpa_conv_desc* d, void* stream);
/* HOST function: the w_layout (0, 1 or 2) the fastest kernel for a (H, W) convolution of this shape reads. */
int mfpa_conv_weight_layout(int H, int W, int Cin, int Cout, int mode, int precision);
/* HOST function (round 5): 1 if the INFERENCE launch of this 3x3 shape (precision 1, w_layout 2, no on-load affine, no training side
 * output) runs on conv_ws64_kernel, whose epilogue is cheapest when the output scale is already IN the weights: pass out_scale = NULL
 * with weights pre-multiplied by the per-output-channel scale (then split / fragment-ordered as usual) and out_shift as before -- the shift
 * becomes the accumulators' start value and the epilogue a bare ReLU.  The same call with out_scale given stays valid everywhere.
 * Reference: the folded eval BatchNorm of DoubleConv, training/unet.py:16-21. */
int mfpa_conv_scale_folds(int H, int W, int Cin, int Cout);
/* HOST function (round 5): the w_layout the FUSED FIRST-LAYER launch of mfpa_conv_mfma (c1_x32 / c1_spec64 given, 64 -> 64 channels,
 * precision 1) reads at this image size: 2 = the fragment image (conv_ws64_kernel computes the first layer in its loader waves on the
 * matrix cores, bf16x3 like every other layer), 0 = the row image (conv_mfma_kernel computes it with exact fp32 FMAs in its loader).
 * Reference: DoubleConv of `inc`, training/unet.py:16-21, 86. */
int mfpa_conv_c1_layout(int H, int W);
/* Round 6: a decoder level's FIRST convolution with the transposed convolution folded into it -- Up.forward of training/unet.py:58-65,
 *   up = ConvTranspose2d(Cl, Cu, 2, 2)(low); pad to the skip's size; y = relu(bn(conv3x3(cat([skip, up]))))        (eval-mode BatchNorm)
 * as ONE launch that reads `skip` and `low` and never forms `up`: nothing non-linear sits between the two convolutions, so the up half is,
 * per output phase (Y & 1, X & 1), a 2 x 2 convolution of `low` with composite weights (exact algebra; float64 check:
 * tools/exp_phase_composite.py) plus a bias that depends only on the pixel's border class.  Products: bf16x3 or exact fp32 (`precision`).
 *   mfpa_upconv_pack (one-time, device): w3 [9][Cout][Cs + Cu] (tap = 3 ky + kx: mfpa_conv_mfma's fp32 kernel layout), wt [4][Cu][Cl]
 *     (mfpa_convT2x2's layout), bt (Cu), scale (Cout, the folded BatchNorm scale, or NULL) -> wc16 [16][Cout][Cl] float32, index
 *     ((py * 2 + px) * 2 + ty) * 2 + tx = phase (py, px), low-resolution tap (ty - 1 + py, tx - 1 + px), and bias_tab (4, 4, Cout) by
 *     (row class, column class): 0 first row / column of the up-sampled extent, 1 interior, 2 its last, 3 the padding row / column of an
 *     odd size; both with `scale` multiplied in, float64 accumulation.
 *   mfpa_upconv_fused: w_skip = the w_layout-2 fragment image of (w3[:, :, :Cs] * scale) and w_up = that of wc16 (as a 16-tap kernel);
 *     shift (Cout) the folded BatchNorm shift.  skip (B,H,W,Cs), low (B,Hl,Wl,Cl), y (B,H,W,Cout) float32 NHWC; H - 2 Hl and W - 2 Wl in {0, 1}.
 *   mfpa_upconv_serves (HOST): 1 if mfpa_upconv_fused takes this shape, else 0 (the caller then runs mfpa_convT2x2 + mfpa_conv_mfma). */
typedef struct mfpa_upconv_desc {
  const float* skip; const float* low;
  const float* w_skip; const float* w_up;
  const float* shift; const float* bias_tab;
  float* y;
  int B, H, W, Cs, Hl, Wl, Cl, Cout, relu;
  int precision;                   /* 1: bf16x3 (w_skip / w_up = w_layout-2 fragment images); 0: exact fp32 products (v_mfma_f32_16x16x4_f32) on the
                                    * FP32 fragment images [tap][Cin / 32][Cout / 16][piece 2][lane 64][4 floats], lane (g = l >> 4, c = l & 15) =
                                    * output channel 16 t + c, input channels 32 chunk + 8 g + 4 piece .. + 3 */
} mfpa_upconv_desc;
int mfpa_upconv_fused(const mfpa_upconv_desc* d, void* stream);
int mfpa_upconv_pack(const float* w3, const float* wt, const float* bt, const float* scale, int Cout, int Cs, int Cu, int Cl,
                     float* wc16, float* bias_tab, void* stream);
int mfpa_upconv_serves(int H, int W, int Hl, int Wl, int Cs, int Cl, int Cout);

/* HOST function: rows of mfpa_conv_desc.stats_part for this shape (0: its kernel does not write them). */
int mfpa_conv_stats_rows(int B, int H, int W, int Cin, int Cout);
/* stats_part (rows, 2, C) -> sums[2C] float64 as mfpa_bn_stats_sums produces them; workspace as for mfpa_bn_stats. */
int mfpa_conv_stats_reduce(const float* part, long long rows, int C, double* sums, double* workspace, void* stream);

/* First layer: 3x3 conv from ONE input channel (inc.double_conv.0, unet.py:86) fused with the
 * spectrogram normalisation: x = (float)(spec / denom) when spec64 != NULL, else x32 as is.
 *   w (9, Cout), y (B,H,W,Cout). */
int mfpa_conv3x3_c1_bn_relu(const float* x32, const double* spec64, const double* denom, int per_clip,
                            int B, int H, int W, const float* w, int Cout,
                            const float* scale, const float* shift, int relu, float* y, int y_is_bf16, float* stats_part /* optional: (B * H, 2, Cout) row partials of the output's (sum, sum of squares), see mfpa_conv_desc.stats_part */, void* stream);

/* MaxPool2d(2), floor (unet.py:34).  x (B,H,W,C) -> y (B,H/2,W/2,C). */
int mfpa_maxpool2(const float* x, int B, int H, int W, int C, float* y, void* stream);

/* ConvTranspose2d(k=2, s=2) + bias (unet.py:51-53).  x (B,H,W,Cin), w (4, Cout, Cin): [tap = dy*2+dx][Cout][Cin],
 * y (B,2H,2W,Cout). */
int mfpa_convT2x2(const float* x, int B, int H, int W, int Cin, const float* w, const float* bias,
                  int Cout, int precision, float* y, void* stream);

/* OutConv 1x1 + bias to ONE class (unet.py:68-74).  x (B*H*W, C), w (C), y (B*H*W). */
int mfpa_conv1x1_out(const float* x, long long npix, int C, const float* w, float bias, float* y,
                     void* stream);

/* ---------------------------------------------------------------------------------------
 * UNet training step, training/train.py:257-317 (spec branch): forward in train mode, L1 loss,
 * backward, Adam.  A layer's BatchNorm+ReLU output is never materialised: convolutions write the raw
 * output z, mfpa_bn_stats reduces the batch statistics into per-channel (scale, shift), and consumers
 * apply relu(z*scale+shift) on load (mfpa_conv_mfma's in_scale0/in_shift0).  Reductions are two-stage
 * float64 and deterministic.  `workspace`: at least mfpa_red_blocks() * max(2*C, 65) doubles.
 */
int mfpa_red_blocks(void);

/* nn.BatchNorm2d in train mode (unet.py:17,20): z (npix, C) -> mean, invstd (biased var, eps),
 * scale = gamma*invstd, shift = beta - mean*scale; running_mean/var (momentum, unbiased var) updated
 * in place when non-NULL. */
int mfpa_bn_stats(const float* z, long long npix, int C, const float* gamma, const float* beta, float eps,
                  float momentum, float* mean, float* invstd, float* scale, float* shift,
                  float* running_mean, float* running_var, double* workspace, int z_is_bf16, void* stream);

/* BatchNorm+ReLU backward: dy (gradient w.r.t. relu(bn(z))) is overwritten with the gradient w.r.t. z;
 * dgamma, dbeta (C) are produced; coef is a (3, C) scratch.  dz_bf16 (optional, may be NULL): a bf16 copy of the result, written in
 * the same pass -- the operand mfpa_wgrad_mfma(precision 3) reads. */
int mfpa_bn_relu_bwd(float* dy, const float* z, long long npix, int C, const float* gamma,
                     const float* scale, const float* shift, const float* mean, const float* invstd,
                     float* dgamma, float* dbeta, float* coef, double* workspace,
                     unsigned drop_seed, unsigned drop_thresh, float drop_scale, void* dz_bf16, int write_f32,
                     int z_is_bf16, void* stream);
/* write_f32 = 0 (needs dz_bf16): only the bf16 copy of dz is written -- every consumer reads it (the plain-bf16 train step: the
 * input-gradient convolution takes it through mfpa_conv_desc.x0_is_bf16, the weight gradient as its bf16 operand) and dy is left as it was. */

/* The same two operations split for synchronised BatchNorm under data parallelism (statistics over the GLOBAL batch, as the
 * single-GPU reference computes them): *_sums reduces this rank's per-channel pairs into sums[2C] float64 -- (sum z, sum z^2)
 * forward, (sum g, sum g*xhat) backward, g = the ReLU/dropout-masked incoming gradient -- the caller all-reduces (SUM) them
 * and the pixel count over the ranks, *_finish completes from the global sums.  Backward: dgamma / dbeta come from the LOCAL
 * sums (the gradient all-reduce adds the ranks), the mean terms of the input gradient from the GLOBAL ones. */
int mfpa_bn_stats_sums(const float* z, long long npix, int C, double* sums, double* workspace, int z_is_bf16, void* stream);
/* round 5, single-GPU statistics (no all-reduce between the halves): mfpa_conv_stats_reduce + mfpa_bn_stats_finish as two launches
 * instead of three, and mfpa_conv_stats_reduce + mfpa_bn_relu_bwd_finish as three instead of four -- the finish kernels sum the row
 * blocks' float64 partials themselves, in mfpa_conv_stats_reduce's order: bit-identical results. */
int mfpa_conv_stats_bn_finish(const float* part, long long rows, int C, double count, const float* gamma, const float* beta, float eps,
                              float momentum, float* mean, float* invstd, float* scale, float* shift, float* running_mean,
                              float* running_var, double* workspace, void* stream);
int mfpa_bn_relu_bwd_from_part(float* dy, const float* z, long long npix, int C, const float* gamma, const float* scale,
                               const float* shift, const float* mean, const float* invstd, const float* part, long long rows,
                               float* dgamma, float* dbeta, float* coef, double* workspace, unsigned drop_seed, unsigned drop_thresh,
                               float drop_scale, void* dz_bf16, int write_f32, int z_is_bf16, int dy_is_bf16, void* stream);
int mfpa_bn_stats_finish(const double* sums, double count, int C, const float* gamma, const float* beta, float eps,
                         float momentum, float* mean, float* invstd, float* scale, float* shift, float* running_mean,
                         float* running_var, void* stream);
int mfpa_bn_relu_bwd_sums(const float* dy, const float* z, long long npix, int C, const float* scale, const float* shift,
                          const float* mean, const float* invstd, double* sums, double* workspace, unsigned drop_seed,
                          unsigned drop_thresh, float drop_scale, int z_is_bf16, void* stream);
int mfpa_bn_relu_bwd_finish(float* dy, const float* z, long long npix, int C, const float* gamma, const float* scale,
                            const float* shift, const float* mean, const float* invstd, const double* local_sums,
                            const double* global_sums, double global_count, float* dgamma, float* dbeta, float* coef,
                            unsigned drop_seed, unsigned drop_thresh, float drop_scale, void* dz_bf16, int write_f32,
                            int z_is_bf16, int dy_is_bf16, void* stream);

/* out[c] = sum over pixels of x[p][c]  (ConvTranspose2d bias gradient). */
int mfpa_colsum(const float* x, long long npix, int C, float* out, double* workspace, void* stream);

/* p = MaxPool2d(2)(relu(z*scale+shift)) (unet.py:34) and its backward: dy += route(dp) to the window's
 * first maximum.  drop_*: the nn.Dropout (unet.py:83,99-103) that follows this DoubleConv in train mode, applied
 * to relu(bn(z)) with the stateless mask of mfpa_conv_desc (thresh 0 = no dropout); mfpa_bn_relu_bwd applies the
 * same mask to the incoming gradient. */
int mfpa_bn_relu_pool(const float* z, int B, int H, int W, int C, const float* scale, const float* shift,
                      float* p, unsigned drop_seed, unsigned drop_thresh, float drop_scale, int z_is_bf16, int p_is_bf16, void* stream);
int mfpa_maxpool2_bwd_add(const float* z, int B, int H, int W, int C, const float* scale,
                          const float* shift, const float* dp, float* dy,
                          unsigned drop_seed, unsigned drop_thresh, float drop_scale, int z_is_bf16, void* stream);

/* The same pass, which also forms the partial sums of the BatchNorm backward that follows on this layer (training/unet.py:17-20 under
 * autograd: sum g and sum g * xhat per channel, g = dy * [z*scale+shift > 0] (* dropout), xhat = (z - mean) * invstd) over ALL pixels of
 * dy -- the pooled windows, the odd last row / column --, one row of `part` (B * (H / 2), 2, C) float per workgroup, to be finished by
 * mfpa_conv_stats_reduce + mfpa_bn_relu_bwd_finish: the separate reduction pass over dy and z is not needed.  C / 4 must divide 256. */
int mfpa_maxpool2_bwd_add_sums(const float* z, int B, int H, int W, int C, const float* scale, const float* shift, const float* mean,
                               const float* invstd, const float* dp, float* dy, unsigned drop_seed, unsigned drop_thresh,
                               float drop_scale, float* part, int z_is_bf16, void* stream);
/* round 5: an encoder block's last BatchNorm + ReLU backward without its finished input gradient in memory -- dy (the skip path's gradient,
 * (B,H,W,C) float32, or bfloat16 with dy_is_bf16) is only READ: g = dy + route(dp) is reduced to the BatchNorm-backward sums (`part`:
 * (B * (H / 2), 2, C) floats of scratch), dgamma / dbeta / coef ([3][C]) are finished, and g -- formed again -- goes through the backward
 * formula into the bfloat16 dz (B,H,W,C).  Replaces mfpa_maxpool2_bwd_add_sums + mfpa_bn_relu_bwd_from_part (same arithmetic per element;
 * single-GPU statistics; C a power of two <= 1024 with 256 % (C / 4) == 0). */
int mfpa_maxpool2_bwd_bn_relu_bwd(const float* z, int B, int H, int W, int C, const float* gamma, const float* scale, const float* shift,
                                  const float* mean, const float* invstd, const float* dp, const void* dy, int dy_is_bf16, unsigned drop_seed,
                                  unsigned drop_thresh, float drop_scale, float* part, float* dgamma, float* dbeta, float* coef,
                                  double* workspace, void* dz_bf16, int z_is_bf16, void* stream);

/* Weight gradient on MFMA, ACCUMULATED into dw (zero it first):
 *   mode 0: dw[tap][co][ci] += sum_p dz[p][co] * xin[p + tap][ci]          (3x3 conv; dw (9,Cout,C0+C1))
 *   mode 1: dw[tap][co][ci] += sum_p dz[2y+dy,2x+dx][co] * xin[y,x][ci]    (transposed conv; dw (4,Cout,C0))
 * xin = [x0 (affine+ReLU on load if in_scale0) | x1 zero-padded] exactly as the forward saw it.
 * C0, C1, Cout multiples of 64.  precision 0: fp32 MFMA; 1: bf16x3 (both operands split hi + lo, 3 bf16 MFMAs per
 * product, fp32 accumulate) -- like mfpa_conv_desc.precision; 2: plain bf16 products (one MFMA; the sum over every pixel of
 * the batch averages the 2^-9 product rounding: relative L1 ~2e-3 per layer, and only the optimiser consumes the result). */
typedef struct mfpa_wgrad_desc {
  const float* dz; const float* x0; const float* in_scale0; const float* in_shift0; const float* x1;
  float* dw;
  int C0, C1, H1, W1;
  int B, H, W, Cout, mode;
  unsigned drop_seed, drop_thresh;
  float drop_scale;
  int precision;
} mfpa_wgrad_desc;
int mfpa_wgrad_mfma(const mfpa_wgrad_desc* d, void* stream);

/* precision 3 of mfpa_wgrad_mfma: dz / x0 / x1 are bfloat16 tensors of the same shapes holding the ACTIVATED operands (in_scale0 /
 * in_shift0 / dropout must be off), one bf16 MFMA per product like precision 2.  mfpa_act_to_bf16 makes such a copy: out[e] =
 * bf16(z[e]) (scale == NULL), or bf16(dropout_e(relu(z[e] * scale[c] + shift[c]))) -- the on-load transform the fp32 kernels apply
 * (the previous layer's BatchNorm + ReLU + Dropout, training/unet.py:8-25,99-103).  z (n) float32, n a multiple of C, C % 4 == 0. */
int mfpa_act_to_bf16(const float* z, long long n, int C, const float* scale, const float* shift, unsigned drop_seed,
                     unsigned drop_thresh, float drop_scale, void* out_bf16, void* stream);

/* First layer (1 input channel): dw[tap][co] += sum_p dz[p][co] * x[p + tap]; x as in
 * mfpa_conv3x3_c1_bn_relu (per-clip denominators). */
int mfpa_wgrad_c1(const float* dz, const float* x32, const double* spec64, const double* denom, int B, int H,
                  int W, int Cout, float* dw, int dz_is_bf16, void* stream);

/* OutConv in training: pred[p] = sum_c relu(z[p][c]*scale[c]+shift[c]) * w[c] + bias[0];
 * backward: dy[p][c] = dpred[p]*w[c], dwb = [C weight gradients, bias gradient]. */
int mfpa_outconv_fwd(const float* z, long long npix, int C, const float* scale, const float* shift,
                     const float* w, const float* bias, float* pred, int z_is_bf16, void* stream);
int mfpa_outconv_bwd(const float* z, const float* dpred, long long npix, int C, const float* scale,
                     const float* shift, const float* w, float* dy, float* dwb, double* workspace, int z_is_bf16, void* stream);
/* The same backward WITHOUT materialising dy (it is rank 1: dpred[p] * w[c]): the pass forms dwb and, holding z and dy of every element
 * anyway, the partial sums of the BatchNorm backward that follows on z's layer ({sum g, sum g * xhat}, one row of `part`
 * (mfpa_outconv_bwd_rows(npix, C), 2, C) float per workgroup, finished by mfpa_conv_stats_reduce); mfpa_bn_relu_bwd_finish_rank1 then
 * forms dy again from dpred and w while it applies the BatchNorm + ReLU backward: dz to dz_f32 (float32, may be null) and / or
 * dz_bf16.  No dropout on this layer (training/unet.py: the last DoubleConv feeds OutConv directly). */
int mfpa_outconv_bwd_rows(long long npix, int C, int* rows);
int mfpa_outconv_bwd_sums(const float* z, const float* dpred, long long npix, int C, const float* scale, const float* shift,
                          const float* mean, const float* invstd, const float* w, float* dwb, double* workspace, float* part,
                          int z_is_bf16, void* stream);
int mfpa_bn_relu_bwd_finish_rank1(const float* dpred, const float* w1, const float* z, long long npix, int C, const float* gamma,
                                  const float* scale, const float* shift, const float* mean, const float* invstd,
                                  const double* local_sums, const double* global_sums, double global_count, float* dgamma,
                                  float* dbeta, float* coef, float* dz_f32, void* dz_bf16, int z_is_bf16, void* stream);

/* nn.L1Loss(mean) of float32 pred against the float64 target (train.py:280): loss[0] (float64) and,
 * if dpred != NULL, dpred = sign(pred - target) / n. */
int mfpa_l1_loss(const float* pred, const double* target, long long n, float* dpred, double* loss,
                 double* workspace, void* stream);

/* Operand image of a convolution's weights, rebuilt from the master fp32 parameters after every optimiser step (replaces the
 * host-side flip / transpose / bf16 split of the packing code).  w: [taps][Co][Ci] fp32.  out, precision 0: [taps][nrows][K] float rows;
 * precision 1: the bf16x3 image [taps][K / 32][nrows][128 B], a row = 32 bf16 hi | 32 bf16 lo in eight 16-byte slots stored at slot
 * index (logical ^ ((row >> 1) & 7)) -- a (tap, chunk, 128-row) tile is 16 KB contiguous (csrc/unet.hip); precision 2: the same split in
 * the FRAGMENT-ORDERED layout of mfpa_conv_desc.w_layout = 1 ([taps][K / 32][nrows / 32][substep][hi | lo][lane][8 bf16]); precision 3:
 * that of w_layout = 2 ([taps][K / 32][nrows / 16][hi | lo][lane][8 bf16]).  flip_transpose = 0: rows = output
 * channels row0 .. row0+nrows-1, K = Ci (forward operand).  flip_transpose = 1: rows = input channels row0 .. row0+nrows-1,
 * K = Co, and for taps == 9 the kernel is flipped (tap t <- 8 - t): the input-gradient operand (training/unet.py's Conv2d /
 * ConvTranspose2d backward).  Co, Ci, row0, nrows multiples of 32. */
int mfpa_pack_conv_weights(const float* w, int taps, int Co, int Ci, int flip_transpose, int row0, int nrows, int precision,
                           float* out, void* stream);
/* Every operand image a training step needs, in ONE launch (round 5): `jobs_dev` = `njobs` of these in DEVICE memory, each the argument set of
 * one mfpa_pack_conv_weights call (same checks apply; the caller validates) plus tile0 = the number of 32 x 32 tiles of all jobs before it
 * (a job has (K / 32) * (nrows / 32) * taps tiles, K = flip_transpose ? Co : Ci); total_tiles = their sum.  Pointers and shapes of a
 * training engine never change, so the table is built once and the launch repeated after every optimiser step. */
typedef struct mfpa_pack_job {
  const float* w; float* out;
  int taps, Co, Ci, flip_transpose, row0, nrows, precision, pad_;
  long long tile0;
} mfpa_pack_job;
int mfpa_pack_conv_weights_batch(const mfpa_pack_job* jobs_dev, int njobs, long long total_tiles, void* stream);
/* torch.optim.Adam step (train.py:661: lr 1e-3, betas (0.9, 0.999), eps 1e-8, no weight decay) on flat
 * arrays; g is multiplied by grad_scale first (1/world_size after a SUM all-reduce). step >= 1. */
int mfpa_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                   float beta2, float eps, int step, float grad_scale, void* stream);

/* ---------------------------------------------------------------------------------------
 * Demucs causal waveform denoiser, forward (next-tier row SURVEY.md §8f-2), training/model.py:22-110,163-326.
 * Activations are time-major (B, L, C) float32.  Every Conv1d / ConvTranspose1d / LSTM projection is one batched
 * GEMM over (possibly overlapping) row windows on the fp32 matrix cores:
 *     C[b][m][n] = epi( sum_k A[b*strideA + m*lda + k] * W[n*K + k] + bias[n] ),  m < M
 *   W (npad, K) with npad a multiple of 64 (rows beyond N zero), K a multiple of 16
 *   mode 0: + bias, optional ReLU (relu = 1);  mode 2: additionally + addend[b*strideAdd + m*ldadd + n]
 *           (relu = 1: after the addition, relu = 2: before it -- `relu(convT(x)) + skip`, model.py:316)
 *   mode 1: GLU -- each 64-row tile of W holds 32 value rows then their 32 gate rows; N = output columns
 *   mode 3: (acc + bias) * (addend[...] > 0) -- the input gradient of a layer whose input was a ReLU output `addend`
 *   C2 (optional, training): mode 1 also stores the pre-activations (bias included) in the packed tile order, row pitch
 *           ldc2 >= npad; relu = 2 also stores relu(acc + bias) before the addition, mode 3 the unmasked acc + bias
 *           (same indexing as C)
 */
typedef struct mfpa_gemm_desc {
  const float* A; long long lda, strideA;
  const float* W; const float* bias;
  const float* addend; long long ldadd, strideAdd;
  float* C; long long ldc, strideC;
  int batch, M, N, K, npad, mode, relu;
  int precision;   /* 0: fp32 MFMA (K multiple of 16); 1: bf16x3 where K >= 128 is a multiple of 32 or K is 48 / 96 (fp32 MFMA otherwise);
                    * 2: bf16x3 with W ALREADY split -- every 32-element chunk of a row as [32 bf16 hi | 32 bf16 lo], w = hi + lo -- for
                    * weights that do not change between calls (K >= 128 a multiple of 32, npad a multiple of 128, no c1_x) */
  /* optional (K <= 256, fp32 kernel): A is COMPUTED while it is staged as the first encoder layer of
   * mfpa_conv1d_c1_relu -- A[b][m][c] = relu(c1_b[c] + sum_j c1_w[j][c] * c1_x[b][4m + j]), c1_x (batch, c1_lin), c1_w (8, K) --
   * so its (B, L, K) output never exists in HBM (model.py:231-238); A may then be NULL. */
  const float* c1_x; long long c1_lin; const float* c1_w; const float* c1_b;
  float* C2; long long ldc2, strideC2;
} mfpa_gemm_desc;
int mfpa_gemm_mfma(const mfpa_gemm_desc* d, void* stream);

/* mix / (floor + std) zero-padded to VL samples, std = unbiased std over time (model.py:293-301). */
int mfpa_demucs_prep(const float* wav, int B, int T, int VL, float floor_, float* out, float* stdv, void* stream);
/* sinc x2 resamplers (model.py:41-88), kernel112 = sinc(t)*hann at the half-sample offsets (model.py:28-38).
 * upsample2: (B,T) -> (B,2T).  downsample2: (B,T) -> first Tkeep (or all ceil(T/2)) samples of the result, row pitch To,
 * each multiplied by scale[b] when scale != NULL (the final `std * x` of model.py:326). */
int mfpa_upsample2(const float* x, int B, int T, const float* kernel112, float* y, void* stream);
int mfpa_downsample2(const float* x, int B, int T, const float* kernel112, float* y, int To, const float* scale,
                     int Tkeep, void* stream);
/* First encoder layer Conv1d(1->C, k8, s4)+ReLU: x (B,Lin) -> y (B,Lout,C), w (8,C) tap-major. */
int mfpa_conv1d_c1_relu(const float* x, int B, int Lin, int Lout, int C, const float* w, const float* bias, float* y,
                        void* stream);
/* The same without the ReLU when relu == 0 (bias may be NULL): with x = the gradient of the last ConvTranspose1d's output
 * and w = its (8, C) taps this is that layer's input gradient. */
int mfpa_conv1d_c1(const float* x, int B, int Lin, int Lout, int C, const float* w, const float* bias, int relu, float* y,
                   void* stream);
/* Last decoder layer ConvTranspose1d(C->1, k8, s4): P (B,L+2,C) with zero first/last rows -> y (B, 4(L+1)), w (8,C). */
int mfpa_convT1d_c1(const float* P, int B, int L, int C, const float* w, float bias, float* y, void* stream);
/* The whole first encoder level in one launch (model.py:66-75,303-307: Conv1d(1, C, 8, 4) + ReLU + Conv1d(C, 2C, 1) + GLU):
 * x (B, Lin) -> y (B, Lout, C), Lout = (Lin - 8) / 4 + 1; the (B, Lout, C) output of the first convolution never exists in
 * memory and, unlike mfpa_gemm_mfma with c1_x set, is evaluated once per row (one workgroup owns all packed GLU columns); the
 * result has the same bits as that form.  w1 (8, C) tap-major, b1 (C) as for mfpa_conv1d_c1; gw (128, C) / gb (128) in the
 * packed GLU tile order.  C must be 48, Lin a multiple of 4 and x 16-byte aligned (a row's 8 samples are read as two float4). */
int mfpa_conv1d_c1_glu(const float* x, int B, int Lin, int Lout, int C, const float* w1, const float* b1, const float* gw,
                       const float* gb, float* y, void* stream);
/* The whole last decoder level in one launch (model.py:80-88,316-318: Conv1d(C, 2C, 1) + GLU + ConvTranspose1d(C, 1, 8, 4)):
 * x (B, L, C) -> y (B, 4 (L + 1)) without the (B, L, C) GLU output in memory.  gw (128, C) / gb (128) = the 1x1 weights and bias
 * in the packed GLU tile order of mfpa_gemm_mfma mode 1, wl (8, C) tap-major as for mfpa_convT1d_c1.  C must be 48 (MFPA_EINVAL
 * otherwise: callers then run mfpa_gemm_mfma mode 1 + mfpa_convT1d_c1). */
int mfpa_glu_convT1d_c1(const float* x, int B, int L, int C, const float* gw, const float* gb, const float* wl, float bias, float* y,
                        void* stream);
/* The same with the bias read from device memory (training: the optimiser updates it there). */
int mfpa_convT1d_c1_dev(const float* P, int B, int L, int C, const float* w, const float* bias_dev, float* y, void* stream);
/* LSTM cell (gate order i,f,g,o; model.py:91-110 via nn.LSTM): gates (B,4H) rows ldg apart, c (B,H) in/out,
 * hout rows ldh apart; optional hsum = h + addend (the first decoder skip). */
int mfpa_lstm_cell(const float* gates, long long ldg, float* c, int B, int H, float* hout, long long ldh, float* hsum,
                   const float* addend, long long ldadd, void* stream);

/* One LSTM time step in ONE launch: gates = hprev W_hh^T + xp (hprev NULL at t = 0: gates = xp), then the cell update of
 * mfpa_lstm_cell.  whh_grouped = W_hh (4H,H) with rows regrouped to [H/16][i16|f16|g16|o16][H] so a workgroup owns all
 * four gates of its 16 hidden units; xp (B,4H) in the standard gate order, rows ldxp apart, includes both biases.
 * bf16x3 products, fp32 accumulate.  H multiple of 128. */
int mfpa_lstm_step(const float* hprev, long long ldhp, const float* whh_grouped, const float* xp, long long ldxp, float* c,
                   int B, int H, float* hout, long long ldh, float* hsum, const float* addend, long long ldadd, void* stream);

/* ---------------------------------------------------------------------------------------
 * Demucs training step (training/train.py:275-312, input_type "audio": loss.backward() through training/model.py:290-326).
 * Input gradients are mfpa_gemm_mfma calls on re-laid-out weights (a Conv1d's input gradient is a ConvTranspose1d of the
 * output gradient and vice versa); the entries below are the rest of the backward pass.  csrc/demucs_train.hip. */
/* mfpa_lstm_step that also keeps what the backward step needs: the gate activations [sig i | sig f | tanh g | sig o] of this
 * step in gsave (B rows ldgs apart; may alias xp) and c_t in cout; the previous cell state is read from cprev (NULL = 0). */
int mfpa_lstm_step_train(const float* hprev, long long ldhp, const float* whh_grouped, const float* xp, long long ldxp,
                         const float* cprev, long long ldcp, float* cout, long long ldco, int B, int H, float* hout, long long ldh,
                         float* hsum, const float* addend, long long ldadd, float* gsave, long long ldgs, void* stream);
/* One LSTM backward time step in one launch: dh = dhout + dgnext W_hh (dgnext = the gate gradients of step t+1, NULL at the
 * last step; whhT = W_hh^T (H, 4H)), then the cell backward: `gates` (the activations mfpa_lstm_step_train saved for step t)
 * is overwritten with the gate pre-activation gradients [di | df | dg | do]; dcstate (B, H) carries dc between steps (zero it
 * before the last step).  ct / cprev: c_t and c_{t-1} (cprev NULL at t = 0).  bf16x3 products.  H multiple of 128. */
int mfpa_lstm_step_bwd(const float* dgnext, long long ldgn, const float* whhT, float* gates, long long ldg, const float* ct,
                       long long ldct, const float* cprev, long long ldcp, const float* dhout, long long lddh, float* dcstate, int B,
                       int H, void* stream);
/* Weight-gradient GEMM: C[m][n] += sum_{b < batch, r < R} A[b*strideA + r*lda + m] * Bm[b*strideB + r*ldb + n]  (C is
 * ACCUMULATED into: zero it first).  Rows are time steps: A = an output gradient (B, L, M), Bm = the layer's input as
 * (possibly overlapping) row windows -- ldb = 4*Cin, N = 8*Cin for the k8/s4 convolutions.  fp32 MFMA
 * (v_mfma_f32_32x32x2_f32: both operands are K-major in HBM, which is that instruction's fragment order), K split over
 * workgroups, float atomics.  M, N, lda, ldb multiples of 4. */
typedef struct mfpa_gemm_tn_desc {
  const float* A; long long lda, strideA;
  const float* Bm; long long ldb, strideB;
  float* C; long long ldc;
  int batch, R, M, N;
  float* colsum;   /* optional (M): colsum[m] += the sum over every row of A[.][m] -- the bias gradient, read off the tiles the
                      workgroups of the first column tile stage anyway */
  int precision;   /* 0: fp32 MFMA; 1: bf16x3 (3 bf16 MFMAs per product); 2: plain bf16 (one MFMA; the sum over every time
                      step of the batch averages the 2^-9 product rounding) -- fragments via ds_read_b64_tr_b16 */
} mfpa_gemm_tn_desc;
int mfpa_gemm_tn(const mfpa_gemm_tn_desc* d, void* stream);
/* GLU backward (nn.GLU(1), model.py:237,246): u (rows, npad) holds the packed pre-activations mfpa_gemm_mfma stored through C2;
 * in place u <- dL/du given dg (rows, N) = dL/d(glu output), row pitch ldg. */
int mfpa_glu_bwd(float* u, long long rows, int npad, int N, const float* dg, long long ldg, void* stream);
/* out[c] += sum_r x[r*ld + c] (bias gradients; out is accumulated into).  C multiple of 4, <= 4096. */
int mfpa_colsum_any(const float* x, long long rows, int C, long long ld, float* out, void* stream);
/* Weight gradient of the two one-channel convolutions (encoder.0.0 and the last ConvTranspose1d), w (8, C) tap-major:
 * dw[j][c] += sum_{b, t < L} x[b*ldx + 4t + j] * g[b*strideG + t*ldg + c].  C <= 256. */
int mfpa_c1_wgrad(const float* x, long long ldx, const float* g, long long ldg, long long strideG, int B, int L, int C, float* dw,
                  void* stream);
/* Adjoint of mfpa_downsample2: dy (B rows ldy apart, nout gradients each, scaled by scale[b] like the forward) -> dx (B, T). */
int mfpa_downsample2_adjoint(const float* dy, int B, int ldy, int nout, const float* kernel112, const float* scale, int T, float* dx,
                             void* stream);

/* Whole-layer forms (one call across the ABI for the Tn launches): (B, Tn, .) contiguous buffers.  mfpa_lstm_layer: xp (B,Tn,4H)
 * input projections incl. biases; inference (train = 0): cstate (B,H) scratch, cseq unused; train = 1: cseq (B,Tn,H) receives
 * c_t and xp is overwritten with the gate activations; xsum / skip as in mfpa_lstm_step (NULL for the first layer).
 * mfpa_lstm_layer_bwd: gates <- gate pre-activation gradients, dcstate (B,H) scratch. */
int mfpa_lstm_layer(const float* whh_grouped, float* xp, float* hseq, float* cseq, float* cstate, int B, int Tn, int H, float* xsum,
                    const float* skip, int train, void* stream);
int mfpa_lstm_layer_bwd(const float* whhT, float* gates, const float* cseq, const float* dhout, float* dcstate, int B, int Tn, int H,
                        void* stream);
/* The same for the time steps [t0, t1) only (backward: t1-1 down to t0), so that the two layers can run as a pipeline on two
 * streams: layer 1 works on chunk k while layer 0 is already on chunk k+1 (backward: the other way round).  The state buffers
 * (cstate / dcstate) are zeroed by the call that holds the first step of the recurrence. */
int mfpa_lstm_layer_range(const float* whh_grouped, float* xp, float* hseq, float* cseq, float* cstate, int B, int Tn, int H,
                          float* xsum, const float* skip, int train, int t0, int t1, void* stream);
int mfpa_lstm_layer_bwd_range(const float* whhT, float* gates, const float* cseq, const float* dhout, float* dcstate, int B, int Tn,
                              int H, int t0, int t1, void* stream);

/* mfpa_lstm_layer_range as ONE persistent launch (lstm_seq_kernel, csrc/demucs.hip): a workgroup keeps the W_hh slice of its 16
 * hidden units in registers for all steps, the workgroups of a 64-clip slab exchange h[t] through `work` (already split into
 * bf16 hi / lo) and meet at a device-memory counter after every step.  Same arguments and results as mfpa_lstm_layer_range
 * (model.py:91-110, torch.nn.LSTM's recurrence); `work` = device scratch of the size mfpa_lstm_seq_work_bytes reports, owned by this
 * layer while the call runs, zeroed once before its first use.  Every wait in the kernel is bounded: if one ever gives up the
 * kernel raises the 32-bit word at byte mfpa_lstm_seq_error_offset() of `work` (and finishes with undefined results); callers
 * read that word before the results leave their operator (the Python host does: ops_demucs.lstm_results_ok).  The grid must be
 * co-resident: the call keeps its own workgroups within `wg_budget` (0 = one per CU of the current device; a caller that runs two
 * such launches side by side, like the chunked two-stream pipeline, passes half the CU count), but it cannot see other work --
 * mfpa_lstm_seq_workgroups reports how many workgroups a launch keeps resident so that a host can account for launches in flight on
 * other streams (ops_demucs._ResidentGuard); a second PROCESS occupying the same GPU is the case the bounded waits and the error word
 * exist for; one process per GPU is the intended deployment.  Shapes outside the persistent kernel's range (H / 128 not in
 * {2,4,6,8}, more workgroups than the budget even with 64-clip slabs) take the per-step path inside the same call. */
int mfpa_lstm_seq_work_bytes(int B, int H, long long* bytes);   /* HOST function: *bytes = size of `work` */
int mfpa_lstm_seq_error_offset(void);
int mfpa_lstm_seq_workgroups(int B, int H, int wg_budget, int* workgroups);   /* HOST function: resident workgroups of the launch, 0 = per-step path */
int mfpa_lstm_layer_seq(const float* whh_grouped, float* xp, float* hseq, float* cseq, float* cstate, int B, int Tn, int H, float* xsum,
                        const float* skip, int train, int t0, int t1, int wg_budget, void* work, void* stream);

/* mfpa_lstm_layer_bwd_range as ONE persistent launch (lstm_bwd_seq_kernel, csrc/demucs_train.hip): the backward recurrence of a layer for
 * steps t1-1 .. t0 (torch.nn.LSTM's backward through time, training/train.py:275-312 via autograd in the reference).  A workgroup keeps
 * the W_hh^T rows of its 16 hidden units in registers (16 x 16 x 32 MFMAs, K = 4H), the workgroups of a 64-clip slab exchange dgates[t]
 * through `work` (already split into bf16 hi / lo) and meet at a device-memory counter after every step.  Same arguments and results as
 * mfpa_lstm_layer_bwd_range; `work`, the bounded waits and the error word as for mfpa_lstm_layer_seq (size: mfpa_lstm_bwd_seq_work_bytes,
 * error word at byte mfpa_lstm_seq_error_offset()).  A workgroup reads its slab's whole dgates[t+1] (K = 4H) every step, so the slab is
 * the smallest of 16 / 32 / 64 clips whose workgroups (slabs x H / 16) fit `wg_budget` (0 = one per CU; a caller that runs two such launches
 * concurrently, like the chunked two-stream pipeline, passes half the CU count: every workgroup of a launch must be resident at once).
 * H / 64 outside {4, 8, 12} or too many workgroups even with 64-clip slabs: the per-step path. */
int mfpa_lstm_bwd_seq_work_bytes(int B, int H, long long* bytes);   /* HOST function */
int mfpa_lstm_bwd_seq_workgroups(int B, int H, int wg_budget, int* workgroups);   /* HOST function, as mfpa_lstm_seq_workgroups */
int mfpa_lstm_layer_bwd_seq(const float* whhT, float* gates, const float* cseq, const float* dhout, float* dcstate, int B, int Tn, int H,
                            int t0, int t1, int wg_budget, void* work, void* stream);

/* ---------------------------------------------------------------------------------------
 * Waveform-domain spectral losses of the Demucs branch, training/loss.py:10-186 (MultiResolutionSTFTLoss; forward).
 * The STFT of a resolution is mfpa_reflect_pad + mfpa_gemm_mfma (precision 0) with a windowed DFT matrix
 * (rows [re bins | zero rows up to im_off | im bins], K = window length padded to a multiple of 16); then:
 *   mfpa_dft_mag        mag[row][k] = sqrt(clamp(re^2 + im^2, 1e-7))                          (loss.py:10-41, `stft`)
 *   mfpa_stft_loss_sums out3 = [sum (|Y|-|X|)^2, sum |Y|^2, sum |log|Y| - log|X||] (float64)   (loss.py:44-83)
 * workspace: 3 * mfpa_loss_blocks() doubles.  mfpa_reflect_pad: out (B, Lout), out[i] = x[reflect(i + shift - pad)]
 * for i + shift < T + 2 pad, else 0 (torch.stft center=True / pad_mode "reflect"; `shift` makes the 16-byte-aligned copy
 * the odd frames of a hop that is not a multiple of 4 need). */
int mfpa_loss_blocks(void);
int mfpa_reflect_pad(const float* x, int B, int T, int pad, int shift, int Lout, float* out, void* stream);
int mfpa_dft_mag(const float* c, long long rows, int bins, long long ldc, int im_off, float* mag, void* stream);
int mfpa_stft_loss_sums(const float* cx, const float* cy, long long rows, int bins, long long ldc, int im_off, double* out3,
                        double* workspace, void* stream);
/* Gradient of one resolution with respect to the PREDICTED signal (the adjoint chain of the forward above):
 *   mfpa_stft_loss_grad       in place on cx: (re, im) <- dL/d(re, im), L = w_sc * sc + w_mag * mag (w_* carry the factors and the
 *                             1/#resolutions); sums = out3 of mfpa_stft_loss_sums (device); zero below the 1e-7 clamp
 *   (GEMM with the transposed DFT matrix: dframes = d(re, im) x W)
 *   mfpa_frames_adjoint       dxp[b][i] = sum_t dframes[b][t][i - t*hop - off]   (overlap-add of the frame gradients)
 *   mfpa_reflect_pad_adjoint  dx (+)= the reflect padding's adjoint of dxp */
int mfpa_stft_loss_grad(float* cx, const float* cy, long long rows, int bins, long long ldc, int im_off, const double* sums,
                        double w_sc, double w_mag, void* stream);
int mfpa_frames_adjoint(const float* dframes, int B, int frames, long long ldf, int win, int hop, int off, int L, float* dxp,
                        void* stream);
int mfpa_reflect_pad_adjoint(const float* dxp, int B, int T, int pad, int L, int accumulate, float* dx, void* stream);

/* ---------------------------------------------------------------------------------------
 * AugmentFP signal chain (next-tier row SURVEY.md §8f-3), augmentation/__init__.py:46-93 and
 * augmentation/transformations/ (every transform).  Waveforms are (B, T) float32; `apply` (B) uint8 is the transform's Bernoulli
 * gate (0 = copy the example through).  The random draws are made by the host like the reference does.
 */
/* Windowed-sinc low-pass taps, julius 0.2.7 design (zeros 8; pass_filters.py:100 -- julius is NOT in the reference
 * tree: parity unpinned): example b owns the 2*half[b]+1 taps at taps + tap_off[b] (ragged: a 0.5 Hz cut-off at 8 kHz
 * has 128 001 taps, a 3.5 kHz one 19), cutoff = f_c / sample_rate. */
int mfpa_lowpass_taps(const float* cutoff, const int* half, const long long* tap_off, int B, float* taps, void* stream);
/* y[t] = sum_{k < ntaps[b]} taps[tap_off[b] + k] * xpad[t + k - off[b]],  pad_mode 0 replicate / 1 zero;
 * out_mode 0: y (LowPassFilter);  1: x - y (HighPassFilter, pass_filters.py:158-171);
 * out_mode 2: impulse response (impulse_response.py:73-117): taps = time-reversed IR, off = n-1, stores t < T and
 *             writes peak[b] = max |y[t]| over ALL t < Tout = T + n_max - 1 (the reference normalises by the peak of the
 *             full convolution before truncating). */
int mfpa_fir(const float* x, int B, int T, int Tout, const float* taps, const long long* tap_off, const int* ntaps,
             const int* off, const uint8_t* apply, int pad_mode, int out_mode, float* y, float* peak, void* stream);
/* y[b] = x[b] * factor[b] (invert 0: Gain, gain.py:62-70) or x[b] / factor[b] (invert 1) where apply[b] (NULL = all). */
int mfpa_scale_rows(const float* x, int B, int T, const float* factor, const uint8_t* apply, int invert, float* y,
                    void* stream);
/* AddBackgroundNoise.random_background (background_noise.py:64-141, non-mixup branch): out[b] (T samples) = the
 * concatenation of up to P slices bank[src[b*P+p] .. + len[b*P+p]) (len 0 ends the list; the lengths of one example sum
 * to T) of a device-resident noise bank, every slice RMS-normalised (x / (rms + 1e-8), utils.py:190-205) and the whole
 * RMS-normalised again.  The host draws scene / file / offset like the reference (python `random`). */
int mfpa_gather_background(const float* bank, const long long* src, const int* len, int B, int P, int T, float* out,
                           void* stream);
/* AddBackgroundNoise (background_noise.py:183-215): y = x + rms(x)/10^(snr/20) * noise, then y /= max|y|.
 * With noise == NULL: PeakNormalization (peak_normalization.py:38-67): y = x / max|x| when the peak is > 0. */
int mfpa_mix_background(const float* x, int B, int T, const float* noise, const float* snr_db, const uint8_t* apply,
                        float* y, void* stream);
/* Clipping (clipping.py:67-100) one example at a time, as AugmentFP.__call__ applies it: clamp to torch.quantile(x, p/2)
 * and torch.quantile(x, 1 - p/2) (linear interpolation), found by an in-LDS radix select. */
int mfpa_clip_quantile(const float* x, int B, int T, const float* pct, const uint8_t* apply, float* y, void* stream);
/* The same transform as the reference's batch_augment applies it to B > 1 examples (clipping.py:77-93: torch.quantile without
 * a dim argument): example b is clamped to the pct[b]/2 and 1 - pct[b]/2 quantiles of ALL selected examples flattened
 * together.  n_apply = number of non-zero entries of apply; n_apply * T <= 16 000 000 (torch.quantile's limit) else EINVAL. */
int mfpa_clip_quantile_flat(const float* x, int B, int T, const float* pct, const uint8_t* apply, int n_apply, float* y,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MFPA_H */
