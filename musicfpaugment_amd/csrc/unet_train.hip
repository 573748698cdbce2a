// Training-step kernels of the UNet denoiser for MI355X (gfx950): the pieces of
// Trainer.train_epoch's spec branch (training/train.py:257-317 of the reference) that are not the
// convolution itself (csrc/unet.hip):
//
//   BatchNorm batch statistics + running-stat update, BN/ReLU/max-pool forward glue,
//   BN/ReLU backward (reduce + apply), max-pool backward, weight gradients on MFMA,
//   bias / OutConv gradients, L1 loss forward+backward, fused Adam.
//
// Design: a layer's BatchNorm+ReLU output is never materialised.  The convolution writes its raw
// output z, one bandwidth pass reduces the batch statistics into a per-channel (scale, shift),
// and every consumer (next conv, transposed conv, pool, weight-gradient kernel) applies
// relu(z*scale+shift) while loading.  Reductions are two-stage float64 (per-workgroup partials,
// then one finishing workgroup) so they are deterministic and independent of the grid.
#include "mfpa_common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RED_BLOCKS = 1024;

// four consecutive channels of an NHWC activation tensor kept as float32 or (z16, round 5: the plain-bf16 step's activations) as bfloat16;
// e = element index of the first of the four
__device__ __forceinline__ f32x4 ld_act4(const float* base, size_t e, int z16) {
  if (z16) {
    const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + e);
    return f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
  }
  return *reinterpret_cast<const f32x4*>(base + e);
}

// ------------------------------------------------------------------ per-channel sums over pixels
// partial[(blk*C + c)*NV + v]: NV float64 sums per channel.  MODE 0: {sum z, sum z^2} (BN statistics);
// MODE 1: {sum g, sum g*xhat} with g = dy*[z*scale+shift > 0], xhat = (z-mean)*invstd (BN backward);
// MODE 2: {sum x} (bias gradient).
template <int MODE>
__global__ __launch_bounds__(256) void chan_reduce_kernel(const float* __restrict__ x, const float* __restrict__ z,
                                                          long long npix, int C, const float* __restrict__ scale,
                                                          const float* __restrict__ shift,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ invstd,
                                                          double* __restrict__ partial, unsigned drop_seed,
                                                          unsigned drop_thresh, float drop_scale, int z16) {
  // z16: the activation tensor (x in MODE 0, z in MODE 1) is bfloat16
  constexpr int NV = (MODE == 2) ? 1 : 2;
  const int C4 = C / 4;
  const int lanes = C4 < 256 ? C4 : 256;          // lanes across channel quads
  const int rows = 256 / lanes;                   // pixels in flight per block
  const int cq0 = threadIdx.x % lanes, prow = threadIdx.x / lanes;
  __shared__ double sh[256 * 8];
  for (int cq = cq0; cq < C4; cq += lanes) {
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    f32x4 sc = {0, 0, 0, 0}, sf = {0, 0, 0, 0}, mu = {0, 0, 0, 0}, is = {0, 0, 0, 0};
    if (MODE == 1) {
      sc = *reinterpret_cast<const f32x4*>(scale + 4 * cq);
      sf = *reinterpret_cast<const f32x4*>(shift + 4 * cq);
      mu = *reinterpret_cast<const f32x4*>(mean + 4 * cq);
      is = *reinterpret_cast<const f32x4*>(invstd + 4 * cq);
    }
    for (long long p = (long long)blockIdx.x * rows + prow; p < npix; p += (long long)gridDim.x * rows) {
      const f32x4 v = (MODE == 0) ? ld_act4(x, (size_t)p * C + 4 * cq, z16) : *reinterpret_cast<const f32x4*>(x + (size_t)p * C + 4 * cq);
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          s0[k] += (double)v[k];
          s1[k] += (double)v[k] * (double)v[k];
        }
      } else if (MODE == 1) {
        const f32x4 zz = ld_act4(z, (size_t)p * C + 4 * cq, z16);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float g = (zz[k] * sc[k] + sf[k] > 0.f) ? v[k] : 0.f;
          if (drop_thresh) g = mfpa_keep(drop_seed, drop_thresh, (unsigned long long)p * C + 4 * cq + k) ? g * drop_scale : 0.f;
          const float xh = (zz[k] - mu[k]) * is[k];
          s0[k] += (double)g;
          s1[k] += (double)g * (double)xh;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) s0[k] += (double)v[k];
      }
    }
    // combine the `rows` pixel-rows of this block for channel quad cq
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sh[threadIdx.x * 8 + k] = s0[k];
      sh[threadIdx.x * 8 + 4 + k] = s1[k];
    }
    __syncthreads();
    if (prow == 0) {
      for (int r = 1; r < rows; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          s0[k] += sh[(r * lanes + cq0) * 8 + k];
          s1[k] += sh[(r * lanes + cq0) * 8 + 4 + k];
        }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double* dst = partial + ((size_t)blockIdx.x * C + 4 * cq + k) * NV;
        dst[0] = s0[k];
        if (NV == 2) dst[1] = s1[k];
      }
    }
    __syncthreads();
  }
}

// Finish kernels: one 64-lane wavefront per channel sums the per-workgroup float64 partials (lane-strided, then a
// fixed xor-shuffle tree: deterministic), lane 0 finalises.
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// BN statistics finish: mean, biased var -> invstd, fused (scale, shift); running stats with momentum
// (unbiased variance), exactly nn.BatchNorm2d in train mode (training/unet.py:17,20).
__global__ __launch_bounds__(256) void bn_stats_finish_kernel(const double* __restrict__ partial, int nblk, int C,
                                                              double count, float eps, float momentum,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ mean,
                                                              float* __restrict__ invstd, float* __restrict__ scale,
                                                              float* __restrict__ shift, float* __restrict__ running_mean,
                                                              float* __restrict__ running_var) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  double s = 0, ss = 0;
  for (int b = lane; b < nblk; b += 64) {
    s += partial[((size_t)b * C + c) * 2];
    ss += partial[((size_t)b * C + c) * 2 + 1];
  }
  s = wave_sum_d(s);
  ss = wave_sum_d(ss);
  if (lane) return;
  const double m = s / count;
  double var = ss / count - m * m;
  if (var < 0) var = 0;
  const double is = 1.0 / sqrt(var + (double)eps);
  mean[c] = (float)m;
  invstd[c] = (float)is;
  const float sc = gamma[c] * (float)is;
  scale[c] = sc;
  shift[c] = beta[c] - (float)m * sc;
  if (running_mean) {
    const double unb = count > 1 ? var * count / (count - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

// BN backward finish: dgamma = sum g*xhat, dbeta = sum g; coefficients so that
// dz = ka*g - kb - kc*xhat  with ka = gamma*invstd, kb = ka*dbeta/N, kc = ka*dgamma/N.
__global__ __launch_bounds__(256) void bn_bwd_finish_kernel(const double* __restrict__ partial, int nblk, int C,
                                                            double count, const float* __restrict__ gamma,
                                                            const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ coef /* [3][C] */) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  double sg = 0, sgx = 0;
  for (int b = lane; b < nblk; b += 64) {
    sg += partial[((size_t)b * C + c) * 2];
    sgx += partial[((size_t)b * C + c) * 2 + 1];
  }
  sg = wave_sum_d(sg);
  sgx = wave_sum_d(sgx);
  if (lane) return;
  dgamma[c] = (float)sgx;
  dbeta[c] = (float)sg;
  const double ka = (double)gamma[c] * (double)invstd[c];
  coef[c] = (float)ka;
  coef[C + c] = (float)(ka * sg / count);
  coef[2 * C + c] = (float)(ka * sgx / count);
}

// SyncBN building blocks: per-channel partial pairs -> one [C][2] float64 row (the caller all-reduces it across ranks) ...
__global__ __launch_bounds__(256) void pair_sums_kernel(const double* __restrict__ partial, int nblk, int C,
                                                        double* __restrict__ sums) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  double s = 0, ss = 0;
  for (int b = lane; b < nblk; b += 64) {
    s += partial[((size_t)b * C + c) * 2];
    ss += partial[((size_t)b * C + c) * 2 + 1];
  }
  s = wave_sum_d(s);
  ss = wave_sum_d(ss);
  if (lane == 0) { sums[2 * c] = s; sums[2 * c + 1] = ss; }
}

// per-wave partial rows of conv_wd16_kernel ([rows][2][C] floats) -> (sum, sum of squares) pairs in float64: a block walks its rows with
// one channel per lane (coalesced), the blocks' results go through the two-stage float64 reduction the other statistics use
__global__ __launch_bounds__(256) void conv_stats_rows_kernel(const float* __restrict__ part, long long rows, int C,
                                                              double* __restrict__ partial) {
  const int lanes = C < 256 ? C : 256, groups = 256 / lanes;
  const int c0 = threadIdx.x % lanes, gq = threadIdx.x / lanes;
  __shared__ double sh[2 * 256];
  for (int c = c0; c < C; c += lanes) {
    double s = 0, q = 0;
    for (long long r = (long long)blockIdx.x * groups + gq; r < rows; r += (long long)gridDim.x * groups) {
      s += (double)part[(size_t)r * 2 * C + c];
      q += (double)part[(size_t)r * 2 * C + C + c];
    }
    sh[threadIdx.x] = s; sh[256 + threadIdx.x] = q;
    __syncthreads();
    if (gq == 0) {
      for (int k = 1; k < groups; ++k) { s += sh[k * lanes + c0]; q += sh[256 + k * lanes + c0]; }
      partial[((size_t)blockIdx.x * C + c) * 2] = s;
      partial[((size_t)blockIdx.x * C + c) * 2 + 1] = q;
    }
    __syncthreads();
  }
}

// ... and the backward finish from LOCAL sums (this rank's dgamma / dbeta: the gradient all-reduce adds the ranks) and GLOBAL
// sums over `count` pixels of all ranks (the mean terms of the input gradient).
__global__ __launch_bounds__(256) void bn_bwd_finish_sync_kernel(const double* __restrict__ local, const double* __restrict__ global,
                                                                 int C, double count, const float* __restrict__ gamma,
                                                                 const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, float* __restrict__ coef) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  dgamma[c] = (float)local[2 * c + 1];
  dbeta[c] = (float)local[2 * c];
  const double ka = (double)gamma[c] * (double)invstd[c];
  coef[c] = (float)ka;
  coef[C + c] = (float)(ka * global[2 * c] / count);
  coef[2 * C + c] = (float)(ka * global[2 * c + 1] / count);
}

__global__ __launch_bounds__(256) void colsum_finish_kernel(const double* __restrict__ partial, int nblk, int C,
                                                            float* __restrict__ out) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  double s = 0;
  for (int b = lane; b < nblk; b += 64) s += partial[(size_t)b * C + c];
  s = wave_sum_d(s);
  if (lane == 0) out[c] = (float)s;
}

// dy <- dz in place (BatchNorm + ReLU backward), all per-channel constants precomputed.
typedef __bf16 wg_bf16x4_t __attribute__((ext_vector_type(4)));
// RANK1: the incoming gradient is the OutConv's, dy[p][c] = dpred[p] * w1[c] (training/unet.py:94-96 backward): it is formed here from the
// (B, H, W) dpred and the 64 weights instead of being written by mfpa_outconv_bwd and read back (2 x 1.06 GB per 64-clip step); dz goes to `dy`.
template <bool RANK1>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(float* __restrict__ dy, const float* __restrict__ z,
                                                           long long npix, int C, const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ coef, unsigned drop_seed,
                                                           unsigned drop_thresh, float drop_scale, __bf16* __restrict__ dz16,
                                                           int write_f32, const float* __restrict__ dpred, const float* __restrict__ w1, int z16, int dy16) {
  // dy16: the incoming gradient is a bfloat16 tensor (the input-gradient convolution left it so: mfpa_conv_desc.y_bf16 without y); then only dz16 is written
  const int C4 = C / 4;                      // a power of two (checked on the host): no 64-bit division per element
  const int c4_shift = __ffs(C4) - 1;
  const long long total = npix * C4;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int cq = (int)(e & (C4 - 1));
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + 4 * cq);
    const f32x4 sf = *reinterpret_cast<const f32x4*>(shift + 4 * cq);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + 4 * cq);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + 4 * cq);
    const f32x4 ka = *reinterpret_cast<const f32x4*>(coef + 4 * cq);
    const f32x4 kb = *reinterpret_cast<const f32x4*>(coef + C + 4 * cq);
    const f32x4 kc = *reinterpret_cast<const f32x4*>(coef + 2 * C + 4 * cq);
    f32x4 g;
    if (RANK1) {
      const float gp = dpred[e >> c4_shift];              // (C4 is a power of two: no 64-bit division per element)
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w1 + 4 * cq);
#pragma unroll
      for (int k = 0; k < 4; ++k) g[k] = gp * wv[k];
    } else {
      g = ld_act4(dy, (size_t)e * 4, dy16);
    }
    const f32x4 zz = ld_act4(z, (size_t)e * 4, z16);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gg = (zz[k] * sc[k] + sf[k] > 0.f) ? g[k] : 0.f;
      if (drop_thresh) gg = mfpa_keep(drop_seed, drop_thresh, (unsigned long long)e * 4 + k) ? gg * drop_scale : 0.f;
      const float xh = (zz[k] - mu[k]) * is[k];
      o[k] = ka[k] * gg - kb[k] - kc[k] * xh;
    }
    if (write_f32) *reinterpret_cast<f32x4*>(dy + e * 4) = o;   // (0: every consumer of dz reads the bf16 copy -- the plain-bf16 train step)
    if (dz16) {                              // the bf16 copy the weight-gradient kernel reads (wgrad precision 3)
      wg_bf16x4_t h;
#pragma unroll
      for (int k = 0; k < 4; ++k) h[k] = (__bf16)o[k];
      *reinterpret_cast<wg_bf16x4_t*>(dz16 + e * 4) = h;
    }
  }
}

// The same pass for bfloat16 z (the plain-bf16 step's activations), EIGHT channels per thread: z arrives as one 16-byte load, dy as two, the
// bf16 dz leaves as one 16-byte store (with four channels per thread the 8-byte z loads / dz stores ran the pass at 4.4 TB/s).
template <bool RANK1>
__global__ __launch_bounds__(256) void bn_bwd_apply8_kernel(float* __restrict__ dy, const unsigned short* __restrict__ z16,
                                                            long long npix, int C, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ coef,
                                                            unsigned drop_seed, unsigned drop_thresh, float drop_scale,
                                                            __bf16* __restrict__ dz16, int write_f32,
                                                            const float* __restrict__ dpred, const float* __restrict__ w1) {
  const int C8 = C / 8;                      // a power of two (checked on the host)
  const int c8_shift = __ffs(C8) - 1;
  const long long total = npix * C8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int c0 = 8 * (int)(e & (C8 - 1));
    f32x4 g[2];
    if (RANK1) {
      const float gp = dpred[e >> c8_shift];                // (C8 is a power of two: no 64-bit division per element)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w1 + c0 + 4 * h);
#pragma unroll
        for (int k = 0; k < 4; ++k) g[h][k] = gp * wv[k];
      }
    } else {
      g[0] = *reinterpret_cast<const f32x4*>(dy + e * 8);
      g[1] = *reinterpret_cast<const f32x4*>(dy + e * 8 + 4);
    }
    const uint4 zr = *reinterpret_cast<const uint4*>(z16 + e * 8);
    const unsigned zu[4] = {zr.x, zr.y, zr.z, zr.w};
    f32x4 o[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c0 + 4 * h), sf = *reinterpret_cast<const f32x4*>(shift + c0 + 4 * h);
      const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c0 + 4 * h), is = *reinterpret_cast<const f32x4*>(invstd + c0 + 4 * h);
      const f32x4 ka = *reinterpret_cast<const f32x4*>(coef + c0 + 4 * h), kb = *reinterpret_cast<const f32x4*>(coef + C + c0 + 4 * h);
      const f32x4 kc = *reinterpret_cast<const f32x4*>(coef + 2 * C + c0 + 4 * h);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned u = zu[2 * h + (k >> 1)];
        const float zz = __uint_as_float((k & 1) ? (u & 0xffff0000u) : (u << 16));
        float gg = (zz * sc[k] + sf[k] > 0.f) ? g[h][k] : 0.f;
        if (drop_thresh) gg = mfpa_keep(drop_seed, drop_thresh, (unsigned long long)e * 8 + 4 * h + k) ? gg * drop_scale : 0.f;
        o[h][k] = ka[k] * gg - kb[k] - kc[k] * ((zz - mu[k]) * is[k]);
      }
    }
    if (write_f32) {
      *reinterpret_cast<f32x4*>(dy + e * 8) = o[0];
      *reinterpret_cast<f32x4*>(dy + e * 8 + 4) = o[1];
    }
    if (dz16) {
      wg_bf16x4_t h0, h1;
#pragma unroll
      for (int k = 0; k < 4; ++k) { h0[k] = (__bf16)o[0][k]; h1[k] = (__bf16)o[1][k]; }
      uint4 pk;
      const uint2 a = __builtin_bit_cast(uint2, h0), b = __builtin_bit_cast(uint2, h1);
      pk.x = a.x; pk.y = a.y; pk.z = b.x; pk.w = b.y;
      *reinterpret_cast<uint4*>(dz16 + e * 8) = pk;
    }
  }
}

// p = maxpool2(relu(z*scale+shift)), floor (unet.py:34 after the BN+ReLU of the DoubleConv).
__global__ __launch_bounds__(256) void bn_relu_pool_kernel(const float* __restrict__ z, int B, int H, int W, int C,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, float* __restrict__ p,
                                                           unsigned drop_seed, unsigned drop_thresh, float drop_scale, int z16, int p16) {
  // z16: z is bfloat16; p16: the pooled activation leaves as bfloat16 (the next convolution's operand as it is: mfpa_conv_desc.x0_is_bf16)
  const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
  const int b = blockIdx.x / Ho, yo = blockIdx.x % Ho;     // one workgroup per pooled row: 32-bit index math only
  for (int e32 = threadIdx.x; e32 < Wo * C4; e32 += 256) {
    const int cq = e32 % C4, xo = e32 / C4;
    const size_t e = ((size_t)blockIdx.x * Wo + xo) * C4 + cq;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + 4 * cq);
    const f32x4 sf = *reinterpret_cast<const f32x4*>(shift + 4 * cq);
    const size_t base = (((size_t)b * H + 2 * yo) * W + 2 * xo) * C + 4 * cq;
    f32x4 m = {0.f, 0.f, 0.f, 0.f};   // relu output is >= 0
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const size_t off = ((size_t)(t >> 1) * W + (t & 1)) * C;
      const f32x4 v = ld_act4(z, base + off, z16);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float y = v[k] * sc[k] + sf[k];
        if (drop_thresh) y = (y > 0.f && mfpa_keep(drop_seed, drop_thresh, (unsigned long long)base + off + k)) ? y * drop_scale : 0.f;
        m[k] = y > m[k] ? y : m[k];
      }
    }
    if (p16) {
      wg_bf16x4_t h;
#pragma unroll
      for (int k = 0; k < 4; ++k) h[k] = (__bf16)m[k];
      *reinterpret_cast<wg_bf16x4_t*>(reinterpret_cast<__bf16*>(p) + e * 4) = h;
    } else *reinterpret_cast<f32x4*>(p + e * 4) = m;
  }
}

// dy_full += route(dp): the gradient of a pooled cell goes to the first maximum of its 2x2 window of
// y = relu(z*scale+shift) (scan order (0,0),(0,1),(1,0),(1,1), strict >), as torch's max_pool2d backward.
// SUMS: the pass also holds everything the BatchNorm backward of this very layer reduces next -- the finished dy and z of every pixel --
// so it forms that reduction's per-channel partial sums {sum g, sum g * xhat} (g = dy * [z*scale+shift > 0] (* dropout), xhat =
// (z - mean) * invstd: chan_reduce_kernel<1>'s terms) for its two rows, the odd last column and (last workgroup of a clip) the odd last
// row included, and writes them as one row of `part` (rows x 2 x C float, the layout of the convolutions' stats_part, finished in
// float64 by mfpa_conv_stats_reduce): the separate reduction pass over dy and z (2.1 GB at the first level) is not run.
// MODE 0: dy += route(dp); 1: ... and the BatchNorm-backward partial sums (SUMS); 2 (round 5): NOTHING of dy is written -- the pass forms
// g = dy + route(dp) again and applies the BatchNorm + ReLU backward to it on the spot (coef = bn_bwd_finish_kernel's [3][C]), writing only the
// bfloat16 dz: with MODE 1 run as a pure reduction in front (`nowrite`), the finished float32 dy of an encoder block never exists.
// d16: dy (the skip path's gradient) is a bfloat16 tensor (then nothing can be written back into it: nowrite or MODE 2).
template <int MODE>
__global__ __launch_bounds__(256) void maxpool_bwd_add_kernel(const float* __restrict__ z, int B, int H, int W, int C,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              const float* __restrict__ dp, float* __restrict__ dy,
                                                              unsigned drop_seed, unsigned drop_thresh, float drop_scale,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              float* __restrict__ part, int z16, int d16, int nowrite,
                                                              const float* __restrict__ coef, __bf16* __restrict__ dz16) {
  constexpr bool SUMS = MODE == 1;
  const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
  const int b = blockIdx.x / Ho, yo = blockIdx.x % Ho;     // one workgroup per pooled row: 32-bit index math only
  // SUMS: 256 % C4 == 0 (checked by the launcher), so a thread meets ONE channel quad in all its trips: tid % C4
  // (float64 like the chan_reduce_kernel<1> pass this replaces and like outconv_bwd_kernel<SUMS>: the two sums cancel heavily, and a float32
  //  running sum over a row's ~500 pixels would put its rounding, relative to sum |g|, into dgamma / dbeta / dz of every engine precision)
  double s0[4] = {0., 0., 0., 0.}, s1[4] = {0., 0., 0., 0.};
  f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = {0.f, 0.f, 0.f, 0.f}, scq = {0.f, 0.f, 0.f, 0.f}, sfq = {0.f, 0.f, 0.f, 0.f};
  f32x4 ka = {0.f, 0.f, 0.f, 0.f}, kb = {0.f, 0.f, 0.f, 0.f}, kc = {0.f, 0.f, 0.f, 0.f};
  if (MODE >= 1) {
    const int cqt = threadIdx.x % C4;
    mu = *reinterpret_cast<const f32x4*>(mean + 4 * cqt);
    is = *reinterpret_cast<const f32x4*>(invstd + 4 * cqt);
    scq = *reinterpret_cast<const f32x4*>(scale + 4 * cqt);
    sfq = *reinterpret_cast<const f32x4*>(shift + 4 * cqt);
    if (MODE == 2) {
      ka = *reinterpret_cast<const f32x4*>(coef + 4 * cqt);
      kb = *reinterpret_cast<const f32x4*>(coef + C + 4 * cqt);
      kc = *reinterpret_cast<const f32x4*>(coef + 2 * C + 4 * cqt);
    }
  }
  // MODE 2: one pixel's four channels through the BatchNorm + ReLU (+ dropout) backward (bn_bwd_apply_kernel's formula), bf16 dz out
  auto apply = [&](const f32x4& d, const f32x4& v, size_t elem0) {
    wg_bf16x4_t h;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gg = (v[k] * scq[k] + sfq[k] > 0.f) ? d[k] : 0.f;
      if (drop_thresh) gg = mfpa_keep(drop_seed, drop_thresh, (unsigned long long)elem0 + k) ? gg * drop_scale : 0.f;
      h[k] = (__bf16)(ka[k] * gg - kb[k] - kc[k] * ((v[k] - mu[k]) * is[k]));
    }
    *reinterpret_cast<wg_bf16x4_t*>(dz16 + elem0) = h;
  };
  // one pixel's four channels into the sums (its finished gradient d, its z)
  auto add = [&](const f32x4& d, const f32x4& v, size_t elem0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float g = (v[k] * scq[k] + sfq[k] > 0.f) ? d[k] : 0.f;
      if (drop_thresh) g = mfpa_keep(drop_seed, drop_thresh, (unsigned long long)elem0 + k) ? g * drop_scale : 0.f;
      s0[k] += (double)g;
      s1[k] += (double)g * (double)((v[k] - mu[k]) * is[k]);
    }
  };
  for (int e32 = threadIdx.x; e32 < Wo * C4; e32 += 256) {
    const int cq = e32 % C4, xo = e32 / C4;
    const size_t e = ((size_t)blockIdx.x * Wo + xo) * C4 + cq;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + 4 * cq);
    const f32x4 sf = *reinterpret_cast<const f32x4*>(shift + 4 * cq);
    const size_t base = (((size_t)b * H + 2 * yo) * W + 2 * xo) * C + 4 * cq;
    const f32x4 g = *reinterpret_cast<const f32x4*>(dp + e * 4);
    float best[4];
    int arg[4];
    f32x4 vz[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 v = ld_act4(z, base + ((size_t)(t >> 1) * W + (t & 1)) * C, z16);
      vz[t] = v;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float y = v[k] * sc[k] + sf[k];
        y = y > 0.f ? y : 0.f;
        if (drop_thresh) y = mfpa_keep(drop_seed, drop_thresh, (unsigned long long)base + ((size_t)(t >> 1) * W + (t & 1)) * C + k) ? y * drop_scale : 0.f;
        if (t == 0 || y > best[k]) {
          best[k] = y;
          arg[k] = t;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const size_t off = base + ((size_t)(t >> 1) * W + (t & 1)) * C;
      f32x4 cur = ld_act4(dy, off, d16);
#pragma unroll
      for (int k = 0; k < 4; ++k) cur[k] += (arg[k] == t) ? g[k] : 0.f;
      if (MODE != 2 && !nowrite) *reinterpret_cast<f32x4*>(dy + off) = cur;
      if (SUMS) add(cur, vz[t], off);
      if (MODE == 2) apply(cur, vz[t], off);
    }
  }
  if (MODE >= 1) {
    if (W & 1) {                                            // the last column lies in no window: its gradient is the skip path's alone
      for (int e32 = threadIdx.x; e32 < 2 * C4; e32 += 256) {
        const int cq = e32 % C4, r = e32 / C4;
        const size_t off = (((size_t)b * H + 2 * yo + r) * W + (W - 1)) * C + 4 * cq;
        if (SUMS) add(ld_act4(dy, off, d16), ld_act4(z, off, z16), off);
        else apply(ld_act4(dy, off, d16), ld_act4(z, off, z16), off);
      }
    }
    if ((H & 1) && yo == Ho - 1) {                          // ... and so does the last row: the clip's last workgroup takes it
      for (int e32 = threadIdx.x; e32 < W * C4; e32 += 256) {
        const int cq = e32 % C4, x = e32 / C4;
        const size_t off = (((size_t)b * H + (H - 1)) * W + x) * C + 4 * cq;
        if (SUMS) add(ld_act4(dy, off, d16), ld_act4(z, off, z16), off);
        else apply(ld_act4(dy, off, d16), ld_act4(z, off, z16), off);
      }
    }
  }
  if (SUMS) {
    __shared__ double red[256 * 8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      red[threadIdx.x * 8 + k] = s0[k];
      red[threadIdx.x * 8 + 4 + k] = s1[k];
    }
    __syncthreads();
    if ((int)threadIdx.x < C4) {                            // fixed order: deterministic
      for (int r = 1; r < 256 / C4; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          s0[k] += red[(r * C4 + threadIdx.x) * 8 + k];
          s1[k] += red[(r * C4 + threadIdx.x) * 8 + 4 + k];
        }
      float* row = part + (size_t)blockIdx.x * 2 * C;
      *reinterpret_cast<f32x4*>(row + 4 * threadIdx.x) = f32x4{(float)s0[0], (float)s0[1], (float)s0[2], (float)s0[3]};
      *reinterpret_cast<f32x4*>(row + C + 4 * threadIdx.x) = f32x4{(float)s1[0], (float)s1[1], (float)s1[2], (float)s1[3]};
    }
  }
}

// ------------------------------------------------------------------ weight gradient on MFMA
// dW[tap][co][ci] += sum over pixels of dz[p][co] * xin[p + tap][ci]   (MODE 0, 3x3 conv, 9 taps)
// dW[tap][co][ci] += sum over pixels of dup[2y+dy, 2x+dx][co] * xin[y,x][ci]   (MODE 1, transposed conv, 4 taps)
// GEMM per tap: M = 64 output channels, N = 64 input channels, K = pixels.  A workgroup owns one 64x64
// (co, ci) tile for ALL taps and walks 2x32-pixel patches (grid-strided over the batch): per patch the
// dz tile and the haloed xin tile are staged in LDS pixel-major (the natural NHWC order, so no transpose:
// MFMA lanes read 32 consecutive channels of one pixel with ds_read_b32); each wave keeps one 32x32
// accumulator per tap (9 x 16 VGPRs) and the A fragment of a k-step is reused by all taps.  Partial sums
// are added to dW with one float atomic per element per workgroup (256-B contiguous segments).
struct WgradArgs {
  const float* dz;        // MODE 0: (B,H,W,Cout);  MODE 1: (B,2H,2W,Cout)
  const float* x0;        // (B,H,W,C0)
  const float* in_scale0; // optional affine+ReLU on load for x0
  const float* in_shift0;
  const float* x1;        // (B,H1,W1,C1) zero-padded (MODE 0 only)
  float* dw;              // [taps][Cout][C0+C1]
  int C0, C1, H1, W1, oy1, ox1;
  int B, H, W, Cout;
  int tiles_x, tiles_y;
  unsigned drop_seed, drop_thresh;
  float drop_scale;
  int xcd;                  // bf16 kernels: XCD-aware (patch group, tile) order
};

constexpr int WG_PH = 2, WG_PW = 32, WG_PIX = 64, WG_T = 64;

template <int MODE>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(WgradArgs a) {
  constexpr int HALO = (MODE == 0) ? 1 : 0;
  constexpr int TAPS = (MODE == 0) ? 9 : 4;
  constexpr int HPW = WG_PW + 2 * HALO, HPH = WG_PH + 2 * HALO, HP = HPW * HPH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Ds = reinterpret_cast<float*>(smem);   // [128][64]  dz tile (pixel-major)
  float* Xs = Ds + WG_PIX * WG_T;               // [HP][64]   xin tile with halo (pixel-major)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int cot = wave & 1, cit = wave >> 1;
  const int co0 = blockIdx.x * WG_T, ci0 = blockIdx.y * WG_T;
  const int Cin = a.C0 + a.C1;
  const bool from0 = ci0 < a.C0;
  const long long npatch = (long long)a.B * a.tiles_x * a.tiles_y;

  floatx16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  for (long long patch = blockIdx.z; patch < npatch; patch += gridDim.z) {
    long long q = patch;
    const int tx = (int)(q % a.tiles_x); q /= a.tiles_x;
    const int ty = (int)(q % a.tiles_y);
    const int b = (int)(q / a.tiles_y);
    const int y0 = ty * WG_PH, x0p = tx * WG_PW;
    __syncthreads();   // previous patch's fragment reads are done
    // stage xin (+halo): HP pixels x 64 channels
    for (int idx = tid; idx < HP * (WG_T / 4); idx += 256) {
      const int pix = idx / (WG_T / 4), c4 = idx % (WG_T / 4);
      const int gy = y0 + pix / HPW - HALO, gx = x0p + pix % HPW - HALO;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
        if (from0) {
          v = *reinterpret_cast<const f32x4*>(a.x0 + (((size_t)b * a.H + gy) * a.W + gx) * a.C0 + ci0 + 4 * c4);
          if (a.in_scale0) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(a.in_scale0 + ci0 + 4 * c4);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(a.in_shift0 + ci0 + 4 * c4);
            v = v * sc + sh;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
            if (a.drop_thresh) {
              const unsigned long long e0 = (((unsigned long long)b * a.H + gy) * a.W + gx) * a.C0 + ci0 + 4 * c4;
#pragma unroll
              for (int k = 0; k < 4; ++k) v[k] = mfpa_keep(a.drop_seed, a.drop_thresh, e0 + k) ? v[k] * a.drop_scale : 0.f;
            }
          }
        } else {
          const int y1 = gy - a.oy1, x1 = gx - a.ox1;
          if (y1 >= 0 && y1 < a.H1 && x1 >= 0 && x1 < a.W1)
            v = *reinterpret_cast<const f32x4*>(a.x1 + (((size_t)b * a.H1 + y1) * a.W1 + x1) * a.C1 + (ci0 - a.C0) + 4 * c4);
        }
      }
      *reinterpret_cast<f32x4*>(Xs + pix * WG_T + 4 * c4) = v;
    }
    for (int tap = 0; tap < (MODE == 0 ? 1 : TAPS); ++tap) {
      if (MODE == 1 && tap > 0) __syncthreads();
      // stage dz: 128 pixels x 64 channels (MODE 1: the tap's strided view of the upsampled gradient)
      for (int idx = tid; idx < WG_PIX * (WG_T / 4); idx += 256) {
        const int pix = idx / (WG_T / 4), c4 = idx % (WG_T / 4);
        const int gy = y0 + pix / WG_PW, gx = x0p + pix % WG_PW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gy < a.H && gx < a.W) {
          if (MODE == 0)
            v = *reinterpret_cast<const f32x4*>(a.dz + (((size_t)b * a.H + gy) * a.W + gx) * a.Cout + co0 + 4 * c4);
          else
            v = *reinterpret_cast<const f32x4*>(a.dz + (((size_t)b * (2 * a.H) + 2 * gy + (tap >> 1)) * (2 * a.W) + 2 * gx + (tap & 1)) * a.Cout + co0 + 4 * c4);
        }
        *reinterpret_cast<f32x4*>(Ds + pix * WG_T + 4 * c4) = v;
      }
      __syncthreads();
      const float* Ap = Ds + cot * 32 + li;
      const float* Bp = Xs + cit * 32 + li;
#pragma unroll 4
      for (int k0 = 0; k0 < WG_PIX; k0 += 2) {
        const int m = k0 + lh;
        const float av = Ap[m * WG_T];
        const int hb = ((m / WG_PW) * HPW + (m % WG_PW)) * WG_T;
        if (MODE == 0) {
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const float bv = Bp[hb + ((t / 3) * HPW + (t % 3)) * WG_T];
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
          }
        } else {
          const float bv = Bp[hb];
#pragma unroll
          for (int t = 0; t < TAPS; ++t)
            if (t == tap) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
        }
      }
    }
  }
  // D[row = co][col = ci]
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int ci = ci0 + cit * 32 + li;
      atomicAdd(a.dw + ((size_t)t * a.Cout + co) * Cin + ci, acc[t][r]);
    }
}

// The same weight gradient with bf16x3 products (precision 1): both operands are split x = hi + lo (bf16) while they are
// staged and each tap's product is dz_lo*x_hi + dz_hi*x_lo + dz_hi*x_hi on v_mfma_f32_32x32x16_bf16 -- 3 matrix
// instructions at 16x the fp32-MFMA rate.  K is the PIXEL index, but NHWC tiles are pixel-major: the fragments (8
// consecutive pixels of one channel per lane) come from gfx950's transposing LDS read ds_read_b64_tr_b16, which hands
// lane i of a 16-lane group column i of a 4-row block -- so the tiles stay in their natural order, and a tap shift is a
// whole-row offset folded into the instruction's immediate.  LDS pixel row = [64 ch hi | 64 ch lo | 64 B pad] = 320 B:
// the four rows of a block land on bank offsets 0 / 64 / 128 / 192 (conflict-free).
typedef __bf16 wg_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef short wg_s16x4 __attribute__((ext_vector_type(4)));
constexpr int WGB_ROW = 320;   // bytes per staged pixel

__device__ __forceinline__ wg_bf16x8 wg_tr_frag(const char* p0, const char* p1) {
  typedef wg_s16x4 __attribute__((address_space(3))) * lds_ptr;
  const wg_s16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0));
  const wg_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p1));
  union { short s[8]; wg_bf16x8 b; } r;
#pragma unroll
  for (int j = 0; j < 4; ++j) { r.s[j] = u[j]; r.s[4 + j] = v[j]; }
  return r.b;
}

template <bool PLAIN>
__device__ __forceinline__ void wg_store_split(char* row, int c4, f32x4 v) {
  wg_bf16x4 hi, lo;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    hi[k] = (__bf16)v[k];
    lo[k] = (__bf16)(v[k] - (float)hi[k]);
  }
  *reinterpret_cast<wg_bf16x4*>(row + 8 * c4) = hi;
  if (!PLAIN) *reinterpret_cast<wg_bf16x4*>(row + 128 + 8 * c4) = lo;
}

// 8 waves: (co half) x (ci half) x (pixel half of a 128-pixel patch, 4x32 or 8x16 for narrow images); one workgroup
// per CU, two waves per SIMD.

// NW = 8: one workgroup per CU on 128-pixel patches; NW = 4: 64-pixel patches, two workgroups per CU whose staging and
// MFMA phases overlap each other.
// PLAIN: one bf16 MFMA per product (precision 2) -- only the hi halves are staged (192-byte pixel rows: 128 B + 64 B pad keep
// the four rows of a transposing block on bank offsets 0 / 192 / 128 / 64).  A weight gradient sums over every pixel of the
// batch, so the 2^-9 rounding of the products averages out (relative L1 2e-3 vs fp32 on one layer, the level of fp32
// autograd's own noise through the BatchNorm backward) and nothing downstream consumes it except the optimiser.
// BF16IN (with PLAIN): dz / x0 / x1 are bf16 tensors that already hold the ACTIVATED values (mfpa_act_to_bf16: the previous layer's
// BatchNorm + ReLU + dropout applied, or a plain cast) -- every (co, ci) tile re-reads the patches of both operands, so for layers
// with many tiles halving the bytes per re-read pays for one cast pass (the kernel was bound by what it pulls from L2: skipping its
// loads returned 24 %); staging is then a 16-byte copy, no split.
template <int MODE, int PW, int NW, bool PLAIN, bool BF16IN = false, bool CI128 = false>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void wgrad_bf16x3_kernel(WgradArgs a) {
  static_assert(!BF16IN || PLAIN, "bf16 operands carry only the hi halves");
  static_assert(!CI128 || (MODE == 1 && BF16IN && NW == 8), "the 128-input-channel tile is a form of ALLTAPS");
  // CI128 (ALLTAPS only): a workgroup owns 64 output x 128 INPUT channels -- waves 2 (co) x 4 (ci), every wave walks all 128 pixels of the patch --
  // so the dz tiles, the larger operand (four taps), are re-read C_in / 128 times instead of C_in / 64: the 64 x 64 form moved 1.25 GB through L2
  // for 64 GFLOP (up1.up) at 5.7 TB/s
  constexpr int CIW = CI128 ? 128 : WG_T;                              // input channels per workgroup
  constexpr int ROWX = CI128 ? 320 : (PLAIN ? 192 : WGB_ROW);          // bytes per staged x pixel (256 B + 64 B pad: rows on bank offsets 0 / 64 / 128 / 192)
  constexpr int ROW = PLAIN ? 192 : WGB_ROW;
  constexpr int WGB_THREADS = 64 * NW, WGB_PIX = 16 * NW;
  constexpr int WGB_PH = WGB_PIX / PW;
  constexpr int HALO = (MODE == 0) ? 1 : 0;
  constexpr int TAPS = (MODE == 0) ? 9 : 4;
  constexpr int HPW = PW + 2 * HALO, HPH = WGB_PH + 2 * HALO, HP = HPW * HPH;
  constexpr int X_F4 = (HP * (WG_T / 4) + WGB_THREADS - 1) / WGB_THREADS;        // float4 loads per thread for the xin tile
  constexpr int D_F4 = WGB_PIX * (WG_T / 4) / WGB_THREADS;            // ... and for the dz tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // ALLTAPS (the transposed convolution's weight gradient from bf16 operands, round 5): the four taps' strided views of dz are staged TOGETHER
  // -- four dz tiles in LDS, requested with the next patch's x tile under the current patch's MFMAs -- instead of one after the other, each
  // behind its own exposed global round trip and two barriers for 8 MFMAs of work (0.085 MFMA-busy, 1.2 ms per train step)
  constexpr bool ALLTAPS = MODE == 1 && BF16IN;
  constexpr int DT = ALLTAPS ? 4 : 1;
  char* Ds = smem;                          // [DT][128 px][ROW]  dz tile(s)
  char* Xs = smem + DT * WGB_PIX * ROW;  // [HP px][ROWX]  xin tile with halo

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int cot = wave & 1, cit = CI128 ? (wave >> 1) : ((wave >> 1) & 1), ph = CI128 ? 0 : (wave >> 2);   // ph: the patch's upper / lower 64 pixels (NW = 8)
  // (co tile, ci tile, patch group) of this workgroup.  Every tile of one patch group reads the same dz / x patches; consecutive
  // workgroup ids are dealt round-robin over the 8 XCDs (one L2 each), so with the plain order every XCD fetched every patch.
  // When the grid size is a multiple of 8, XCD k owns a contiguous range of the (group, tile) order instead, tiles fastest.
  unsigned bxi = blockIdx.x, byi = blockIdx.y, bzi = blockIdx.z;
  {
    const unsigned tiles = gridDim.x * gridDim.y, total = tiles * gridDim.z;
    if (a.xcd && total % 8 == 0) {
      const unsigned hw = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      const unsigned lin = (hw % 8) * (total / 8) + hw / 8;
      const unsigned tile = lin % tiles;
      bzi = lin / tiles; bxi = tile % gridDim.x; byi = tile / gridDim.x;
    }
  }
  const int co0 = bxi * WG_T, ci0 = byi * CIW;
  const int Cin = a.C0 + a.C1;
  const bool from0 = ci0 < a.C0;
  const bool affine = from0 && a.in_scale0 != nullptr;
  const long long npatch = (long long)a.B * a.tiles_x * a.tiles_y;
  const int c4 = tid % (WG_T / 4);          // this thread's channel quad (the same for every staged pixel: 512 % 16 == 0)
  // transposing read: lane 4q+p of a 16-lane group addresses row q (pixel), columns 4p..4p+3 (channels) of its block
  const int gl = lane & 15, tq = gl >> 2, tp = gl & 3, gsel = (lane >> 4) & 1;
  const char* a_lane = Ds + (64 * ph + 8 * lh + tq) * ROW + (32 * cot + 16 * gsel + 4 * tp) * 2;
  const char* b_lane = Xs + ((64 / PW) * ph * HPW + 8 * lh + tq) * ROWX + (32 * cit + 16 * gsel + 4 * tp) * 2;

  f32x4 a_sc = {1.f, 1.f, 1.f, 1.f}, a_sh = {0.f, 0.f, 0.f, 0.f};
  if (affine) {
    a_sc = *reinterpret_cast<const f32x4*>(a.in_scale0 + ci0 + 4 * c4);
    a_sh = *reinterpret_cast<const f32x4*>(a.in_shift0 + ci0 + 4 * c4);
  }

  floatx16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // Software pipeline over patches: the global loads of patch n+1 are issued (into registers, no dependent use) before
  // the MFMA block of patch n and are split / written to LDS after it.
  f32x4 xr[X_F4], dr[D_F4];
  auto decode = [&](long long patch, int& b, int& y0, int& x0p) __attribute__((always_inline)) {
    long long q = patch;
    const int tx = (int)(q % a.tiles_x); q /= a.tiles_x;
    const int ty = (int)(q % a.tiles_y);
    b = (int)(q / a.tiles_y);
    y0 = ty * WGB_PH; x0p = tx * PW;
  };
  // A thread's staging slots map to fixed pixels of the patch (pixel = tid / 16 + it * THREADS / 16, channel quad c4): a load is a
  // per-patch SCALAR base plus an element offset that depends only on the in-patch (row, column) -- two integer multiply-adds and
  // four compares for the image border, re-derived from compile-time divisors (no register arrays next to the 144 accumulators)
  // -- instead of a division / 64-bit index chain per load (the loads' address work was the largest single cost of this kernel:
  // skipping the next patch's loads returned 24 %).  Loads are unconditional: a pixel outside the image reads the element at offset
  // 0 of its channel quad and is zeroed when it is staged, so there are no exec-mask branches.
  const int t16 = tid / (WG_T / 4);
  constexpr int PIX_STEP = WGB_THREADS / (WG_T / 4);
  auto x_inside = [&](int pix, int y0, int x0p) __attribute__((always_inline)) {            // inside the image (source 0's extent)?
    const int gy = y0 + pix / HPW - HALO, gx = x0p + pix % HPW - HALO;
    return pix < HP && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
  };
  auto x1_inside = [&](int pix, int y0, int x0p) __attribute__((always_inline)) {           // ... and inside the zero-padded second source?
    const int y1 = y0 + pix / HPW - HALO - a.oy1, x1 = x0p + pix % HPW - HALO - a.ox1;
    return pix < HP && y1 >= 0 && y1 < a.H1 && x1 >= 0 && x1 < a.W1;
  };
  auto load_x = [&](int b, int y0, int x0p) __attribute__((always_inline)) {
    // a slot outside the image loads its nearest image pixel (a line its neighbours fetch anyway; one fixed address for all of them
    // would be a hot spot) and is zeroed when it is staged
    if (from0) {
      const float* base = a.x0 + (size_t)b * a.H * a.W * a.C0 + ci0 + 4 * c4;
#pragma unroll
      for (int it = 0; it < X_F4; ++it) {
        const int pix = t16 + it * PIX_STEP;
        const int gy = min(max(y0 + pix / HPW - HALO, 0), a.H - 1), gx = min(max(x0p + pix % HPW - HALO, 0), a.W - 1);
        xr[it] = *reinterpret_cast<const f32x4*>(base + (gy * a.W + gx) * a.C0);
      }
    } else {
      const float* base = a.x1 + (size_t)b * a.H1 * a.W1 * a.C1 + (ci0 - a.C0) + 4 * c4;
#pragma unroll
      for (int it = 0; it < X_F4; ++it) {
        const int pix = t16 + it * PIX_STEP;
        const int y1 = min(max(y0 + pix / HPW - HALO - a.oy1, 0), a.H1 - 1), x1 = min(max(x0p + pix % HPW - HALO - a.ox1, 0), a.W1 - 1);
        xr[it] = *reinterpret_cast<const f32x4*>(base + (y1 * a.W1 + x1) * a.C1);
      }
    }
  };
  auto store_x = [&](int b, int y0, int x0p) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < X_F4; ++it) {
      const int pix = t16 + it * PIX_STEP;
      if (pix < HP) {
        const bool inside = from0 ? x_inside(pix, y0, x0p) : x1_inside(pix, y0, x0p);
        f32x4 v = xr[it];
        if (!inside) v = f32x4{0.f, 0.f, 0.f, 0.f};               // zero padding
        if (affine && inside) {                                    // padding stays exactly zero
          v = v * a_sc + a_sh;
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
          if (a.drop_thresh) {
            const int gy = y0 + pix / HPW - HALO, gx = x0p + pix % HPW - HALO;
            const unsigned long long e0 = (((unsigned long long)b * a.H + gy) * a.W + gx) * a.C0 + ci0 + 4 * c4;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = mfpa_keep(a.drop_seed, a.drop_thresh, e0 + k) ? v[k] * a.drop_scale : 0.f;
          }
        }
        wg_store_split<PLAIN>(Xs + pix * ROW, c4, v);
      }
    }
  };
  // dz tile: zero past the image border is applied when the tile is staged (store_d keeps the patch position for it)
  int d_y0 = 0, d_x0 = 0;
  auto load_d = [&](int b, int y0, int x0p, int tap) __attribute__((always_inline)) {
    const float* base = (MODE == 0) ? a.dz + (size_t)b * a.H * a.W * a.Cout + co0 + 4 * c4
                                    : a.dz + (size_t)b * (2 * a.H) * (2 * a.W) * a.Cout + co0 + 4 * c4;
#pragma unroll
    for (int it = 0; it < D_F4; ++it) {
      const int pix = t16 + it * PIX_STEP;
      const int gy = min(y0 + pix / PW, a.H - 1), gx = min(x0p + pix % PW, a.W - 1);       // clamped; zeroed in store_d when outside
      const int off = (MODE == 0) ? (gy * a.W + gx) * a.Cout : ((2 * gy + (tap >> 1)) * (2 * a.W) + 2 * gx + (tap & 1)) * a.Cout;
      dr[it] = *reinterpret_cast<const f32x4*>(base + off);
    }
    d_y0 = y0; d_x0 = x0p;
  };
  auto store_d = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < D_F4; ++it) {
      const int pix = t16 + it * PIX_STEP;
      const bool in = d_y0 + pix / PW < a.H && d_x0 + pix % PW < a.W;
      wg_store_split<PLAIN>(Ds + pix * ROW, c4, in ? dr[it] : f32x4{0.f, 0.f, 0.f, 0.f});
    }
  };
  // ---- BF16IN staging: 8 threads per pixel, one 16-byte piece (8 channels) each
  typedef unsigned int wg_u32x4 __attribute__((ext_vector_type(4)));
  constexpr int X_L = BF16IN ? (HP * 8 + WGB_THREADS - 1) / WGB_THREADS : 1;
  constexpr int D_L = BF16IN ? WGB_PIX * 8 / WGB_THREADS : 1;
  constexpr int PIX_STEP8 = WGB_THREADS / 8;
  const int c8 = tid % 8, t8 = tid / 8;
  wg_u32x4 xr16[X_L], dr16[D_L];
  auto load_x16 = [&](int b, int y0, int x0p) __attribute__((always_inline)) {
    if (from0) {
      const char* base = reinterpret_cast<const char*>(a.x0) + ((size_t)b * a.H * a.W * a.C0 + ci0 + 8 * c8) * 2;
#pragma unroll
      for (int it = 0; it < X_L; ++it) {
        const int pix = t8 + it * PIX_STEP8;
        const int gy = min(max(y0 + pix / HPW - HALO, 0), a.H - 1), gx = min(max(x0p + pix % HPW - HALO, 0), a.W - 1);
        xr16[it] = *reinterpret_cast<const wg_u32x4*>(base + (size_t)((gy * a.W + gx) * a.C0) * 2);
      }
    } else {
      const char* base = reinterpret_cast<const char*>(a.x1) + ((size_t)b * a.H1 * a.W1 * a.C1 + (ci0 - a.C0) + 8 * c8) * 2;
#pragma unroll
      for (int it = 0; it < X_L; ++it) {
        const int pix = t8 + it * PIX_STEP8;
        const int y1 = min(max(y0 + pix / HPW - HALO - a.oy1, 0), a.H1 - 1), x1 = min(max(x0p + pix % HPW - HALO - a.ox1, 0), a.W1 - 1);
        xr16[it] = *reinterpret_cast<const wg_u32x4*>(base + (size_t)((y1 * a.W1 + x1) * a.C1) * 2);
      }
    }
  };
  auto store_x16 = [&](int y0, int x0p) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < X_L; ++it) {
      const int pix = t8 + it * PIX_STEP8;
      if (pix < HP) {
        const bool inside = from0 ? x_inside(pix, y0, x0p) : x1_inside(pix, y0, x0p);
        *reinterpret_cast<wg_u32x4*>(Xs + pix * ROW + 16 * c8) = inside ? xr16[it] : wg_u32x4{0u, 0u, 0u, 0u};
      }
    }
  };
  auto load_d16 = [&](int b, int y0, int x0p, int tap) __attribute__((always_inline)) {
    const char* base = reinterpret_cast<const char*>(a.dz) +
                       ((size_t)b * (MODE == 0 ? 1 : 4) * a.H * a.W * a.Cout + co0 + 8 * c8) * 2;
#pragma unroll
    for (int it = 0; it < D_L; ++it) {
      const int pix = t8 + it * PIX_STEP8;
      const int gy = min(y0 + pix / PW, a.H - 1), gx = min(x0p + pix % PW, a.W - 1);
      const int off = (MODE == 0) ? (gy * a.W + gx) * a.Cout : ((2 * gy + (tap >> 1)) * (2 * a.W) + 2 * gx + (tap & 1)) * a.Cout;
      dr16[it] = *reinterpret_cast<const wg_u32x4*>(base + (size_t)off * 2);
    }
    d_y0 = y0; d_x0 = x0p;
  };
  auto store_d16 = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < D_L; ++it) {
      const int pix = t8 + it * PIX_STEP8;
      const bool in = d_y0 + pix / PW < a.H && d_x0 + pix % PW < a.W;
      *reinterpret_cast<wg_u32x4*>(Ds + pix * ROW + 16 * c8) = in ? dr16[it] : wg_u32x4{0u, 0u, 0u, 0u};
    }
  };
  // CI128: the x tile is 128 pixels x 16 pieces of 16 bytes (no halo in mode 1): four pieces per thread
  constexpr int X_L2 = CI128 ? WGB_PIX * 16 / WGB_THREADS : 1;
  const int c16 = tid % 16, t16b = tid / 16;
  wg_u32x4 xr16w[X_L2];
  auto load_x16w = [&](int b, int y0, int x0p) __attribute__((always_inline)) {
    const char* base = reinterpret_cast<const char*>(a.x0) + ((size_t)b * a.H * a.W * a.C0 + ci0 + 8 * c16) * 2;
#pragma unroll
    for (int it = 0; it < X_L2; ++it) {
      const int pix = t16b + it * (WGB_THREADS / 16);
      const int gy = min(y0 + pix / PW, a.H - 1), gx = min(x0p + pix % PW, a.W - 1);
      xr16w[it] = *reinterpret_cast<const wg_u32x4*>(base + (size_t)((gy * a.W + gx) * a.C0) * 2);
    }
  };
  auto store_x16w = [&](int y0, int x0p) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < X_L2; ++it) {
      const int pix = t16b + it * (WGB_THREADS / 16);
      const bool inside = y0 + pix / PW < a.H && x0p + pix % PW < a.W;
      *reinterpret_cast<wg_u32x4*>(Xs + pix * ROWX + 16 * c16) = inside ? xr16w[it] : wg_u32x4{0u, 0u, 0u, 0u};
    }
  };
  // ALLTAPS: the four taps' tiles at once
  wg_u32x4 dr16q[ALLTAPS ? 4 : 1][D_L];
  auto load_d16_all = [&](int b, int y0, int x0p) __attribute__((always_inline)) {
    const char* base = reinterpret_cast<const char*>(a.dz) + ((size_t)b * 4 * a.H * a.W * a.Cout + co0 + 8 * c8) * 2;
#pragma unroll
    for (int tap = 0; tap < (ALLTAPS ? 4 : 1); ++tap)
#pragma unroll
      for (int it = 0; it < D_L; ++it) {
        const int pix = t8 + it * PIX_STEP8;
        const int gy = min(y0 + pix / PW, a.H - 1), gx = min(x0p + pix % PW, a.W - 1);
        const int off = ((2 * gy + (tap >> 1)) * (2 * a.W) + 2 * gx + (tap & 1)) * a.Cout;
        dr16q[tap][it] = *reinterpret_cast<const wg_u32x4*>(base + (size_t)off * 2);
      }
    d_y0 = y0; d_x0 = x0p;
  };
  auto store_d16_all = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int tap = 0; tap < (ALLTAPS ? 4 : 1); ++tap)
#pragma unroll
      for (int it = 0; it < D_L; ++it) {
        const int pix = t8 + it * PIX_STEP8;
        const bool in = d_y0 + pix / PW < a.H && d_x0 + pix % PW < a.W;
        *reinterpret_cast<wg_u32x4*>(Ds + (tap * WGB_PIX + pix) * ROW + 16 * c8) = in ? dr16q[tap][it] : wg_u32x4{0u, 0u, 0u, 0u};
      }
  };
  // one set of names for the patch loop below
  auto LOAD_X = [&](int b, int y0, int x0p) __attribute__((always_inline)) { if constexpr (BF16IN) load_x16(b, y0, x0p); else load_x(b, y0, x0p); };
  auto STORE_X = [&](int b, int y0, int x0p) __attribute__((always_inline)) { if constexpr (BF16IN) store_x16(y0, x0p); else store_x(b, y0, x0p); };
  auto LOAD_D = [&](int b, int y0, int x0p, int tap) __attribute__((always_inline)) { if constexpr (BF16IN) load_d16(b, y0, x0p, tap); else load_d(b, y0, x0p, tap); };
  auto STORE_D = [&]() __attribute__((always_inline)) { if constexpr (BF16IN) store_d16(); else store_d(); };
  auto mfma3 = [&](floatx16& c, wg_bf16x8 ah, wg_bf16x8 al, wg_bf16x8 bh, wg_bf16x8 bl) __attribute__((always_inline)) {
    if (!PLAIN) {
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    }
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
  };

  int b = 0, y0 = 0, x0p = 0;
  long long patch = bzi;
  if constexpr (ALLTAPS) {
    auto LOAD_XA = [&](int b_, int y_, int x_) __attribute__((always_inline)) { if constexpr (CI128) load_x16w(b_, y_, x_); else LOAD_X(b_, y_, x_); };
    auto STORE_XA = [&](int b_, int y_, int x_) __attribute__((always_inline)) { if constexpr (CI128) store_x16w(y_, x_); else STORE_X(b_, y_, x_); };
    if (patch < npatch) {
      decode(patch, b, y0, x0p);
      LOAD_XA(b, y0, x0p);
      load_d16_all(b, y0, x0p);
    }
    for (; patch < npatch; patch += gridDim.z) {
      __syncthreads();                     // the previous patch's fragment reads are done
      STORE_XA(b, y0, x0p);
      store_d16_all();
      __syncthreads();
      const long long next = patch + gridDim.z;
      int nb = 0, ny0 = 0, nx0 = 0;
      if (next < npatch) {                 // the next patch's five tiles: their loads land during the MFMA block below
        decode(next, nb, ny0, nx0);
        LOAD_XA(nb, ny0, nx0);
        load_d16_all(nb, ny0, nx0);
      }
#pragma unroll 2
      for (int ks = 0; ks < (CI128 ? WGB_PIX : WG_PIX) / 16; ++ks) {
        const char* bp = b_lane + (((16 * ks) / PW) * HPW + (16 * ks) % PW) * ROWX;
        const wg_bf16x8 bh = wg_tr_frag(bp, bp + 4 * ROWX);          // the x fragment serves all four taps
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const char* ap = a_lane + (t * WGB_PIX + 16 * ks) * ROW;
          const wg_bf16x8 ah = wg_tr_frag(ap, ap + 4 * ROW);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
        }
      }
      b = nb; y0 = ny0; x0p = nx0;
    }
  } else
  if (patch < npatch) {
    decode(patch, b, y0, x0p);
    LOAD_X(b, y0, x0p);
    LOAD_D(b, y0, x0p, 0);
  }
  for (; !ALLTAPS && patch < npatch; patch += gridDim.z) {
    __syncthreads();                       // the previous patch's fragment reads are done
    STORE_X(b, y0, x0p);
    STORE_D();
    __syncthreads();
    const long long next = patch + gridDim.z;
    int nb = 0, ny0 = 0, nx0 = 0;
    if (MODE == 0 && next < npatch) {      // prefetch the next patch; the loads land during the MFMA block below
      decode(next, nb, ny0, nx0);
      LOAD_X(nb, ny0, nx0);
      LOAD_D(nb, ny0, nx0, 0);
    }
    for (int tap = 0; tap < (MODE == 0 ? 1 : TAPS); ++tap) {
      if (MODE == 1 && tap > 0) {          // transposed conv: the tap's strided view of dz replaces the dz tile
        __syncthreads();
        LOAD_D(b, y0, x0p, tap);
        STORE_D();
        __syncthreads();
      }
#pragma unroll 1   // one k-step's 9 taps in flight at a time: bounds the live fragment registers next to 144 accumulators
      for (int ks = 0; ks < WG_PIX / 16; ++ks) {
        // this lane's 8 pixels: k = 16 ks + 8 lh + (0..7) of the wave's 64: patch row 16 ks / PW, columns 16 ks % PW + 8 lh + (0..7)
        const char* ap = a_lane + (16 * ks) * ROW;
        const wg_bf16x8 ah = wg_tr_frag(ap, ap + 4 * ROW);
        const wg_bf16x8 al = PLAIN ? ah : wg_tr_frag(ap + 128, ap + 4 * ROW + 128);
        if (MODE == 0) {
          const char* bk = b_lane + (((16 * ks) / PW) * HPW + (16 * ks) % PW) * ROW;
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const char* bp = bk + ((t / 3) * HPW + (t % 3)) * ROW;
            const wg_bf16x8 bh = wg_tr_frag(bp, bp + 4 * ROW);
            mfma3(acc[t], ah, al, bh, PLAIN ? bh : wg_tr_frag(bp + 128, bp + 4 * ROW + 128));
          }
        } else {
          const char* bp = b_lane + (((16 * ks) / PW) * HPW + (16 * ks) % PW) * ROW;
          const wg_bf16x8 bh = wg_tr_frag(bp, bp + 4 * ROW);
          const wg_bf16x8 bl = PLAIN ? bh : wg_tr_frag(bp + 128, bp + 4 * ROW + 128);
#pragma unroll
          for (int t = 0; t < TAPS; ++t)
            if (t == tap) mfma3(acc[t], ah, al, bh, bl);
        }
      }
    }
    if (MODE == 1 && next < npatch) {
      decode(next, nb, ny0, nx0);
      LOAD_X(nb, ny0, nx0);
      LOAD_D(nb, ny0, nx0, 0);
    }
    b = nb; y0 = ny0; x0p = nx0;
  }
  // D[row = co][col = ci]
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int ci = ci0 + cit * 32 + li;
      atomicAdd(a.dw + ((size_t)t * a.Cout + co) * Cin + ci, acc[t][r]);
    }
}

// 3x3 weight gradient from bf16 operands, one bf16 MFMA per product ("V2" of the PLAIN + BF16IN form above; precision 3, mode 0).
// What the skip experiments on that form said (tools/exp_wgrad.py, 512 -> 512 @ 32 x 31, 64 clips, 453 us): fragment reads + MFMAs alone
// 220 us, the staging chain alone 176 us, the two together 369 -- the staging sat in front of the MFMA block as a bubble all eight
// waves share (they are in lockstep behind the barriers) -- and the final atomics 130 us of every launch.  Here:
//   * a workgroup owns 32 COT output channels x 64 input channels x 9 taps; with COT = 4 (C_out % 128 == 0) the eight waves are 4 x 2
//     tiles and every wave walks all 128 pixels of a patch -- no two waves hold the same (co, ci) tile, so half the atomics and half the
//     operand bytes per MFMA of the 64 x 64 form (COT = 2: two pixel halves, as before);
//   * two LDS stages, ONE barrier per patch: the staging registers (patch n + 1, requested most of an iteration ago) are written to the
//     other stage between the MFMAs of k-step 0, and take patch n + 2 between the MFMAs of k-step 1 (sched_group_barrier-pinned; the
//     staging code is branch-free so that it can sit inside the MFMA block);
//   * the three dx taps of a halo row share their transposing reads: a lane's pixels P .. P+9 of one x column come from THREE
//     ds_read_b64_tr_b16 (r0..r4 = pixel pairs); tap dx = 0 is r0..r3, dx = 2 is r1..r4 (no instruction), dx = 1 four v_alignbit --
//     11 reads per k-step instead of 20, 22 fragment registers instead of 80, so the next k-step's reads run under this one's MFMAs;
//   * one workgroup per CU and as few patch groups as fill the chip once or twice: the atomics are per workgroup.
template <int PW, int COT>
__global__ __launch_bounds__(512, 1) void wgrad_bf16_kernel(WgradArgs a) {
  constexpr int THREADS = 512, PIX = 128, PH_ = PIX / PW, TAPS = 9;
  constexpr int PHS = 4 / COT;                                         // pixel halves (waves that share a (co, ci) tile)
  constexpr int KS = PIX / PHS / 16;                                   // k-steps per wave and patch
  constexpr int COW = 32 * COT;                                        // output channels per workgroup
  constexpr int ROW_D = COW * 2 + 64, ROW_X = 192;                     // staged pixel rows: data + 64 B (the four rows of a transposing block
                                                                       // land on bank offsets 0 / 64 / 128 / 192)
  constexpr int HPW = PW + 2, HPH = PH_ + 2, HP = HPW * HPH;
  constexpr int XROWS = HP + 4;                                        // the third read of a row block runs two pixels past the halo tile
  constexpr int STAGE = PIX * ROW_D + XROWS * ROW_X;
  constexpr int DPP = COW / 8;                                         // 16-byte pieces per dz pixel
  constexpr int D_L = PIX * DPP / THREADS, X_L = (HP * 8 + THREADS - 1) / THREADS;
  static_assert(KS % 2 == 0 && KS >= 4, "fragment sets alternate; k-steps 0 and 1 carry the staging");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int cot = wave % COT, cit = (wave / COT) & 1, ph = wave / (2 * COT);
  unsigned bxi = blockIdx.x, byi = blockIdx.y, bzi = blockIdx.z;
  {                                                                     // XCD k owns a contiguous range of the (patch group, tile) order
    const unsigned tiles = gridDim.x * gridDim.y, total = tiles * gridDim.z;
    if (a.xcd && total % 8 == 0) {
      const unsigned hw = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      const unsigned lin = (hw % 8) * (total / 8) + hw / 8;
      const unsigned tile = lin % tiles;
      bzi = lin / tiles; bxi = tile % gridDim.x; byi = tile / gridDim.x;
    }
  }
  const int co0 = bxi * COW, ci0 = byi * WG_T;
  const int Cin = a.C0 + a.C1;
  const bool from0 = ci0 < a.C0;
  const long long npatch = (long long)a.B * a.tiles_x * a.tiles_y;
  const int gl = lane & 15, tq = gl >> 2, tp = gl & 3, gsel = (lane >> 4) & 1;
  const char* a_lane = smem + ((PIX / PHS) * ph + 8 * lh + tq) * ROW_D + (32 * cot + 16 * gsel + 4 * tp) * 2;
  const char* b_lane = smem + PIX * ROW_D + ((PIX / PHS / PW) * ph * HPW + 8 * lh + tq) * ROW_X + (32 * cit + 16 * gsel + 4 * tp) * 2;

  floatx16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // ---- staging: one code path for both sources of the input (extent, channel count and zero-pad offset of the one this tile reads)
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int sH = from0 ? a.H : a.H1, sW = from0 ? a.W : a.W1, sC = from0 ? a.C0 : a.C1, soy = from0 ? 0 : a.oy1, sox = from0 ? 0 : a.ox1;
  const int c8 = tid % 8, t8 = tid / 8, cd = tid % DPP, td = tid / DPP;
  const char* xsrc = (from0 ? reinterpret_cast<const char*>(a.x0) + (size_t)(ci0 + 8 * c8) * 2
                            : reinterpret_cast<const char*>(a.x1) + (size_t)(ci0 - a.C0 + 8 * c8) * 2);
  const char* dsrc = reinterpret_cast<const char*>(a.dz) + (size_t)(co0 + 8 * cd) * 2;
  u32x4 xr[X_L], dr[D_L];
  int pb = 0, py = 0, px = 0;                                           // patch in the staging registers
  auto decode = [&](long long patch) __attribute__((always_inline)) {
    long long q = patch;
    const int tx = (int)(q % a.tiles_x); q /= a.tiles_x;
    const int ty = (int)(q % a.tiles_y);
    pb = (int)(q / a.tiles_y); py = ty * PH_; px = tx * PW;
  };
  auto load = [&]() __attribute__((always_inline)) {                    // clamped addresses; what lies outside is zeroed by stage()
    const char* xb = xsrc + (size_t)pb * sH * sW * sC * 2;
#pragma unroll
    for (int it = 0; it < X_L; ++it) {
      const int pix = t8 + it * (THREADS / 8);
      const int gy = min(max(py + pix / HPW - 1 - soy, 0), sH - 1), gx = min(max(px + pix % HPW - 1 - sox, 0), sW - 1);
      xr[it] = *reinterpret_cast<const u32x4*>(xb + (size_t)((gy * sW + gx) * sC) * 2);
    }
    const char* db = dsrc + (size_t)pb * a.H * a.W * a.Cout * 2;
#pragma unroll
    for (int it = 0; it < D_L; ++it) {
      const int pix = td + it * (THREADS / DPP);
      const int gy = min(py + pix / PW, a.H - 1), gx = min(px + pix % PW, a.W - 1);
      dr[it] = *reinterpret_cast<const u32x4*>(db + (size_t)((gy * a.W + gx) * a.Cout) * 2);
    }
  };
  auto stage = [&](int soff) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < X_L; ++it) {
      const int pix = t8 + it * (THREADS / 8);
      const unsigned sy = (unsigned)(py + pix / HPW - 1 - soy), sx = (unsigned)(px + pix % HPW - 1 - sox);
      const bool inside = (int)(pix < HP) & (int)(sy < (unsigned)sH) & (int)(sx < (unsigned)sW);
      // slots past the tile write zeros into its last slack row
      *reinterpret_cast<u32x4*>(smem + soff + PIX * ROW_D + min(pix, XROWS - 1) * ROW_X + 16 * c8) = inside ? xr[it] : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int it = 0; it < D_L; ++it) {
      const int pix = td + it * (THREADS / DPP);
      const bool in = (int)(py + pix / PW < a.H) & (int)(px + pix % PW < a.W);
      *reinterpret_cast<u32x4*>(smem + soff + pix * ROW_D + 16 * cd) = in ? dr[it] : u32x4{0u, 0u, 0u, 0u};
    }
  };
  // ---- fragments
  typedef wg_s16x4 __attribute__((address_space(3))) * lds_ptr;
  struct Fr { wg_bf16x8 a; unsigned r[3][5]; };
  auto read_fr = [&](Fr& f, int soff, int ks) __attribute__((always_inline)) {
    const char* ap = a_lane + soff + (16 * ks) * ROW_D;
    f.a = wg_tr_frag(ap, ap + 4 * ROW_D);
    const char* bk = b_lane + soff + (((16 * ks) / PW) * HPW + (16 * ks) % PW) * ROW_X;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const char* bp = bk + dy * HPW * ROW_X;
      union { wg_s16x4 s; unsigned u[2]; } u0, u1, u2;
      u0.s = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(bp));
      u1.s = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(bp + 4 * ROW_X));
      u2.s = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(bp + 8 * ROW_X));
      f.r[dy][0] = u0.u[0]; f.r[dy][1] = u0.u[1]; f.r[dy][2] = u1.u[0]; f.r[dy][3] = u1.u[1]; f.r[dy][4] = u2.u[0];
    }
  };
  auto mfma9 = [&](const Fr& f) __attribute__((always_inline)) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      union { unsigned u[4]; wg_bf16x8 b; } f0, f1, f2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f0.u[j] = f.r[dy][j];
        f1.u[j] = __builtin_amdgcn_alignbit(f.r[dy][j + 1], f.r[dy][j], 16);
        f2.u[j] = f.r[dy][j + 1];
      }
      acc[3 * dy + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a, f0.b, acc[3 * dy + 0], 0, 0, 0);
      acc[3 * dy + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a, f1.b, acc[3 * dy + 1], 0, 0, 0);
      acc[3 * dy + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a, f2.b, acc[3 * dy + 2], 0, 0, 0);
    }
  };

  long long patch = bzi;
  if (patch < npatch) {
    decode(patch);
    load();
    stage(0);
    decode(patch + gridDim.z < npatch ? patch + gridDim.z : patch);     // past the end: the same patch again, never used
    load();
  }
  __syncthreads();
  int sel = 0;
  Fr fa, fb;
  for (; patch < npatch; patch += gridDim.z) {
    const int soff = sel * STAGE, noff = (sel ^ 1) * STAGE;
    read_fr(fa, soff, 0);
    __builtin_amdgcn_sched_barrier(0);
    // k-step 0: MFMAs || k-step 1's reads, the staging registers -> the other stage
    read_fr(fb, soff, 1);
    stage(noff);
    mfma9(fa);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
      if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // k-step 1: MFMAs || k-step 2's reads, the staging registers <- patch n + 2
    read_fr(fa, soff, 2);
    {
      const long long n2 = patch + 2 * (long long)gridDim.z;
      decode(n2 < npatch ? n2 : patch);
      load();
    }
    mfma9(fb);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 2; ks < KS; ks += 2) {
      read_fr(fb, soff, ks + 1);
      mfma9(fa);
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 2 < KS) {
        read_fr(fa, soff, ks + 2);
        mfma9(fb);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      } else {
        mfma9(fb);
      }
    }
    __syncthreads();
    sel ^= 1;
  }
  // D[row = co][col = ci]
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int ci = ci0 + cit * 32 + li;
      atomicAdd(a.dw + ((size_t)t * a.Cout + co) * Cin + ci, acc[t][r]);
    }
}

// First layer weight gradient (1 input channel): dW[tap][co] += sum_p dz[p][co] * x[p + tap].
__global__ __launch_bounds__(256) void wgrad_c1_kernel(const float* __restrict__ dz, const float* __restrict__ x32,
                                                       const double* __restrict__ spec64,
                                                       const double* __restrict__ denom, int B, int H, int W, int Cout,
                                                       float* __restrict__ dw, int dz16) {
  // a workgroup walks image rows (b, gy) and a row's pixels 256 / lanes at a time: 32-bit index arithmetic, one division per ROW (the first
  // form divided a 64-bit pixel index three times per pixel: 441 us for 1.06 GB); dz16: dz is the bfloat16 copy the BatchNorm backward wrote
  const int lanes = Cout / 4, rows = 256 / lanes;
  const int cq = threadIdx.x % lanes, prow = threadIdx.x / lanes;
  float acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[t][k] = 0.f;
  const int nrows = B * H;
  for (int r = blockIdx.x; r < nrows; r += gridDim.x)
  for (int gx = prow; gx < W + rows - 1 - (W + rows - 1) % rows; gx += rows) {      // every lane runs every trip: the shuffles below need whole pixel groups
    const int b = r / H, gy = r % H;
    const bool live = gx < W;
    const size_t p = (size_t)r * W + (live ? gx : 0);
    f32x4 g = ld_act4(dz, p * Cout + 4 * cq, dz16);
    if (!live) g = f32x4{0.f, 0.f, 0.f, 0.f};
    const double den = (spec64 && denom) ? denom[b] : 1.0;
    if (lanes >= 9) {
      // the lanes of a pixel share its 3x3 input window: lane cq < 9 loads (and normalises: one float64 division) tap cq,
      // the others receive it by shuffle -- like conv3x3_c1_kernel
      float mine = 0.f;
      if (cq < 9) {
        const int yy = gy + cq / 3 - 1, xx = gx + cq % 3 - 1;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
          const size_t o = ((size_t)b * H + yy) * W + xx;
          mine = spec64 ? (float)(spec64[o] / den) : x32[o];
        }
      }
      const int base = (threadIdx.x & 63) - cq;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float v = __shfl(mine, base + t);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[t][k] += g[k] * v;
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
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[t][k] += g[k] * v;
      }
    }
  }
  __shared__ float sh[256 * 36];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) sh[threadIdx.x * 36 + t * 4 + k] = acc[t][k];
  __syncthreads();
  if (prow == 0) {
    for (int r = 1; r < rows; ++r)
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[t][k] += sh[(r * lanes + cq) * 36 + t * 4 + k];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(dw + (size_t)t * Cout + 4 * cq + k, acc[t][k]);
  }
}

// ------------------------------------------------------------------ OutConv (1x1 -> 1 class) in training
// pred[p] = sum_c relu(z[p][c]*scale[c]+shift[c]) * w[c] + bias
__global__ __launch_bounds__(256) void outconv_fwd_kernel(const float* __restrict__ z, long long npix, int C,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ pred, int z16) {
  const int lpp = C / 4, sub = threadIdx.x % lpp, pl = threadIdx.x / lpp, ppb = 256 / lpp;
  const f32x4 wv = *reinterpret_cast<const f32x4*>(w + 4 * sub);
  const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + 4 * sub);
  const f32x4 sf = *reinterpret_cast<const f32x4*>(shift + 4 * sub);
  const float b0 = bias[0];
  const long long iters = (npix + (long long)gridDim.x * ppb - 1) / ((long long)gridDim.x * ppb);
  for (long long it = 0; it < iters; ++it) {
    const long long p = (it * gridDim.x + blockIdx.x) * ppb + pl;
    float s = 0.f;
    if (p < npix) {
      const f32x4 v = ld_act4(z, (size_t)p * C + 4 * sub, z16);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float y = v[k] * sc[k] + sf[k];
        s += (y > 0.f ? y : 0.f) * wv[k];
      }
    }
    for (int o = lpp >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (sub == 0 && p < npix) pred[p] = s + b0;
  }
}

// dy[p][c] = dpred[p]*w[c];  partial[blk][c] = sum_p dpred[p]*y[p][c] (c < C), partial[blk][C] = sum_p dpred[p]
// SUMS: dy is not written (the BatchNorm backward forms it again from dpred and w: bn_bwd_apply_kernel<RANK1>); instead the pass, which holds
// z and dy of every element anyway, also forms the partial sums of that BatchNorm backward -- {sum g, sum g * xhat}, g = dy * [y > 0], xhat =
// (z - mean) * invstd -- as one row of `part` (blocks x 2 x C float, the convolutions' stats_part layout) per workgroup.
template <bool SUMS>
__global__ __launch_bounds__(256) void outconv_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dpred,
                                                          long long npix, int C, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ w,
                                                          float* __restrict__ dy, double* __restrict__ partial,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          float* __restrict__ part, int z16) {
  const int lpp = C / 4, sub = threadIdx.x % lpp, pl = threadIdx.x / lpp, ppb = 256 / lpp;
  const f32x4 wv = *reinterpret_cast<const f32x4*>(w + 4 * sub);
  const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + 4 * sub);
  const f32x4 sf = *reinterpret_cast<const f32x4*>(shift + 4 * sub);
  f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = {0.f, 0.f, 0.f, 0.f};
  if (SUMS) {
    mu = *reinterpret_cast<const f32x4*>(mean + 4 * sub);
    is = *reinterpret_cast<const f32x4*>(invstd + 4 * sub);
  }
  double sw[4] = {0, 0, 0, 0}, sb = 0, s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  for (long long p = (long long)blockIdx.x * ppb + pl; p < npix; p += (long long)gridDim.x * ppb) {
    const float g = dpred[p];
    const f32x4 v = ld_act4(z, (size_t)p * C + 4 * sub, z16);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float y = v[k] * sc[k] + sf[k];
      sw[k] += (double)g * (double)(y > 0.f ? y : 0.f);
      o[k] = g * wv[k];
      if (SUMS) {
        const float gg = y > 0.f ? o[k] : 0.f;
        s0[k] += (double)gg;
        s1[k] += (double)gg * (double)((v[k] - mu[k]) * is[k]);
      }
    }
    if (!SUMS) *reinterpret_cast<f32x4*>(dy + (size_t)p * C + 4 * sub) = o;
    if (sub == 0) sb += (double)g;
  }
  if (SUMS) {
    __shared__ double sh2[256 * 8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sh2[threadIdx.x * 8 + k] = s0[k];
      sh2[threadIdx.x * 8 + 4 + k] = s1[k];
    }
    __syncthreads();
    if (pl == 0) {
      for (int r = 1; r < ppb; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          s0[k] += sh2[(r * lpp + sub) * 8 + k];
          s1[k] += sh2[(r * lpp + sub) * 8 + 4 + k];
        }
      float* row = part + (size_t)blockIdx.x * 2 * C;
      *reinterpret_cast<f32x4*>(row + 4 * sub) = f32x4{(float)s0[0], (float)s0[1], (float)s0[2], (float)s0[3]};
      *reinterpret_cast<f32x4*>(row + C + 4 * sub) = f32x4{(float)s1[0], (float)s1[1], (float)s1[2], (float)s1[3]};
    }
  }
  __shared__ double sh[256 * 5];
#pragma unroll
  for (int k = 0; k < 4; ++k) sh[threadIdx.x * 5 + k] = sw[k];
  sh[threadIdx.x * 5 + 4] = sb;
  __syncthreads();
  if (pl == 0) {
    for (int r = 1; r < ppb; ++r) {
#pragma unroll
      for (int k = 0; k < 4; ++k) sw[k] += sh[(r * lpp + sub) * 5 + k];
      if (sub == 0) sb += sh[(r * lpp) * 5 + 4];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) partial[(size_t)blockIdx.x * (C + 1) + 4 * sub + k] = sw[k];
    if (sub == 0) partial[(size_t)blockIdx.x * (C + 1) + C] = sb;
  }
}

// ------------------------------------------------------------------ L1 loss (mean), float32 pred vs float64 target
__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ pred, const double* __restrict__ target,
                                                 long long n, float* __restrict__ dpred, double* __restrict__ partial) {
  double s = 0;
  const float inv = (float)(1.0 / (double)n);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const double d = (double)pred[i] - target[i];
    s += fabs(d);
    if (dpred) dpred[i] = d > 0 ? inv : (d < 0 ? -inv : 0.f);
  }
  __shared__ double sh[256];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

__global__ void l1_finish_kernel(const double* __restrict__ partial, int nblk, long long n, double* __restrict__ loss) {
  __shared__ double sh[256];
  double s = 0;
  for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = sh[0] / (double)n;
}

// ------------------------------------------------------------------ Adam (torch.optim.Adam, no weight decay, no amsgrad)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n, float lr,
                                                   float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                   float grad_scale) {
  const float step_size = lr / bc1;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gi = g[i] * grad_scale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  }
}

inline int grid_for(long long work, int per_block = 256, int cap = 256 * 16) {
  long long b = (work + per_block - 1) / per_block;
  if (b > cap) b = cap;
  return (int)(b < 1 ? 1 : b);
}

// Convolution weights [tap][Co][Ci] (the master fp32 layout) -> the operand image a conv launch reads: precision 0: [tap][row n][K]
// floats; precision 1: the bf16x3 image of csrc/unet.hip, [tap][chunk of 32 k][row n][128 B = 32 bf16 hi | 32 bf16 lo in 16-byte
// slots XOR-swizzled by the row] (what ops_unet.split_bf16x3 builds on the host).  TR = false: n = co in [row0, row0 + nrows), k = ci (forward).  TR = true: the input-gradient operand --
// tap t reads source tap taps-1-t when taps == 9 (the 3x3 kernel flipped; the 2x2 transposed conv keeps its tap), n = ci in
// [row0, row0 + nrows), k = co.  One workgroup per 32 x 32 (n, k) tile and tap, transposed through LDS so both sides coalesce.
template <bool TR>
__device__ __forceinline__ void pack_conv_weights_tile(float (*tile)[33], const float* __restrict__ w, int taps, int Co, int Ci, int row0,
                                                       int nrows, int precision, float* __restrict__ out, int kx, int ny, int t) {
  const int k0 = kx * 32, n0 = ny * 32;
  const int ts = (TR && taps == 9) ? taps - 1 - t : t;
  const int K = TR ? Co : Ci;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    // source element (n = n0 + .., k = k0 + ..): TR: w[ts][co = k][ci = row0 + n], contiguous along n; else w[ts][co = row0 + n][ci = k]
    if (TR) tile[r][tx] = w[((size_t)ts * Co + (k0 + r)) * Ci + row0 + n0 + tx];          // tile[k][n]
    else tile[r][tx] = w[((size_t)ts * Co + (row0 + n0 + r)) * Ci + k0 + tx];             // tile[n][k]
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const float v = TR ? tile[tx][r] : tile[r][tx];                  // (n = n0 + r, k = k0 + tx)
    if (precision == 0) {
      out[((size_t)t * nrows + n0 + r) * K + k0 + tx] = v;             // [tap][row][K] floats
    } else if (precision == 3) {
      // the fragment-ordered image of the 16 x 16 x 32 weights-direct kernel (conv_wd16_kernel, ops_unet.split_bf16x3_frag(w, 2)):
      // [tap][chunk][row / 16][hi | lo][lane = (k / 8) * 16 + row % 16][k % 8]
      const int row = n0 + r, k = tx;
      __bf16* ob = reinterpret_cast<__bf16*>(out) + ((((size_t)t * (K / 32) + k0 / 32) * (nrows / 16) + row / 16) << 10);
      const __bf16 hi = (__bf16)v;
      const __bf16 lo = (__bf16)(v - (float)hi);
      const int at = (((k >> 3) * 16 + (row & 15)) << 3) + (k & 7);
      ob[at] = hi;
      ob[at + 512] = lo;
    } else if (precision == 2) {
      // the FRAGMENT-ORDERED bf16x3 image of the weights-direct kernels (csrc/unet.hip BDIR, ops_unet.split_bf16x3_frag):
      // [tap][chunk][row / 32][substep][hi | lo][lane = (k % 16 / 8) * 32 + row % 32][k % 8]
      const int row = n0 + r, k = tx;                                  // k0 is a multiple of 32: k = position inside the chunk
      __bf16* ob = reinterpret_cast<__bf16*>(out) + ((((size_t)t * (K / 32) + k0 / 32) * (nrows / 32) + row / 32) << 11);
      const __bf16 hi = (__bf16)v;
      const __bf16 lo = (__bf16)(v - (float)hi);
      const int at = ((k >> 4) << 10) + ((((k >> 3) & 1) * 32 + (row & 31)) << 3) + (k & 7);
      ob[at] = hi;
      ob[at + 512] = lo;
    } else {
      // the bf16x3 image of csrc/unet.hip: [tap][chunk][row][8 slots of 16 B], logical slot (hi: 0-3, lo: 4-7) at physical slot ^ swz(row)
      const int row = n0 + r, swz = (row >> 1) & 7;
      __bf16* ob = reinterpret_cast<__bf16*>(out + (((size_t)t * (K / 32) + k0 / 32) * nrows + row) * 32);
      const __bf16 hi = (__bf16)v;
      const __bf16 lo = (__bf16)(v - (float)hi);
      ob[(((tx >> 3)) ^ swz) * 8 + (tx & 7)] = hi;
      ob[((4 + (tx >> 3)) ^ swz) * 8 + (tx & 7)] = lo;
    }
  }
}

template <bool TR>
__global__ __launch_bounds__(256) void pack_conv_weights_kernel(const float* __restrict__ w, int taps, int Co, int Ci, int row0,
                                                                int nrows, int precision, float* __restrict__ out) {
  __shared__ float tile[32][33];
  pack_conv_weights_tile<TR>(tile, w, taps, Co, Ci, row0, nrows, precision, out, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Every operand image of a training step in ONE launch (mfpa_pack_conv_weights_batch): workgroup -> (job, tile) through the jobs' tile
// prefix (tile0; a job's tiles are (K / 32) x (nrows / 32) x taps, k fastest), then the single-launch kernel's tile code.
__global__ __launch_bounds__(256) void pack_conv_weights_batch_kernel(const mfpa_pack_job* __restrict__ jobs, int njobs) {
  __shared__ float tile[32][33];
  int lo = 0, hi = njobs - 1;                                // the last job whose tile0 <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].tile0 <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const mfpa_pack_job jb = jobs[lo];
  const int K = jb.flip_transpose ? jb.Co : jb.Ci;
  int r = (int)((long long)blockIdx.x - jb.tile0);
  const int kx = r % (K / 32); r /= (K / 32);
  const int ny = r % (jb.nrows / 32);
  const int t = r / (jb.nrows / 32);
  if (t >= jb.taps) return;
  if (jb.flip_transpose) pack_conv_weights_tile<true>(tile, jb.w, jb.taps, jb.Co, jb.Ci, jb.row0, jb.nrows, jb.precision, jb.out, kx, ny, t);
  else pack_conv_weights_tile<false>(tile, jb.w, jb.taps, jb.Co, jb.Ci, jb.row0, jb.nrows, jb.precision, jb.out, kx, ny, t);
}


// Activation -> bf16 copy for the bf16-operand weight-gradient kernel: out[e] = bf16(act(z[e])), act = the consumer's on-load
// transform (relu(z * scale[c] + shift[c]), then the stateless dropout mask of element e) or the identity (scale == NULL).
__global__ __launch_bounds__(256) void act_to_bf16_kernel(const float* __restrict__ z, long long n4, int C, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, unsigned drop_seed, unsigned drop_thresh,
                                                          float drop_scale, __bf16* __restrict__ out) {
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f32x4 v = *reinterpret_cast<const f32x4*>(z + 4 * i);
    if (scale) {
      const int c = (int)((4 * i) % C);
      const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
      v = v * sc + sh;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
      if (drop_thresh) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = mfpa_keep(drop_seed, drop_thresh, (unsigned long long)(4 * i + k)) ? v[k] * drop_scale : 0.f;
      }
    }
    bf16x4_t o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (__bf16)v[k];
    *reinterpret_cast<bf16x4_t*>(out + 4 * i) = o;
  }
}

}  // namespace

template <bool RANK1>
static void launch_bn_bwd_apply(hipStream_t s, float* dy, const float* z, long long npix, int C, const float* scale, const float* shift,
                                const float* mean, const float* invstd, const float* coef, unsigned drop_seed, unsigned drop_thresh,
                                float drop_scale, __bf16* dz16, int write_f32, const float* dpred, const float* w1, int z16, int dy16) {
  if (z16 && !dy16 && C % 8 == 0)
    hipLaunchKernelGGL(bn_bwd_apply8_kernel<RANK1>, dim3(grid_for(npix * (C / 8))), dim3(256), 0, s, dy, reinterpret_cast<const unsigned short*>(z),
                       npix, C, scale, shift, mean, invstd, coef, drop_seed, drop_thresh, drop_scale, dz16, write_f32, dpred, w1);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<RANK1>, dim3(grid_for(npix * (C / 4))), dim3(256), 0, s, dy, z, npix, C, scale, shift, mean, invstd,
                       coef, drop_seed, drop_thresh, drop_scale, dz16, write_f32, dpred, w1, z16, dy16);
}

extern "C" {

int mfpa_red_blocks(void) { return RED_BLOCKS; }

int mfpa_bn_stats(const float* z, long long npix, int C, const float* gamma, const float* beta, float eps,
                  float momentum, float* mean, float* invstd, float* scale, float* shift, float* running_mean,
                  float* running_var, double* workspace, int z_is_bf16, void* stream) {
  if (npix == 0) return MFPA_OK;
  if (!z || !gamma || !beta || !mean || !invstd || !scale || !shift || !workspace) return MFPA_EINVAL;
  if (npix < 0 || C < 4 || C % 4 || (C / 4 < 256 && 256 % (C / 4) != 0) || (C / 4 > 256 && (C / 4) % 256 != 0)) return MFPA_EINVAL;
  hipStream_t s = mfpa_stream(stream);
  const int rows = 256 / (C / 4 < 256 ? C / 4 : 256);
  const int nblk = grid_for(npix, rows * 8, RED_BLOCKS);
  hipLaunchKernelGGL(chan_reduce_kernel<0>, dim3(nblk), dim3(256), 0, s, z, nullptr, npix, C, nullptr, nullptr,
                     nullptr, nullptr, workspace, 0u, 0u, 1.f, z_is_bf16);
  MFPA_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_stats_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, s, workspace, nblk, C, (double)npix,
                     eps, momentum, gamma, beta, mean, invstd, scale, shift, running_mean, running_var);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_bn_relu_bwd(float* dy, const float* z, long long npix, int C, const float* gamma, const float* scale,
                     const float* shift, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                     float* coef, double* workspace, unsigned drop_seed, unsigned drop_thresh, float drop_scale,
                     void* dz_bf16, int write_f32, int z_is_bf16, void* stream) {
  if (npix == 0) return MFPA_OK;
  if (!write_f32 && !dz_bf16) return MFPA_EINVAL;
  if (!dy || !z || !gamma || !scale || !shift || !mean || !invstd || !dgamma || !dbeta || !coef || !workspace) return MFPA_EINVAL;
  if (C & (C - 1)) return MFPA_EINVAL;   // channel counts of this UNet are powers of two (64 ... 1024)
  if (npix < 0 || C < 4 || C % 4 || (C / 4 < 256 && 256 % (C / 4) != 0) || (C / 4 > 256 && (C / 4) % 256 != 0)) return MFPA_EINVAL;
  hipStream_t s = mfpa_stream(stream);
  const int rows = 256 / (C / 4 < 256 ? C / 4 : 256);
  const int nblk = grid_for(npix, rows * 8, RED_BLOCKS);
  hipLaunchKernelGGL(chan_reduce_kernel<1>, dim3(nblk), dim3(256), 0, s, dy, z, npix, C, scale, shift, mean, invstd,
                     workspace, drop_seed, drop_thresh, drop_scale, z_is_bf16);
  MFPA_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, s, workspace, nblk, C, (double)npix,
                     gamma, invstd, dgamma, dbeta, coef);
  MFPA_CHECK_LAUNCH();
  launch_bn_bwd_apply<false>(s, dy, z, npix, C, scale, shift,
                     mean, invstd, coef, drop_seed, drop_thresh, drop_scale, reinterpret_cast<__bf16*>(dz_bf16), write_f32,
                     (const float*)nullptr, (const float*)nullptr, z_is_bf16, 0);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

static bool bn_shape_ok(long long npix, int C) {
  return npix >= 0 && C >= 4 && C % 4 == 0 && !(C / 4 < 256 && 256 % (C / 4) != 0) && !(C / 4 > 256 && (C / 4) % 256 != 0);
}

int mfpa_bn_stats_sums(const float* z, long long npix, int C, double* sums, double* workspace, int z_is_bf16, void* stream) {
  if (!sums || !workspace || !bn_shape_ok(npix, C) || (npix > 0 && !z)) return MFPA_EINVAL;
  hipStream_t s = mfpa_stream(stream);
  if (npix == 0) { MFPA_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, s)); return MFPA_OK; }
  const int rows = 256 / (C / 4 < 256 ? C / 4 : 256);
  const int nblk = grid_for(npix, rows * 8, RED_BLOCKS);
  hipLaunchKernelGGL(chan_reduce_kernel<0>, dim3(nblk), dim3(256), 0, s, z, nullptr, npix, C, nullptr, nullptr,
                     nullptr, nullptr, workspace, 0u, 0u, 1.f, z_is_bf16);
  MFPA_CHECK_LAUNCH();
  hipLaunchKernelGGL(pair_sums_kernel, dim3((C + 3) / 4), dim3(256), 0, s, workspace, nblk, C, sums);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_conv_stats_reduce(const float* part, long long rows, int C, double* sums, double* workspace, void* stream) {
  if (!part || !sums || !workspace || rows < 1 || C < 1 || (C < 256 && 256 % C) || (C > 256 && C % 256)) return MFPA_EINVAL;
  hipStream_t s = mfpa_stream(stream);
  const int groups = 256 / (C < 256 ? C : 256);
  const int nblk = grid_for(rows, groups * 8, RED_BLOCKS);
  hipLaunchKernelGGL(conv_stats_rows_kernel, dim3(nblk), dim3(256), 0, s, part, rows, C, workspace);
  MFPA_CHECK_LAUNCH();
  hipLaunchKernelGGL(pair_sums_kernel, dim3((C + 3) / 4), dim3(256), 0, s, workspace, nblk, C, sums);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

// mfpa_conv_stats_reduce + mfpa_bn_stats_finish in two launches instead of three (single-GPU statistics: no all-reduce between them): the
// finish kernel sums the row blocks' partials itself, in the order pair_sums_kernel does -- bit-identical statistics.
int mfpa_conv_stats_bn_finish(const float* part, long long rows, int C, double count, const float* gamma, const float* beta, float eps,
                              float momentum, float* mean, float* invstd, float* scale, float* shift, float* running_mean,
                              float* running_var, double* workspace, void* stream) {
  if (!part || !workspace || rows < 1 || C < 1 || (C < 256 && 256 % C) || (C > 256 && C % 256)) return MFPA_EINVAL;
  if (!gamma || !beta || !mean || !invstd || !scale || !shift || !(count >= 1.0)) return MFPA_EINVAL;
  hipStream_t s = mfpa_stream(stream);
  const int groups = 256 / (C < 256 ? C : 256);
  const int nblk = grid_for(rows, groups * 8, RED_BLOCKS);
  hipLaunchKernelGGL(conv_stats_rows_kernel, dim3(nblk), dim3(256), 0, s, part, rows, C, workspace);
  MFPA_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_stats_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, s, workspace, nblk, C, count, eps,
                     momentum, gamma, beta, mean, invstd, scale, shift, running_mean, running_var);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

// the BatchNorm + ReLU backward from a convolution's (or pool / OutConv backward's) row partials {sum g, sum g * xhat}: rows -> float64
// block sums -> dgamma / dbeta / coefficients -> apply, three launches (mfpa_conv_stats_reduce + mfpa_bn_relu_bwd_finish: four).  Single-GPU
// statistics only (with SyncBN the sums are all-reduced between the two halves: use those two calls).
int mfpa_bn_relu_bwd_from_part(float* dy, const float* z, long long npix, int C, const float* gamma, const float* scale,
                               const float* shift, const float* mean, const float* invstd, const float* part, long long rows,
                               float* dgamma, float* dbeta, float* coef, double* workspace, unsigned drop_seed, unsigned drop_thresh,
                               float drop_scale, void* dz_bf16, int write_f32, int z_is_bf16, int dy_is_bf16, void* stream) {
  if (!gamma || !scale || !shift || !mean || !invstd || !part || !dgamma || !dbeta || !coef || !workspace || rows < 1) return MFPA_EINVAL;
  if (!write_f32 && !dz_bf16) return MFPA_EINVAL;
  if (dy_is_bf16 && write_f32) return MFPA_EINVAL;
  if (npix < 1 || !bn_shape_ok(npix, C) || (C & (C - 1)) || !dy || !z || (C < 256 && 256 % C) || (C > 256 && C % 256)) return MFPA_EINVAL;
  hipStream_t s = mfpa_stream(stream);
  const int groups = 256 / (C < 256 ? C : 256);
  const int nblk = grid_for(rows, groups * 8, RED_BLOCKS);
  hipLaunchKernelGGL(conv_stats_rows_kernel, dim3(nblk), dim3(256), 0, s, part, rows, C, workspace);
  MFPA_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, s, workspace, nblk, C, (double)npix, gamma, invstd, dgamma,
                     dbeta, coef);
  MFPA_CHECK_LAUNCH();
  launch_bn_bwd_apply<false>(s, dy, z, npix, C, scale, shift,
                     mean, invstd, coef, drop_seed, drop_thresh, drop_scale, reinterpret_cast<__bf16*>(dz_bf16), write_f32,
                     (const float*)nullptr, (const float*)nullptr, z_is_bf16, dy_is_bf16);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_bn_stats_finish(const double* sums, double count, int C, const float* gamma, const float* beta, float eps,
                         float momentum, float* mean, float* invstd, float* scale, float* shift, float* running_mean,
                         float* running_var, void* stream) {
  if (!sums || !gamma || !beta || !mean || !invstd || !scale || !shift || C < 1 || !(count >= 1.0)) return MFPA_EINVAL;
  hipLaunchKernelGGL(bn_stats_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, mfpa_stream(stream), sums, 1, C, count, eps,
                     momentum, gamma, beta, mean, invstd, scale, shift, running_mean, running_var);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_bn_relu_bwd_sums(const float* dy, const float* z, long long npix, int C, const float* scale, const float* shift,
                          const float* mean, const float* invstd, double* sums, double* workspace, unsigned drop_seed,
                          unsigned drop_thresh, float drop_scale, int z_is_bf16, void* stream) {
  if (!sums || !workspace || !scale || !shift || !mean || !invstd || !bn_shape_ok(npix, C) || (C & (C - 1))) return MFPA_EINVAL;
  if (npix > 0 && (!dy || !z)) return MFPA_EINVAL;
  hipStream_t s = mfpa_stream(stream);
  if (npix == 0) { MFPA_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, s)); return MFPA_OK; }
  const int rows = 256 / (C / 4 < 256 ? C / 4 : 256);
  const int nblk = grid_for(npix, rows * 8, RED_BLOCKS);
  hipLaunchKernelGGL(chan_reduce_kernel<1>, dim3(nblk), dim3(256), 0, s, dy, z, npix, C, scale, shift, mean, invstd,
                     workspace, drop_seed, drop_thresh, drop_scale, z_is_bf16);
  MFPA_CHECK_LAUNCH();
  hipLaunchKernelGGL(pair_sums_kernel, dim3((C + 3) / 4), dim3(256), 0, s, workspace, nblk, C, sums);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_bn_relu_bwd_finish(float* dy, const float* z, long long npix, int C, const float* gamma, const float* scale,
                            const float* shift, const float* mean, const float* invstd, const double* local_sums,
                            const double* global_sums, double global_count, float* dgamma, float* dbeta, float* coef,
                            unsigned drop_seed, unsigned drop_thresh, float drop_scale, void* dz_bf16, int write_f32, int z_is_bf16, int dy_is_bf16, void* stream) {
  if (!gamma || !scale || !shift || !mean || !invstd || !local_sums || !global_sums || !dgamma || !dbeta || !coef) return MFPA_EINVAL;
  if (!write_f32 && !dz_bf16) return MFPA_EINVAL;
  if (dy_is_bf16 && write_f32) return MFPA_EINVAL;                    // a bfloat16 dy cannot take the float32 dz in place
  if (!bn_shape_ok(npix, C) || (C & (C - 1)) || !(global_count >= 1.0) || (npix > 0 && (!dy || !z))) return MFPA_EINVAL;
  hipStream_t s = mfpa_stream(stream);
  hipLaunchKernelGGL(bn_bwd_finish_sync_kernel, dim3((C + 255) / 256), dim3(256), 0, s, local_sums, global_sums, C, global_count,
                     gamma, invstd, dgamma, dbeta, coef);
  MFPA_CHECK_LAUNCH();
  if (npix == 0) return MFPA_OK;
  launch_bn_bwd_apply<false>(s, dy, z, npix, C, scale, shift,
                     mean, invstd, coef, drop_seed, drop_thresh, drop_scale, reinterpret_cast<__bf16*>(dz_bf16), write_f32,
                     (const float*)nullptr, (const float*)nullptr, z_is_bf16, dy_is_bf16);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_bn_relu_bwd_finish_rank1(const float* dpred, const float* w1, const float* z, long long npix, int C, const float* gamma,
                                  const float* scale, const float* shift, const float* mean, const float* invstd, const double* local_sums,
                                  const double* global_sums, double global_count, float* dgamma, float* dbeta, float* coef,
                                  float* dz_f32, void* dz_bf16, int z_is_bf16, void* stream) {
  if (!dpred || !w1 || !gamma || !scale || !shift || !mean || !invstd || !local_sums || !global_sums || !dgamma || !dbeta || !coef) return MFPA_EINVAL;
  if (!dz_f32 && !dz_bf16) return MFPA_EINVAL;
  if (!bn_shape_ok(npix, C) || (C & (C - 1)) || !(global_count >= 1.0) || (npix > 0 && !z)) return MFPA_EINVAL;
  hipStream_t s = mfpa_stream(stream);
  hipLaunchKernelGGL(bn_bwd_finish_sync_kernel, dim3((C + 255) / 256), dim3(256), 0, s, local_sums, global_sums, C, global_count,
                     gamma, invstd, dgamma, dbeta, coef);
  MFPA_CHECK_LAUNCH();
  if (npix == 0) return MFPA_OK;
  launch_bn_bwd_apply<true>(s, dz_f32, z, npix, C, scale, shift,
                     mean, invstd, coef, 0u, 0u, 1.f, reinterpret_cast<__bf16*>(dz_bf16), dz_f32 != nullptr ? 1 : 0, dpred, w1, z_is_bf16, 0);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_colsum(const float* x, long long npix, int C, float* out, double* workspace, void* stream) {
  if (npix == 0) return MFPA_OK;
  if (!x || !out || !workspace || npix < 0 || C < 4 || C % 4 || (C / 4 < 256 && 256 % (C / 4) != 0) || (C / 4 > 256 && (C / 4) % 256 != 0)) return MFPA_EINVAL;
  hipStream_t s = mfpa_stream(stream);
  const int rows = 256 / (C / 4 < 256 ? C / 4 : 256);
  const int nblk = grid_for(npix, rows * 8, RED_BLOCKS);
  hipLaunchKernelGGL(chan_reduce_kernel<2>, dim3(nblk), dim3(256), 0, s, x, nullptr, npix, C, nullptr, nullptr, nullptr,
                     nullptr, workspace, 0u, 0u, 1.f, 0);
  MFPA_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, s, workspace, nblk, C, out);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_bn_relu_pool(const float* z, int B, int H, int W, int C, const float* scale, const float* shift, float* p,
                      unsigned drop_seed, unsigned drop_thresh, float drop_scale, int z_is_bf16, int p_is_bf16, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!z || !scale || !shift || !p || B < 0 || H < 2 || W < 2 || C < 4 || C % 4) return MFPA_EINVAL;
  if ((long long)B * (H / 2) > 0x7fffffffLL) return MFPA_EINVAL;
  hipLaunchKernelGGL(bn_relu_pool_kernel, dim3((unsigned)(B * (H / 2))), dim3(256), 0, mfpa_stream(stream), z, B, H, W, C, scale,
                     shift, p, drop_seed, drop_thresh, drop_scale, z_is_bf16, p_is_bf16);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_maxpool2_bwd_add(const float* z, int B, int H, int W, int C, const float* scale, const float* shift,
                          const float* dp, float* dy, unsigned drop_seed, unsigned drop_thresh, float drop_scale,
                          int z_is_bf16, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!z || !scale || !shift || !dp || !dy || B < 0 || H < 2 || W < 2 || C < 4 || C % 4) return MFPA_EINVAL;
  if ((long long)B * (H / 2) > 0x7fffffffLL) return MFPA_EINVAL;
  hipLaunchKernelGGL(maxpool_bwd_add_kernel<0>, dim3((unsigned)(B * (H / 2))), dim3(256), 0, mfpa_stream(stream), z, B, H, W, C,
                     scale, shift, dp, dy, drop_seed, drop_thresh, drop_scale, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, z_is_bf16,
                     0, 0, (const float*)nullptr, (__bf16*)nullptr);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_maxpool2_bwd_add_sums(const float* z, int B, int H, int W, int C, const float* scale, const float* shift, const float* mean,
                               const float* invstd, const float* dp, float* dy, unsigned drop_seed, unsigned drop_thresh,
                               float drop_scale, float* part, int z_is_bf16, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!z || !scale || !shift || !mean || !invstd || !dp || !dy || !part || B < 0 || H < 2 || W < 2 || C < 4 || C % 4) return MFPA_EINVAL;
  if ((long long)B * (H / 2) > 0x7fffffffLL || C / 4 > 256 || 256 % (C / 4)) return MFPA_EINVAL;   // a thread keeps ONE channel quad's sums
  hipLaunchKernelGGL(maxpool_bwd_add_kernel<1>, dim3((unsigned)(B * (H / 2))), dim3(256), 0, mfpa_stream(stream), z, B, H, W, C,
                     scale, shift, dp, dy, drop_seed, drop_thresh, drop_scale, mean, invstd, part, z_is_bf16, 0, 0, (const float*)nullptr, (__bf16*)nullptr);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

// An encoder block's last BatchNorm + ReLU backward WITHOUT its finished input gradient in memory (round 5): dy = the skip path's gradient
// (float32, or bfloat16 with dy_is_bf16) is only read.  Launch 1: g = dy + route(dp) reduced to the BatchNorm-backward row partials (nothing
// written back); 2 + 3: mfpa_conv_stats_reduce's row kernel and the finish (dgamma, dbeta, coefficients); 4: the same g formed again and pushed
// through the backward formula, bfloat16 dz out.  Against mfpa_maxpool2_bwd_add_sums + mfpa_bn_relu_bwd_from_part: the float32 dy is neither
// written nor read back (2 of 4.75 tensor passes at float32 dy, 1.75 of 4.75 more with a bfloat16 dy).  Single-GPU statistics only.
int mfpa_maxpool2_bwd_bn_relu_bwd(const float* z, int B, int H, int W, int C, const float* gamma, const float* scale, const float* shift,
                                  const float* mean, const float* invstd, const float* dp, const void* dy, int dy_is_bf16, unsigned drop_seed,
                                  unsigned drop_thresh, float drop_scale, float* part, float* dgamma, float* dbeta, float* coef,
                                  double* workspace, void* dz_bf16, int z_is_bf16, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!z || !gamma || !scale || !shift || !mean || !invstd || !dp || !dy || !part || !dgamma || !dbeta || !coef || !workspace || !dz_bf16) return MFPA_EINVAL;
  if (B < 0 || H < 2 || W < 2 || C < 4 || C % 4 || (C & (C - 1))) return MFPA_EINVAL;
  if ((long long)B * (H / 2) > 0x7fffffffLL || C / 4 > 256 || 256 % (C / 4) || (C < 256 && 256 % C)) return MFPA_EINVAL;
  hipStream_t s = mfpa_stream(stream);
  const unsigned grid = (unsigned)(B * (H / 2));
  float* dyf = const_cast<float*>(reinterpret_cast<const float*>(dy));
  hipLaunchKernelGGL(maxpool_bwd_add_kernel<1>, dim3(grid), dim3(256), 0, s, z, B, H, W, C, scale, shift, dp, dyf, drop_seed, drop_thresh,
                     drop_scale, mean, invstd, part, z_is_bf16, dy_is_bf16, 1, (const float*)nullptr, (__bf16*)nullptr);
  MFPA_CHECK_LAUNCH();
  const long long rows = (long long)grid;
  const int groups = 256 / (C < 256 ? C : 256);
  const int nblk = grid_for(rows, groups * 8, RED_BLOCKS);
  hipLaunchKernelGGL(conv_stats_rows_kernel, dim3(nblk), dim3(256), 0, s, part, rows, C, workspace);
  MFPA_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, s, workspace, nblk, C, (double)((long long)B * H * W), gamma, invstd,
                     dgamma, dbeta, coef);
  MFPA_CHECK_LAUNCH();
  hipLaunchKernelGGL(maxpool_bwd_add_kernel<2>, dim3(grid), dim3(256), 0, s, z, B, H, W, C, scale, shift, dp, dyf, drop_seed, drop_thresh,
                     drop_scale, mean, invstd, (float*)nullptr, z_is_bf16, dy_is_bf16, 1, coef, reinterpret_cast<__bf16*>(dz_bf16));
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_wgrad_mfma(const mfpa_wgrad_desc* d, void* stream) {
  if (!d) return MFPA_EINVAL;
  if (d->B == 0) return MFPA_OK;
  if (!d->dz || !d->x0 || !d->dw || d->B < 0 || d->H < 1 || d->W < 1) return MFPA_EINVAL;
  if (d->C0 < 64 || d->C0 % 64 || d->C1 < 0 || d->C1 % 64 || d->Cout < 64 || d->Cout % 64) return MFPA_EINVAL;
  if (d->mode != 0 && d->mode != 1) return MFPA_EINVAL;
  if (d->C1 > 0 && (d->mode != 0 || !d->x1 || d->H1 < 1 || d->W1 < 1 || d->H1 > d->H || d->W1 > d->W)) return MFPA_EINVAL;
  if ((d->in_scale0 == nullptr) != (d->in_shift0 == nullptr)) return MFPA_EINVAL;
  WgradArgs a{};
  a.dz = d->dz; a.x0 = d->x0; a.in_scale0 = d->in_scale0; a.in_shift0 = d->in_shift0;
  a.x1 = d->C1 ? d->x1 : nullptr; a.dw = d->dw;
  a.C0 = d->C0; a.C1 = d->C1; a.H1 = d->C1 ? d->H1 : 0; a.W1 = d->C1 ? d->W1 : 0;
  a.oy1 = d->C1 ? (d->H - d->H1) / 2 : 0;
  a.ox1 = d->C1 ? (d->W - d->W1) / 2 : 0;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cout = d->Cout;
  if (d->drop_thresh && !d->in_scale0) return MFPA_EINVAL;
  a.drop_seed = d->drop_seed; a.drop_thresh = d->drop_thresh; a.drop_scale = d->drop_scale;
  a.tiles_x = (d->W + WG_PW - 1) / WG_PW;
  a.tiles_y = (d->H + WG_PH - 1) / WG_PH;
  const long long npatch = (long long)a.B * a.tiles_x * a.tiles_y;
  const int tiles = (d->Cout / WG_T) * ((d->C0 + d->C1) / WG_T);
  long long split = (2048 + tiles - 1) / tiles;   // ~2048 workgroups in flight in total
  if (split > npatch) split = npatch;
  if (split < 1) split = 1;
  if (split > 65535) split = 65535;
  dim3 grid(d->Cout / WG_T, (d->C0 + d->C1) / WG_T, (unsigned)split);
  hipStream_t s = mfpa_stream(stream);
  if (d->precision < 0 || d->precision > 3) return MFPA_EINVAL;
  if (d->precision == 3 && (d->in_scale0 || d->drop_thresh)) return MFPA_EINVAL;      // bf16 operands are already activated
  if (d->precision >= 1) {
    // transposing LDS reads need every lane live (512-thread workgroups, no early exits) -- guaranteed by the kernel shape
    static const int nw_env = MFPA_EXP_ENV("MFPA_WGRAD_NW", 0);   // experiments: 4 or 8 waves
    const int nw = (d->precision != 3 && (nw_env == 4 || nw_env == 8)) ? nw_env : 8;
    const int pix = 16 * nw;
    const int pw = d->W <= 16 ? 16 : 32, phh = pix / pw;
    a.tiles_x = (d->W + pw - 1) / pw;
    a.tiles_y = (d->H + phh - 1) / phh;
    const long long npatch_b = (long long)a.B * a.tiles_x * a.tiles_y;
    long long split_b = ((nw == 8 ? 1024 : 2048) + tiles - 1) / tiles;   // ~4 workgroup rounds over the launch
    if (split_b > npatch_b) split_b = npatch_b;
    if (split_b < 1) split_b = 1;
    if (split_b > 65535) split_b = 65535;
    grid.z = (unsigned)split_b;
    static const int xcd_env = MFPA_EXP_ENV("MFPA_GEMM_XCD", 1);   // 0: plain order (experiments)
    a.xcd = xcd_env;
    const bool plain = d->precision >= 2;
    const size_t lds = (size_t)(plain ? 192 : WGB_ROW) * ((d->mode == 1 && d->precision == 3 ? 4 : 1) * pix + (d->mode == 0 ? (phh + 2) * (pw + 2) : pix));
    const dim3 blk(64 * nw);
#define MFPA_WG_LAUNCH(M, P, N, Q) hipLaunchKernelGGL((wgrad_bf16x3_kernel<M, P, N, Q>), grid, blk, lds, s, a)
#define MFPA_WG_PICK(N, Q)                                  \
    do {                                                    \
      if (d->mode == 0 && pw == 32) MFPA_WG_LAUNCH(0, 32, N, Q); \
      else if (d->mode == 0) MFPA_WG_LAUNCH(0, 16, N, Q);   \
      else if (pw == 32) MFPA_WG_LAUNCH(1, 32, N, Q);       \
      else MFPA_WG_LAUNCH(1, 16, N, Q);                     \
    } while (0)
    if (d->precision == 3 && d->mode == 0) {
      // wgrad_bf16_kernel: 128-channel output tiles when C_out allows; as few patch groups as give every CU one workgroup (the atomics
      // are per workgroup: 256 / 512 / 1024 workgroups measured 291 / 333 / 385 us on 512 -> 512 @ 32 x 31, 64 clips)
      const int cot = d->Cout % 128 == 0 ? 4 : 2;
      const int tiles2 = (d->Cout / (32 * cot)) * ((d->C0 + d->C1) / WG_T);
      const int cus = mfpa_current_device_cus();
      const int target = MFPA_EXP_ENV("MFPA_WGRAD_WGS", cus > 0 ? cus : 256);
      long long split2 = (target + tiles2 - 1) / tiles2;
      if (split2 > npatch_b) split2 = npatch_b;
      if (split2 < 1) split2 = 1;
      if (split2 > 65535) split2 = 65535;
      const dim3 grid2(d->Cout / (32 * cot), (d->C0 + d->C1) / WG_T, (unsigned)split2);
      const size_t lds2 = 2 * ((size_t)128 * (64 * cot + 64) + (size_t)((phh + 2) * (pw + 2) + 4) * 192);
      if (cot == 4 && pw == 32) hipLaunchKernelGGL((wgrad_bf16_kernel<32, 4>), grid2, dim3(512), lds2, s, a);
      else if (cot == 4) hipLaunchKernelGGL((wgrad_bf16_kernel<16, 4>), grid2, dim3(512), lds2, s, a);
      else if (pw == 32) hipLaunchKernelGGL((wgrad_bf16_kernel<32, 2>), grid2, dim3(512), lds2, s, a);
      else hipLaunchKernelGGL((wgrad_bf16_kernel<16, 2>), grid2, dim3(512), lds2, s, a);
    } else if (d->precision == 3) {
      static const int ci128_env = MFPA_EXP_ENV("MFPA_WGRAD_T_CI128", 1);
      if (ci128_env && d->C0 % 128 == 0 && d->C1 == 0) {
        // 64 output x 128 input channels per workgroup (halves the re-reads of the four dz tap tiles)
        const int tiles3 = (d->Cout / WG_T) * (d->C0 / 128);
        long long split3 = (1024 + tiles3 - 1) / tiles3;
        if (split3 > npatch_b) split3 = npatch_b;
        if (split3 < 1) split3 = 1;
        if (split3 > 65535) split3 = 65535;
        const dim3 grid3(d->Cout / WG_T, d->C0 / 128, (unsigned)split3);
        const size_t lds3 = (size_t)192 * 4 * pix + (size_t)320 * pix;
        if (pw == 32) hipLaunchKernelGGL((wgrad_bf16x3_kernel<1, 32, 8, true, true, true>), grid3, dim3(512), lds3, s, a);
        else hipLaunchKernelGGL((wgrad_bf16x3_kernel<1, 16, 8, true, true, true>), grid3, dim3(512), lds3, s, a);
      } else if (pw == 32) hipLaunchKernelGGL((wgrad_bf16x3_kernel<1, 32, 8, true, true>), grid, dim3(512), lds, s, a);
      else hipLaunchKernelGGL((wgrad_bf16x3_kernel<1, 16, 8, true, true>), grid, dim3(512), lds, s, a);
    }
    else if (nw == 8 && plain) MFPA_WG_PICK(8, true);
    else if (nw == 8) MFPA_WG_PICK(8, false);
    else if (plain) MFPA_WG_PICK(4, true);
    else MFPA_WG_PICK(4, false);
#undef MFPA_WG_PICK
#undef MFPA_WG_LAUNCH
  } else if (d->mode == 0) {
    const size_t lds = sizeof(float) * ((size_t)WG_PIX * WG_T + (size_t)(WG_PH + 2) * (WG_PW + 2) * WG_T);
    hipLaunchKernelGGL(wgrad_mfma_kernel<0>, grid, dim3(256), lds, s, a);
  } else {
    const size_t lds = sizeof(float) * ((size_t)WG_PIX * WG_T + (size_t)WG_PH * WG_PW * WG_T);
    hipLaunchKernelGGL(wgrad_mfma_kernel<1>, grid, dim3(256), lds, s, a);
  }
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_act_to_bf16(const float* z, long long n, int C, const float* scale, const float* shift, unsigned drop_seed,
                     unsigned drop_thresh, float drop_scale, void* out_bf16, void* stream) {
  if (n == 0) return MFPA_OK;
  if (!z || !out_bf16 || n < 0 || C < 4 || C % 4 || n % C) return MFPA_EINVAL;
  if ((scale == nullptr) != (shift == nullptr) || (drop_thresh && !scale)) return MFPA_EINVAL;
  hipLaunchKernelGGL(act_to_bf16_kernel, dim3(grid_for(n / 4, 256, 256 * 32)), dim3(256), 0, mfpa_stream(stream), z, n / 4, C, scale, shift,
                     drop_seed, drop_thresh, drop_scale, reinterpret_cast<__bf16*>(out_bf16));
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_wgrad_c1(const float* dz, const float* x32, const double* spec64, const double* denom, int B, int H, int W,
                  int Cout, float* dw, int dz_is_bf16, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!dz || (!x32 && !spec64) || !dw || B < 0 || H < 1 || W < 1) return MFPA_EINVAL;
  if (Cout % 4 || Cout < 4 || Cout > 1024 || (256 % (Cout / 4)) != 0) return MFPA_EINVAL;
  if ((long long)B * H > 0x7fffffffLL) return MFPA_EINVAL;
  const int nrows = B * H;
  hipLaunchKernelGGL(wgrad_c1_kernel, dim3(nrows < 2048 ? nrows : 2048), dim3(256), 0, mfpa_stream(stream), dz, x32,
                     spec64, denom, B, H, W, Cout, dw, dz_is_bf16);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_outconv_fwd(const float* z, long long npix, int C, const float* scale, const float* shift, const float* w,
                     const float* bias, float* pred, int z_is_bf16, void* stream) {
  if (npix == 0) return MFPA_OK;
  if (!z || !scale || !shift || !w || !bias || !pred || npix < 0 || C < 4 || C > 256 || (C & (C - 1))) return MFPA_EINVAL;
  hipLaunchKernelGGL(outconv_fwd_kernel, dim3(grid_for(npix, 256 / (C / 4), 256 * 32)), dim3(256), 0,
                     mfpa_stream(stream), z, npix, C, scale, shift, w, bias, pred, z_is_bf16);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_outconv_bwd(const float* z, const float* dpred, long long npix, int C, const float* scale, const float* shift,
                     const float* w, float* dy, float* dwb, double* workspace, int z_is_bf16, void* stream) {
  if (npix == 0) return MFPA_OK;
  if (!z || !dpred || !scale || !shift || !w || !dy || !dwb || !workspace) return MFPA_EINVAL;
  if (npix < 0 || C < 4 || C > 256 || (C & (C - 1))) return MFPA_EINVAL;
  hipStream_t s = mfpa_stream(stream);
  const int nblk = grid_for(npix, (256 / (C / 4)) * 16, RED_BLOCKS);
  hipLaunchKernelGGL(outconv_bwd_kernel<false>, dim3(nblk), dim3(256), 0, s, z, dpred, npix, C, scale, shift, w, dy, workspace,
                     (const float*)nullptr, (const float*)nullptr, (float*)nullptr, z_is_bf16);
  MFPA_CHECK_LAUNCH();
  // finish: column sums of the (nblk, C+1) partial matrix: C weight gradients then the bias gradient
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((C + 1 + 3) / 4), dim3(256), 0, s, workspace, nblk, C + 1, dwb);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_outconv_bwd_rows(long long npix, int C, int* rows) {
  if (!rows || npix < 0 || C < 4 || C > 256 || (C & (C - 1))) return MFPA_EINVAL;
  *rows = npix == 0 ? 0 : grid_for(npix, (256 / (C / 4)) * 16, RED_BLOCKS);
  return MFPA_OK;
}

int mfpa_outconv_bwd_sums(const float* z, const float* dpred, long long npix, int C, const float* scale, const float* shift,
                          const float* mean, const float* invstd, const float* w, float* dwb, double* workspace, float* part,
                          int z_is_bf16, void* stream) {
  if (npix == 0) return MFPA_OK;
  if (!z || !dpred || !scale || !shift || !mean || !invstd || !w || !dwb || !workspace || !part) return MFPA_EINVAL;
  if (npix < 0 || C < 4 || C > 256 || (C & (C - 1))) return MFPA_EINVAL;
  hipStream_t s = mfpa_stream(stream);
  const int nblk = grid_for(npix, (256 / (C / 4)) * 16, RED_BLOCKS);
  hipLaunchKernelGGL(outconv_bwd_kernel<true>, dim3(nblk), dim3(256), 0, s, z, dpred, npix, C, scale, shift, w, (float*)nullptr, workspace,
                     mean, invstd, part, z_is_bf16);
  MFPA_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((C + 1 + 3) / 4), dim3(256), 0, s, workspace, nblk, C + 1, dwb);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_l1_loss(const float* pred, const double* target, long long n, float* dpred, double* loss, double* workspace,
                 void* stream) {
  if (!pred || !target || !loss || !workspace || n <= 0) return MFPA_EINVAL;
  hipStream_t s = mfpa_stream(stream);
  const int nblk = grid_for(n, 256 * 8, RED_BLOCKS);
  hipLaunchKernelGGL(l1_kernel, dim3(nblk), dim3(256), 0, s, pred, target, n, dpred, workspace);
  MFPA_CHECK_LAUNCH();
  hipLaunchKernelGGL(l1_finish_kernel, dim3(1), dim3(256), 0, s, workspace, nblk, n, loss);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_pack_conv_weights(const float* w, int taps, int Co, int Ci, int flip_transpose, int row0, int nrows, int precision,
                            float* out, void* stream) {
  if (!w || !out || taps < 1 || Co < 32 || Ci < 32 || Co % 32 || Ci % 32 || nrows < 32 || nrows % 32 || row0 < 0 || row0 % 32)
    return MFPA_EINVAL;
  if (precision < 0 || precision > 3) return MFPA_EINVAL;     // 2 / 3: the fragment-ordered bf16x3 images (w_layout 1 / 2)
  if (row0 + nrows > (flip_transpose ? Ci : Co)) return MFPA_EINVAL;
  const int K = flip_transpose ? Co : Ci;
  dim3 grid(K / 32, nrows / 32, taps);
  if (flip_transpose) hipLaunchKernelGGL(pack_conv_weights_kernel<true>, grid, dim3(256), 0, mfpa_stream(stream), w, taps, Co, Ci, row0, nrows, precision, out);
  else hipLaunchKernelGGL(pack_conv_weights_kernel<false>, grid, dim3(256), 0, mfpa_stream(stream), w, taps, Co, Ci, row0, nrows, precision, out);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_pack_conv_weights_batch(const mfpa_pack_job* jobs_dev, int njobs, long long total_tiles, void* stream) {
  if (njobs == 0) return MFPA_OK;
  if (!jobs_dev || njobs < 0 || total_tiles < 1 || total_tiles > 0x7fffffffLL) return MFPA_EINVAL;
  hipLaunchKernelGGL(pack_conv_weights_batch_kernel, dim3((unsigned)total_tiles), dim3(256), 0, mfpa_stream(stream), jobs_dev, njobs);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                   float eps, int step, float grad_scale, void* stream) {
  if (n == 0) return MFPA_OK;
  if (!p || !g || !m || !v || n < 0 || step < 1) return MFPA_EINVAL;
  const double bc1 = 1.0 - pow((double)beta1, step);
  const double bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256 * 4, 256 * 16)), dim3(256), 0, mfpa_stream(stream), p, g, m, v, n,
                     lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), grad_scale);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

}  // extern "C"
