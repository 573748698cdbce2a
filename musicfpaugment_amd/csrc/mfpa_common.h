// Shared helpers for the libmfpa HIP sources (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mfpa.h"

#define MFPA_WAVE 64

// Experiment switches (A/B timing runs of tools/, some of which skip work and give WRONG results) exist only in a library
// built with -DMFPA_EXPERIMENTS (`python -m musicfpaugment_amd.csrc.build --experiments` -> libmfpa_exp.so, never loaded by the
// package on its own).  The product library does not read the environment: every switch is its compile-time default.
#ifdef MFPA_EXPERIMENTS
#include <cstdlib>
#define MFPA_EXP_ENV(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#define MFPA_EXP_FLAG(word, bit) (((word) & (bit)) != 0)
#elif defined(MFPA_SKIP_BITS)
// timing-only variants of the PRODUCT code (tools/: `build(extra_flags=["-DMFPA_SKIP_BITS=<bits>"], out=...)`): the skip switches of the
// experiments build as compile-time constants, so that the variant differs from the shipped kernel by the skipped work only (the
// run-time switches cost conv_wd16_kernel's 64-channel form 12 % by themselves).  Wrong results by design; never loaded by the package.
#define MFPA_EXP_ENV(name, dflt) (dflt)
#define MFPA_EXP_FLAG(word, bit) (((MFPA_SKIP_BITS) & (bit)) != 0)
#else
#define MFPA_EXP_ENV(name, dflt) (dflt)
#define MFPA_EXP_FLAG(word, bit) false
#endif

#define MFPA_CHECK_LAUNCH()                                   \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return MFPA_EHIP - (int)e__;       \
  } while (0)

#define MFPA_HIP(call)                                        \
  do {                                                        \
    hipError_t e__ = (call);                                  \
    if (e__ != hipSuccess) return MFPA_EHIP - (int)e__;       \
  } while (0)

// Kernels carrying this attribute are compiled without packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).
// hipcc pairs scalar float expressions into those freely and, when one operand sits in the high half of a register pair, selects it
// with op_sel:[..1..] (the LOW lane reads a HIGH half).  That encoding returned sporadically wrong low lanes next to MFMA waves on
// this toolchain (profiles/r02_pk_fma_op_sel.md), so the product library contains none: every kernel in which hipcc chose it carries
// this attribute (same IEEE results: a packed FMA is two scalar FMAs), and tests/test_isa_scan.py disassembles the shipped
// libmfpa.so and fails on any packed-fp32 instruction with an op_sel:[...] operand selection.
#if defined(__HIP_DEVICE_COMPILE__)
#define MFPA_NO_PK_F32 __attribute__((target("no-packed-fp32-ops")))
#else
#define MFPA_NO_PK_F32      // host pass of the same translation unit: the attribute names a gfx950 feature
#endif

// {x, x} for a packed-fp32 multiply-add by a scalar, built so that hipcc CANNOT take x out of the high half of a register pair
// (the op_sel:[...] form): x is moved into the LOW half of a fresh pair (the empty asm pins it there) and broadcast from it, which is
// the op_sel_hi form.  One v_mov per scalar; the FMAs stay packed -- hot VALU kernels (conv1d_c1_kernel, c1_wgrad_kernel, the UNet's
// first layer in the loader) lose far less than under MFPA_NO_PK_F32, which doubles their FMA count.
typedef float mfpa_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ mfpa_f32x2 mfpa_bcast2(float x) {
  mfpa_f32x2 lo;
  lo.x = x;                                  // the high half is never read
  asm("" : "+v"(lo));                        // NOT volatile: volatile asm statements keep their program order, which serialised the nine LDS sample
                                             // reads of the UNet's first layer (each waited for before the next was issued: 7 us per chunk and workgroup)
  return mfpa_f32x2{lo.x, lo.x};
}

// CU count of the CURRENT device (cached per device id: a process may drive several GPUs).  The persistent LSTM kernels size their
// grids against it -- one workgroup per CU is the residency they rely on (their LDS / register footprint allows no second one).
static inline int mfpa_current_device_cus() {
  static int cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 0;
  if (dev >= 64) {
    int n = 0;
    return hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess ? n : 0;
  }
  if (cus[dev] <= 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    cus[dev] = n;
  }
  return cus[dev];
}

static inline hipStream_t mfpa_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

__device__ __forceinline__ double mfpa_wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}

// Non-negative doubles order like their bit patterns: max via integer atomic.
__device__ __forceinline__ void mfpa_atomic_max_nonneg(double* addr, double v) {
  atomicMax(reinterpret_cast<unsigned long long*>(addr),
            static_cast<unsigned long long>(__double_as_longlong(v)));
}

// Counter-based dropout mask (training): keep element `idx` of a tensor iff hash(seed, idx) >= thresh, where
// thresh = rate * 2^32.  Stateless, so the mask is recomputed wherever the dropped activation is consumed
// (forward loaders and the backward pass) and never stored.  Mirrored in numpy by tests/test_gpu_train.py.
__host__ __device__ __forceinline__ uint32_t mfpa_mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ bool mfpa_keep(uint32_t seed, uint32_t thresh, unsigned long long idx) {
  const uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
  return mfpa_mix32(mfpa_mix32(lo + seed) ^ (hi * 0x7F4A7C15u + seed)) >= thresh;
}
