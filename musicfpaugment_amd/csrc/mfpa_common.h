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
