// Landmark pairing and hashing for MI355X (gfx950) -- the step right after the peak pickers
// (SURVEY.md §8f-1).  Integer-only, so results are identical to the reference's.
//
//   audfprint: peaks2landmarks (afp/audfprint/peak_extractor.py:313-346), landmarks2hashes (:40-58) and the
//              duplicate removal of wavfile2hashes (:443-460: merge to time<<32|hash, unique, sort).
//   dejavu   : generate_hashes (afp/dejavu/fingerprint.py:174-213): peaks in time order, each paired with its next
//              fan_value-1 peaks, SHA-1("f1|f2|dt") truncated to 20 hex digits (10 bytes).
//
// One workgroup per clip: the peak mask never leaves the device between the picker and the hash list.
#include "mfpa_common.h"

namespace {

constexpr int HT = 256;       // threads
constexpr int MAXPK = 8;      // peaks per frame (the pruner keeps <= 8)
constexpr int SORT_CAP = 8192;

struct AudLds {
  size_t pk, npk, col_off, sh, keys, total;
};
__host__ __device__ inline AudLds aud_lds(int T, int npow) {   // byte offsets of the landmark kernel's LDS carve
  AudLds l;
  auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
  l.pk = 0;
  l.npk = up(l.pk + sizeof(short) * (size_t)T * MAXPK);
  l.col_off = up(l.npk + sizeof(short) * (size_t)T);
  l.sh = up(l.col_off + sizeof(int) * (size_t)T);
  l.keys = up(l.sh + sizeof(int) * (HT + 2));
  l.total = l.keys + sizeof(unsigned long long) * (size_t)npow;
  return l;
}

__device__ __forceinline__ int block_excl_scan(int v, int* sh, int tid) {  // exclusive scan over HT threads; sh[HT+1]
  sh[tid] = v;
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int i = 0; i < HT; ++i) {
      const int t = sh[i];
      sh[i] = acc;
      acc += t;
    }
    sh[HT] = acc;
  }
  __syncthreads();
  return sh[tid];
}

// mask (R, T) uint8, R <= 256.  landmarks (cap, 4) int32 in the reference's list order, hashes (cap, 2) int32 in
// the same order, uniq (cap, 2) int32 sorted unique, counts[0] = landmarks (or -1 on overflow), counts[1] = unique.
__global__ __launch_bounds__(HT) void audfprint_landmarks_kernel(const uint8_t* __restrict__ mask, int R, int T, int cap,
                                                                 int mindt, int targetdt, int targetdf, int maxpairs,
                                                                 int32_t* __restrict__ landmarks, int32_t* __restrict__ hashes,
                                                                 int32_t* __restrict__ uniq, int32_t* __restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const AudLds L = aud_lds(T, 1);
  short* pk = reinterpret_cast<short*>(smem + L.pk);             // [T][MAXPK]
  short* npk = reinterpret_cast<short*>(smem + L.npk);           // [T]
  int* col_off = reinterpret_cast<int*>(smem + L.col_off);       // [T]
  int* sh = reinterpret_cast<int*>(smem + L.sh);                 // [HT + 2] scan scratch
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem + L.keys);  // [pow2 >= n]
  __shared__ int s_flag, s_scols;

  const int tid = threadIdx.x, b = blockIdx.x;
  const uint8_t* M = mask + (size_t)b * R * T;
  int32_t* LM = landmarks + (size_t)b * cap * 4;
  int32_t* HS = hashes + (size_t)b * cap * 2;
  int32_t* UQ = uniq + (size_t)b * cap * 2;
  if (tid == 0) { s_flag = 0; s_scols = 0; }
  __syncthreads();

  // per-frame peak lists (bins ascending): thread = frame, consecutive threads read consecutive bytes of a bin row
  for (int c = tid; c < T; c += HT) {
    int n = 0;
    for (int r = 0; r < R; ++r) {
      if (M[(size_t)r * T + c]) {
        if (n < MAXPK) pk[c * MAXPK + n] = (short)r;
        ++n;
      }
    }
    if (n > MAXPK) { atomicOr(&s_flag, 1); n = MAXPK; }
    npk[c] = (short)n;
    if (n) atomicMax(&s_scols, c + 1);
  }
  __syncthreads();
  const int scols = s_scols;   // pklist[-1][0] + 1  (peak_extractor.py:327)

  // pass 1: landmarks per frame -> offsets
  auto pairs_of = [&](int c, int i, int32_t* out /* nullable */) {
    const int p = pk[c * MAXPK + i];
    int pairs = 0;
    const int c_end = min(scols, c + targetdt);
    for (int c2 = c + mindt; c2 < c_end && pairs < maxpairs; ++c2) {
      const int n2 = npk[c2];
      for (int k = 0; k < n2 && pairs < maxpairs; ++k) {
        const int p2 = pk[c2 * MAXPK + k];
        const int d = p2 - p;
        if ((d < 0 ? -d : d) < targetdf) {
          if (out) { out[4 * pairs] = c; out[4 * pairs + 1] = p; out[4 * pairs + 2] = p2; out[4 * pairs + 3] = c2 - c; }
          ++pairs;
        }
      }
    }
    return pairs;
  };
  int total = 0;
  {
    // frames are distributed in contiguous chunks so the block scan yields list order
    const int per = (T + HT - 1) / HT;
    const int c0 = tid * per, c1 = min(T, c0 + per);
    int mine = 0;
    for (int c = c0; c < c1; ++c) {
      int cc = 0;
      for (int i = 0; i < npk[c]; ++i) cc += pairs_of(c, i, nullptr);
      col_off[c] = cc;
      mine += cc;
    }
    int off = block_excl_scan(mine, sh, tid);
    total = sh[HT];
    for (int c = c0; c < c1; ++c) {
      const int cc = col_off[c];
      col_off[c] = off;
      off += cc;
    }
  }
  __syncthreads();
  const bool overflow = (total > cap) || (total > SORT_CAP) || s_flag;
  if (overflow) {
    if (tid == 0) { counts[2 * b] = -1; counts[2 * b + 1] = -1; }
    return;
  }
  // pass 2: emit landmarks + hashes in list order, keys for the unique/sort step
  int npow = 1;
  while (npow < total) npow <<= 1;
  for (int i = tid; i < npow; i += HT) keys[i] = ~0ull;
  __syncthreads();
  for (int c = tid; c < T; c += HT) {
    int off = col_off[c];
    for (int i = 0; i < npk[c]; ++i) {
      int32_t tmp[12];
      const int n = pairs_of(c, i, tmp);
      for (int k = 0; k < n; ++k) {
        const int e = off + k;
        LM[4 * e] = tmp[4 * k]; LM[4 * e + 1] = tmp[4 * k + 1]; LM[4 * e + 2] = tmp[4 * k + 2]; LM[4 * e + 3] = tmp[4 * k + 3];
        const int32_t h = ((tmp[4 * k + 1] & 255) << 12) | (((tmp[4 * k + 2] - tmp[4 * k + 1]) & 63) << 6) | (tmp[4 * k + 3] & 63);
        HS[2 * e] = tmp[4 * k];
        HS[2 * e + 1] = h;
        keys[e] = ((unsigned long long)(unsigned)tmp[4 * k] << 32) + (unsigned long long)(unsigned)h;
      }
      off += n;
    }
  }
  __syncthreads();
  // bitonic sort of npow keys in LDS
  for (int k = 2; k <= npow; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npow; i += HT) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = keys[i], c = keys[ixj];
          const bool up = ((i & k) == 0);
          if ((a > c) == up) { keys[i] = c; keys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
  // unique compaction (keys sorted ascending; padding = ~0 never equals a real key)
  {
    const int per = (total + HT - 1) / HT;
    const int i0 = tid * per, i1 = min(total, i0 + per);
    int mine = 0;
    for (int i = i0; i < i1; ++i) mine += (i == 0 || keys[i] != keys[i - 1]);
    int off = block_excl_scan(mine, sh, tid);
    for (int i = i0; i < i1; ++i) {
      if (i == 0 || keys[i] != keys[i - 1]) {
        UQ[2 * off] = (int32_t)(keys[i] >> 32);
        UQ[2 * off + 1] = (int32_t)(keys[i] & 0xFFFFFFFFull);
        ++off;
      }
    }
    if (tid == 0) { counts[2 * b] = total; counts[2 * b + 1] = sh[HT]; }
  }
}

// ------------------------------------------------------------------------------------------ dejavu + SHA-1
__device__ __forceinline__ uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }

__device__ int put_dec(unsigned char* m, int pos, int v) {   // decimal digits of v >= 0
  char tmp[12];
  int n = 0;
  do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (n) m[pos++] = (unsigned char)tmp[--n];
  return pos;
}

// SHA-1 of the ASCII string "f1|f2|dt" (<= 55 bytes: one block); writes the first 10 digest bytes.
__device__ void sha1_trunc10(int f1, int f2, int dt, uint8_t* out10) {
  unsigned char m[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) m[i] = 0;
  int len = put_dec(m, 0, f1);
  m[len++] = '|';
  len = put_dec(m, len, f2);
  m[len++] = '|';
  len = put_dec(m, len, dt);
  m[len] = 0x80;
  const unsigned bits = (unsigned)len * 8u;
  m[62] = (unsigned char)(bits >> 8);
  m[63] = (unsigned char)(bits & 0xFF);
  uint32_t w[80];
  for (int i = 0; i < 16; ++i)
    w[i] = ((uint32_t)m[4 * i] << 24) | ((uint32_t)m[4 * i + 1] << 16) | ((uint32_t)m[4 * i + 2] << 8) | (uint32_t)m[4 * i + 3];
  for (int i = 16; i < 80; ++i) w[i] = rol(w[i - 3] ^ w[i - 8] ^ w[i - 14] ^ w[i - 16], 1);
  uint32_t a = 0x67452301u, b = 0xEFCDAB89u, c = 0x98BADCFEu, d = 0x10325476u, e = 0xC3D2E1F0u;
  for (int i = 0; i < 80; ++i) {
    uint32_t f, k;
    if (i < 20) { f = (b & c) | (~b & d); k = 0x5A827999u; }
    else if (i < 40) { f = b ^ c ^ d; k = 0x6ED9EBA1u; }
    else if (i < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDCu; }
    else { f = b ^ c ^ d; k = 0xCA62C1D6u; }
    const uint32_t t = rol(a, 5) + f + e + k + w[i];
    e = d; d = c; c = rol(b, 30); b = a; a = t;
  }
  const uint32_t h0 = 0x67452301u + a, h1 = 0xEFCDAB89u + b, h2 = 0x98BADCFEu + c;
  out10[0] = (uint8_t)(h0 >> 24); out10[1] = (uint8_t)(h0 >> 16); out10[2] = (uint8_t)(h0 >> 8); out10[3] = (uint8_t)h0;
  out10[4] = (uint8_t)(h1 >> 24); out10[5] = (uint8_t)(h1 >> 16); out10[6] = (uint8_t)(h1 >> 8); out10[7] = (uint8_t)h1;
  out10[8] = (uint8_t)(h2 >> 24); out10[9] = (uint8_t)(h2 >> 16);
}

// mask (F, T) uint8.  Peaks in (time, freq) order = the reference's stable sort by time of the row-major list.
// digests (cap, 10) uint8, t1 (cap) int32, counts[b] = number of hashes (-1 on overflow).
__global__ __launch_bounds__(HT) void dejavu_hashes_kernel(const uint8_t* __restrict__ mask, int F, int T, int cap,
                                                           int peak_cap, int fan, int min_dt, int max_dt,
                                                           uint8_t* __restrict__ digests, int32_t* __restrict__ t1,
                                                           int32_t* __restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* sh = reinterpret_cast<int*>(smem);          // [HT + 2]
  int* col_off = sh + HT + 2;                      // [T]
  short* pf = reinterpret_cast<short*>(col_off + T);   // [peak_cap] freq
  short* pt = pf + peak_cap;                       // [peak_cap] time
  const int tid = threadIdx.x, b = blockIdx.x;
  const uint8_t* M = mask + (size_t)b * F * T;
  const int per = (T + HT - 1) / HT;
  const int c0 = tid * per, c1 = min(T, c0 + per);
  int mine = 0;
  for (int c = c0; c < c1; ++c) {
    int n = 0;
    for (int r = 0; r < F; ++r) n += M[(size_t)r * T + c] != 0;
    col_off[c] = n;
    mine += n;
  }
  int off = block_excl_scan(mine, sh, tid);
  const int npeaks = sh[HT];
  if (npeaks > peak_cap) {
    if (tid == 0) counts[b] = -1;
    return;
  }
  for (int c = c0; c < c1; ++c) {
    for (int r = 0; r < F; ++r)
      if (M[(size_t)r * T + c]) { pf[off] = (short)r; pt[off] = (short)c; ++off; }
  }
  __syncthreads();
  // hashes per peak i: j = 1 .. fan-1, kept when min_dt <= t2 - t1 <= max_dt
  const int pper = (npeaks + HT - 1) / HT;
  const int i0 = tid * pper, i1 = min(npeaks, i0 + pper);
  mine = 0;
  for (int i = i0; i < i1; ++i)
    for (int j = 1; j < fan; ++j)
      if (i + j < npeaks) { const int dt = pt[i + j] - pt[i]; mine += (dt >= min_dt && dt <= max_dt); }
  off = block_excl_scan(mine, sh, tid);
  const int total = sh[HT];
  if (total > cap) {
    if (tid == 0) counts[b] = -1;
    return;
  }
  uint8_t* D = digests + (size_t)b * cap * 10;
  int32_t* TT = t1 + (size_t)b * cap;
  for (int i = i0; i < i1; ++i)
    for (int j = 1; j < fan; ++j)
      if (i + j < npeaks) {
        const int dt = pt[i + j] - pt[i];
        if (dt >= min_dt && dt <= max_dt) {
          sha1_trunc10(pf[i], pf[i + j], dt, D + (size_t)off * 10);
          TT[off] = pt[i];
          ++off;
        }
      }
  if (tid == 0) counts[b] = total;
}

}  // namespace

extern "C" {

int mfpa_audfprint_landmarks(const uint8_t* mask, int B, int R, int T, int cap, int mindt, int targetdt, int targetdf,
                             int maxpairs, int32_t* landmarks, int32_t* hashes, int32_t* uniq, int32_t* counts,
                             void* stream) {
  if (B == 0) return MFPA_OK;
  if (!mask || !landmarks || !hashes || !uniq || !counts || B < 0) return MFPA_EINVAL;
  if (R < 1 || R > 256 || T < 1 || T > 4096 || cap < 1 || cap > SORT_CAP || maxpairs < 1 || maxpairs > 3 || mindt < 0) return MFPA_EINVAL;
  int npow = 1;
  while (npow < cap) npow <<= 1;
  const size_t lds = aud_lds(T, npow).total;
  if (lds > 150 * 1024) return MFPA_EINVAL;
  hipLaunchKernelGGL(audfprint_landmarks_kernel, dim3(B), dim3(HT), lds, mfpa_stream(stream), mask, R, T, cap, mindt, targetdt,
                     targetdf, maxpairs, landmarks, hashes, uniq, counts);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

int mfpa_dejavu_hashes(const uint8_t* mask, int B, int F, int T, int cap, int peak_cap, int fan, int min_dt, int max_dt,
                       uint8_t* digests, int32_t* t1, int32_t* counts, void* stream) {
  if (B == 0) return MFPA_OK;
  if (!mask || !digests || !t1 || !counts || B < 0) return MFPA_EINVAL;
  if (F < 1 || F > 32767 || T < 1 || T > 32767 || cap < 1 || peak_cap < 1 || peak_cap > 16384 || fan < 1 || fan > 64) return MFPA_EINVAL;
  const size_t lds = sizeof(int) * ((size_t)HT + 2 + T) + sizeof(short) * 2 * (size_t)peak_cap;
  if (lds > 150 * 1024) return MFPA_EINVAL;
  hipLaunchKernelGGL(dejavu_hashes_kernel, dim3(B), dim3(HT), lds, mfpa_stream(stream), mask, F, T, cap, peak_cap, fan, min_dt,
                     max_dt, digests, t1, counts);
  MFPA_CHECK_LAUNCH();
  return MFPA_OK;
}

}  // extern "C"
