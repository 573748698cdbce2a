// numpy's float summation order, reproduced on the device (shared by the audfprint and dejavu
// pre-processing kernels).  np.mean / np.add.reduce of a contiguous array walks MEMORY order in
// chunks of 8192 elements (the ufunc buffer size): acc = 0; acc += pairwise_sum(chunk), where
// pairwise_sum splits at n/2 rounded down to a multiple of 8 until blocks of <= 128 elements,
// each summed with eight strided accumulators (numpy/_core/src/umath/loops_utils.h.src).
#pragma once
#include "mfpa_common.h"

namespace mfpa_np {

constexpr int NPY_BUFSIZE = 8192;  // numpy reduces a contiguous array in chunks of 8192 elements
constexpr int PW_BLOCK = 128;      // pairwise-sum leaf size
constexpr int HEAP = 256;          // nodes per chunk (tree depth <= 7)
constexpr int MAX_CHUNKS = 64;

struct NodeInfo {
  bool exists;
  int off, n;
};

// Walk numpy's pairwise split (n2 = n/2 rounded down to a multiple of 8) from the chunk root to heap node `id`.
__device__ __forceinline__ NodeInfo pw_node(int chunk_n, int id) {
  NodeInfo r{true, 0, chunk_n};
  const int depth = 31 - __clz(id);
  for (int bit = depth - 1; bit >= 0; --bit) {
    if (r.n <= PW_BLOCK) {
      r.exists = false;
      return r;
    }
    int n2 = r.n / 2;
    n2 -= n2 % 8;
    if ((id >> bit) & 1) {
      r.off += n2;
      r.n -= n2;
    } else {
      r.n = n2;
    }
  }
  return r;
}

template <typename S>
__device__ __forceinline__ S group8_sum(S v) {  // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) on every lane of the group
  v = v + __shfl_xor(v, 1);
  v = v + __shfl_xor(v, 2);
  v = v + __shfl_xor(v, 4);
  return v;
}


// Sum of N = F*T values stored bin-major in L (element (f,t) at f*T+t), taken in numpy's order for
// a C-contiguous (order 0: bin-major) or F-contiguous (order 1: frame-major) array.  S is the
// accumulation dtype (the array's dtype).  All `nthreads` threads of the block must call; `heap`
// is nchunks*HEAP elements of LDS.  Returns the sum on every thread via `bcast` (1 LDS slot).
template <typename S>
__device__ __forceinline__ S block_numpy_sum(const double* __restrict__ L, int N, int F, int T, int order, S* heap,
                                             double* bcast, int tid, int nthreads) {
  const int nchunks = (N + NPY_BUFSIZE - 1) / NPY_BUFSIZE;
  const int lane8 = tid & 7, grp = tid >> 3;
  for (int cand = grp; cand < nchunks * HEAP; cand += nthreads / 8) {
    const int c = cand / HEAP, id = cand % HEAP;
    if (id == 0) continue;
    const int cn = min(NPY_BUFSIZE, N - c * NPY_BUFSIZE);
    const NodeInfo nd = pw_node(cn, id);
    if (!nd.exists || nd.n > PW_BLOCK) continue;
    const int base = c * NPY_BUFSIZE + nd.off;
    auto at = [&](int i) -> S {  // i-th element in numpy's memory order
      const int e = base + i;
      const int addr = order ? (e % F) * T + (e / F) : e;
      return (S)L[addr];
    };
    S res;
    if (nd.n < 8) {
      res = 0;
      for (int i = 0; i < nd.n; ++i) res = res + at(i);
    } else {
      // all of a lane's <= 16 strided elements are loaded before the (ordered) additions: independent loads in flight
      const int n8 = nd.n - (nd.n % 8);
      S vals[PW_BLOCK / 8];
#pragma unroll
      for (int j = 0; j < PW_BLOCK / 8; ++j) vals[j] = (8 * j < n8) ? at(8 * j + lane8) : (S)0;
      S acc = vals[0];
#pragma unroll
      for (int j = 1; j < PW_BLOCK / 8; ++j)
        if (8 * j < n8) acc = acc + vals[j];
      res = group8_sum(acc);
      for (int i = n8; i < nd.n; ++i) res = res + at(i);
    }
    if (lane8 == 0) heap[c * HEAP + id] = res;
  }
  __syncthreads();
  for (int d = 6; d >= 0; --d) {
    const int first = 1 << d, cnt = 1 << d;
    for (int k = tid; k < nchunks * cnt; k += nthreads) {
      const int c = k / cnt, id = first + k % cnt;
      const int cn = min(NPY_BUFSIZE, N - c * NPY_BUFSIZE);
      const NodeInfo nd = pw_node(cn, id);
      if (nd.exists && nd.n > PW_BLOCK) heap[c * HEAP + id] = heap[c * HEAP + 2 * id] + heap[c * HEAP + 2 * id + 1];
    }
    __syncthreads();
  }
  if (tid == 0) {
    S acc = 0;
    for (int c = 0; c < nchunks; ++c) acc = acc + heap[c * HEAP + 1];
    *bcast = (double)acc;
  }
  __syncthreads();
  return (S)*bcast;
}

}  // namespace mfpa_np
