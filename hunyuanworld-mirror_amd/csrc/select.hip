// Confidence top-k mask (reference: infer.py:25-59 create_confidence_mask, used by infer.py on the
// predicted confidences right after the path).
//
//   conf' = conf <= 1e-5 ? -inf : conf;   K = p > 0 ? max(1, ceil(N (100 - p) / 100)) : N;   mask = top-K of conf'
//
// HBM-bound integer work: a 4 x 8-bit radix SELECT on the order-preserving uint32 image of conf' finds the K-th
// largest key exactly (per pass: per-block LDS histogram -> integer atomics -> one small block walks the 256 bins from
// the top), then one pass writes the mask.  Ties at the K-th value are broken by lowest index (the reference's
// torch.topk leaves the choice unspecified): a per-block count of the tied elements, a single-block exclusive scan and
// an in-block ordered rank.  Everything is integer / comparison arithmetic: the result is exact and deterministic.
#include "wm_common.h"
#include "wm_kernels.h"

namespace {

struct SelState {
  unsigned int prefix;       // bits of the K-th key fixed so far (high bits)
  unsigned int remaining;    // how many elements with the fixed prefix (and below-the-top bins) are still to take
  unsigned int hist[256];
  unsigned int kth;          // final K-th key
  unsigned int need_eq;      // number of elements equal to kth that belong to the top-K
  unsigned int total_eq;     // number of elements equal to kth
};

__device__ __forceinline__ unsigned int conf_key(float c) {
  const float v = c <= 1e-5f ? -INFINITY : c;  // NaN compares false -> kept as is: sorts above +inf like torch.topk's NaN-largest
  const unsigned int u = __builtin_bit_cast(unsigned int, v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone: larger float <=> larger key
}

constexpr int SEL_EPT = 16;                 // consecutive elements per thread
constexpr int SEL_CHUNK = 256 * SEL_EPT;    // elements per block

__global__ __launch_bounds__(256) void sel_init_kernel(SelState* st, unsigned int K) {
  st->hist[threadIdx.x] = 0;
  if (threadIdx.x == 0) { st->prefix = 0; st->remaining = K; st->kth = 0; st->need_eq = 0; st->total_eq = 0; }
}

// histogram of byte `shift/8` of the keys whose higher bits equal st->prefix
__global__ __launch_bounds__(256) void sel_hist_kernel(const float* __restrict__ conf, size_t n, SelState* st, int shift) {
  __shared__ unsigned int lh[256];
  lh[threadIdx.x] = 0;
  __syncthreads();
  const unsigned int prefix = st->prefix;
  const unsigned int himask = shift == 24 ? 0u : ~0u << (shift + 8);
  const size_t base = (size_t)blockIdx.x * SEL_CHUNK + (size_t)threadIdx.x * SEL_EPT;
#pragma unroll
  for (int e = 0; e < SEL_EPT; ++e) {
    const size_t i = base + e;
    if (i < n) {
      const unsigned int k = conf_key(conf[i]);
      if ((k & himask) == prefix) atomicAdd(&lh[(k >> shift) & 255u], 1u);
    }
  }
  __syncthreads();
  if (lh[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], lh[threadIdx.x]);
}

// one block: walk the bins from the top until `remaining` is covered; fix the next byte of the K-th key
__global__ __launch_bounds__(256) void sel_pick_kernel(SelState* st, int shift) {
  __shared__ unsigned int h[256];
  h[threadIdx.x] = st->hist[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int rem = st->remaining, bin = 0;
    for (int b = 255; b >= 0; --b) {
      if (h[b] >= rem) { bin = (unsigned int)b; break; }
      rem -= h[b];
    }
    st->prefix |= bin << shift;
    st->remaining = rem;  // elements to take from inside the chosen bin
    if (shift == 0) { st->kth = st->prefix; st->need_eq = rem; st->total_eq = h[bin]; }
  }
  __syncthreads();
  st->hist[threadIdx.x] = 0;
}

// per-block number of elements equal to the K-th key
__global__ __launch_bounds__(256) void sel_count_eq_kernel(const float* __restrict__ conf, size_t n, const SelState* st,
                                                           unsigned int* __restrict__ blk_eq) {
  __shared__ unsigned int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  const unsigned int kth = st->kth;
  const size_t base = (size_t)blockIdx.x * SEL_CHUNK + (size_t)threadIdx.x * SEL_EPT;
  unsigned int c = 0;
#pragma unroll
  for (int e = 0; e < SEL_EPT; ++e) {
    const size_t i = base + e;
    if (i < n && conf_key(conf[i]) == kth) ++c;
  }
  if (c) atomicAdd(&cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) blk_eq[blockIdx.x] = cnt;
}

// single block: exclusive scan of the per-block tie counts (in place)
__global__ __launch_bounds__(256) void sel_scan_kernel(unsigned int* __restrict__ blk_eq, unsigned int nblk) {
  __shared__ unsigned int part[256];
  const unsigned int per = (nblk + 255) / 256, lo = threadIdx.x * per, hi = lo + per < nblk ? lo + per : nblk;
  unsigned int s = 0;
  for (unsigned int i = lo; i < hi; ++i) s += blk_eq[i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int run = 0;
    for (int i = 0; i < 256; ++i) { const unsigned int t = part[i]; part[i] = run; run += t; }
  }
  __syncthreads();
  unsigned int run = part[threadIdx.x];
  for (unsigned int i = lo; i < hi; ++i) { const unsigned int t = blk_eq[i]; blk_eq[i] = run; run += t; }
}

// mask: key > kth -> 1; key == kth -> 1 for the first need_eq of them in index order
__global__ __launch_bounds__(256) void sel_mask_kernel(const float* __restrict__ conf, size_t n, const SelState* st,
                                                       const unsigned int* __restrict__ blk_eq, unsigned char* __restrict__ mask) {
  __shared__ unsigned int tcnt[256];
  const unsigned int kth = st->kth, need = st->need_eq;
  const bool all_ties = need == st->total_eq;
  const size_t base = (size_t)blockIdx.x * SEL_CHUNK + (size_t)threadIdx.x * SEL_EPT;
  unsigned int keys[SEL_EPT];
  unsigned int c = 0;
#pragma unroll
  for (int e = 0; e < SEL_EPT; ++e) {
    const size_t i = base + e;
    keys[e] = i < n ? conf_key(conf[i]) : 0u;
    if (i < n && keys[e] == kth) ++c;
  }
  unsigned int rank = 0;
  if (!all_ties) {  // ordered rank of this thread's first tied element: block offset + ties of lower threads
    tcnt[threadIdx.x] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned int run = blk_eq[blockIdx.x];
      for (int i = 0; i < 256; ++i) { const unsigned int t = tcnt[i]; tcnt[i] = run; run += t; }
    }
    __syncthreads();
    rank = tcnt[threadIdx.x];
  }
#pragma unroll
  for (int e = 0; e < SEL_EPT; ++e) {
    const size_t i = base + e;
    if (i >= n) break;
    unsigned char m = keys[e] > kth ? 1 : 0;
    if (keys[e] == kth) { m = (all_ties || rank < need) ? 1 : 0; ++rank; }
    mask[i] = m;
  }
}

}  // namespace

size_t wm_confidence_mask_workspace(size_t n) {
  const size_t nblk = (n + SEL_CHUNK - 1) / SEL_CHUNK;
  return sizeof(SelState) + 256 + nblk * sizeof(unsigned int);
}

hipError_t wm_launch_confidence_mask(const float* conf, size_t n, unsigned int K, unsigned char* mask, void* workspace,
                                     hipStream_t s) {
  if (n == 0) return hipSuccess;
  if (K < 1 || K > n || n >= ((size_t)1 << 32)) return hipErrorInvalidValue;
  SelState* st = (SelState*)workspace;
  unsigned int* blk_eq = (unsigned int*)((char*)workspace + ((sizeof(SelState) + 255) / 256) * 256);
  const unsigned int nblk = (unsigned int)((n + SEL_CHUNK - 1) / SEL_CHUNK);
  hipLaunchKernelGGL(sel_init_kernel, dim3(1), dim3(256), 0, s, st, K);
  for (int shift = 24; shift >= 0; shift -= 8) {
    hipLaunchKernelGGL(sel_hist_kernel, dim3(nblk), dim3(256), 0, s, conf, n, st, shift);
    hipLaunchKernelGGL(sel_pick_kernel, dim3(1), dim3(256), 0, s, st, shift);
  }
  hipLaunchKernelGGL(sel_count_eq_kernel, dim3(nblk), dim3(256), 0, s, conf, n, (const SelState*)st, blk_eq);
  hipLaunchKernelGGL(sel_scan_kernel, dim3(1), dim3(256), 0, s, blk_eq, nblk);
  hipLaunchKernelGGL(sel_mask_kernel, dim3(nblk), dim3(256), 0, s, conf, n, (const SelState*)st, (const unsigned int*)blk_eq, mask);
  return hipGetLastError();
}
