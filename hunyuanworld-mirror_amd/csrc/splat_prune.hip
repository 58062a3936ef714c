// Voxel merge of the per-pixel Gaussian splats — GaussianSplatRenderer.prune_gs (src/models/models/rasterization.py:301-387),
// the only cross-view step of the 3D-Gaussian branch.  HBM-bound integer / fp32 work:
//   keys     voxel = floor(mean / voxel_size) per axis (true fp32 division, as the reference), min / max per axis by
//            64-bit integer atomics, flat key = ((x - minx) * dimy + (y - miny)) * dimz + (z - minz)       (:320-331)
//   sort     stable 64-bit radix sort of (key, index) (hipCUB) — ascending keys is torch.unique's output order (:334)
//   heads    first element of every run of equal keys, exclusive scan -> voxel number and run boundaries
//   merge    one thread per voxel walks its run IN ORIGINAL INDEX ORDER (the sort is stable): weighted sums of means, sh,
//            scales, quats, sum of w and of w^2, then the reference's normalisations (:341-376).  The reference's
//            scatter_add_ adds in index order on the CPU, so the sums here are bit-identical to its CPU result.
#include "wm_common.h"
#include "wm_kernels.h"

#include <hipcub/hipcub.hpp>

namespace {

typedef long long i64;
typedef unsigned long long u64;

__global__ __launch_bounds__(256) void prune_minmax_kernel(const float* __restrict__ means, int N, float voxel, i64* __restrict__ mm) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  i64 v[3] = {0, 0, 0};
  const bool ok = i < N;
  if (ok) {
#pragma unroll
    for (int d = 0; d < 3; ++d) v[d] = (i64)floorf(means[3 * i + d] / voxel);
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {  // wave reduction first: 6 atomics per wave instead of per thread
    i64 lo = ok ? v[d] : LLONG_MAX, hi = ok ? v[d] : LLONG_MIN;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const i64 l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
      lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&mm[d], lo); atomicMax(&mm[3 + d], hi); }
  }
}

__global__ __launch_bounds__(256) void prune_keys_kernel(const float* __restrict__ means, int N, float voxel, const i64* __restrict__ mm,
                                                         u64* __restrict__ keys, unsigned int* __restrict__ idx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const i64 dy = mm[4] - mm[1] + 1, dz = mm[5] - mm[2] + 1;
  const i64 x = (i64)floorf(means[3 * i] / voxel) - mm[0], y = (i64)floorf(means[3 * i + 1] / voxel) - mm[1], z = (i64)floorf(means[3 * i + 2] / voxel) - mm[2];
  keys[i] = (u64)(x * dy * dz + y * dz + z);
  idx[i] = (unsigned int)i;
}

__global__ __launch_bounds__(256) void prune_heads_kernel(const u64* __restrict__ keys, int N, unsigned int* __restrict__ head) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// run starts: start[voxel] = position of its head; start[K] = N
__global__ __launch_bounds__(256) void prune_starts_kernel(const unsigned int* __restrict__ head, const unsigned int* __restrict__ excl, int N,
                                                           unsigned int* __restrict__ start) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i > N) return;
  if (i == N) { start[excl[N]] = (unsigned int)N; return; }
  if (head[i]) start[excl[i]] = (unsigned int)i;
}

__global__ __launch_bounds__(256) void prune_merge_kernel(const float* __restrict__ means, const float* __restrict__ quats, const float* __restrict__ scales,
                                                          const float* __restrict__ sh, const float* __restrict__ weights,
                                                          const unsigned int* __restrict__ order, const unsigned int* __restrict__ start,
                                                          const unsigned int* __restrict__ K_dev, float* __restrict__ o_means, float* __restrict__ o_quats,
                                                          float* __restrict__ o_scales, float* __restrict__ o_opac, float* __restrict__ o_sh) {
#pragma clang fp contract(off)  // the reference multiplies, rounds, then adds (x * w, scatter_add_): no fused multiply-add here
  const unsigned int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= *K_dev) return;
  float m[3] = {0, 0, 0}, q[4] = {0, 0, 0, 0}, s[3] = {0, 0, 0}, c[3] = {0, 0, 0}, ws = 0.f, w2 = 0.f;
  for (unsigned int p = start[k]; p < start[k + 1]; ++p) {
    const unsigned int i = order[p];
    const float w = weights[i];
    ws += w; w2 += w * w;
#pragma unroll
    for (int d = 0; d < 3; ++d) { m[d] += means[3 * i + d] * w; s[d] += scales[3 * i + d] * w; c[d] += sh[3 * i + d] * w; }
#pragma unroll
    for (int d = 0; d < 4; ++d) q[d] += quats[4 * i + d] * w;
  }
  ws = fmaxf(ws, 1e-8f);
#pragma unroll
  for (int d = 0; d < 3; ++d) { o_means[3 * k + d] = m[d] / ws; o_scales[3 * k + d] = s[d] / ws; o_sh[3 * k + d] = c[d] / ws; }
  o_opac[k] = w2 / ws;
  const float qn = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-8f);
#pragma unroll
  for (int d = 0; d < 4; ++d) o_quats[4 * k + d] = q[d] / qn;
}

size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct PruneWs { i64* mm; u64* keys[2]; unsigned int* idx[2]; unsigned int* head; unsigned int* excl; unsigned int* start; void* cub; size_t cub_bytes, total; };

PruneWs carve(char* base, size_t N) {
  PruneWs w;
  size_t o = 0;
  auto take = [&](size_t b) { char* p = base ? base + o : nullptr; o += al256(b); return p; };
  w.mm = (i64*)take(6 * 8);
  w.keys[0] = (u64*)take(N * 8); w.keys[1] = (u64*)take(N * 8);
  w.idx[0] = (unsigned int*)take(N * 4); w.idx[1] = (unsigned int*)take(N * 4);
  w.head = (unsigned int*)take((N + 1) * 4); w.excl = (unsigned int*)take((N + 1) * 4); w.start = (unsigned int*)take((N + 2) * 4);
  size_t a = 0, b = 0;
  hipcub::DoubleBuffer<u64> dk((u64*)nullptr, (u64*)nullptr);
  hipcub::DoubleBuffer<unsigned int> dv((unsigned int*)nullptr, (unsigned int*)nullptr);
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, dk, dv, (int)(N ? N : 1), 0, 64);
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (unsigned int*)nullptr, (unsigned int*)nullptr, (int)(N + 1));
  w.cub_bytes = a > b ? a : b;
  w.cub = take(w.cub_bytes);
  w.total = o;
  return w;
}

}  // namespace

size_t wm_prune_workspace_bytes(size_t N) { return carve(nullptr, N).total; }

// outputs have room for N rows; *K_out (host) = number of occupied voxels.  One stream synchronisation (K sizes the outputs
// the caller slices, as len(unique_voxels) does in the reference).
hipError_t wm_launch_prune_gs(const float* means, const float* quats, const float* scales, const float* opac, const float* sh,
                              const float* weights, int N, float voxel, float* o_means, float* o_quats, float* o_scales, float* o_opac,
                              float* o_sh, int* K_out, void* workspace, size_t ws_bytes, hipStream_t s) {
  (void)opac;  // the merged opacity is sum(w^2) / sum(w): the input opacities are not read by the reference either (:359-361)
  if (N <= 0) { if (K_out) *K_out = 0; return hipSuccess; }
  PruneWs w = carve((char*)workspace, (size_t)N);
  if (w.total > ws_bytes) return hipErrorInvalidValue;
  const i64 init[6] = {LLONG_MAX, LLONG_MAX, LLONG_MAX, LLONG_MIN, LLONG_MIN, LLONG_MIN};
  hipError_t e = hipMemcpyAsync(w.mm, init, sizeof(init), hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return e;
  const unsigned nb = (unsigned)((N + 255) / 256);
  hipLaunchKernelGGL(prune_minmax_kernel, dim3(nb), dim3(256), 0, s, means, N, voxel, w.mm);
  hipLaunchKernelGGL(prune_keys_kernel, dim3(nb), dim3(256), 0, s, means, N, voxel, w.mm, w.keys[0], w.idx[0]);
  hipcub::DoubleBuffer<u64> dk(w.keys[0], w.keys[1]);
  hipcub::DoubleBuffer<unsigned int> dv(w.idx[0], w.idx[1]);
  size_t tb = w.cub_bytes;
  e = hipcub::DeviceRadixSort::SortPairs(w.cub, tb, dk, dv, N, 0, 64, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(prune_heads_kernel, dim3(nb), dim3(256), 0, s, dk.Current(), N, w.head);
  e = hipMemsetAsync(w.head + N, 0, 4, s);
  if (e != hipSuccess) return e;
  tb = w.cub_bytes;
  e = hipcub::DeviceScan::ExclusiveSum(w.cub, tb, w.head, w.excl, N + 1, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(prune_starts_kernel, dim3((unsigned)((N + 1 + 255) / 256)), dim3(256), 0, s, w.head, w.excl, N, w.start);
  hipLaunchKernelGGL(prune_merge_kernel, dim3(nb), dim3(256), 0, s, means, quats, scales, sh, weights, dv.Current(), w.start, w.excl + N, o_means, o_quats,
                     o_scales, o_opac, o_sh);
  unsigned int K = 0;
  e = hipMemcpyAsync(&K, w.excl + N, 4, hipMemcpyDeviceToHost, s);
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(s);
  if (e != hipSuccess) return e;
  if (K_out) *K_out = (int)K;
  return hipGetLastError();
}
