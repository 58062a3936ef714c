// Micro-benchmark: do MFMA (wave A) and VALU (wave B) of the SAME SIMD overlap?  8 waves per block: waves 0-3 = role A,
// waves 4-7 = role B (wave w and w+4 share SIMD w%4).  mode bit0: A runs MFMAs, bit1: B runs VALU, bit2: A runs both
// interleaved in its own stream (B idle).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 b16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(512, 1) void k(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  const bool roleA = wave < 4;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  b16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.5f); }
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 0.01f + j;
  if (roleA && (mode & 1)) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
  } else if (roleA && (mode & 4)) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
          // 4 exp + 4 add per MFMA  (issue cost 4*8 + 4*4 = 48 > the 24 free cycles of a 32-cycle gap)
          const int o = (u * 4 + i) & 3;
#pragma unroll
          for (int q = 0; q < 4; ++q) { v[o * 4 + q] = __builtin_amdgcn_exp2f(v[o * 4 + q]); v[o * 4 + q] += 1.0f; }
        }
    }
  } else if (!roleA && (mode & 2)) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 16; ++j) { v[j] = __builtin_amdgcn_exp2f(v[j]); v[j] += 1.0f; }   // 64 exp + 64 add per iteration
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int j = 0; j < 16; ++j) s += v[j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode : {1, 2, 3, 4, 1, 2, 3, 4}) {
    k<<<256, 512>>>(out, 100, mode); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<256, 512>>>(out, iters, mode); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per iteration: A = 16 MFMA (512 cycles of matrix pipe), B = 64 exp + 64 add (64*8 + 64*4 = 768 issue cycles)
    printf("mode %d: %.3f ms  -> %.1f ns/iter = %.0f cycles/iter @2.4GHz\n", mode, ms, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
  }
  return 0;
}
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o tools/micro/coexec tools/micro/coexec.hip && tools/micro/coexec
