// Shared device/host helpers for the WorldMirror HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
// Packed-fp32 VALU instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) are kept out of every kernel except the GELU GEMM
// instantiations of gemm.hip: with two or more HIP queues active such an instruction can lose a half result
// (profiles/r02_multiqueue_hazard.md), and the forward runs the DPT heads, the camera head and the sharded K/V all-gather on their own
// queues.  How: the build's -fno-slp-vectorize removes the compiler-formed ones; WM_NO_PACKED_FP32 on a function removes what is left
// where vector-typed source still packs (a whole-translation-unit attribute was tried: it also doubled the v_mov_b32 count of the
// conv kernels, -13 % on the 148^2 convs); tests/test_kernel_resources_cpu.py compiles every object to assembly and holds the line.
#if defined(__HIP_DEVICE_COMPILE__)
#define WM_NO_PACKED_FP32 __attribute__((target("no-packed-fp32-ops")))
#else
#define WM_NO_PACKED_FP32
#endif

typedef unsigned short u16;
typedef __attribute__((ext_vector_type(8))) short s16x8;     // 8 x 16-bit MFMA A/B fragment
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;  // same, typed for the f16 builtin
typedef __attribute__((ext_vector_type(8))) __bf16 b16x8;    // same, typed for the bf16 builtin
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;   // 32x32 accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;

// 16-bit operand element types understood by the kernels
enum { WM_T_BF16 = 0, WM_T_F16 = 1 };

__device__ __forceinline__ u16 f2bf(float x) {  // round-to-nearest-even, NaN preserved by the cast
  __bf16 b = (__bf16)x;
  return __builtin_bit_cast(u16, b);
}
__device__ __forceinline__ float bf2f(u16 v) { return __builtin_bit_cast(float, ((uint32_t)v) << 16); }
// fp32 -> f16 SATURATES at +-65504 (one v_med3_f32): the reference's DPT heads are fp32 (worldmirror.py:146) and an unbounded
// ReLU / residual chain of a real checkpoint may exceed f16's range; an inf operand would turn a whole MFMA row into NaN.
// A NaN stays a NaN (v_med3 alone would order it low and return -65504: an upstream fault would come out as plausible finite
// values where the reference's fp32 heads return NaN): one v_cmp + v_cndmask more.
__device__ __forceinline__ u16 f2h(float x) {
  const float c = __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f);
  x = x != x ? x : c;
  _Float16 h = (_Float16)x;
  return __builtin_bit_cast(u16, h);
}
__device__ __forceinline__ float h2f(u16 v) { return (float)__builtin_bit_cast(_Float16, v); }

template <int T> __device__ __forceinline__ u16 f2t(float x) { return T == WM_T_BF16 ? f2bf(x) : f2h(x); }
template <int T> __device__ __forceinline__ float t2f(u16 v) { return T == WM_T_BF16 ? bf2f(v) : h2f(v); }

// D(32x32 f32) += A(32x16) * B(16x32); lane l: A[row l&31][k=8(l>>5)+j], B[k=8(l>>5)+j][col l&31];
// D: col = l&31, row = (r&3) + 8(r>>2) + 4(l>>5)  (guides/cdna_hip_programming.md §3)
template <int T>
__device__ __forceinline__ f32x16 mfma32(s16x8 a, s16x8 b, f32x16 c) {
  if constexpr (T == WM_T_BF16)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b16x8, a), __builtin_bit_cast(b16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}

// D(16x16 f32) += A(16x32) * B(32x16); lane l: A[row l&15][k=8(l>>4)+j], B[k=8(l>>4)+j][col l&15];
// D: col = l&15, row = 4(l>>4) + r  (guides §3)
template <int T>
__device__ __forceinline__ f32x4 mfma16(s16x8 a, s16x8 b, f32x4 c) {
  if constexpr (T == WM_T_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b16x8, a), __builtin_bit_cast(b16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}

// exact-erf GELU (nn.GELU() default, mlp.py:17).  erf via Abramowitz-Stegun 7.1.26 (|abs err| < 1.5e-7, far
// below the 16-bit output rounding) on v_rcp/v_exp: ~12 VALU ops instead of libm erff's ~40 — the GELU epilogue
// was 26 % of the fc1 GEMM's wave time.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
  const float r = fmaf(-p * t, e, 1.0f);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f)); }

// The same GELU on two values at a time with packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32: 7.5 full-rate VALU per element
// instead of ~17): s = |x| / sqrt 2 is folded into the constants, and with hx = x / 2, pte = p t exp(-x^2 / 2)
//   gelu = hx (1 + sign(x) (1 - pte)) = (hx + |hx|) - |hx| pte        (no cancellation for x < 0: hx + |hx| = 0 exactly)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t x) {
  auto k = [](float c) { return f32x2_t{c, c}; };
  const f32x2_t ax = __builtin_elementwise_abs(x);
  const f32x2_t d = __builtin_elementwise_fma(ax, k(0.3275911f * 0.70710678118654752440f), k(1.0f));
  const f32x2_t t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  f32x2_t p = __builtin_elementwise_fma(t, k(1.061405429f), k(-1.453152027f));
  p = __builtin_elementwise_fma(p, t, k(1.421413741f));
  p = __builtin_elementwise_fma(p, t, k(-0.284496736f));
  p = __builtin_elementwise_fma(p, t, k(0.254829592f));
  const f32x2_t q = x * x * k(-0.5f * 1.4426950408889634f);
  const f32x2_t e = {__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
  const f32x2_t hx = x * k(0.5f), ah = ax * k(0.5f);
  return __builtin_elementwise_fma(-ah, p * t * e, hx + ah);
}
// WM_UNPACKED_GELU (make EXTRA=-DWM_UNPACKED_GELU): the scalar form, no packed-fp32 instruction anywhere in the library — for a
// deployment that drives several handles on one device from different threads / streams (wm_share_weights) and wants the
// multi-queue guarantee of profiles/r02_multiqueue_hazard.md for the fc1 GEMM too (~ +0.4 ms per 8-view forward).
__device__ __forceinline__ float4 gelu_erf4(float4 x) {
#ifdef WM_UNPACKED_GELU
  return make_float4(gelu_erf(x.x), gelu_erf(x.y), gelu_erf(x.z), gelu_erf(x.w));
#else
  const f32x2_t a = gelu_erf2(f32x2_t{x.x, x.y}), b = gelu_erf2(f32x2_t{x.z, x.w});
  return make_float4(a.x, a.y, b.x, b.y);
#endif
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// Combine a value with the one held by lane ^ 32.  v_permlane32_swap exchanges lanes 32-63 of its
// first operand with lanes 0-31 of its second; inline asm on two distinct registers because hipcc
// folds the builtin's two results when both inputs are the same SSA value.  The s_nop covers the
// VALU-write -> permlane-read hazard (guides T21).
__device__ __forceinline__ void xhalf_pair(float x, float& a, float& b) {
  a = x; b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float xhalf_max(float x) { float a, b; xhalf_pair(x, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float xhalf_sum(float x) { float a, b; xhalf_pair(x, a, b); return a + b; }

// v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of its second.  With
// a = this lane's packed columns of sub-tile j0 and b = those of sub-tile j1 (MFMA 16x16 D layout: lane = (row l15,
// quad lq) owns 4 consecutive columns), afterwards an even-lq lane holds {a, b} = 8 consecutive columns of j0
// (its own 4 + those of lq + 1) and an odd-lq lane 8 consecutive columns of j1 (those of lq - 1 + its own): 16-B
// stores of 64-B row segments instead of 8-B stores of 32-B segments.  Every lane of the wave must be active.
__device__ __forceinline__ void swap16(uint32_t& a, uint32_t& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// XCD-aware bijective block remap (guides T1): blocks b and b+8 share an XCD; give each XCD a
// contiguous chunk of the logical tile space so neighbouring tiles hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

#define WM_CHECK_HIP(expr)                                                            \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) return wm_fail(hipGetErrorString(_e), __FILE__, __LINE__);  \
  } while (0)

