// A/B of the two bf16 MFMA shapes inside the cross-view attention loop's instruction mix (guides/MI355X_MICROARCH.md, DVFS
// give-back item 7: under the power limit the chip may hold a higher clock on 16x16x32 than on 32x32x16).
//
// Not a kernel of the product: a register-resident loop with exactly the per-gap mix of attn_v4_kernel (attention_v4.hip) —
// 2 v_exp + 2 v_add + 1 v_cvt_pk per 32x32x16-MFMA-worth of flops, half the MFMAs writing score tiles in arch VGPRs (read by
// the exps half a step later), half accumulating into AGPRs — once with one 32x32x16 MFMA per gap and once with two 16x16x32
// MFMAs per gap (same flops, same accumulator registers).  No LDS, no global traffic inside the loop: what differs is the MFMA
// shape only.  wave 0 of every block stamps s_memtime / s_memrealtime around the loop.
//   build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/micro/libattn_loop_shapes.so tools/micro/attn_loop_shapes.hip
//   run:   python tools/attn_loop_shapes.py
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ORDER (build-time A/B of the statement's instruction order; 0 is what attention_v4.hip uses):
//   0: exp exp MFMA add add pack    1: exp MFMA exp add add pack    2: MFMA exp exp add add pack    3: exp MFMA add exp add pack
#ifndef ORDER
#define ORDER 0
#endif
#ifndef XOR_SINK
#define XOR_SINK 0
#endif
#define SM_HEAD "v_exp_f32 %[e0], %[s0]\n\tv_exp_f32 %[e1], %[s1]\n\t"
#define SM_OUT [l0] "+v"(l0), [l1] "+v"(l1), [p] "=&v"(p), [e0] "+v"(e0), [e1] "+v"(e1)
#define SM_IN [s0] "v"(s0), [s1] "v"(s1), [ones] "v"(0x3f803f80u)

#define EXP0 "v_exp_f32 %[e0], %[s0]\n\t"
#define EXP1 "v_exp_f32 %[e1], %[s1]\n\t"
#define ADD0 "v_add_f32 %[l0], %[l0], %[e0]\n\t"
#define ADD1 "v_add_f32 %[l1], %[l1], %[e1]\n\t"
#define PACK "v_cvt_pk_bf16_f32 %[p], %[e0], %[e1]"
#if ORDER == 0
#define GAP32(M) EXP0 EXP1 M "\n\t" ADD0 ADD1 PACK
#elif ORDER == 1
#define GAP32(M) EXP0 M "\n\t" EXP1 ADD0 ADD1 PACK
#elif ORDER == 2
#define GAP32(M) M "\n\t" EXP0 EXP1 "s_nop 0\n\t" ADD0 ADD1 PACK
#elif ORDER == 3
#define GAP32(M) EXP0 M "\n\t" ADD0 EXP1 "s_nop 0\n\t" ADD1 PACK
#elif ORDER == 4   // pricing: no pack (p = e0 moved)
#define GAP32(M) EXP0 EXP1 M "\n\t" ADD0 ADD1 "v_mov_b32 %[p], %[e0]"
#elif ORDER == 5   // pricing: one add only
#define GAP32(M) EXP0 EXP1 M "\n\t" ADD0 PACK
#elif ORDER == 6   // pricing: one exp only (e1 = s1 moved)
#define GAP32(M) EXP0 "v_mov_b32 %[e1], %[s1]\n\t" M "\n\t" ADD0 ADD1 PACK
#elif ORDER == 7   // pricing: no MFMA
#define GAP32(M) EXP0 EXP1 ADD0 ADD1 PACK
#elif ORDER == 9   // pricing: consumers of the exp results replaced by independent moves
#define GAP32(M) EXP0 EXP1 M "\n\tv_mov_b32 %[l0], %[s0]\n\tv_mov_b32 %[l1], %[s1]\n\tv_mov_b32 %[p], %[s0]"
#elif ORDER == 10  // pricing: no transcendental at all
#define GAP32(M) "v_mov_b32 %[e0], %[s0]\n\tv_mov_b32 %[e1], %[s1]\n\t" M "\n\t" ADD0 ADD1 PACK
#elif ORDER == 12  // candidate: the row sum from the PACKED words (one v_dot2c with a pair of ones instead of two adds)
#define GAP32(M) EXP0 EXP1 M "\n\t" PACK "\n\tv_dot2c_f32_bf16 %[l0], %[p], %[ones]"
#elif ORDER == 13  // the same, the dot one gap late is not expressible here; dot before the MFMA's successor: exp exp pack MFMA dot
#define GAP32(M) EXP0 EXP1 PACK "\n\t" M "\n\tv_dot2c_f32_bf16 %[l0], %[p], %[ones]"
#elif ORDER == 11  // pricing: MFMA + five moves (no dependency between any two instructions of the gap)
#define GAP32(M) "v_mov_b32 %[e0], %[s0]\n\tv_mov_b32 %[e1], %[s1]\n\t" M "\n\tv_mov_b32 %[l0], %[s0]\n\tv_mov_b32 %[l1], %[s1]\n\tv_mov_b32 %[p], %[s0]"
#endif
// ---- 32x32x16: one MFMA per gap
__device__ __forceinline__ void gap32_first(f32x16& d, const s16x8& a, const s16x8& b, float s0, float s1, float& l0, float& l1, uint32_t& p, float& e0, float& e1) {
  asm volatile(GAP32("v_mfma_f32_32x32x16_bf16 %[d], %[a], %[b], 0")
               : [d] "=&v"(d), SM_OUT : [a] "v"(a), [b] "a"(b), SM_IN);
}
__device__ __forceinline__ void gap32_next(f32x16& d, const s16x8& a, const s16x8& b, float s0, float s1, float& l0, float& l1, uint32_t& p, float& e0, float& e1) {
  asm volatile(GAP32("v_mfma_f32_32x32x16_bf16 %[d], %[a], %[b], %[d]")
               : [d] "+v"(d), SM_OUT : [a] "v"(a), [b] "a"(b), SM_IN);
}
__device__ __forceinline__ void gap32_acc(f32x16& c, const s16x8& a, const s16x8& b, float s0, float s1, float& l0, float& l1, uint32_t& p, float& e0, float& e1) {
  asm volatile(GAP32("v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]")
               : [c] "+a"(c), SM_OUT : [a] "v"(a), [b] "v"(b), SM_IN);
}
// ---- 16x16x32: two MFMAs per gap (a score tile's two k-steps / two accumulator tiles)
__device__ __forceinline__ void gap16_qk(f32x4& d, const s16x8& a0, const s16x8& b0, const s16x8& a1, const s16x8& b1, float s0, float s1, float& l0, float& l1, uint32_t& p, float& e0, float& e1) {
  asm volatile(SM_HEAD "v_mfma_f32_16x16x32_bf16 %[d], %[a0], %[b0], 0\n\tv_add_f32 %[l0], %[l0], %[e0]\n\tv_mfma_f32_16x16x32_bf16 %[d], %[a1], %[b1], %[d]\n\tv_add_f32 %[l1], %[l1], %[e1]\n\tv_cvt_pk_bf16_f32 %[p], %[e0], %[e1]"
               : [d] "=&v"(d), SM_OUT : [a0] "v"(a0), [b0] "a"(b0), [a1] "v"(a1), [b1] "a"(b1), SM_IN);
}
__device__ __forceinline__ void gap16_acc(f32x4& c0, f32x4& c1, const s16x8& a0, const s16x8& a1, const s16x8& b, float s0, float s1, float& l0, float& l1, uint32_t& p, float& e0, float& e1) {
  asm volatile(SM_HEAD "v_mfma_f32_16x16x32_bf16 %[c0], %[a0], %[b], %[c0]\n\tv_add_f32 %[l0], %[l0], %[e0]\n\tv_mfma_f32_16x16x32_bf16 %[c1], %[a1], %[b], %[c1]\n\tv_add_f32 %[l1], %[l1], %[e1]\n\tv_cvt_pk_bf16_f32 %[p], %[e0], %[e1]"
               : [c0] "+a"(c0), [c1] "+a"(c1), SM_OUT : [a0] "v"(a0), [a1] "v"(a1), [b] "v"(b), SM_IN);
}

// ---- bare MFMA loops (no softmax beside them): what the matrix pipes deliver on random operands under the board's power limit
__device__ __forceinline__ void bare32(f32x16& c, const s16x8& a, const s16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]" : [c] "+a"(c) : [a] "v"(a), [b] "v"(b));
}
__device__ __forceinline__ void bare16(f32x4& c, const s16x8& a, const s16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %[c], %[a], %[b], %[c]" : [c] "+a"(c) : [a] "v"(a), [b] "v"(b));
}
template <int SHAPE>
__global__ __launch_bounds__(256, 1) void bare_kernel(const uint4* __restrict__ ab, float* __restrict__ sink, unsigned long long* __restrict__ stamps, int iters) {
  const int tid = threadIdx.x;
  s16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = __builtin_bit_cast(s16x8, ab[(i * 256 + tid) & 4095]);
    b[i] = __builtin_bit_cast(s16x8, ab[((4 + i) * 256 + tid) & 4095]);
  }
  unsigned long long c0 = 0, r0 = 0;
  float acc = 0.f;
  if constexpr (SHAPE == 0) {
    f32x16 O[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[t][r] = 0.f;
    if (tid == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int g = 0; g < 32; ++g) bare32(O[g & 7], a[g & 3], b[(g >> 2) & 3]);
    if (tid == 0) { c0 = __builtin_amdgcn_s_memtime() - c0; r0 = __builtin_amdgcn_s_memrealtime() - r0; }
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(O[0]), "+a"(O[1]), "+a"(O[2]), "+a"(O[3]), "+a"(O[4]), "+a"(O[5]), "+a"(O[6]), "+a"(O[7]));
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc += O[t][r];
  } else {
    f32x4 O[32];
#pragma unroll
    for (int t = 0; t < 32; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) O[t][r] = 0.f;
    if (tid == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int g = 0; g < 64; ++g) bare16(O[g & 31], a[g & 3], b[(g >> 2) & 3]);
    if (tid == 0) { c0 = __builtin_amdgcn_s_memtime() - c0; r0 = __builtin_amdgcn_s_memrealtime() - r0; }
    asm volatile("s_nop 15" ::: "memory");
#pragma unroll
    for (int t = 0; t < 32; ++t) { asm volatile("" : "+a"(O[t])); 
#pragma unroll
      for (int r = 0; r < 4; ++r) acc += O[t][r]; }
  }
  sink[blockIdx.x * 256 + tid] = acc;
  if (tid == 0) { stamps[blockIdx.x * 2] = c0; stamps[blockIdx.x * 2 + 1] = r0; }
}

template <int SHAPE, int MINW>
__global__ __launch_bounds__(256, MINW) void loop_kernel(const uint4* __restrict__ ab, float* __restrict__ sink, unsigned long long* __restrict__ stamps, int iters) {
  const int tid = threadIdx.x;
  s16x8 a[4], b[4], q[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = __builtin_bit_cast(s16x8, ab[(i * 256 + tid) & 4095]);
    b[i] = __builtin_bit_cast(s16x8, ab[((4 + i) * 256 + tid) & 4095]);
    q[i] = __builtin_bit_cast(s16x8, ab[((8 + i) * 256 + tid) & 4095]);
  }
  float l[4] = {0.f, 0.f, 0.f, 0.f}, et[4] = {0.f, 0.f, 0.f, 0.f};
  uint32_t acc_p = 0, w0 = 0, w1 = 0;
  s16x8 pf[2] = {a[0], a[1]};
  unsigned long long c0 = 0, r0 = 0;
  if constexpr (SHAPE == 0) {
    f32x16 S[4], O[8];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) S[t][r] = 0.01f * (float)((tid * 7 + t * 16 + r) % 200) - 1.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[t][r] = 0.f;
    if (tid == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
#ifndef REP
#define REP 1     // copies of the 32-gap body per loop iteration (prices the taken branch at the loop's end: one wave per SIMD has nobody to hide it)
#endif
    for (int it = 0; it < iters; it += REP) {
#pragma unroll
      for (int g4 = 0; g4 < 4 * REP; ++g4) {        // per score tile
        const int g = g4 & 3;              // its 4 k-steps alternate with 4 accumulating MFMAs (consecutive statements share no
        const int rd = (g + 2) & 3;        // register one of them writes: hipcc pads a wait state between statements that do); exps read the
        uint32_t u[8];                     // tile written two tiles ago
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i == 0) gap32_first(S[g], a[0], q[0], S[rd][4 * i], S[rd][4 * i + 1], l[0], l[1], u[2 * i], et[0], et[1]);
          else gap32_next(S[g], a[i], q[i], S[rd][4 * i], S[rd][4 * i + 1], l[0], l[1], u[2 * i], et[0], et[1]);
          gap32_acc(O[2 * g + (i & 1)], b[i], pf[i >> 1], S[rd][4 * i + 2], S[rd][4 * i + 3], l[2], l[3], u[2 * i + 1], et[2], et[3]);
        }
#if XOR_SINK   // rounds 3's first rows: the packed words kept alive by one v_xor per gap (an instruction the attention loop does not have)
        acc_p ^= u[0] ^ u[1] ^ u[2] ^ u[3] ^ u[4] ^ u[5] ^ u[6] ^ u[7];
#else          // as in attention_v4.hip: the packed words ARE the next tile's MFMA operand, nothing else reads them
        pf[0] = __builtin_bit_cast(s16x8, make_uint4(u[0], u[2], u[4], u[6]));
        pf[1] = __builtin_bit_cast(s16x8, make_uint4(u[1], u[3], u[5], u[7]));
#endif
      }
    }
    if (tid == 0) { c0 = __builtin_amdgcn_s_memtime() - c0; r0 = __builtin_amdgcn_s_memrealtime() - r0; }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc += O[t][r];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc += S[t][3];
    sink[blockIdx.x * 256 + tid] = acc + l[0] + l[1] + l[2] + l[3] + (float)((acc_p ^ w0 ^ w1 ^ (uint32_t)pf[0][0] ^ (uint32_t)pf[1][1]) & 3);
  } else {
    f32x4 S[16], O[32];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) S[t][r] = 0.01f * (float)((tid * 7 + t * 4 + r) % 200) - 1.0f;
#pragma unroll
    for (int t = 0; t < 32; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) O[t][r] = 0.f;
    if (tid == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {       // 16 pairs of gaps: a score tile (two k-steps), then two accumulator tiles
        const int rd = (g + 8) & 15;       // score tile read by this pair's exps: written half a step ago
        uint32_t u0, u1;
        gap16_qk(S[g], a[g & 3], q[g & 3], a[(g + 1) & 3], q[(g + 1) & 3], S[rd][0], S[rd][1], l[0], l[1], u0, et[0], et[1]);
        gap16_acc(O[2 * g], O[2 * g + 1], b[g & 3], b[(g + 2) & 3], pf[0], S[rd][2], S[rd][3], l[2], l[3], u1, et[2], et[3]);
#if XOR_SINK
        acc_p ^= u0 ^ u1;
#else
        if (g & 1) pf[0] = __builtin_bit_cast(s16x8, make_uint4(w0, w1, u0, u1));
        else { w0 = u0; w1 = u1; }
#endif
      }
    }
    if (tid == 0) { c0 = __builtin_amdgcn_s_memtime() - c0; r0 = __builtin_amdgcn_s_memrealtime() - r0; }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 32; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc += O[t][r];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc += S[t][3];
    sink[blockIdx.x * 256 + tid] = acc + l[0] + l[1] + l[2] + l[3] + (float)((acc_p ^ w0 ^ w1 ^ (uint32_t)pf[0][0] ^ (uint32_t)pf[1][1]) & 3);
  }
  if (tid == 0) { stamps[blockIdx.x * 2] = c0; stamps[blockIdx.x * 2 + 1] = r0; }
}

// shape 0 = 32x32x16, 1 = 16x16x32 (attention mix); 2 / 3 = the same MFMAs bare; waves_per_simd 1 or 2 (blocks of 256 threads per CU); grid = blocks
extern "C" int attn_loop_launch(int shape, int waves_per_simd, int blocks, int iters, const void* ab, float* sink, unsigned long long* stamps, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (shape == 0 && waves_per_simd == 1) hipLaunchKernelGGL((loop_kernel<0, 1>), dim3(blocks), dim3(256), 0, s, (const uint4*)ab, sink, stamps, iters);
  else if (shape == 1 && waves_per_simd == 1) hipLaunchKernelGGL((loop_kernel<1, 1>), dim3(blocks), dim3(256), 0, s, (const uint4*)ab, sink, stamps, iters);
  else if (shape == 2) hipLaunchKernelGGL((bare_kernel<0>), dim3(blocks), dim3(256), 0, s, (const uint4*)ab, sink, stamps, iters);
  else if (shape == 3) hipLaunchKernelGGL((bare_kernel<1>), dim3(blocks), dim3(256), 0, s, (const uint4*)ab, sink, stamps, iters);
  else return -1;
  return (int)hipGetLastError();
}
