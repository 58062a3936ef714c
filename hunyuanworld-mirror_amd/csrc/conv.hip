// NHWC implicit-GEMM convolution (k x k, stride 1/2, zero pad) on 16-bit MFMA with fp32 activations.
//
// Replaces every spatial nn.Conv2d of the DPT heads (reference: src/models/heads/dense_head.py:
// resize_layers[3] :65-67, scratch.layer*_rn :394-406, ResidualConvUnit :426-455, out_conv :496,
// output_conv1/2 :97-105).  Activations stay fp32 NHWC in HBM (the reference runs the heads in fp32,
// worldmirror.py:146; SURVEY §7 "hard parts": 16-bit inter-layer storage breaks the point-map
// tolerance); they are converted to f16/bf16 while being staged into LDS, with the RCU's ReLU
// applied on the fly.  GEMM view: M = N*Ho*Wo output pixels, N = Cout, K = k*k*Cin with the weight
// repacked [Cout][ky][kx][Cin] so that one K-tile is one tap x BK consecutive channels.
//
// Epilogue: y = acc + bias (+ relu?(resid)) (+ resid2) — covers conv2 of an RCU including the
// in-place-ReLU skip and the fusion block's "x + RCU1(skip)".
#include "wm_common.h"
#include "wm_kernels.h"

#include <cstdlib>

namespace {

typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

template <int BK>
__device__ __forceinline__ int swz(int row) {  // 16-B slot permutation making ds_read_b128 conflict-free
  return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3);
}

template <int BK>
__device__ __forceinline__ s16x8 frag(const char* tile, int row, int chunk) {
  return *(const s16x8*)(tile + row * (BK * 2) + ((chunk ^ swz<BK>(row)) << 4));
}

// weight tile [BN rows][BK] via LDS-DMA, swizzle on the source address
template <int BK, int BN>
__device__ __forceinline__ void stage_w(const u16* __restrict__ w, int K, int n0, int Cout, int k0, char* tile, int wave,
                                        int lane) {
  constexpr int CPR = BK / 8;          // 16-B chunks per row
  constexpr int RPI = 64 / CPR;        // rows per wave-instruction
  constexpr int NI = BN / RPI;
  for (int i = wave; i < NI; i += 4) {
    const int r = i * RPI + lane / CPR;
    const int c = (lane % CPR) ^ swz<BK>(r);
    int gr = n0 + r;
    gr = gr < Cout ? gr : Cout - 1;
    __builtin_amdgcn_global_load_lds((glb_vp)(w + (size_t)gr * K + k0 + c * 8), (lds_vp)(tile + i * 1024), 16, 0, 0);
  }
}

template <int T, int BK, int WGM, int WGN, int NI, int NJ>
__global__ __launch_bounds__(256) void conv_kernel(const WmConvArgs p) {
  constexpr int BM = WGM * NI * 32, BN = WGN * NJ * 32;
  constexpr int CPR = BK / 8;                 // chunks per row
  constexpr int APT = BM * CPR / 256;         // A chunks (8 channels) per thread per tile
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A | B]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int M = p.N * p.Ho * p.Wo;
  const int K = p.ksize * p.ksize * p.Cin;
  const int ntn = (p.Cout + BN - 1) / BN, ntm = (M + BM - 1) / BM;
  const int lid = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (lid / ntn) * BM, n0 = (lid % ntn) * BN;

  // per-thread A rows: chunk id = tid + i*256 -> row = id / CPR, chunk = id % CPR
  int a_iy0[APT], a_ix0[APT];
  const float* a_base[APT];
  bool a_ok[APT];
#pragma unroll
  for (int i = 0; i < APT; ++i) {
    const int id = tid + i * 256, r = id / CPR;
    const int m = m0 + r;
    a_ok[i] = m < M;
    const int mm = a_ok[i] ? m : 0;
    const int ox = mm % p.Wo, t = mm / p.Wo, oy = t % p.Ho, n = t / p.Ho;
    a_iy0[i] = oy * p.stride - p.pad;
    a_ix0[i] = ox * p.stride - p.pad;
    a_base[i] = p.x + (size_t)n * p.Hi * p.Wi * p.Cin + (id % CPR) * 8;
  }
  const int cin_tiles = p.Cin / BK;
  const int nk = p.ksize * p.ksize * cin_tiles;

  f32x4 areg[APT][2];  // native vectors: each is tied to the explicit wait in store_a
  auto load_a = [&](int kt) {
    const int tap = kt / cin_tiles, ci0 = (kt - tap * cin_tiles) * BK;
    const int ky = tap / p.ksize, kx = tap - ky * p.ksize;
#pragma unroll
    for (int i = 0; i < APT; ++i) {
      const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
      if (a_ok[i] && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi) {
        const float* s = a_base[i] + ((size_t)iy * p.Wi + ix) * p.Cin + ci0;
        areg[i][0] = *(const f32x4*)s;
        areg[i][1] = *(const f32x4*)(s + 4);
      } else {
        areg[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        areg[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto store_a = [&](char* tile) {
#pragma unroll
    for (int i = 0; i < APT; ++i) {
      const int id = tid + i * 256, r = id / CPR, c = id % CPR;
      // released by a full drain the registers are tied to (hipcc's own waits consumed these loop-carried loads
      // too early with LDS-DMA in flight; untied register math may be hoisted above a bare asm wait)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(areg[i][0]), "+v"(areg[i][1]) : : "memory");
      float v[8] = {areg[i][0][0], areg[i][0][1], areg[i][0][2], areg[i][0][3],
                    areg[i][1][0], areg[i][1][1], areg[i][1][2], areg[i][1][3]};
      if (p.relu_in) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      uint4 u;
      u.x = (uint32_t)f2t<T>(v[0]) | ((uint32_t)f2t<T>(v[1]) << 16);
      u.y = (uint32_t)f2t<T>(v[2]) | ((uint32_t)f2t<T>(v[3]) << 16);
      u.z = (uint32_t)f2t<T>(v[4]) | ((uint32_t)f2t<T>(v[5]) << 16);
      u.w = (uint32_t)f2t<T>(v[6]) | ((uint32_t)f2t<T>(v[7]) << 16);
      *(uint4*)(tile + r * (BK * 2) + ((c ^ swz<BK>(r)) << 4)) = u;
    }
  };

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const u16* W = (const u16*)p.w;
  load_a(0);
  stage_w<BK, BN>(W, K, n0, p.Cout, 0, smem + A_BYTES, wave, lane);
  // Explicit drain BEFORE the register loads are consumed: (i) hipcc does not reliably drain LDS-DMA before a barrier,
  // (ii) LDS-DMA and register loads return out of order with respect to each other, so a compiler-counted vmcnt(N > 0)
  // for the register loads is not safe while DMA is in flight (see conv3x3.hip).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  store_a(smem);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const char* tA = smem + cur * (A_BYTES + B_BYTES);
    const char* tB = tA + A_BYTES;
    char* nA = smem + (cur ^ 1) * (A_BYTES + B_BYTES);
    if (kt + 1 < nk) {
      load_a(kt + 1);
      stage_w<BK, BN>(W, K, n0, p.Cout, (kt + 1) * BK, nA + A_BYTES, wave, lane);
    }
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int ch = 2 * ks + (lane >> 5);
      s16x8 a[NI], b[NJ];
#pragma unroll
      for (int i = 0; i < NI; ++i) a[i] = frag<BK>(tA, (wm * NI + i) * 32 + (lane & 31), ch);
#pragma unroll
      for (int j = 0; j < NJ; ++j) b[j] = frag<BK>(tB, (wn * NJ + j) * 32 + (lane & 31), ch);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32<T>(a[i], b[j], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (kt + 1 < nk) store_a(nA);
    __syncthreads();
    cur ^= 1;
  }

#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = n0 + (wn * NJ + j) * 32 + (lane & 31);
    if (col >= p.Cout) continue;
    const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * NI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= M) continue;
        const size_t o = (size_t)row * p.Cout + col;
        float v = acc[i][j][r] + bias;
        if (p.resid) {
          const float rr = p.resid[o];
          v += p.resid_relu ? fmaxf(rr, 0.f) : rr;
        }
        if (p.resid2) v += p.resid2[o];
        if (p.relu_out) v = fmaxf(v, 0.f);
        p.y[o] = v;
      }
  }
}

template <int T, int BK, int WGM, int WGN, int NI, int NJ>
hipError_t launch_cfg(const WmConvArgs& a, hipStream_t s) {
  constexpr int BM = WGM * NI * 32, BN = WGN * NJ * 32;
  const int M = a.N * a.Ho * a.Wo;
  const int ntn = (a.Cout + BN - 1) / BN, ntm = (M + BM - 1) / BM;
  const size_t shm = 2 * (BM + BN) * BK * 2;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_kernel<T, BK, WGM, WGN, NI, NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    attr = true;
  }
  hipLaunchKernelGGL((conv_kernel<T, BK, WGM, WGN, NI, NJ>), dim3(ntm * ntn), dim3(256), shm, s, a);
  return hipGetLastError();
}

template <int T>
hipError_t launch_T(const WmConvArgs& a, hipStream_t s) {
  const bool narrow = a.Cout <= 32;
  if (a.Cin % 64 == 0) {
    // 128 x 128 tiles that do not cover the chip (the stride-2 conv of the deepest DPT level: 184 blocks of 4 waves, each a
    // 144-step K loop waiting ~1 us of memory latency per 0.25 us of MFMA work: 232 us = 235 TF/s): 64-pixel tiles double the
    // blocks, two or three of which share a CU and overlap each other's waits
    static const int ncu = [] { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&pr, d) == hipSuccess ? pr.multiProcessorCount : 256; }();
    const long M = (long)a.N * a.Ho * a.Wo;
    const long t128 = ((M + 127) / 128) * ((a.Cout + 127) / 128);
    static const int bm_env = [] { const char* e = wm_env("WM_CONV_BM"); return e ? atoi(e) : 0; }();
    if (!narrow && (bm_env == 64 || (bm_env == 0 && t128 < ncu))) return launch_cfg<T, 64, 2, 2, 1, 2>(a, s);
    return narrow ? launch_cfg<T, 64, 4, 1, 1, 1>(a, s) : launch_cfg<T, 64, 2, 2, 2, 2>(a, s);
  } else {
    return narrow ? launch_cfg<T, 32, 4, 1, 1, 1>(a, s) : launch_cfg<T, 32, 2, 2, 2, 2>(a, s);
  }
}

}  // namespace

bool wm_conv3x3_applicable(const WmConvArgs& a);
hipError_t wm_launch_conv3x3(const WmConvArgs& a, hipStream_t s);

hipError_t wm_launch_conv(const WmConvArgs& a, hipStream_t s) {
  if (a.N <= 0) return hipSuccess;
  const bool no_halo = wm_conv_force_generic();   // (wm_conv3x3_out16_ok knows the switch too: no out16 grant that this launch cannot honour)
  if (a.up_hs > 0) {  // fused input upsample exists in the halo kernel only; callers test wm_conv3x3_applicable first
    if (!wm_conv3x3_applicable(a) || a.up_ws <= 0 || (a.up_addx && (a.Cin & 15))) return hipErrorInvalidValue;
    return wm_launch_conv3x3(a, s);
  }
  if (!no_halo && wm_conv3x3_applicable(a)) return wm_launch_conv3x3(a, s);
  if (a.in16) return hipErrorInvalidValue;   // only the register-staged 3x3 kernel reads a 16-bit input
  if (a.Cin % 32 != 0 || a.ksize < 1 || a.out16) return hipErrorInvalidValue;  // (16-bit output: register-staged 3x3 kernel only)
  if (a.Ho != (a.Hi + 2 * a.pad - a.ksize) / a.stride + 1 || a.Wo != (a.Wi + 2 * a.pad - a.ksize) / a.stride + 1)
    return hipErrorInvalidValue;
  return a.dtype == WM_T_BF16 ? launch_T<WM_T_BF16>(a, s) : launch_T<WM_T_F16>(a, s);
}
