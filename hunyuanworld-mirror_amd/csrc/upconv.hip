// Conv2d(C, Co, 3, padding=1) of F.interpolate(x, (Ho, Wo), bilinear, align_corners=True) WITHOUT the upsampled tensor and with a quarter of the
// MFMA work — the DPT head's output_conv1 behind the last fusion block (dense_head.py:217-225,265-295).
//
// Both steps are linear and the interpolation acts per channel, so the conv's channel mixing commutes with it:
//     conv(U x)(p) = sum_tap W_tap (U x)(p + tap) = sum_tap U (W_tap x) (p + tap)          (taps outside the (Ho, Wo) image: zero padding)
// (1) the nine 1x1 products y[q][tap][co] = W_tap x[q] at the LOW resolution are one GEMM of the ping-pong kernel on the 16-bit NHWC x:
//     M = N Hi Wi, K = C, N = 9 Co, weights repacked tap-major (repack_tap_major_kernel, once per weight) — a quarter of the direct conv's
//     flops when the resize doubles each side;
// (2) upconv_gather_kernel: out[p][co] = bias[co] + sum over the taps inside the image of the bilinear sample of y[.][tap][co] at
//     (p + tap) scaled back — 36 (coefficient, 16-byte load) pairs per output pixel and 8 channels, fp32 accumulation, fp32 NHWC out.
// y is stored in f16 (the heads' operand type): its rounding (2^-11 per tap product) is of the size of the operand rounding the direct
// conv applies to the interpolated values; |y| must stay in f16's range like the operands themselves.
#include <hip/hip_fp16.h>
#include "wm_common.h"
#include "wm_kernels.h"

namespace {

// w16 [Co][9][C] (the conv weight as the other kernels use it) -> wt [9][Co][C]
__global__ __launch_bounds__(256) void repack_tap_major_kernel(const uint16_t* __restrict__ w, uint16_t* __restrict__ wt, int Co, int C) {
  const size_t total = (size_t)Co * 9 * (C / 8);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c8 = (int)(i % (C / 8));
    const size_t r = i / (C / 8);
    const int tap = (int)(r % 9), co = (int)(r / 9);
    *(uint4*)(wt + ((size_t)tap * Co + co) * C + c8 * 8) = *(const uint4*)(w + ((size_t)co * 9 + tap) * C + c8 * 8);
  }
}

__device__ __forceinline__ void acc8(float (&a)[8], const uint4 v, const float w) {
  const __half2* h = reinterpret_cast<const __half2*>(&v);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float2 f = __half22float2(h[i]);
    a[2 * i] = fmaf(w, f.x, a[2 * i]);
    a[2 * i + 1] = fmaf(w, f.y, a[2 * i + 1]);
  }
}

// One lane: one output pixel x 8 channels; 16 / (Co / 8) ... a wave covers 64 / (Co / 8) consecutive pixels of a row x all Co channels, so
// every load instruction reads whole 2 Co-byte runs.  PX = pixels per 256-thread block.
template <int CO8>   // Co / 8 lanes per pixel: 16 (Co = 128) or 8 (Co = 64) or 4 (Co = 32)
__global__ __launch_bounds__(256) void upconv_gather_kernel(const uint16_t* __restrict__ y, const float* __restrict__ bias, float* __restrict__ out,
                                                           int N, int Hi, int Wi, int Ho, int Wo) {
  constexpr int Co = CO8 * 8, PX = 256 / CO8;
  const int nbx = (Wo + PX - 1) / PX;
  const int bx = blockIdx.x % nbx;
  const int rest = blockIdx.x / nbx;
  const int Y = rest % Ho, n = rest / Ho;
  const int X = bx * PX + (int)threadIdx.x / CO8, c = ((int)threadIdx.x % CO8) * 8;
  if (X >= Wo) return;
  const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
  const float sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  float a[8];
  if (bias) {
    const float4 b0 = *(const float4*)(bias + c), b1 = *(const float4*)(bias + c + 4);
    a[0] = b0.x; a[1] = b0.y; a[2] = b0.z; a[3] = b0.w; a[4] = b1.x; a[5] = b1.y; a[6] = b1.z; a[7] = b1.w;
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = 0.f;
  }
  // 32-bit byte offsets from the (scalar) tensor base: the launcher checks that y is below 4 GB
  const char* yb = (const char*)y;
  constexpr uint32_t PIX = 9u * Co * 2u;   // bytes of one low-resolution pixel's nine products
  const uint32_t img = (uint32_t)n * (uint32_t)(Hi * Wi) * PIX + (uint32_t)c * 2u;
  // the three taps along x: columns, weights, validity (zero padding of the HIGH-resolution image)
  uint32_t co0[3], co1[3];
  float wx0[3], wx1[3];
#pragma unroll
  for (int tx = 0; tx < 3; ++tx) {
    const int Xs = X + tx - 1;
    const bool in = Xs >= 0 && Xs < Wo;
    const int Xc = Xs < 0 ? 0 : (Xs >= Wo ? Wo - 1 : Xs);
    const float fx = sx * Xc;
    int c0 = (int)fx;
    c0 = c0 < Wi - 1 ? c0 : Wi - 1;
    const int c1 = c0 < Wi - 1 ? c0 + 1 : c0;
    const float wx = fx - c0;
    wx1[tx] = in ? wx : 0.f;
    wx0[tx] = in ? 1.f - wx : 0.f;
    co0[tx] = img + (uint32_t)c0 * PIX + (uint32_t)tx * (Co * 2u);
    co1[tx] = img + (uint32_t)c1 * PIX + (uint32_t)tx * (Co * 2u);
  }
#pragma unroll
  for (int ty = 0; ty < 3; ++ty) {
    const int Ys = Y + ty - 1;
    if (Ys < 0 || Ys >= Ho) continue;   // block-uniform
    const float fy = sy * Ys;
    int r0 = (int)fy;
    r0 = r0 < Hi - 1 ? r0 : Hi - 1;
    const int r1 = r0 < Hi - 1 ? r0 + 1 : r0;
    const float wy = fy - r0;
    const uint32_t ro0 = (uint32_t)(r0 * Wi) * PIX + (uint32_t)ty * (3u * Co * 2u), ro1 = (uint32_t)(r1 * Wi) * PIX + (uint32_t)ty * (3u * Co * 2u);   // scalar
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      const uint4 v00 = *(const uint4*)(yb + (co0[tx] + ro0));
      const uint4 v01 = *(const uint4*)(yb + (co1[tx] + ro0));
      const uint4 v10 = *(const uint4*)(yb + (co0[tx] + ro1));
      const uint4 v11 = *(const uint4*)(yb + (co1[tx] + ro1));
      acc8(a, v00, (1.f - wy) * wx0[tx]); acc8(a, v01, (1.f - wy) * wx1[tx]); acc8(a, v10, wy * wx0[tx]); acc8(a, v11, wy * wx1[tx]);
    }
  }
  float* o = out + (((size_t)n * Ho + Y) * Wo + X) * Co + c;
  *(float4*)o = make_float4(a[0], a[1], a[2], a[3]);
  *(float4*)(o + 4) = make_float4(a[4], a[5], a[6], a[7]);
}

// The same gather with the products staged in LDS (resizes by about two per side: the DPT case).  A block owns an 8 x 16 output tile and 32
// channels; the low-resolution pixels its 36 samples per output can touch (at most 7 x 11) are copied once, by LDS-DMA, as records of nine
// 64-byte tap pieces (+ 16 B of padding: consecutive columns then fall on different banks), and every sample is a ds_read_b128 instead of a
// 16-byte trip through the vector cache (the direct kernel moves 18 x its output through L1: 325 us at 8 x 296^2 x 128).
constexpr int UG_TY = 8, UG_TX = 16, UG_RMAX = 7, UG_CMAX = 11, UG_REC = 37;
template <int CO8>
__global__ __launch_bounds__(256) void upconv_gather_lds_kernel(const uint16_t* __restrict__ y, const float* __restrict__ bias, float* __restrict__ out,
                                                               int N, int Hi, int Wi, int Ho, int Wo) {
  constexpr int Co = CO8 * 8, NG = Co / 32;
  __shared__ uint4 lds[UG_RMAX * UG_CMAX * UG_REC + 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntx = (Wo + UG_TX - 1) / UG_TX, nty = (Ho + UG_TY - 1) / UG_TY;
  // The NG channel-group blocks of one tile read the same 128-byte lines (64 B of each tap piece per group): keep them on ONE XCD's L2 — workgroups
  // go round-robin over the 8 XCDs, so the group index is taken from bits above the low three of the block index (PMC, first form: the groups of a
  // tile sat on NG different XCDs and the launch fetched 650 MB for 404 MB of products)
  const int nb = gridDim.x;
  int g, tile;
  if (NG > 1 && (nb / NG) % 8 == 0) { const int b = blockIdx.x; g = (b >> 3) % NG; tile = (b / (8 * NG)) * 8 + (b & 7); }
  else { g = blockIdx.x % NG; tile = blockIdx.x / NG; }
  const int X0 = (tile % ntx) * UG_TX;
  tile /= ntx;
  const int Y0 = (tile % nty) * UG_TY, n = tile / nty;
  if (n >= N) return;   // padding blocks
  const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
  const float sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  auto lo_of = [](float sc, int p, int lim) { int v = (int)(sc * (float)p); return v < lim - 1 ? v : lim - 1; };
  const int ylo = Y0 > 0 ? Y0 - 1 : 0, yhi = Y0 + UG_TY < Ho ? Y0 + UG_TY : Ho - 1;
  const int xlo = X0 > 0 ? X0 - 1 : 0, xhi = X0 + UG_TX < Wo ? X0 + UG_TX : Wo - 1;
  const int r_lo = lo_of(sy, ylo, Hi), c_lo = lo_of(sx, xlo, Wi);
  int r_hi = lo_of(sy, yhi, Hi) + 1, c_hi = lo_of(sx, xhi, Wi) + 1;
  r_hi = r_hi < Hi ? r_hi : Hi - 1;
  c_hi = c_hi < Wi ? c_hi : Wi - 1;
  int nr = r_hi - r_lo + 1, nc = c_hi - c_lo + 1;
  nr = nr < UG_RMAX ? nr : UG_RMAX;   // (the launcher admits only scales for which these never bind)
  nc = nc < UG_CMAX ? nc : UG_CMAX;
  const int total = nr * nc * UG_REC;   // 16-byte units
  const uint16_t* img = y + (size_t)n * Hi * Wi * 9 * Co + g * 32;
  for (int base = wave * 64; base < total; base += 256) {
    int u = base + lane;
    u = u < total ? u : total - 1;
    const int rec = u / UG_REC, k = u - rec * UG_REC, kk = k < 36 ? k : 35;
    const int row = rec / nc, col = rec - row * nc;
    const uint16_t* src = img + ((size_t)((r_lo + row) * Wi + c_lo + col) * 9 + (kk >> 2)) * Co + (kk & 3) * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(lds + base), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int sub = tid & 3, ch = g * 32 + sub * 8;
  float b8[8];
  if (bias) {
    const float4 b0 = *(const float4*)(bias + ch), b1 = *(const float4*)(bias + ch + 4);
    b8[0] = b0.x; b8[1] = b0.y; b8[2] = b0.z; b8[3] = b0.w; b8[4] = b1.x; b8[5] = b1.y; b8[6] = b1.z; b8[7] = b1.w;
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) b8[i] = 0.f;
  }
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int pix = pass * 64 + (tid >> 2);
    const int Y = Y0 + pix / UG_TX, X = X0 + pix % UG_TX;
    if (Y >= Ho || X >= Wo) continue;
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = b8[i];
    int u0[3], u1[3];
    float wx0[3], wx1[3];
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      const int Xs = X + tx - 1;
      const bool in = Xs >= 0 && Xs < Wo;
      const int Xc = Xs < 0 ? 0 : (Xs >= Wo ? Wo - 1 : Xs);
      const float fx = sx * Xc;
      int c0 = (int)fx;
      c0 = c0 < Wi - 1 ? c0 : Wi - 1;
      const int c1 = c0 < Wi - 1 ? c0 + 1 : c0;
      const float wx = fx - c0;
      wx1[tx] = in ? wx : 0.f;
      wx0[tx] = in ? 1.f - wx : 0.f;
      int d0 = c0 - c_lo, d1 = c1 - c_lo;   // inside [0, nc) for every tap that counts; clamped for the masked ones
      d0 = d0 < 0 ? 0 : (d0 < nc ? d0 : nc - 1);
      d1 = d1 < 0 ? 0 : (d1 < nc ? d1 : nc - 1);
      u0[tx] = d0 * UG_REC + tx * 4 + sub;
      u1[tx] = d1 * UG_REC + tx * 4 + sub;
    }
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      const int Ys = Y + ty - 1;
      if (Ys < 0 || Ys >= Ho) continue;
      const float fy = sy * Ys;
      int r0 = (int)fy;
      r0 = r0 < Hi - 1 ? r0 : Hi - 1;
      const int r1 = r0 < Hi - 1 ? r0 + 1 : r0;
      const float wy = fy - r0;
      int e0 = r0 - r_lo, e1 = r1 - r_lo;
      e0 = e0 < 0 ? 0 : (e0 < nr ? e0 : nr - 1);
      e1 = e1 < 0 ? 0 : (e1 < nr ? e1 : nr - 1);
      const int ro0 = e0 * nc * UG_REC + ty * 12, ro1 = e1 * nc * UG_REC + ty * 12;
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        const uint4 v00 = lds[ro0 + u0[tx]], v01 = lds[ro0 + u1[tx]], v10 = lds[ro1 + u0[tx]], v11 = lds[ro1 + u1[tx]];
        acc8(a, v00, (1.f - wy) * wx0[tx]); acc8(a, v01, (1.f - wy) * wx1[tx]); acc8(a, v10, wy * wx0[tx]); acc8(a, v11, wy * wx1[tx]);
      }
    }
    float* o = out + (((size_t)n * Ho + Y) * Wo + X) * Co + ch;
    *(float4*)o = make_float4(a[0], a[1], a[2], a[3]);
    *(float4*)(o + 4) = make_float4(a[4], a[5], a[6], a[7]);
  }
}

// fp32 [rows][cols] (pitch ld_src) -> 16-bit [rows][cols] (pitch ld_dst): the combined token-conv matrices into their place in the B operand
template <int T>
__global__ __launch_bounds__(256) void f32_to_16_2d_kernel(const float* __restrict__ src, int ld_src, uint16_t* __restrict__ dst, int ld_dst, int rows, int cols) {
  const size_t total = (size_t)rows * cols;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / cols), c = (int)(i - (size_t)r * cols);
    dst[(size_t)r * ld_dst + c] = f2t<T>(src[(size_t)r * ld_src + c]);
  }
}

// Token-conv border fix (gemm.hip, WM_EPI_CONV with tc_k > 0): the composed GEMM adds the ConvTranspose's bias through ALL nine taps of the 3x3
// conv; on the image border the taps that fall outside see zero padding, not bias: subtract bmiss[tap][f] = (W_rn[tap] b_ct)[f] for those.
__global__ __launch_bounds__(256) void tconv_border_kernel(float* __restrict__ out, const float* __restrict__ bmiss, int N, int H, int W, int F) {
  const int per = 2 * W + 2 * (H - 2 > 0 ? H - 2 : 0), f4n = F / 4;
  const size_t total = (size_t)N * per * f4n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int f = (int)(i % f4n) * 4;
    size_t r = i / f4n;
    const int b = (int)(r % per), n = (int)(r / per);
    int Y, X;
    if (b < W) { Y = 0; X = b; }
    else if (b < 2 * W) { Y = H - 1; X = b - W; }
    else { const int q = b - 2 * W; Y = 1 + (q >> 1); X = (q & 1) ? W - 1 : 0; }
    if (H == 1 && b >= W) continue;          // a single row: its pixels are listed once
    if (W == 1 && b >= 2 * W && (b & 1)) continue;
    float4 sub = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        if (Y + dy >= 0 && Y + dy < H && X + dx >= 0 && X + dx < W) continue;
        const float4 m = *(const float4*)(bmiss + (size_t)((dy + 1) * 3 + dx + 1) * F + f);
        sub.x += m.x; sub.y += m.y; sub.z += m.z; sub.w += m.w;
      }
    float4* o = (float4*)(out + (((size_t)n * H + Y) * W + X) * F + f);
    float4 v = *o;
    v.x -= sub.x; v.y -= sub.y; v.z -= sub.z; v.w -= sub.w;
    *o = v;
  }
}

}  // namespace

hipError_t wm_launch_f32_to_16_2d(const float* src, int ld_src, void* dst16, int ld_dst, int rows, int cols, int dtype, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return hipSuccess;
  const size_t total = (size_t)rows * cols;
  const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == WM_T_BF16) hipLaunchKernelGGL(f32_to_16_2d_kernel<WM_T_BF16>, dim3(grid), dim3(256), 0, s, src, ld_src, (uint16_t*)dst16, ld_dst, rows, cols);
  else hipLaunchKernelGGL(f32_to_16_2d_kernel<WM_T_F16>, dim3(grid), dim3(256), 0, s, src, ld_src, (uint16_t*)dst16, ld_dst, rows, cols);
  return hipGetLastError();
}

hipError_t wm_launch_tconv_border(float* out, const float* bmiss, int N, int H, int W, int F, hipStream_t s) {
  if (N <= 0 || H <= 0 || W <= 0 || (F & 3)) return F & 3 ? hipErrorInvalidValue : hipSuccess;
  const size_t total = (size_t)N * (2 * W + 2 * (H - 2 > 0 ? H - 2 : 0)) * (F / 4);
  const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(tconv_border_kernel, dim3(grid), dim3(256), 0, s, out, bmiss, N, H, W, F);
  return hipGetLastError();
}

hipError_t wm_launch_repack_tap_major(const void* w16, void* wt16, int Co, int C, hipStream_t s) {
  if (Co <= 0 || C <= 0 || (C & 7)) return hipErrorInvalidValue;
  const size_t total = (size_t)Co * 9 * (C / 8);
  hipLaunchKernelGGL(repack_tap_major_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const uint16_t*)w16, (uint16_t*)wt16, Co, C);
  return hipGetLastError();
}

hipError_t wm_launch_upconv_gather(const void* y16, const float* bias, float* out, int N, int Hi, int Wi, int Ho, int Wo, int Co, hipStream_t s) {
  if (N <= 0) return hipSuccess;
  const bool lds_ok = wm_tuning[WM_TUNE_UP1_GATHER] != 2;   // 2: the direct (cache-fed) kernel on every shape (A/B)
  if (Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1 || (Co != 128 && Co != 64 && Co != 32)) return hipErrorInvalidValue;
  if ((unsigned long long)N * Hi * Wi * 9ull * Co * 2ull >= (1ull << 32)) return hipErrorInvalidValue;   // the kernel's 32-bit byte offsets
  // resizes by about two per side (the rows / columns an 8 x 16 output tile samples fit the LDS tile): the LDS-staged kernel
  const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  // rows a tile samples: floor(sy yhi) + 1 - floor(sy ylo) + 1 with yhi - ylo <= TY + 1, i.e. at most ceil(sy (TY + 1)) + 2 (same for columns)
  if (lds_ok && (int)ceilf(sy * (UG_TY + 1)) + 2 <= UG_RMAX && (int)ceilf(sx * (UG_TX + 1)) + 2 <= UG_CMAX) {
    const size_t tiles = (size_t)N * ((Ho + UG_TY - 1) / UG_TY) * ((Wo + UG_TX - 1) / UG_TX);
    const dim3 grid((unsigned)(((tiles + 7) / 8 * 8) * (Co / 32)));   // tiles padded to eights (XCD-aware group mapping; surplus blocks exit)
    if (Co == 128) hipLaunchKernelGGL(upconv_gather_lds_kernel<16>, grid, dim3(256), 0, s, (const uint16_t*)y16, bias, out, N, Hi, Wi, Ho, Wo);
    else if (Co == 64) hipLaunchKernelGGL(upconv_gather_lds_kernel<8>, grid, dim3(256), 0, s, (const uint16_t*)y16, bias, out, N, Hi, Wi, Ho, Wo);
    else hipLaunchKernelGGL(upconv_gather_lds_kernel<4>, grid, dim3(256), 0, s, (const uint16_t*)y16, bias, out, N, Hi, Wi, Ho, Wo);
    return hipGetLastError();
  }
  const int px = 256 / (Co / 8);
  const dim3 grid((unsigned)((size_t)N * Ho * ((Wo + px - 1) / px)));
  if (Co == 128) hipLaunchKernelGGL(upconv_gather_kernel<16>, grid, dim3(256), 0, s, (const uint16_t*)y16, bias, out, N, Hi, Wi, Ho, Wo);
  else if (Co == 64) hipLaunchKernelGGL(upconv_gather_kernel<8>, grid, dim3(256), 0, s, (const uint16_t*)y16, bias, out, N, Hi, Wi, Ho, Wo);
  else hipLaunchKernelGGL(upconv_gather_kernel<4>, grid, dim3(256), 0, s, (const uint16_t*)y16, bias, out, N, Hi, Wi, Ho, Wo);
  return hipGetLastError();
}
