// HBM-bound kernels of the WorldMirror path: LayerNorm, QKV post-processing (per-head LayerNorm +
// 2-D RoPE + head-major relayout), patchify im2col, token assembly, bilinear resize, DPT tail.
// All are vectorised 16 B/lane where the layout allows and sized to ~8 blocks/CU (guides §6 G11/G13).
#include "wm_common.h"
#include "wm_kernels.h"
#include <cstdlib>

namespace {

constexpr float RESNET_MEAN[3] = {0.485f, 0.456f, 0.406f};  // visual_transformer.py:16-17
constexpr float RESNET_STD[3] = {0.229f, 0.224f, 0.225f};

// ------------------------------------------------------------------------------------------ LayerNorm
// One wave per row, row cached in registers (D <= 2048), two-pass mean/variance like torch.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const WmLnArgs p) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long nrows = (long long)p.groups * p.rows_per_group;
  if (row >= nrows) return;
  const int g = (int)(row / p.rows_per_group), q = (int)(row - (long long)g * p.rows_per_group);
  const float* x = p.x + ((size_t)g * p.in_group + p.in_off + q) * p.ld_in;
  const size_t orow = (size_t)g * p.out_group + p.out_off + q;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    v[i] = c < p.D ? *(const float4*)(x + c) : make_float4(0, 0, 0, 0);
    s += v[i].x + v[i].y + v[i].z + v[i].w;
  }
  const float mean = wave_sum(s) / p.D;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < p.D) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      ss += a * a + b * b + cc * cc + d * d;
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / p.D + p.eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c >= p.D) continue;
    float4 w = p.w ? *(const float4*)(p.w + c) : make_float4(1, 1, 1, 1);
    float4 b = p.b ? *(const float4*)(p.b + c) : make_float4(0, 0, 0, 0);
    float4 y;
    y.x = (v[i].x - mean) * rstd * w.x + b.x;
    y.y = (v[i].y - mean) * rstd * w.y + b.y;
    y.z = (v[i].z - mean) * rstd * w.z + b.z;
    y.w = (v[i].w - mean) * rstd * w.w + b.w;
    if (p.out_f32) {
      *(float4*)((float*)p.y + orow * p.ld_out + c) = y;
    } else {
      uint2 u;
      if (p.dtype == WM_T_BF16) {
        u.x = (uint32_t)f2bf(y.x) | ((uint32_t)f2bf(y.y) << 16);
        u.y = (uint32_t)f2bf(y.z) | ((uint32_t)f2bf(y.w) << 16);
      } else {
        u.x = (uint32_t)f2h(y.x) | ((uint32_t)f2h(y.y) << 16);
        u.y = (uint32_t)f2h(y.z) | ((uint32_t)f2h(y.w) << 16);
      }
      *(uint2*)((u16*)p.y + orow * p.ld_out + c) = u;
    }
  }
}

// The backbone's case (D = NV * 256 exactly, affine, 16-bit output) without a branch and with every load of the row's life — x, weight
// and bias — requested up front.  The general kernel above tests `c < D` per chunk, so hipcc emits, per chunk, {load w, load b,
// s_waitcnt vmcnt(0), compute, store}: four serialised L2 round trips per row behind the reductions (round 4: 14.3 -> 12.x us for the
// 11008 x 1024 rows of a backbone LayerNorm).  Same expressions in the same order as the general kernel: bit-identical results.
template <int NV, int T, int RPW>
__global__ __launch_bounds__(256) void layernorm_fast_kernel(const WmLnArgs p) {
  // RPW rows per wave (A/B: 2 makes the 11 K-row launches of an 8-view forward one round of wave slots instead of 1.34 — measured equal),
  // all of their loads in flight before the first reduction
  const int lane = threadIdx.x & 63;
  const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
  const long long nrows = (long long)p.groups * p.rows_per_group;
  if (row0 >= nrows) return;
  float4 v[RPW][NV], w[NV], b[NV];
  const float* x[RPW];
  size_t orow[RPW];
  bool ok[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    ok[r] = row0 + r < nrows;                              // wave-uniform
    const long long row = ok[r] ? row0 + r : row0;
    const int g = (int)(row / p.rows_per_group), q = (int)(row - (long long)g * p.rows_per_group);
    x[r] = p.x + ((size_t)g * p.in_group + p.in_off + q) * p.ld_in;
    orow[r] = (size_t)g * p.out_group + p.out_off + q;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[r][i] = *(const float4*)(x[r] + (i * 64 + lane) * 4);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    w[i] = *(const float4*)(p.w + (i * 64 + lane) * 4);
    b[i] = *(const float4*)(p.b + (i * 64 + lane) * 4);
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[r][i].x + v[r][i].y + v[r][i].z + v[r][i].w;
    const float mean = wave_sum(s) / p.D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float a = v[r][i].x - mean, bb = v[r][i].y - mean, cc = v[r][i].z - mean, d = v[r][i].w - mean;
      ss += a * a + bb * bb + cc * cc + d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / p.D + p.eps);
    if (!ok[r]) continue;
    u16* y = (u16*)p.y + orow[r] * p.ld_out;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float y0 = (v[r][i].x - mean) * rstd * w[i].x + b[i].x, y1 = (v[r][i].y - mean) * rstd * w[i].y + b[i].y;
      const float y2 = (v[r][i].z - mean) * rstd * w[i].z + b[i].z, y3 = (v[r][i].w - mean) * rstd * w[i].w + b[i].w;
      uint2 u;
      u.x = (uint32_t)f2t<T>(y0) | ((uint32_t)f2t<T>(y1) << 16);
      u.y = (uint32_t)f2t<T>(y2) | ((uint32_t)f2t<T>(y3) << 16);
      *(uint2*)(y + (i * 64 + lane) * 4) = u;
    }
  }
}

// ------------------------------------------------------------------------------------------ QKV post
// One wave = one token x 4 heads; 16 lanes per head vector (4 elements per lane).
// attention.py:50-56: split heads, q/k LayerNorm(64, eps 1e-5, affine), 2-D RoPE (rope.py:148-181).
template <int T>
__global__ __launch_bounds__(256) void qkv_post_kernel(const WmQkvArgs p) {
  const int lane = threadIdx.x & 63, sub = lane & 15, hg = lane >> 4;
  const int hgroups = (p.H + 3) >> 2;
  const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (long long)p.M * hgroups) return;
  const int m = (int)(wid / hgroups);
  const int head_raw = ((int)(wid - (long long)m * hgroups) << 2) + hg;
  const bool head_ok = head_raw < p.H;            // H not a multiple of 4: surplus lanes compute on a
  const int head = head_ok ? head_raw : p.H - 1;  // clamped head (shuffles stay in-group) and skip the store
  const int D = p.H * 64;
  const float* row = p.qkv + (size_t)m * 3 * D + head * 64 + sub * 4;

  // rotary angles for this token (specials sit at (0,0): identity)
  float c[4] = {1, 1, 1, 1}, s[4] = {0, 0, 0, 0};
  if (p.rope_cos) {
    const int t = m % p.tokens_per_view;
    int pos = 0;
    if (t >= p.patch_start) {
      const int idx = t - p.patch_start;
      const int y = idx / p.grid_w + 1, x = idx - (y - 1) * p.grid_w + 1;
      pos = sub < 8 ? y : x;  // elements 0..31 rotate by y, 32..63 by x
    }
    const int f = (sub * 4) & 15;  // frequency index = element % 16
    const float4 cc = *(const float4*)(p.rope_cos + pos * 16 + f);
    const float4 sn = *(const float4*)(p.rope_sin + pos * 16 + f);
    c[0] = cc.x; c[1] = cc.y; c[2] = cc.z; c[3] = cc.w;
    s[0] = sn.x; s[1] = sn.y; s[2] = sn.z; s[3] = sn.w;
  }
  const size_t obase = ((size_t)head * p.head_stride + m) * 64 + sub * 4;

#pragma unroll
  for (int which = 0; which < 3; ++which) {
    float4 v4 = *(const float4*)(row + which * D);
    float v[4] = {v4.x, v4.y, v4.z, v4.w};
    if (which < 2) {
      const float* nw = which == 0 ? p.qn_w : p.kn_w;
      const float* nb = which == 0 ? p.qn_b : p.kn_b;
      if (nw) {
        float sm = v[0] + v[1] + v[2] + v[3];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
        const float mean = sm * (1.0f / 64.0f);
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] -= mean; ss += v[i] * v[i]; }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        const float rstd = 1.0f / sqrtf(ss * (1.0f / 64.0f) + 1e-5f);
        const float4 w4 = *(const float4*)(nw + sub * 4), b4 = *(const float4*)(nb + sub * 4);
        v[0] = v[0] * rstd * w4.x + b4.x; v[1] = v[1] * rstd * w4.y + b4.y;
        v[2] = v[2] * rstd * w4.z + b4.z; v[3] = v[3] * rstd * w4.w + b4.w;
      }
      if (p.rope_cos) {
        // partner holds element e^16 (lane sub^4): first half of each 32 gets -x2, second half +x1
        const float sign = (sub & 4) ? 1.0f : -1.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float o = __shfl_xor(v[i], 4);
          v[i] = v[i] * c[i] + sign * o * s[i];
        }
      }
      if (which == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] *= p.q_scale;
      }
    }
    uint2 u;
    u.x = (uint32_t)f2t<T>(v[0]) | ((uint32_t)f2t<T>(v[1]) << 16);
    u.y = (uint32_t)f2t<T>(v[2]) | ((uint32_t)f2t<T>(v[3]) << 16);
    u16* dst = (u16*)(which == 0 ? p.q : which == 1 ? p.k : p.v);
    if (head_ok) *(uint2*)(dst + obase) = u;
  }
}

// ------------------------------------------------------------------------------------------ im2col
// patch_embed.py:70 conv k=s=ps as GEMM: row (n,py,px), column c*ps*ps + ky*ps + kx, zero-padded to Kpad.
template <int T>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, u16* __restrict__ out, int N, int C,
                                                     int H, int W, int ps, int Kpad, int normalize) {
  const int gh = H / ps, gw = W / ps, K = C * ps * ps;
  const size_t total = (size_t)N * gh * gw * Kpad;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % Kpad);
    const size_t r = i / Kpad;
    float v = 0.f;
    if (k < K) {
      const int c = k / (ps * ps), kk = k - c * ps * ps, ky = kk / ps, kx = kk - ky * ps;
      const int px = (int)(r % gw), py = (int)((r / gw) % gh), n = (int)(r / ((size_t)gw * gh));
      v = img[(((size_t)n * C + c) * H + py * ps + ky) * W + px * ps + kx];
      if (normalize) v = (v - RESNET_MEAN[c]) / RESNET_STD[c];
    }
    out[i] = f2t<T>(v);
  }
}

// The same matrix, INPUT-indexed (round 3): one block per (image, patch row, channel) reads its ps image rows with coalesced
// full-line loads (lane i at x = 2 i: the whole 2 072-B row of a 518-px image in 5 wave-loads), normalises and rounds once, parks
// the ps x W tile in LDS and writes it out as gw runs of ps * ps consecutive 16-bit elements (392 B).  The output-indexed kernel
// above reads the image in 56-B patch-row segments: rocprofv3 showed 81 MB fetched for the 25.8 MB image of 8 views (3.2 x;
// profiles/r02_hbm_by_kernel.md) — harmless in time (38 us per forward) but not the coalesced image read the design promises.
// Same arithmetic, same rounding point (every e2e golden runs through it; WM_IM2COL_ROWS=0 selects the output-indexed kernel for an A/B).
// Needs ps * W * 2 bytes of LDS and an even W.
template <int T>
WM_NO_PACKED_FP32 __global__ __launch_bounds__(256) void im2col_rows_kernel(const float* __restrict__ img, u16* __restrict__ out, int N, int C,
                                                          int H, int W, int ps, int Kpad, int normalize) {
  extern __shared__ __attribute__((aligned(16))) u16 tile[];   // [ps][W]
  const int gh = H / ps, gw = W / ps, K = C * ps * ps;
  const int c = blockIdx.x % C, py = (blockIdx.x / C) % gh, n = blockIdx.x / (C * gh);
  const float mean = normalize ? RESNET_MEAN[c] : 0.f, sd = normalize ? RESNET_STD[c] : 1.f;
  const float* src = img + (((size_t)n * C + c) * H + (size_t)py * ps) * W;   // ps consecutive image rows = one contiguous span
  const int half = (ps * W) >> 1;
  for (int i = threadIdx.x; i < half; i += 256) {
    const float2 v = *(const float2*)(src + 2 * i);
    const float a = normalize ? (v.x - mean) / sd : v.x, b = normalize ? (v.y - mean) / sd : v.y;
    *(uint32_t*)(tile + 2 * i) = (uint32_t)f2t<T>(a) | ((uint32_t)f2t<T>(b) << 16);
  }
  __syncthreads();
  const int pp = ps * ps, hp = pp >> 1;   // ps even: a pair (kk, kk + 1) never straddles a patch row
  u16* dst = out + ((size_t)n * gh * gw + (size_t)py * gw) * Kpad + (size_t)c * pp;
  for (int i = threadIdx.x; i < gw * hp; i += 256) {
    const int px = i / hp, kk = 2 * (i - px * hp), ky = kk / ps, kx = kk - ky * ps;
    *(uint32_t*)(dst + (size_t)px * Kpad + kk) = *(const uint32_t*)(tile + ky * W + px * ps + kx);
  }
  if (c == C - 1) {   // zero padding of the rows' tails [K, Kpad)
    const int tail = (Kpad - K) >> 1;
    u16* z = out + ((size_t)n * gh * gw + (size_t)py * gw) * Kpad + K;
    for (int i = threadIdx.x; i < gw * tail; i += 256) {
      const int px = i / tail, j = i - px * tail;
      *(uint32_t*)(z + (size_t)px * Kpad + 2 * j) = 0u;
    }
  }
}

// Conv2d(3, C, 7, 1, 3) on the NCHW image as a GEMM (dense_head.py:91-95): row = pixel, col = c*49 + ky*7 + kx
template <int T>
__global__ __launch_bounds__(256) void im2col7_kernel(const float* __restrict__ img, u16* __restrict__ out, int N, int H, int W,
                                                      int Kpad) {
  const size_t total = (size_t)N * H * W * Kpad;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % Kpad);
    const size_t r = i / Kpad;
    float v = 0.f;
    if (k < 147) {
      const int c = k / 49, kk = k - c * 49, ky = kk / 7, kx = kk - ky * 7;
      const int x = (int)(r % W), y = (int)((r / W) % H), n = (int)(r / ((size_t)W * H));
      const int iy = y + ky - 3, ix = x + kx - 3;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = img[(((size_t)n * 3 + c) * H + iy) * W + ix];
    }
    out[i] = f2t<T>(v);
  }
}

// prepare_splats (rasterization.py:389-498, position_from="gsdepth+predcamera"): activations
// (act_gs.py), RGB->SH residual (sh_utils.py:112-113) and depth unprojection with the predicted
// camera (geometry.py:5-89, closed-form SE3 inverse :95-110).  One thread per pixel.
__global__ __launch_bounds__(256) void gs_splat_kernel(const float* __restrict__ gp, const float* __restrict__ img,
                                                       const float* __restrict__ depth, const float* __restrict__ cam,
                                                       float* __restrict__ means, float* __restrict__ quats,
                                                       float* __restrict__ scales, float* __restrict__ opac,
                                                       float* __restrict__ sh, float* __restrict__ wts, int N, int H, int W, int dbg) {
  const size_t npix = (size_t)N * H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % W), y = (int)((i / W) % H), n = (int)(i / ((size_t)W * H));
    const float* g = gp + i * 12;
    const float4 g0 = *(const float4*)g, g1 = *(const float4*)(g + 4), g2 = *(const float4*)(g + 8);
    const float qn = sqrtf(g0.x * g0.x + g0.y * g0.y + g0.z * g0.z + g0.w * g0.w) + 1e-8f;
    quats[i * 4 + 0] = g0.x / qn; quats[i * 4 + 1] = g0.y / qn; quats[i * 4 + 2] = g0.z / qn; quats[i * 4 + 3] = g0.w / qn;
    scales[i * 3 + 0] = fminf(expf(g1.x), 0.3f); scales[i * 3 + 1] = fminf(expf(g1.y), 0.3f); scales[i * 3 + 2] = fminf(expf(g1.z), 0.3f);
    opac[i] = 1.0f / (1.0f + expf(-g1.w));
    const float C0 = 0.28209479177387814f;
    const size_t hw = (size_t)H * W, p0 = (size_t)n * 3 * hw + (size_t)y * W + x;
    sh[i * 3 + 0] = (img[p0] - 0.5f) / C0 + g2.x;
    sh[i * 3 + 1] = (img[p0 + hw] - 0.5f) / C0 + g2.y;
    sh[i * 3 + 2] = (img[p0 + 2 * hw] - 0.5f) / C0 + g2.z;
    wts[i] = 1.0f / (1.0f + expf(-g2.w));
    // camera: [t, quat xyzw, fov_v, fov_u] is w2c; c2w = [R^T | -R^T t]
    const float* v = cam + n * 9;
    const float qi = v[3], qj = v[4], qk = v[5], qr = v[6];
    const float s2 = 2.0f / (qi * qi + qj * qj + qk * qk + qr * qr);
    const float R[9] = {1 - s2 * (qj * qj + qk * qk), s2 * (qi * qj - qk * qr), s2 * (qi * qk + qj * qr),
                        s2 * (qi * qj + qk * qr), 1 - s2 * (qi * qi + qk * qk), s2 * (qj * qk - qi * qr),
                        s2 * (qi * qk - qj * qr), s2 * (qj * qk + qi * qr), 1 - s2 * (qi * qi + qj * qj)};
    const float fy = H * 0.5f / tanf(v[7] * 0.5f), fx = W * 0.5f / tanf(v[8] * 0.5f);
    const float d = depth[i];
    const float xc = ((float)x - W * 0.5f) * d / fx, yc = ((float)y - H * 0.5f) * d / fy, zc = d;
    float tc[3];
    for (int a = 0; a < 3; ++a) tc[a] = -(R[0 * 3 + a] * v[0] + R[1 * 3 + a] * v[1] + R[2 * 3 + a] * v[2]);
    for (int a = 0; a < 3; ++a) means[i * 3 + a] = R[0 * 3 + a] * xc + R[1 * 3 + a] * yc + R[2 * 3 + a] * zc + tc[a];
#ifdef WM_DBG_SPLAT_BUILD   // diagnostic builds only (-DWM_DBG_SPLAT_BUILD, then WM_DBG_SPLAT=1): dump the camera vector this thread READ in
    if (dbg) {              // place of the other attributes (tools/dbg_c5.py) — a product build cannot be made to corrupt its splats
      opac[i] = v[0]; wts[i] = v[1]; scales[i * 3 + 0] = v[2]; scales[i * 3 + 1] = v[7]; scales[i * 3 + 2] = v[8];
      quats[i * 4 + 0] = v[3]; quats[i * 4 + 1] = v[4]; quats[i * 4 + 2] = v[5]; quats[i * 4 + 3] = v[6];
      sh[i * 3 + 0] = tc[0]; sh[i * 3 + 1] = tc[1]; sh[i * 3 + 2] = tc[2];
    }
#else
    (void)dbg;
#endif
  }
}

// ------------------------------------------------------------------------------------------ tokens
// vision_transformer.py:215-219: cls + pos[0], then R registers (patch rows are written by the
// patchify GEMM epilogue with pos[1+j] added).
__global__ __launch_bounds__(256) void dino_special_kernel(const float* cls, const float* reg, const float* pos, float* X,
                                                           int N, int T, int R, int D) {
  const int total = N * (1 + R) * D;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int d = i % D, t = (i / D) % (1 + R), n = i / (D * (1 + R));
    X[((size_t)n * T + t) * D + d] = t == 0 ? cls[d] + pos[d] : reg[(t - 1) * D + d];
  }
}

// visual_transformer.py:285-295,397-416: [cam, reg x R, (pose, ray)] ; token slot 0 for global view 0.
__global__ __launch_bounds__(256) void vgt_special_kernel(float* X, const float* cam, const float* reg, const float* pose,
                                                          const float* ray, int N, int P, int R, int D, int cond,
                                                          int first_view_global) {
  const int psi = 1 + R + (cond ? 2 : 0);
  const int total = N * psi * D;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int d = i % D, t = (i / D) % psi, n = i / (D * psi);
    const int slot = (first_view_global + n) == 0 ? 0 : 1;
    float v;
    if (t == 0) v = cam[slot * D + d];
    else if (t <= R) v = reg[(slot * R + (t - 1)) * D + d];
    else if (t == R + 1) v = pose ? pose[(size_t)n * D + d] : 0.f;
    else v = ray ? ray[(size_t)n * D + d] : 0.f;
    X[((size_t)n * P + t) * D + d] = v;
  }
}

__global__ __launch_bounds__(256) void copy2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows,
                                                     int cols4, int ld_src, int ld_dst) {
  const size_t total = (size_t)rows * cols4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / cols4;
    const int c = (int)(i - r * cols4) * 4;
    *(float4*)(dst + r * ld_dst + c) = *(const float4*)(src + r * ld_src + c);
  }
}

// F.interpolate(mode="bilinear", align_corners=True) on NHWC f32 (dense_head.py:217-225,535)
// OUT16 = 0: fp32 out.  OUT16 = 1 / 2: bf16 / f16 out (the operand of conv_n32.hip: rounded exactly where the fused-resize conv rounds)
template <int OUT16>
__global__ __launch_bounds__(256) void bilinear_kernel(const float* __restrict__ in, void* __restrict__ outp, int N, int Hi,
                                                       int Wi, int Ho, int Wo, int C4, const float* __restrict__ addx,
                                                       const float* __restrict__ addy) {
  const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
  const float sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  const size_t total = (size_t)N * Ho * Wo * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    size_t r = i / C4;
    const int x = (int)(r % Wo); r /= Wo;
    const int y = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const float fy = sy * y, fx = sx * x;
    int y0 = (int)fy, x0 = (int)fx;
    y0 = y0 < Hi - 1 ? y0 : Hi - 1;
    x0 = x0 < Wi - 1 ? x0 : Wi - 1;
    const int y1 = y0 < Hi - 1 ? y0 + 1 : y0, x1 = x0 < Wi - 1 ? x0 + 1 : x0;
    const float wy = fy - y0, wx = fx - x0;
    const size_t C = (size_t)C4 * 4;
    const float* b = in + (size_t)n * Hi * Wi * C + c;
    const float4 v00 = *(const float4*)(b + ((size_t)y0 * Wi + x0) * C);
    const float4 v01 = *(const float4*)(b + ((size_t)y0 * Wi + x1) * C);
    const float4 v10 = *(const float4*)(b + ((size_t)y1 * Wi + x0) * C);
    const float4 v11 = *(const float4*)(b + ((size_t)y1 * Wi + x1) * C);
    const float w00 = (1.f - wy) * (1.f - wx), w01 = (1.f - wy) * wx, w10 = wy * (1.f - wx), w11 = wy * wx;
    float4 o;
    o.x = w00 * v00.x + w01 * v01.x + w10 * v10.x + w11 * v11.x;
    o.y = w00 * v00.y + w01 * v01.y + w10 * v10.y + w11 * v11.y;
    o.z = w00 * v00.z + w01 * v01.z + w10 * v10.z + w11 * v11.z;
    o.w = w00 * v00.w + w01 * v01.w + w10 * v10.w + w11 * v11.w;
    if (addx) {  // separable UV position embedding: first C/2 channels depend on x, the rest on y
      const int half = (int)(C >> 1);
      const float4 a = c < half ? *(const float4*)(addx + (size_t)x * half + c) : *(const float4*)(addy + (size_t)y * half + (c - half));
      o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    }
    const size_t oi = (((size_t)n * Ho + y) * Wo + x) * C + c;
    if constexpr (OUT16 == 0) {
      *(float4*)((float*)outp + oi) = o;
    } else {
      uint2 u;
      if constexpr (OUT16 == 1) { u.x = (uint32_t)f2bf(o.x) | ((uint32_t)f2bf(o.y) << 16); u.y = (uint32_t)f2bf(o.z) | ((uint32_t)f2bf(o.w) << 16); }
      else { u.x = (uint32_t)f2h(o.x) | ((uint32_t)f2h(o.y) << 16); u.y = (uint32_t)f2h(o.z) | ((uint32_t)f2h(o.w) << 16); }
      *(uint2*)((u16*)outp + oi) = u;
    }
  }
}

// The same resize, LDS-tiled: a block owns a 16 x 16 output tile; per 64-channel chunk the source patch under it (at most
// 12 x 12 pixels when the scale is <= ~0.66, i.e. an upsample by >= 1.5) is read ONCE, coalesced, into LDS and the 4-corner
// blend reads it from there.  The gather kernel above fetches every source value ~5 times through L1/L2 (4.4 GB of L2 -> CU
// traffic for 0.36 GB of source at 296 -> 518, 128 channels, 8 views: 2.1 TB/s of useful HBM traffic).  Same arithmetic, same
// results.  Thread t of a pass owns (pixel = item >> 3, 8 channels = item & 7): 16-B (16-bit) or 2 x 16-B (fp32) stores.
constexpr int RT_P = 12;  // max patch edge
template <int OUT16>
__global__ __launch_bounds__(256) void bilinear_tiled_kernel(const float* __restrict__ in, void* __restrict__ outp, int N, int Hi, int Wi,
                                                             int Ho, int Wo, int C, const float* __restrict__ addx,
                                                             const float* __restrict__ addy) {
  __shared__ __attribute__((aligned(16))) float patch[RT_P * RT_P * 64];
  const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
  const float sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  const int tiles_x = (Wo + 15) / 16, tiles_y = (Ho + 15) / 16;
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int n = t / tiles_y;
  const int y0 = ty * 16, x0 = tx * 16;
  const int y_last = min(y0 + 15, Ho - 1), x_last = min(x0 + 15, Wo - 1);
  auto lo = [](float s, int o, int lim) { int v = (int)(s * o); return v < lim - 1 ? v : lim - 1; };
  const int py0 = lo(sy, y0, Hi), px0 = lo(sx, x0, Wi);
  const int py1 = min(lo(sy, y_last, Hi) + 1, Hi - 1), px1 = min(lo(sx, x_last, Wi) + 1, Wi - 1);
  const int ph = py1 - py0 + 1, pw = px1 - px0 + 1;  // <= RT_P (checked by the launcher)
  const int tid = threadIdx.x;
  const float* src = in + (size_t)n * Hi * Wi * C;
  for (int c0 = 0; c0 < C; c0 += 64) {
    __syncthreads();  // the previous chunk's reads are done
    const int npieces = ph * pw * 16;
    for (int q = tid; q < npieces; q += 256) {
      const int pix = q >> 4, k = q & 15;
      const int yy = pix / pw, xx = pix - yy * pw;
      *(float4*)(patch + pix * 64 + k * 4) = *(const float4*)(src + ((size_t)(py0 + yy) * Wi + (px0 + xx)) * C + c0 + k * 4);
    }
    __syncthreads();
#pragma unroll 2
    for (int it = 0; it < 8; ++it) {
      const int item = it * 256 + tid, pix = item >> 3, g = item & 7;
      const int y = y0 + (pix >> 4), x = x0 + (pix & 15);
      if (y >= Ho || x >= Wo) continue;
      const float fy = sy * y, fx = sx * x;
      int ya = (int)fy, xa = (int)fx;
      ya = ya < Hi - 1 ? ya : Hi - 1;
      xa = xa < Wi - 1 ? xa : Wi - 1;
      const int yb = ya < Hi - 1 ? ya + 1 : ya, xb = xa < Wi - 1 ? xa + 1 : xa;
      const float wy = fy - ya, wx = fx - xa;
      const float w00 = (1.f - wy) * (1.f - wx), w01 = (1.f - wy) * wx, w10 = wy * (1.f - wx), w11 = wy * wx;
      const float* p00 = patch + ((ya - py0) * pw + (xa - px0)) * 64 + g * 8;
      const float* p01 = patch + ((ya - py0) * pw + (xb - px0)) * 64 + g * 8;
      const float* p10 = patch + ((yb - py0) * pw + (xa - px0)) * 64 + g * 8;
      const float* p11 = patch + ((yb - py0) * pw + (xb - px0)) * 64 + g * 8;
      float o[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4 v00 = *(const float4*)(p00 + 4 * h), v01 = *(const float4*)(p01 + 4 * h);
        const float4 v10 = *(const float4*)(p10 + 4 * h), v11 = *(const float4*)(p11 + 4 * h);
        o[4 * h + 0] = w00 * v00.x + w01 * v01.x + w10 * v10.x + w11 * v11.x;
        o[4 * h + 1] = w00 * v00.y + w01 * v01.y + w10 * v10.y + w11 * v11.y;
        o[4 * h + 2] = w00 * v00.z + w01 * v01.z + w10 * v10.z + w11 * v11.z;
        o[4 * h + 3] = w00 * v00.w + w01 * v01.w + w10 * v10.w + w11 * v11.w;
      }
      const int c = c0 + g * 8;
      if (addx) {
        const int half = C >> 1;
        const float* a = c < half ? addx + (size_t)x * half + c : addy + (size_t)y * half + (c - half);
        const float4 a0 = *(const float4*)a, a1 = *(const float4*)(a + 4);
        o[0] += a0.x; o[1] += a0.y; o[2] += a0.z; o[3] += a0.w; o[4] += a1.x; o[5] += a1.y; o[6] += a1.z; o[7] += a1.w;
      }
      const size_t oi = (((size_t)n * Ho + y) * Wo + x) * C + c;
      if constexpr (OUT16 == 0) {
        *(float4*)((float*)outp + oi) = make_float4(o[0], o[1], o[2], o[3]);
        *(float4*)((float*)outp + oi + 4) = make_float4(o[4], o[5], o[6], o[7]);
      } else {
        uint4 u;
        if constexpr (OUT16 == 1) {
          u.x = (uint32_t)f2bf(o[0]) | ((uint32_t)f2bf(o[1]) << 16); u.y = (uint32_t)f2bf(o[2]) | ((uint32_t)f2bf(o[3]) << 16);
          u.z = (uint32_t)f2bf(o[4]) | ((uint32_t)f2bf(o[5]) << 16); u.w = (uint32_t)f2bf(o[6]) | ((uint32_t)f2bf(o[7]) << 16);
        } else {
          u.x = (uint32_t)f2h(o[0]) | ((uint32_t)f2h(o[1]) << 16); u.y = (uint32_t)f2h(o[2]) | ((uint32_t)f2h(o[3]) << 16);
          u.z = (uint32_t)f2h(o[4]) | ((uint32_t)f2h(o[5]) << 16); u.w = (uint32_t)f2h(o[6]) | ((uint32_t)f2h(o[7]) << 16);
        }
        *(uint4*)((u16*)outp + oi) = u;
      }
    }
  }
}
// the tiled kernel applies when the source patch of a 16 x 16 output tile fits RT_P x RT_P and the channels come in 64s
static bool bilinear_tiled_ok(int Hi, int Wi, int Ho, int Wo, int C) {
  if (C % 64 || Ho < 16 || Wo < 16) return false;
  const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  return 15.f * sy + 3.f <= (float)RT_P && 15.f * sx + 3.f <= (float)RT_P;
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ o, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 x = ((const float4*)a)[i], y = ((const float4*)b)[i];
    ((float4*)o)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
}

// dense_head.py:97-105,297-344: ReLU -> 1x1 conv 32->C -> split (attr C-1, conf) -> activations.
__global__ __launch_bounds__(256) void dpt_tail_kernel(const float* __restrict__ y32, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ attr,
                                                       float* __restrict__ conf, size_t npix, int C, int act) {
  __shared__ float sw[4 * 32 + 4];
  if (threadIdx.x < C * 32) sw[threadIdx.x] = w[threadIdx.x];
  if (threadIdx.x < C) sw[128 + threadIdx.x] = b[threadIdx.x];
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
    float o[4] = {sw[128], sw[129], sw[130], sw[131]};
    const float4* src = (const float4*)(y32 + i * 32);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float4 v = src[k];
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < C) o[c] += v.x * sw[c * 32 + 4 * k] + v.y * sw[c * 32 + 4 * k + 1] + v.z * sw[c * 32 + 4 * k + 2] + v.w * sw[c * 32 + 4 * k + 3];
    }
    const int A = C - 1;
    if (act == WM_ACT_NORM) {
      float nn = 0.f;
      for (int c = 0; c < A; ++c) nn += o[c] * o[c];
      nn = sqrtf(nn);
      for (int c = 0; c < A; ++c) attr[i * A + c] = o[c] / nn;
    } else if (act == WM_ACT_EXP) {
      for (int c = 0; c < A; ++c) attr[i * A + c] = expf(o[c]);
    } else {
      for (int c = 0; c < A; ++c) {
        const float e = expm1f(fabsf(o[c]));
        attr[i * A + c] = o[c] > 0.f ? e : (o[c] < 0.f ? -e : 0.f);
      }
    }
    conf[i] = 1.0f + expf(o[A]);
  }
}

inline int grid_for(size_t n, int per_block = 256) {
  size_t g = (n + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

hipError_t wm_launch_layernorm(const WmLnArgs& a, hipStream_t s) {
  const long long rows = (long long)a.groups * a.rows_per_group;
  if (rows <= 0) return hipSuccess;
  if (a.D % 4 || a.D > 2048 || a.ld_in % 4 || a.ld_out % 4) return hipErrorInvalidValue;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const int nv = (a.D + 255) / 256;
  if (a.D == nv * 256 && (nv == 4 || nv == 8) && a.w && a.b && !a.out_f32) {   // the backbone's LayerNorms (D = 1024) and the DPT heads' (D = 2048)
    // rows per wave: 1; 2 (tuning ln_rpw, D = 1024 only: one round of wave slots instead of 1.34 at 8 views) measured the same 12.5 us
    // (tools/bench_ln.py) and stays an A/B variant
    const int rpw = nv == 4 && wm_tuning[WM_TUNE_LN_RPW] == 2 ? 2 : 1;
    const dim3 g2((unsigned)((rows + 4 * rpw - 1) / (4 * rpw)));
    if (nv == 4 && rpw == 2) {
      if (a.dtype == WM_T_BF16) hipLaunchKernelGGL((layernorm_fast_kernel<4, WM_T_BF16, 2>), g2, block, 0, s, a);
      else hipLaunchKernelGGL((layernorm_fast_kernel<4, WM_T_F16, 2>), g2, block, 0, s, a);
    } else if (nv == 4) {
      if (a.dtype == WM_T_BF16) hipLaunchKernelGGL((layernorm_fast_kernel<4, WM_T_BF16, 1>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((layernorm_fast_kernel<4, WM_T_F16, 1>), grid, block, 0, s, a);
    } else {
      if (a.dtype == WM_T_BF16) hipLaunchKernelGGL((layernorm_fast_kernel<8, WM_T_BF16, 1>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((layernorm_fast_kernel<8, WM_T_F16, 1>), grid, block, 0, s, a);
    }
    return hipGetLastError();
  }
  if (nv <= 1) hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, a);
  else if (nv <= 2) hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, a);
  else if (nv <= 4) hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, s, a);
  return hipGetLastError();
}

hipError_t wm_launch_qkv_post(const WmQkvArgs& a, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  const long long waves = (long long)a.M * ((a.H + 3) / 4);
  dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  if (a.dtype == WM_T_BF16) hipLaunchKernelGGL(qkv_post_kernel<WM_T_BF16>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(qkv_post_kernel<WM_T_F16>, grid, block, 0, s, a);
  return hipGetLastError();
}

hipError_t wm_launch_im2col(const float* img, void* out, int N, int C, int H, int W, int ps, int Kpad, int normalize,
                            int dtype, hipStream_t s) {
  const size_t total = (size_t)N * (H / ps) * (W / ps) * Kpad;
  if (!total) return hipSuccess;
  static const bool rows_env = [] { const char* e = wm_env("WM_IM2COL_ROWS"); return !e || atoi(e) != 0; }();
  const size_t lds = (size_t)ps * W * sizeof(u16);
  if (rows_env && ps % 2 == 0 && W % 2 == 0 && (C * ps * ps) % 2 == 0 && Kpad % 2 == 0 && lds <= 64 * 1024) {   // input-indexed form (coalesced image reads)
    const dim3 grid((unsigned)(N * (H / ps) * C));
    if (dtype == WM_T_BF16) hipLaunchKernelGGL(im2col_rows_kernel<WM_T_BF16>, grid, dim3(256), lds, s, img, (u16*)out, N, C, H, W, ps, Kpad, normalize);
    else hipLaunchKernelGGL(im2col_rows_kernel<WM_T_F16>, grid, dim3(256), lds, s, img, (u16*)out, N, C, H, W, ps, Kpad, normalize);
    return hipGetLastError();
  }
  if (dtype == WM_T_BF16)
    hipLaunchKernelGGL(im2col_kernel<WM_T_BF16>, dim3(grid_for(total)), dim3(256), 0, s, img, (u16*)out, N, C, H, W, ps, Kpad, normalize);
  else
    hipLaunchKernelGGL(im2col_kernel<WM_T_F16>, dim3(grid_for(total)), dim3(256), 0, s, img, (u16*)out, N, C, H, W, ps, Kpad, normalize);
  return hipGetLastError();
}

hipError_t wm_launch_im2col7(const float* img, void* out, int N, int H, int W, int Kpad, int dtype, hipStream_t s) {
  const size_t total = (size_t)N * H * W * Kpad;
  if (!total) return hipSuccess;
  if (dtype == WM_T_BF16) hipLaunchKernelGGL(im2col7_kernel<WM_T_BF16>, dim3(grid_for(total)), dim3(256), 0, s, img, (u16*)out, N, H, W, Kpad);
  else hipLaunchKernelGGL(im2col7_kernel<WM_T_F16>, dim3(grid_for(total)), dim3(256), 0, s, img, (u16*)out, N, H, W, Kpad);
  return hipGetLastError();
}

hipError_t wm_launch_gs_splat(const float* gp, const float* img, const float* depth, const float* cam, float* means, float* quats,
                              float* scales, float* opac, float* sh, float* wts, int N, int H, int W, hipStream_t s) {
  if (!N) return hipSuccess;
#ifdef WM_DBG_SPLAT_BUILD
  static const int dbg = wm_env("WM_DBG_SPLAT") ? 1 : 0;
#else
  const int dbg = 0;
#endif
  hipLaunchKernelGGL(gs_splat_kernel, dim3(grid_for((size_t)N * H * W)), dim3(256), 0, s, gp, img, depth, cam, means, quats, scales,
                     opac, sh, wts, N, H, W, dbg);
  return hipGetLastError();
}

hipError_t wm_launch_dino_tokens(const float* /*patch*/, const float* cls, const float* reg, const float* pos, float* X,
                                 int N, int hw, int R, int D, hipStream_t s) {
  hipLaunchKernelGGL(dino_special_kernel, dim3(grid_for((size_t)N * (1 + R) * D)), dim3(256), 0, s, cls, reg, pos, X, N,
                     1 + R + hw, R, D);
  return hipGetLastError();
}

hipError_t wm_launch_vgt_special(float* X, const float* cam_tok, const float* reg_tok, const float* pose_tok,
                                 const float* ray_tok, int N, int P, int R, int D, int cond, int first_view_global,
                                 hipStream_t s) {
  hipLaunchKernelGGL(vgt_special_kernel, dim3(grid_for((size_t)N * (3 + R) * D)), dim3(256), 0, s, X, cam_tok, reg_tok,
                     pose_tok, ray_tok, N, P, R, D, cond, first_view_global);
  return hipGetLastError();
}

hipError_t wm_launch_copy2d(const float* src, float* dst, int rows, int cols, int ld_src, int ld_dst, hipStream_t s) {
  if (cols % 4 || ld_src % 4 || ld_dst % 4) return hipErrorInvalidValue;
  if (!rows) return hipSuccess;
  hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for((size_t)rows * cols / 4)), dim3(256), 0, s, src, dst, rows, cols / 4, ld_src, ld_dst);
  return hipGetLastError();
}

hipError_t wm_launch_bilinear(const float* in, float* out, int N, int Hi, int Wi, int Ho, int Wo, int C, const float* addx,
                              const float* addy, hipStream_t s) {
  if (C % 8) return hipErrorInvalidValue;
  const size_t total = (size_t)N * Ho * Wo * (C / 4);
  if (!total) return hipSuccess;
  static const int tiled_env = [] { const char* e = wm_env("WM_BILINEAR_TILED"); return e ? atoi(e) : 1; }();
  if (tiled_env && bilinear_tiled_ok(Hi, Wi, Ho, Wo, C)) {
    hipLaunchKernelGGL(bilinear_tiled_kernel<0>, dim3((unsigned)(N * ((Ho + 15) / 16) * ((Wo + 15) / 16))), dim3(256), 0, s, in, (void*)out, N, Hi, Wi, Ho, Wo, C, addx, addy);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(bilinear_kernel<0>, dim3(grid_for(total)), dim3(256), 0, s, in, (void*)out, N, Hi, Wi, Ho, Wo, C / 4, addx, addy);
  return hipGetLastError();
}

hipError_t wm_launch_bilinear16(const float* in, void* out16, int N, int Hi, int Wi, int Ho, int Wo, int C, const float* addx,
                                const float* addy, int dtype, hipStream_t s) {
  if (C % 4) return hipErrorInvalidValue;
  const size_t total = (size_t)N * Ho * Wo * (C / 4);
  if (!total) return hipSuccess;
  static const int tiled_env = [] { const char* e = wm_env("WM_BILINEAR_TILED"); return e ? atoi(e) : 1; }();
  if (tiled_env && bilinear_tiled_ok(Hi, Wi, Ho, Wo, C)) {
    const dim3 grid((unsigned)(N * ((Ho + 15) / 16) * ((Wo + 15) / 16)));
    if (dtype == WM_T_BF16) hipLaunchKernelGGL(bilinear_tiled_kernel<1>, grid, dim3(256), 0, s, in, out16, N, Hi, Wi, Ho, Wo, C, addx, addy);
    else hipLaunchKernelGGL(bilinear_tiled_kernel<2>, grid, dim3(256), 0, s, in, out16, N, Hi, Wi, Ho, Wo, C, addx, addy);
    return hipGetLastError();
  }
  if (dtype == WM_T_BF16) hipLaunchKernelGGL(bilinear_kernel<1>, dim3(grid_for(total)), dim3(256), 0, s, in, out16, N, Hi, Wi, Ho, Wo, C / 4, addx, addy);
  else hipLaunchKernelGGL(bilinear_kernel<2>, dim3(grid_for(total)), dim3(256), 0, s, in, out16, N, Hi, Wi, Ho, Wo, C / 4, addx, addy);
  return hipGetLastError();
}

hipError_t wm_launch_add(const float* a, const float* b, float* out, size_t n, hipStream_t s) {
  if (n % 4) return hipErrorInvalidValue;
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, a, b, out, n / 4);
  return hipGetLastError();
}

// depth_to_world_coords_points (src/models/utils/geometry.py:5-89; callers infer.py:303, app.py:151): per pixel
//   cam = ((u - cx) z / fx, (v - cy) z / fy, z),  world = R cam + t  (extrinsic = camera-to-world),  mask = z > eps.
// HBM-bound: 4 B read, 12 + 12 + 1 B written per pixel.  Same operation order as the reference's fp32 expressions
// ((u - cx) * z, then / fx; no contraction of the mul into the divide).
// VEC = 4: one thread owns 4 consecutive pixels -> one 16-B depth load, three 16-B stores per output tensor.
template <int VEC>
__global__ __launch_bounds__(256) void depth_to_world_kernel(const float* __restrict__ depth, const float* __restrict__ ext,
                                                             const float* __restrict__ intr, float* __restrict__ world,
                                                             float* __restrict__ cam, unsigned char* __restrict__ mask, int B,
                                                             int H, int W, float eps) {
  const size_t hw = (size_t)H * W, total = (size_t)B * hw / VEC;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    float zz[VEC], cw[3 * VEC], cc[3 * VEC];
    if constexpr (VEC == 4) {
      const float4 d4 = *(const float4*)(depth + 4 * i);
      zz[0] = d4.x; zz[1] = d4.y; zz[2] = d4.z; zz[3] = d4.w;
    } else {
      zz[0] = depth[i];
    }
    unsigned int mbits = 0;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const size_t px = VEC * i + e;
      const int b = (int)(px / hw);
      const size_t r = px - (size_t)b * hw;
      const int v = (int)(r / W), u = (int)(r - (size_t)v * W);
      const float* K = intr + b * 9;
      const float* E = ext + b * 16;
      const float z = zz[e];
      const float x = __fdiv_rn(__fmul_rn((float)u - K[2], z), K[0]);
      const float y = __fdiv_rn(__fmul_rn((float)v - K[5], z), K[4]);
      cc[3 * e] = x; cc[3 * e + 1] = y; cc[3 * e + 2] = z;
#pragma unroll
      for (int j = 0; j < 3; ++j) cw[3 * e + j] = E[4 * j] * x + E[4 * j + 1] * y + E[4 * j + 2] * z + E[4 * j + 3];
      mbits |= (z > eps ? 1u : 0u) << (8 * e);
    }
    if constexpr (VEC == 4) {
      if (cam) {
#pragma unroll
        for (int q = 0; q < 3; ++q) *(float4*)(cam + 12 * i + 4 * q) = make_float4(cc[4 * q], cc[4 * q + 1], cc[4 * q + 2], cc[4 * q + 3]);
      }
      if (world) {
#pragma unroll
        for (int q = 0; q < 3; ++q) *(float4*)(world + 12 * i + 4 * q) = make_float4(cw[4 * q], cw[4 * q + 1], cw[4 * q + 2], cw[4 * q + 3]);
      }
      if (mask) *(unsigned int*)(mask + 4 * i) = mbits;
    } else {
      if (cam) { cam[3 * i] = cc[0]; cam[3 * i + 1] = cc[1]; cam[3 * i + 2] = cc[2]; }
      if (world) { world[3 * i] = cw[0]; world[3 * i + 1] = cw[1]; world[3 * i + 2] = cw[2]; }
      if (mask) mask[i] = (unsigned char)mbits;
    }
  }
}

hipError_t wm_launch_depth_to_world(const float* depth, const float* ext, const float* intr, float* world, float* cam,
                                    unsigned char* mask, int B, int H, int W, float eps, hipStream_t s) {
  const size_t total = (size_t)B * H * W;
  if (!total) return hipSuccess;
  const bool al = (((size_t)depth | (size_t)world | (size_t)cam) & 15) == 0 && ((size_t)mask & 3) == 0;
  if (total % 4 == 0 && al)
    hipLaunchKernelGGL(depth_to_world_kernel<4>, dim3(grid_for(total / 4)), dim3(256), 0, s, depth, ext, intr, world, cam, mask, B, H, W, eps);
  else
    hipLaunchKernelGGL(depth_to_world_kernel<1>, dim3(grid_for(total)), dim3(256), 0, s, depth, ext, intr, world, cam, mask, B, H, W, eps);
  return hipGetLastError();
}

hipError_t wm_launch_dpt_tail(const float* y32, const float* w, const float* b, float* attr, float* conf, size_t npix,
                              int C, int act, hipStream_t s) {
  if (C < 2 || C > 4) return hipErrorInvalidValue;
  if (!npix) return hipSuccess;
  hipLaunchKernelGGL(dpt_tail_kernel, dim3(grid_for(npix)), dim3(256), 0, s, y32, w, b, attr, conf, npix, C, act);
  return hipGetLastError();
}
