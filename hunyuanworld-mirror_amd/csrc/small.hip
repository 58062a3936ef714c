// fp32 kernels for the camera head and the pose/ray prior encoders (M = number of views, tiny).
//
// The reference runs the camera head in fp32 (worldmirror.py:146) and camera_params is the most
// precision-sensitive output, so this path keeps fp32 weights and arithmetic.  With
// M <= 64 rows these layers are weight-streaming (HBM) bound: 216 M parameters re-read on each of
// the 4 refinement iterations (camera_head.py:84-102).
#include "wm_common.h"
#include "wm_kernels.h"

#include <cstdlib>

namespace {

__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

constexpr int LBM = 64, LBN = 32, LBK = 64, LPAD = 68;

// Y[M][N] (+)= gamma * post(pre(X) W^T + b)
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                         const float* __restrict__ b, float* __restrict__ Y, int M, int N,
                                                         int K, int ldx, int ldy, int pre_act, int post_act,
                                                         const float* __restrict__ gamma, int accumulate) {
  __shared__ __attribute__((aligned(16))) float xs[LBM * LPAD];
  __shared__ __attribute__((aligned(16))) float ws[LBN * LPAD];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int n0 = blockIdx.x * LBN, m0 = blockIdx.y * LBM;
  float acc[4][2] = {};
  for (int k0 = 0; k0 < K; k0 += LBK) {
    // X tile 64 x 64: 1024 float4, 4 per thread
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = tid + i * 256, r = id >> 4, c = (id & 15) * 4;
      float4 v = make_float4(0, 0, 0, 0);
      if (m0 + r < M && k0 + c < K) {
        const float* s = X + (size_t)(m0 + r) * ldx + k0 + c;
        if (k0 + c + 3 < K) v = *(const float4*)s;
        else { v.x = s[0]; if (k0 + c + 1 < K) v.y = s[1]; if (k0 + c + 2 < K) v.z = s[2]; }
        if (pre_act == 1) { v.x = silu(v.x); v.y = silu(v.y); v.z = silu(v.z); v.w = silu(v.w); }
      }
      *(float4*)(xs + r * LPAD + c) = v;
    }
    // W tile 32 x 64: 512 float4, 2 per thread
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int id = tid + i * 256, r = id >> 4, c = (id & 15) * 4;
      float4 v = make_float4(0, 0, 0, 0);
      if (n0 + r < N && k0 + c < K) {
        const float* s = W + (size_t)(n0 + r) * K + k0 + c;
        if (k0 + c + 3 < K && (K & 3) == 0) v = *(const float4*)s;
        else { v.x = s[0]; if (k0 + c + 1 < K) v.y = s[1]; if (k0 + c + 2 < K) v.z = s[2]; if (k0 + c + 3 < K) v.w = s[3]; }
      }
      *(float4*)(ws + r * LPAD + c) = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < LBK; k += 4) {
      float4 xv[4], wv[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) xv[i] = *(const float4*)(xs + (ty * 4 + i) * LPAD + k);
#pragma unroll
      for (int j = 0; j < 2; ++j) wv[j] = *(const float4*)(ws + (tx * 2 + j) * LPAD + k);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] += xv[i].x * wv[j].x + xv[i].y * wv[j].y + xv[i].z * wv[j].z + xv[i].w * wv[j].w;
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + ty * 4 + i, n = n0 + tx * 2 + j;
      if (m >= M || n >= N) continue;
      float v = acc[i][j] + (b ? b[n] : 0.f);
      if (post_act == 1) v = silu(v);
      else if (post_act == 2) v = gelu_erf(v);
      if (gamma) v *= gamma[n];
      float* y = Y + (size_t)m * ldy + n;
      *y = accumulate ? *y + v : v;
    }
}

// Weight-streaming form for M <= 32 rows (the camera head: M = number of views).  A block owns NC output
// columns; its 4 waves split K in 1-KiB (256-float) slices so every W load is a fully coalesced
// 16 B/lane wave-instruction and W is read exactly once; X (tiny) is re-read from L1/L2.
template <int MT, int NC>
__global__ __launch_bounds__(256) void linear_f32_stream_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                                const float* __restrict__ b, float* __restrict__ Y, int M,
                                                                int N, int K, int ldx, int ldy, int pre_act, int post_act,
                                                                const float* __restrict__ gamma, int accumulate) {
  __shared__ float red[4][MT * NC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * NC, m0 = blockIdx.y * MT;
  float acc[MT][NC];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NC; ++j) acc[i][j] = 0.f;
  for (int k0 = wave * 256; k0 < K; k0 += 1024) {
    const int k = k0 + lane * 4;
    if (k < K) {
      float4 w[NC];
#pragma unroll
      for (int j = 0; j < NC; ++j)
        w[j] = n0 + j < N ? *(const float4*)(W + (size_t)(n0 + j) * K + k) : make_float4(0, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (m0 + i < M) {
          float4 x = *(const float4*)(X + (size_t)(m0 + i) * ldx + k);
          if (pre_act == 1) { x.x = silu(x.x); x.y = silu(x.y); x.z = silu(x.z); x.w = silu(x.w); }
#pragma unroll
          for (int j = 0; j < NC; ++j) acc[i][j] += x.x * w[j].x + x.y * w[j].y + x.z * w[j].z + x.w * w[j].w;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const float v = wave_sum(acc[i][j]);
      if (lane == 0) red[wave][i * NC + j] = v;
    }
  __syncthreads();
  if (tid < MT * NC) {
    const int i = tid / NC, j = tid - i * NC;
    const int m = m0 + i, n = n0 + j;
    if (m < M && n < N) {
      float v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid] + (b ? b[n] : 0.f);
      if (post_act == 1) v = silu(v);
      else if (post_act == 2) v = gelu_erf(v);
      if (gamma) v *= gamma[n];
      float* y = Y + (size_t)m * ldy + n;
      *y = accumulate ? *y + v : v;
    }
  }
}

// Weight streaming on the fp32 MFMA (M <= 64 rows, N % 16 == 0, K % (64 KW) == 0: every Linear of the camera trunk at up to
// 64 views).  The VALU form above re-reads X for every pair of output columns (8 X loads per 2 W loads) and needs a
// 64-lane reduction per (row, column); here a wave owns 16 output columns, its lanes hold W[n0 + l%16][k .. k+3] and
// X[16 t + l%16][k .. k+3] (k = kb + 4 (l/16)) straight from 16-B loads, and four v_mfma_f32_16x16x4_f32 per 16 k's and
// 16-row tile t do the dot products — exact fp32 FMA chains, no cross-lane reduction; the KW waves of a block split K and
// meet in LDS.  MT = row tiles (1, 2, 4): the weights are streamed once whatever M is (at 32 views the VALU kernel streamed
// them at 0.5 TB/s: 130 us per layer, 7 ms of the C3 forward).  Four weight loads in flight per lane; 8 and 16 were
// measured no faster (tools/bench_lin.py), so the limit is the access pattern (16 rows x 64 B per load instruction), not
// the depth.
template <int KW, int MT>
__global__ __launch_bounds__(KW * 64) void linear_f32_mfma_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                                  const float* __restrict__ b, float* __restrict__ Y, int M, int N,
                                                                  int K, int ldx, int ldy, int pre_act, int post_act,
                                                                  const float* __restrict__ gamma, int accumulate) {
  __shared__ float red[KW][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 16;
  const int lr = lane & 15, lk = lane >> 4;
  const int kper = K / KW;  // K % (64 KW) == 0 (launcher)
  const float* wp = W + (size_t)(n0 + lr) * K + wave * kper + 4 * lk;
  const float* xp[MT];
  bool xok[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    xok[t] = 16 * t + lr < M;
    xp[t] = X + (size_t)(xok[t] ? 16 * t + lr : 0) * ldx + wave * kper + 4 * lk;
  }
  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int kb = 0; kb < kper; kb += 64) {  // 4 steps of 16 k's in flight
    f32x4 wv[4], xv[MT][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) wv[u] = *(const f32x4*)(wp + kb + 16 * u);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) xv[t][u] = xok[t] ? *(const f32x4*)(xp[t] + kb + 16 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (pre_act == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) xv[t][u][e] = silu(xv[t][u][e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[t][u][e], wv[u][e], acc[t], 0, 0, 0);
      }
  }
  // D[row m = 16 t + 4 lk + r][col n = lr]; one row tile at a time through the same LDS buffer
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    if (t) __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(4 * lk + r) * 16 + lr] = acc[t][r];
    __syncthreads();
    if (tid < 256) {
      const int m = 16 * t + (tid >> 4), n = n0 + (tid & 15);
      if (m < M) {
        float v = b ? b[n] : 0.f;
#pragma unroll
        for (int w = 0; w < KW; ++w) v += red[w][tid];
        if (post_act == 1) v = silu(v);
        else if (post_act == 2) v = gelu_erf(v);
        if (gamma) v *= gamma[n];
        float* y = Y + (size_t)m * ldy + n;
        *y = accumulate ? *y + v : v;
      }
    }
  }
}

// softmax(q k^T / sqrt(hd)) v over S tokens; one wave per (head, query)
__global__ __launch_bounds__(64) void small_attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int S,
                                                             int heads, int hd) {
  extern __shared__ float sc[];  // S scores
  const int lane = threadIdx.x, head = blockIdx.x % heads, i = blockIdx.x / heads;
  const int D = heads * hd;
  const float* q = qkv + (size_t)i * 3 * D + head * hd;
  const float scale = 1.0f / sqrtf((float)hd);
  // scores: the head dimension across the lanes, one wave reduction per key (S is the number of views: a handful; the former
  // one-key-per-lane form ran 8 lanes through hd-long scalar dot products: 12 us at S = 8)
  float mx = -INFINITY;
  for (int j = 0; j < S; ++j) {
    const float* k = qkv + (size_t)j * 3 * D + D + head * hd;
    float part = 0.f;
    for (int d = lane; d < hd; d += 64) part += q[d] * scale * k[d];
    const float sj = wave_sum(part);
    if (lane == (j & 63)) sc[j] = sj;
    mx = fmaxf(mx, sj);   // wave-uniform
  }
  __syncthreads();
  float sum = 0.f;
  for (int j = lane; j < S; j += 64) {
    const float e = expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __syncthreads();
  for (int d = lane; d < hd; d += 64) {
    float o = 0.f;
    for (int j = 0; j < S; ++j) o += sc[j] * qkv[(size_t)j * 3 * D + 2 * D + head * hd + d];
    out[(size_t)i * D + head * hd + d] = o / sum;
  }
}

// camera_head.py:84-88: h = gate * (LN_noaffine(tok) * (1 + scale) + shift) + tok ; one wave per row
WM_NO_PACKED_FP32 __global__ __launch_bounds__(64) void adaln_kernel(const float* __restrict__ tok, const float* __restrict__ mod,
                                                   float* __restrict__ h, int D, float eps) {
  const int lane = threadIdx.x, row = blockIdx.x;
  const float* x = tok + (size_t)row * D;
  const float* m = mod + (size_t)row * 3 * D;
  if (D % 256 == 0 && D <= 2048) {  // the row in registers, 16-B accesses (D = 2048: 8 per lane)
    float4 v[8];
    const int nv = D / 256;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < nv) { v[i] = *(const float4*)(x + (i * 64 + lane) * 4); s += v[i].x + v[i].y + v[i].z + v[i].w; }
    const float mean = wave_sum(s) / D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < nv) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        ss += a * a + b * b + c * c + d * d;
      }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / D + eps);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < nv) {
        const int c = (i * 64 + lane) * 4;
        const float4 sh = *(const float4*)(m + c), scl = *(const float4*)(m + D + c), g = *(const float4*)(m + 2 * D + c);
        float4 o;
        o.x = g.x * ((v[i].x - mean) * rstd * (1.0f + scl.x) + sh.x) + v[i].x;
        o.y = g.y * ((v[i].y - mean) * rstd * (1.0f + scl.y) + sh.y) + v[i].y;
        o.z = g.z * ((v[i].z - mean) * rstd * (1.0f + scl.z) + sh.z) + v[i].z;
        o.w = g.w * ((v[i].w - mean) * rstd * (1.0f + scl.w) + sh.w) + v[i].w;
        *(float4*)(h + (size_t)row * D + c) = o;
      }
    return;
  }
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += x[c];
  const float mean = wave_sum(s) / D;
  float ss = 0.f;
  for (int c = lane; c < D; c += 64) { const float d = x[c] - mean; ss += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / D + eps);
  for (int c = lane; c < D; c += 64) {
    const float ln = (x[c] - mean) * rstd;
    h[(size_t)row * D + c] = m[2 * D + c] * (ln * (1.0f + m[D + c]) + m[c]) + x[c];
  }
}

__global__ void cam_update_kernel(float* pred, const float* delta, float* out, int n, int first) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = i / 9, c = i - s * 9;
  const float v = first ? delta[s * 12 + c] : pred[s * 12 + c] + delta[s * 12 + c];
  pred[s * 12 + c] = v;
  out[i] = c >= 7 ? fmaxf(v, 0.f) : v;  // camera_head.py:116-147: t, quat linear; fov ReLU
}

// camera_utils.py:46-75 + rotation.py:8-38 + worldmirror.py:165-175 (inverse of [R|t;0 0 0 1])
__global__ void cam_matrices_kernel(const float* __restrict__ p, float* __restrict__ poses, float* __restrict__ intrs,
                                    int S, int H, int W) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const float* v = p + s * 9;
  const float i = v[3], j = v[4], k = v[5], r = v[6];
  const float two_s = 2.0f / (i * i + j * j + k * k + r * r);
  float R[9] = {1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)};
  // general 3x3 inverse (the reference calls torch.linalg.inv on the 4x4), then t' = -R^-1 t
  const float c00 = R[4] * R[8] - R[5] * R[7], c01 = R[5] * R[6] - R[3] * R[8], c02 = R[3] * R[7] - R[4] * R[6];
  const float det = R[0] * c00 + R[1] * c01 + R[2] * c02, id = 1.0f / det;
  float Ri[9] = {c00 * id, (R[2] * R[7] - R[1] * R[8]) * id, (R[1] * R[5] - R[2] * R[4]) * id,
                 c01 * id, (R[0] * R[8] - R[2] * R[6]) * id, (R[2] * R[3] - R[0] * R[5]) * id,
                 c02 * id, (R[1] * R[6] - R[0] * R[7]) * id, (R[0] * R[4] - R[1] * R[3]) * id};
  float* o = poses + s * 16;
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) o[a * 4 + b] = Ri[a * 3 + b];
    o[a * 4 + 3] = -(Ri[a * 3] * v[0] + Ri[a * 3 + 1] * v[1] + Ri[a * 3 + 2] * v[2]);
  }
  o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
  float* K = intrs + s * 9;
  for (int a = 0; a < 9; ++a) K[a] = 0.f;
  K[4] = H * 0.5f / tanf(v[7] * 0.5f);
  K[0] = W * 0.5f / tanf(v[8] * 0.5f);
  K[2] = W * 0.5f;
  K[5] = H * 0.5f;
  K[8] = 1.0f;
}

}  // namespace

hipError_t wm_launch_linear_f32(const float* X, const float* W, const float* b, float* Y, int M, int N, int K, int ldx,
                                int ldy, int pre_act, int post_act, const float* gamma, int accumulate, hipStream_t s) {
  if (M <= 0 || N <= 0) return hipSuccess;
  if (ldx % 4) return hipErrorInvalidValue;
  if (M <= 64 && N % 16 == 0 && K % 1024 == 0 && wm_tuning[WM_TUNE_LIN_MFMA] != 0) {  // fp32-MFMA weight streaming
    // waves per block (they split K) chosen so that the launch has ~2000 waves: N / 16 blocks alone would leave the
    // 2048-column layers at 2 waves per CU; row tiles of 16 (M <= 16 / 32 / 64)
#define WM_LINM(KW_, MT_) hipLaunchKernelGGL((linear_f32_mfma_kernel<KW_, MT_>), dim3(N / 16), dim3(KW_ * 64), 0, s, X, W, b, Y, M, N, K, ldx, ldy, pre_act, post_act, gamma, accumulate)
#define WM_LINMT(KW_) do { if (M <= 16) WM_LINM(KW_, 1); else if (M <= 32) WM_LINM(KW_, 2); else WM_LINM(KW_, 4); } while (0)
    if (N <= 2048) WM_LINMT(16);
    else if (N <= 4096) WM_LINMT(8);
    else WM_LINMT(4);
#undef WM_LINMT
#undef WM_LINM
    return hipGetLastError();
  }
  if (K % 4 == 0 && K >= 256) {  // weight-streaming path
#define WM_STREAM(MT)                                                                                              \
  hipLaunchKernelGGL((linear_f32_stream_kernel<MT, 4>), dim3((N + 3) / 4, (M + MT - 1) / MT), dim3(256), 0, s, X, W, b, Y, \
                     M, N, K, ldx, ldy, pre_act, post_act, gamma, accumulate)
    static const int nc_force = [] { const char* e = wm_env("WM_LIN_NC"); return e ? atoi(e) : 0; }();
    const int nc = nc_force ? nc_force : 2;  // measured best (tools/bench_lin.py): ~2 TB/s
    if (M <= 8 && nc == 8) {
      hipLaunchKernelGGL((linear_f32_stream_kernel<8, 8>), dim3((N + 7) / 8, 1), dim3(256), 0, s, X, W, b, Y, M, N, K, ldx, ldy,
                         pre_act, post_act, gamma, accumulate);
    } else if (M <= 8 && nc == 2) {
      hipLaunchKernelGGL((linear_f32_stream_kernel<8, 2>), dim3((N + 1) / 2, 1), dim3(256), 0, s, X, W, b, Y, M, N, K, ldx, ldy,
                         pre_act, post_act, gamma, accumulate);
    } else if (M <= 8 && nc == 1) {
      hipLaunchKernelGGL((linear_f32_stream_kernel<8, 1>), dim3(N, 1), dim3(256), 0, s, X, W, b, Y, M, N, K, ldx, ldy,
                         pre_act, post_act, gamma, accumulate);
    } else if (M <= 8) WM_STREAM(8);
    else if (M <= 16) WM_STREAM(16);
    else WM_STREAM(32);
#undef WM_STREAM
    return hipGetLastError();
  }
  dim3 grid((N + LBN - 1) / LBN, (M + LBM - 1) / LBM), block(256);
  hipLaunchKernelGGL(linear_f32_kernel, grid, block, 0, s, X, W, b, Y, M, N, K, ldx, ldy, pre_act, post_act, gamma, accumulate);
  return hipGetLastError();
}

hipError_t wm_launch_small_attention(const float* qkv, float* out, int S, int heads, int hd, hipStream_t s) {
  if (S <= 0) return hipSuccess;
  if (S > 8192) return hipErrorInvalidValue;
  hipLaunchKernelGGL(small_attention_kernel, dim3(S * heads), dim3(64), S * sizeof(float), s, qkv, out, S, heads, hd);
  return hipGetLastError();
}

hipError_t wm_launch_adaln(const float* tok, const float* mod, float* h, int S, int D, float eps, hipStream_t s) {
  if (S <= 0) return hipSuccess;
  hipLaunchKernelGGL(adaln_kernel, dim3(S), dim3(64), 0, s, tok, mod, h, D, eps);
  return hipGetLastError();
}

hipError_t wm_launch_cam_matrices(const float* params, float* poses, float* intrs, int S, int H, int W, hipStream_t s) {
  if (S <= 0) return hipSuccess;
  hipLaunchKernelGGL(cam_matrices_kernel, dim3((S + 63) / 64), dim3(64), 0, s, params, poses, intrs, S, H, W);
  return hipGetLastError();
}

hipError_t wm_launch_cam_update(float* pred, const float* delta, float* out, int S, int first, hipStream_t s) {
  if (S <= 0) return hipSuccess;
  hipLaunchKernelGGL(cam_update_kernel, dim3((S * 9 + 255) / 256), dim3(256), 0, s, pred, delta, out, S * 9, first);
  return hipGetLastError();
}
