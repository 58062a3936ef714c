// C[M,N] = A[M,K] * W[N,K]^T with fused epilogues — 16-bit (bf16/f16) MFMA, fp32 accumulate.
//
// Replaces every nn.Linear / 1x1 conv / k==stride ConvTranspose on the hot path
// (reference: src/models/layers/attention.py:50,67; mlp.py:29-35; patch_embed.py:70;
//  heads/dense_head.py:53-68,204-208).  Both operands are K-contiguous ("NT"), which is what
// torch's Linear weight layout gives for free and what the 32x32x16 MFMA fragments want.
//
// Tile: 128x128x64, 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32 tiles.  Global->LDS staging
// is direct LDS-DMA (global_load_lds, 16 B/lane); the XOR swizzle that makes the ds_read_b128
// fragment reads conflict-free is applied to the per-lane SOURCE address (guides §5.4 rule 21).
#include "wm_common.h"
#include "wm_kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

// stage one 128 x 64 (16-bit) operand tile: 16 wave-instructions of 1 KiB (8 rows x 128 B)
__device__ __forceinline__ void stage_tile(const u16* __restrict__ g, int ld, int row0, int nrows, int k0,
                                           char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rg = (wave * 4 + i) * 8;       // first row of this 8-row group (wave-uniform)
    const int r = rg + (lane >> 3);          // tile row this lane fetches
    const int c = (lane & 7) ^ ((r >> 1) & 7);  // source chunk so that LDS[r][p] = G[r][p ^ swz(r)]
    int gr = row0 + r;
    gr = gr < nrows ? gr : nrows - 1;        // clamp: out-of-range rows are masked in the epilogue
    const u16* src = g + (size_t)gr * ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(lds_tile + rg * 128), 16, 0, 0);
  }
}

__device__ __forceinline__ s16x8 lds_frag(const char* tile, int row, int chunk) {
  return *(const s16x8*)(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

template <int T, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const WmGemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A 16K | B 16K]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int lid = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (lid / ntn) * BM, n0 = (lid % ntn) * BN;
  const u16* A = (const u16*)p.A;
  const u16* W = (const u16*)p.W;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
  stage_tile(A, p.lda, m0, p.M, 0, smem, wave, lane);
  stage_tile(W, p.ldw, n0, p.N, 0, smem + TILE_BYTES, wave, lane);
  __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and publishes the tile
  int cur = 0;
  for (int t = 0; t < nk; ++t) {
    const char* tA = smem + cur * 2 * TILE_BYTES;
    const char* tB = tA + TILE_BYTES;
    if (t + 1 < nk) {
      char* nA = smem + (cur ^ 1) * 2 * TILE_BYTES;
      stage_tile(A, p.lda, m0, p.M, (t + 1) * BK, nA, wave, lane);
      stage_tile(W, p.ldw, n0, p.N, (t + 1) * BK, nA + TILE_BYTES, wave, lane);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int ch = 2 * ks + (lane >> 5);
      s16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = lds_frag(tA, wm * 64 + i * 32 + (lane & 31), ch);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = lds_frag(tB, wn * 64 + j * 32 + (lane & 31), ch);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32<T>(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---------------- epilogue ----------------
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + (lane & 31);
    if (col >= p.N) continue;
    const float bias = p.bias ? p.bias[col] : 0.f;
    float gamma = 1.f;
    if constexpr (EPI == WM_EPI_RESID) gamma = p.gamma[col];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= p.M) continue;
        float v = acc[i][j][r] + bias;
        if constexpr (EPI == WM_EPI_F32) {
          ((float*)p.C)[(size_t)row * p.ldc + col] = v;
        } else if constexpr (EPI == WM_EPI_T16) {
          ((u16*)p.C)[(size_t)row * p.ldc + col] = f2t<T>(v);
        } else if constexpr (EPI == WM_EPI_GELU_T16) {
          ((u16*)p.C)[(size_t)row * p.ldc + col] = f2t<T>(gelu_erf(v));
        } else if constexpr (EPI == WM_EPI_RESID) {
          float* c = (float*)p.C + (size_t)row * p.ldc + col;
          *c = *c + gamma * v;
        } else if constexpr (EPI == WM_EPI_ROWMAP_ADD) {
          // out[(row / rpg) * out_group + out_off + row % rpg][col] (+)= v + add[(row % rpg)][col]
          const int g = row / p.rows_per_group, q = row - g * p.rows_per_group;
          const size_t o = ((size_t)g * p.out_group + p.out_off + q) * p.ldc + col;
          float x = v;
          if (p.add) x += p.add[(size_t)q * p.N + col];
          if (p.out16) {
            ((u16*)p.C)[o] = f2t<T>(x);
          } else {
            float* c = (float*)p.C + o;
            if (p.accumulate) x += *c;
            *c = x;
          }
        } else if constexpr (EPI == WM_EPI_CONVT) {
          // k==stride ConvTranspose2d as GEMM: col = (i*ks + j)*Cout + co; row = (n*gh + y)*gw + x
          const int co = col % p.ct_cout, ij = col / p.ct_cout;
          const int ii = ij / p.ct_k, jj = ij - ii * p.ct_k;
          const int hw = p.ct_gh * p.ct_gw;
          const int n = row / hw, yx = row - n * hw, y = yx / p.ct_gw, x = yx - y * p.ct_gw;
          const size_t o = (((size_t)n * p.ct_gh * p.ct_k + (y * p.ct_k + ii)) * (p.ct_gw * p.ct_k) + (x * p.ct_k + jj)) * p.ct_cout + co;
          ((float*)p.C)[o] = acc[i][j][r] + p.bias[co];
        }
      }
    }
  }
}

template <int T>
hipError_t launch_T(const WmGemmArgs& a, hipStream_t s) {
  const int ntn = (a.N + BN - 1) / BN, ntm = (a.M + BM - 1) / BM;
  dim3 grid(ntm * ntn), block(256);
  const size_t shm = 4 * TILE_BYTES;
#define WM_LAUNCH(E)                                                                           \
  case E: {                                                                                    \
    static bool attr = false;                                                                  \
    if (!attr) {                                                                               \
      hipFuncSetAttribute((const void*)gemm_nt_kernel<T, E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
      attr = true;                                                                             \
    }                                                                                          \
    hipLaunchKernelGGL((gemm_nt_kernel<T, E>), grid, block, shm, s, a);                         \
    break;                                                                                     \
  }
  switch (a.epi) {
    WM_LAUNCH(WM_EPI_F32)
    WM_LAUNCH(WM_EPI_T16)
    WM_LAUNCH(WM_EPI_GELU_T16)
    WM_LAUNCH(WM_EPI_RESID)
    WM_LAUNCH(WM_EPI_ROWMAP_ADD)
    WM_LAUNCH(WM_EPI_CONVT)
    default: return hipErrorInvalidValue;
  }
#undef WM_LAUNCH
  return hipGetLastError();
}

}  // namespace

hipError_t wm_launch_gemm(const WmGemmArgs& a, hipStream_t s) {
  if (a.M <= 0 || a.N <= 0) return hipSuccess;
  if (a.K <= 0 || a.K % BK != 0) return hipErrorInvalidValue;
  if ((a.lda & 7) || (a.ldw & 7)) return hipErrorInvalidValue;  // 16-B aligned rows
  return a.dtype == WM_T_BF16 ? launch_T<WM_T_BF16>(a, s) : launch_T<WM_T_F16>(a, s);
}
