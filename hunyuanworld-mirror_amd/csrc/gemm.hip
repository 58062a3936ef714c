// C[M,N] = A[M,K] * W[N,K]^T with fused epilogues — 16-bit (bf16/f16) MFMA, fp32 accumulate.
//
// Replaces every nn.Linear / 1x1 conv / k==stride ConvTranspose on the hot path
// (reference: src/models/layers/attention.py:50,67; mlp.py:29-35; patch_embed.py:70;
//  heads/dense_head.py:53-68,204-208).  Both operands are K-contiguous ("NT"), which is what
// torch's Linear weight layout gives for free and what the 32x32x16 MFMA fragments want.
//
// Structure
//  * tile (WM*TM*32) x (WN*TN*32) x 64, WM x WN waves, each wave TM x TN MFMA 32x32 tiles;
//  * global -> LDS by LDS-DMA (global_load_lds, 16 B/lane, 1 KiB per wave-instruction); the XOR
//    swizzle that makes the ds_read_b128 fragment reads conflict-free is applied to the per-lane
//    SOURCE address (guides §5.4 rule 21);
//  * NSTAGE-deep LDS ring, one raw s_barrier per K-tile, counted s_waitcnt vmcnt(N) so NSTAGE-2
//    tiles stay in flight across the barrier (guides "Pipelining across barriers", T3+T4);
//  * the MFMA is issued as D = W_frag * A_frag (operands swapped), so a lane owns one output ROW
//    and 4 consecutive COLUMNS per register group: the epilogue reads/writes float4 (16 B/lane).
#include "wm_common.h"
#include "wm_kernels.h"

#include <cstdlib>
#include <type_traits>

int wm_tuning[WM_TUNE_COUNT] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};

namespace {

#ifndef WM_GEMM_PRIO
#define WM_GEMM_PRIO 0
#endif
constexpr int PRIO = WM_GEMM_PRIO;
constexpr int BK = 64;  // default K-tile; the 32x32 kernel is also instantiated with BK = 32 (KT template parameter)

typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

// K-tile 32: rows are 64 B, a 1-KiB piece is 16 rows x 4 chunks; slot swizzle (row>>2)&3 keeps ds_read_b128 conflict-free
__device__ __forceinline__ void stage_piece32(const u16* __restrict__ g, int ld, int row0, int nrows, int k0, char* lds_tile,
                                              int piece, int lane) {
  const int r = piece * 16 + (lane >> 2);
  const int c = (lane & 3) ^ ((r >> 2) & 3);
  int gr = row0 + r;
  gr = gr < nrows ? gr : nrows - 1;
  __builtin_amdgcn_global_load_lds((glb_vp)(g + (size_t)gr * ld + k0 + c * 8), (lds_vp)(lds_tile + piece * 1024), 16, 0, 0);
}
__device__ __forceinline__ s16x8 lds_frag32(const char* tile, int row, int chunk) {
  return *(const s16x8*)(tile + row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4));
}

// one 1-KiB LDS-DMA piece = 8 rows x 128 B of an operand tile
__device__ __forceinline__ void stage_piece(const u16* __restrict__ g, int ld, int row0, int nrows, int k0, char* lds_tile,
                                            int piece, int lane) {
  const int r = piece * 8 + (lane >> 3);
  const int c = (lane & 7) ^ ((r >> 1) & 7);  // LDS[r][p] = G[r][p ^ swz(r)]
  int gr = row0 + r;
  gr = gr < nrows ? gr : nrows - 1;  // clamp: out-of-range rows are masked in the epilogue
  __builtin_amdgcn_global_load_lds((glb_vp)(g + (size_t)gr * ld + k0 + c * 8), (lds_vp)(lds_tile + piece * 1024), 16, 0, 0);
}

__device__ __forceinline__ s16x8 lds_frag(const char* tile, int row, int chunk) {
  return *(const s16x8*)(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

// logical tile id -> (row band, column tile).  With more than 4 column tiles the tiles that run together on an XCD (32 CUs,
// consecutive ids) are arranged as G row bands x 8 column tiles instead of 2 x 16: per round the XCD's L2 then fetches
// (G x A band + 32 / G x W slab) instead of (2 x A + 16 x W) — e.g. fc1 at 8 views 144 instead of 216 MB of L2 misses.
__device__ __forceinline__ void tile_of(int lid, int ntm, int ntn, int G, int& band, int& nt) {
  if (G <= 1 || ntn <= 4) { band = lid / ntn; nt = lid - band * ntn; return; }
  const int per = G * ntn, sg = lid / per, r = lid - sg * per;
  const int here = ntm - sg * G < G ? ntm - sg * G : G;
  nt = r / here;
  band = sg * G + (r - nt * here);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int T, int EPI, int WM, int WN, int TM, int TN, int NSTAGE, int ILV, int KT = 64>
__global__ __launch_bounds__(WM * WN * 64) void gemm_nt_kernel(const WmGemmArgs p) {
  constexpr int BK = KT;
  constexpr int NW = WM * WN, BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int PA = BM * BK * 2 / 1024, PB = BN * BK * 2 / 1024;  // 1-KiB pieces per operand tile
  constexpr int PPW = (PA + PB) / NW;           // pieces per wave per K-tile
  static_assert((PA + PB) % NW == 0, "pieces must divide over waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int lid = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (lid / ntn) * BM, n0 = (lid % ntn) * BN;
  const u16* A = (const u16*)p.A;
  const u16* W = (const u16*)p.W;

  auto stage_part = [&](int kt, int s, int i0, int i1) {
    char* base = smem + s * STAGE;
#pragma unroll
    for (int i = i0; i < i1; ++i) {
      const int pc = wave * PPW + i;  // wave-uniform
      if constexpr (BK == 64) {
        if (pc < PA) stage_piece(A, p.lda, m0, p.M, kt * BK, base, pc, lane);
        else stage_piece(W, p.ldw, n0, p.N, kt * BK, base + A_BYTES, pc - PA, lane);
      } else {
        if (pc < PA) stage_piece32(A, p.lda, m0, p.M, kt * BK, base, pc, lane);
        else stage_piece32(W, p.ldw, n0, p.N, kt * BK, base + A_BYTES, pc - PA, lane);
      }
    }
  };
  auto stage = [&](int kt, int s) { stage_part(kt, s, 0, PPW); };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) stage(s, s);

  for (int t = 0; t < nk; ++t) {
    // tile t must have landed; tiles t+1 .. t+NSTAGE-2 may stay in flight
    const int ahead = min(nk - 1 - t, NSTAGE - 2);
    if (ahead >= 3) wait_vmcnt<3 * PPW>();
    else if (ahead >= 2) wait_vmcnt<2 * PPW>();
    else if (ahead == 1) wait_vmcnt<PPW>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // everyone's pieces of tile t landed; everyone finished tile t-1
    const bool more = t + NSTAGE - 1 < nk;
    if (!ILV && more) stage(t + NSTAGE - 1, (t + NSTAGE - 1) % NSTAGE);
    const char* tA = smem + (t % NSTAGE) * STAGE;
    const char* tB = tA + A_BYTES;
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      if (ILV && more) {  // spread the LDS-DMA issue over the four MFMA groups of this K-tile
        constexpr int Q = (PPW + 3) / 4;
        stage_part(t + NSTAGE - 1, (t + NSTAGE - 1) % NSTAGE, ks * Q < PPW ? ks * Q : PPW, (ks + 1) * Q < PPW ? (ks + 1) * Q : PPW);
      }
      const int ch = 2 * ks + (lane >> 5);
      s16x8 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = BK == 64 ? lds_frag(tA, (wm * TM + i) * 32 + (lane & 31), ch) : lds_frag32(tA, (wm * TM + i) * 32 + (lane & 31), ch);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = BK == 64 ? lds_frag(tB, (wn * TN + j) * 32 + (lane & 31), ch) : lds_frag32(tB, (wn * TN + j) * 32 + (lane & 31), ch);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma32<T>(b[j], a[i], acc[i][j]);  // D[n][m]: lane = row m
      if (ILV) __builtin_amdgcn_sched_barrier(0);
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
  }

  // ---------------- epilogue: lane (m = lane&31, h = lane>>5), reg 4g+e <-> col 8g + 4h + e ----------------
  const int h4 = (lane >> 5) * 4;
  if constexpr (EPI == WM_EPI_QKV) {
    // attention.py:50-56 fused: the wave's 64 columns are exactly one head of q, k or v (TN == 2, tiles
    // aligned to 64): bias -> LayerNorm(64, eps 1e-5, affine) -> 2-D RoPE (rope.py:148-181) -> q scale ->
    // 16-bit store in the head-major layout the attention kernel reads.  A lane holds 32 of its row's 64
    // values (the other 32 sit in lane^32), and every RoPE pair (c, c+16) is lane-local (g <-> g+2).
    static_assert(EPI != WM_EPI_QKV || TN == 2, "QKV epilogue needs 64 columns per wave");
    const WmQkvArgs& q = p.qkv;
    const int D = q.H * 64;
    const int col0 = n0 + wn * 64;             // wave-uniform
    const int which = col0 / D, head = (col0 - which * D) >> 6;
    if (col0 < p.N) {
      const float* nw = which == 0 ? q.qn_w : q.kn_w;
      const float* nb = which == 0 ? q.qn_b : q.kn_b;
      const bool do_norm = which < 2 && nw != nullptr;
      const bool do_rope = which < 2 && q.rope_cos != nullptr;
      u16* dst = (u16*)(which == 0 ? q.q : which == 1 ? q.k : q.v);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = m0 + (wm * TM + i) * 32 + (lane & 31);
        const bool row_ok = row < p.M;
        float v[2][4][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 bs = p.bias ? *(const float4*)(p.bias + col0 + 32 * j + 8 * g + h4) : make_float4(0, 0, 0, 0);
            v[j][g][0] = acc[i][j][4 * g] + bs.x; v[j][g][1] = acc[i][j][4 * g + 1] + bs.y;
            v[j][g][2] = acc[i][j][4 * g + 2] + bs.z; v[j][g][3] = acc[i][j][4 * g + 3] + bs.w;
          }
        if (do_norm) {
          float sm = 0.f;
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
              for (int e = 0; e < 4; ++e) sm += v[j][g][e];
          const float mean = xhalf_sum(sm) * (1.0f / 64.0f);
          float ss = 0.f;
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[j][g][e] -= mean; ss += v[j][g][e] * v[j][g][e]; }
          const float rstd = 1.0f / sqrtf(xhalf_sum(ss) * (1.0f / 64.0f) + 1e-5f);
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 w4 = *(const float4*)(nw + 32 * j + 8 * g + h4), b4 = *(const float4*)(nb + 32 * j + 8 * g + h4);
              v[j][g][0] = v[j][g][0] * rstd * w4.x + b4.x; v[j][g][1] = v[j][g][1] * rstd * w4.y + b4.y;
              v[j][g][2] = v[j][g][2] * rstd * w4.z + b4.z; v[j][g][3] = v[j][g][3] * rstd * w4.w + b4.w;
            }
        }
        if (do_rope) {
          const int rr = row_ok ? row : 0;
          // row / tokens_per_view and idx / grid_w by float multiplication (exact here: rows < 2^20, quotients < 2^11,
          // the +0.5 keeps the product > 1e-4 away from an integer) — two runtime integer divisions were ~80 VALU per row
          const int t = rr - (int)(((float)rr + 0.5f) * q.inv_tpv) * q.tokens_per_view;
          int py = 0, px = 0;
          if (t >= q.patch_start) {
            const int idx = t - q.patch_start;
            py = (int)(((float)idx + 0.5f) * q.inv_gw) + 1;
            px = idx - (py - 1) * q.grid_w + 1;
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int pos = j == 0 ? py : px;  // first 32 channels rotate by y, last 32 by x
#pragma unroll
            for (int g = 0; g < 2; ++g) {      // pair (g, g+2): elements c and c+16; frequency index 8g + 4h + e
              const float4 cs = *(const float4*)(q.rope_cos + pos * 16 + 8 * g + h4);
              const float4 sn = *(const float4*)(q.rope_sin + pos * 16 + 8 * g + h4);
              const float c4[4] = {cs.x, cs.y, cs.z, cs.w}, s4[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float lo = v[j][g][e], hi = v[j][g + 2][e];
                v[j][g][e] = lo * c4[e] - hi * s4[e];
                v[j][g + 2][e] = hi * c4[e] + lo * s4[e];
              }
            }
          }
        }
        const float sc = which == 0 ? q.q_scale : 1.0f;
        if (row_ok) {
          u16* o = dst + ((size_t)head * q.head_stride + row) * 64 + h4;
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              uint2 u;
              u.x = (uint32_t)f2t<T>(v[j][g][0] * sc) | ((uint32_t)f2t<T>(v[j][g][1] * sc) << 16);
              u.y = (uint32_t)f2t<T>(v[j][g][2] * sc) | ((uint32_t)f2t<T>(v[j][g][3] * sc) << 16);
              *(uint2*)(o + 32 * j + 8 * g) = u;
            }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = m0 + (wm * TM + i) * 32 + (lane & 31);
    if (row >= p.M) continue;
    size_t orow = (size_t)row;
    int q = 0;
    if constexpr (EPI == WM_EPI_ROWMAP_ADD) {
      const int g = row / p.rows_per_group;
      q = row - g * p.rows_per_group;
      orow = (size_t)g * p.out_group + p.out_off + q;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int cb = n0 + (wn * TN + j) * 32 + h4;
      if constexpr (EPI == WM_EPI_CONVT) {
        // k==stride ConvTranspose2d as GEMM: col = (ii*k + jj)*Cout + co; row = (n*gh + y)*gw + x
        const int hw = p.ct_gh * p.ct_gw;
        const int n = row / hw, yx = row - n * hw, y = yx / p.ct_gw, x = yx - y * p.ct_gw;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = cb + 8 * g;
          if (col >= p.N) continue;
          const int co = col % p.ct_cout, ij = col / p.ct_cout;  // 4 consecutive cols share (ii,jj): Cout % 4 == 0
          const int ii = ij / p.ct_k, jj = ij - ii * p.ct_k;
          const size_t o = (((size_t)n * p.ct_gh * p.ct_k + (y * p.ct_k + ii)) * (p.ct_gw * p.ct_k) + (x * p.ct_k + jj)) * p.ct_cout + co;
          const float4 bs = *(const float4*)(p.bias + co);
          *(float4*)((float*)p.C + o) = make_float4(acc[i][j][4 * g] + bs.x, acc[i][j][4 * g + 1] + bs.y,
                                                    acc[i][j][4 * g + 2] + bs.z, acc[i][j][4 * g + 3] + bs.w);
        }
      } else {
        float4 v[4];
        bool ok[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = cb + 8 * g;
          ok[g] = col < p.N;  // N % 4 == 0: a float4 is all-in or all-out
          const int cc = ok[g] ? col : 0;
          const float4 bs = p.bias ? *(const float4*)(p.bias + cc) : make_float4(0, 0, 0, 0);
          v[g] = make_float4(acc[i][j][4 * g] + bs.x, acc[i][j][4 * g + 1] + bs.y, acc[i][j][4 * g + 2] + bs.z,
                             acc[i][j][4 * g + 3] + bs.w);
        }
        if constexpr (EPI == WM_EPI_F32) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            if (ok[g]) *(float4*)((float*)p.C + orow * p.ldc + cb + 8 * g) = v[g];
        } else if constexpr (EPI == WM_EPI_T16 || EPI == WM_EPI_GELU_T16) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            if (!ok[g]) continue;
            float4 x = v[g];
            if constexpr (EPI == WM_EPI_GELU_T16) x = gelu_erf4(x);
            uint2 u;
            u.x = (uint32_t)f2t<T>(x.x) | ((uint32_t)f2t<T>(x.y) << 16);
            u.y = (uint32_t)f2t<T>(x.z) | ((uint32_t)f2t<T>(x.w) << 16);
            *(uint2*)((u16*)p.C + orow * p.ldc + cb + 8 * g) = u;
          }
        } else if constexpr (EPI == WM_EPI_RESID) {
          float4 old[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) old[g] = ok[g] ? *(const float4*)((const float*)p.C + orow * p.ldc + cb + 8 * g) : make_float4(0, 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            if (!ok[g]) continue;
            const float4 gm = *(const float4*)(p.gamma + cb + 8 * g);
            const float4 nv = make_float4(old[g].x + gm.x * v[g].x, old[g].y + gm.y * v[g].y, old[g].z + gm.z * v[g].z, old[g].w + gm.w * v[g].w);
            *(float4*)((float*)p.C + orow * p.ldc + cb + 8 * g) = nv;
            if (p.C2) *(float4*)(p.C2 + orow * p.ldc2 + cb + 8 * g) = nv;
          }
        } else if constexpr (EPI == WM_EPI_ROWMAP_ADD) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            if (!ok[g]) continue;
            const int col = cb + 8 * g;
            float4 x = v[g];
            if (p.relu) x = make_float4(fmaxf(x.x, 0.f), fmaxf(x.y, 0.f), fmaxf(x.z, 0.f), fmaxf(x.w, 0.f));
            if (p.add) {
              const float4 ad = *(const float4*)(p.add + (size_t)q * p.N + col);
              x.x += ad.x; x.y += ad.y; x.z += ad.z; x.w += ad.w;
            }
            const size_t o = orow * p.ldc + col;
            if (p.out16) {
              uint2 u;
              u.x = (uint32_t)f2t<T>(x.x) | ((uint32_t)f2t<T>(x.y) << 16);
              u.y = (uint32_t)f2t<T>(x.z) | ((uint32_t)f2t<T>(x.w) << 16);
              *(uint2*)((u16*)p.C + o) = u;
            } else {
              float* c = (float*)p.C + o;
              if (p.accumulate) {
                const float4 od = *(const float4*)c;
                x.x += od.x; x.y += od.y; x.z += od.z; x.w += od.w;
              }
              *(float4*)c = x;
            }
          }
        }
      }
    }
  }
}

template <int T, int EPI, int WM, int WN, int TM, int TN, int NSTAGE, int ILV, int KT = 64>
hipError_t launch_cfg(const WmGemmArgs& a, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr size_t shm = (size_t)NSTAGE * (BM + BN) * KT * 2;
  const int ntn = (a.N + BN - 1) / BN, ntm = (a.M + BM - 1) / BM;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<T, EPI, WM, WN, TM, TN, NSTAGE, ILV, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    attr = true;
  }
  hipLaunchKernelGGL((gemm_nt_kernel<T, EPI, WM, WN, TM, TN, NSTAGE, ILV, KT>), dim3(ntm * ntn), dim3(WM * WN * 64), shm, s, a);
  return hipGetLastError();
}


// Epilogue shared by the 16x16x32 kernels.  acc[i][j] is the 16x16 sub-tile at rows rowb + 16 i, columns colb + 16 j of
// the wave's SM x SN grid; operands were swapped (D = W_frag * A_frag), so lane (l15 = lane & 15, lq = lane >> 4) owns
// row 16 i + l15 and the 4 consecutive columns 16 j + 4 lq .. + 3.
template <int T, int EPI, int SM, int SN>
__device__ __forceinline__ void epilogue16(const WmGemmArgs& p, f32x4 (&acc)[SM][SN], int rowb, int colb, int lane, int mlim) {
  // mlim: first row this wave must NOT write (p.M, or the end of the wave's share of a shortened row band)
  const int l15 = lane & 15, lq = lane >> 4;
  // ---------------- epilogue ----------------
  if constexpr (EPI == WM_EPI_QKV) {
    static_assert(EPI != WM_EPI_QKV || SN == 4, "QKV epilogue needs 64 columns per wave");
    const WmQkvArgs& q = p.qkv;
    const int D = q.H * 64;
    const int col0 = colb;
    const int which = col0 / D, head = (col0 - which * D) >> 6;
    if (col0 >= p.N) return;
    const float* nw = which == 0 ? q.qn_w : q.kn_w;
    const float* nb = which == 0 ? q.qn_b : q.kn_b;
    const bool do_norm = which < 2 && nw != nullptr, do_rope = which < 2 && q.rope_cos != nullptr;
    u16* dst = (u16*)(which == 0 ? q.q : which == 1 ? q.k : q.v);
    const float sc = which == 0 ? q.q_scale : 1.0f;
    // column vectors of this head are the same for every row group: loaded once.  The per-row RoPE table rows are
    // requested one row group ahead, i.e. BEFORE the previous group's stores (the compiler may not move a load across
    // a possibly aliasing store, so loads placed after them would each pay a full L2 round trip per row group).
    float4 bs4[4], w4[4], b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bs4[j] = p.bias ? *(const float4*)(p.bias + col0 + 16 * j + 4 * lq) : make_float4(0, 0, 0, 0);
      w4[j] = do_norm ? *(const float4*)(nw + 16 * j + 4 * lq) : make_float4(1, 1, 1, 1);
      b4[j] = do_norm ? *(const float4*)(nb + 16 * j + 4 * lq) : make_float4(0, 0, 0, 0);
    }
    auto rope_rows = [&](int i, float4 (&cs)[2], float4 (&sn)[2]) {
      const int row = rowb + i * 16 + l15;
      const int rr = row < mlim ? row : 0;
      const int t = rr - (int)(((float)rr + 0.5f) * q.inv_tpv) * q.tokens_per_view;  // see gemm_nt_kernel
      int py = 0, px = 0;
      if (t >= q.patch_start) {
        const int idx = t - q.patch_start;
        py = (int)(((float)idx + 0.5f) * q.inv_gw) + 1;
        px = idx - (py - 1) * q.grid_w + 1;
      }
      cs[0] = *(const float4*)(q.rope_cos + py * 16 + 4 * lq); sn[0] = *(const float4*)(q.rope_sin + py * 16 + 4 * lq);
      cs[1] = *(const float4*)(q.rope_cos + px * 16 + 4 * lq); sn[1] = *(const float4*)(q.rope_sin + px * 16 + 4 * lq);
    };
    float4 csn[2][2], snn[2][2];  // [parity of the row group][y | x]
    if (do_rope) rope_rows(0, csn[0], snn[0]);
#pragma unroll
    for (int i = 0; i < SM; ++i) {
      if (rowb + i * 16 >= mlim) break;  // wave-uniform: the rest of the wave's rows are past the band / matrix end
      const int row = rowb + i * 16 + l15;
      const bool row_ok = row < mlim;
      if (do_rope && i + 1 < SM) rope_rows(i + 1, csn[(i + 1) & 1], snn[(i + 1) & 1]);
      float v[4][4];  // [j: 16-col group][e]: column 16j + 4*lq + e of this head
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j][0] = acc[i][j][0] + bs4[j].x; v[j][1] = acc[i][j][1] + bs4[j].y; v[j][2] = acc[i][j][2] + bs4[j].z; v[j][3] = acc[i][j][3] + bs4[j].w;
      }
      if (do_norm) {  // the row's 64 values live in the 4 lanes l15 + 16*{0..3}
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) sm += v[j][e];
        sm += __shfl_xor(sm, 16);
        const float mean = xhalf_sum(sm) * (1.0f / 64.0f);
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[j][e] -= mean; ss += v[j][e] * v[j][e]; }
        ss += __shfl_xor(ss, 16);
        const float rstd = 1.0f / sqrtf(xhalf_sum(ss) * (1.0f / 64.0f) + 1e-5f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j][0] = v[j][0] * rstd * w4[j].x + b4[j].x; v[j][1] = v[j][1] * rstd * w4[j].y + b4[j].y;
          v[j][2] = v[j][2] * rstd * w4[j].z + b4[j].z; v[j][3] = v[j][3] * rstd * w4[j].w + b4[j].w;
        }
      }
      if (do_rope) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {  // columns [0,32) rotate by y, [32,64) by x; pair (c, c+16) = groups (2hh, 2hh+1)
          const float4 cs = csn[i & 1][hh], sn = snn[i & 1][hh];
          const float c4[4] = {cs.x, cs.y, cs.z, cs.w}, s4[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = v[2 * hh][e], hi = v[2 * hh + 1][e];
            v[2 * hh][e] = lo * c4[e] - hi * s4[e];
            v[2 * hh + 1][e] = hi * c4[e] + lo * s4[e];
          }
        }
      }
      // 16-B stores (swap16: even quads take sub-tile 2jp, odd quads 2jp + 1); all lanes take part in the swaps
      uint2 u[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u[j].x = (uint32_t)f2t<T>(v[j][0] * sc) | ((uint32_t)f2t<T>(v[j][1] * sc) << 16);
        u[j].y = (uint32_t)f2t<T>(v[j][2] * sc) | ((uint32_t)f2t<T>(v[j][3] * sc) << 16);
      }
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        swap16(u[2 * jp].x, u[2 * jp + 1].x);
        swap16(u[2 * jp].y, u[2 * jp + 1].y);
      }
      if (row_ok) {
        u16* o = dst + ((size_t)head * q.head_stride + row) * 64 + 16 * (lq & 1) + 4 * (lq & 2);
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) *(uint4*)(o + 32 * jp) = make_uint4(u[2 * jp].x, u[2 * jp].y, u[2 * jp + 1].x, u[2 * jp + 1].y);
      }
    }
    return;
  } else {
    // Column vectors (bias, LayerScale gamma) depend on j only: loaded once, ahead of every store (the compiler may not
    // move a load across a possibly aliasing store, so a load inside the (i, j) loop pays its latency per iteration).
    float4 bs4[SN], gm4[SN];
#pragma unroll
    for (int j = 0; j < SN; ++j) {
      const int col = colb + j * 16 + 4 * lq;
      bs4[j] = (p.bias && col < p.N) ? *(const float4*)(p.bias + col) : make_float4(0, 0, 0, 0);
      if constexpr (EPI == WM_EPI_RESID) gm4[j] = col < p.N ? *(const float4*)(p.gamma + col) : make_float4(0, 0, 0, 0);
    }
    if constexpr ((EPI == WM_EPI_T16 || EPI == WM_EPI_GELU_T16) && SN % 2 == 0) {
      // 16-bit outputs: pair the sub-tiles (2jp, 2jp + 1) through swap16 and store 16 B per lane
      if ((p.N & 7) == 0 && (p.ldc & 7) == 0 && ((uintptr_t)p.C & 15) == 0) {
#pragma unroll
        for (int i = 0; i < SM; ++i) {
          if (rowb + i * 16 >= mlim) break;  // wave-uniform: the rest of the wave's rows are past the band / matrix end
          const int row = rowb + i * 16 + l15;
#pragma unroll
          for (int jp = 0; jp < SN / 2; ++jp) {
            uint2 u[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int j = 2 * jp + h;
              float4 x = make_float4(acc[i][j][0] + bs4[j].x, acc[i][j][1] + bs4[j].y, acc[i][j][2] + bs4[j].z, acc[i][j][3] + bs4[j].w);
              if constexpr (EPI == WM_EPI_GELU_T16) x = gelu_erf4(x);
              u[h].x = (uint32_t)f2t<T>(x.x) | ((uint32_t)f2t<T>(x.y) << 16);
              u[h].y = (uint32_t)f2t<T>(x.z) | ((uint32_t)f2t<T>(x.w) << 16);
            }
            swap16(u[0].x, u[1].x);
            swap16(u[0].y, u[1].y);
            const int col = colb + (2 * jp + (lq & 1)) * 16 + 4 * (lq & 2);  // 8 columns from here
            if (row < mlim && col < p.N) *(uint4*)((u16*)p.C + (size_t)row * p.ldc + col) = make_uint4(u[0].x, u[0].y, u[1].x, u[1].y);
          }
        }
        return;
      }
    }
    if constexpr (EPI == WM_EPI_RESID) {
      // in-place X += gamma * (acc + bias): the old values of row group i + 1 are requested before row group i is stored
      float4 old[2][SN];
      auto load_old = [&](int i, float4 (&o)[SN]) {
        const int row = rowb + i * 16 + l15;
#pragma unroll
        for (int j = 0; j < SN; ++j) {
          const int col = colb + j * 16 + 4 * lq;
          o[j] = (row < mlim && col < p.N) ? *(const float4*)((const float*)p.C + (size_t)row * p.ldc + col) : make_float4(0, 0, 0, 0);
        }
      };
      load_old(0, old[0]);
#pragma unroll
      for (int i = 0; i < SM; ++i) {
        if (rowb + i * 16 >= mlim) break;  // wave-uniform: the rest of the wave's rows are past the band / matrix end
        if (i + 1 < SM) load_old(i + 1, old[(i + 1) & 1]);
        const int row = rowb + i * 16 + l15;
#pragma unroll
        for (int j = 0; j < SN; ++j) {
          const int col = colb + j * 16 + 4 * lq;
          const float4 o = old[i & 1][j], gm = gm4[j], bs = bs4[j];
          if (row < mlim && col < p.N) {
            const float4 nv = make_float4(o.x + gm.x * (acc[i][j][0] + bs.x), o.y + gm.y * (acc[i][j][1] + bs.y), o.z + gm.z * (acc[i][j][2] + bs.z), o.w + gm.w * (acc[i][j][3] + bs.w));
            *(float4*)((float*)p.C + (size_t)row * p.ldc + col) = nv;
            if (p.C2) *(float4*)(p.C2 + (size_t)row * p.ldc2 + col) = nv;  // tap half (block-uniform branch)
          }
        }
      }
      return;
    }
    if constexpr (EPI == WM_EPI_CONV) {
      // y = acc + bias + relu?(resid) + resid2 (ResidualConvUnit skip and fusion add, dense_head.py:435-455), optional ReLU; fp32 NHWC, or
      // 16-bit when the only consumer rounds to the operand type anyway.  The residuals of row group i + 1 are requested before row
      // group i is stored (the output may alias neither).
      // token-conv form: row = token (n, i, j), this column tile = phase (a, b): the output row is pixel (k i + a, k j + b) and the tile's
      // 256 columns are the channels (the ConvTranspose's depth-to-space, dense_head.py:57-66, composed with the 3x3 conv behind it)
      auto out_row = [&](int row) -> size_t {
        if (p.tc_k <= 0) return (size_t)row;
        const int hw = p.cv_h * p.cv_w, n = row / hw, rem = row - n * hw, ti = rem / p.cv_w, tj = rem - ti * p.cv_w;
        const int ph = colb / 256, a = ph / p.tc_k, b = ph - a * p.tc_k;
        return ((size_t)(n * p.cv_h + ti) * p.tc_k + a) * (size_t)(p.cv_w * p.tc_k) + (size_t)(tj * p.tc_k + b);
      };
      const int cshift = p.tc_k > 0 ? (colb / 256) * 256 : 0;
      float4 rs[2][SN], r2[2][SN];
      auto load_res = [&](int i, float4 (&a)[SN], float4 (&b)[SN]) {
        const int row = rowb + i * 16 + l15;
#pragma unroll
        for (int j = 0; j < SN; ++j) {
          const int col = colb + j * 16 + 4 * lq;
          const bool ok = row < mlim && col < p.N;
          a[j] = (p.cv_resid && ok) ? *(const float4*)(p.cv_resid + (size_t)row * p.ldc + col) : make_float4(0, 0, 0, 0);
          b[j] = (p.cv_resid2 && ok) ? *(const float4*)(p.cv_resid2 + (size_t)row * p.ldc + col) : make_float4(0, 0, 0, 0);
        }
      };
      load_res(0, rs[0], r2[0]);
#pragma unroll
      for (int i = 0; i < SM; ++i) {
        if (rowb + i * 16 >= mlim) break;
        if (i + 1 < SM) load_res(i + 1, rs[(i + 1) & 1], r2[(i + 1) & 1]);
        const int row = rowb + i * 16 + l15;
        float4 y[SN];
#pragma unroll
        for (int j = 0; j < SN; ++j) {
          float4 a = rs[i & 1][j];
          if (p.cv_resid_relu) a = make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f));
          const float4 b = r2[i & 1][j], bs = bs4[j];
          y[j] = make_float4(acc[i][j][0] + bs.x + a.x + b.x, acc[i][j][1] + bs.y + a.y + b.y, acc[i][j][2] + bs.z + a.z + b.z, acc[i][j][3] + bs.w + a.w + b.w);
          if (p.relu) y[j] = make_float4(fmaxf(y[j].x, 0.f), fmaxf(y[j].y, 0.f), fmaxf(y[j].z, 0.f), fmaxf(y[j].w, 0.f));
        }
        if (p.out16) {
          if constexpr (SN % 2 == 0) {
#pragma unroll
            for (int jp = 0; jp < SN / 2; ++jp) {
              uint2 u[2];
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const float4 x = y[2 * jp + h];
                u[h].x = (uint32_t)f2t<T>(x.x) | ((uint32_t)f2t<T>(x.y) << 16);
                u[h].y = (uint32_t)f2t<T>(x.z) | ((uint32_t)f2t<T>(x.w) << 16);
              }
              swap16(u[0].x, u[1].x);
              swap16(u[0].y, u[1].y);
              const int col = colb + (2 * jp + (lq & 1)) * 16 + 4 * (lq & 2);
              if (row < mlim && col < p.N) *(uint4*)((u16*)p.C + out_row(row) * p.ldc + col - cshift) = make_uint4(u[0].x, u[0].y, u[1].x, u[1].y);
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < SN; ++j) {
            const int col = colb + j * 16 + 4 * lq;
            if (row < mlim && col < p.N) *(float4*)((float*)p.C + out_row(row) * p.ldc + col - cshift) = y[j];
          }
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < SM; ++i) {
      if (rowb + i * 16 >= mlim) break;  // wave-uniform: the rest of the wave's rows are past the band / matrix end
      const int row = rowb + i * 16 + l15;
      if (row >= mlim) continue;
#pragma unroll
      for (int j = 0; j < SN; ++j) {
        const int col = colb + j * 16 + 4 * lq;
        if (col >= p.N) continue;
        const float4 bs = bs4[j];
        float4 x = make_float4(acc[i][j][0] + bs.x, acc[i][j][1] + bs.y, acc[i][j][2] + bs.z, acc[i][j][3] + bs.w);
        const size_t o = (size_t)row * p.ldc + col;
        if constexpr (EPI == WM_EPI_F32) {
          *(float4*)((float*)p.C + o) = x;
        } else if constexpr (EPI == WM_EPI_T16 || EPI == WM_EPI_GELU_T16) {
          if constexpr (EPI == WM_EPI_GELU_T16) x = gelu_erf4(x);
          uint2 u;
          u.x = (uint32_t)f2t<T>(x.x) | ((uint32_t)f2t<T>(x.y) << 16);
          u.y = (uint32_t)f2t<T>(x.z) | ((uint32_t)f2t<T>(x.w) << 16);
          *(uint2*)((u16*)p.C + o) = u;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Same tile / staging / pipeline, but the inner product runs on v_mfma_f32_16x16x32 (guides "DVFS give-back"
// item 7: at equal cycles per FLOP the chip holds a higher clock on this shape).  Backbone epilogues only.
// Operands swapped as above: D[n = 4(lane>>4) + r][m = lane & 15] -> a lane owns row m and 4 consecutive columns.
template <int T, int EPI, int WM, int WN, int TM, int TN, int NSTAGE>
__global__ __launch_bounds__(WM * WN * 64) void gemm_nt16_kernel(const WmGemmArgs p) {
  constexpr int NW = WM * WN, BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int SM = TM * 2, SN = TN * 2;  // 16-wide sub-tiles per wave
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int PA = BM / 8, PB = BN / 8, PPW = (PA + PB) / NW;
  static_assert((PA + PB) % NW == 0, "pieces must divide over waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int lid = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (lid / ntn) * BM, n0 = (lid % ntn) * BN;
  const u16* A = (const u16*)p.A;
  const u16* W = (const u16*)p.W;
  auto stage = [&](int kt, int s) {
    char* base = smem + s * STAGE;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pc = wave * PPW + i;
      if (pc < PA) stage_piece(A, p.lda, m0, p.M, kt * BK, base, pc, lane);
      else stage_piece(W, p.ldw, n0, p.N, kt * BK, base + A_BYTES, pc - PA, lane);
    }
  };
  f32x4 acc[SM][SN];
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int j = 0; j < SN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = p.K / BK;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) stage(s, s);
  const int l15 = lane & 15, lq = lane >> 4;
  for (int t = 0; t < nk; ++t) {
    const int ahead = min(nk - 1 - t, NSTAGE - 2);
    if (ahead >= 2) wait_vmcnt<2 * PPW>();
    else if (ahead == 1) wait_vmcnt<PPW>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (t + NSTAGE - 1 < nk) stage(t + NSTAGE - 1, (t + NSTAGE - 1) % NSTAGE);
    const char* tA = smem + (t % NSTAGE) * STAGE;
    const char* tB = tA + A_BYTES;
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = 4 * ks + lq;
      s16x8 a[SM], b[SN];
#pragma unroll
      for (int i = 0; i < SM; ++i) a[i] = lds_frag(tA, wm * TM * 32 + i * 16 + l15, ch);
#pragma unroll
      for (int j = 0; j < SN; ++j) b[j] = lds_frag(tB, wn * TN * 32 + j * 16 + l15, ch);
#pragma unroll
      for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int j = 0; j < SN; ++j) acc[i][j] = mfma16<T>(b[j], a[i], acc[i][j]);
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
  }
  epilogue16<T, EPI, SM, SN>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, lane, p.M);
}

template <int T, int EPI, int WM, int WN, int TM, int TN, int NSTAGE>
hipError_t launch16_cfg(const WmGemmArgs& a, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr size_t shm = (size_t)NSTAGE * (BM + BN) * BK * 2;
  const int ntn = (a.N + BN - 1) / BN, ntm = (a.M + BM - 1) / BM;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_nt16_kernel<T, EPI, WM, WN, TM, TN, NSTAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    attr = true;
  }
  hipLaunchKernelGGL((gemm_nt16_kernel<T, EPI, WM, WN, TM, TN, NSTAGE>), dim3(ntm * ntn), dim3(WM * WN * 64), shm, s, a);
  return hipGetLastError();
}

template <int T, int EPI>
hipError_t launch16_E(const WmGemmArgs& a, int cfg, hipStream_t s) {
  return cfg == 5 ? launch16_cfg<T, EPI, 2, 4, 3, 2, 2>(a, s) : launch16_cfg<T, EPI, 2, 4, 4, 2, 2>(a, s);
}

// ---------------------------------------------------------------------------------------------------------
// Ping-pong kernel: (2 QI 32) x 256 x 64 tile, 8 waves as 2 (M) x 4 (N), wave tile (QI 32) x 64 on 16x16x32 MFMAs.
//
// The lock-step kernels above lose ~half their time to the K-tile rendezvous: all 8 waves wait for the DMA, then all
// read LDS (the matrix pipe idles), then the two waves of a SIMD want the matrix pipe at the same moment.  Here the
// two wave groups (wr = 0 / 1; waves w and w + 4 share a SIMD) run ONE BARRIER APART: a K-tile is four phases (one
// 64 x 32 quadrant of the wave tile x K = 64 each: 16 MFMAs), every phase is {load stage: ds_read the quadrant's
// fragments, issue a share of the next K-tile's LDS-DMA | barrier | MFMA stage | barrier}, and group 1 executes one
// extra barrier before the loop, so on every SIMD one wave is in its MFMA stage (at s_setprio 1) while its partner
// is in its load stage (guides "The 256^2 8-phase template", MI355X_MICROARCH "Two waves per SIMD").
//
// Ordering (all waits are explicit):
//  RAW  the next K-tile's DMA is issued in phases 0-2 and drained by each wave (vmcnt(0)) in its phase-3 load stage,
//       i.e. before a barrier that every reader passes before its first read of that tile;
//  WAR  a wave retires its own ds_reads (lgkmcnt(0)) BEFORE the barrier that ends its load stage, so when any wave
//       starts overwriting the other buffer (phase 0 of the next K-tile, at least one barrier later) nobody is
//       still reading it.
template <int T, int EPI, int QI, int PRIO, int DBG = 0>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const WmGemmArgs p) {
  constexpr int SM = 2 * QI, SN = 4;
  constexpr int WROWS = SM * 16, BM = 2 * WROWS, BN = 256;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int PA = BM / 8, PB = BN / 8, PPW = (PA + PB) / 8;  // 1-KiB DMA pieces: 8 or 7 per wave per K-tile
  static_assert((PA + PB) % 8 == 0, "pieces must divide over the 8 waves");
  constexpr int D0 = 3, D1 = PPW >= 8 ? 6 : 5;                   // DMA pieces issued in phases 0 / 1 / 2: [0,D0) [D0,D1) [D1,PPW)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int lid = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (lid / ntn) * BM, n0 = (lid % ntn) * BN;
  const int l15 = lane & 15, lq = lane >> 4;

  // per-lane global source of each DMA piece of this wave (advanced by 64 elements per K-tile) and its LDS offset
  const u16* gp[PPW];
  int loff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pc = wave * PPW + i;  // wave-uniform
    const bool isA = pc < PA;
    const int pl = isA ? pc : pc - PA;
    const int r = pl * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);  // LDS[r][chunk] = G[r][chunk ^ swz(r)]
    int gr = (isA ? m0 : n0) + r;
    const int lim = (isA ? p.M : p.N) - 1;
    gr = gr < lim ? gr : lim;  // clamped rows are masked in the epilogue
    gp[i] = (isA ? (const u16*)p.A + (size_t)gr * p.lda : (const u16*)p.W + (size_t)gr * p.ldw) + c * 8;
    loff[i] = (isA ? 0 : A_BYTES) + pl * 1024;
  }
  auto dma = [&](int buf, int i0, int i1) {
#pragma unroll
    for (int i = i0; i < i1; ++i) {
      __builtin_amdgcn_global_load_lds((glb_vp)gp[i], (lds_vp)(smem + buf * STAGE + loff[i]), 16, 0, 0);
      gp[i] += 64;
    }
  };

  // fragment read offsets: row (base + l15), 16-B chunk 4 kh + lq, swizzle ((row >> 1) & 7) = ((l15 >> 1) & 7)
  const int sw = (l15 >> 1) & 7;
  const int foff0 = l15 * 128 + ((lq ^ sw) << 4), foff1 = l15 * 128 + (((4 + lq) ^ sw) << 4);
  const int a_base = wr * WROWS * 128, b_base = A_BYTES + wc * 64 * 128;

  f32x4 acc[SM][SN];
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int j = 0; j < SN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  s16x8 a[QI][2], b[2][2];

  const int nk = p.K / 64;
  dma(0, 0, PPW);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0

  for (int t = 0; t < nk; ++t) {
    const char* tile = smem + (t & 1) * STAGE;
    const bool more = t + 1 < nk;
    const int nbuf = (t + 1) & 1;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      const int qm = ph >> 1, qn = (ph == 1 || ph == 2) ? 1 : 0;
      // ---- load stage
      if (ph != 2 && (DBG != 2 || t == 0)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          b[j][0] = *(const s16x8*)(tile + b_base + (qn * 2 + j) * 2048 + foff0);
          b[j][1] = *(const s16x8*)(tile + b_base + (qn * 2 + j) * 2048 + foff1);
        }
      }
      if ((ph == 0 || ph == 2) && (DBG != 2 || t == 0)) {
#pragma unroll
        for (int i = 0; i < QI; ++i) {
          a[i][0] = *(const s16x8*)(tile + a_base + (qm * QI + i) * 2048 + foff0);
          a[i][1] = *(const s16x8*)(tile + a_base + (qm * QI + i) * 2048 + foff1);
        }
      }
      if (more && DBG != 1) {
        if (ph == 0) dma(nbuf, 0, D0);
        if (ph == 1) dma(nbuf, D0, D1);
        if (ph == 2) dma(nbuf, D1, PPW);
      }
      if (ph == 3) wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (DBG != 3) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- MFMA stage
      if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int i = 0; i < QI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[qm * QI + i][qn * 2 + j] = mfma16<T>(b[j][kh], a[i][kh], acc[qm * QI + i][qn * 2 + j]);
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (DBG != 3) __builtin_amdgcn_s_barrier();
    }
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();  // balance group 1's extra barrier
  epilogue16<T, EPI, SM, SN>(p, acc, m0 + wr * WROWS, n0 + wc * 64, lane, p.M);
}

// ---------------------------------------------------------------------------------------------------------
// WM_EPI_RESID + the following LayerNorm (see WmGemmArgs::ln_out).  Numerics: the row statistics are combined from partials
// (64 columns per wave -> 256 per tile -> 1024 per row) with the equal-count form of Chan's update, mean = avg(mean_k),
// M2 = sum(M2_k) + n_k sum((mean_k - mean)^2): as accurate as the two-pass form of layernorm_kernel, not bit-identical to it.
// ln_norm() is the one expression both this epilogue and the fallback kernel apply, with explicit roundings and one fma.
typedef __attribute__((address_space(1))) unsigned long long wm_gu64;
typedef __attribute__((address_space(1))) int wm_gi32;
#define WM_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ float ln_norm(float v, float mean, float rstd, float w, float b) {
  return __builtin_fmaf(__fmul_rn(__fsub_rn(v, mean), rstd), w, b);
}
// (mean, M2) of four equal-count partials; cnt = elements per partial
__device__ __forceinline__ void ln_combine4(const float (&m)[4], const float (&q)[4], float cnt, float& mean, float& M2) {
  mean = __fmul_rn(__fadd_rn(__fadd_rn(m[0], m[1]), __fadd_rn(m[2], m[3])), 0.25f);
  const float d0 = __fsub_rn(m[0], mean), d1 = __fsub_rn(m[1], mean), d2 = __fsub_rn(m[2], mean), d3 = __fsub_rn(m[3], mean);
  const float dd = __fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fadd_rn(__fmul_rn(d2, d2), __fmul_rn(d3, d3)));
  M2 = __fadd_rn(__fadd_rn(__fadd_rn(q[0], q[1]), __fadd_rn(q[2], q[3])), __fmul_rn(cnt, dd));
}
constexpr int WM_LN_SPIN_LIMIT = 4000;   // polls of ~64 cycles each: ~120 us, far beyond a band's skew when its four blocks are resident

// acc holds this wave's (SM x 16 rows) x 64 columns of acc; on return X (and the tap half) are updated and, unless the rendezvous
// timed out, ln_out holds the normalised rows of this block's 256 columns.
template <int T, int SM>
__device__ __forceinline__ void epilogue_resid_ln(const WmGemmArgs& p, f32x4 (&acc)[SM][4], char* smem, int tid, int lane, int wr, int wc,
                                                  int band, int nt, int rowb, int colb, int rl0, int mlim) {
  const int l15 = lane & 15, lq = lane >> 4;
  float4 bs4[4], gm4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = colb + j * 16 + 4 * lq;
    bs4[j] = p.bias ? *(const float4*)(p.bias + col) : make_float4(0, 0, 0, 0);
    gm4[j] = *(const float4*)(p.gamma + col);
  }
  // ---- A: new values gamma (acc + bias) + X into acc.  The X stores are NOT issued yet: the block's signal has to wait for every store
  // in front of it, so the 8-byte statistics go out first and the bulk stores (phase X below) run while the band's other blocks arrive
  float4 old[2][4];
  auto load_old = [&](int i, float4 (&o)[4]) {
    const int row = rowb + i * 16 + l15;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = row < mlim ? *(const float4*)((const float*)p.C + (size_t)row * p.ldc + colb + j * 16 + 4 * lq) : make_float4(0, 0, 0, 0);
  };
  load_old(0, old[0]);
#pragma unroll
  for (int i = 0; i < SM; ++i) {
    if (rowb + i * 16 >= mlim) break;
    if (i + 1 < SM) load_old(i + 1, old[(i + 1) & 1]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 o = old[i & 1][j], gm = gm4[j], bs = bs4[j];
      acc[i][j] = f32x4{o.x + gm.x * (acc[i][j][0] + bs.x), o.y + gm.y * (acc[i][j][1] + bs.y), o.z + gm.z * (acc[i][j][2] + bs.z), o.w + gm.w * (acc[i][j][3] + bs.w)};
    }
  }
  // ---- B: per row (mean, M2) over this wave's 64 columns -> LDS [tile row][wc]
  float2* lst = (float2*)smem;     // [256][4] (mean, M2): the K-tile ring is dead
  int* lflag = (int*)(smem + 256 * 4 * 8);
  __syncthreads();                 // every wave is out of the K loop: the ring may be overwritten
#pragma unroll
  for (int i = 0; i < SM; ++i) {
    if (rowb + i * 16 >= mlim) break;   // wave-uniform
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s1 += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
    s1 += __shfl_xor(s1, 16);
    const float mw = __fmul_rn(xhalf_sum(s1), 1.0f / 64.0f);
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = acc[i][j][e] - mw; s2 = __builtin_fmaf(d, d, s2); }
    s2 += __shfl_xor(s2, 16);
    s2 = xhalf_sum(s2);
    if (lq == 0) lst[(rl0 + i * 16 + l15) * 4 + wc] = make_float2(mw, s2);
  }
  __syncthreads();
  // ---- C: the tile's (mean, M2) per row; the wc == 0 waves publish them (8-byte sc1 stores), then one arrival per block
#pragma unroll
  for (int i = 0; i < SM; ++i) {
    if (rowb + i * 16 >= mlim) break;
    const float4 a0 = *(const float4*)(lst + (rl0 + i * 16 + l15) * 4), a1 = *(const float4*)(lst + (rl0 + i * 16 + l15) * 4 + 2);
    const float m[4] = {a0.x, a0.z, a1.x, a1.z}, q[4] = {a0.y, a0.w, a1.y, a1.w};
    float tmean, tm2;
    ln_combine4(m, q, 64.0f, tmean, tm2);
    const int row = rowb + i * 16 + l15;
    if (wc == 0 && lq == 0 && row < mlim) {
      const unsigned long long bits = ((unsigned long long)__builtin_bit_cast(unsigned, tm2) << 32) | __builtin_bit_cast(unsigned, tmean);
      __hip_atomic_store((wm_gu64*)(p.ln_stats + ((size_t)row * 4 + nt) * 2), bits, WM_RLX_AGENT);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains before the block signals
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add((wm_gi32*)(p.ln_sync + band * 2), 1, WM_RLX_AGENT);
  // ---- X: the residual stream (and the tap half), while the other three blocks of the band arrive
#pragma unroll
  for (int i = 0; i < SM; ++i) {
    if (rowb + i * 16 >= mlim) break;
    const int row = rowb + i * 16 + l15;
    if (row < mlim) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 nv = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        *(float4*)((float*)p.C + (size_t)row * p.ldc + colb + j * 16 + 4 * lq) = nv;
        if (p.C2) *(float4*)(p.C2 + (size_t)row * p.ldc2 + colb + j * 16 + 4 * lq) = nv;
      }
    }
  }
  int ok = 0;
  if (tid == 0) {
    int spins = 0;
    while (true) {
      if (__hip_atomic_load((wm_gi32*)(p.ln_sync + band * 2), WM_RLX_AGENT) >= 4) { ok = 1; break; }
      if (++spins > WM_LN_SPIN_LIMIT) break;
      __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) __hip_atomic_store((wm_gi32*)(p.ln_fallback + band), 1, WM_RLX_AGENT);
    *lflag = ok;
  }
  __syncthreads();
  ok = *lflag;
  // ---- D: the row's statistics from the four tiles' partials (sc1 loads of sc1-stored, drained bytes), normalise, 16-bit stores
  if (ok) {
    float4 w4[4], b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w4[j] = *(const float4*)(p.ln_w + colb + j * 16 + 4 * lq);
      b4[j] = *(const float4*)(p.ln_b + colb + j * 16 + 4 * lq);
    }
#pragma unroll
    for (int i = 0; i < SM; ++i) {
      if (rowb + i * 16 >= mlim) break;
      const int row = rowb + i * 16 + l15;
      const int rr = row < mlim ? row : mlim - 1;
      float m[4], q[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned long long bits = __hip_atomic_load((wm_gu64*)(p.ln_stats + ((size_t)rr * 4 + k) * 2), WM_RLX_AGENT);
        m[k] = __builtin_bit_cast(float, (unsigned)bits); q[k] = __builtin_bit_cast(float, (unsigned)(bits >> 32));
      }
      float mean, M2;
      ln_combine4(m, q, 256.0f, mean, M2);
      const float rstd = 1.0f / sqrtf(__fadd_rn(__fmul_rn(M2, 1.0f / 1024.0f), p.ln_eps));
      uint2 u[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float y0 = ln_norm(acc[i][j][0], mean, rstd, w4[j].x, b4[j].x), y1 = ln_norm(acc[i][j][1], mean, rstd, w4[j].y, b4[j].y);
        const float y2 = ln_norm(acc[i][j][2], mean, rstd, w4[j].z, b4[j].z), y3 = ln_norm(acc[i][j][3], mean, rstd, w4[j].w, b4[j].w);
        u[j].x = (uint32_t)f2t<T>(y0) | ((uint32_t)f2t<T>(y1) << 16);
        u[j].y = (uint32_t)f2t<T>(y2) | ((uint32_t)f2t<T>(y3) << 16);
      }
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {   // 16-B stores of 64-B row segments (see the T16 epilogue)
        swap16(u[2 * jp].x, u[2 * jp + 1].x);
        swap16(u[2 * jp].y, u[2 * jp + 1].y);
        const int col = colb + (2 * jp + (lq & 1)) * 16 + 4 * (lq & 2);
        if (row < mlim) *(uint4*)((u16*)p.ln_out + (size_t)row * p.ln_ld + col) = make_uint4(u[2 * jp].x, u[2 * jp].y, u[2 * jp + 1].x, u[2 * jp + 1].y);
      }
    }
  }
  // ---- E: departure; the band's last block to leave re-arms the counters for the next launch
  __syncthreads();
  if (tid == 0) {
    const int d = __hip_atomic_fetch_add((wm_gi32*)(p.ln_sync + band * 2 + 1), 1, WM_RLX_AGENT);
    if (d == 3) {
      __hip_atomic_store((wm_gi32*)(p.ln_sync + band * 2), 0, WM_RLX_AGENT);
      __hip_atomic_store((wm_gi32*)(p.ln_sync + band * 2 + 1), 0, WM_RLX_AGENT);
    }
  }
}

#ifdef WM_GEMM_STAMPS
__device__ int wm_ln_fallback_count;
#endif
// Bands whose rendezvous timed out (ln_fallback set): LayerNorm of their rows from X and the published partials, the epilogue's arithmetic.
template <int T>
__global__ __launch_bounds__(256) void gemm_ln_fallback_kernel(const WmGemmArgs p, int bands, int full_units) {
  const int band = blockIdx.x;
  if (__hip_atomic_load((wm_gi32*)(p.ln_fallback + band), WM_RLX_AGENT) == 0) return;   // block-uniform
  int u0, S;
  if (p.sched_bands > 0) {
    u0 = (int)((long long)band * p.sched_units / bands);
    S = (int)((long long)(band + 1) * p.sched_units / bands) - u0;
  } else {
    u0 = band * full_units; S = full_units;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#ifdef WM_GEMM_STAMPS
  if (threadIdx.x == 0) atomicAdd(&wm_ln_fallback_count, 1);   // diagnostic build: bands recomputed since the last read
#endif
  const int r1 = (u0 + S) * 16 < p.M ? (u0 + S) * 16 : p.M;
  for (int row = u0 * 16 + wave; row < r1; row += 4) {
    float m[4], q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned long long bits = __hip_atomic_load((wm_gu64*)(p.ln_stats + ((size_t)row * 4 + k) * 2), WM_RLX_AGENT);
      m[k] = __builtin_bit_cast(float, (unsigned)bits); q[k] = __builtin_bit_cast(float, (unsigned)(bits >> 32));
    }
    float mean, M2;
    ln_combine4(m, q, 256.0f, mean, M2);
    const float rstd = 1.0f / sqrtf(__fadd_rn(__fmul_rn(M2, 1.0f / 1024.0f), p.ln_eps));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = (i * 64 + lane) * 4;
      const float4 v = *(const float4*)((const float*)p.C + (size_t)row * p.ldc + c);
      const float4 w = *(const float4*)(p.ln_w + c), b = *(const float4*)(p.ln_b + c);
      uint2 u;
      u.x = (uint32_t)f2t<T>(ln_norm(v.x, mean, rstd, w.x, b.x)) | ((uint32_t)f2t<T>(ln_norm(v.y, mean, rstd, w.y, b.y)) << 16);
      u.y = (uint32_t)f2t<T>(ln_norm(v.z, mean, rstd, w.z, b.z)) | ((uint32_t)f2t<T>(ln_norm(v.w, mean, rstd, w.w, b.w)) << 16);
      *(uint2*)((u16*)p.ln_out + (size_t)row * p.ln_ld + c) = u;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store((wm_gi32*)(p.ln_fallback + band), 0, WM_RLX_AGENT);
}

#ifdef WM_GEMM_STAMPS
// diagnostic build only (make stamps): wave 0 of every block records, on the 100 MHz s_memrealtime counter, its entry, the start and
// the end of its K loop and its exit (after its own stores have drained), the shader clock count of the loop and the CU it ran on;
// read back by wm_debug_gemm_stamps (tools/gemm_timeline.py).  No output depends on the stamps.
__device__ unsigned long long wm_gemm_stamp_buf[8 * 8192];
#endif

// Ping-pong v2 (256 x 256 only): same two-group structure, but the LDS-DMA runs TWO K-tiles ahead inside the same two
// 64-KiB buffers.  Quadrant order (0,0) (0,1) (1,1) (1,0) with B(qn=0) kept in registers for the whole K-tile, so the
// four regions of a buffer die one after the other — {A rows of qm=0, B cols of qn=0} after phase 0, B(qn=1) after
// phase 1, A(qm=1) after phase 2 — and each is refilled with K-tile t+2 in the very next phase (every wave issues
// 2 pieces per region).  In-order counted waits: with R = 4 / 2 / 2 pieces per wave for {A0,B0} / B1 / A1,
//   end of phase 3: {A0,B0}(t+1) landed  <=> at most  4 + 8[t+2 < nk]                  younger pieces outstanding
//   end of phase 0:  B1(t) landed        <=> at most  2 + 8[t+1 < nk]
//   end of phase 1:  A1(t) landed        <=> at most  8[t+1 < nk] + 4[t+2 < nk]
// RAW/WAR argument as for v1 (wait, then a barrier every reader passes; own reads retired before the stage's barrier).
// QI = 3 (192-row tile: wave tile 96 x 64, quadrants of 3 sub-tiles): the A regions have 12 pieces, so waves 0-3 carry
// two per region and waves 4-7 one; the counted waits then differ per wave half (R_A0B0 = nA + 2, R_B1 = 2, R_A1 = nA,
// nA = 2 | 1  ->  12 / 10 / 12 / 8 / 4 / 2  |  9 / 7 / 9 / 6 / 3 / 1), selected by a wave-uniform branch.
//
// VER = 3 (round 4) — HALF THE BARRIERS.  In-kernel stamps (profiles/r04_gemm_timeline.md) put a K-tile at 2 260 - 2 720 cycles
// against 1 536 / 2 048 of MFMA issue, the same with every operand row an L2 hit, 8 % less with the LDS-DMA removed, and
// barely less with MFMAs removed (a band cut from 12 to 10.75 units: -3 %): the eight s_barrier hand-offs of a K-tile
// (~ 90 cycles each in which the matrix pipe has nothing queued) are what the loop is made of.  v3 keeps the schedule of v2
// but lets group 0 ("X") put its barrier BEFORE its MFMA stage only and group 1 ("Y") BEFORE its load stage only:
//      X:  L(0) | B  M(k)  L(k+1) | B  M(k+1)  L(k+2) | ...
//      Y:       | B  L(k)  M(k)   | B  L(k+1)  M(k+1) | ...
// so inside an interval X's MFMAs run beside Y's loads, Y's MFMAs queue right behind them (no barrier in between) beside
// X's next loads, and only one hand-off per phase is left.  Ordering, with I_k the interval after barrier k = 4 t + p:
//  WAR  a region is refilled two phases after the phase whose load stage read it, instead of one: {A0,B0}(t+2) in L(t,2),
//       B1(t+2) in L(t,3), A1(t+2) in L(t+1,0).  Its readers were X's L(t,p) in I_(k-1) and Y's L(t,p) in I_k; the first
//       refill is X's in I_(k+1), behind barrier k+1, and Y retired its reads (lgkmcnt(0)) before its M(t,p).
//  RAW  a region of K-tile t+1 is first read by X in the interval before Y reads it; every wave waits for its own pieces of
//       it before the barrier that opens that interval: X at the end of the load stage in front of that barrier, Y after
//       the MFMA stage in front of it.  Pieces are issued in the order A0B0(k) B1(k) A1(k), k = 0, 1, ... by every wave
//       (c0 = nA + 2, c1 = 2, c2 = nA pieces, c = 2 nA + 4; nA = 2, or 1 for waves 4-7 of the 192-row tile), so with
//       n1 = [t+1 < nk], n2 = [t+2 < nk] the counted waits are
//         X  end of L(t,0): B1(t)       c2 + n1 c              end of L(t,1): A1(t)   n1 c
//            end of L(t,3): A0B0(t+1)   c1 + c2 + n2 (c0 + c1)
//         Y  end of M(t,0): A1(t)       n1 c                   end of M(t,2): A0B0(t+1)   c1 + c2 + n2 c0
//            end of M(t,3): B1(t+1)     c2 + n2 (c0 + c1)
//       and everybody waits for A0B0(0) and B1(0) (c2 + [nk > 1] c younger) before one common barrier in front of the loop.
template <int T, int EPI, int PRIO, int DBG = 0, int QI = 4, int VER = 2>
__global__ __launch_bounds__(512) void gemm_pp2_kernel(const WmGemmArgs p) {
  constexpr int SM = 2 * QI, SN = 4, WROWS = SM * 16, BM = 2 * WROWS, BN = 256;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int APC = QI * 2;  // 8-row pieces per (wave row half, quadrant): 8 or 6
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef WM_GEMM_STAMPS
  const unsigned long long st_entry = __builtin_amdgcn_s_memrealtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int ntn = (p.N + BN - 1) / BN, ntm = p.sched_bands > 0 ? p.sched_bands : (p.M + BM - 1) / BM;
  const int lid = xcd_remap(blockIdx.x, ntm * ntn);
  int band, nt;
  tile_of(lid, ntm, ntn, p.group_bands, band, nt);
  // Row band of this block, in 16-row units.  With a schedule (wm_launch_gemm) the M rows are cut into `ntm` bands of
  // floor / ceil(units / ntm) units, spread evenly (band b starts at floor(b units / ntm)), so that ntm x ntn fills whole
  // rounds of the CUs instead of leaving the last round a third empty (M = 11008: 696 tiles of 192 rows = 2.72 rounds ->
  // 768 tiles of 160 / 176 rows = 3.0).  A band of S units is S0 = ceil(S / 2) units for wave group 0 and S1 = S - S0 for
  // group 1; the LDS image keeps the full-tile layout (group 1's rows start at WROWS), only the SOURCE rows of group 1 move
  // up to follow group 0's, the MFMAs of the units a group does not have are skipped, and the epilogue stops at the band's end.
  int u0, S;
  if (p.sched_bands > 0) {
    u0 = (int)((long long)band * p.sched_units / ntm);
    S = (int)((long long)(band + 1) * p.sched_units / ntm) - u0;
  } else {
    u0 = band * 2 * SM;
    S = 2 * SM;
  }
  const int S0 = (S + 1) >> 1, Sw = wr ? S - S0 : S0;        // units of this wave's group (block- / wave-uniform)
  const int nB = Sw - QI;                                    // units in the group's second quadrant row: QI, QI - 1 or QI - 2
  const int m0 = u0 * 16, n0 = nt * BN;
  const int l15 = lane & 15, lq = lane >> 4;
  const bool two = QI == 4 || wave < 4;  // this wave carries two A pieces per region (wave-uniform)

  // this wave's DMA pieces per K-tile, in issue order: A0 A0 B0 B0 | B1 B1 | A1 A1 (the second A piece only if `two`)
  const u16* gp[8];
  int loff[8];
  // WM_EPI_CONV: the A rows are pixels; per A piece the lane's pixel, its border flags and its 16-B chunk, and (wave-uniform) the
  // tap / channel chunk of the K-tile the piece is issued for next (K-tile kt = tap * chunks + chunk: the weight's own K order)
  constexpr bool CONV = EPI == WM_EPI_CONV;
  int cv_pix[4], cv_fl[4], cv_c8[4], cv_tap[4] = {0, 0, 0, 0}, cv_cc[4] = {0, 0, 0, 0};
  const int cv_nch = CONV ? p.cv_cin >> 6 : 1;
  // token-conv form (p.tc_k > 0): column tile nt is an output phase; its word = neighbour count << 16 | 4 bits (di + 1, dj + 1) per neighbour
  const unsigned tc_word = CONV && p.tc_k > 0 ? p.tc_list[nt & 15] : 0u;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const bool isA = i < 2 || i >= 6;
    const int half = (i >= 4) ? 1 : 0;              // A1 / B1
    int pl;
    if (isA) {
      const int cl = QI == 4 ? 2 * wave + (i & 1) : (wave < 4 ? 2 * wave + (i & 1) : 8 + (wave - 4));  // index in the 2*APC-piece region
      pl = (cl < APC ? cl : cl - APC + 2 * APC) + APC * half;   // rows wr' * WROWS + 8 * APC * qm + ...
    } else {
      const int cl = 2 * wave + (i & 1);
      pl = (cl >> 2) * 8 + (cl & 3) + 4 * half;                 // cols wc' * 64 + 32 qn + ...
    }
    const int r = pl * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int gr = (DBG == 4 ? 0 : (isA ? m0 : n0)) + (isA && r >= WROWS ? r - WROWS + 16 * S0 : r);
    const int lim = (isA ? p.M : p.N) - 1;
    gr = gr < lim ? gr : lim;
    gp[i] = (isA ? (const u16*)p.A + (size_t)gr * p.lda : (const u16*)p.W + (size_t)gr * p.ldw) + c * 8;
    loff[i] = (isA ? 0 : A_BYTES) + pl * 1024;
    if constexpr (CONV) {
      if (isA) {   // the row is a pixel (n, y, x) of the NHWC input: remember it and which image borders it touches
        const int ai = i < 2 ? i : i - 4;
        const int hw = p.cv_h * p.cv_w, n = gr / hw, rem = gr - n * hw, y = rem / p.cv_w, x = rem - y * p.cv_w;
        cv_pix[ai] = gr;
        cv_fl[ai] = (y == 0 ? 1 : 0) | (y == p.cv_h - 1 ? 2 : 0) | (x == 0 ? 4 : 0) | (x == p.cv_w - 1 ? 8 : 0);
        cv_c8[ai] = c * 8;
      }
    }
  }
  auto dma = [&](int buf, int i0, int i1, bool in_loop = false) {
    if ((DBG == 6 || (DBG >= 16 && (DBG & 1))) && in_loop) return;  // timing experiment: no LDS-DMA inside the K loop (the counted waits then never block)
#pragma unroll
    for (int i = i0; i < i1; ++i) {
      if ((i == 1 || i == 7) && !two) continue;  // wave-uniform
      if constexpr (CONV) {
        if (i < 2 || i >= 6) {   // an A piece: the pixel shifted by the K-tile's tap, or the zero page outside the image
          const int ai = i < 2 ? i : i - 4;
          const int tap = cv_tap[ai], cc = cv_cc[ai];
          int dy, dx;
          if (p.tc_k > 0) { const unsigned nb = tc_word >> (4 * tap); dy = (int)(nb & 3) - 1; dx = (int)((nb >> 2) & 3) - 1; }
          else { const int ty = tap / 3; dy = ty - 1; dx = tap - ty * 3 - 1; }
          const int mask = (dy < 0 ? 1 : 0) | (dy > 0 ? 2 : 0) | (dx < 0 ? 4 : 0) | (dx > 0 ? 8 : 0);   // wave-uniform
          const u16* inside = (const u16*)p.A + ((long long)(cv_pix[ai] + dy * p.cv_w + dx) * p.cv_cin + cc * 64 + cv_c8[ai]);
          const u16* src = (cv_fl[ai] & mask) ? (const u16*)p.cv_zero + cv_c8[ai] : inside;
          __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(smem + buf * STAGE + loff[i]), 16, 0, 0);
          if (++cv_cc[ai] == cv_nch) { cv_cc[ai] = 0; ++cv_tap[ai]; }
          continue;
        }
      }
      __builtin_amdgcn_global_load_lds((glb_vp)gp[i], (lds_vp)(smem + buf * STAGE + loff[i]), 16, 0, 0);
      gp[i] += 64;
    }
  };
  auto wait2 = [&](auto n2c, auto n1c) {  // counted wait: immediates for the two-piece / one-piece wave halves
    if (two) wait_vmcnt<decltype(n2c)::value>(); else wait_vmcnt<decltype(n1c)::value>();
  };
#define WM_W2(a_, b_) wait2(std::integral_constant<int, a_>{}, std::integral_constant<int, b_>{})
  const int sw = (l15 >> 1) & 7;
  const int foff0 = l15 * 128 + ((lq ^ sw) << 4), foff1 = l15 * 128 + (((4 + lq) ^ sw) << 4);
  const int a_base = wr * WROWS * 128, b_base = A_BYTES + wc * 64 * 128;

  f32x4 acc[SM][SN];
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int j = 0; j < SN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  s16x8 a[QI][2], b0[2][2], b1[2][2];

  // (token-conv form of WM_EPI_CONV: this column tile's own neighbour list sets the K length)
  const int nk = CONV && p.tc_k > 0 ? (int)(tc_word >> 16) * cv_nch : p.K / 64;
  dma(0, 0, 8);
  if (nk > 1) dma(1, 0, 8);
  if constexpr (VER == 3) {   // A0B0(0) and B1(0): group X reads B1(0) in the first interval, before group Y has had a wait of its own
    if (nk > 1) WM_W2(10, 7); else WM_W2(2, 1);
  } else {
    if (nk > 1) WM_W2(12, 9); else WM_W2(4, 3);
  }
  __builtin_amdgcn_s_barrier();
  if (VER == 2 && wr == 1) __builtin_amdgcn_s_barrier();

#ifdef WM_GEMM_STAMPS
  const unsigned long long st_loop0 = __builtin_amdgcn_s_memrealtime(), st_cyc0 = __builtin_amdgcn_s_memtime();
#endif
  if constexpr (VER == 3) {
    constexpr int NAY = QI == 4 ? 2 : 1;                      // A pieces per region of a group-1 wave (group 0 always carries 2)
    constexpr int CY = 2 * NAY + 4, C0Y = NAY + 2;
    auto lgk0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    auto bar = [&]() { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); };
    auto mfma_on = [&]() { __builtin_amdgcn_sched_barrier(0); if (PRIO) __builtin_amdgcn_s_setprio(1); };
    auto mfma_off = [&]() { if (PRIO) __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); };
    auto kloop3 = [&](auto nbc) __attribute__((always_inline)) {
      constexpr int NB = decltype(nbc)::value;
      // (Tried and dropped: reading only the k-half 0 fragments in the load stage and the k-half 1 fragments behind the k-half 0
      // MFMAs of the MFMA stage itself — the wait in the middle of the stage cost more than the shorter load stages gave: fc1
      // 101 -> 105 us, profiles/r04_gemm_timeline.md.)
      auto rd_a = [&](int t, int q0, int n) {   // A fragments of quadrant row q0 (first unit index), both k-halves
        const char* tile = smem + (t & 1) * STAGE;
#pragma unroll
        for (int i = 0; i < n; ++i) {
          a[i][0] = *(const s16x8*)(tile + a_base + (q0 + i) * 2048 + foff0);
          a[i][1] = *(const s16x8*)(tile + a_base + (q0 + i) * 2048 + foff1);
        }
      };
      auto rd_b = [&](int t, int j0, s16x8 (&b)[2][2]) {
        const char* tile = smem + (t & 1) * STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          b[j][0] = *(const s16x8*)(tile + b_base + (j0 + j) * 2048 + foff0);
          b[j][1] = *(const s16x8*)(tile + b_base + (j0 + j) * 2048 + foff1);
        }
      };
      auto L0 = [&](int t) {
        rd_b(t, 0, b0); rd_a(t, 0, QI);
        if (t >= 1 && t + 1 < nk) dma((t + 1) & 1, 6, 8, true);   // A1(t+1): its last readers were the L(t-1,2)
      };
      auto L1 = [&](int t) { rd_b(t, 2, b1); };
      auto L2 = [&](int t) {
        rd_a(t, QI, NB);
        if (t + 2 < nk) dma(t & 1, 0, 4, true);                   // {A0,B0}(t+2): last readers the L(t,0)
      };
      auto L3 = [&](int t) {
        if (t + 2 < nk) dma(t & 1, 4, 6, true);                   // B1(t+2): last readers the L(t,1)
      };
      auto mm = [&](int r0, int n, int c0, s16x8 (&b)[2][2]) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
          for (int i = 0; i < n; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[r0 + i][c0 + j] = mfma16<T>(b[j][kh], a[i][kh], acc[r0 + i][c0 + j]);
      };
      auto M0 = [&](int) { mfma_on(); mm(0, QI, 0, b0); mfma_off(); };
      auto M1 = [&](int) { mfma_on(); mm(0, QI, 2, b1); mfma_off(); };
      auto M2 = [&](int) { mfma_on(); mm(QI, NB, 2, b1); mfma_off(); };
      auto M3 = [&](int) { mfma_on(); mm(QI, NB, 0, b0); mfma_off(); };
      if (wr == 0) {
        // ---- group X (nA = 2: c0 4, c1 2, c2 2, c 8): load stage one phase ahead of its barrier
        L0(0);
        if (nk > 1) wait_vmcnt<10>(); else wait_vmcnt<2>();
        for (int t = 0; t < nk; ++t) {
          const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
          lgk0(); bar(); M0(t);
          L1(t);
          if (n1) wait_vmcnt<8>(); else wait_vmcnt<0>();
          lgk0(); bar(); M1(t);
          L2(t);
          lgk0(); bar(); M2(t);
          L3(t);
          if (n1) { if (n2) wait_vmcnt<10>(); else wait_vmcnt<4>(); }
          bar(); M3(t);
          if (n1) {
            L0(t + 1);
            if (n2) wait_vmcnt<10>(); else wait_vmcnt<2>();
          }
        }
      } else {
        // ---- group Y (nA = NAY): barrier, load stage, MFMA stage; its waits sit behind the MFMA stages
        for (int t = 0; t < nk; ++t) {
          const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
          bar(); L0(t); lgk0(); M0(t);
          if (n1) wait_vmcnt<CY>(); else wait_vmcnt<0>();
          bar(); L1(t); lgk0(); M1(t);
          bar(); L2(t); lgk0(); M2(t);
          if (n1) { if (n2) wait_vmcnt<2 + NAY + C0Y>(); else wait_vmcnt<2 + NAY>(); }
          bar(); L3(t); M3(t);
          if (n1) { if (n2) wait_vmcnt<NAY + C0Y + 2>(); else wait_vmcnt<NAY>(); }
        }
      }
    };
    if (nB == QI) kloop3(std::integral_constant<int, QI>{});
    else if (nB == QI - 1) kloop3(std::integral_constant<int, QI - 1>{});
    else kloop3(std::integral_constant<int, QI - 2>{});
  } else {
  // timing experiments (stamps build, DBG = 16 + mask; results are wrong): 1 no LDS-DMA in the loop, 2 no fragment reads after the
  // first K-tile, 4 no barriers in the loop, 8 no s_setprio
  constexpr bool NO_RD = DBG >= 16 && (DBG & 2), NO_BAR = DBG >= 16 && (DBG & 4), NO_PRIO = DBG >= 16 && (DBG & 8);
  auto stage_end = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!NO_BAR) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (PRIO && !NO_PRIO) __builtin_amdgcn_s_setprio(1);
  };
  auto mfma_end = [&]() {
    if (PRIO && !NO_PRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (!NO_BAR) __builtin_amdgcn_s_barrier();
  };

  // WM_EPI_RESID (p.pf_c): the epilogue's read-modify-write of the fp32 residual tile is a burst every CU issues at the same moment
  // (profiles/r04_gemm_ceiling.md: 11.9 us against 2 us for a plain epilogue).  In the last two K-tiles no refill is in flight: each lane
  // then touches one 128-B line of the block's old C tile per instruction (QI instructions cover the 64 QI rows x 8 lines) with a 4-byte
  // LDS-DMA into a scratch corner behind the ring — no destination register to keep alive, counted by vmcnt like the refills, never
  // read — so that the epilogue's loads find the lines on their way or in the L2.  The waits behind it allow PF more operations.
  constexpr bool PFX = EPI == WM_EPI_RESID && DBG == 0;
  constexpr int PF = QI;
  const bool do_pf = PFX && p.pf_c != 0 && nk >= 2;
  auto prefetch_c = [&]() __attribute__((always_inline)) {
    if constexpr (PFX) {
      const int rows = 16 * S, last = (p.M - 1 - m0) < rows - 1 ? (p.M - 1 - m0) : rows - 1;
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int l = tid + 512 * j;
        int row = l >> 3;
        row = row < last ? row : last;
        int col = n0 + (l & 7) * 32;
        col = col < p.N - 1 ? col : p.N - 4;
        const float* src = (const float*)p.C + (size_t)(m0 + row) * p.ldc + col;
        __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(smem + 2 * STAGE + wave * 256), 4, 0, 0);
      }
    }
  };
  // The K loop, instantiated per number NB of 16-row units in this wave group's SECOND quadrant row (a shortened band drops
  // units from the end of the group): the choice is made once, outside the loop (a guard per MFMA group inside it measured
  // +30 % on the loop).  The first quadrant row is always whole (wm_launch_gemm keeps bands >= 4 QI - 4 units).
  auto kloop = [&](auto nbc) __attribute__((always_inline)) {
  constexpr int NB = decltype(nbc)::value;
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    const char* tile = smem + buf * STAGE;
    const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
    // ---- phase 0: quadrant (0,0)
    if (!NO_RD || t == 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      b0[j][0] = *(const s16x8*)(tile + b_base + j * 2048 + foff0);
      b0[j][1] = *(const s16x8*)(tile + b_base + j * 2048 + foff1);
    }
#pragma unroll
    for (int i = 0; i < QI; ++i) {
      a[i][0] = *(const s16x8*)(tile + a_base + i * 2048 + foff0);
      a[i][1] = *(const s16x8*)(tile + a_base + i * 2048 + foff1);
    }
    }
    if (n1) WM_W2(10, 7); else if (do_pf) WM_W2(2 + PF, 1 + PF); else WM_W2(2, 1);
    stage_end();
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < QI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<T>(b0[j][kh], a[i][kh], acc[i][j]);
    mfma_end();
    // ---- phase 1: quadrant (0,1); refill {A0,B0} with K-tile t+2
    if (!NO_RD || t == 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      b1[j][0] = *(const s16x8*)(tile + b_base + (2 + j) * 2048 + foff0);
      b1[j][1] = *(const s16x8*)(tile + b_base + (2 + j) * 2048 + foff1);
    }
    }
    if (n2) { dma(buf, 0, 4, true); WM_W2(12, 9); }
    else if (n1) { if (do_pf) { prefetch_c(); WM_W2(8 + PF, 6 + PF); } else WM_W2(8, 6); }
    else if (do_pf) wait_vmcnt<PF>(); else wait_vmcnt<0>();
    stage_end();
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < QI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][2 + j] = mfma16<T>(b1[j][kh], a[i][kh], acc[i][2 + j]);
    mfma_end();
    // ---- phase 2: quadrant (1,1); refill B1
    if (!NO_RD) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      a[i][0] = *(const s16x8*)(tile + a_base + (QI + i) * 2048 + foff0);
      a[i][1] = *(const s16x8*)(tile + a_base + (QI + i) * 2048 + foff1);
    }
    }
    if (n2) dma(buf, 4, 6, true);
    stage_end();
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[QI + i][2 + j] = mfma16<T>(b1[j][kh], a[i][kh], acc[QI + i][2 + j]);
    mfma_end();
    // ---- phase 3: quadrant (1,0) from registers; refill A1; next K-tile's {A0,B0} must have landed
    if (n2) { dma(buf, 6, 8, true); WM_W2(12, 9); } else if (n1) { if (do_pf) WM_W2(4 + PF, 3 + PF); else WM_W2(4, 3); }
    stage_end();
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[QI + i][j] = mfma16<T>(b0[j][kh], a[i][kh], acc[QI + i][j]);
    mfma_end();
  }
  };
  if (nB == QI) kloop(std::integral_constant<int, QI>{});
  else if (nB == QI - 1) kloop(std::integral_constant<int, QI - 1>{});
  else kloop(std::integral_constant<int, QI - 2>{});
  }
#undef WM_W2
#ifdef WM_GEMM_STAMPS
  const unsigned long long st_loop1 = __builtin_amdgcn_s_memrealtime(), st_cyc1 = __builtin_amdgcn_s_memtime();
#endif
  if (VER == 2 && wr == 0) __builtin_amdgcn_s_barrier();
  if (DBG == 5 && acc[0][0][0] != 1.2345e-30f) return;  // timing experiment: no epilogue
  {
    const int rowb = m0 + (wr ? 16 * S0 : 0), rend = rowb + 16 * Sw;
    if constexpr (EPI == WM_EPI_RESID && DBG == 0 && QI == 3) {   // (192-row tile only: on the 256-row tile the fused epilogue does not fit 256 registers)
      if (p.ln_out) {   // block-uniform: the following LayerNorm rides in this epilogue
        epilogue_resid_ln<T, SM>(p, acc, smem, tid, lane, wr, wc, band, nt, rowb, n0 + wc * 64, wr ? 16 * S0 : 0, rend < p.M ? rend : p.M);
        return;
      }
    }
    epilogue16<T, EPI, SM, SN>(p, acc, rowb, n0 + wc * 64, lane, rend < p.M ? rend : p.M);
  }
#ifdef WM_GEMM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (wave == 0 && blockIdx.x < 8192) {
    const unsigned long long st_exit = __builtin_amdgcn_s_memrealtime();
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_HW_ID, HW_REG_XCC_ID
    if (lane == 0) {
      unsigned long long* o = wm_gemm_stamp_buf + (size_t)blockIdx.x * 8;
      o[0] = st_entry; o[1] = st_loop0; o[2] = st_loop1; o[3] = st_exit; o[4] = st_cyc1 - st_cyc0;
      o[5] = ((unsigned long long)xcc << 32) | hw; o[6] = ((unsigned long long)m0 << 32) | (unsigned)n0; o[7] = 1;
    }
  }
#endif
}

template <int T, int EPI, int DBG = 0, int QI = 4, int VER = 2>
hipError_t launch_pp2(const WmGemmArgs& a, hipStream_t s) {
  constexpr int BM = 64 * QI;
  constexpr size_t shm = (size_t)2 * (BM + 256) * 128 + (EPI == WM_EPI_RESID ? 2048 : 0);   // + the residual prefetch's scratch (8 waves x 256 B)
  const int ntn = (a.N + 255) / 256, ntm = a.sched_bands > 0 ? a.sched_bands : (a.M + BM - 1) / BM;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_pp2_kernel<T, EPI, 1, DBG, QI, VER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    attr = true;
  }
  hipLaunchKernelGGL((gemm_pp2_kernel<T, EPI, 1, DBG, QI, VER>), dim3(ntm * ntn), dim3(512), shm, s, a);
  return hipGetLastError();
}

template <int T, int EPI, int QI, int DBG = 0>
hipError_t launch_pp(const WmGemmArgs& a, hipStream_t s) {
  constexpr int BM = 64 * QI, BN = 256;
  constexpr size_t shm = (size_t)2 * (BM + BN) * 128;
  const int ntn = (a.N + BN - 1) / BN, ntm = (a.M + BM - 1) / BM;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<T, EPI, QI, 1, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    attr = true;
  }
  hipLaunchKernelGGL((gemm_pp_kernel<T, EPI, QI, 1, DBG>), dim3(ntm * ntn), dim3(512), shm, s, a);
  return hipGetLastError();
}

template <int T, int EPI>
hipError_t launch_pp_E(const WmGemmArgs& a, int cfg, hipStream_t s) {
  // gemm_pp tuning: 3 = v1, 4 = v3 (half the barriers: bit-identical, measured +0.1 - 0.3 % on the forward, i.e. nothing — kept
  // selectable and under test as the record of that experiment), anything else = v2 (a barrier on both sides of every MFMA stage)
  const int ver = EPI == WM_EPI_CONV ? 2 : wm_tuning[WM_TUNE_GEMM_PP];   // the pixel-row addressing of WM_EPI_CONV lives in v2 only
  if constexpr (EPI == WM_EPI_CONV) {
    return cfg == 4 ? launch_pp2<T, EPI>(a, s) : launch_pp2<T, EPI, 0, 3>(a, s);
  } else {
    if (ver == 4) return cfg == 4 ? launch_pp2<T, EPI, 0, 4, 3>(a, s) : launch_pp2<T, EPI, 0, 3, 3>(a, s);
    if (ver != 3) return cfg == 4 ? launch_pp2<T, EPI>(a, s) : launch_pp2<T, EPI, 0, 3>(a, s);
    return cfg == 5 ? launch_pp<T, EPI, 3>(a, s) : launch_pp<T, EPI, 4>(a, s);
  }
}

// tile configurations: id -> (WM, WN, TM, TN, NSTAGE)
//  0: 128x128, 4 waves, 2-stage (64 KiB, 2 blocks/CU)     1: 256x128, 8 waves, 3-stage (144 KiB)
//  2: 128x256, 8 waves, 3-stage (144 KiB)                 3: 128x128, 8 waves (32x64 per wave), 4-stage (128 KiB)
//  4: 256x256, 8 waves (128x64 per wave), 2-stage (128 KiB)   5: 192x256, 8 waves (96x64 per wave), 2-stage (112 KiB)
template <int T, int EPI>
hipError_t launch_E(const WmGemmArgs& a, int cfg, hipStream_t s) {
  switch (cfg) {
    case 0: return launch_cfg<T, EPI, 2, 2, 2, 2, 2, 0>(a, s);
    case 1: return launch_cfg<T, EPI, 4, 2, 2, 2, 3, 0>(a, s);
    case 2: return launch_cfg<T, EPI, 4, 2, 2, 2, 3, 1>(a, s);
    case 3: return launch_cfg<T, EPI, 2, 4, 4, 2, 2, 1>(a, s);
    case 4: return launch_cfg<T, EPI, 2, 4, 4, 2, 2, 0>(a, s);
    case 5: return launch_cfg<T, EPI, 2, 4, 3, 2, 2, 0>(a, s);
    case 6: return launch_cfg<T, EPI, 2, 4, 4, 2, 4, 0, 32>(a, s);  // 256x256, K-tile 32, 4-stage ring (128 KiB)
    default: return hipErrorInvalidValue;
  }
}

template <int T>
hipError_t launch_T(const WmGemmArgs& a, int cfg, hipStream_t s) {
  // 16x16x32 MFMA main loop: measured +3..8 % on the N=1024 GEMMs (proj, fc2) and +0..2 % on QKV, neutral/negative
  // on fc1 (tools/bench_gemm.py (rounds 1-2; git history)); WM_GEMM_MFMA16 = 0 / 2 forces it off / on for every backbone epilogue
  static const int mf16_env = [] { const char* e = wm_env("WM_GEMM_MFMA16"); return e ? atoi(e) : 1; }();
  static const int pp_env = [] { const char* e = wm_env("WM_GEMM_PP"); return e ? atoi(e) : 1; }();
  const int mf16 = wm_tuning[WM_TUNE_GEMM_MFMA16] >= 0 ? wm_tuning[WM_TUNE_GEMM_MFMA16] : mf16_env;
  const int pp = wm_tuning[WM_TUNE_GEMM_PP] >= 0 ? wm_tuning[WM_TUNE_GEMM_PP] : pp_env;
#ifdef WM_GEMM_PP_DEBUG  // timing experiments only (results are wrong): 11 no DMA, 12 no ds_read, 13 no barriers
  if (pp > 10 && a.epi == WM_EPI_F32 && T == WM_T_BF16)
    if (pp == 14) return launch_pp2<T, WM_EPI_F32, 4>(a, s);
  if (pp == 15 && a.epi == WM_EPI_F32 && T == WM_T_BF16) return cfg == 5 ? launch_pp2<T, WM_EPI_F32, 5, 3>(a, s) : launch_pp2<T, WM_EPI_F32, 5>(a, s);
  if (pp == 16 && a.epi == WM_EPI_F32 && T == WM_T_BF16) return cfg == 5 ? launch_pp2<T, WM_EPI_F32, 6, 3>(a, s) : launch_pp2<T, WM_EPI_F32, 6>(a, s);
  if (pp >= 100 && pp < 116 && a.epi == WM_EPI_F32 && T == WM_T_BF16) {
    switch (pp - 100) {
#define WM_DBGM(m_) case m_: return cfg == 5 ? launch_pp2<T, WM_EPI_F32, 16 + m_, 3>(a, s) : launch_pp2<T, WM_EPI_F32, 16 + m_>(a, s);
      WM_DBGM(1) WM_DBGM(2) WM_DBGM(4) WM_DBGM(8) WM_DBGM(3) WM_DBGM(7) WM_DBGM(15) WM_DBGM(5) WM_DBGM(12)
#undef WM_DBGM
      default: break;
    }
  }
  if (pp == 17 && a.epi == WM_EPI_F32 && T == WM_T_BF16) return cfg == 5 ? launch_pp2<T, WM_EPI_F32, 4, 3>(a, s) : launch_pp2<T, WM_EPI_F32, 4>(a, s);
  if (pp > 10 && a.epi == WM_EPI_F32 && T == WM_T_BF16)
    return pp == 11 ? launch_pp<T, WM_EPI_F32, 4, 1>(a, s) : pp == 12 ? launch_pp<T, WM_EPI_F32, 4, 2>(a, s) : launch_pp<T, WM_EPI_F32, 4, 3>(a, s);
#endif
  if (a.epi == WM_EPI_CONV) return cfg == 4 || cfg == 5 ? launch_pp_E<T, WM_EPI_CONV>(a, cfg, s) : hipErrorInvalidValue;   // the ping-pong v2 kernel only
  if (pp && (cfg == 4 || cfg == 5)) {
    switch (a.epi) {
      case WM_EPI_F32: return launch_pp_E<T, WM_EPI_F32>(a, cfg, s);
      case WM_EPI_T16: return launch_pp_E<T, WM_EPI_T16>(a, cfg, s);
      case WM_EPI_GELU_T16: return launch_pp_E<T, WM_EPI_GELU_T16>(a, cfg, s);
      case WM_EPI_RESID: return launch_pp_E<T, WM_EPI_RESID>(a, cfg, s);
      case WM_EPI_QKV: return launch_pp_E<T, WM_EPI_QKV>(a, cfg, s);
      default: break;
    }
  }
  if (mf16 && (cfg == 4 || cfg == 5)) {
    switch (a.epi) {
      case WM_EPI_F32: if (mf16 == 2) return launch16_E<T, WM_EPI_F32>(a, cfg, s); break;
      case WM_EPI_T16: if (mf16 == 2) return launch16_E<T, WM_EPI_T16>(a, cfg, s); break;
      case WM_EPI_GELU_T16: if (mf16 == 2) return launch16_E<T, WM_EPI_GELU_T16>(a, cfg, s); break;
      case WM_EPI_RESID: return launch16_E<T, WM_EPI_RESID>(a, cfg, s);
      case WM_EPI_QKV: return launch16_E<T, WM_EPI_QKV>(a, cfg, s);
      default: break;
    }
  }
  switch (a.epi) {
    case WM_EPI_F32: return launch_E<T, WM_EPI_F32>(a, cfg, s);
    case WM_EPI_T16: return launch_E<T, WM_EPI_T16>(a, cfg, s);
    case WM_EPI_GELU_T16: return launch_E<T, WM_EPI_GELU_T16>(a, cfg, s);
    case WM_EPI_RESID: return launch_E<T, WM_EPI_RESID>(a, cfg, s);
    case WM_EPI_ROWMAP_ADD: return launch_E<T, WM_EPI_ROWMAP_ADD>(a, cfg, s);
    case WM_EPI_CONVT: return launch_E<T, WM_EPI_CONVT>(a, cfg, s);
    case WM_EPI_QKV: return launch_E<T, WM_EPI_QKV>(a, cfg, s);
    default: return hipErrorInvalidValue;
  }
}

int pick_cfg(const WmGemmArgs& a) {
  static const int forced = [] { const char* e = wm_env("WM_GEMM_CFG"); return e ? atoi(e) : -1; }();
  const bool conv = a.epi == WM_EPI_CONV;   // lives in the ping-pong v2 kernel only: tile 4 or 5 whatever the size
  if (wm_tuning[WM_TUNE_GEMM_CFG] >= 0 && (!conv || wm_tuning[WM_TUNE_GEMM_CFG] == 4 || wm_tuning[WM_TUNE_GEMM_CFG] == 5)) return wm_tuning[WM_TUNE_GEMM_CFG];
  if (forced >= 0 && (!conv || forced == 4 || forced == 5)) return forced;
  if (!conv && (a.M <= 128 || a.N <= 128)) return 0;
  // minimise (rounds over the CUs) x (tile area / relative tile efficiency): tile quantisation is the
  // first-order loss at M = 11008 (e.g. 516 tiles of 256^2 on 256 CUs = 3 rounds)
  static const int ncu = [] { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&pr, d) == hipSuccess ? pr.multiProcessorCount : 256; }();
  struct Cand { int id, bm, bn, per_cu; float eff; };
  // relative tile efficiencies measured with tools/check_gemm_pp.py (rounds 1-2; git history) (cfg 4 = ping-pong v2, cfg 5 = ping-pong v1 at 192 x 256)
  static const Cand cands[] = {{4, 256, 256, 1, 1.00f}, {5, 192, 256, 1, 0.85f}, {1, 256, 128, 1, 0.70f}, {0, 128, 128, 2, 0.58f}};
  int best = 4;
  float best_cost = 1e30f;
  for (const Cand& c : cands) {
    if (conv && c.id != 4 && c.id != 5) continue;
    const long tiles = (long)((a.M + c.bm - 1) / c.bm) * ((a.N + c.bn - 1) / c.bn);
    const long rounds = (tiles + (long)ncu * c.per_cu - 1) / ((long)ncu * c.per_cu);
    const float cost = (float)rounds * c.per_cu * c.bm * c.bn / c.eff;  // co-resident blocks share the CU
    if (cost < best_cost) { best_cost = cost; best = c.id; }
  }
  return best;
}

// Row-band schedule of the ping-pong v2 kernel (see gemm_pp2_kernel): how many row bands B to cut the M rows into, and on
// which tile height (cfg 4 = up to 16 units of 16 rows, cfg 5 = up to 12).  Candidates are the B that fill R = 1, 2, ... whole
// rounds of the CUs (B = floor(R ncu / ntn)) and the fewest bands the tile height allows; cost = rounds x one block's time,
// with the block's time from profiles/r04_gemm_timeline.md (prologue 1.6 us; per 1024 of K the full tile's K loop takes
// 1.5 us per 16-row unit on the 12-unit tile and 1.43 on the 16-unit one, but a unit CUT from a tile gives back only ~0.5 us:
// the loop is bound by its LDS traffic, which does not shrink with the MFMAs; ~0.35 us of epilogue per unit).
// forced_cfg >= 0 keeps the tile height.
void pick_sched(const WmGemmArgs& a, int ncu, int forced_cfg, int& cfg, int& bands) {
  const int U = (a.M + 15) / 16, ntn = (a.N + 255) / 256;
  const float kf = (float)a.K / 1024.0f;
  float best = 1e30f;
  int best_cfg = cfg, best_b = 0;
  for (int c = 4; c <= 5; ++c) {
    if (forced_cfg >= 0 && c != forced_cfg) continue;
    const int full = c == 4 ? 16 : 12;
    const float unit = c == 4 ? 1.43f : 1.5f;
    const int bmin = (U + full - 1) / full;
    for (int k = 0; k < 6; ++k) {
      int B;
      if (k == 0) B = bmin;
      else {
        const long r0 = ((long)bmin * ntn + ncu - 1) / ncu;   // rounds of the fewest-bands candidate
        B = (int)(((r0 + k - 1) * ncu) / ntn);
      }
      if (B < bmin) continue;
      if (B > U) B = U;
      const int smax = (U + B - 1) / B;
      if (U / B < full - 4) continue;   // the kernel drops at most 2 units per wave group
      const long rounds = ((long)B * ntn + ncu - 1) / ncu;
      const float loop = kf * (full * unit - 0.5f * (float)(full - smax));
      const float cost = (float)rounds * (1.6f + loop + 0.35f * (float)smax);
      if (cost < best) { best = cost; best_cfg = c; best_b = B; }
    }
  }
  cfg = best_cfg;
  bands = best_b;
}

}  // namespace

static int wm_ncu() {
  static const int ncu = [] { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&pr, d) == hipSuccess ? pr.multiProcessorCount : 256; }();
  return ncu;
}
// tile configuration, row-band schedule (0 = full-height grid) and whether the launch is a ping-pong v2 / v3 one
static void plan_gemm(const WmGemmArgs& a, int& cfg, int& sched_b, bool& pp2_out);

bool wm_gemm_fuses_ln(const WmGemmArgs& a) {
  if (!a.ln_out || a.epi != WM_EPI_RESID || a.N != 1024 || !a.ln_w || !a.ln_b || !a.ln_stats || !a.ln_sync || !a.ln_fallback || !a.gamma) return false;
  // OPT-IN (tuning ln_fuse = 1).  Built and proven in round 4 (tests/test_gpu_ops.py::test_gemm_residual_with_fused_layernorm, zero fallbacks in
  // the forward: tools/ln_fallback_count.py) and measured: per launch -2.5 us (proj) / -0.8 us (fc2) against GEMM + LayerNorm kernel in isolation,
  // and between -0.15 and +0.87 ms on the 8-view forward by box (interleaved: 52.38 / 52.50 fused vs 52.39 / 52.74 on one box, 53.49 vs 52.62 on
  // another): the rendezvous' signal -> poll -> partial loads chain costs what the LayerNorm kernel's second read of a stream that the residual
  // epilogue has just left in the memory-side cache costs.  No gain to ship a second synchronisation structure for: off by default.
  if (wm_tuning[WM_TUNE_LN_FUSE] != 1 || (a.ln_ld & 7) || (a.ldc & 3)) return false;
  int cfg, sched_b; bool pp2;
  plan_gemm(a, cfg, sched_b, pp2);
  if (!pp2 || cfg != 5) return false;   // the 192-row tile's kernel carries the fused epilogue
  const int bm = cfg == 4 ? 256 : 192;
  const long bands = sched_b > 0 ? sched_b : (a.M + bm - 1) / bm;
  return bands * 4 <= wm_ncu();   // every block of the launch resident at once (one per CU): the rendezvous of a band's four blocks cannot wait on an undispatched one
}

hipError_t wm_launch_gemm_ln_fallback(const WmGemmArgs& a, hipStream_t s) {
  if (!wm_gemm_fuses_ln(a)) return hipSuccess;
  int cfg, sched_b; bool pp2;
  plan_gemm(a, cfg, sched_b, pp2);
  const int bm = cfg == 4 ? 256 : 192;
  const int bands = sched_b > 0 ? sched_b : (a.M + bm - 1) / bm;
  WmGemmArgs b = a;
  b.sched_bands = sched_b; b.sched_units = (a.M + 15) / 16;
  if (a.dtype == WM_T_BF16) hipLaunchKernelGGL((gemm_ln_fallback_kernel<WM_T_BF16>), dim3(bands), dim3(256), 0, s, b, bands, bm / 16);
  else hipLaunchKernelGGL((gemm_ln_fallback_kernel<WM_T_F16>), dim3(bands), dim3(256), 0, s, b, bands, bm / 16);
  return hipGetLastError();
}

static void plan_gemm(const WmGemmArgs& a, int& cfg, int& sched_b, bool& pp2_out) {
  cfg = pick_cfg(a);
  sched_b = 0;
  {
    // ping-pong v2 launches only (launch_T): backbone epilogues on the 256- / 192-row tiles
    const int ncu = wm_ncu();
    const int pp = wm_tuning[WM_TUNE_GEMM_PP];
    const bool pp2 = (cfg == 4 || cfg == 5) && (pp < 0 || pp == 1 || pp == 2 || pp == 4 || a.epi == WM_EPI_CONV) && (a.epi == WM_EPI_F32 || a.epi == WM_EPI_T16 || a.epi == WM_EPI_GELU_T16 || a.epi == WM_EPI_RESID || a.epi == WM_EPI_QKV || a.epi == WM_EPI_CONV);
    const int ts = wm_tuning[WM_TUNE_GEMM_SCHED];   // -1 choose, 0 off (full-height tiles), > 0 that many bands
    if (pp2 && ts != 0) {
      // Measured (profiles/r04_gemm_timeline.md, `sched` rows): at M = 11008 the schedule takes 2.4 - 5.1 % off all four backbone
      // GEMMs (2.7 -> 3.0 and 0.9 -> 1.0 rounds of shorter blocks); at M = 44032 (8 - 11 rounds, where a ragged last round
      // costs at most a tenth) it measured +5 / +4 / +1.6 / -2.3 %.  So: only launches of at most three rounds.
      const long legacy_tiles = (long)((a.M + (cfg == 4 ? 255 : 191)) / (cfg == 4 ? 256 : 192)) * ((a.N + 255) / 256);
      if (ts > 0) sched_b = ts;
      else if (legacy_tiles <= 3L * ncu && a.M <= 16384) pick_sched(a, ncu, wm_tuning[WM_TUNE_GEMM_CFG], cfg, sched_b);
      const int U = (a.M + 15) / 16, full = cfg == 4 ? 16 : 12;
      if (sched_b > U) sched_b = U;
      // bands taller than the tile, or shorter than the kernel's instantiations go (a wave group drops at most 2 units): full-height grid
      if (sched_b <= 0 || (U + sched_b - 1) / sched_b > full || U / sched_b < full - 4) sched_b = 0;
    }
    pp2_out = pp2 && wm_tuning[WM_TUNE_GEMM_PP] != 0;
    static const int pp_env = [] { const char* e = wm_env("WM_GEMM_PP"); return e ? atoi(e) : 1; }();
    if (wm_tuning[WM_TUNE_GEMM_PP] < 0 && !pp_env) pp2_out = false;
    if (a.epi == WM_EPI_CONV) pp2_out = true;
  }
}

hipError_t wm_launch_gemm(const WmGemmArgs& a, hipStream_t s) {
  if (a.M <= 0 || a.N <= 0) return hipSuccess;
  if (a.K <= 0 || a.K % 64 != 0 || a.N % 4 != 0) return hipErrorInvalidValue;
  if ((a.lda & 7) || (a.ldw & 7)) return hipErrorInvalidValue;  // 16-B aligned rows
  if (a.epi == WM_EPI_CONVT && (a.ct_cout & 3)) return hipErrorInvalidValue;
  if (a.epi != WM_EPI_CONVT && a.epi != WM_EPI_QKV && (a.ldc & 3)) return hipErrorInvalidValue;
  if (a.epi == WM_EPI_QKV && (a.N % 64 || a.N != 3 * a.qkv.H * 64)) return hipErrorInvalidValue;
  if (a.epi == WM_EPI_CONV && a.tc_k <= 0 && (a.cv_h <= 0 || a.cv_w <= 0 || a.cv_cin % 64 || a.K != 9 * a.cv_cin || a.M % (a.cv_h * a.cv_w) || !a.cv_zero || (a.N & 7))) return hipErrorInvalidValue;
  if (a.epi == WM_EPI_CONV && a.tc_k > 0 && (a.cv_h <= 0 || a.cv_w <= 0 || a.cv_cin % 64 || a.K != 4 * a.cv_cin || a.M % (a.cv_h * a.cv_w) || !a.cv_zero || a.tc_k > 4 ||
                                             a.N != a.tc_k * a.tc_k * 256 || a.ldc != 256 || a.cv_resid || a.cv_resid2))
    return hipErrorInvalidValue;
  int cfg, sched_b;
  bool is_pp2;
  plan_gemm(a, cfg, sched_b, is_pp2);
  const bool fuse_ln = wm_gemm_fuses_ln(a);
  const int gb = wm_tuning[WM_TUNE_GEMM_GROUP] >= 0 ? wm_tuning[WM_TUNE_GEMM_GROUP] : 6;  // row bands per supertile: 6 measured 2-3 % ahead of 4 / 8 at 32 views, equal at 8 (tools/bench_gemm_group.py)
  if (a.epi == WM_EPI_QKV) {
    if (a.qkv.tokens_per_view <= 0 || a.qkv.grid_w <= 0 || a.M >= (1 << 20) || a.qkv.tokens_per_view >= (1 << 16)) return hipErrorInvalidValue;
    WmGemmArgs b = a;
    b.group_bands = gb;
    b.sched_bands = sched_b; b.sched_units = (a.M + 15) / 16;
    b.qkv.inv_tpv = 1.0f / (float)a.qkv.tokens_per_view;
    b.qkv.inv_gw = 1.0f / (float)a.qkv.grid_w;
    return b.dtype == WM_T_BF16 ? launch_T<WM_T_BF16>(b, cfg, s) : launch_T<WM_T_F16>(b, cfg, s);
  }
  WmGemmArgs c = a;
  c.group_bands = gb;
  c.sched_bands = sched_b; c.sched_units = (a.M + 15) / 16;
  c.pf_c = a.epi == WM_EPI_RESID && wm_tuning[WM_TUNE_RESID_PREFETCH] == 1 ? 1 : 0;
  if (!fuse_ln) c.ln_out = nullptr;   // the kernel takes the fused epilogue iff ln_out is set
  return c.dtype == WM_T_BF16 ? launch_T<WM_T_BF16>(c, cfg, s) : launch_T<WM_T_F16>(c, cfg, s);
}

#ifdef WM_GEMM_STAMPS
extern "C" int wm_debug_gemm_stamps(unsigned long long* host_out, int nblocks) {
  if (nblocks > 8192) nblocks = 8192;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(wm_gemm_stamp_buf), (size_t)nblocks * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
extern "C" int wm_debug_ln_fallback_count() {   // bands the fused-LayerNorm fallback kernel recomputed since the last call (then reset)
  int v = -1, z = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(wm_ln_fallback_count), 4) != hipSuccess) return -1;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(wm_ln_fallback_count), &z, 4);
  return v;
}
extern "C" int wm_debug_gemm_stamps_clear() {
  void* d = nullptr;
  if (hipGetSymbolAddress(&d, HIP_SYMBOL(wm_gemm_stamp_buf)) != hipSuccess) return 1;
  return hipMemset(d, 0, sizeof(unsigned long long) * 8 * 8192) == hipSuccess ? 0 : 1;
}
#endif
