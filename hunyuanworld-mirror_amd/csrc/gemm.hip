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

int wm_tuning[WM_TUNE_COUNT] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};

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
__device__ __forceinline__ void epilogue16(const WmGemmArgs& p, f32x4 (&acc)[SM][SN], int rowb, int colb, int lane) {
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
      const int rr = row < p.M ? row : 0;
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
      const int row = rowb + i * 16 + l15;
      const bool row_ok = row < p.M;
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
            if (row < p.M && col < p.N) *(uint4*)((u16*)p.C + (size_t)row * p.ldc + col) = make_uint4(u[0].x, u[0].y, u[1].x, u[1].y);
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
          o[j] = (row < p.M && col < p.N) ? *(const float4*)((const float*)p.C + (size_t)row * p.ldc + col) : make_float4(0, 0, 0, 0);
        }
      };
      load_old(0, old[0]);
#pragma unroll
      for (int i = 0; i < SM; ++i) {
        if (i + 1 < SM) load_old(i + 1, old[(i + 1) & 1]);
        const int row = rowb + i * 16 + l15;
#pragma unroll
        for (int j = 0; j < SN; ++j) {
          const int col = colb + j * 16 + 4 * lq;
          const float4 o = old[i & 1][j], gm = gm4[j], bs = bs4[j];
          if (row < p.M && col < p.N) {
            const float4 nv = make_float4(o.x + gm.x * (acc[i][j][0] + bs.x), o.y + gm.y * (acc[i][j][1] + bs.y), o.z + gm.z * (acc[i][j][2] + bs.z), o.w + gm.w * (acc[i][j][3] + bs.w));
            *(float4*)((float*)p.C + (size_t)row * p.ldc + col) = nv;
            if (p.C2) *(float4*)(p.C2 + (size_t)row * p.ldc2 + col) = nv;  // tap half (block-uniform branch)
          }
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < SM; ++i) {
      const int row = rowb + i * 16 + l15;
      if (row >= p.M) continue;
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
  epilogue16<T, EPI, SM, SN>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, lane);
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
  epilogue16<T, EPI, SM, SN>(p, acc, m0 + wr * WROWS, n0 + wc * 64, lane);
}

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
template <int T, int EPI, int PRIO, int DBG = 0, int QI = 4>
__global__ __launch_bounds__(512) void gemm_pp2_kernel(const WmGemmArgs p) {
  constexpr int SM = 2 * QI, SN = 4, WROWS = SM * 16, BM = 2 * WROWS, BN = 256;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int APC = QI * 2;  // 8-row pieces per (wave row half, quadrant): 8 or 6
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int lid = xcd_remap(blockIdx.x, ntm * ntn);
  int band, nt;
  tile_of(lid, ntm, ntn, p.group_bands, band, nt);
  const int m0 = band * BM, n0 = nt * BN;
  const int l15 = lane & 15, lq = lane >> 4;
  const bool two = QI == 4 || wave < 4;  // this wave carries two A pieces per region (wave-uniform)

  // this wave's DMA pieces per K-tile, in issue order: A0 A0 B0 B0 | B1 B1 | A1 A1 (the second A piece only if `two`)
  const u16* gp[8];
  int loff[8];
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
    int gr = (DBG == 4 ? 0 : (isA ? m0 : n0)) + r;
    const int lim = (isA ? p.M : p.N) - 1;
    gr = gr < lim ? gr : lim;
    gp[i] = (isA ? (const u16*)p.A + (size_t)gr * p.lda : (const u16*)p.W + (size_t)gr * p.ldw) + c * 8;
    loff[i] = (isA ? 0 : A_BYTES) + pl * 1024;
  }
  auto dma = [&](int buf, int i0, int i1) {
#pragma unroll
    for (int i = i0; i < i1; ++i) {
      if ((i == 1 || i == 7) && !two) continue;  // wave-uniform
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

  const int nk = p.K / 64;
  dma(0, 0, 8);
  if (nk > 1) { dma(1, 0, 8); WM_W2(12, 9); } else WM_W2(4, 3);
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();

  auto stage_end = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (PRIO) __builtin_amdgcn_s_setprio(1);
  };
  auto mfma_end = [&]() {
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };

  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    const char* tile = smem + buf * STAGE;
    const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
    // ---- phase 0: quadrant (0,0)
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
    if (n1) WM_W2(10, 7); else WM_W2(2, 1);
    stage_end();
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < QI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<T>(b0[j][kh], a[i][kh], acc[i][j]);
    mfma_end();
    // ---- phase 1: quadrant (0,1); refill {A0,B0} with K-tile t+2
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      b1[j][0] = *(const s16x8*)(tile + b_base + (2 + j) * 2048 + foff0);
      b1[j][1] = *(const s16x8*)(tile + b_base + (2 + j) * 2048 + foff1);
    }
    if (n2) { dma(buf, 0, 4); WM_W2(12, 9); } else if (n1) WM_W2(8, 6); else wait_vmcnt<0>();
    stage_end();
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < QI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][2 + j] = mfma16<T>(b1[j][kh], a[i][kh], acc[i][2 + j]);
    mfma_end();
    // ---- phase 2: quadrant (1,1); refill B1
#pragma unroll
    for (int i = 0; i < QI; ++i) {
      a[i][0] = *(const s16x8*)(tile + a_base + (QI + i) * 2048 + foff0);
      a[i][1] = *(const s16x8*)(tile + a_base + (QI + i) * 2048 + foff1);
    }
    if (n2) dma(buf, 4, 6);
    stage_end();
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < QI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[QI + i][2 + j] = mfma16<T>(b1[j][kh], a[i][kh], acc[QI + i][2 + j]);
    mfma_end();
    // ---- phase 3: quadrant (1,0) from registers; refill A1; next K-tile's {A0,B0} must have landed
    if (n2) { dma(buf, 6, 8); WM_W2(12, 9); } else if (n1) WM_W2(4, 3);
    stage_end();
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < QI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[QI + i][j] = mfma16<T>(b0[j][kh], a[i][kh], acc[QI + i][j]);
    mfma_end();
  }
#undef WM_W2
  if (wr == 0) __builtin_amdgcn_s_barrier();
  if (DBG == 5 && acc[0][0][0] != 1.2345e-30f) return;  // timing experiment: no epilogue
  epilogue16<T, EPI, SM, SN>(p, acc, m0 + wr * WROWS, n0 + wc * 64, lane);
}

template <int T, int EPI, int DBG = 0, int QI = 4>
hipError_t launch_pp2(const WmGemmArgs& a, hipStream_t s) {
  constexpr int BM = 64 * QI;
  constexpr size_t shm = (size_t)2 * (BM + 256) * 128;
  const int ntn = (a.N + 255) / 256, ntm = (a.M + BM - 1) / BM;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_pp2_kernel<T, EPI, 1, DBG, QI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    attr = true;
  }
  hipLaunchKernelGGL((gemm_pp2_kernel<T, EPI, 1, DBG, QI>), dim3(ntm * ntn), dim3(512), shm, s, a);
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
  if (cfg == 4 && wm_tuning[WM_TUNE_GEMM_PP] != 3) return launch_pp2<T, EPI>(a, s);  // gemm_pp = 3 forces v1
  if (cfg == 5 && wm_tuning[WM_TUNE_GEMM_PP] != 3) return launch_pp2<T, EPI, 0, 3>(a, s);
  return cfg == 5 ? launch_pp<T, EPI, 3>(a, s) : launch_pp<T, EPI, 4>(a, s);
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
  // on fc1 (tools/bench_gemm.py); WM_GEMM_MFMA16 = 0 / 2 forces it off / on for every backbone epilogue
  static const int mf16_env = [] { const char* e = getenv("WM_GEMM_MFMA16"); return e ? atoi(e) : 1; }();
  static const int pp_env = [] { const char* e = getenv("WM_GEMM_PP"); return e ? atoi(e) : 1; }();
  const int mf16 = wm_tuning[WM_TUNE_GEMM_MFMA16] >= 0 ? wm_tuning[WM_TUNE_GEMM_MFMA16] : mf16_env;
  const int pp = wm_tuning[WM_TUNE_GEMM_PP] >= 0 ? wm_tuning[WM_TUNE_GEMM_PP] : pp_env;
#ifdef WM_GEMM_PP_DEBUG  // timing experiments only (results are wrong): 11 no DMA, 12 no ds_read, 13 no barriers
  if (pp > 10 && a.epi == WM_EPI_F32 && T == WM_T_BF16)
    if (pp == 14) return launch_pp2<T, WM_EPI_F32, 4>(a, s);
  if (pp == 15 && a.epi == WM_EPI_F32 && T == WM_T_BF16) return cfg == 5 ? launch_pp2<T, WM_EPI_F32, 5, 3>(a, s) : launch_pp2<T, WM_EPI_F32, 5>(a, s);
  if (pp > 10 && a.epi == WM_EPI_F32 && T == WM_T_BF16)
    return pp == 11 ? launch_pp<T, WM_EPI_F32, 4, 1>(a, s) : pp == 12 ? launch_pp<T, WM_EPI_F32, 4, 2>(a, s) : launch_pp<T, WM_EPI_F32, 4, 3>(a, s);
#endif
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
  static const int forced = [] { const char* e = getenv("WM_GEMM_CFG"); return e ? atoi(e) : -1; }();
  if (wm_tuning[WM_TUNE_GEMM_CFG] >= 0) return wm_tuning[WM_TUNE_GEMM_CFG];
  if (forced >= 0) return forced;
  if (a.M <= 128 || a.N <= 128) return 0;
  // minimise (rounds over the CUs) x (tile area / relative tile efficiency): tile quantisation is the
  // first-order loss at M = 11008 (e.g. 516 tiles of 256^2 on 256 CUs = 3 rounds)
  static const int ncu = [] { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&pr, d) == hipSuccess ? pr.multiProcessorCount : 256; }();
  struct Cand { int id, bm, bn, per_cu; float eff; };
  // relative tile efficiencies measured with tools/check_gemm_pp.py (cfg 4 = ping-pong v2, cfg 5 = ping-pong v1 at 192 x 256)
  static const Cand cands[] = {{4, 256, 256, 1, 1.00f}, {5, 192, 256, 1, 0.85f}, {1, 256, 128, 1, 0.70f}, {0, 128, 128, 2, 0.58f}};
  int best = 4;
  float best_cost = 1e30f;
  for (const Cand& c : cands) {
    const long tiles = (long)((a.M + c.bm - 1) / c.bm) * ((a.N + c.bn - 1) / c.bn);
    const long rounds = (tiles + (long)ncu * c.per_cu - 1) / ((long)ncu * c.per_cu);
    const float cost = (float)rounds * c.per_cu * c.bm * c.bn / c.eff;  // co-resident blocks share the CU
    if (cost < best_cost) { best_cost = cost; best = c.id; }
  }
  return best;
}

}  // namespace

hipError_t wm_launch_gemm(const WmGemmArgs& a, hipStream_t s) {
  if (a.M <= 0 || a.N <= 0) return hipSuccess;
  if (a.K <= 0 || a.K % 64 != 0 || a.N % 4 != 0) return hipErrorInvalidValue;
  if ((a.lda & 7) || (a.ldw & 7)) return hipErrorInvalidValue;  // 16-B aligned rows
  if (a.epi == WM_EPI_CONVT && (a.ct_cout & 3)) return hipErrorInvalidValue;
  if (a.epi != WM_EPI_CONVT && a.epi != WM_EPI_QKV && (a.ldc & 3)) return hipErrorInvalidValue;
  if (a.epi == WM_EPI_QKV && (a.N % 64 || a.N != 3 * a.qkv.H * 64)) return hipErrorInvalidValue;
  const int cfg = pick_cfg(a);
  const int gb = wm_tuning[WM_TUNE_GEMM_GROUP] >= 0 ? wm_tuning[WM_TUNE_GEMM_GROUP] : 6;  // row bands per supertile: 6 measured 2-3 % ahead of 4 / 8 at 32 views, equal at 8 (tools/bench_gemm_group.py)
  if (a.epi == WM_EPI_QKV) {
    if (a.qkv.tokens_per_view <= 0 || a.qkv.grid_w <= 0 || a.M >= (1 << 20) || a.qkv.tokens_per_view >= (1 << 16)) return hipErrorInvalidValue;
    WmGemmArgs b = a;
    b.group_bands = gb;
    b.qkv.inv_tpv = 1.0f / (float)a.qkv.tokens_per_view;
    b.qkv.inv_gw = 1.0f / (float)a.qkv.grid_w;
    return b.dtype == WM_T_BF16 ? launch_T<WM_T_BF16>(b, cfg, s) : launch_T<WM_T_F16>(b, cfg, s);
  }
  WmGemmArgs c = a;
  c.group_bands = gb;
  return c.dtype == WM_T_BF16 ? launch_T<WM_T_BF16>(c, cfg, s) : launch_T<WM_T_F16>(c, cfg, s);
}
