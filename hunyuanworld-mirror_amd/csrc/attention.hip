// Flash-attention forward, head_dim 64, 16-bit operands, fp32 online softmax — gfx950.
//
// Replaces F.scaled_dot_product_attention at src/models/layers/attention.py:59 for all three users:
// DINO blocks (L=1374), frame blocks (L=1376) and the cross-view global blocks (L = N*1376).
//
// Layout: Q,K,V are [H][rows][64] (head-major; produced by the QKV epilogue kernel), so frame and
// global attention share one buffer: a "sequence" is a contiguous row range.  O is token-major
// [rows][H*64], i.e. directly the A operand of the projection GEMM.  For the view-sharded multi-GPU
// path K/V may be `kv_chunks` gathered shards; softmax is permutation-invariant over keys, so the
// all-gather's natural [rank][H][rows][64] order is consumed as is.
//
// Structure (per guides' "Fused attention prefill" recipe, adapted to D=64): one wave owns 32 query
// rows with Q fragments in registers; S^T = K Q^T is computed with K as the MFMA A operand so every
// lane holds one query row's scores (row max/sum need a single permlane32 swap); the S^T accumulator
// is fed straight back as the B operand of O^T += V^T P^T; V^T fragments come from
// ds_read_b64_tr_b16 on a [4 keys][32 d] blocked image (each 256-B block covers all 64 banks);
// K rows are XOR-swizzled for conflict-free ds_read_b128; K/V tiles are register-staged and double
// buffered (global loads issued before the MFMA phase, LDS writes after it).
#include "wm_common.h"
#include "wm_kernels.h"

#include <cstdlib>

namespace {

// One 1-KiB LDS-DMA piece (16 B per lane) from inline asm: M0 carries the LDS destination; it is saved and restored
// around the load so the compiler's own view of M0 stays valid (M0 is a reserved register: a clobber would not be honoured).
__device__ __forceinline__ void wm_dma16(const void* g, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}

constexpr int KVB = 64;                 // keys per tile
constexpr int TILE_B = KVB * 64 * 2;    // 8 KiB per K or V tile
constexpr float LOG2E = 1.4426950408889634f;

typedef __attribute__((address_space(3))) s16x4* lds_s16x4p;

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
// two fp32 -> one packed 16-bit pair: a single v_cvt_pk_{bf16,f16}_f32
__device__ __forceinline__ uint32_t pack2(float a, float b, int T) {
  const f32x2 v = {a, b};
  return T == WM_T_BF16 ? __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2))
                        : __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}

// smallest T-representable INTEGER >= x (x finite): used for the running max so that (a) the value the MFMA subtracts
// (carried as a 16-bit operand) is exactly m_run and scores never exceed it, and (b) P = 2^(S - m_run) has the same
// mantissa whatever m_run is: the 16-bit rounding of P — and with it the result up to fp32 summation order — does not
// depend on the order the key tiles are visited in, on the lazy-max policy, on KV splits or on how keys are sharded
// over ranks (a fractional max would re-round every P by half an ulp of the 16-bit type when it moves).
// ceilf(x) is T-representable for |x| <= 256 (bf16) / 2048 (f16); beyond that every T value is an integer anyway.
template <int T>
__device__ __forceinline__ float ceil_t16(float x) {
  x = ceilf(x);
  u16 u = f2t<T>(x);
  float y = t2f<T>(u);
  if (y < x) {
    if (u == 0x8000) u = 0;
    u = (u & 0x8000) ? (u16)(u - 1) : (u16)(u + 1);
    y = t2f<T>(u);
  }
  return y;
}

template <int T, int NW, int QB, int MINW, int LZ = 1, int STG = 1>
__global__ __launch_bounds__(NW * 64, MINW) void attn_fwd_kernel(const WmAttnArgs p) {
  constexpr int NT = NW * 64;
  constexpr int QT = NW * 32 * QB;  // query rows per block; each wave owns QB blocks of 32 rows
  constexpr int NBUF = STG == 2 ? 3 : 2;  // STG 2: three-deep ring, tiles fetched two ahead
  __shared__ __attribute__((aligned(16))) char smem[NBUF * 2 * TILE_B];  // [buf][K|V]

  if (p.only_if && p.only_if[blockIdx.x] == 0) return;  // recompute pass behind attn_v3_kernel: only the blocks it flagged (block-uniform)
  if (p.only_if && p.unit_hint && threadIdx.x == 0) {    // sticky hint: one call of this unit on the general kernel used up (nothing here reads the hint)
    const int hv = p.unit_hint[blockIdx.x];
    if (hv > 0) p.unit_hint[blockIdx.x] = hv - 1;
  }
  if (p.only_if && p.unit_stat && threadIdx.x == 0) atomicAdd(p.unit_stat, 1);   // units taken by the recompute pass, for the host's policy
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, ql = lane & 31;
  const int tiles_per_seq = (p.seq_len + QT - 1) / QT;
  const int nseq = p.q_rows / p.seq_len;
  const int tiles_per_head = tiles_per_seq * nseq;
  // Units (q-tile, head) [0, full_units) are processed whole by the first full_units blocks; every later unit is walked by
  // kv_splits blocks, each over a slice of the key tiles (partials + combine pass).  full_units = 0 is the uniform split;
  // the launcher uses full_units > 0 to cut only the LAST, partly filled round of a launch into short blocks.
  const int nfull = p.kv_splits > 1 ? p.full_units : tiles_per_head * p.H;
  const bool whole = (int)blockIdx.x < nfull;  // block-uniform
  const int nsplit = whole ? 1 : p.kv_splits;
  int lid, split = 0;
  if (whole) {
    lid = xcd_remap(blockIdx.x, nfull);
  } else {
    lid = xcd_remap(blockIdx.x - nfull, (tiles_per_head * p.H - nfull) * nsplit);
    split = lid % nsplit;
    lid = nfull + lid / nsplit;
  }
  const int head = lid / tiles_per_head;
  const int tile = lid - head * tiles_per_head;
  const int seq = tile / tiles_per_seq;
  const int qt = tile - seq * tiles_per_seq;
  const int seq_row0 = seq * p.seq_len;

  // ---- Q fragments (B operand of S^T = K Q^T): lane (q=ql, h) holds Q[q][16ks + 8h + j]
  const u16* Qh = (const u16*)p.Q + (size_t)head * p.q_head_stride * 64;
  int qrow[QB];
  bool q_valid[QB];
  s16x8 qf[QB][4];
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    int r = qt * QT + (wave * QB + b) * 32 + ql;  // row within the sequence
    q_valid[b] = r < p.seq_len;
    r = q_valid[b] ? r : p.seq_len - 1;
    qrow[b] = r;
    const u16* qptr = Qh + (size_t)(seq_row0 + r) * 64;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[b][ks] = *(const s16x8*)(qptr + ks * 16 + h * 8);
  }

  // ---- K/V segments
  const int seg_rows = p.kv_chunks > 1 ? p.kv_rows_per_chunk : p.seq_len;
  const int seg_off = p.kv_chunks > 1 ? 0 : seq_row0;
  const int ntpc = (seg_rows + KVB - 1) / KVB;
  const int ntiles = ntpc * p.kv_chunks;
  const u16* Kb = (const u16*)p.K + (size_t)head * p.kv_head_stride * 64 + (size_t)seg_off * 64;
  const u16* Vb = (const u16*)p.V + (size_t)head * p.kv_head_stride * 64 + (size_t)seg_off * 64;

  // Tiles stream by LDS-DMA (1-KiB pieces; the source-side permutation builds the swizzled K rows and the blocked V
  // image; issued from inline asm — hipcc would order every later ds_read behind a DMA it can see with a vmcnt(0));
  // the wait is the explicit one before the barrier.  STG 2: three-deep ring, two tiles ahead (measured equal).
  typedef __attribute__((address_space(3))) void* lds_vp0;
  int dma_c = 0, dma_j = 0;  // (chunk, tile in chunk) of the next tile to fetch: tiles are fetched in order from t0
  auto dma_tile = [&](int buf) {
    const u16* kp = Kb + (size_t)dma_c * p.kv_chunk_stride;
    const u16* vp = Vb + (size_t)dma_c * p.kv_chunk_stride;
    const uint32_t dst = (uint32_t)(size_t)(lds_vp0)(smem + buf * 2 * TILE_B);
    constexpr int PPW = 8 / NW;  // pieces of K (and of V) per wave
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pc = wave * PPW + i;
      const int kkey = pc * 8 + (lane >> 3), kd8 = (lane & 7) ^ ((kkey >> 1) & 7);
      const int off = pc * 1024 + lane * 16, blk = off >> 8;
      const int vkey = (blk >> 1) * 4 + ((off >> 6) & 3), vd8 = (blk & 1) * 4 + ((off >> 4) & 3);
      int kr = dma_j * KVB + kkey, vr = dma_j * KVB + vkey;
      kr = kr < seg_rows ? kr : seg_rows - 1;  // clamped keys: masked (K), multiplied by P = 0 (V)
      vr = vr < seg_rows ? vr : seg_rows - 1;
      const u16* gk = kp + (size_t)kr * 64 + kd8 * 8;
      const u16* gv = vp + (size_t)vr * 64 + vd8 * 8;
      wm_dma16(gk, dst + pc * 1024);
      wm_dma16(gv, dst + TILE_B + pc * 1024);
    }
    if (++dma_j == ntpc) { dma_j = 0; ++dma_c; }
  };
  // Scores arrive in log2 units (q is pre-scaled by log2(e)/sqrt(d)).  The running max is subtracted
  // INSIDE the MFMA: one extra k-step multiplies a constant K-side fragment (1.0 at k=0) with a Q-side
  // fragment holding -m_run at k=0, so S' = S - m_run comes out of the matrix pipe and the softmax
  // needs no per-score FMA (the VALU, not the MFMA, bounds this kernel at head_dim 64).
  f32x16 ot[QB][2];
  float m_run[QB], l_run[QB];
  s16x8 qx[QB];
  s16x8 kx;
#pragma unroll
  for (int j = 0; j < 8; ++j) kx[j] = 0;
  if (h == 0) kx[0] = (short)f2t<T>(1.0f);
#pragma unroll
  for (int b = 0; b < QB; ++b) {
#pragma unroll
    for (int j = 0; j < 8; ++j) qx[b][j] = 0;
    m_run[b] = 0.f;
    l_run[b] = 0.f;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[b][d][r] = 0.f;
  }

  // this block's slice of the key tiles (every slice is non-empty: the launcher keeps kv_splits <= ntiles)
  const int t0 = (int)((long long)split * ntiles / nsplit), t1 = (int)((long long)(split + 1) * ntiles / nsplit);
  constexpr int DPW = 2 * (8 / NW);  // DMA instructions per wave per tile
  dma_c = t0 / ntpc; dma_j = t0 - dma_c * ntpc;
  dma_tile(0);
  if (STG == 2 && t0 + 1 < t1) { dma_tile(1); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW) : "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // per-lane constant part of the transposed V read address (see header)
  const int vtr_lane = ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;

  for (int t = t0; t < t1; ++t) {
    const int cur = STG == 2 ? (t - t0) % 3 : (t - t0) & 1;
    const char* kt = smem + cur * 2 * TILE_B;
    const char* vt = kt + TILE_B;
    if (STG == 2) {
      if (t + 2 < t1) dma_tile((t - t0 + 2) % 3);  // the buffer of tile t-1: everybody passed the barrier that ended it
    } else if (t + 1 < t1) {
      dma_tile(cur ^ 1);  // everybody passed the barrier that ended tile t-1: the other buffer is free
    }

    // ---- S^T = K Q^T : st[b][k2][r] = S[key = 32k2 + (r&3)+8(r>>2)+4h][q = ql of block b]
    f32x16 st[QB][2];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const int key = k2 * 32 + ql;
      const f32x16 zero = {0};
#pragma unroll
      for (int b = 0; b < QB; ++b) st[b][k2] = mfma32<T>(kx, qx[b], zero);  // -m_run broadcast to every key row
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const s16x8 kf = *(const s16x8*)(kt + key * 128 + (((2 * ks + h) ^ ((key >> 1) & 7)) << 4));
#pragma unroll
        for (int b = 0; b < QB; ++b)
          st[b][k2] = mfma32<T>(kf, qf[b][ks], st[b][k2]);
      }
    }
    // ---- tail mask (wave-uniform branch; only the last tile of a segment)
    {
      const int j = t % ntpc;
      const int valid = seg_rows - j * KVB;
      if (valid < KVB) {
#pragma unroll
        for (int b = 0; b < QB; ++b)
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = k2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
              if (key >= valid) st[b][k2][r] = -INFINITY;
            }
      }
    }
    // ---- online softmax in base 2 (q was pre-scaled by log2(e)/sqrt(d)).  LZ: the scores already carry -m_run (MFMA),
    // so P = 2^S' is taken FIRST and the row max only when it is needed: the lane's partial row-sum bounds every P it
    // contains (sum <= 2^11 => each P <= 2^11: full 16-bit relative precision, fp32 l and O far from overflow; inf and
    // NaN fail the test), and only then — or on the first tile — the max is found, O and l rescaled and P redone.
    // Saves the 64 v_max + cross-half exchange per tile of the eager form (the VALU, not the MFMA, is the scarce pipe).
    s16x8 pf[QB][2][2];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
      float pv[2][16], sum = 0.f;
      bool redo = (t == t0) || !LZ;
      if (LZ) {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int r = 0; r < 16; ++r) { pv[k2][r] = __builtin_amdgcn_exp2f(st[b][k2][r]); sum += pv[k2][r]; }
        redo = redo || !__all(sum <= 2048.0f);
      }
      if (redo) {  // wave-uniform
        float mloc = st[b][0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, st[b][0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[b][1][r]);
        mloc = xhalf_max(mloc);  // max of S' = S - m_run over this tile (same value in both lane halves)
        if (t == t0 || !__all(mloc <= 0.f)) {  // some row's max grew (always on the first tile)
          const float cand = m_run[b] + (t == t0 ? mloc : fmaxf(mloc, 0.f));
          const float m_new = ceil_t16<T>(cand);
          const float d2 = m_new - m_run[b];
          m_run[b] = m_new;
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[b][k2][r] -= d2;
          if (t > t0) {
            const float alpha = __builtin_amdgcn_exp2f(-d2);
            l_run[b] *= alpha;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
              for (int r = 0; r < 16; ++r) ot[b][d][r] *= alpha;
          }
          if (h == 0) qx[b][0] = (short)f2t<T>(-m_new);  // exact: m_new is T-representable
        }
        sum = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            pv[k2][r] = __builtin_amdgcn_exp2f(st[b][k2][r]);
            sum += pv[k2][r];
          }
      }
      l_run[b] += sum;
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          uint4 u;
          u.x = pack2(pv[k2][8 * s2 + 0], pv[k2][8 * s2 + 1], T);
          u.y = pack2(pv[k2][8 * s2 + 2], pv[k2][8 * s2 + 3], T);
          u.z = pack2(pv[k2][8 * s2 + 4], pv[k2][8 * s2 + 5], T);
          u.w = pack2(pv[k2][8 * s2 + 6], pv[k2][8 * s2 + 7], T);
          pf[b][k2][s2] = __builtin_bit_cast(s16x8, u);
        }
    }
    // ---- O^T += V^T P^T : A = V^T fragment via transposed LDS reads, B = P^T (the S^T accumulator)
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int key0 = k2 * 32 + s2 * 16 + 4 * h;  // elements 0-3: key0..key0+3 ; 4-7: key0+8..key0+11
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const char* b0 = vt + (((key0 >> 2) * 2 + d) << 8) + vtr_lane;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4p)(b0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4p)(b0 + 2 * 2 * 256));
          s16x8 vf;
          vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
          vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
#pragma unroll
          for (int b = 0; b < QB; ++b)
            ot[b][d] = mfma32<T>(vf, pf[b][k2][s2], ot[b][d]);
        }
      }
    if (STG == 2) {  // tile t+1 must have landed; tile t+2 (if any) may stay in flight
      if (t + 2 < t1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  if (nsplit > 1 || p.force_partial) {  // unnormalised partial: O^T (fp32), running max and sum; the combine pass finishes the softmax
    const int slot = p.part_slot0 + split;
#pragma unroll
    for (int b = 0; b < QB; ++b) {
      const float lsum = xhalf_sum(l_run[b]);
      if (!q_valid[b]) continue;
      const size_t row = (size_t)(seq_row0 + qrow[b]);
      float* op = p.part_o + ((size_t)slot * p.q_rows + row) * (p.H * 64) + head * 64;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(float4*)(op + 32 * d + 8 * g + 4 * h) = make_float4(ot[b][d][4 * g], ot[b][d][4 * g + 1], ot[b][d][4 * g + 2], ot[b][d][4 * g + 3]);
      if (h == 0) *(float2*)(p.part_ml + (((size_t)slot * p.H + head) * p.q_rows + row) * 2) = make_float2(m_run[b], lsum);
    }
    return;
  }

  // ---- epilogue: O[q][head*64 + d] = O^T / l.  A lane holds 4-column pieces of its row (columns 32d + 8g + 4h ..+3); v_permlane32_swap
  // pairs the pieces g and g+1 of the two lane halves into 8 consecutive columns: 16-B stores instead of 8-B ones (guides T21; the
  // fast kernels' epilogues do the same).  Every lane takes part in the swaps; only the stores are masked.
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    const float inv = 1.0f / xhalf_sum(l_run[b]);
    u16* op = (u16*)p.O + ((size_t)(seq_row0 + (q_valid[b] ? qrow[b] : 0)) * p.H + head) * 64;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        uint32_t a0 = pack2(ot[b][d][4 * g + 0] * inv, ot[b][d][4 * g + 1] * inv, T);
        uint32_t a1 = pack2(ot[b][d][4 * g + 2] * inv, ot[b][d][4 * g + 3] * inv, T);
        uint32_t b0 = pack2(ot[b][d][4 * g + 4] * inv, ot[b][d][4 * g + 5] * inv, T);
        uint32_t b1 = pack2(ot[b][d][4 * g + 6] * inv, ot[b][d][4 * g + 7] * inv, T);
        // swap(a, b): lanes 32-63 of a <-> lanes 0-31 of b.  After: h = 0 {a, b} = columns 8g..8g+7, h = 1 {a, b} = 8g+8..8g+15
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a0), "+v"(b0));
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a1), "+v"(b1));
        if (q_valid[b]) *(uint4*)(op + 32 * d + 8 * g + 8 * h) = make_uint4(a0, a1, b0, b1);
      }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Software-pipelined variant for long sequences (cross-view attention).
//
// Measured on the kernel above (tools/attn_exp.py (rounds 1-2; git history), 32 views): deleting the softmax VALU saves 18 %, deleting the K/V
// streaming + barrier 19 %, three quarters of the MFMAs 35 %, and one wave per SIMD is only 1.46x slower than two:
// a wave runs QK^T -> softmax -> PV strictly one after the other and its SIMD partner fills the gaps only by chance.
// Here the tile is processed as two 32-key halves, staggered: every half-step is ONE basic block that holds
//    PV of the half whose P was finished in the previous half-step            ( 8 MFMAs)
//    QK^T of the half that will be soft-maxed in the next half-step           (10 MFMAs, running max folded in)
//    exp2 / row-sum / 16-bit pack of the current half                          (VALU, no dependence on either)
// so the matrix pipe and the VALU of one wave overlap by construction.  The exp uses the running max the scores were
// computed with (it is already subtracted by the MFMA); only if a score exceeds it by more than 2^THRESH, if the max
// moved since the QK^T was issued, or on a masked tail tile, a wave-uniform branch at the END of the half-step redoes
// that half exactly (new max, rescale O and l, recompute P).  P up to 2^THRESH keeps full 16-bit relative precision.
// Iteration j needs {K tile j, V tile j-1}: both arrive by LDS-DMA (source-side permutation builds the swizzled K
// rows and the [4 key][32 d] blocked V image), one barrier per iteration, no staging registers.
template <int T, int MINW>
__global__ __launch_bounds__(256, MINW) void attn_sp_kernel(const WmAttnArgs p) {
  constexpr int QB = 2, QT = 256;
  constexpr float PBOUND = 2048.0f;  // 2^11: fine for bf16 and f16 P, and far from fp32 trouble in l and O
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_B];  // [set][K|V]
  typedef __attribute__((address_space(3))) void* lds_vp;
  typedef const __attribute__((address_space(1))) void* glb_vp;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, ql = lane & 31;
  const int tiles_per_seq = (p.seq_len + QT - 1) / QT;
  const int nseq = p.q_rows / p.seq_len;
  const int tiles_per_head = tiles_per_seq * nseq;
  const int lid = xcd_remap(blockIdx.x, tiles_per_head * p.H);
  const int head = lid / tiles_per_head;
  const int tile = lid - head * tiles_per_head;
  const int seq = tile / tiles_per_seq;
  const int qt = tile - seq * tiles_per_seq;
  const int seq_row0 = seq * p.seq_len;

  const u16* Qh = (const u16*)p.Q + (size_t)head * p.q_head_stride * 64;
  int qrow[QB];
  bool q_valid[QB];
  s16x8 qf[QB][4];
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    int r = qt * QT + (wave * QB + b) * 32 + ql;
    q_valid[b] = r < p.seq_len;
    r = q_valid[b] ? r : p.seq_len - 1;
    qrow[b] = r;
    const u16* qptr = Qh + (size_t)(seq_row0 + r) * 64;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[b][ks] = *(const s16x8*)(qptr + ks * 16 + h * 8);
  }

  const int seg_rows = p.kv_chunks > 1 ? p.kv_rows_per_chunk : p.seq_len;
  const int seg_off = p.kv_chunks > 1 ? 0 : seq_row0;
  const int ntpc = (seg_rows + KVB - 1) / KVB;
  const int nt = ntpc * p.kv_chunks;
  const u16* Kb = (const u16*)p.K + (size_t)head * p.kv_head_stride * 64 + (size_t)seg_off * 64;
  const u16* Vb = (const u16*)p.V + (size_t)head * p.kv_head_stride * 64 + (size_t)seg_off * 64;

  // ---- LDS-DMA of one set: waves 0,1 bring K tile `js` (4 pieces each), waves 2,3 bring V tile `js - 1`.
  // The loads are issued from inline asm: hipcc orders every later ds_read behind a pending LDS-DMA it can see
  // (s_waitcnt vmcnt(0) right after the issue: +2000 cycles per iteration, measured), and the ordering this kernel
  // needs is the explicit vmcnt(0) + barrier at the end of the iteration.
  int nxt_c = 0, nxt_j = wave < 2 ? 0 : -1;  // (chunk, tile in chunk) of the next tile this wave loads; calls come with js = 0, 1, 2, ...
  auto dma_set = [&](int js) {
    const int tl = wave < 2 ? js : js - 1;  // wave-uniform
    if (tl >= 0 && tl < nt) {
      const u16* src = (wave < 2 ? Kb : Vb) + (size_t)nxt_c * p.kv_chunk_stride;
      const uint32_t dst = (uint32_t)(size_t)(lds_vp)(smem + (js & 1) * 2 * TILE_B + (wave < 2 ? 0 : TILE_B) + (wave & 1) * 4096);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pc = (wave & 1) * 4 + i;
        int key, d8;
        if (wave < 2) {
          key = pc * 8 + (lane >> 3);
          d8 = (lane & 7) ^ ((key >> 1) & 7);
        } else {
          const int off = pc * 1024 + lane * 16, blk = off >> 8;
          key = (blk >> 1) * 4 + ((off >> 6) & 3);
          d8 = (blk & 1) * 4 + ((off >> 4) & 3);
        }
        int row = nxt_j * KVB + key;
        row = row < seg_rows ? row : seg_rows - 1;  // clamped keys are masked (K) / multiplied by P = 0 (V)
        const u16* g = src + (size_t)row * 64 + d8 * 8;
        wm_dma16(g, dst + i * 1024);
      }
    }
    if (tl >= -1) { if (++nxt_j == ntpc) { nxt_j = 0; ++nxt_c; } }
  };
  auto valid_keys = [&](int tl) { return seg_rows - (tl % ntpc) * KVB; };  // >= KVB except on a segment's last tile

  f32x16 ot[QB][2], st[QB][2];
  s16x8 pf[QB][2][2];
  float m_run[QB], l_run[QB];
  bool stale[2] = {false, false};  // wave-uniform: the running max moved after this half's QK^T was issued
  float dlt[QB][2];                // ... by this much (only read on the slow path)
  s16x8 qx[QB], kx;
#pragma unroll
  for (int j = 0; j < 8; ++j) kx[j] = 0;
  if (h == 0) kx[0] = (short)f2t<T>(1.0f);
#pragma unroll
  for (int b = 0; b < QB; ++b) {
#pragma unroll
    for (int j = 0; j < 8; ++j) qx[b][j] = 0;
    m_run[b] = 0.f; l_run[b] = 0.f; dlt[b][0] = dlt[b][1] = 0.f;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[b][d][r] = 0.f;
  }
  const int vtr_lane = ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;

  // K / V^T fragments of one half-step live in registers and are read from LDS ONE HALF-STEP AHEAD (a lone wave on
  // its SIMD has nobody to hide the ds_read latency behind; measured 8 exposed waits = ~650 cycles per half-step).
  struct Frags { s16x8 k[4]; s16x8 v[2][2]; };
  auto load_frags = [&](Frags& f, int hf, const char* set) {
    const char* kt = set;
    const char* vt = set + TILE_B;
    const int key = hf * 32 + ql;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f.k[ks] = *(const s16x8*)(kt + key * 128 + (((2 * ks + h) ^ ((key >> 1) & 7)) << 4));
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int key0 = hf * 32 + s2 * 16 + 4 * h;
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const char* b0 = vt + (((key0 >> 2) * 2 + d) << 8) + vtr_lane;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4p)(b0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4p)(b0 + 2 * 2 * 256));
        s16x8 vf;
        vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
        vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
        f.v[s2][d] = vf;
      }
    }
  };
  // S'[b][hf] = K(half hf) Q^T - m_run  (keys 32 hf + (r&3) + 8(r>>2) + 4h, query ql of block b)
  auto qk = [&](int hf, const Frags& f) {
    const f32x16 zero = {0};
#pragma unroll
    for (int b = 0; b < QB; ++b) { st[b][hf] = mfma32<T>(kx, qx[b], zero); dlt[b][hf] = 0.f; }
    stale[hf] = false;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int b = 0; b < QB; ++b) st[b][hf] = mfma32<T>(f.k[ks], qf[b][ks], st[b][hf]);
  };
  // O^T += V^T(half hf) P^T(half hf)
  auto pvmm = [&](int hf, const Frags& f) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int b = 0; b < QB; ++b) ot[b][d] = mfma32<T>(f.v[s2][d], pf[b][hf][s2], ot[b][d]);
  };
  auto pack_half = [&](int b, int hf, const float* pv) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      uint4 u;
      u.x = pack2(pv[8 * s2 + 0], pv[8 * s2 + 1], T);
      u.y = pack2(pv[8 * s2 + 2], pv[8 * s2 + 3], T);
      u.z = pack2(pv[8 * s2 + 4], pv[8 * s2 + 5], T);
      u.w = pack2(pv[8 * s2 + 6], pv[8 * s2 + 7], T);
      pf[b][hf][s2] = __builtin_bit_cast(s16x8, u);
    }
  };
  // fast path: P = 2^S' with the max S' was computed with; returns whether it may be kept (per lane).  No row max is
  // taken: the half's row-sum bounds every P of the row (sum <= 2^PBOUND => each P <= 2^PBOUND; inf / NaN fail the test)
  float lsum[QB];
  auto softmax_fast = [&](int hf) -> bool {
    bool ok = true;
#pragma unroll
    for (int b = 0; b < QB; ++b) {
      float pv[16], sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { pv[r] = __builtin_amdgcn_exp2f(st[b][hf][r]); sum += pv[r]; }
      pack_half(b, hf, pv);
      lsum[b] = sum;
      ok = ok && (sum <= PBOUND);
    }
    return ok;
  };
  // exact redo of one half: (masked) scores -> new running max -> rescale O, l -> P
  auto softmax_slow = [&](int hf, int valid, bool first) {
#pragma unroll
    for (int b = 0; b < QB; ++b) {
      const float delta = dlt[b][hf];  // the max moved after this half's QK^T was issued
      float sv[16], mloc = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = hf * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        sv[r] = key < valid ? st[b][hf][r] - delta : -INFINITY;
        mloc = fmaxf(mloc, sv[r]);
      }
      mloc = xhalf_max(mloc);  // vs the current m_run; valid >= 1 on every tile, so finite
      const float cand = m_run[b] + (first ? mloc : fmaxf(mloc, 0.f));
      const float m_new = ceil_t16<T>(cand);
      const float d2 = m_new - m_run[b];
      if (!first) {
        const float alpha = __builtin_amdgcn_exp2f(-d2);
        l_run[b] *= alpha;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) ot[b][d][r] *= alpha;
      }
      m_run[b] = m_new;
      dlt[b][hf ^ 1] += d2;  // the other half's scores (if already issued) were computed with the old max
      if (h == 0) qx[b][0] = (short)f2t<T>(-m_new);
      float pv[16], sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { pv[r] = __builtin_amdgcn_exp2f(sv[r] - d2); sum += pv[r]; }
      pack_half(b, hf, pv);
      lsum[b] = sum;
    }
  };
  auto softmax = [&](int hf, int valid, bool first) {
    bool ok = softmax_fast(hf);
    int slow = (first || valid < KVB || stale[hf]) ? 1 : 0;
    // opaque: neither may the decision be hoisted above the fast path nor the fast path sunk below the branch — either
    // would separate the softmax VALU from the MFMAs it is meant to overlap with
    asm volatile("" : "+v"(slow), "+v"(pf[0][hf][0]), "+v"(pf[0][hf][1]), "+v"(pf[1][hf][0]), "+v"(pf[1][hf][1]), "+v"(lsum[0]), "+v"(lsum[1]));
    if (!__all(ok && slow == 0)) { softmax_slow(hf, valid, first); stale[hf ^ 1] = true; }
#pragma unroll
    for (int b = 0; b < QB; ++b) l_run[b] += lsum[b];
  };
  // Schedule.  Set j = {K tile j, V tile j-1} lives in buffer j & 1.  Iteration j:
  //   half-step A(j): softmax(j-1, half 1) || PV(j-1, half 0) + QK(j, half 0)     while reading B(j)'s fragments (set j)
  //   vmcnt(0) [set j+1 landed], lgkmcnt(0) [own reads of set j retired], barrier
  //   issue the DMA of set j+2 into buffer j & 1  (set j is dead: its last fragments were read in A(j))
  //   half-step B(j): softmax(j, half 0)   || PV(j-1, half 1) + QK(j, half 1)     while reading A(j+1)'s fragments (set j+1)
  auto set_ptr = [&](int js) { return (const char*)smem + (js & 1) * 2 * TILE_B; };
  auto mid_barrier = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  Frags fa, fb;
  dma_set(0);
  mid_barrier();
  dma_set(1);
  // ---- iteration 0: A(0) = QK(0, half 0) only; B(0) = softmax(0, half 0) + QK(0, half 1)
  load_frags(fa, 0, set_ptr(0));   // only the K part is meaningful (V tile -1 does not exist; never multiplied)
  load_frags(fb, 1, set_ptr(0));
  qk(0, fa);
  mid_barrier();                   // set 1 landed
  dma_set(2);
  load_frags(fa, 0, set_ptr(1));
  softmax(0, valid_keys(0), true);
  qk(1, fb);
  for (int j = 1; j < nt; ++j) {
    const int vprev = valid_keys(j - 1), vcur = valid_keys(j);
    load_frags(fb, 1, set_ptr(j));
    pvmm(0, fa);
    qk(0, fa);
    softmax(1, vprev, false);
    mid_barrier();
    dma_set(j + 2);
    load_frags(fa, 0, set_ptr(j + 1));
    pvmm(1, fb);
    qk(1, fb);
    softmax(0, vcur, false);
  }
  // ---- iteration nt: softmax(nt-1, half 1) || PV(nt-1, half 0) ; PV(nt-1, half 1)   (set nt = {-, V tile nt-1})
  load_frags(fb, 1, set_ptr(nt));
  pvmm(0, fa);
  softmax(1, valid_keys(nt - 1), false);
  pvmm(1, fb);

#pragma unroll
  for (int b = 0; b < QB; ++b) {
    const float inv = 1.0f / xhalf_sum(l_run[b]);
    if (q_valid[b]) {
      u16* op = (u16*)p.O + ((size_t)(seq_row0 + qrow[b]) * p.H + head) * 64;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 u;
          u.x = pack2(ot[b][d][4 * g + 0] * inv, ot[b][d][4 * g + 1] * inv, T);
          u.y = pack2(ot[b][d][4 * g + 2] * inv, ot[b][d][4 * g + 3] * inv, T);
          *(uint2*)(op + 32 * d + 8 * g + 4 * h) = u;
        }
    }
  }
}

template <int T, int MINW>
hipError_t launch_sp(const WmAttnArgs& a, hipStream_t s) {
  const int tiles_per_seq = (a.seq_len + 255) / 256;
  const int nseq = a.q_rows / a.seq_len;
  hipLaunchKernelGGL((attn_sp_kernel<T, MINW>), dim3(tiles_per_seq * nseq * a.H), dim3(256), 0, s, a);
  return hipGetLastError();
}

// O[row][head*64 + d] = sum_s 2^(m_s - M) O_s / sum_s 2^(m_s - M) l_s for the split units [full_units, units); a block
// of 256 threads = 16 query rows of one unit x 16 groups of 4 channels (QT = query rows per unit)
template <int T>
__global__ __launch_bounds__(256) void attn_combine_kernel(const WmAttnArgs p, int QT) {
  const int bpu = QT / 16;  // blocks per unit
  const int unit = p.full_units + (int)(blockIdx.x / bpu);
  const int tiles_per_seq = (p.seq_len + QT - 1) / QT;
  const int tiles_per_head = tiles_per_seq * (p.q_rows / p.seq_len);
  const int head = unit / tiles_per_head, tile = unit - head * tiles_per_head;
  const int seq = tile / tiles_per_seq, qt = tile - seq * tiles_per_seq;
  const int r = qt * QT + (int)(blockIdx.x % bpu) * 16 + (threadIdx.x >> 4);
  if (r >= p.seq_len) return;
  const size_t row = (size_t)seq * p.seq_len + r;
  const int d4 = threadIdx.x & 15;
  float m[WM_ATTN_MAX_SPLITS], l[WM_ATTN_MAX_SPLITS], M = -INFINITY;
  for (int s = 0; s < p.kv_splits; ++s) {
    const float2 ml = *(const float2*)(p.part_ml + (((size_t)s * p.H + head) * p.q_rows + row) * 2);
    m[s] = ml.x; l[s] = ml.y;
    M = fmaxf(M, ml.x);
  }
  float L = 0.f;
  float4 o = make_float4(0, 0, 0, 0);
  for (int s = 0; s < p.kv_splits; ++s) {
    const float w = __builtin_amdgcn_exp2f(m[s] - M);
    L += w * l[s];
    const float4 v = *(const float4*)(p.part_o + ((size_t)s * p.q_rows + row) * (p.H * 64) + head * 64 + d4 * 4);
    o.x += w * v.x; o.y += w * v.y; o.z += w * v.z; o.w += w * v.w;
  }
  const float inv = 1.0f / L;
  uint2 u;
  u.x = pack2(o.x * inv, o.y * inv, T);
  u.y = pack2(o.z * inv, o.w * inv, T);
  *(uint2*)((u16*)p.O + (row * p.H + head) * 64 + d4 * 4) = u;
}

template <int T, int NW, int QB, int MINW, int LZ = 1, int STG = 1>
hipError_t launch(const WmAttnArgs& a_in, hipStream_t s, int fast = 0) {   // fast: 0 general kernel only, 3 attn_v3 first, 4 attn_v4 first
  const bool use_v3 = fast != 0;
  constexpr int QT = NW * 32 * QB;
  WmAttnArgs a = a_in;
  a.only_if = nullptr;
  const int tiles_per_seq = (a.seq_len + QT - 1) / QT;
  const int nseq = a.q_rows / a.seq_len;
  {  // split-KV choice: rounds over the resident slots, per unit of work; a split must beat 1 by > 6 % (combine cost)
    static const int ncu = [] { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&pr, d) == hipSuccess ? pr.multiProcessorCount : 256; }();
    const long slots = (long)ncu * MINW * 4 / NW, blocks = (long)tiles_per_seq * nseq * a.H;
    const int seg_rows = a.kv_chunks > 1 ? a.kv_rows_per_chunk : a.seq_len;
    const int ntiles = (seg_rows + KVB - 1) / KVB * a.kv_chunks;
    int lim = a.part_o && a.part_ml ? (a.max_splits > 0 ? a.max_splits : WM_ATTN_MAX_SPLITS) : 1;
    lim = lim < WM_ATTN_MAX_SPLITS ? lim : WM_ATTN_MAX_SPLITS;
    lim = lim < ntiles / 8 ? lim : (ntiles / 8 > 1 ? ntiles / 8 : 1);  // keep slices >= 8 tiles
    int best = 1;
    if (a.kv_splits > 0) best = a.kv_splits < lim ? a.kv_splits : lim;
    else if (wm_tuning[WM_TUNE_ATTN_SPLITS] > 0) best = wm_tuning[WM_TUNE_ATTN_SPLITS] < lim ? wm_tuning[WM_TUNE_ATTN_SPLITS] : lim;
    // Automatic split only on the view-sharded path (kv_chunks > 1: 8 local views of queries against all ranks' keys):
    // the q-tiles alone give 1.3 rounds of long blocks there and 4 slices measure +18 % at 8 chunks, +7 % at 4, +6 % at
    // 2 (tools/bench_attn_split_chunks.py).  On one GPU (kv_chunks == 1) the combine pass costs what the better
    // balance gains (8 views: 940 vs 940 TF/s) and the short per-frame sequences lose 20 %.
    else if (a.kv_chunks > 1 && lim >= 2) best = lim < 4 ? lim : 4;
    if (a.force_partial) best = a_in.kv_splits > 0 ? (a_in.kv_splits < lim ? a_in.kv_splits : lim) : 1;  // explicit uniform slices, no tail split
    a.kv_splits = best;
    a.full_units = 0;
    // Tail split: the launch's last, partly filled round — e.g. 688 units on 512 slots at 8 views: 176 whole blocks, one
    // per CU, with 80 CUs idle — is cut into S short blocks per unit, so that it fills the chip again.  It replaces the
    // uniform split wherever the launch spans more than one round: also on the view-sharded path (8 local views against
    // 2 / 4 / 8 gathered chunks: 1040 / 1051 / 1050 vs 1003 / 1030 / 1037 TF/s for the uniform 4-way split, with a combine
    // pass over a quarter of the rows, tools/bench_attn_split_chunks.py).
    // Model, in units of a full round: a partial round whose blocks sit alone on their CU runs 1.37x faster (measured:
    // one wave per SIMD is 1.46x slower per wave); a slice costs 1 / S; the combine pass ~0.01 per slice.  Measured
    // (tools/bench_attn_tail.py): 8 views 530 -> 480-495 us (+8-10 %), 16 views +8 %, 32 views +4-7 %; the short
    // per-frame sequences (22 key tiles) lose 10 % and are left alone (ntiles >= 64).
    const long tail = blocks % slots;
    // (the short per-frame sequences — 22 key tiles — lose 10 % with the general kernel and are left alone there; the pipelined
    //  kernel splits them too: 768 units on 512 slots = 512 whole + 256 x 2 halves = two even rounds)
    if (!a.force_partial && a_in.kv_splits == 0 && wm_tuning[WM_TUNE_ATTN_SPLITS] <= 0 && lim >= 2 && ntiles >= (use_v3 ? 16 : 64) && blocks > slots && tail > 0 && wm_tuning[WM_TUNE_ATTN_TAIL] != 0) {
      // (one block per CU — attn_v4 — has no partner to lose: a half-filled round costs a whole one)
      auto round_cost = [&](long n) { const long rem = n % slots; return (double)(n / slots) + (rem == 0 ? 0.0 : (rem * 2 <= slots && fast != 4) ? 0.73 : 1.0); };
      double best_cost = round_cost(tail);
      int bs = 1;
      for (int S = 2; S <= lim; ++S) {
        const double c = round_cost(tail * S) / S + 0.01 * S;
        if (c < best_cost - 0.05) { best_cost = c; bs = S; }
      }
      if (wm_tuning[WM_TUNE_ATTN_TAIL] > 1) bs = wm_tuning[WM_TUNE_ATTN_TAIL] < lim ? wm_tuning[WM_TUNE_ATTN_TAIL] : lim;  // A/B: forced slice count
      if (bs > 1) { a.kv_splits = bs; a.full_units = (int)(blocks - tail); }
      else if (best > 1) a.kv_splits = best;  // keep the uniform choice
    }
  }
  const int units = tiles_per_seq * nseq * a.H;
  const int nfull = a.kv_splits > 1 ? a.full_units : units;
  dim3 grid(nfull + (units - nfull) * a.kv_splits), block(NW * 64);
  if (use_v3) {  // same unit / split numbering (QT = 256 / 512): the fast kernel, then the general kernel on the blocks it flagged
    static const int v3_minw = [] { const char* e = wm_env("WM_ATTN_V3_MINW"); return e ? atoi(e) : 2; }();
    hipError_t e = fast == 4 ? wm_launch_attention_v4(a, (int)grid.x, a.unit_flags, s) : wm_launch_attention_v3(a, (int)grid.x, a.unit_flags, v3_minw, s);
    if (e != hipSuccess) return e;
    a.only_if = a.unit_flags;
  }
#ifdef WM_ATTN_TIMING_EXPERIMENT
  // (timing-only build for tools/attn_launch_cost.py, make EXTRA="-DWM_ATTN_TIMING_EXPERIMENT -DWM_DIAG_ENV": 1 skips the recompute pass behind
  //  a fast kernel, 2 the combine pass, 3 both — wrong results; not in the shipped library)
  static const int dbg_skip = [] { const char* e = wm_env("WM_ATTN_DEBUG_SKIP"); return e ? atoi(e) : 0; }();
#else
  constexpr int dbg_skip = 0;
#endif
  if (!(use_v3 && (dbg_skip & 1))) hipLaunchKernelGGL((attn_fwd_kernel<T, NW, QB, MINW, LZ, STG>), grid, block, 0, s, a);
  if (a.kv_splits > 1 && !a.force_partial && !(dbg_skip & 2))
    hipLaunchKernelGGL((attn_combine_kernel<T>), dim3((unsigned)((units - nfull) * (QT / 16))), dim3(256), 0, s, a, QT);
  return hipGetLastError();
}

}  // namespace

hipError_t wm_launch_attention_combine(const WmAttnArgs& a_in, int slots, hipStream_t s) {
  if (a_in.q_rows <= 0) return hipSuccess;
  if (slots < 1 || slots > WM_ATTN_MAX_SPLITS || !a_in.part_o || !a_in.part_ml) return hipErrorInvalidValue;
  constexpr int QT = 256;  // only the row -> (unit, block) arithmetic of the combine kernel depends on it
  WmAttnArgs a = a_in;
  a.kv_splits = slots;
  a.full_units = 0;
  const int units = ((a.seq_len + QT - 1) / QT) * (a.q_rows / a.seq_len) * a.H;
  if (a.dtype == WM_T_BF16) hipLaunchKernelGGL((attn_combine_kernel<WM_T_BF16>), dim3((unsigned)(units * (QT / 16))), dim3(256), 0, s, a, QT);
  else hipLaunchKernelGGL((attn_combine_kernel<WM_T_F16>), dim3((unsigned)(units * (QT / 16))), dim3(256), 0, s, a, QT);
  return hipGetLastError();
}

// Which kernel a launch takes (the attn_qb numbering of the tuning interface), or -1 for an invalid request:
//   7  attn_v3 (attention_v3.hip): bf16, software-pipelined, no running max, two waves per SIMD, 256-row units: the per-frame
//      sequences from 16 frames up
//   8  attn_v4 (attention_v4.hip): one wave per SIMD, 128 query rows per wave, 512-row units; bf16 (no max) and f16 (fixed
//      integer max in the QK chain's initial accumulator): every long (cross-view) sequence
//   3  general kernel (integer running max), 64 rows per wave at 2 waves / SIMD: everything else on long sequences, and every
//      piecewise (force_partial) launch that is not 7 / 8
//   4  general kernel, 32 rows per wave at 3 waves / SIMD: the short per-frame / DINO sequences
// 7 and 8 need a flag workspace and key segments of >= 512 keys (a ragged last tile is padded with zero keys); other values: A/B variants (forced only).
int wm_attention_variant(const WmAttnArgs& a) {
  static const int forced_env = [] { const char* e = wm_env("WM_ATTN_QB"); return e ? atoi(e) : 0; }();
  const int forced = wm_tuning[WM_TUNE_ATTN_QB] >= 0 ? wm_tuning[WM_TUNE_ATTN_QB] : forced_env;
  const int seg_rows = a.kv_chunks > 1 ? a.kv_rows_per_chunk : a.seq_len;
  const bool fast_ok = a.unit_flags != nullptr && seg_rows >= 512;   // (a ragged last tile is padded with zero keys)
  const bool v3ok = a.dtype == WM_T_BF16 && fast_ok;
  // long (cross-view) sequences: attn_v4 for both types (tools/bench_attn_v4.py, 32 views bf16: 1245-1250 TF/s vs attn_v3's
  // 1217-1227; 8 views: 1065-1103 vs 1051-1089; f16: 1186-1203 vs the general kernel's 1028-1039)
  int qb = forced ? forced : 8;
  // The short per-frame / DINO sequences (1376 / 1374 keys = 22 tiles per unit): the fast kernels' fixed cost per unit and their
  // coarser units do not pay below 16 sequences (8 frames: 85 us general (4) vs 87-93 attn_v3 vs 89-94 attn_v4; 32 frames,
  // bf16: 300-312 vs 265-297 attn_v3 (256-row units suit 1376 rows better than 512-row ones); f16: 321-337 vs 304-333 attn_v4,
  // left on 4)
  if (!forced && a.kv_chunks == 1 && a.seq_len <= 2048) qb = (a.dtype == WM_T_BF16 && a.q_rows / a.seq_len >= 16) ? 7 : 4;
  if ((qb == 7 && !v3ok) || (qb == 8 && !fast_ok) || qb == 0) qb = (a.kv_chunks == 1 && a.seq_len <= 2048) ? 4 : 3;
  // piecewise launches (the overlapped K/V gather of a sharded forward) write partials for every unit: kernels 3, 7 and 8 do;
  // a short or ragged local chunk (one view per rank: 1376 keys) would otherwise pick 4
  if (a.force_partial && qb != 7 && qb != 8 && qb != 3) qb = forced ? -1 : 3;
  return qb;
}
// query rows per unit and resident blocks per CU of the kernel a launch takes (for callers that size key slices)
void wm_attention_geometry(const WmAttnArgs& a, int* unit_rows, int* blocks_per_cu) {
  const int qb = wm_attention_variant(a);
  *unit_rows = qb == 8 ? 512 : qb == 4 || qb == 11 ? 128 : 256;
  *blocks_per_cu = qb == 8 ? 1 : qb == 4 || qb == 11 ? 3 : 2;
}

hipError_t wm_launch_attention(const WmAttnArgs& a, hipStream_t s) {
  if (a.q_rows <= 0) return hipSuccess;
  if (a.seq_len <= 0 || a.q_rows % a.seq_len != 0 || a.kv_chunks < 1) return hipErrorInvalidValue;
  // Which kernel runs is decided by wm_attention_variant (above): attn_v4 (8) on every long sequence with a flag workspace, attn_v3 (7)
  // on the per-frame sequences from 16 frames up (bf16), the general kernel at 32 rows per wave and 3 waves / SIMD (4) on the
  // short per-frame / DINO sequences below that, at 64 rows per wave and 2 waves / SIMD (3) otherwise.  The fast kernels flag the
  // units that left their range; the general kernel recomputes exactly those (and a unit's sticky hint, WmAttnArgs::unit_hint,
  // sends it there directly on the following calls).
  const int qb = wm_attention_variant(a);
  if (qb < 0) return hipErrorInvalidValue;
  const int seg_rows = a.kv_chunks > 1 ? a.kv_rows_per_chunk : a.seq_len;
  if (qb == 7) return launch<WM_T_BF16, 4, 2, 2>(a, s, 3);
  if (qb == 8) {  // attn_v4: 512-row units, one block per CU; the flagged units re-run on 8 waves x 64 rows of the general kernel
    if (!a.unit_flags) return hipErrorInvalidValue;
    return a.dtype == WM_T_BF16 ? launch<WM_T_BF16, 8, 2, 2>(a, s, 4) : launch<WM_T_F16, 8, 2, 2>(a, s, 4);
  }
  if (qb == 6) return a.dtype == WM_T_BF16 ? launch_sp<WM_T_BF16, 1>(a, s) : launch_sp<WM_T_F16, 1>(a, s);
  if (qb == 10) return a.dtype == WM_T_BF16 ? launch<WM_T_BF16, 4, 2, 2, 0>(a, s) : launch<WM_T_F16, 4, 2, 2, 0>(a, s);  // eager max (A/B)
  if (qb == 11) return a.dtype == WM_T_BF16 ? launch<WM_T_BF16, 4, 1, 3, 0>(a, s) : launch<WM_T_F16, 4, 1, 3, 0>(a, s);
  if (qb == 13) return a.dtype == WM_T_BF16 ? launch<WM_T_BF16, 4, 2, 2, 1, 2>(a, s) : launch<WM_T_F16, 4, 2, 2, 1, 2>(a, s);  // 3-deep DMA ring
  if (qb == 2) return a.dtype == WM_T_BF16 ? launch<WM_T_BF16, 4, 2, 1>(a, s) : launch<WM_T_F16, 4, 2, 1>(a, s);
  if (qb == 3) return a.dtype == WM_T_BF16 ? launch<WM_T_BF16, 4, 2, 2>(a, s) : launch<WM_T_F16, 4, 2, 2>(a, s);
  if (qb == 4) return a.dtype == WM_T_BF16 ? launch<WM_T_BF16, 4, 1, 3>(a, s) : launch<WM_T_F16, 4, 1, 3>(a, s);
  return a.dtype == WM_T_BF16 ? launch<WM_T_BF16, 4, 1, 2>(a, s) : launch<WM_T_F16, 4, 1, 2>(a, s);
}
