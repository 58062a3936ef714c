// Cross-view attention, software-pipelined inside one wave — gfx950, bf16, head_dim 64, long sequences.
//
// Replaces F.scaled_dot_product_attention (src/models/layers/attention.py:59) for the global blocks, where one
// sequence holds all views' tokens (L = N * 1376: 11 008 keys at 8 views, 44 032 at 32, 88 064 over 8 ranks).
//
// Why a second kernel: attn_fwd_kernel (attention.hip) runs QK^T -> softmax -> PV one after the other inside a wave
// and relies on its SIMD partner to fill the gaps (PMC r01: matrix pipe 48 % busy, 39 % of the wave cycles stalled at
// issue).  Here a wave overlaps the three by construction: a key tile is walked as two 32-key halves ("steps"), and
// step s issues, in ONE basic block whose instruction order is pinned with sched_group_barrier,
//      PV(s-2): O^T += V^T P^T of the half whose P was finished in step s-1            8 MFMA 32x32x16
//      QK(s)  : S^T = K Q^T of the half that is soft-maxed in step s+1                  8 MFMA
//      SM(s-1): P = 2^S, row sums, bf16 pack of the half whose S was finished in s-1    32 v_exp + 32 v_add + 16 v_cvt_pk
// i.e. per MFMA one gap holding 2 exp + 2 add + 1 pack (+ the LDS fragment reads of the next step).
//
// No running max.  Scores arrive in log2 units (q pre-scaled by log2(e)/sqrt(d)); bf16 has fp32's exponent range, so
// P = 2^S needs no max subtraction for correctness as long as nothing overflows or vanishes: 2^S / sum 2^S is the same
// number whatever power of two all P are scaled by, and the bf16 rounding of P is scale-invariant too — the result
// equals the integer-running-max kernel's up to fp32 summation order.  That is checked, not assumed: at the end every
// row's sum l must lie in [2^-80, 2^100] (q-k-normed scores sit within +-30); a unit with a row outside it raises its
// flag and the general kernel (running max, attention.hip) recomputes exactly the flagged units.  This removes the
// max MFMAs (4 of 36 per tile), the max search and every data-dependent branch from the loop.
//
// Order inside a step: the 8 QK MFMAs first, then the 8 PV MFMAs — S of step s is complete half a step before SM(s)
// reads it (no MFMA -> VALU result stall at a step's head), P of step s-1 a full step before PV reads it.  Fragments
// are read from LDS one half-step ahead of their MFMAs: V^T of this step's PV behind the QK MFMAs, K of the next
// step's QK behind the PV MFMAs.
// K/V tiles stream by LDS-DMA into rings (K 4 deep, V 3 deep: V of tile t is consumed one tile after K of tile t);
// at the barrier that opens tile t a wave requests K(t+3) and V(t+1): two tile times of flight for every piece, one
// barrier per tile, counted vmcnt.  f16 P would need the max (5-bit exponent): f16 stays on attention.hip.
// A key segment that is not a whole number of tiles (the per-frame sequences: 1376 = 21.5 tiles, DINO 1374; an odd number of
// views in a cross-view sequence or gathered chunk) gets its last tile PADDED WITH ZERO ROWS by the DMA: a zero key scores
// S = 0 exactly, P = 2^0 = 1 exactly, its zero V row adds nothing to O, and the epilogue takes the pads out of the row sums
// again — no masking, no second instantiation, the loop and its 236 registers untouched (round 2's masked instantiation
// needed one wave per SIMD and lost to the general kernel on exactly the sequences it was for).
#include "wm_common.h"
#include "wm_kernels.h"

#include <type_traits>

namespace {

constexpr int KVB = 64;
constexpr int TILE_B = KVB * 64 * 2;   // 8 KiB per K or V tile
constexpr int KRING = 4, VRING = 3;
constexpr int T = WM_T_BF16;

typedef __attribute__((address_space(3))) s16x4* lds_s16x4p;
typedef __attribute__((address_space(3))) void* lds_vp0;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

__device__ __forceinline__ uint32_t pack2bf(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// One 1-KiB LDS-DMA piece (16 B per lane): source = wave-uniform base (SGPR pair) + per-lane byte offset (VGPR);
// M0 carries the LDS destination (saved / restored: M0 is reserved).
__device__ __forceinline__ void dma16(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// The same piece with a per-lane 64-bit source pointer (ragged last tile: rows beyond the segment come from wm_zero_rows_v3)
__device__ __forceinline__ void dma16p(const void* g, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}
__device__ __attribute__((aligned(128))) const uint4 wm_zero_rows_v3[8] = {};

// sched_group_barrier masks (LLVM SchedGroupMask)
#ifdef WM_ATTN_STAMPS
__device__ unsigned long long wm_attn3_stamp_buf[4 * 8192];   // diagnostic build only (see attention_v4.hip)
#endif
#define SG_VALU 0x002
#define SG_MFMA 0x008
#define SG_DSRD 0x100
#define SG_TRANS 0x400

// (the second template parameter is kept for the launcher's A/B history: both values are the same kernel now)
template <int MINW, bool RAGGED>
__global__ __launch_bounds__(256, MINW) void attn_v3_kernel(const WmAttnArgs p, int* __restrict__ flags) {
  constexpr int QB = 2, QT = 256;
  __shared__ __attribute__((aligned(16))) char smem[(KRING + VRING) * TILE_B];  // K ring | V ring
  constexpr int VBASE = KRING * TILE_B;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (p.unit_hint && p.unit_hint[blockIdx.x] > 0) {   // block-uniform: see attention_v4.hip / WmAttnArgs::unit_hint
    if (tid == 0) flags[blockIdx.x] = 17;
    return;
  }
  const int h = lane >> 5, ql = lane & 31;
  const int tiles_per_seq = (p.seq_len + QT - 1) / QT;
  const int nseq = p.q_rows / p.seq_len;
  const int tiles_per_head = tiles_per_seq * nseq;
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

  // ---- Q fragments (B operand of S^T = K Q^T): lane (q = ql, h) holds Q[q][16 ks + 8 h + j]
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

  // ---- K/V segments; rem = valid keys of a segment's last tile (0: whole tiles)
  const int seg_rows = p.kv_chunks > 1 ? p.kv_rows_per_chunk : p.seq_len;
  const int seg_off = p.kv_chunks > 1 ? 0 : seq_row0;
  const int ntpc = (seg_rows + KVB - 1) / KVB;
  const int rem = seg_rows % KVB;
  const int ntiles = ntpc * p.kv_chunks;
  const u16* Kb = (const u16*)p.K + (size_t)head * p.kv_head_stride * 64 + (size_t)seg_off * 64;
  const u16* Vb = (const u16*)p.V + (size_t)head * p.kv_head_stride * 64 + (size_t)seg_off * 64;
  const int t0 = (int)((long long)split * ntiles / nsplit), t1 = (int)((long long)(split + 1) * ntiles / nsplit);
  const int nt = t1 - t0;  // >= 2 (launcher)
  const int npad = rem ? (t1 / ntpc - t0 / ntpc) * (KVB - rem) : 0;   // zero keys this block walks (one ragged tile per segment end in [t0, t1))

  // DMA: this wave moves pieces {2 wave, 2 wave + 1} of every K tile and of every V tile (source-side permutation builds the
  // XOR-swizzled K rows and the [4 key][32 d] blocked V image, as in attention.hip).  Source = a wave-uniform tile pointer
  // (advanced by one tile per request, by a chunk stride at a chunk's end: scalar arithmetic only) + fixed per-lane offsets.
  // K and V run on separate counters: V lags K by two tiles.
  uint32_t koff[2], voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pc = wave * 2 + i;
    const int kkey = pc * 8 + (lane >> 3), kd8 = (lane & 7) ^ ((kkey >> 1) & 7);
    const int off = pc * 1024 + lane * 16, blk = off >> 8;
    const int vkey = (blk >> 1) * 4 + ((off >> 6) & 3), vd8 = (blk & 1) * 4 + ((off >> 4) & 3);
    koff[i] = (uint32_t)(kkey * 64 + kd8 * 8) * 2;
    voff[i] = (uint32_t)(vkey * 64 + vd8 * 8) * 2;
  }
  const uint32_t smem_base = (uint32_t)(size_t)(lds_vp0)smem;
  const long long chunk_jump = (p.kv_chunk_stride - (long long)ntpc * KVB * 64) * 2;  // bytes from a chunk's end to the next chunk's start (whole tiles)
  const int c0 = t0 / ntpc, j0 = t0 - c0 * ntpc;
  const char* ksrc = (const char*)(Kb + (size_t)c0 * p.kv_chunk_stride + (size_t)j0 * KVB * 64);  // next K tile (wave-uniform)
  const char* vsrc = (const char*)(Vb + (size_t)c0 * p.kv_chunk_stride + (size_t)j0 * KVB * 64);
  int kj = j0, vj = j0;        // tile-in-chunk of the next K / V tile
  int kslot = 0, vslot = 0;    // ring slots of those tiles
  int kidx = 0, vidx = 0;      // index (within this block's range) of those tiles
  // ragged tile of a segment (wave-uniform, once per segment): per-lane pointers, rows >= rem from the zero rows
  auto pad_ptr = [&](bool isv, int i, const char* src) -> const char* {
    const int pc = wave * 2 + i;
    int key;
    if (!isv) key = pc * 8 + (lane >> 3);
    else { const int off = pc * 1024 + lane * 16, blk = off >> 8; key = (blk >> 1) * 4 + ((off >> 6) & 3); }
    return key < rem ? src + (isv ? voff[i] : koff[i]) : (const char*)wm_zero_rows_v3 + (lane & 7) * 16;
  };
  auto dma_k = [&]() {
    const uint32_t dst = smem_base + kslot * TILE_B + wave * 2048;
    if (rem && kj == ntpc - 1) {
      dma16p(pad_ptr(false, 0, ksrc), dst);
      dma16p(pad_ptr(false, 1, ksrc), dst + 1024);
    } else {
      dma16(ksrc, koff[0], dst);
      dma16(ksrc, koff[1], dst + 1024);
    }
    ksrc += TILE_B; ++kidx;
    if (++kj == ntpc) { kj = 0; ksrc += chunk_jump; }
    kslot = kslot == KRING - 1 ? 0 : kslot + 1;
  };
  auto dma_v = [&]() {
    const uint32_t dst = smem_base + VBASE + vslot * TILE_B + wave * 2048;
    if (rem && vj == ntpc - 1) {
      dma16p(pad_ptr(true, 0, vsrc), dst);
      dma16p(pad_ptr(true, 1, vsrc), dst + 1024);
    } else {
      dma16(vsrc, voff[0], dst);
      dma16(vsrc, voff[1], dst + 1024);
    }
    vsrc += TILE_B; ++vidx;
    if (++vj == ntpc) { vj = 0; vsrc += chunk_jump; }
    vslot = vslot == VRING - 1 ? 0 : vslot + 1;
  };

  f32x16 ot[QB][2];
  f32x16 st[2][QB];      // [step parity][q-block]: scores of one 32-key half
  s16x8 pf[2][QB][2];    // [step parity][q-block][s2]: packed P of one half
  float lsum[QB][2];
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    lsum[b][0] = lsum[b][1] = 0.f;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[b][d][r] = 0.f;
  }

  // Fragment addresses (LDS byte addresses).  K: row (32 hf + ql), 16-B chunk (2 ks + h) ^ swizzle(row); the swizzle of row
  // 32 + ql equals that of row ql, so half 1 = half 0 + 4096 (an immediate).  V^T: transposed 8-byte reads of the
  // [4 key][32 d] blocked image.  The ring-slot part is added OUTSIDE the scheduled regions (lds_k / lds_v below), so a
  // region holds exactly the instruction mix its sched_group_barrier pipeline names.
  typedef const __attribute__((address_space(3))) s16x8* lds_frag_p;
  uint32_t kaddr0[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kaddr0[ks] = smem_base + ql * 128 + (((2 * ks + h) ^ ((ql >> 1) & 7)) << 4);
  const uint32_t vaddr0 = smem_base + VBASE + ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8 + h * 512;

  s16x8 kfr[4], vfr[2][2];
  auto load_k = [&](const uint32_t (&ka)[4], int hf) {   // ka = kaddr0 + slot * TILE_B
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kfr[ks] = *(lds_frag_p)(uintptr_t)(ka[ks] + hf * 4096);
  };
  auto load_v = [&](uint32_t va, int hf) {                // va = vaddr0 + slot * TILE_B
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const uint32_t b0 = va + ((((hf * 8 + s2 * 4) * 2) + d) << 8);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4p)(uintptr_t)(b0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4p)(uintptr_t)(b0 + 2 * 2 * 256));
        s16x8 vf;
        vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
        vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
        vfr[s2][d] = vf;
      }
  };
  auto lds_k = [&](uint32_t (&ka)[4], int slot) {  // materialised here and now (the empty asm pins the values)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { ka[ks] = kaddr0[ks] + slot * TILE_B; asm volatile("" : "+v"(ka[ks])); }
  };
  auto lds_v = [&](int slot) { uint32_t va = vaddr0 + slot * TILE_B; asm volatile("" : "+v"(va)); return va; };
  auto qk = [&](int par) {  // S^T(half) = K(half) Q^T : st[par][b][r] = S[key 32 hf + (r&3) + 8(r>>2) + 4h][q = ql of block b]
    const f32x16 zero = {0};
#pragma unroll
    for (int b = 0; b < QB; ++b) st[par][b] = mfma32<T>(kfr[0], qf[b][0], zero);
#pragma unroll
    for (int ks = 1; ks < 4; ++ks)
#pragma unroll
      for (int b = 0; b < QB; ++b) st[par][b] = mfma32<T>(kfr[ks], qf[b][ks], st[par][b]);
  };
  auto pv = [&](int par) {  // O^T += V^T(half) P^T(half)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int b = 0; b < QB; ++b) ot[b][d] = mfma32<T>(vfr[s2][d], pf[par][b][s2], ot[b][d]);
  };
  auto sm_half = [&](int par, int b) {  // P = 2^S, row sums (two chains), bf16 pack — q-block b of one 32-key half
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float e[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        e[i] = __builtin_amdgcn_exp2f(st[par][b][8 * s2 + i]);
        lsum[b][i & 1] += e[i];
      }
      uint4 u;
      u.x = pack2bf(e[0], e[1]); u.y = pack2bf(e[2], e[3]); u.z = pack2bf(e[4], e[5]); u.w = pack2bf(e[6], e[7]);
      pf[par][b][s2] = __builtin_bit_cast(s16x8, u);
    }
  };
  auto sm = [&](int par) { sm_half(par, 0); sm_half(par, 1); };
  // A step is two scheduling regions of 8 MFMA gaps, each gap = 1 MFMA + (1 LDS read) + 2 exp + 3 plain VALU:
  //   region A: QK MFMAs | the 8 V^T fragment reads of THIS step's PV | softmax of q-block 0
  //   region B: PV MFMAs | the 4 K fragment reads of the NEXT step's QK | softmax of q-block 1
  // (one kind of LDS read per region, so the scheduler cannot pull the next step's K reads in front of this step's V reads)
  auto pipeline8 = [&](int nds) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
      if (i < nds) __builtin_amdgcn_sched_group_barrier(SG_DSRD, 1, 0);
      __builtin_amdgcn_sched_group_barrier(SG_TRANS, 2, 0);
      __builtin_amdgcn_sched_group_barrier(SG_VALU, 3, 0);
    }
  };

  // ---- prologue: K(0), V(0), K(1), K(2) requested; all but K(2) landed
  dma_k();
  dma_v();
  if (nt > 1) dma_k();
  if (nt > 2) { dma_k(); asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (nt > 3) dma_k();   // K(3)
  if (nt > 1) dma_v();   // V(1)
  __builtin_amdgcn_sched_barrier(0);

#ifdef WM_ATTN_STAMPS
  unsigned long long stamp_c0 = 0, stamp_r0 = 0;
  if (tid == 0) { stamp_c0 = __builtin_amdgcn_s_memtime(); stamp_r0 = __builtin_amdgcn_s_memrealtime(); }
  __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
  // tile 0: step 0 = QK(0); step 1 = QK(1) + SM(0)
  uint32_t ka[4], kb[4];
  lds_k(ka, 0);
  lds_k(kb, 1);
  load_k(ka, 0);
  qk(0);
  load_k(ka, 1);
  __builtin_amdgcn_sched_barrier(0);
  qk(1);
  sm(0);
  load_k(kb, 0);
  __builtin_amdgcn_sched_barrier(0);

  // ---- tiles 1 .. nt-1.  Tile j (K ring slot j & 3; V(j-1) in V ring slot (j-1) % 3):
  //   step 2j  : QK(2j)   [K half 0 of tile j, read in step 2j-1]   V^T reads of (tile j-1, half 0)   PV(2j-2)   SM(2j-1)   K reads of (tile j, half 1)
  //   step 2j+1: QK(2j+1) [K half 1 of tile j]                      V^T reads of (tile j-1, half 1)   PV(2j-1)   SM(2j)     K reads of (tile j+1, half 0)
  // barrier B_j opens step 2j: everybody finished step 2j-1, so K(j-1) and V(j-2) are dead = the slots K(j+3) and V(j+1) go to;
  // the wait leaves only the four youngest pieces (K(j+2), V(j), requested at B_{j-1}) in flight: K(j+1), V(j-1) landed.
  int vs = 0;  // V ring slot of tile j-1
  auto tile_body = [&](int j) {
    if (j + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (j + 3 < nt) dma_k();
    if (j + 1 < nt) dma_v();
    lds_k(ka, j & 3);
    lds_k(kb, (j + 1) & 3);
    const uint32_t va = lds_v(vs);
    __builtin_amdgcn_sched_barrier(0);
    // step 2j
    qk(0);
    load_v(va, 0);
    sm_half(1, 0);
    pipeline8(8);
    __builtin_amdgcn_sched_barrier(0);
    pv(0);
    load_k(ka, 1);
    sm_half(1, 1);
    pipeline8(4);
    __builtin_amdgcn_sched_barrier(0);
    // step 2j+1
    qk(1);
    load_v(va, 1);
    sm_half(0, 0);
    pipeline8(8);
    __builtin_amdgcn_sched_barrier(0);
    pv(1);
    load_k(kb, 0);   // past the last tile: a dead read of a valid slot (keeps the step branch-free)
    sm_half(0, 1);
    pipeline8(4);
    __builtin_amdgcn_sched_barrier(0);
    vs = vs == VRING - 1 ? 0 : vs + 1;
  };
  for (int j = 1; j < nt; ++j) tile_body(j);
  // ---- epilogue: V(nt-1) landed (the last waits were vmcnt(0) + barrier);
  //      step 2nt: PV(2nt-2) + SM(2nt-1); step 2nt+1: PV(2nt-1)
  {
    const uint32_t va = lds_v(vs);
    load_v(va, 0);
    pv(0);
    sm(1);
    __builtin_amdgcn_sched_barrier(0);
    load_v(va, 1);
    pv(1);
  }

#ifdef WM_ATTN_STAMPS
  if (tid == 0) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x < 8192) {
      unsigned long long* o = wm_attn3_stamp_buf + (size_t)blockIdx.x * 4;
      o[0] = c1 - stamp_c0; o[1] = r1 - stamp_r0; o[2] = (unsigned long long)nt; o[3] = stamp_r0;
    }
  }
#endif
  // ---- row sums; the no-max form is valid iff every row's sum is a comfortably normal number
  float l[QB];
  bool bad = false;
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    l[b] = xhalf_sum(lsum[b][0] + lsum[b][1]);
    bad = bad || !(l[b] >= 8.2718061e-25f && l[b] <= 1.2676506e30f);  // [2^-80, 2^100]; NaN fails
    if (npad) {   // the zero keys of a ragged tile contributed P = 1.0 each: take them out again; if the true sum is lost against them
      l[b] -= (float)npad;                                       // (below 2^-12 of the pads' sum), the general kernel recomputes the unit
      bad = bad || !(l[b] >= (float)npad * 2.44140625e-4f);
    }
  }
  const int any_bad = __any(bad) ? 1 : 0;  // wave-uniform
  __shared__ int bad_sh[4];
  if (lane == 0) bad_sh[wave] = any_bad;
  __syncthreads();
  if (tid == 0) {
    const int f = bad_sh[0] | bad_sh[1] | bad_sh[2] | bad_sh[3];
    flags[blockIdx.x] = f;
    if (f && p.unit_hint) p.unit_hint[blockIdx.x] = WM_ATTN_HINT_TTL + 1;
  }

  if (nsplit > 1 || p.force_partial) {  // unnormalised partial (running max 0): the combine pass finishes the softmax
    const int slot = p.part_slot0 + split;
#pragma unroll
    for (int b = 0; b < QB; ++b) {
      if (!q_valid[b]) continue;
      const size_t row = (size_t)(seq_row0 + qrow[b]);
      float* op = p.part_o + ((size_t)slot * p.q_rows + row) * (p.H * 64) + head * 64;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(float4*)(op + 32 * d + 8 * g + 4 * h) = make_float4(ot[b][d][4 * g], ot[b][d][4 * g + 1], ot[b][d][4 * g + 2], ot[b][d][4 * g + 3]);
      if (h == 0) *(float2*)(p.part_ml + (((size_t)slot * p.H + head) * p.q_rows + row) * 2) = make_float2(0.f, l[b]);
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    const float inv = 1.0f / l[b];
    if (q_valid[b]) {
      u16* op = (u16*)p.O + ((size_t)(seq_row0 + qrow[b]) * p.H + head) * 64;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 u;
          u.x = pack2bf(ot[b][d][4 * g + 0] * inv, ot[b][d][4 * g + 1] * inv);
          u.y = pack2bf(ot[b][d][4 * g + 2] * inv, ot[b][d][4 * g + 3] * inv);
          *(uint2*)(op + 32 * d + 8 * g + 4 * h) = u;
        }
    }
  }
}

}  // namespace

// grid / split decisions are the caller's (attention.hip: the same unit numbering as attn_fwd_kernel<.., 4, 2, ..>)
hipError_t wm_launch_attention_v3(const WmAttnArgs& a, int grid, int* flags, int minw, hipStream_t s) {
  const int seg_rows = a.kv_chunks > 1 ? a.kv_rows_per_chunk : a.seq_len;
  if (seg_rows < KVB) return hipErrorInvalidValue;
  if (minw >= 2) hipLaunchKernelGGL((attn_v3_kernel<2, false>), dim3(grid), dim3(256), 0, s, a, flags);
  else hipLaunchKernelGGL((attn_v3_kernel<1, false>), dim3(grid), dim3(256), 0, s, a, flags);
  return hipGetLastError();
}

#ifdef WM_ATTN_STAMPS
extern "C" int wm_debug_attn3_stamps(unsigned long long* host_out, int nblocks) {
  if (nblocks > 8192) nblocks = 8192;
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(wm_attn3_stamp_buf), (size_t)nblocks * 4 * sizeof(unsigned long long));
}
#endif
