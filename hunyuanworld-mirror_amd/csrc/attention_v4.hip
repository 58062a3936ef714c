// Cross-view attention, one wave per SIMD, 128 query rows per wave — gfx950, head_dim 64, bf16 and f16.
//
// Replaces F.scaled_dot_product_attention (src/models/layers/attention.py:59) for the global blocks (L = N * 1376 keys).
//
// attn_v3_kernel (attention_v3.hip) runs two 64-row waves per SIMD: PMC r02 showed 37.8 % of the wave cycles stalled at
// issue (the two waves contend for one matrix pipe and one VALU issue port) and 0.75 LDS fragment reads per MFMA.  Here a
// workgroup is four waves, ONE per SIMD (512-register budget), each owning 128 query rows (4 q-blocks of 32): a K or V^T
// fragment read from LDS feeds four MFMAs instead of two (0.375 reads per MFMA), a 64-key tile moved into LDS serves 512
// query rows instead of 256 (half the L2 -> LDS bytes per flop), and nothing contends with the wave for its SIMD.
//
// Schedule.  A key tile is walked as two 32-key "steps".  Step s is 32 MFMA gaps in 8 groups of 4; every gap also holds
// 2 v_exp + 2 v_add + 1 v_cvt_pk (the softmax of a quarter of a q-block's half) and at most 2 LDS reads:
//   G0  QK(s-1, b3)   SM(s-1, b0) first 16 keys
//   G1  PV(s-2, b3)   SM(s-1, b0) second 16 keys     K fragments of step s      (4 ds_read_b128)
//   G2  QK(s,   b0)   SM(s-1, b1)                    V^T fragments of step s-1  (8 ds_read_b64_tr_b16)
//   G3  PV(s-1, b0)   SM(s-1, b1)
//   G4  QK(s,   b1)   SM(s-1, b2)
//   G5  PV(s-1, b1)   SM(s-1, b2)
//   G6  QK(s,   b2)   SM(s-1, b3)
//   G7  PV(s-1, b2)   SM(s-1, b3)
// QK(s, b) overwrites the score registers of q-block b right after SM(s-1, b) has consumed them, PV(s-1, b) reads the packed P
// of q-block b a group after SM(s-1, b) wrote it and before SM(s, b) rewrites it: scores, P, K and V^T fragments are all
// SINGLE-buffered (64 + 32 + 16 + 16 registers; attn_v3 keeps two sets of scores and P for half as many rows), and every
// producer -> consumer distance is at least four MFMA gaps.
//
// bf16: no running max (as attn_v3: P = 2^S, bf16 has fp32's exponent range; row sums checked at the end, a unit with a row
// outside [2^-80, 2^100] raises its flag and the general kernel recomputes it).
// f16 (BASELINE config 5's dtype): P = 2^(S - m) needs a max so that P stays inside f16's range.  m is an INTEGER per query
// row (same mantissa of P whatever m is: the f16 rounding of P does not depend on it), carried as the INITIAL ACCUMULATOR of
// the QK chain (16 registers holding -m: the subtraction costs no instruction), taken once, from the block's first tile plus a
// headroom (F16_HEADROOM below): no rescale, no branch, nothing per tile.  A row that grows beyond the window overflows a P:
// O turns non-finite, the unit raises its flag and the general kernel (running max) recomputes it.
#include "wm_common.h"
#include "wm_kernels.h"

#include <type_traits>

// Timing-only diagnostic variants of the tile loop (stamps build only; the results are WRONG with any bit set): 1 no tile barrier,
// 2 no LDS fragment reads in the loop, 4 no K/V DMA in the loop, 8 no vmcnt wait in front of the barrier, 16 (LDS-DMA form) only
// wave 0 requests its pieces.  tools/attn_v4_diag.sh
#ifndef WM_V4_DIAG
#define WM_V4_DIAG 0
#endif
#ifndef WM_V4_STAGED
#define WM_V4_STAGED 0   // 1: (bf16) tiles 2.. travel global -> registers -> LDS (buffer_load_dwordx4, ds_write_b128 a tile later) instead of
#endif                   // by LDS-DMA.  Measured, not adopted: 47.2 vs 46.8 cycles per MFMA, 1 205 vs 1 217 TF/s at 32 views — a 64-lane
                         // 16-byte VMEM instruction holds a lone wave ~40 cycles whichever kind it is (profiles/r03_attention_ceiling.md)
#if WM_V4_DIAG && !defined(WM_ATTN_STAMPS)
#error "WM_V4_DIAG is for the stamps build only"
#endif

namespace {

constexpr int KVB = 64;
constexpr int TILE_B = KVB * 64 * 2;   // 8 KiB per K or V tile
constexpr int KRING = 3, VRING = 3;    // at barrier j: tile j is read, tile j+1 in flight, tile j+2 requested into the slot of tile j-1

typedef __attribute__((address_space(3))) s16x4* lds_s16x4p;
typedef __attribute__((address_space(3))) void* lds_vp0;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

template <int T>
__device__ __forceinline__ uint32_t pack2t(float a, float b) {
  const f32x2 v = {a, b};
  if constexpr (T == WM_T_BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
  else return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}

// Two 1-KiB LDS-DMA pieces of one tile (16 B per lane each): source = wave-uniform tile pointer (SGPR pair) + per-lane byte
// offsets (the source-side permutation builds the LDS image); destination = M0 (+ 1024 for the second piece: the
// instruction offset is added on BOTH sides, so voff1 is pre-biased by -1024).  M0 is reserved: saved and restored.
__device__ __forceinline__ void dma2(const void* sbase, uint32_t voff0, uint32_t voff1m, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %3\n\t"
      "global_load_lds_dwordx4 %2, %3 offset:1024\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep) : "v"(voff0), "v"(voff1m), "s"(sbase), "s"(lds_dst) : "memory");
}

// Every MFMA gap of the loop is ONE asm statement: the MFMA and the quarter-softmax beside it (2 exp, 2 add, 1 pack), hand-placed.
// Why asm: the scores must sit in arch VGPRs (the VALU reads them) while O (128 registers, touched by nothing but MFMAs until
// the epilogue) and the Q fragments (64, read-only) only fit in the ACCUMULATOR half of the 512-register file.  hipcc selects
// ONE form (AGPR or VGPR C/D) for all MFMA builtins of a function and with either choice copies through v_accvgpr_* inside
// the loop; and an MFMA statement alone is opaque to its scheduler, which then has to be held in place by dummy operands that
// cost copies of their own.  Here the "a" / "v" constraints place every operand, the compiler allocates the registers,
// issues the LDS fragment reads between the statements and owns every wait.
// Order inside a gap: exp, exp, MFMA, add, add, pack — a transcendental's result is not used by the next VALU instruction
// (one wait state is required: the MFMA provides it), and the two exps are the two wait states an MFMA needs behind a compiler
// instruction that wrote one of its operands just before the statement (a v_accvgpr_write of a Q fragment, a v_mov): hipcc pads
// nothing in front of a statement.  The bare forms (no softmax beside the MFMA: prologue, drain) open with s_nop 1 for the same
// reason — without it the f16 prologue read stale Q fragments on some rows of q-block 0 (caught by the op test).  What hipcc does not see are the MFMAs' result hazards: every reader of
// an MFMA result is >= 16 instructions behind it (the schedule in the header), the epilogue is fenced by s_nops, and the
// f16 prologue's reads of its first scores are fenced by s_nops too.
#define WM_SM_HEAD "v_exp_f32 %[e0], %[s0]\n\tv_exp_f32 %[e1], %[s1]\n\t"
#define WM_SM_TAIL_BF "\n\tv_add_f32 %[l0], %[l0], %[e0]\n\tv_add_f32 %[l1], %[l1], %[e1]\n\tv_cvt_pk_bf16_f32 %[p], %[e0], %[e1]"
#define WM_SM_TAIL_H "\n\tv_add_f32 %[l0], %[l0], %[e0]\n\tv_add_f32 %[l1], %[l1], %[e1]\n\tv_cvt_pk_f16_f32 %[p], %[e0], %[e1]"
#define WM_SM_OUT [l0] "+v"(l0), [l1] "+v"(l1), [p] "=v"(p), [e0] "+v"(e0), [e1] "+v"(e1)
#define WM_SM_IN [s0] "v"(s0), [s1] "v"(s1)
// O^T += V^T P^T (accumulator in AGPRs), bare and with the softmax quarter
template <int T> __device__ __forceinline__ void gap_pv(f32x16& acc, const s16x8& a, const s16x8& b) {
  if constexpr (T == WM_T_BF16) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]" : [c] "+a"(acc) : [a] "v"(a), [b] "v"(b));
  else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %[c], %[a], %[b], %[c]" : [c] "+a"(acc) : [a] "v"(a), [b] "v"(b));
}
template <int T> __device__ __forceinline__ void gap_pv(f32x16& acc, const s16x8& a, const s16x8& b, float s0, float s1, float& l0, float& l1, uint32_t& p, float& e0, float& e1) {
  if constexpr (T == WM_T_BF16)
    asm volatile(WM_SM_HEAD "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]" WM_SM_TAIL_BF : [c] "+a"(acc), WM_SM_OUT : [a] "v"(a), [b] "v"(b), WM_SM_IN);
  else
    asm volatile(WM_SM_HEAD "v_mfma_f32_32x32x16_f16 %[c], %[a], %[b], %[c]" WM_SM_TAIL_H : [c] "+a"(acc), WM_SM_OUT : [a] "v"(a), [b] "v"(b), WM_SM_IN);
}
// S^T = K Q^T: FIRST 1 = first MFMA of a chain with C = 0, 2 = with C = the -m tile (f16), 0 = accumulate.  Q fragments in AGPRs.
template <int T, int FIRST> __device__ __forceinline__ void gap_qk(f32x16& d, const s16x8& a, const s16x8& bq, const f32x16& c0) {
  if constexpr (FIRST == 1) {
    if constexpr (T == WM_T_BF16) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %[d], %[a], %[b], 0" : [d] "=&v"(d) : [a] "v"(a), [b] "a"(bq));
    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %[d], %[a], %[b], 0" : [d] "=&v"(d) : [a] "v"(a), [b] "a"(bq));
  } else if constexpr (FIRST == 2) {
    if constexpr (T == WM_T_BF16) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %[d], %[a], %[b], %[c]" : [d] "=&v"(d) : [a] "v"(a), [b] "a"(bq), [c] "v"(c0));
    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %[d], %[a], %[b], %[c]" : [d] "=&v"(d) : [a] "v"(a), [b] "a"(bq), [c] "v"(c0));
  } else {
    if constexpr (T == WM_T_BF16) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %[d], %[a], %[b], %[d]" : [d] "+v"(d) : [a] "v"(a), [b] "a"(bq));
    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %[d], %[a], %[b], %[d]" : [d] "+v"(d) : [a] "v"(a), [b] "a"(bq));
  }
}
template <int T, int FIRST> __device__ __forceinline__ void gap_qk(f32x16& d, const s16x8& a, const s16x8& bq, const f32x16& c0, float s0, float s1, float& l0, float& l1, uint32_t& p, float& e0, float& e1) {
  if constexpr (FIRST == 1) {
    if constexpr (T == WM_T_BF16) asm volatile(WM_SM_HEAD "v_mfma_f32_32x32x16_bf16 %[d], %[a], %[b], 0" WM_SM_TAIL_BF : [d] "=&v"(d), WM_SM_OUT : [a] "v"(a), [b] "a"(bq), WM_SM_IN);
    else asm volatile(WM_SM_HEAD "v_mfma_f32_32x32x16_f16 %[d], %[a], %[b], 0" WM_SM_TAIL_H : [d] "=&v"(d), WM_SM_OUT : [a] "v"(a), [b] "a"(bq), WM_SM_IN);
  } else if constexpr (FIRST == 2) {
    if constexpr (T == WM_T_BF16) asm volatile(WM_SM_HEAD "v_mfma_f32_32x32x16_bf16 %[d], %[a], %[b], %[c]" WM_SM_TAIL_BF : [d] "=&v"(d), WM_SM_OUT : [a] "v"(a), [b] "a"(bq), [c] "v"(c0), WM_SM_IN);
    else asm volatile(WM_SM_HEAD "v_mfma_f32_32x32x16_f16 %[d], %[a], %[b], %[c]" WM_SM_TAIL_H : [d] "=&v"(d), WM_SM_OUT : [a] "v"(a), [b] "a"(bq), [c] "v"(c0), WM_SM_IN);
  } else {
    if constexpr (T == WM_T_BF16) asm volatile(WM_SM_HEAD "v_mfma_f32_32x32x16_bf16 %[d], %[a], %[b], %[d]" WM_SM_TAIL_BF : [d] "+v"(d), WM_SM_OUT : [a] "v"(a), [b] "a"(bq), WM_SM_IN);
    else asm volatile(WM_SM_HEAD "v_mfma_f32_32x32x16_f16 %[d], %[a], %[b], %[d]" WM_SM_TAIL_H : [d] "+v"(d), WM_SM_OUT : [a] "v"(a), [b] "a"(bq), WM_SM_IN);
  }
}

// The four pieces of one tile (K: two, V: two) in one statement.  M0 is written without save / restore: nothing else in this
// kernel uses it (LDS instructions do not need M0 on gfx9+; tests/test_kernel_resources_cpu.py checks the listing), and the
// statement that reads M0 is the statement that writes it.
__device__ __forceinline__ void dma4(const void* kbase, const void* vbase, uint32_t k0, uint32_t k1m, uint32_t v0, uint32_t v1m, uint32_t kdst, uint32_t vdst) {
  asm volatile(
      "s_mov_b32 m0, %[kd]\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %[k0], %[kb]\n\t"
      "global_load_lds_dwordx4 %[k1], %[kb] offset:1024\n\t"
      "s_mov_b32 m0, %[vd]\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %[v0], %[vb]\n\t"
      "global_load_lds_dwordx4 %[v1], %[vb] offset:1024"
      :: [k0] "v"(k0), [k1] "v"(k1m), [v0] "v"(v0), [v1] "v"(v1m), [kb] "s"(kbase), [vb] "s"(vbase), [kd] "s"(kdst), [vd] "s"(vdst) : "memory");
}
// The same two pieces with per-lane 64-bit source pointers (the ragged last tile of a key segment: rows beyond the segment
// come from wm_zero_row).  The second pointer is pre-biased by -1024 like voff1m above.
__device__ __forceinline__ void dma2p(const void* g0, const void* g1m, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "global_load_lds_dwordx4 %2, off offset:1024\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep) : "v"(g0), "v"(g1m), "s"(lds_dst) : "memory");
}
// 128 B of zeros (+ 1 KiB in front: the -1024 bias of the second piece must stay inside the object)
__device__ __attribute__((aligned(128))) const uint4 wm_zero_rows_v4[72] = {};

#define SG_VALU 0x002
#define SG_MFMA 0x008
#define SG_DSRD 0x100
#define SG_TRANS 0x400

#ifdef WM_ATTN_STAMPS
// diagnostic build only: per block {s_memtime, s_memrealtime} before and after the tile loop (wave 0), read back by
// wm_debug_attn_stamps; no output depends on them
__device__ unsigned long long wm_attn_stamp_buf[4 * 8192];
__device__ unsigned long long wm_attn_stamp_buf2[2 * 8192];   // {s_memrealtime at the kernel's first instruction, after wave 0's last store}
#endif

// f16: P = 2^(S - m) is in f16's normal range for m - 14 <= S < m + 16.  m = ceil(first tile's row max) + 4 puts that window at
// [max0 - 10, max0 + 20): a row may still grow 2^20 (13.9 nats) above what its first 64 keys showed before a P overflows (-> the
// unit is flagged and recomputed), and a P below 2^-10 of the first tile's maximum keeps an absolute error of 2^-25.
constexpr float F16_HEADROOM = 4.0f;

// step parts (compile-time mask)
enum { P_QKB3 = 1, P_PVB3 = 2, P_QK = 4, P_SM = 8, P_PV = 16 };

template <int T>
__global__ __launch_bounds__(256, 1) void attn_v4_kernel(const WmAttnArgs p, int* __restrict__ flags) {
  constexpr int QB = 4, QT = 512;
  constexpr bool F16 = T == WM_T_F16;
  __shared__ __attribute__((aligned(16))) char smem[(KRING + VRING) * TILE_B];  // K ring | V ring
  constexpr int VBASE = KRING * TILE_B;

#ifdef WM_ATTN_STAMPS
  const unsigned long long stamp_entry = __builtin_amdgcn_s_memrealtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (p.unit_hint && p.unit_hint[blockIdx.x] > 0) {   // block-uniform: this unit left the fast range in a recent call — straight to the general kernel
    if (tid == 0) flags[blockIdx.x] = 17;            // (16: by hint; nothing in this kernel writes the hint before its flag write at the end)
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
  s16x8 qf[QB][4];
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    int r = qt * QT + (wave * QB + b) * 32 + ql;
    r = r < p.seq_len ? r : p.seq_len - 1;
    const u16* qptr = Qh + (size_t)(seq_row0 + r) * 64;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[b][ks] = *(const s16x8*)(qptr + ks * 16 + h * 8);
  }

  // Q goes to the accumulator half of the register file here, once (every use is an "a" operand), well before its first MFMA
#pragma unroll
  for (int b = 0; b < QB; ++b)
    asm volatile("" :: "a"(qf[b][0]), "a"(qf[b][1]), "a"(qf[b][2]), "a"(qf[b][3]));
  asm volatile("s_nop 3" ::: "memory");

  // ---- K/V segments.  A segment (a sequence, or one gathered chunk) need not be whole 64-key tiles: its last tile is PADDED WITH
  // ZERO ROWS by the DMA (rem = its valid keys, 0 = none ragged).  A zero key scores S = 0 exactly, its P is 2^0 (bf16) or
  // 2^-m (f16) exactly, its zero V row adds nothing to O: the loop runs unmasked and the epilogue takes the pads' P out of the
  // row sums again (a unit whose true sum is lost against them raises its flag).
  const int seg_rows = p.kv_chunks > 1 ? p.kv_rows_per_chunk : p.seq_len;
  const int seg_off = p.kv_chunks > 1 ? 0 : seq_row0;
  const int ntpc = (seg_rows + KVB - 1) / KVB;
  const int rem = seg_rows % KVB;
  const int ntiles = ntpc * p.kv_chunks;
  const u16* Kb = (const u16*)p.K + (size_t)head * p.kv_head_stride * 64 + (size_t)seg_off * 64;
  const u16* Vb = (const u16*)p.V + (size_t)head * p.kv_head_stride * 64 + (size_t)seg_off * 64;
  const int t0 = (int)((long long)split * ntiles / nsplit), t1 = (int)((long long)(split + 1) * ntiles / nsplit);
  const int nt = t1 - t0;  // >= 1
  const int npad = rem ? (t1 / ntpc - t0 / ntpc) * (KVB - rem) : 0;   // zero keys this block walks (one ragged tile per segment end in [t0, t1))

  // DMA: this wave moves pieces {2 wave, 2 wave + 1} of every K tile and of every V tile (source-side permutation: XOR-swizzled
  // K rows, [4 key][32 d] blocked V image, as attention.hip)
  uint32_t koff[2], voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pc = wave * 2 + i;
    const int kkey = pc * 8 + (lane >> 3), kd8 = (lane & 7) ^ ((kkey >> 1) & 7);
    const int off = pc * 1024 + lane * 16, blk = off >> 8;
    const int vkey = (blk >> 1) * 4 + ((off >> 6) & 3), vd8 = (blk & 1) * 4 + ((off >> 4) & 3);
    koff[i] = (uint32_t)(kkey * 64 + kd8 * 8) * 2 - (i ? 1024u : 0u);
    voff[i] = (uint32_t)(vkey * 64 + vd8 * 8) * 2 - (i ? 1024u : 0u);
  }
  const uint32_t smem_base = (uint32_t)(size_t)(lds_vp0)smem;
  const long long chunk_jump = (p.kv_chunk_stride - (long long)ntpc * KVB * 64) * 2;  // bytes from a chunk's end to the next chunk's start
  const int c0 = t0 / ntpc, j0 = t0 - c0 * ntpc;
  const char* ksrc = (const char*)(Kb + (size_t)c0 * p.kv_chunk_stride + (size_t)j0 * KVB * 64);  // next tile (wave-uniform)
  const char* vsrc = (const char*)(Vb + (size_t)c0 * p.kv_chunk_stride + (size_t)j0 * KVB * 64);
  // The request state points at one tile (wave-uniform, always an existing tile of [t0, t1)): dma_advance() moves it on by one
  int dleft = ntpc - j0;       // tiles left in its segment, itself included (the last one is the ragged one, if any)
  const int dleft_ragged = rem ? 1 : -1;   // value of dleft at which the tile is a ragged one (never, for whole tiles)
  const uint32_t wbase = __builtin_amdgcn_readfirstlane(smem_base + wave * 2048);   // this wave's share of a ring slot
  // staged form: the tile's segment as two raw-buffer descriptors of seg_rows * 128 bytes (rows beyond the segment read as zeros:
  // the hardware's range check — which counts the scalar offset, tools/micro/buffer_oob.hip — pads the ragged tile) + the tile's
  // byte offset in the segment
  constexpr bool STAGED = WM_V4_STAGED && !F16;   // f16: no registers left for the staged tile (its -m tiles hold 64)
  const char* segk = (const char*)(Kb + (size_t)c0 * p.kv_chunk_stride);
  const char* segv = (const char*)(Vb + (size_t)c0 * p.kv_chunk_stride);
  __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)segk, 0, seg_rows * 128, 0x00020000);
  __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)segv, 0, seg_rows * 128, 0x00020000);
  int nsoff = j0 * TILE_B;
  auto dma_advance = [&]() __attribute__((always_inline)) {
    ksrc += TILE_B; vsrc += TILE_B; nsoff += TILE_B;
    if (__builtin_expect(--dleft == 0, 0)) {   // next segment (the empty statement keeps this a branch: if-converted it is a dozen
      asm volatile("" ::: "memory");          // scalar instructions on every tile)
      dleft = ntpc; ksrc += chunk_jump; vsrc += chunk_jump;
      if constexpr (STAGED) {
        nsoff = 0; segk += p.kv_chunk_stride * 2; segv += p.kv_chunk_stride * 2;
        krs = __builtin_amdgcn_make_buffer_rsrc((void*)segk, 0, seg_rows * 128, 0x00020000);
        vrs = __builtin_amdgcn_make_buffer_rsrc((void*)segv, 0, seg_rows * 128, 0x00020000);
      }
    }
  };
  // K and V of the tile into ring slot SL by LDS-DMA (the rings run in lockstep): 4 pieces per wave, one statement; the slot is
  // static (the tile loop is unrolled over the ring), so the destinations are constants added to one SGPR
  auto dma_tile = [&](auto slot_c) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot_c)::value;
    if (dleft == dleft_ragged) {   // wave-uniform, once per segment: rows >= rem of this tile are zero rows
      const char* zero = (const char*)wm_zero_rows_v4 + 1024;
      const char* kp[2]; const char* vp[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int pc = wave * 2 + i;
        const int kkey = pc * 8 + (lane >> 3);
        const int off = pc * 1024 + lane * 16, blk = off >> 8;
        const int vkey = (blk >> 1) * 4 + ((off >> 6) & 3);
        kp[i] = (kkey < rem ? ksrc + koff[i] : zero + (lane & 7) * 16 - (i ? 1024 : 0));
        vp[i] = (vkey < rem ? vsrc + voff[i] : zero + (lane & 7) * 16 - (i ? 1024 : 0));
      }
      dma2p(kp[0], kp[1], wbase + SL * TILE_B);
      dma2p(vp[0], vp[1], wbase + VBASE + SL * TILE_B);
    } else {
      dma4(ksrc, vsrc, koff[0], koff[1], voff[0], voff[1], wbase + SL * TILE_B, wbase + VBASE + SL * TILE_B);
    }
  };
  // Staged form (bf16; tiles 2 ..): a global_load_lds_dwordx4 holds the issuing wave for ~57 cycles wherever in the tile it stands
  // (one wave per SIMD: nobody runs meanwhile; 3.6 of the loop's 46.7 cycles per MFMA, profiles/r03_attn_v4_diag.log); a
  // buffer_load_dwordx4 into registers and a ds_write_b128 a tile later do not.  Tile j's first step writes the staged tile j + 1
  // into ring slot (j + 1) % 3 (dead since barrier B_j-1) behind the gaps of G1, then requests tile j + 2 behind G3; the compiler
  // counts these loads itself (builtin loads, plain stores: no hand-counted vmcnt in the loop).  Same lane -> (source, LDS) mapping
  // as the DMA pieces.  Past the last tile the last tile is requested again and written to a dead slot: no branch around them.
  uint4 stg[4];                                   // K piece 0, K piece 1, V piece 0, V piece 1 of the staged tile
  char* const stg_dst = smem + wave * 2048 + lane * 16;
  int tile_j = 0;                                 // the tile the loop is at (for the group that requests tile_j + 2)
  auto stage_load = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t o = i < 2 ? koff[i] + (i ? 1024u : 0u) : voff[i - 2] + (i == 3 ? 1024u : 0u);
      stg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(i < 2 ? krs : vrs, (int)o, nsoff, 0));
    }
  };

  f32x16 ot[QB][2];
  f32x16 st[QB];         // scores of one 32-key step, per q-block
  s16x8 pf[QB][2];       // packed P of one step, per q-block and 16-key group
  float lsum[QB][4];
  float eA0 = 0.f, eA1 = 0.f, eB0 = 0.f, eB1 = 0.f;   // exp scratch of even / odd gaps (four scalars: as an array the compiler
                                                      // keeps one pair in a 64-bit register and pads a wait state in front of every statement using it)
  f32x16 cinit[QB];      // f16: -m broadcast (the QK chain's initial accumulator)
  float m_run[QB];
  float s_run[QB];   // f16: sum of the first tile's scores per row (its mean places the fixed row max when the tile holds a sink)
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    lsum[b][0] = lsum[b][1] = lsum[b][2] = lsum[b][3] = 0.f;
    m_run[b] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { cinit[b][r] = 0.f; st[b][r] = 0.f; }
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[b][d][r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[b][s2][j] = 0;
  }

  // Fragment addresses (LDS byte addresses), as attn_v3: K row (32 hf + ql), chunk (2 ks + h) ^ swizzle(row) — half 1 = half 0
  // + 4096; V^T: transposed 8-byte reads of the [4 key][32 d] blocked image.
  typedef const __attribute__((address_space(3))) s16x8* lds_frag_p;
  uint32_t kaddr0[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kaddr0[ks] = smem_base + ql * 128 + (((2 * ks + h) ^ ((ql >> 1) & 7)) << 4);
  const uint32_t vaddr0 = smem_base + VBASE + ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8 + h * 512;

  s16x8 kfr[4], kfn[4], vfr[2][2];
  s16x4 vnx[8];   // the next step's V^T fragments as they arrive: [2 (2 s2 + d) + part]
  auto read_k = [&](uint32_t a, int hf) { return *(lds_frag_p)(uintptr_t)(a + hf * 4096); };
  auto read_v = [&](uint32_t va, int hf, int r) {   // r = 2 (2 s2 + d) + part: one 8-byte transposed read
    const int s2 = r >> 2, d = (r >> 1) & 1, part = r & 1;
    const uint32_t b0 = va + ((((hf * 8 + s2 * 4) * 2) + d) << 8) + part * (2 * 2 * 256);
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4p)(uintptr_t)(b0));
  };
  auto v_next = [&]() {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      s16x8 vf;
      vf[0] = vnx[2 * f][0]; vf[1] = vnx[2 * f][1]; vf[2] = vnx[2 * f][2]; vf[3] = vnx[2 * f][3];
      vf[4] = vnx[2 * f + 1][0]; vf[5] = vnx[2 * f + 1][1]; vf[6] = vnx[2 * f + 1][2]; vf[7] = vnx[2 * f + 1][3];
      vfr[f >> 1][f & 1] = vf;
    }
  };
  auto k_next = [&]() {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kfr[ks] = kfn[ks];
  };
  // One group of four gaps.  MM: which MFMAs (0 none, 1 the QK chain of q-block b, 2 the PV of q-block b); SM: the softmax of 16
  // keys (s2) of q-block sb beside them; DS: LDS fragment reads issued behind the gaps, at most one per gap (they also separate
  // consecutive statements of a dependent MFMA chain, between which hipcc would otherwise pad a wait state): 0 none, 1 this
  // step's K fragments (4), 2 + r0: three of this step's V^T fragment reads from r0 (r0 = 0, 3), 8: the last two (6, 7).
  // The row sums run in four chains per q-block (two per gap, alternating) and the exp results go through two alternating
  // pairs of scratch registers (read-write operands, so that they stay distinct): consecutive statements share no register
  // one writes (hipcc pads a wait state between statements that do).
  // Softmax order: a q-block's 32 scores per lane are 2 x 4 packed words (s2 = 16-key group, w = word).  The first group of a
  // q-block (half 0) visits (s2, w) = (0,0) (1,0) (0,1) (1,1), the second (half 1) words 2 and 3: consecutive statements then
  // write words of DIFFERENT packed fragments.  (Words w and w+1 of one fragment are the halves of a 64-bit register pair: hipcc
  // treats the second half's definition as a read of the pair and pads a wait state behind the statement that wrote the first.)
  uint32_t pw[QB][2][4];
  // DM (staged form): 1 = the staged tile's four pieces are written to LDS, one behind each gap; 2 = the next tile is requested.
  auto group = [&](auto mm_c, auto sm_c, auto ds_c, auto b_c, auto sb_c, auto half_c, auto slot_c, int kh, auto dm_c) __attribute__((always_inline)) {
    constexpr int MM = decltype(mm_c)::value, DS = decltype(ds_c)::value, DM = decltype(dm_c)::value;
    constexpr int b = decltype(b_c)::value, sb = decltype(sb_c)::value, half = decltype(half_c)::value;   // (compile-time: register arrays)
    constexpr uint32_t KOFF = decltype(slot_c)::value * TILE_B;   // ring offsets fold into the ds_read offset fields
    const uint32_t va = vaddr0 + KOFF;
    constexpr bool SM = decltype(sm_c)::value;
    constexpr int LC = 2;             // four row-sum chains (alternating pairs)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int s2 = i & 1, w = 2 * half + (i >> 1);
      const float s0 = st[sb][8 * s2 + 2 * w], s1 = st[sb][8 * s2 + 2 * w + 1];
      uint32_t u = 0;
      if constexpr (MM == 1 && SM) {
        if (i == 0) gap_qk<T, F16 ? 2 : 1>(st[b], kfr[0], qf[b][0], cinit[b], s0, s1, lsum[sb][LC * (i & 1)], lsum[sb][LC * (i & 1) + 1], u, (i & 1) ? eB0 : eA0, (i & 1) ? eB1 : eA1);
        else gap_qk<T, 0>(st[b], kfr[i], qf[b][i], cinit[b], s0, s1, lsum[sb][LC * (i & 1)], lsum[sb][LC * (i & 1) + 1], u, (i & 1) ? eB0 : eA0, (i & 1) ? eB1 : eA1);
      } else if constexpr (MM == 1) {
        if (i == 0) gap_qk<T, F16 ? 2 : 1>(st[b], kfr[0], qf[b][0], cinit[b]);
        else gap_qk<T, 0>(st[b], kfr[i], qf[b][i], cinit[b]);
      } else if constexpr (MM == 2 && SM) {
        gap_pv<T>(ot[b][i & 1], vfr[i >> 1][i & 1], pf[b][i >> 1], s0, s1, lsum[sb][LC * (i & 1)], lsum[sb][LC * (i & 1) + 1], u, (i & 1) ? eB0 : eA0, (i & 1) ? eB1 : eA1);
      } else if constexpr (MM == 2) {
        gap_pv<T>(ot[b][i & 1], vfr[i >> 1][i & 1], pf[b][i >> 1]);
      } else if constexpr (SM) {   // prologue only: no MFMA beside the softmax
        const float e0 = __builtin_amdgcn_exp2f(s0), e1 = __builtin_amdgcn_exp2f(s1);
        lsum[sb][LC * (i & 1)] += e0; lsum[sb][LC * (i & 1) + 1] += e1;
        u = pack2t<T>(e0, e1);
      }
      if constexpr (SM) pw[sb][s2][w] = u;
      if constexpr (DM == 1)   // the staged tile's piece i -> ring slot (slot + 1) % 3
        *(uint4*)(stg_dst + ((decltype(slot_c)::value + 1) % 3) * TILE_B + (i >> 1) * VBASE + (i & 1) * 1024) = stg[i];
      if constexpr (DM == 2) { if (i == 0) { if (tile_j + 2 < nt) dma_advance(); stage_load(); } }
      if constexpr (!(WM_V4_DIAG & 2)) {
        if constexpr (DS == 1) kfn[i] = read_k(kaddr0[i] + KOFF, kh);
        if constexpr (DS == 2 || DS == 5) { if (i < 3) vnx[DS - 2 + i] = read_v(va, kh, DS - 2 + i); }
        if constexpr (DS == 8) { if (i < 2) vnx[6 + i] = read_v(va, kh, 6 + i); }
      }
    }
    if constexpr (SM) {
      if constexpr (half == 1) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) pf[sb][s2] = __builtin_bit_cast(s16x8, make_uint4(pw[sb][s2][0], pw[sb][s2][1], pw[sb][s2][2], pw[sb][s2][3]));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  // One step (the header's table) on the half kh of the tile in ring slot SL: this step's K fragments, and this step's V^T
  // fragments (same tile, same half), which are read during the QK groups into a second register set — the reads also separate the
  // statements of the QK chains — and become the operands of the next step's PV groups.
  auto step = [&](auto parts_c, auto slot_c, int kh, auto dma_c) __attribute__((always_inline)) {
    constexpr int PARTS = decltype(parts_c)::value;
    constexpr bool DMA = decltype(dma_c)::value != 0;
    using D0 = I0; using D1 = std::integral_constant<int, DMA ? 1 : 0>; using D2 = std::integral_constant<int, DMA ? 2 : 0>;
    using D3 = I0; using D4 = I0;
    using SMc = std::integral_constant<bool, (PARTS & P_SM) != 0>;
    constexpr bool QK = (PARTS & P_QK) != 0, PV = (PARTS & P_PV) != 0;
    using MQ = std::integral_constant<int, QK ? 1 : 0>; using MP = std::integral_constant<int, PV ? 2 : 0>;
    using B0 = I0; using B1 = I1; using B2 = I2; using B3 = std::integral_constant<int, 3>;
    group(std::integral_constant<int, (PARTS & P_QKB3) ? 1 : 0>{}, SMc{}, std::integral_constant<int, QK ? 1 : 0>{}, B3{}, B0{}, I0{}, slot_c, kh, D0{});   // G0
    group(std::integral_constant<int, (PARTS & P_PVB3) ? 2 : 0>{}, SMc{}, I0{}, B3{}, B0{}, I1{}, slot_c, kh, D1{});                                        // G1
    if constexpr (QK) k_next();
    if constexpr ((PARTS & (P_PV | P_PVB3)) != 0) v_next();
    group(MQ{}, SMc{}, std::integral_constant<int, QK ? 2 : 0>{}, B0{}, B1{}, I0{}, slot_c, kh, D0{});   // G2
    group(MP{}, SMc{}, I0{}, B0{}, B1{}, I1{}, slot_c, kh, D2{});                                        // G3
    group(MQ{}, SMc{}, std::integral_constant<int, QK ? 5 : 0>{}, B1{}, B2{}, I0{}, slot_c, kh, D0{});   // G4
    group(MP{}, SMc{}, I0{}, B1{}, B2{}, I1{}, slot_c, kh, D3{});                                        // G5
    group(MQ{}, SMc{}, std::integral_constant<int, QK ? 8 : 0>{}, B2{}, B3{}, I0{}, slot_c, kh, D0{});   // G6
    group(MP{}, SMc{}, I0{}, B2{}, B3{}, I1{}, slot_c, kh, D4{});                                        // G7
  };
  using PC_FIRST = std::integral_constant<int, P_QK>;
  using PC_SECOND = std::integral_constant<int, P_QKB3 | P_QK | P_SM | P_PV>;
  using PC_FULL = std::integral_constant<int, P_QKB3 | P_PVB3 | P_QK | P_SM | P_PV>;
  using PC_DRAIN1 = std::integral_constant<int, P_QKB3 | P_PVB3 | P_SM | P_PV>;
  using PC_DRAIN2 = std::integral_constant<int, P_PVB3>;

#ifdef WM_ATTN_STAMPS
  unsigned long long stamp_c0 = 0, stamp_r0 = 0;
#endif

  // ---- prologue: tiles 0 and 1 requested, tile 0 landed; then tile 2 requested
  dma_tile(I0{});
  if (nt > 1) { dma_advance(); dma_tile(I1{}); }
  if constexpr (STAGED) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tiles 0 and 1 (LDS-DMA) have landed; every later tile travels through registers
  } else {
    if (nt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (nt > 2) dma_advance();
  if constexpr (STAGED) stage_load();            // tile 2 (or the last tile again: written to a slot nobody reads)
  else { if (nt > 2) dma_tile(I2{}); }
  __builtin_amdgcn_sched_barrier(0);

  if constexpr (F16) {
    // m per query row = ceil(max over the first tile's 64 keys) + F16_HEADROOM, fixed for the whole key range of the block
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      // (the same asm MFMAs as the loop: Q is then used through "a" operands only and lives in AGPRs; an MFMA builtin here made
      // the compiler keep Q in 64 VGPRs and copy four registers into AGPRs in front of every QK statement of the loop)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kfr[ks] = read_k(kaddr0[ks], hf);
#pragma unroll
      for (int b = 0; b < QB; ++b) {
        gap_qk<T, 1>(st[b], kfr[0], qf[b][0], cinit[b]);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) gap_qk<T, 0>(st[b], kfr[ks], qf[b][ks], cinit[b]);
      }
      // MFMA result -> VALU read: the compiler does not see the asm MFMAs' latency, and it may move a register-only reader of
      // an earlier statement's result ABOVE a fence that does not name that result ("memory" orders loads and stores only): the
      // score tiles are operands of the fence.  (Without them the row max of q-block 0 was taken right behind its MFMA chain.)
      asm volatile("s_nop 15\n\ts_nop 7" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3]));
#pragma unroll
      for (int b = 0; b < QB; ++b) {
        float mx = st[b][0], sm = st[b][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) { mx = fmaxf(mx, st[b][r]); sm += st[b][r]; }
        m_run[b] = hf == 0 ? mx : fmaxf(m_run[b], mx);
        s_run[b] = hf == 0 ? sm : s_run[b] + sm;
      }
    }
#pragma unroll
    for (int b = 0; b < QB; ++b) {
      // m = max0 + headroom, unless the first tile shows a SINK (its maximum far above its mean): then the bulk of the row's keys sits
      // far below max0 and max0 + 4 would push their P = 2^(S - m) below f16's normal range (flushed in O, still counted in the fp32
      // row sum: with 44 K keys 18 log2 units under a sink that loses ~8 % of the row's mass from O — ADVICE r03).  m = mean0 + 14 puts
      // the bulk at 2^-14 and the sink at 2^(gap - 14); never below max0 - 12 (the first tile's own maximum must stay < 2^16).
      const float mx0 = xhalf_max(m_run[b]), mean0 = xhalf_sum(s_run[b]) * (1.0f / 64.0f);
      m_run[b] = ceilf(fmaxf(fminf(mx0 + F16_HEADROOM, mean0 + 14.0f), mx0 - 12.0f));
#pragma unroll
      for (int r = 0; r < 16; ++r) cinit[b][r] = -m_run[b];
    }
    // the -m tiles are MFMA operands (SrcC) of statements the compiler cannot see into: written here, fenced here
    asm volatile("s_nop 7" :: "v"(cinit[0]), "v"(cinit[1]), "v"(cinit[2]), "v"(cinit[3]) : "memory");
    __builtin_amdgcn_sched_barrier(0);
  }

#ifdef WM_ATTN_STAMPS
  if (tid == 0) { stamp_c0 = __builtin_amdgcn_s_memtime(); stamp_r0 = __builtin_amdgcn_s_memrealtime(); }
  __builtin_amdgcn_s_waitcnt(0xC07F);
#endif

#if WM_V4_DIAG & 2
#pragma unroll
  for (int i = 0; i < 4; ++i) kfn[i] = read_k(kaddr0[i], 0);
#pragma unroll
  for (int r = 0; r < 8; ++r) vnx[r] = read_v(vaddr0, 0, r);
#endif
  // steps 0 and 1 (tile 0, ring slot 0)
  step(PC_FIRST{}, I0{}, 0, I0{});
  step(PC_SECOND{}, I0{}, 1, I0{});

  // ---- tiles 1 .. nt-1, unrolled over the ring (tile j sits in slot j % 3: every LDS address of the loop is a constant).
  // Barrier B_j opens step 2j: every wave has finished step 2j-1, so tile j-1 is dead (its fragments are in registers) = the ring
  // slot tile j+2 goes to; the wait leaves only the four youngest pieces (tile j+1) in flight.
  auto tile_at = [&](auto slot_c, int j) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot_c)::value;
    tile_j = j;
    if constexpr (STAGED) {
      // (LDS operations of a wave complete in order and every step waits for its fragment reads: the ds_writes of the previous
      // tile's G1 are long done; the wait states that in the code)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (!(WM_V4_DIAG & 1)) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      step(PC_FULL{}, slot_c, 0, std::integral_constant<int, (WM_V4_DIAG & 4) ? 0 : 1>{});
    } else {
      // lgkmcnt(0): the V^T fragment reads of step 2j-1 (consumed only in step 2j) must have LEFT the ring slot before another wave's
      // DMA of tile j+2 may land in it — they are ~6 MFMA gaps old here, so the wait is free; it states what the timing implied
      if constexpr (!(WM_V4_DIAG & 8)) {
        if (j + 1 < nt) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if constexpr (!(WM_V4_DIAG & 1)) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(WM_V4_DIAG & 4)) {
        if (j + 2 < nt) {
          dma_advance();
          if constexpr ((WM_V4_DIAG & 16) != 0) { if (wave == 0) dma_tile(std::integral_constant<int, (SL + 2) % 3>{}); }   // one wave's pieces only
          else dma_tile(std::integral_constant<int, (SL + 2) % 3>{});
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      step(PC_FULL{}, slot_c, 0, I0{});
    }
    step(PC_FULL{}, slot_c, 1, I0{});
  };
  int j = 1;
  for (; j + 2 < nt; j += 3) { tile_at(I1{}, j); tile_at(I2{}, j + 1); tile_at(I0{}, j + 2); }
  if (j < nt) {
    tile_at(I1{}, j);
    if (j + 1 < nt) tile_at(I2{}, j + 1);
  }
  // ---- drain: step 2nt (softmax + PV of the last step), step 2nt+1 (its q-block 3)
  step(PC_DRAIN1{}, I0{}, 0, I0{});
  step(PC_DRAIN2{}, I0{}, 0, I0{});
  // MFMA result -> the epilogue's reads of O: the fence names the accumulators, so no read can be scheduled above it
  asm volatile("s_nop 15\n\ts_nop 15" : "+a"(ot[0][0]), "+a"(ot[0][1]), "+a"(ot[1][0]), "+a"(ot[1][1]), "+a"(ot[2][0]), "+a"(ot[2][1]), "+a"(ot[3][0]), "+a"(ot[3][1]));

#ifdef WM_ATTN_STAMPS
  if (tid == 0) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x < 8192) {
      unsigned long long* o = wm_attn_stamp_buf + (size_t)blockIdx.x * 4;
      o[0] = c1 - stamp_c0; o[1] = r1 - stamp_r0; o[2] = (unsigned long long)nt; o[3] = stamp_r0;
    }
  }
#endif

  // ---- row sums; the fast form is valid iff every row's sum is a comfortably normal number (and, f16, no P left the range)
  float l[QB];
  int bad = 0;   // reason bits (any non-zero value flags the unit): 4 an O value is not finite (f16: a P left the range),
                 // 8 a row sum is not a comfortably normal number
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    l[b] = xhalf_sum((lsum[b][0] + lsum[b][1]) + (lsum[b][2] + lsum[b][3]));
    if (npad) {   // take the zero keys' P (2^(0 - m) each, the same v_exp_f32 the loop evaluated) out of the sum again
      const float pads = (float)npad * (F16 ? __builtin_amdgcn_exp2f(-m_run[b]) : 1.0f);
      l[b] -= pads;
      // the true sum must not be lost against the pads': fp32 keeps it to 2^-24 of the pads' sum, so at a ratio of 2^-12 (bf16: 8 significant
      // bits out, 2^-12 relative is plenty) or 2^-8 (f16: 11 bits out — at 2^-12 the normaliser's cancellation error would be one f16 ulp
      // on every output of the row) the result is still right; below it the general kernel recomputes the unit
      bad |= !(l[b] >= pads * (F16 ? 3.90625e-3f : 2.44140625e-4f)) ? 8 : 0;
    }
    if constexpr (F16) {
      float t = 0.f;   // stays 0 iff every O value of the row is finite (x * 0 is NaN for inf and NaN)
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) t = fmaf(ot[b][d][r], 0.f, t);
      bad |= !(t == 0.f) ? 4 : 0;
      bad |= !(l[b] >= 1.0e-30f && l[b] <= 1.0e30f) ? 8 : 0;
    } else {
      bad |= !(l[b] >= 8.2718061e-25f && l[b] <= 1.2676506e30f) ? 8 : 0;  // [2^-80, 2^100]; NaN fails
    }
  }
  int any_bad = 0;  // wave-uniform
#pragma unroll
  for (int bit = 2; bit <= 8; bit <<= 1) any_bad |= __any((bad & bit) != 0) ? bit : 0;
  __shared__ int bad_sh[4];
  if (lane == 0) bad_sh[wave] = any_bad;
  __syncthreads();
  if (tid == 0) {
    const int f = bad_sh[0] | bad_sh[1] | bad_sh[2] | bad_sh[3];
    flags[blockIdx.x] = f ? (f | 1) : 0;
    if (f && p.unit_hint) p.unit_hint[blockIdx.x] = WM_ATTN_HINT_TTL + 1;   // (+1: the recompute pass of this very call counts it down once)
  }

  if (nsplit > 1 || p.force_partial) {  // unnormalised partial (running max m_run): the combine pass finishes the softmax
    const int slot = p.part_slot0 + split;
#pragma unroll
    for (int b = 0; b < QB; ++b) {
      const int r = qt * QT + (wave * QB + b) * 32 + ql;
      if (r >= p.seq_len) continue;
      const size_t row = (size_t)(seq_row0 + r);
      float* op = p.part_o + ((size_t)slot * p.q_rows + row) * (p.H * 64) + head * 64;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(float4*)(op + 32 * d + 8 * g + 4 * h) = make_float4(ot[b][d][4 * g], ot[b][d][4 * g + 1], ot[b][d][4 * g + 2], ot[b][d][4 * g + 3]);
      if (h == 0) *(float2*)(p.part_ml + (((size_t)slot * p.H + head) * p.q_rows + row) * 2) = make_float2(m_run[b], l[b]);
    }
    return;
  }
  // ---- O[q][head*64 + d] = O^T / l.  A lane holds 4-column pieces of its row (columns 32d + 8g + 4h ..+3); v_permlane32_swap
  // pairs the pieces g and g+1 of the two lane halves into 8 consecutive columns: 16-B stores (guides T21).
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    const float inv = 1.0f / l[b];
    const int r = qt * QT + (wave * QB + b) * 32 + ql;
    const bool ok = r < p.seq_len;
    u16* op = (u16*)p.O + ((size_t)(seq_row0 + (ok ? r : 0)) * p.H + head) * 64;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        uint32_t a0 = pack2t<T>(ot[b][d][4 * g + 0] * inv, ot[b][d][4 * g + 1] * inv);
        uint32_t a1 = pack2t<T>(ot[b][d][4 * g + 2] * inv, ot[b][d][4 * g + 3] * inv);
        uint32_t b0 = pack2t<T>(ot[b][d][4 * g + 4] * inv, ot[b][d][4 * g + 5] * inv);
        uint32_t b1 = pack2t<T>(ot[b][d][4 * g + 6] * inv, ot[b][d][4 * g + 7] * inv);
        // before: h = 0 lanes hold columns 8g..8g+3 (a) and 8g+8..8g+11 (b); h = 1 lanes 8g+4..8g+7 (a) and 8g+12..8g+15 (b).
        // swap(a, b): lanes 32-63 of a <-> lanes 0-31 of b.  After: h = 0 {a, b} = columns 8g..8g+7, h = 1 {a, b} = 8g+8..8g+15.
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a0), "+v"(b0));
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a1), "+v"(b1));
        if (ok) *(uint4*)(op + 32 * d + 8 * g + 8 * h) = make_uint4(a0, a1, b0, b1);
      }
  }
#ifdef WM_ATTN_STAMPS
  if (tid == 0 && blockIdx.x < 8192) {   // (the partial-writing blocks return above: their exit stamp stays 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wm_attn_stamp_buf2[2 * blockIdx.x] = stamp_entry; wm_attn_stamp_buf2[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

}  // namespace

// grid / split decisions are the caller's (attention.hip: the unit numbering of attn_fwd_kernel<.., 8, 2, ..>, 512-row units)
hipError_t wm_launch_attention_v4(const WmAttnArgs& a, int grid, int* flags, hipStream_t s) {
  const int seg_rows = a.kv_chunks > 1 ? a.kv_rows_per_chunk : a.seq_len;
  if (seg_rows < KVB) return hipErrorInvalidValue;
  if (a.dtype == WM_T_BF16) hipLaunchKernelGGL((attn_v4_kernel<WM_T_BF16>), dim3(grid), dim3(256), 0, s, a, flags);
  else hipLaunchKernelGGL((attn_v4_kernel<WM_T_F16>), dim3(grid), dim3(256), 0, s, a, flags);
  return hipGetLastError();
}

#ifdef WM_ATTN_STAMPS
extern "C" int wm_debug_attn_stamps(unsigned long long* host_out, int nblocks) {
  if (nblocks > 8192) nblocks = 8192;
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(wm_attn_stamp_buf), (size_t)nblocks * 4 * sizeof(unsigned long long));
}
extern "C" int wm_debug_attn_stamps_clear() {   // before the launch whose stamps are read: entries of blocks it does not have stay 0
  void* p0 = nullptr; void* p1 = nullptr;
  if (hipGetSymbolAddress(&p0, HIP_SYMBOL(wm_attn_stamp_buf)) != hipSuccess || hipGetSymbolAddress(&p1, HIP_SYMBOL(wm_attn_stamp_buf2)) != hipSuccess) return 1;
  return (int)hipMemset(p0, 0, sizeof(unsigned long long) * 4 * 8192) | (int)hipMemset(p1, 0, sizeof(unsigned long long) * 2 * 8192);
}
extern "C" int wm_debug_attn_stamps2(unsigned long long* host_out, int nblocks) {
  if (nblocks > 8192) nblocks = 8192;
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(wm_attn_stamp_buf2), (size_t)nblocks * 2 * sizeof(unsigned long long));
}
#endif
