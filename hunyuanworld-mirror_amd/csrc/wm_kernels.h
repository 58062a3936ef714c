// Internal launcher interface between the host orchestrator (wm_model.cpp) and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

// The shipped library reads NO environment variable: kernel variants are chosen by the launchers and, for tests and A/B tools,
// through wm_set_tuning.  The historical WM_* environment switches only exist in diagnostic builds (make EXTRA=-DWM_DIAG_ENV).
#include <cstdlib>
inline const char* wm_env(const char* name) {
#ifdef WM_DIAG_ENV
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}


// ------------------------------------------------------------------ GEMM (gemm.hip)
enum {
  WM_EPI_F32 = 0,         // C f32 = acc + bias
  WM_EPI_T16 = 1,         // C 16-bit = acc + bias
  WM_EPI_GELU_T16 = 2,    // C 16-bit = gelu_erf(acc + bias)
  WM_EPI_RESID = 3,       // C f32 += gamma[col] * (acc + bias)           (LayerScale + residual)
  WM_EPI_ROWMAP_ADD = 4,  // C f32 [row-remapped] (+)= acc + bias + add[row % rpg][col]
  WM_EPI_CONVT = 5,       // k==stride ConvTranspose2d pixel-shuffle scatter, NHWC f32 out
  WM_EPI_QKV = 6,         // N = 3*D: bias, per-head LayerNorm(64) on q/k, 2-D RoPE, q scale, 16-bit [H][M][64] q/k/v
  WM_EPI_CONV = 7,        // 3x3 conv as a GEMM over (pixel rows) x (tap, channel) K (WmGemmArgs::cv_*): acc + bias + relu?(resid) + resid2, fp32 NHWC or 16-bit out
};

// qkv f32 [M][3*D] -> Q,K,V 16-bit [H][M][64] with optional per-head LayerNorm(64) and 2-D RoPE
struct WmQkvArgs {
  const float* qkv; void* q; void* k; void* v;
  const float* qn_w; const float* qn_b; const float* kn_w; const float* kn_b;  // null = no qk-norm
  const float* rope_cos; const float* rope_sin;                                  // [max_pos][16], null = no rope
  int M, H, head_stride;   // head_stride = rows per head in the outputs
  int tokens_per_view, patch_start, grid_w;  // position of token t: special (0,0) or (y+1, x+1)
  float inv_tpv, inv_gw;                     // 1/tokens_per_view, 1/grid_w (filled by wm_launch_gemm: the epilogue divides by multiplication)
  float q_scale; int dtype;
};

struct WmGemmArgs {
  const void* A; const void* W; void* C;
  const float* bias; const float* gamma; const float* add;
  float* C2; int ldc2;                                 // WM_EPI_RESID: the updated value is also stored to C2[row*ldc2 + col] (tap halves, visual_transformer.py:337-339)
  int M, N, K, lda, ldw, ldc;
  int dtype, epi;
  int group_bands;                                     // set by wm_launch_gemm: row bands per L2 supertile (ping-pong kernel)
  int pf_c;                                            // set by wm_launch_gemm (WM_EPI_RESID, ping-pong v2): touch the old C tile's lines during the last two K-tiles
  int sched_bands, sched_units;                        // set by wm_launch_gemm (ping-pong v2): the M rows, counted in 16-row units (sched_units), are cut into
                                                       // sched_bands row bands of floor / ceil(units / bands) units each, so that bands x column tiles fills whole
                                                       // rounds of the CUs; 0 = bands of the kernel's full tile height
  int rows_per_group, out_group, out_off, accumulate, out16, relu;  // WM_EPI_ROWMAP_ADD (out16: C is 16-bit, no accumulate; relu before add)
  int ct_k, ct_cout, ct_gh, ct_gw;                     // WM_EPI_CONVT
  WmQkvArgs qkv;                                       // WM_EPI_QKV (qkv.qkv unused; qkv.H*64 = D)
  // WM_EPI_CONV (round 4): a 3x3 / stride 1 / pad 1 convolution of a 16-BIT NHWC tensor A = x[n][y][x][cv_cin] as the ping-pong GEMM itself:
  // row r = pixel (n, y, x) in raster order, K-tile kt = (tap = kt / chunks, 64-channel chunk = kt % chunks), W = [Cout][tap][cin] (the conv
  // weight layout: K-contiguous as it is).  The A piece of a K-tile is DMA-ed from the pixel shifted by the tap, or from cv_zero (>= 128 B of
  // zeros) when that pixel lies outside the image: no im2col, no halo staging, no conversion pass.  M = N H W, K = 9 cv_cin, lda unused.
  int cv_h, cv_w, cv_cin;                              // cv_h > 0 selects the mode
  const void* cv_zero;
  const float* cv_resid; const float* cv_resid2;       // fp32 [M][N] or null; cv_resid_relu: relu(resid) is added
  int cv_resid_relu;                                   // (out16 / relu above: 16-bit output, ReLU on the result)
  // Token-conv form of WM_EPI_CONV (tc_k > 0): Conv2d(3x3, pad 1, no bias) of ConvTranspose2d(kernel = stride = tc_k) of a token grid
  // (dense_head.py:57-66 resize_layers[0 / 1] -> :394-399 layer{1,2}_rn) composed into one block-sparse GEMM at the TOKEN resolution: the A rows
  // are tokens [n][cv_h][cv_w][cv_cin] (16-bit), column tile nt is the output phase (a, b) = (nt / tc_k, nt % tc_k) with N = tc_k^2 x 256 columns,
  // and its K runs over that phase's own neighbour tokens only: tc_list[nt] = count << 16 | (di + 1 | (dj + 1) << 2) << 4 idx; W row (nt 256 + co)
  // holds the phase's combined matrices (sum over the taps landing in a neighbour of W_rn[tap] W_ct[phase']) neighbour after neighbour (ldw = 4 cv_cin).
  // Output: C[((n cv_h + i) tc_k + a) (cv_w tc_k) + j tc_k + b][co] (ldc = 256), bias = the interior bias repeated per phase.
  int tc_k;
  unsigned tc_list[16];
  // WM_EPI_RESID with the FOLLOWING LayerNorm fused into the epilogue (round 4; block.py:44,61 behind :90-92): when ln_out is set and
  // wm_gemm_fuses_ln(args) holds (N = 1024 = four column tiles, every block of the launch resident at once), the epilogue keeps the new
  // residual values in registers, the four column tiles of a row band exchange per-row (mean, M2) partials through ln_stats, and every
  // block writes LayerNorm(x) * ln_w + ln_b for its own 256 columns as a 16-bit tensor — the LayerNorm kernel's second read of the
  // stream is gone.  The rendezvous is a bounded spin; a block that gives up sets ln_fallback[band] and wm_launch_gemm_ln_fallback (always
  // launched behind, normally a no-op) redoes such bands from X and the partials with the same arithmetic.
  void* ln_out; int ln_ld;                             // 16-bit [M][ln_ld] (the GEMM's operand type)
  const float* ln_w; const float* ln_b; float ln_eps;
  float* ln_stats;                                     // [M][4][2] fp32 (mean, M2) of the row over each column tile
  int* ln_sync;                                        // [bands][2] arrivals / departures of a band's blocks, zero between launches
  int* ln_fallback;                                    // [bands]
};
hipError_t wm_launch_gemm(const WmGemmArgs& a, hipStream_t s);
// whether wm_launch_gemm(a) would take the fused-LayerNorm epilogue (a.ln_out set, shape / residency conditions): the caller skips its LayerNorm launch
bool wm_gemm_fuses_ln(const WmGemmArgs& a);
// behind a fused launch, on the same stream: recomputes the bands whose rendezvous timed out (grid = the launch's row bands)
hipError_t wm_launch_gemm_ln_fallback(const WmGemmArgs& a, hipStream_t s);

// ------------------------------------------------------------------ attention (attention.hip)
#define WM_ATTN_MAX_SPLITS_C 8
struct WmAttnArgs {
  const void* Q; const void* K; const void* V;  // 16-bit [H][rows][64]; q pre-scaled by log2(e)/sqrt(64): P = 2^(q.k - max)
  void* O;                                      // 16-bit [q_rows][H*64] token-major
  int H;
  int q_rows;          // total query rows (all sequences)
  int seq_len;         // rows per sequence; query row r attends keys of sequence r / seq_len
  int q_head_stride;   // rows per head in Q
  int kv_head_stride;  // rows per head in one K/V chunk
  int kv_chunks;       // global attention over gathered shards: K/V = kv_chunks x [H][kv_head_stride][64]
  long long kv_chunk_stride;  // elements between chunks
  int kv_rows_per_chunk;      // valid rows per head in each chunk (when kv_chunks > 1)
  int dtype;
  // split-KV (fills the chip when q-tiles x heads does not): each of kv_splits blocks per q-tile walks a slice of
  // the key tiles and writes an unnormalised partial; wm_launch_attention runs the combine pass itself.
  // kv_splits 0 = choose (1 when no workspace is given); part_o: fp32 [kv_splits][q_rows][H*64]; part_ml: fp32 [kv_splits][H][q_rows][2]
  int kv_splits, max_splits;
  int full_units;      // set by the launcher: units (q-tile, head) processed whole; the rest is split kv_splits ways (0 = all split)
  float* part_o; float* part_ml;
  // software-pipelined no-max kernel for long bf16 sequences (attention_v3.hip): unit_flags = int[grid blocks] workspace in which
  // every block reports whether one of its rows left the no-max range; the general kernel then runs with only_if = unit_flags
  // and recomputes exactly those blocks.  unit_flags null = general kernel only.
  int* unit_flags;
  const int* only_if;  // set by the launcher
  // Sticky fallback hint (round 4), int[grid blocks] that PERSISTS across calls of the same call site (same shapes), or null: a unit
  // the fast kernel had to flag is remembered (hint = WM_ATTN_HINT_TTL); while its hint is > 0 the fast kernel's block exits at entry
  // (flag set) and only the general kernel computes the unit, which also counts the hint down — so a checkpoint whose scores leave the
  // no-max range on some units (attention sinks) pays the general kernel for those units, not fast + general, and the fast kernel
  // is tried again every WM_ATTN_HINT_TTL + 1 calls.  Correctness never depends on the hint's content: a flagged unit is always
  // recomputed by the general kernel.
  int* unit_hint;
  // one device counter per call site, or null: every unit the recompute pass takes adds 1 (never reset).  The host reads it back
  // asynchronously (a D2H copy at the end of the forward, no synchronisation) and, when a quarter or more of a site's units went to
  // the general kernel, launches ONLY the general kernel there for the next WM_ATTN_HINT_TTL calls: two partly filled launches
  // (fast kernel on the units in range, general kernel on the rest) cost up to 1.25 x the general kernel alone.
  int* unit_stat;
  // Key range processed piecewise (sharded forward with the K/V all-gather overlapped, wm_model.cpp): with force_partial every
  // unit — split or not — writes an unnormalised partial into slot part_slot0 + split (kv_splits = the explicit, uniform slice
  // count of THIS launch, >= 1), nothing is combined; wm_launch_attention_combine then finishes all units over all slots.
  int force_partial, part_slot0;
};
// finishes every unit (q_rows x H) over `slots` partial slots written by force_partial launches; O 16-bit [q_rows][H*64]
hipError_t wm_launch_attention_combine(const WmAttnArgs& a, int slots, hipStream_t s);
hipError_t wm_launch_attention(const WmAttnArgs& a, hipStream_t s);
int wm_attention_variant(const WmAttnArgs& a);                                            // which kernel the launch takes (attention.hip)
void wm_attention_geometry(const WmAttnArgs& a, int* unit_rows, int* blocks_per_cu);   // its unit size and residency
hipError_t wm_launch_attention_v3(const WmAttnArgs& a, int grid, int* flags, int minw, hipStream_t s);
// one wave per SIMD, 128 query rows per wave, 512-row units (attention_v4.hip): bf16 (no max) and f16 (lazy integer max)
hipError_t wm_launch_attention_v4(const WmAttnArgs& a, int grid, int* flags, hipStream_t s);
// upper bound of the launch grid (units x splits) for sizing unit_flags
inline size_t wm_attention_max_blocks(int q_rows, int seq_len, int H) {
  const size_t nseq = (size_t)(q_rows / (seq_len > 0 ? seq_len : 1));
  return ((size_t)(seq_len + 127) / 128 + 1) * nseq * (size_t)H * WM_ATTN_MAX_SPLITS_C;
}
constexpr int WM_ATTN_MAX_SPLITS = 8;
constexpr int WM_ATTN_HINT_TTL = 15;

// ------------------------------------------------------------------ elementwise (elementwise.hip)
// LayerNorm over the last dim with row remapping: out row (g*out_group + out_off + q) <- in row
// (g*in_group + in_off + q), q < rows_per_group, g < groups.  Output 16-bit or f32.
struct WmLnArgs {
  const float* x; void* y; const float* w; const float* b;
  int D, ld_in, ld_out; float eps;
  int groups, rows_per_group, in_group, in_off, out_group, out_off;
  int out_f32, dtype;
};
hipError_t wm_launch_layernorm(const WmLnArgs& a, hipStream_t s);

hipError_t wm_launch_qkv_post(const WmQkvArgs& a, hipStream_t s);

// images f32 [N][C][H][W] -> im2col rows [N*gh*gw][Kpad] 16-bit, (x-mean)/std per channel when norm
hipError_t wm_launch_im2col(const float* img, void* out, int N, int C, int H, int W, int ps, int Kpad,
                            int normalize, int dtype, hipStream_t s);
// 7x7 pad-3 im2col of the NCHW image: [N*H*W][Kpad] 16-bit, col = c*49 + ky*7 + kx
hipError_t wm_launch_im2col7(const float* img, void* out, int N, int H, int W, int Kpad, int dtype, hipStream_t s);
// Gaussian-splat assembly per pixel from raw params [npix][12], image, gs_depth and predicted cameras [N][9]
hipError_t wm_launch_gs_splat(const float* gp, const float* img, const float* depth, const float* cam, float* means, float* quats,
                              float* scales, float* opac, float* sh, float* wts, int N, int H, int W, hipStream_t s);
// DINO tokens: X[n][0]=cls+pos[0]; X[n][1..R]=reg; X[n][1+R+j] = patch[n][j] + pos[1+j]   (f32)
hipError_t wm_launch_dino_tokens(const float* patch, const float* cls, const float* reg, const float* pos,
                                 float* X, int N, int hw, int R, int D, hipStream_t s);
// VGT special tokens: rows [0, psi) of every view: cam, reg x R, (pose, ray)
hipError_t wm_launch_vgt_special(float* X, const float* cam_tok, const float* reg_tok, const float* pose_tok,
                                 const float* ray_tok, int N, int P, int R, int D, int cond, int first_view_global,
                                 hipStream_t s);
// strided f32 copy (tap halves): dst[r*ld_dst + c] = src[r*ld_src + c]
hipError_t wm_launch_copy2d(const float* src, float* dst, int rows, int cols, int ld_src, int ld_dst, hipStream_t s);
// NHWC f32 bilinear resize, align_corners=True, optional separable add: c < C/2 ? addx[x][c] : addy[y][c - C/2]
hipError_t wm_launch_bilinear(const float* in, float* out, int N, int Hi, int Wi, int Ho, int Wo, int C,
                              const float* addx, const float* addy, hipStream_t s);
// out = a + b (f32), n elements
hipError_t wm_launch_add(const float* a, const float* b, float* out, size_t n, hipStream_t s);
// DPT tail: y32 f32 NHWC [n][H][W][32] (pre-ReLU) -> relu -> 1x1 (32->C) -> activation; writes attr [n][H][W][C-1], conf [n][H][W]
hipError_t wm_launch_depth_to_world(const float* depth, const float* ext, const float* intr, float* world, float* cam,
                                    unsigned char* mask, int B, int H, int W, float eps, hipStream_t s);
hipError_t wm_launch_resample_h(const unsigned char* in, unsigned char* out, int H, int Wi, int Wo, const int* bounds, const int* kk,
                                int ksize, hipStream_t s);
hipError_t wm_launch_resample_v_tensor(const unsigned char* tmp, float* out, int Hi, int W, int Ho, int Hf, int Wf, int ry0, int rx0,
                                       const int* bounds, const int* kk, int ksize, hipStream_t s);
size_t wm_confidence_mask_workspace(size_t n);
hipError_t wm_launch_confidence_mask(const float* conf, size_t n, unsigned int K, unsigned char* mask, void* workspace, hipStream_t s);
hipError_t wm_launch_dpt_tail(const float* y32, const float* w, const float* b, float* attr, float* conf,
                              size_t npix, int C, int act, hipStream_t s);
enum { WM_ACT_INV_LOG = 0, WM_ACT_EXP = 1, WM_ACT_NORM = 2 };

// ------------------------------------------------------------------ conv (conv.hip)
struct WmConvArgs {
  const float* x;      // NHWC f32 [N][Hi][Wi][Cin]
  const void* w;       // 16-bit [Cout][ky][kx][Cin]
  const float* bias;   // [Cout] or null
  const float* resid;  // NHWC f32 [N][Ho][Wo][Cout] or null; added after bias (relu'd first when resid_relu)
  const float* resid2; // second, plain residual or null
  float* y;            // NHWC f32 [N][Ho][Wo][Cout]
  int N, Hi, Wi, Cin, Ho, Wo, Cout, ksize, stride, pad;
  int relu_in, resid_relu, relu_out, dtype;
  // Fused align_corners bilinear resize of the INPUT (3x3 / s1 / p1 halo kernel only): when up_hs > 0, x is
  // [N][up_hs][up_ws][Cin] and the conv sees interpolate(x, (Hi, Wi)) (+ the separable position tables: channel
  // c < Cin/2 gets up_addx[ix][c], the others up_addy[iy][c - Cin/2]) without that tensor ever being stored.
  int up_hs, up_ws;
  const float* up_addx; const float* up_addy;
  int out16;           // y is a 16-bit NHWC tensor of the conv's operand type (f16 / bf16) and no fp32 is written: for outputs whose only consumer
                       // rounds them to that type anyway (x2 -> out_conv).  Register-staged 3x3 kernel only: ask wm_conv3x3_out16_ok first
  int in16;            // x is a 16-bit NHWC tensor of the conv's operand type (the out16 output of the conv before: RCU conv1 -> conv2); same kernels
                       // as out16 (wm_conv3x3_out16_ok), relu_in must be 0 (the producer applied it)
  int dbg;             // timing experiments only (builds with -DWM_CONV_TIMING_EXPERIMENT; results are wrong): 1 no halo refill, 2 no epilogue, 4 no weight refill
};
bool wm_conv3x3_applicable(const WmConvArgs& a);
bool wm_conv_force_generic();                     // WM_CONV_GENERIC set
bool wm_conv3x3_out16_ok(const WmConvArgs& a);   // the launch would take a kernel that implements out16
// 3x3 / s1 / p1 conv with 32 output channels on a 16-bit NHWC input (conv_n32.hip); zero: >= 16 B of device zeros
struct WmConvN32Args {
  const uint16_t* x; const uint16_t* w; const float* bias; float* y; const uint16_t* zero;
  int N, H, W, Cin, relu_out, dtype;
  // optional fused DPT tail (replaces y): ReLU -> 1x1 conv 32 -> tail_C (fp32 weights [tail_C][32], bias) -> activations (WM_ACT_*)
  const float* tail_w; const float* tail_b; float* tail_attr; float* tail_conf; int tail_C, tail_act;
};
hipError_t wm_launch_conv3x3_n32_in16(const WmConvN32Args& a, hipStream_t s);
// F.interpolate(bilinear, align_corners) (+ separable position tables) written as 16-bit NHWC (elementwise.hip)
hipError_t wm_launch_bilinear16(const float* in, void* out16, int N, int Hi, int Wi, int Ho, int Wo, int C, const float* addx,
                                const float* addy, int dtype, hipStream_t s);
hipError_t wm_launch_conv(const WmConvArgs& a, hipStream_t s);
// conv3x3(bilinear resize(x)) as nine low-resolution 1x1 products + a bilinear gather (upconv.hip): the weight's tap-major copy, and the gather
// over y16 [N][Hi][Wi][9][Co] (f16) -> out fp32 [N][Ho][Wo][Co] (+ bias); Co in {32, 64, 128}
hipError_t wm_launch_f32_to_16_2d(const float* src, int ld_src, void* dst16, int ld_dst, int rows, int cols, int dtype, hipStream_t s);
// token-conv (WmGemmArgs::tc_k): subtract, on the image border, the ConvTranspose bias seen through the 3x3 taps that fall outside; bmiss [9][F]
hipError_t wm_launch_tconv_border(float* out, const float* bmiss, int N, int H, int W, int F, hipStream_t s);
hipError_t wm_launch_repack_tap_major(const void* w16, void* wt16, int Co, int C, hipStream_t s);
hipError_t wm_launch_upconv_gather(const void* y16, const float* bias, float* out, int N, int Hi, int Wi, int Ho, int Wo, int Co, hipStream_t s);

// ------------------------------------------------------------------ camera head / small fp32 ops (small.hip)
// Y[M][N] = act(X[M][K]) * W[N][K]^T + b ; all f32.  pre_act: 0 none, 1 SiLU on X.  post: 0 none, 1 SiLU, 2 GELU(erf)
hipError_t wm_launch_linear_f32(const float* X, const float* W, const float* b, float* Y, int M, int N, int K,
                                int ldx, int ldy, int pre_act, int post_act, const float* gamma, int accumulate,
                                hipStream_t s);
// f32 attention over S tokens: qkv [S][3*D] -> out [S][D], heads x hd
hipError_t wm_launch_small_attention(const float* qkv, float* out, int S, int heads, int hd, hipStream_t s);
// h = gate * (LN_noaffine(tok) * (1 + scale) + shift) + tok ; mod = [S][3*D] (shift, scale, gate)
hipError_t wm_launch_adaln(const float* tok, const float* mod, float* h, int S, int D, float eps, hipStream_t s);
// camera activation + accumulate: pred[S][12] (+)= delta[S][12] ; out[S][9] = [t, quat, relu(fov)]
hipError_t wm_launch_cam_update(float* pred, const float* delta, float* out, int S, int first, hipStream_t s);
// camera_params [S][9] -> c2w [S][16], K [S][9]  (camera_utils.py:46-75, worldmirror.py:165-175)
hipError_t wm_launch_cam_matrices(const float* params, float* poses, float* intrs, int S, int H, int W, hipStream_t s);

// Process-wide tuning overrides (wm_set_tuning in the C ABI; tests and A/B tools).  -1 = not set: the kernel's
// launcher falls back to its environment variable, then to its built-in choice.
// ------------------------------------------------------------------ 3D-Gaussian rasteriser (raster.hip)
struct WmRasterArgs {
  const float* means; const float* quats; const float* scales; const float* opacities;  // [N,3] [N,4 wxyz] [N,3] [N]
  const float* colors; int is_sh;       // [N,3]: degree-0 SH coefficients (is_sh = 1) or final colours
  int N;
  const float* viewmats; const float* Ks; int C;  // world-to-camera [C,4,4], intrinsics [C,3,3]
  int width, height;
  float* out_rgb; float* out_depth; float* out_alpha;  // [C,H,W,3] [C,H,W] [C,H,W]
  int* radii_out;                       // optional [C,N,2] (tests)
  void* workspace; size_t workspace_bytes; size_t max_isects;
};
size_t wm_raster_workspace_bytes(int N, int C, int width, int height, size_t max_isects);
hipError_t wm_launch_rasterize(const WmRasterArgs& a, hipStream_t s, unsigned long long* n_isects_out);

// ------------------------------------------------------------------ voxel merge of splats (splat_prune.hip)
size_t wm_prune_workspace_bytes(size_t N);
hipError_t wm_launch_prune_gs(const float* means, const float* quats, const float* scales, const float* opac, const float* sh,
                              const float* weights, int N, float voxel, float* o_means, float* o_quats, float* o_scales, float* o_opac,
                              float* o_sh, int* K_out, void* workspace, size_t ws_bytes, hipStream_t s);

enum { WM_TUNE_GEMM_CFG = 0, WM_TUNE_GEMM_PP, WM_TUNE_GEMM_MFMA16, WM_TUNE_ATTN_QB, WM_TUNE_OP_LDPAD, WM_TUNE_ATTN_SPLITS, WM_TUNE_CONV_FUSE_UP, WM_TUNE_CONV_NARROW, WM_TUNE_CONV_BN, WM_TUNE_CONV_RS, WM_TUNE_LIN_MFMA, WM_TUNE_CONV_TPX, WM_TUNE_ATTN_TAIL, WM_TUNE_GEMM_GROUP, WM_TUNE_COMM_OVERLAP, WM_TUNE_HEADS_CONC, WM_TUNE_RCU_MID16, WM_TUNE_GEMM_SCHED, WM_TUNE_FORCE_GATHER, WM_TUNE_ATTN_OP_POLICY, WM_TUNE_COMM_P2P, WM_TUNE_LN_RPW, WM_TUNE_LN_FUSE, WM_TUNE_HEADS_MAIN, WM_TUNE_CONV_GEMM, WM_TUNE_RESID_PREFETCH, WM_TUNE_UP1_GATHER, WM_TUNE_TCONV, WM_TUNE_UP1_COMP, WM_TUNE_COUNT };
extern int wm_tuning[WM_TUNE_COUNT];

