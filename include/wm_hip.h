/* libwm_hip.so — C ABI of the MI355X-native WorldMirror forward pass.
 *
 * The reference (zubair-irshad/HunyuanWorld-Mirror) has no FFI/operator layer on this path: its
 * boundary is the Python class WorldMirror (src/models/models/worldmirror.py:16) whose forward()
 * (:120-152) is pure torch.nn.  This header is the boundary a maintainer binds instead (ctypes stub
 * in INTEGRATION.md); each entry cites the reference call it replaces.  Plain pointers and sizes
 * only; device pointers are HIP device memory owned by the caller (torch-ROCm tensors in practice).
 *
 * Threading: one handle per (process, device); calls on a handle are serialised by the caller and
 * are stream-ordered on the hipStream_t passed in (hipStream_t is passed as void*).
 */
#ifndef WM_HIP_H
#define WM_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wm_handle wm_handle;

typedef enum { WM_OK = 0, WM_ERR_INVALID = 1, WM_ERR_HIP = 2, WM_ERR_STATE = 3, WM_ERR_COMM = 4 } wm_status;
typedef enum { WM_DT_BF16 = 0, WM_DT_F16 = 1 } wm_dtype;

/* Architecture: reference ctor kwargs (worldmirror.py:17-34) + the sub-module defaults they imply
 * (visual_transformer.py:48-70, vision_transformer.py:364-375, camera_head.py:16-27,
 * dense_head.py:34-46). */
typedef struct {
  int32_t img_size, patch_size, embed_dim, gs_dim;
  int32_t enable_cond, enable_cam, enable_pts, enable_depth, enable_norm, enable_gs;
  int32_t depth, num_heads, mlp_ratio, num_register_tokens;
  int32_t intermediate_idxs[4];
  float rope_freq;
  int32_t dino_depth, dino_heads;
  int32_t cam_trunk_depth, cam_heads, cam_steps;
  int32_t dpt_features;
  int32_t dpt_out_channels[4];
  int32_t backbone_dtype; /* wm_dtype of GEMM/attention operands inside the backbone (reference GPU recipe: bf16) */
  int32_t head_dtype;     /* wm_dtype of DPT conv operands (fp32 accumulate/activations; default f16)          */
} wm_config;

/* Outputs of WorldMirror.forward (worldmirror.py:157-216), caller-allocated f32 device buffers for
 * the LOCAL views; any pointer may be NULL to skip that output (a NULL dense output skips its head). */
typedef struct {
  float* camera_params; /* [N_total][9]  (camera head attends across all views) */
  float* camera_poses;  /* [N_total][4][4] c2w */
  float* camera_intrs;  /* [N_total][3][3] */
  float* depth;         /* [N][H][W][1] */
  float* depth_conf;    /* [N][H][W]    */
  float* pts3d;         /* [N][H][W][3] */
  float* pts3d_conf;    /* [N][H][W]    */
  float* normals;       /* [N][H][W][3] */
  float* normals_conf;  /* [N][H][W]    */
  float* gs_depth;      /* [N][H][W][1] */
  float* gs_depth_conf; /* [N][H][W]    */
  /* Gaussian splats before voxel pruning (rasterization.py:389-498), one per pixel of the local views */
  float* splat_means;     /* [N][H][W][3] */
  float* splat_quats;     /* [N][H][W][4] */
  float* splat_scales;    /* [N][H][W][3] */
  float* splat_opacities; /* [N][H][W]    */
  float* splat_sh;        /* [N][H][W][3] (sh degree 0) */
  float* splat_weights;   /* [N][H][W]    */
  float* taps[4];       /* optional: the 4 backbone taps [N][P][2*D] (visual_transformer.py:337-339) */
} wm_outputs;

/* ---- lifecycle (replaces WorldMirror.__init__ / from_pretrained / load_state_dict) ---- */
wm_status wm_create(const wm_config* cfg, int device, wm_handle** out);
void wm_destroy(wm_handle* h);
const char* wm_last_error(const wm_handle* h);
/* One reference state_dict tensor (fp32, host memory). Unknown names are ignored (strict=False). */
wm_status wm_set_weight(wm_handle* h, const char* name, const float* host, const int64_t* shape, int ndim);
/* Number of spec'd parameters still missing is written to *missing.  strict=False semantics of the reference's loader
 * (PyTorchModelHubMixin, src/models/models/worldmirror.py:13,16): a missing tensor keeps its init value where that
 * value is deterministic — LayerNorm 1 / 0, LayerScale gamma 1.0 in the DINOv2 encoder and 0.01 elsewhere
 * (visual_transformer.py:65,152-160; camera_head.py:24) — the rest (randomly initialised in the reference) is 0.
 * wm_missing_name(h, i), i < *missing, lists them (NULL past the end). */
wm_status wm_finalize_weights(wm_handle* h, int* missing);
const char* wm_missing_name(const wm_handle* h, int i);
/* Several handles on ONE device (one per stream / per in-process rank) can share one copy of the repacked weights:
 * dst (same wm_config, same device) refers to src's device tensors without copying; src must outlive dst and must not
 * be re-loaded while dst exists.  The nn.Module analogue is several callers sharing one model's parameters. */
wm_status wm_share_weights(wm_handle* dst, const wm_handle* src);
/* DINO pos-embed resample for a non-native grid is done inside the library (host, once per shape);
 * exported for tests: in [gs*gs][D] -> out [gh*gw][D], bicubic antialias (vision_transformer.py:175-207). */
void wm_host_resample_pos(const float* in, int gs, int D, int gh, int gw, float* out);

/* ---- forward (replaces WorldMirror.forward, worldmirror.py:120-152) ---- */
size_t wm_workspace_bytes(const wm_handle* h, int n_local, int n_total, int H, int W);
/* Workspace ownership ("no hidden allocation in wm_forward").  wm_forward / wm_forward_sharded never allocate,
 * synchronise or build tables: the workspace for the call's shape must have been prepared by wm_reserve, which sizes
 * (wm_workspace_bytes) and lays out the arena and uploads the shape- and weight-derived tables (RoPE, resampled
 * pos_embed of vision_transformer.py:175-207, the DPT UV tables of dense_head.py:253-263).  Call it again after a
 * shape change or after wm_set_weight / wm_finalize_weights; a forward without it returns WM_ERR_STATE.
 * The arena is the library's (hipMalloc inside wm_reserve, grown only when a shape needs more) unless the caller
 * provides device memory with wm_set_workspace (NULL returns ownership to the library): then wm_reserve fails with
 * WM_ERR_STATE instead of allocating when `bytes` < wm_workspace_bytes(...). */
wm_status wm_reserve(wm_handle* h, int n_local, int n_total, int H, int W);
wm_status wm_set_workspace(wm_handle* h, void* device_ptr, size_t bytes);
/* img [N][3][H][W] f32 in [0,1]; priors already normalised as extract_priors returns them
 * (worldmirror.py:218-251): pose7 [N][7], depth [N][H][W], ray4 [N][4]; any may be NULL.
 * cond_flags = [pose, depth, rays]. */
wm_status wm_forward(wm_handle* h, const float* img, int N, int H, int W, const float* pose7, const float* depth,
                     const float* ray4, const int32_t cond_flags[3], const wm_outputs* out, void* stream);
/* View-sharded forward: this rank owns views [first_view, first_view + n_local) of n_total; K/V of
 * every global-attention layer and the camera tokens are all-gathered through the handle's comm. */
wm_status wm_forward_sharded(wm_handle* h, const float* img, int n_local, int first_view, int n_total, int H, int W,
                             const float* pose7, const float* depth, const float* ray4, const int32_t cond_flags[3],
                             const wm_outputs* out, void* stream);

/* ---- communicator for the sharded path ---- */
#define WM_RCCL_ID_BYTES 128
wm_status wm_rccl_unique_id(uint8_t id[WM_RCCL_ID_BYTES]);
wm_status wm_comm_init_rccl(wm_handle* h, const uint8_t id[WM_RCCL_ID_BYTES], int rank, int world);
/* In-process group (ranks = host threads, one handle each, possibly sharing one GPU): used by tests. */
typedef struct wm_local_group wm_local_group;
wm_local_group* wm_local_group_create(int world);
void wm_local_group_destroy(wm_local_group* g);
wm_status wm_comm_init_local(wm_handle* h, wm_local_group* g, int rank);
/* All-gather over the handle's communicator (RCCL, or the in-process group), stream-ordered: recv = [world][bytes_per_rank].
 * For the ONE cross-view step behind the forward: the reference's prune_gs merges the splats of ALL views
 * (src/models/models/rasterization.py:301-387), so a sharded forward gathers the per-rank raw splats before the merge. */
wm_status wm_allgather(wm_handle* h, const void* send, void* recv, size_t bytes_per_rank, void* stream);

/* ---- timing hooks used by bench.py: HIP events on the launch stream around kernel classes ---- */
/* kind: 0 = global attention, 1 = frame+dino attention, 2 = GEMM (epilogues other than the three below), 3 = DPT conv,
 * 4 = whole forward, 5 = GEMM with the fused qkv epilogue, 6 = GEMM with the LayerScale+residual epilogue (proj, fc2),
 * 7 = GEMM with the GELU epilogue (fc1), 8 / 9 = DPT 3x3 convs F -> F on the 4x / 2x pyramid levels, 10 = output_conv1 with the
 * fused resize, 11 = output_conv2 (32 channels) with the fused tail; 3 then holds the remaining convs; 12 = the all-gathers of a
 * sharded forward (K|V per global layer + the camera tokens), timed on the queue they run on: with the gather on the compute
 * queue (the default) that is the time the collective is exposed */
wm_status wm_profile_enable(wm_handle* h, int on);
wm_status wm_profile_read(wm_handle* h, int kind, double* total_ms, int64_t* launches);

/* ---- operator-level entry points (device pointers) used by the parity tests ---- */
wm_status wm_op_gemm(int dtype, int epi, const void* A, const void* W, void* C, const float* bias, const float* gamma,
                     int M, int N, int K, void* stream);
/* QKV projection with the q/k-norm + 2-D RoPE + head-major relayout epilogue (attention.py:50-56): A [M][K] 16-bit,
 * W [3*H*64][K] 16-bit -> q,k,v 16-bit [H][M][64] */
wm_status wm_op_gemm_qkv(int dtype, const void* A, const void* W, const float* bias, void* q, void* k, void* v, const float* qn_w,
                         const float* qn_b, const float* kn_w, const float* kn_b, const float* rope_cos, const float* rope_sin, int M,
                         int H, int K, int tokens_per_view, int patch_start, int grid_w, float q_scale, void* stream);
/* X[M][1024] += gamma * (A W^T + bias) with the FOLLOWING LayerNorm fused into the epilogue (round 4: block.py:44,61 behind :90-92;
   mlp.py:29-35 / attention.py:67): ln_out[M][1024] (16-bit, the operand type) = LayerNorm(X_new) * ln_w + ln_b.  stats: M * 8 floats of
   scratch; sync: 3 * (M / 16 + 2) ints, zero before the first call (the kernels leave them zero).  *fused_out tells whether the launch
   took the fused epilogue (N = 1024, every block resident at once); if not, X is updated and ln_out is left untouched. */
wm_status wm_op_gemm_resid_ln(int dtype, const void* A, const void* W, float* X, const float* bias, const float* gamma, const float* ln_w,
                              const float* ln_b, float ln_eps, void* ln_out, float* stats, int* sync, int M, int N, int K, int* fused_out,
                              void* stream);
/* Q must be pre-scaled by log2(e)/sqrt(64) (what wm_op_qkv_post does with q_scale): softmax is evaluated in base 2 */
wm_status wm_op_attention(int dtype, const void* Q, const void* K, const void* V, void* O, int H, int q_rows, int seq_len,
                          int kv_chunks, int kv_rows_per_chunk, void* stream);
/* Same with split-KV: kv_splits (1..4; 0 = choose) blocks per query tile each walk a slice of the keys and write
 * unnormalised partials into part_o (fp32 [kv_splits][q_rows][H*64]) and part_ml (fp32 [kv_splits][H][q_rows][2]);
 * a combine kernel finishes the softmax.  Used by the forward for cross-view attention when q-tiles x heads does
 * not fill the chip. */
wm_status wm_op_attention_split(int dtype, const void* Q, const void* K, const void* V, void* O, int H, int q_rows, int seq_len,
                                int kv_chunks, int kv_rows_per_chunk, int kv_splits, float* part_o, float* part_ml, void* stream);
/* The same with a flag workspace: unit_flags = int[wm_op_attention_flag_count(q_rows, seq_len, H)].  With it (bf16, whole
 * 64-key tiles, long sequences; selected by wm_set_tuning("attn_qb", 7) or the forward's own choice) the software-pipelined
 * kernel without a running max runs first and the general kernel recomputes the blocks it flagged (attention_v3.hip). */
wm_status wm_op_attention_ex(int dtype, const void* Q, const void* K, const void* V, void* O, int H, int q_rows, int seq_len,
                             int kv_chunks, int kv_rows_per_chunk, int kv_splits, float* part_o, float* part_ml, int* unit_flags,
                             void* stream);
size_t wm_op_attention_flag_count(int q_rows, int seq_len, int H);
/* prepare_splats' per-pixel assembly (src/models/models/rasterization.py:389-498, position_from = "gsdepth+predcamera"): gp [N*H*W][12]
 * raw head outputs, img [N][3][H][W], depth [N][H][W], cam [N][9] -> means / quats / scales / opacities / sh / weights */
wm_status wm_op_gs_splat(const float* gp, const float* img, const float* depth, const float* cam, float* means, float* quats,
                         float* scales, float* opac, float* sh, float* wts, int N, int H, int W, void* stream);
wm_status wm_op_layernorm(const float* x, void* y, const float* w, const float* b, int rows, int D, float eps, int out_f32,
                          int dtype, void* stream);
wm_status wm_op_qkv_post(int dtype, const float* qkv, void* q, void* k, void* v, const float* qn_w, const float* qn_b,
                         const float* kn_w, const float* kn_b, const float* rope_cos, const float* rope_sin, int M, int H,
                         int tokens_per_view, int patch_start, int grid_w, float q_scale, void* stream);
wm_status wm_op_conv(int dtype, const float* x, const void* w16, const float* bias, const float* resid, const float* resid2,
                     float* y, int N, int Hi, int Wi, int Cin, int Cout, int ksize, int stride, int pad, int relu_in,
                     int resid_relu, void* stream);
/* 3x3 / stride 1 / pad 1 conv of interpolate(x, (Hi, Wi), bilinear, align_corners=True) [+ separable position tables
 * addx [Wi][Cin/2], addy [Hi][Cin/2] or NULL], x NHWC [N][Hs][Ws][Cin]: the resize (dense_head.py:217-225) is fused
 * into the conv's input staging, the resized tensor is never stored.  Cin % 64 == 0, Cout % 4 == 0. */
wm_status wm_op_conv3x3_up(int dtype, const float* x, const void* w16, const float* bias, float* y, int N, int Hs, int Ws, int Hi, int Wi,
                           int Cin, int Cout, const float* addx, const float* addy, void* stream);
wm_status wm_op_bilinear(const float* in, float* out, int N, int Hi, int Wi, int Ho, int Wo, int C, void* stream);
wm_status wm_op_linear_f32(const float* X, const float* W, const float* b, float* Y, int M, int N, int K, int ldx, int pre_act,
                           int post_act, void* stream);
/* host helper: fp32 -> 16-bit (round to nearest even), for building test operands */
/* ---- post-path geometry (the step right after the path in infer.py:303 / app.py:151) ----
 * depth_to_world_coords_points (src/models/utils/geometry.py:57-89): depth [B][H][W] f32, extrinsic [B][4][4]
 * camera-to-world, intrinsic [B][3][3] -> world [B][H][W][3], cam [B][H][W][3], mask [B][H][W] (u8, depth > eps).
 * Any of world / cam / mask may be NULL.  Device pointers, stream-ordered. */
wm_status wm_depth_to_world(const float* depth, const float* extrinsic, const float* intrinsic, float* world, float* cam,
                            unsigned char* mask, int B, int H, int W, float eps, void* stream);

/* ---- image ingest after decode ----
 * load_and_preprocess_images (src/utils/inference_utils.py:67-108) for ONE decoded image: rgb [H][W][3] uint8 (device)
 * -> Pillow-exact BICUBIC resize to (518, round(H*518/W/14)*14) ["crop", mode 0] or the longer side to 518 ["pad",
 * mode 1] -> /255 -> centre crop of the height / white padding to the square -> out planar float32 [3][out_h][out_w]
 * (device; sizes from wm_preprocess_image_size).  Decoding, alpha compositing and stacking stay on the host side of
 * the binding.  workspace: wm_preprocess_image_workspace_bytes bytes of device memory. */
wm_status wm_preprocess_image_size(int H, int W, int mode, int output_size, int* out_h, int* out_w);
size_t wm_preprocess_image_workspace_bytes(int H, int W, int mode, int output_size);
wm_status wm_preprocess_image(const unsigned char* rgb, int H, int W, int mode, int output_size, float* out, void* workspace,
                              size_t workspace_bytes, void* stream);

/* create_confidence_mask (infer.py:25-59): mask[i] = 1 for the top ceil(n (100 - p) / 100) (at least 1; p <= 0: all)
 * confidences after conf <= 1e-5 -> -inf; exact radix select on the device, ties at the threshold value broken by
 * lowest index (the reference's torch.topk leaves them unspecified).  workspace: wm_confidence_mask_workspace_bytes(n)
 * bytes of device memory.  n < 2^32. */
size_t wm_confidence_mask_workspace_bytes(size_t n);
wm_status wm_confidence_mask(const float* conf, size_t n, float conf_threshold_percent, unsigned char* mask, void* workspace,
                             size_t workspace_bytes, void* stream);

/* Operator-level entry (parity tests / A-B): F.interpolate(x, (Hi, Wi), bilinear, align_corners=True) (+ separable position tables
 * addx [Wi][Cin/2], addy [Hi][Cin/2], may be NULL) rounded to the 16-bit operand type, then Conv2d(Cin, 32, 3, padding=1) (+ ReLU):
 * the un-fused form the DPT tail uses for output_conv2[0] (dense_head.py:97-105,217-251).  up16: caller-owned device scratch of
 * N*Hi*Wi*Cin 16-bit elements + 32 bytes. */
wm_status wm_op_up_conv_n32(int dtype, const float* x, const void* w16, const float* bias, float* y, int N, int Hs, int Ws, int Hi,
                            int Wi, int Cin, const float* addx, const float* addy, int relu_out, void* up16, void* stream);

/* Operator-level entry (parity tests / A-B): wm_op_conv's 3x3 / stride 1 / pad 1 case with the 16-bit tensor forms the DPT heads use
 * between their convs (dense_head.py:435-455): x is a 16-bit NHWC tensor of the operand type when in16 (relu_in must be 0: the producer
 * applied it), y is one when out16 (+ ReLU when relu_out).  WM_ERR_INVALID when the shape's kernel has no such form. */
wm_status wm_op_conv_ex(int dtype, const void* x, int in16, const void* w16, const float* bias, const float* resid, const float* resid2,
                        void* y, int out16, int N, int Hi, int Wi, int Cin, int Cout, int relu_in, int resid_relu, int relu_out, void* stream);

/* Operator-level entry (parity tests / A-B): Conv2d(Cm, 256, 3, padding=1, bias=False)(ConvTranspose2d(Cin, Cm, k, stride=k)(tokens)) — the DPT
 * head's resize_layers[0 / 1] followed by scratch.layer{1,2}_rn (dense_head.py:57-66,196-214,277-278,394-399) — composed into ONE block-sparse GEMM
 * at the token resolution: per output phase the 3x3 taps land in at most 2 x 2 neighbouring tokens, so 36 (k = 4) / 16 (k = 2) combined
 * matrices replace k^2 x 9 tap products per token.  tokens16: device, 16-bit NHWC [N][gh][gw][Cin]; wct [Cin][Cm][k][k], bct [Cm], wrn
 * [256][Cm][3][3]: HOST fp32, torch layouts (combined in fp32 on the device, rounded once); out: device fp32 NHWC [N][k*gh][k*gw][256];
 * zero16: >= 128 B of device zeros.  k in {2, 4}; Cin % 64 == 0.  Synchronises the stream (it builds and frees the combined weights). */
wm_status wm_op_tconv(int dtype, const void* tokens16, const float* wct, const float* bct, const float* wrn, float* out, int N, int gh, int gw,
                      int k, int Cin, int Cm, const void* zero16, void* stream);

/* Operator-level entry (parity tests / A-B): Conv2d(C, Co, 3, padding=1)(F.interpolate(x, (Ho, Wo), mode="bilinear", align_corners=True)) —
 * the DPT head's output_conv1 behind its last resize (dense_head.py:217-225,265-295) — in the tap form: the nine 1x1 products W_tap x at the
 * LOW resolution (one GEMM, a quarter of the direct conv's flops) and a bilinear gather of them.  x16: f16 NHWC [N][Hi][Wi][C]; w16:
 * [Co][3][3][C]; out: fp32 NHWC [N][Ho][Wo][Co]; wt16 / y16: caller-owned device scratch of 9*Co*C and N*Hi*Wi*9*Co 16-bit elements.
 * dtype must be f16 (the products are stored in the operand type); C % 64 == 0; Co in {32, 64, 128}. */
wm_status wm_op_upconv3x3_tap(int dtype, const void* x16, const void* w16, const float* bias, float* out, int N, int Hi, int Wi, int Ho,
                              int Wo, int C, int Co, void* wt16, void* y16, void* stream);

/* The gather half of wm_op_upconv3x3_tap on its own (timing tools): y16 f16 [N][Hi][Wi][9][Co] -> out fp32 [N][Ho][Wo][Co] (+ bias). */
wm_status wm_op_upconv_gather(const void* y16, const float* bias, float* out, int N, int Hi, int Wi, int Ho, int Wo, int Co, void* stream);

/* Operator-level entry (parity tests / A-B): Conv2d(Cin, Cout, 3, padding=1) on a 16-BIT NHWC tensor x16 [N][H][W][Cin] of the operand
 * type, run as the ping-pong GEMM itself (rows = pixels, K = (tap, channel); no im2col): y = conv(x16) + bias + relu?(resid) + resid2,
 * optional ReLU; y is fp32 NHWC, or 16-bit NHWC when out16.  The form the ResidualConvUnit's second conv takes (dense_head.py:435-455)
 * when tuning "conv_gemm" = 1.  zero16: >= 128 B of device zeros.  Cin % 64 == 0, Cout % 8 == 0. */
wm_status wm_op_conv3x3_gemm16(int dtype, const void* x16, const void* w16, const float* bias, const float* resid, int resid_relu,
                               const float* resid2, void* y, int out16, int relu_out, int N, int H, int W, int Cin, int Cout,
                               const void* zero16, void* stream);

/* Voxel merge of the per-pixel splats — GaussianSplatRenderer.prune_gs (src/models/models/rasterization.py:301-387; called at
 * :216 and on predictions["splats"] by the callers).  Inputs [n, ...] device fp32: means [n,3], quats [n,4], scales [n,3],
 * opacities [n] (not read: the merged opacity is sum w^2 / sum w, as in the reference), sh [n,3] (degree-0 coefficients), weights [n].
 * Outputs have room for n rows; *n_voxels (host) = occupied voxels K, rows [0, K) are valid and ordered by ascending voxel
 * index (torch.unique's order).  Sums run in original index order, i.e. bit-identical to the reference on CPU.  One stream
 * synchronisation (K is data dependent). */
size_t wm_prune_gs_workspace_bytes(size_t n);
wm_status wm_prune_gs(const float* means, const float* quats, const float* scales, const float* opacities, const float* sh,
                      const float* weights, int n, float voxel_size, float* out_means, float* out_quats, float* out_scales,
                      float* out_opacities, float* out_sh, int* n_voxels, void* workspace, size_t workspace_bytes, void* stream);

/* 3D-Gaussian-splat rasteriser forward — replaces gsplat.rasterization as the reference calls it through
 * Rasterizer.rasterize_splats (src/models/models/rasterization.py:29-66; callers: GaussianSplatRenderer.render :221-241,
 * render_interpolated_video src/utils/render_utils.py:242-312 <- infer.py:264): packed, rasterize_mode "classic", pinhole,
 * render_mode "RGB+ED" (colour + expected depth), tile size 16, eps2d 0.3, near plane 0.01, no background.
 * means [N,3], quats [N,4] (wxyz, normalised inside), scales [N,3], opacities [N], colors [N,3]: degree-0 SH coefficients
 * (colors_are_sh0 = 1: colour = max(0.2820948 sh + 0.5, 0), the reference's sh_degree = 0 call) or final colours (0: the
 * reference's sh_degree = None call); viewmats [C,4,4] WORLD-TO-CAMERA (the reference inverts its camtoworlds before the
 * call, :48), Ks [C,3,3].  Outputs (caller-owned device buffers): out_rgb [C,H,W,3], out_depth [C,H,W] (expected depth
 * = sum w z / sum w), out_alpha [C,H,W]; radii_out optional [C,N,2] int32 (the projection's screen radii, 0 = culled).
 * The number of (Gaussian, tile) pairs is data dependent: the call synchronises the stream once to read it (as the
 * reference's isect_tiles does); if it exceeds max_isects the call returns WM_ERR_STATE with *n_isects = the required count
 * and renders nothing — re-size the workspace with wm_rasterize_workspace_bytes and call again. */
size_t wm_rasterize_workspace_bytes(int n_gaussians, int n_cameras, int width, int height, size_t max_isects);
wm_status wm_rasterize_splats(const float* means, const float* quats, const float* scales, const float* opacities,
                              const float* colors, int colors_are_sh0, int n_gaussians, const float* viewmats, const float* Ks,
                              int n_cameras, int width, int height, float* out_rgb, float* out_depth, float* out_alpha,
                              int* radii_out, void* workspace, size_t workspace_bytes, size_t max_isects,
                              unsigned long long* n_isects, void* stream);

/* Process-wide kernel-selection override for tests and A/B tools (no reference counterpart).  key: "gemm_cfg"
 * (tile config id), "gemm_pp" (0/1 ping-pong GEMM), "gemm_mfma16" (0/1/2), "attn_qb" (attention variant);
 * value -1 restores the default.  Returns 0, or -1 for an unknown key. */
int wm_set_tuning(const char* key, int value);

void wm_host_to_16(const float* in, uint16_t* out, size_t n, int dtype);

#ifdef __cplusplus
}
#endif
#endif
