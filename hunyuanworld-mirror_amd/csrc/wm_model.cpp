// Host orchestrator + C ABI (include/wm_hip.h) of the MI355X-native WorldMirror forward pass.
//
// One wm_handle = one process/GPU: owns the repacked weights and a workspace arena, and turns
// WorldMirror.forward (reference src/models/models/worldmirror.py:120-216) into a fixed sequence of
// HIP kernel launches on the caller's stream.  No torch, no hidden per-call allocation once the
// workspace for a shape exists.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <pthread.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/wm_hip.h"
#include "wm_kernels.h"

namespace {

inline int ru(int x, int m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------ host 16-bit conversion
inline uint16_t h_f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
inline uint16_t h_f2h(float f) {
  _Float16 h = (_Float16)f;
  uint16_t r;
  memcpy(&r, &h, 2);
  return r;
}
inline uint16_t h_to16(float f, int dt) { return dt == WM_DT_BF16 ? h_f2bf(f) : h_f2h(f); }

struct Weight {
  std::vector<int64_t> shape;
  float* f32 = nullptr;  // device
  void* w16 = nullptr;   // device 16-bit repack
  int k16 = 0;           // padded K of the repack
  bool tap_major = false;  // w16 holds a second copy [tap][Cout][Cin] of a 3x3 conv weight behind the first (numel elements in)
  std::vector<float> host;  // kept only for the few tensors the host needs (pos_embed, init_token)
  bool set = false;
  bool owned = true;  // false: device memory belongs to another handle (wm_share_weights)
};

// the composed ConvTranspose -> 3x3 conv of a DPT head level (build_tconv below)
struct TconvPack {
  void* w16 = nullptr;        // [k^2 * 256][4 * Cin] 16-bit: row (phase, co), the phase's neighbour matrices side by side
  float* bias_rep = nullptr;  // [k^2][256]: (sum_tap W_rn[tap]) b_ct, once per phase
  float* bmiss = nullptr;     // [9][256]: W_rn[tap] b_ct (wm_launch_tconv_border)
  unsigned list[16] = {0};
  int k = 0, cin = 0;
};
static void free_tconv(TconvPack& t) {
  if (t.w16) (void)hipFree(t.w16);
  if (t.bias_rep) (void)hipFree(t.bias_rep);
  if (t.bmiss) (void)hipFree(t.bmiss);
  t = TconvPack();
}

static hipError_t build_tconv(int dt, int k, int Cin, int Cm, int F_, const float* wct, const float* bct, const float* wrn, TconvPack& out, hipStream_t s);
// refinenet1.out_conv (1x1) composed into the nine tap matrices of output_conv1 (upconv.hip): the tap GEMM then reads the last fusion block's
// 16-bit tensor directly — W[tap][co][ci] = sum_c W_oc1[co][c][tap] W_out[c][ci], bias[tap][co] = sum_c W_oc1[co][c][tap] b_out[c] (build_up1comp)
struct Up1Comp {
  void* w16 = nullptr;     // [9 Co][F] 16-bit
  float* bias = nullptr;   // [9 Co]
  int co = 0;
};
static void free_up1comp(Up1Comp& u) {
  if (u.w16) (void)hipFree(u.w16);
  if (u.bias) (void)hipFree(u.bias);
  u = Up1Comp();
}
static hipError_t build_up1comp(int dt, int F_, int Co, const float* woc1, const float* wout, const float* bout, Up1Comp& out, hipStream_t s);

struct EvPair { hipEvent_t a, b; };

struct Comm {
  int kind = 0;  // 0 none, 1 rccl, 2 local
  int rank = 0, world = 1;
  ncclComm_t nccl = nullptr;
  wm_local_group* grp = nullptr;
};

}  // namespace

struct wm_local_group {
  int world;
  pthread_barrier_t bar;
  std::vector<void*> recv;
};

struct wm_handle {
  wm_config cfg;
  int device = 0;
  std::string err;
  std::unordered_map<std::string, Weight> w;
  std::map<std::string, std::vector<int64_t>> spec;
  bool finalized = false;
  Comm comm;
  // workspace
  char* arena = nullptr;
  size_t arena_bytes = 0;
  bool arena_owned = true;   // false: caller-provided (wm_set_workspace)
  int plan_n = -1, plan_nt = -1, plan_H = -1, plan_W = -1;
  std::map<std::string, TconvPack> tconv;   // per head and level ("pts_head.0"): rebuilt by wm_reserve after a weight change
  bool tconv_valid = false;
  std::map<std::string, Up1Comp> up1comp;   // per head ("pts_head."), rebuilt with the token-conv packs
  std::vector<std::string> missing;  // names wm_finalize_weights filled with their init values
  // shape-dependent device tables (allocated inside the arena by plan())
  std::map<std::string, void*> buf;
  // the DPT heads are mutually independent: each runs on its own stream (forked/joined with events)
  hipStream_t hstream[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t hfork = nullptr, hjoin[4] = {nullptr, nullptr, nullptr, nullptr};
  // K/V all-gather of the sharded forward runs on its own queue, under the attention over the local keys
  hipStream_t cstream = nullptr;
  hipEvent_t cfork = nullptr, cjoin = nullptr;
  // the camera head (HBM-bound weight streaming) runs beside the DPT heads (MFMA-bound) on its own queue
  hipStream_t camstream = nullptr;
  hipEvent_t camjoin = nullptr;
  // fast-attention fallback statistics per call site (WmAttnArgs::unit_stat): pinned host copy written by an async D2H at the end of
  // every forward, the last value seen per site, and the calls left in general-kernel-only mode
  int* att_stat_host = nullptr;
  int att_stat_n = 0;
  std::vector<int> att_seen, att_general_ttl;
  // profiling
  bool prof = false;
  static constexpr int NKIND = 13;
  std::vector<EvPair> ev[NKIND];
  size_t ev_used[NKIND] = {};
};

namespace {

wm_status fail(wm_handle* h, wm_status st, const std::string& msg) {
  if (h) h->err = msg;
  return st;
}
#define HIPCHK(h, e)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (e);                                                                               \
    if (_e != hipSuccess)                                                                              \
      return fail(h, WM_ERR_HIP, std::string(hipGetErrorString(_e)) + " at " + __FILE__ + ":" + std::to_string(__LINE__)); \
  } while (0)

bool ends_with(const std::string& s, const char* suf) {
  const size_t n = strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}
bool starts_with(const std::string& s, const char* p) { return s.compare(0, strlen(p), p) == 0; }

enum WKind { WK_F32, WK_LIN16_BACKBONE, WK_LIN16_HEAD, WK_CONV16_HEAD, WK_CONVT16_HEAD };

WKind classify(const std::string& n, int ndim) {
  const bool vgt = starts_with(n, "visual_geometry_transformer.");
  if (vgt && ndim >= 2) {
    if (ends_with(n, "attn.qkv.weight") || ends_with(n, "attn.proj.weight") || ends_with(n, "mlp.fc1.weight") ||
        ends_with(n, "mlp.fc2.weight") || ends_with(n, "patch_embed.proj.weight") || ends_with(n, "depth_embed.proj.2.fc1.weight") ||
        ends_with(n, "depth_embed.proj.2.fc2.weight"))
      return WK_LIN16_BACKBONE;
    return WK_F32;
  }
  const bool head = starts_with(n, "pts_head.") || starts_with(n, "depth_head.") || starts_with(n, "norm_head.") || starts_with(n, "gs_head.");
  if (head && ndim == 4 && ends_with(n, ".weight")) {
    if (ends_with(n, "scratch.output_conv2.2.weight")) return WK_F32;
    if (n.find("input_merger") != std::string::npos) return WK_LIN16_HEAD;  // 7x7 conv on the image as a GEMM, K = 3*49
    if (n.find("resize_layers.0.") != std::string::npos || n.find("resize_layers.1.") != std::string::npos) return WK_CONVT16_HEAD;
    if (n.find(".projects.") != std::string::npos) return WK_LIN16_HEAD;
    return WK_CONV16_HEAD;
  }
  if (n == "gs_renderer.gs_head.0.weight" || n == "gs_renderer.gs_head.2.weight") return WK_CONV16_HEAD;
  return WK_F32;
}

// ------------------------------------------------------------------ parameter list (mirrors config.param_spec)
void spec_block(std::vector<std::pair<std::string, std::vector<int64_t>>>& s, const std::string& p, int D, int hid, int qk) {
  auto add = [&](const char* n, std::vector<int64_t> sh) { s.push_back({p + n, sh}); };
  add("norm1.weight", {D}); add("norm1.bias", {D});
  add("attn.qkv.weight", {3 * D, D}); add("attn.qkv.bias", {3 * D});
  if (qk) { add("attn.q_norm.weight", {qk}); add("attn.q_norm.bias", {qk}); add("attn.k_norm.weight", {qk}); add("attn.k_norm.bias", {qk}); }
  add("attn.proj.weight", {D, D}); add("attn.proj.bias", {D}); add("ls1.gamma", {D});
  add("norm2.weight", {D}); add("norm2.bias", {D});
  add("mlp.fc1.weight", {hid, D}); add("mlp.fc1.bias", {hid}); add("mlp.fc2.weight", {D, hid}); add("mlp.fc2.bias", {D});
  add("ls2.gamma", {D});
}

std::vector<std::pair<std::string, std::vector<int64_t>>> param_spec(const wm_config& c) {
  std::vector<std::pair<std::string, std::vector<int64_t>>> s;
  const int D = c.embed_dim, g = c.img_size / c.patch_size, R = c.num_register_tokens, H4 = c.mlp_ratio * D;
  const std::string v = "visual_geometry_transformer.", d = v + "patch_embed.";
  s.push_back({v + "cam_token", {1, 2, 1, D}});
  s.push_back({v + "reg_token", {1, 2, R, D}});
  s.push_back({d + "cls_token", {1, 1, D}});
  s.push_back({d + "pos_embed", {1, 1 + g * g, D}});
  s.push_back({d + "register_tokens", {1, R, D}});
  s.push_back({d + "patch_embed.proj.weight", {D, 3, c.patch_size, c.patch_size}});
  s.push_back({d + "patch_embed.proj.bias", {D}});
  for (int i = 0; i < c.dino_depth; ++i) spec_block(s, d + "blocks." + std::to_string(i) + ".", D, H4, 0);
  s.push_back({d + "norm.weight", {D}});
  s.push_back({d + "norm.bias", {D}});
  if (c.enable_cond) {
    s.push_back({v + "pose_embed.0.weight", {D, 7}}); s.push_back({v + "pose_embed.0.bias", {D}});
    s.push_back({v + "pose_embed.2.weight", {D, D}}); s.push_back({v + "pose_embed.2.bias", {D}});
    s.push_back({v + "depth_embed.proj.2.fc1.weight", {4 * D, c.patch_size * c.patch_size}});
    s.push_back({v + "depth_embed.proj.2.fc1.bias", {4 * D}});
    s.push_back({v + "depth_embed.proj.2.fc2.weight", {D, 4 * D}}); s.push_back({v + "depth_embed.proj.2.fc2.bias", {D}});
    s.push_back({v + "ray_embed.0.weight", {D, 4}}); s.push_back({v + "ray_embed.0.bias", {D}});
    s.push_back({v + "ray_embed.2.weight", {D, D}}); s.push_back({v + "ray_embed.2.bias", {D}});
  }
  for (int i = 0; i < c.depth; ++i) spec_block(s, v + "frame_blocks." + std::to_string(i) + ".", D, H4, D / c.num_heads);
  for (int i = 0; i < c.depth; ++i) spec_block(s, v + "global_blocks." + std::to_string(i) + ".", D, H4, D / c.num_heads);
  const int D2 = 2 * D;
  if (c.enable_cam) {
    const std::string ch = "cam_head.";
    for (int i = 0; i < c.cam_trunk_depth; ++i) spec_block(s, ch + "refine_net." + std::to_string(i) + ".", D2, 4 * D2, 0);
    s.push_back({ch + "token_norm.weight", {D2}}); s.push_back({ch + "token_norm.bias", {D2}});
    s.push_back({ch + "out_norm.weight", {D2}}); s.push_back({ch + "out_norm.bias", {D2}});
    s.push_back({ch + "init_token", {1, 1, 9}});
    s.push_back({ch + "param_embed.weight", {D2, 9}}); s.push_back({ch + "param_embed.bias", {D2}});
    s.push_back({ch + "adapt_norm_gen.1.weight", {3 * D2, D2}}); s.push_back({ch + "adapt_norm_gen.1.bias", {3 * D2}});
    s.push_back({ch + "param_predictor.fc1.weight", {D2 / 2, D2}}); s.push_back({ch + "param_predictor.fc1.bias", {D2 / 2}});
    s.push_back({ch + "param_predictor.fc2.weight", {9, D2 / 2}}); s.push_back({ch + "param_predictor.fc2.bias", {9}});
  }
  auto dpt = [&](const std::string& p, int F, int od, bool gs) {
    const int32_t* oc = c.dpt_out_channels;
    s.push_back({p + "norm.weight", {D2}}); s.push_back({p + "norm.bias", {D2}});
    for (int i = 0; i < 4; ++i) {
      s.push_back({p + "projects." + std::to_string(i) + ".weight", {oc[i], D2, 1, 1}});
      s.push_back({p + "projects." + std::to_string(i) + ".bias", {oc[i]}});
    }
    s.push_back({p + "resize_layers.0.weight", {oc[0], oc[0], 4, 4}}); s.push_back({p + "resize_layers.0.bias", {oc[0]}});
    s.push_back({p + "resize_layers.1.weight", {oc[1], oc[1], 2, 2}}); s.push_back({p + "resize_layers.1.bias", {oc[1]}});
    s.push_back({p + "resize_layers.3.weight", {oc[3], oc[3], 3, 3}}); s.push_back({p + "resize_layers.3.bias", {oc[3]}});
    for (int i = 0; i < 4; ++i) s.push_back({p + "scratch.layer" + std::to_string(i + 1) + "_rn.weight", {F, oc[i], 3, 3}});
    for (int r = 1; r <= 4; ++r) {
      const std::string q = p + "scratch.refinenet" + std::to_string(r) + ".";
      s.push_back({q + "out_conv.weight", {F, F, 1, 1}}); s.push_back({q + "out_conv.bias", {F}});
      for (int u = (r == 4 ? 2 : 1); u <= 2; ++u)
        for (int cv = 1; cv <= 2; ++cv) {
          const std::string nm = q + "resConfUnit" + std::to_string(u) + ".conv" + std::to_string(cv);
          s.push_back({nm + ".weight", {F, F, 3, 3}}); s.push_back({nm + ".bias", {F}});
        }
    }
    s.push_back({p + "scratch.output_conv1.weight", {F / 2, F, 3, 3}}); s.push_back({p + "scratch.output_conv1.bias", {F / 2}});
    s.push_back({p + "scratch.output_conv2.0.weight", {32, F / 2, 3, 3}}); s.push_back({p + "scratch.output_conv2.0.bias", {32}});
    s.push_back({p + "scratch.output_conv2.2.weight", {od, 32, 1, 1}}); s.push_back({p + "scratch.output_conv2.2.bias", {od}});
    if (gs) { s.push_back({p + "input_merger.0.weight", {F / 2, 3, 7, 7}}); s.push_back({p + "input_merger.0.bias", {F / 2}}); }
  };
  if (c.enable_pts) dpt("pts_head.", c.dpt_features, 4, false);
  if (c.enable_depth) dpt("depth_head.", c.dpt_features, 2, false);
  if (c.enable_norm) dpt("norm_head.", c.dpt_features, 4, false);
  if (c.enable_gs) {
    dpt("gs_head.", c.gs_dim, 2, true);
    s.push_back({"gs_renderer.gs_head.0.weight", {c.gs_dim, c.gs_dim / 2, 3, 3}});
    s.push_back({"gs_renderer.gs_head.2.weight", {12, c.gs_dim, 1, 1}});
    s.push_back({"gs_renderer.gs_head.2.bias", {12}});
  }
  return s;
}

// ------------------------------------------------------------------ small accessors
const Weight* W(const wm_handle* h, const std::string& n) {
  auto it = h->w.find(n);
  return it == h->w.end() ? nullptr : &it->second;
}
const float* F(const wm_handle* h, const std::string& n) {
  const Weight* w = W(h, n);
  return w ? w->f32 : nullptr;
}
const void* W16(const wm_handle* h, const std::string& n) {
  const Weight* w = W(h, n);
  return w ? w->w16 : nullptr;
}

struct Dims {
  int n, nt, H, W, gh, gw, hw, Td, P, psi, D, heads, R, Md, Mv, Mx, kpad_patch, kpad_depth, world, chunk;
  int gh2, gw2;
};

Dims make_dims(const wm_handle* h, int n, int nt, int H, int W) {
  const wm_config& c = h->cfg;
  Dims d;
  d.n = n; d.nt = nt; d.H = H; d.W = W;
  d.gh = H / c.patch_size; d.gw = W / c.patch_size; d.hw = d.gh * d.gw;
  d.R = c.num_register_tokens;
  d.Td = 1 + d.R + d.hw;
  d.psi = 1 + d.R + (c.enable_cond ? 2 : 0);
  d.P = d.psi + d.hw;
  d.D = c.embed_dim; d.heads = c.num_heads;
  d.Md = n * d.Td; d.Mv = n * d.P; d.Mx = std::max(d.Md, d.Mv);
  d.kpad_patch = ru(3 * c.patch_size * c.patch_size, 64);
  d.kpad_depth = ru(c.patch_size * c.patch_size, 64);
  d.world = nt / std::max(n, 1);
  d.chunk = std::min(n, 8);
  d.gh2 = (d.gh + 2 - 3) / 2 + 1; d.gw2 = (d.gw + 2 - 3) / 2 + 1;
  return d;
}

// arena layout: name -> bytes (in a fixed order)
std::vector<std::pair<std::string, size_t>> arena_layout(const wm_handle* h, const Dims& d) {
  const wm_config& c = h->cfg;
  std::vector<std::pair<std::string, size_t>> L;
  auto add = [&](const char* n, size_t b) { L.push_back({n, (b + 255) / 256 * 256}); };
  const size_t D = d.D, Mx = d.Mx, Mv = d.Mv;
  add("Xd", (size_t)d.Md * D * 4);
  add("Xv", Mv * D * 4);
  add("A16", Mx * std::max<size_t>({D, (size_t)d.kpad_patch, (size_t)d.kpad_depth}) * 2);
  add("QKV16", 3 * Mx * D * 2);
  add("O16", Mx * D * 2);
  add("H16", Mx * 4 * D * 2);
  for (int i = 0; i < 4; ++i) add(("tap" + std::to_string(i)).c_str(), Mv * 2 * D * 4);
  add("KVG", (size_t)d.world * 2 * Mv * D * 2);
  add("ATT_PO", (size_t)WM_ATTN_MAX_SPLITS * Mx * D * 4);   // split-KV attention partials (tail round of a launch cut into up to 8 key slices; uniform 4-way split when a sharded launch fits one round): unnormalised O, (max, sum)
  add("ATT_ML", (size_t)WM_ATTN_MAX_SPLITS * Mx * (D / 64) * 2 * 4);
  add("ATT_FLAGS", wm_attention_max_blocks((int)Mx, d.P < d.Td ? d.P : d.Td, d.heads) * 4);  // per-block fallback flags of the no-max attention kernel
  // sticky fallback hints of the fast attention kernels (WmAttnArgs::unit_hint): one int per launch block, per attention call site
  // (DINO + frame + global blocks) x up to 3 launches (the piecewise form under the overlapped gather); zeroed by wm_reserve
  add("ATT_HINT", (size_t)(c.dino_depth + 2 * c.depth) * 3 * wm_attention_max_blocks((int)Mx, d.P < d.Td ? d.P : d.Td, d.heads) * 4);
  // fused LayerNorm of the residual GEMMs (WmGemmArgs::ln_out): per-row partials of the four column tiles, per-band arrival / departure
  // counters and fallback flags (zeroed by wm_reserve; the kernels leave the counters and flags at zero)
  add("LN_STATS", Mx * 4 * 2 * 4);
  add("LN_SYNC", (Mx / 16 + 2) * 3 * 4);
  add("ATT_STAT", (size_t)(c.dino_depth + 2 * c.depth) * 3 * 4);   // WmAttnArgs::unit_stat, one counter per (call site, launch)
  add("ZERO256", 256);  // zero page for the out-of-image halo pieces of the DMA-fed conv (conv_n32.hip); cleared at the start of every forward
  add("rope_cos", (size_t)(std::max(d.gh, d.gw) + 1) * 16 * 4);
  add("rope_sin", (size_t)(std::max(d.gh, d.gw) + 1) * 16 * 4);
  add("dino_pos", (size_t)(1 + d.hw) * D * 4);
  // priors
  add("pr_in", (size_t)d.n * 16 * 4);
  add("pr_h", (size_t)d.n * D * 4);
  add("pose_tok", (size_t)d.n * D * 4);
  add("ray_tok", (size_t)d.n * D * 4);
  // camera head (fp32, all views)
  const size_t D2 = 2 * D, nt = d.nt;
  add("cam_tok_local", (size_t)d.n * D2 * 4);
  add("cam_tok", nt * D2 * 4);
  add("cam_e", nt * D2 * 4);
  add("cam_mod", nt * 3 * D2 * 4);
  add("cam_h", nt * D2 * 4);
  add("cam_a", nt * D2 * 4);
  add("cam_qkv", nt * 3 * D2 * 4);
  add("cam_o", nt * D2 * 4);
  add("cam_f", nt * 4 * D2 * 4);
  add("cam_pred", nt * 12 * 4);
  add("cam_delta", nt * 12 * 4);
  add("cam_init", nt * 12 * 4);
  add("cam_params", nt * 9 * 4);
  // DPT heads
  const int32_t* oc = c.dpt_out_channels;
  const size_t ch = d.chunk, hw = d.hw;
  const int Fm = std::max(c.dpt_features, c.enable_gs ? c.gs_dim : 0);
  const int nslots = (c.enable_depth ? 1 : 0) + (c.enable_pts ? 1 : 0) + (c.enable_norm ? 1 : 0) + (c.enable_gs ? 1 : 0);
  const size_t big = ch * std::max<size_t>({64 * hw * (size_t)Fm, (size_t)d.H * d.W * (Fm / 2), (size_t)d.H * d.W * 32,
                                            c.enable_gs ? (size_t)d.H * d.W * c.gs_dim : 0}) * 4;
  for (int sl = 0; sl < std::max(nslots, 1); ++sl) {
    const std::string x = "_h" + std::to_string(sl);
    auto addh = [&](const char* n, size_t b) { add((std::string(n) + x).c_str(), b); };
    addh("dpt_T16", ch * hw * D2 * 2);
    addh("dpt_P16", ch * hw * std::max(oc[0], oc[1]) * 2);
    addh("dpt_f0", ch * 16 * hw * oc[0] * 4);
    addh("dpt_f1", ch * 4 * hw * oc[1] * 4);
    addh("dpt_f2", ch * hw * oc[2] * 4);
    addh("dpt_f3in", ch * hw * oc[3] * 4);
    addh("dpt_f3", ch * (size_t)d.gh2 * d.gw2 * oc[3] * 4);
    addh("dpt_rn1", ch * 16 * hw * Fm * 4);
    addh("dpt_rn2", ch * 4 * hw * Fm * 4);
    addh("dpt_rn3", ch * hw * Fm * 4);
    addh("dpt_rn4", ch * (size_t)d.gh2 * d.gw2 * Fm * 4);
    for (int i = 0; i < 4; ++i) addh(("dpt_s" + std::to_string(i)).c_str(), big);
  }
  if (c.enable_gs) add("gs_im2col", ch * (size_t)d.H * d.W * 192 * 2);
  for (int i = 0; i < 4; ++i) add(("dpt_pos" + std::to_string(i)).c_str(), hw * oc[i] * 4);
  add("dpt_posx", (size_t)d.W * (Fm / 4) * 4 * 2);  // one table per feature width (F/2 channels -> F/4 per axis)
  add("dpt_posy", (size_t)d.H * (Fm / 4) * 4 * 2);
  add("gs_posx", (size_t)d.W * (Fm / 4) * 4 * 2);
  add("gs_posy", (size_t)d.H * (Fm / 4) * 4 * 2);
  return L;
}

template <class T> T* B(wm_handle* h, const char* n) { return (T*)h->buf.at(n); }

// torch.linspace(start, end, steps, dtype=float32) (symmetric evaluation, as ATen does)
void linspace_f32(float start, float end, int steps, std::vector<float>& out) {
  out.resize(steps);
  if (steps == 1) { out[0] = start; return; }
  const float step = (end - start) / (float)(steps - 1);
  const int half = steps / 2;
  for (int i = 0; i < steps; ++i) out[i] = i < half ? start + step * (float)i : end - step * (float)(steps - i - 1);
}

// src/models/utils/grid.py:4-90 — separable halves of 0.1 * position_grid_to_embed(create_uv_grid(w,h,aspect), C)
void uv_tables(int w, int hgt, int C, double aspect, std::vector<float>& tx, std::vector<float>& ty) {
  const double diag = std::sqrt(aspect * aspect + 1.0), sx = aspect / diag, sy = 1.0 / diag;
  std::vector<float> u, v;
  linspace_f32((float)(-sx * (w - 1) / w), (float)(sx * (w - 1) / w), w, u);
  linspace_f32((float)(-sy * (hgt - 1) / hgt), (float)(sy * (hgt - 1) / hgt), hgt, v);
  const int q = C / 4;
  std::vector<double> om(q);
  for (int k = 0; k < q; ++k) om[k] = 1.0 / std::pow(100.0, (double)k / (C / 4.0));
  tx.assign((size_t)w * 2 * q, 0.f);
  ty.assign((size_t)hgt * 2 * q, 0.f);
  for (int x = 0; x < w; ++x)
    for (int k = 0; k < q; ++k) {
      tx[(size_t)x * 2 * q + k] = (float)std::sin((double)u[x] * om[k]) * 0.1f;
      tx[(size_t)x * 2 * q + q + k] = (float)std::cos((double)u[x] * om[k]) * 0.1f;
    }
  for (int y = 0; y < hgt; ++y)
    for (int k = 0; k < q; ++k) {
      ty[(size_t)y * 2 * q + k] = (float)std::sin((double)v[y] * om[k]) * 0.1f;
      ty[(size_t)y * 2 * q + q + k] = (float)std::cos((double)v[y] * om[k]) * 0.1f;
    }
}

// ATen's separable anti-aliased bicubic (a = -0.5), align_corners=False — vision_transformer.py:196-201
inline double cubic_aa(double x) {
  const double a = -0.5;
  x = std::fabs(x);
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0;
  if (x < 2.0) return (((x - 5.0) * x + 8.0) * x - 4.0) * a;
  return 0.0;
}
void aa_weights(int in, int out, std::vector<int>& xmin, std::vector<int>& xsize, std::vector<float>& wts, int& maxk) {
  const double scale = (double)in / out;
  const double support = scale >= 1.0 ? 2.0 * scale : 2.0;
  const double invscale = scale >= 1.0 ? 1.0 / scale : 1.0;
  maxk = (int)std::ceil(support) * 2 + 1;
  xmin.resize(out); xsize.resize(out); wts.assign((size_t)out * maxk, 0.f);
  for (int i = 0; i < out; ++i) {
    const double center = scale * (i + 0.5);
    const int lo = std::max(0, (int)(center - support + 0.5));
    const int hi = std::min(in, (int)(center + support + 0.5));
    xmin[i] = lo; xsize[i] = hi - lo;
    double tot = 0;
    std::vector<double> w(hi - lo);
    for (int j = 0; j < hi - lo; ++j) { w[j] = cubic_aa((j + lo - center + 0.5) * invscale); tot += w[j]; }
    for (int j = 0; j < hi - lo; ++j) wts[(size_t)i * maxk + j] = (float)(w[j] / tot);
  }
}

}  // namespace

extern "C" void wm_host_resample_pos(const float* in, int gs, int D, int gh, int gw, float* out) {
  // in [gs][gs][D] -> out [gh][gw][D]; horizontal pass then vertical pass (ATen order)
  std::vector<int> xm, xs, ym, ys;
  std::vector<float> xw, yw;
  int kx, ky;
  aa_weights(gs, gw, xm, xs, xw, kx);
  aa_weights(gs, gh, ym, ys, yw, ky);
  std::vector<float> tmp((size_t)gs * gw * D);
  for (int y = 0; y < gs; ++y)
    for (int x = 0; x < gw; ++x) {
      float* o = &tmp[((size_t)y * gw + x) * D];
      for (int c = 0; c < D; ++c) o[c] = 0.f;
      for (int j = 0; j < xs[x]; ++j) {
        const float w = xw[(size_t)x * kx + j];
        const float* s = in + ((size_t)y * gs + xm[x] + j) * D;
        for (int c = 0; c < D; ++c) o[c] += w * s[c];
      }
    }
  for (int y = 0; y < gh; ++y)
    for (int x = 0; x < gw; ++x) {
      float* o = out + ((size_t)y * gw + x) * D;
      for (int c = 0; c < D; ++c) o[c] = 0.f;
      for (int j = 0; j < ys[y]; ++j) {
        const float w = yw[(size_t)y * ky + j];
        const float* s = &tmp[((size_t)(ym[y] + j) * gw + x) * D];
        for (int c = 0; c < D; ++c) o[c] += w * s[c];
      }
    }
}

extern "C" int wm_set_tuning(const char* key, int value) {
  static const char* keys[WM_TUNE_COUNT] = {"gemm_cfg", "gemm_pp", "gemm_mfma16", "attn_qb", "op_ldpad", "attn_splits", "conv_fuse_up", "conv_narrow", "conv_bn", "conv_rs", "lin_mfma", "conv_tpx", "attn_tail", "gemm_group", "comm_overlap", "heads_concurrent", "rcu_mid16", "gemm_sched", "force_gather", "attn_op_policy", "comm_p2p", "ln_rpw", "ln_fuse", "heads_main", "conv_gemm", "resid_prefetch", "up1_gather", "tconv", "up1_comp"};
  for (int i = 0; i < WM_TUNE_COUNT; ++i)
    if (key && strcmp(key, keys[i]) == 0) { wm_tuning[i] = value; return 0; }
  return -1;
}

extern "C" void wm_host_to_16(const float* in, uint16_t* out, size_t n, int dtype) {
  for (size_t i = 0; i < n; ++i) out[i] = h_to16(in[i], dtype);
}

// ====================================================================================== lifecycle
extern "C" wm_status wm_create(const wm_config* cfg, int device, wm_handle** out) {
  if (!cfg || !out) return WM_ERR_INVALID;
  wm_handle* h = new wm_handle();
  h->cfg = *cfg;
  h->device = device;
  *out = h;
  if (cfg->embed_dim / cfg->num_heads != 64 || cfg->embed_dim / cfg->dino_heads != 64)
    return fail(h, WM_ERR_INVALID, "backbone head_dim must be 64 (reference: 1024/16)");
  if (cfg->embed_dim % 64) return fail(h, WM_ERR_INVALID, "embed_dim must be a multiple of 64");
  if (hipSetDevice(device) != hipSuccess) return fail(h, WM_ERR_HIP, "hipSetDevice failed");
  return WM_OK;
}

extern "C" void wm_destroy(wm_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  for (auto& kv : h->tconv) free_tconv(kv.second);
  for (auto& kv : h->up1comp) free_up1comp(kv.second);
  for (auto& kv : h->w) {
    if (!kv.second.owned) continue;
    if (kv.second.f32) (void)hipFree(kv.second.f32);
    if (kv.second.w16) (void)hipFree(kv.second.w16);
  }
  if (h->arena && h->arena_owned) (void)hipFree(h->arena);
  for (int i = 0; i < 4; ++i) {
    if (h->hstream[i]) (void)hipStreamDestroy(h->hstream[i]);
    if (h->hjoin[i]) (void)hipEventDestroy(h->hjoin[i]);
  }
  if (h->att_stat_host) (void)hipHostFree(h->att_stat_host);
  if (h->hfork) (void)hipEventDestroy(h->hfork);
  if (h->camstream) (void)hipStreamDestroy(h->camstream);
  if (h->camjoin) (void)hipEventDestroy(h->camjoin);
  if (h->cstream) (void)hipStreamDestroy(h->cstream);
  if (h->cfork) (void)hipEventDestroy(h->cfork);
  if (h->cjoin) (void)hipEventDestroy(h->cjoin);
  for (int k = 0; k < wm_handle::NKIND; ++k)
    for (auto& e : h->ev[k]) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  if (h->comm.kind == 1 && h->comm.nccl) ncclCommDestroy(h->comm.nccl);
  delete h;
}

extern "C" const char* wm_last_error(const wm_handle* h) { return h ? h->err.c_str() : "null handle"; }

extern "C" wm_status wm_set_weight(wm_handle* h, const char* name, const float* host, const int64_t* shape, int ndim) {
  if (!h || !name || !host || !shape) return WM_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  const std::string n(name);
  // strict=False: names outside the spec are ignored
  auto& spec = h->spec;
  if (spec.empty())
    for (auto& kv : param_spec(h->cfg)) spec[kv.first] = kv.second;
  auto it = spec.find(n);
  if (it == spec.end()) return WM_OK;
  std::vector<int64_t> sh(shape, shape + ndim);
  if (sh != it->second) return fail(h, WM_ERR_INVALID, "shape mismatch for " + n);
  size_t numel = 1;
  for (auto s : sh) numel *= (size_t)s;
  Weight& w = h->w[n];
  h->plan_n = -1;  // weight-derived workspace tables (resampled pos_embed, camera init token) are rebuilt by the next wm_reserve
  h->tconv_valid = false;
  if (w.owned) {
    if (w.f32) (void)hipFree(w.f32);
    if (w.w16) (void)hipFree(w.w16);
  }
  w.f32 = nullptr; w.w16 = nullptr; w.owned = true; w.tap_major = false;
  w.shape = sh;
  w.set = true;
  const WKind k = classify(n, ndim);
  if (k == WK_F32) {
    HIPCHK(h, hipMalloc((void**)&w.f32, std::max<size_t>(numel, 4) * 4));
    HIPCHK(h, hipMemcpy(w.f32, host, numel * 4, hipMemcpyHostToDevice));
    if (ends_with(n, "pos_embed") || ends_with(n, "init_token")) w.host.assign(host, host + numel);
    if (n.find(".resize_layers.0.bias") != std::string::npos || n.find(".resize_layers.1.bias") != std::string::npos ||
        ends_with(n, ".scratch.refinenet1.out_conv.bias"))
      w.host.assign(host, host + numel);
    return WM_OK;
  }
  // the ConvTranspose / layer_rn pairs that wm_reserve composes into one token-resolution GEMM (build_tconv) are kept in fp32 on the host
  if (n.find(".resize_layers.0.weight") != std::string::npos || n.find(".resize_layers.1.weight") != std::string::npos ||
      ends_with(n, ".scratch.layer1_rn.weight") || ends_with(n, ".scratch.layer2_rn.weight") ||
      ends_with(n, ".scratch.refinenet1.out_conv.weight") || ends_with(n, ".scratch.output_conv1.weight"))   // (+ build_up1comp)
    w.host.assign(host, host + numel);
  const int dt = k == WK_LIN16_BACKBONE ? h->cfg.backbone_dtype : h->cfg.head_dtype;
  std::vector<uint16_t> r;
  if (k == WK_LIN16_BACKBONE || k == WK_LIN16_HEAD) {
    const int N = (int)sh[0];
    const int K = (int)(numel / N);
    const int Kp = ru(K, 64);
    r.assign((size_t)N * Kp, 0);
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < K; ++j) r[(size_t)i * Kp + j] = h_to16(host[(size_t)i * K + j], dt);
    w.k16 = Kp;
  } else if (k == WK_CONV16_HEAD) {  // [Cout][Cin][kh][kw] -> [Cout][kh][kw][Cin]
    const int Co = (int)sh[0], Ci = (int)sh[1], kh = (int)sh[2], kw = (int)sh[3];
    r.resize(numel);
    for (int o = 0; o < Co; ++o)
      for (int c = 0; c < Ci; ++c)
        for (int y = 0; y < kh; ++y)
          for (int x = 0; x < kw; ++x)
            r[(((size_t)o * kh + y) * kw + x) * Ci + c] = h_to16(host[(((size_t)o * Ci + c) * kh + y) * kw + x], dt);
    w.k16 = kh * kw * Ci;
    if (kh == 3 && kw == 3 && ends_with(n, "output_conv1.weight")) {
      // a second, tap-major copy [tap][Cout][Cin] behind the first: the B operand of the low-resolution tap GEMM (upconv.hip)
      r.resize(2 * numel);
      for (int o = 0; o < Co; ++o)
        for (int t = 0; t < 9; ++t)
          for (int c = 0; c < Ci; ++c) r[numel + ((size_t)t * Co + o) * Ci + c] = r[((size_t)o * 9 + t) * Ci + c];
      w.tap_major = true;
    }
  } else {  // ConvTranspose2d [Cin][Cout][k][k] -> rows (i*k + j)*Cout + co, K = Cin
    const int Ci = (int)sh[0], Co = (int)sh[1], kk = (int)sh[2];
    r.resize(numel);
    for (int c = 0; c < Ci; ++c)
      for (int o = 0; o < Co; ++o)
        for (int i = 0; i < kk; ++i)
          for (int j = 0; j < kk; ++j)
            r[((size_t)(i * kk + j) * Co + o) * Ci + c] = h_to16(host[(((size_t)c * Co + o) * kk + i) * kk + j], dt);
    w.k16 = Ci;
  }
  HIPCHK(h, hipMalloc(&w.w16, r.size() * 2));
  HIPCHK(h, hipMemcpy(w.w16, r.data(), r.size() * 2, hipMemcpyHostToDevice));
  return WM_OK;
}

// The value a parameter has in a freshly constructed reference model, where that value is deterministic: what a tensor
// missing from the checkpoint keeps under load_state_dict(strict=False) (huggingface_hub mixin, worldmirror.py:13,16).
// LayerNorm weight 1 / bias 0 (torch default); LayerScale gamma = init_values: 1.0 in the DINOv2 encoder
// (visual_transformer.py:152-160), 0.01 in the multi-view blocks and the camera trunk (visual_transformer.py:65,
// camera_head.py:24).  Everything else (Linear / conv weights and biases, learned tokens) is RANDOM in the reference
// (trunc_normal / kaiming-uniform / normal(1e-6)): not reproducible, filled with 0 and reported through wm_missing_name.
static float init_value_of(const std::string& n) {
  const bool norm = n.find(".norm1.") != std::string::npos || n.find(".norm2.") != std::string::npos || n.find(".norm.") != std::string::npos ||
                    n.find("q_norm.") != std::string::npos || n.find("k_norm.") != std::string::npos || n.find("token_norm.") != std::string::npos ||
                    n.find("out_norm.") != std::string::npos;
  if (norm) return ends_with(n, ".weight") ? 1.0f : 0.0f;
  if (ends_with(n, ".gamma")) return n.find(".patch_embed.") != std::string::npos ? 1.0f : 0.01f;
  return 0.0f;
}

extern "C" wm_status wm_finalize_weights(wm_handle* h, int* missing) {
  if (!h) return WM_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  h->missing.clear();
  for (auto& kv : param_spec(h->cfg)) {
    auto it = h->w.find(kv.first);
    if (it != h->w.end() && it->second.set) continue;
    h->missing.push_back(kv.first);  // strict=False: a missing tensor keeps its init value (see init_value_of)
    size_t numel = 1;
    for (auto s : kv.second) numel *= (size_t)s;
    std::vector<float> z(numel, init_value_of(kv.first));
    wm_status st = wm_set_weight(h, kv.first.c_str(), z.data(), kv.second.data(), (int)kv.second.size());
    if (st != WM_OK) return st;
    h->w[kv.first].set = false;
  }
  if (missing) *missing = (int)h->missing.size();
  h->finalized = true;
  return WM_OK;
}

extern "C" wm_status wm_share_weights(wm_handle* dst, const wm_handle* src) {
  if (!dst || !src || dst == src) return WM_ERR_INVALID;
  if (!src->finalized) return fail(dst, WM_ERR_STATE, "source weights not finalized");
  if (memcmp(&dst->cfg, &src->cfg, sizeof(wm_config)) != 0 || dst->device != src->device)
    return fail(dst, WM_ERR_INVALID, "wm_share_weights: configuration or device differs");
  for (auto& kv : dst->w) {
    if (!kv.second.owned) continue;
    if (kv.second.f32) (void)hipFree(kv.second.f32);
    if (kv.second.w16) (void)hipFree(kv.second.w16);
  }
  dst->w = src->w;
  for (auto& kv : dst->w) kv.second.owned = false;
  dst->spec = src->spec;
  dst->missing = src->missing;
  dst->finalized = true;
  dst->plan_n = -1;
  dst->tconv_valid = false;
  return WM_OK;
}

extern "C" const char* wm_missing_name(const wm_handle* h, int i) {
  return (h && i >= 0 && (size_t)i < h->missing.size()) ? h->missing[i].c_str() : nullptr;
}

// ====================================================================================== workspace
extern "C" size_t wm_workspace_bytes(const wm_handle* h, int n_local, int n_total, int H, int W) {
  if (!h) return 0;
  const Dims d = make_dims(h, n_local, n_total, H, W);
  size_t tot = 0;
  for (auto& kv : arena_layout(h, d)) tot += kv.second;
  return tot;
}

namespace {

// Builds the workspace for a shape: arena (library-owned unless the caller gave one), buffer map and the shape- /
// weight-derived device tables.  Called by wm_reserve only — never from the forward pass.
wm_status plan(wm_handle* h, const Dims& d) {
  if (h->plan_n == d.n && h->plan_nt == d.nt && h->plan_H == d.H && h->plan_W == d.W) return WM_OK;
  const wm_config& c = h->cfg;
  auto L = arena_layout(h, d);
  size_t tot = 0;
  for (auto& kv : L) tot += kv.second;
  HIPCHK(h, hipDeviceSynchronize());  // a forward of the previous shape may still be using the arena
  if (tot > h->arena_bytes) {
    if (!h->arena_owned)
      return fail(h, WM_ERR_STATE, "caller workspace too small: " + std::to_string(h->arena_bytes) + " < " + std::to_string(tot) + " bytes (wm_workspace_bytes)");
    if (h->arena) HIPCHK(h, hipFree(h->arena));
    h->arena = nullptr;
    h->arena_bytes = 0;
    HIPCHK(h, hipMalloc((void**)&h->arena, tot));
    h->arena_bytes = tot;
  }
  h->buf.clear();
  size_t off = 0;
  for (auto& kv : L) { h->buf[kv.first] = h->arena + off; off += kv.second; }

  // RoPE tables (rope.py:80-111): angle = pos * 100^(-i/16), i < 16 (table is cat(ang, ang): index e % 16)
  {
    const int np = std::max(d.gh, d.gw) + 1;
    std::vector<float> cs((size_t)np * 16), sn((size_t)np * 16);
    for (int p = 0; p < np; ++p)
      for (int i = 0; i < 16; ++i) {
        const float expo = (float)(2 * i) / 32.0f;
        const float inv = 1.0f / std::pow(c.rope_freq, expo);
        const float ang = (float)p * inv;
        cs[(size_t)p * 16 + i] = std::cos(ang);
        sn[(size_t)p * 16 + i] = std::sin(ang);
      }
    HIPCHK(h, hipMemcpy(h->buf["rope_cos"], cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->buf["rope_sin"], sn.data(), sn.size() * 4, hipMemcpyHostToDevice));
  }
  {  // fallback statistics: counters at 0, host mirror allocated once (pinned: the forward copies into it without synchronising)
    const int n = (c.dino_depth + 2 * c.depth) * 3;
    if (h->att_stat_n != n) {
      if (h->att_stat_host) HIPCHK(h, hipHostFree(h->att_stat_host));
      h->att_stat_host = nullptr;
      HIPCHK(h, hipHostMalloc((void**)&h->att_stat_host, (size_t)n * 4, hipHostMallocDefault));
      h->att_stat_n = n;
    }
    memset(h->att_stat_host, 0, (size_t)n * 4);
    h->att_seen.assign(n, 0);
    h->att_general_ttl.assign(n, 0);
    HIPCHK(h, hipMemset(h->buf["ATT_STAT"], 0, (size_t)n * 4));
  }
  HIPCHK(h, hipMemset(h->buf["LN_SYNC"], 0, ((size_t)d.Mx / 16 + 2) * 3 * 4));
  // the fast attention kernels' sticky fallback hints start empty for a new shape (they index launch blocks)
  HIPCHK(h, hipMemset(h->buf["ATT_HINT"], 0, (size_t)(c.dino_depth + 2 * c.depth) * 3 * wm_attention_max_blocks((int)d.Mx, d.P < d.Td ? d.P : d.Td, d.heads) * 4));
  // DINO pos-embed for this grid (vision_transformer.py:175-207)
  {
    const Weight* pe = W(h, "visual_geometry_transformer.patch_embed.pos_embed");
    if (!pe || pe->host.empty()) return fail(h, WM_ERR_STATE, "pos_embed not set");
    const int g = c.img_size / c.patch_size, D = d.D;
    if (d.gh == g && d.gw == g) {
      HIPCHK(h, hipMemcpy(h->buf["dino_pos"], pe->host.data(), (size_t)(1 + d.hw) * D * 4, hipMemcpyHostToDevice));
    } else {
      std::vector<float> r((size_t)(1 + d.hw) * D);
      memcpy(r.data(), pe->host.data(), (size_t)D * 4);
      wm_host_resample_pos(pe->host.data() + D, g, D, d.gh, d.gw, r.data() + D);
      HIPCHK(h, hipMemcpy(h->buf["dino_pos"], r.data(), r.size() * 4, hipMemcpyHostToDevice));
    }
  }
  // DPT UV position tables (dense_head.py:253-263)
  {
    const double aspect = (double)d.W / (double)d.H;
    std::vector<float> tx, ty;
    for (int i = 0; i < 4; ++i) {
      const int C = c.dpt_out_channels[i];
      uv_tables(d.gw, d.gh, C, aspect, tx, ty);
      std::vector<float> full((size_t)d.hw * C);
      for (int y = 0; y < d.gh; ++y)
        for (int x = 0; x < d.gw; ++x)
          for (int k = 0; k < C; ++k)
            full[((size_t)y * d.gw + x) * C + k] = k < C / 2 ? tx[(size_t)x * (C / 2) + k] : ty[(size_t)y * (C / 2) + k - C / 2];
      HIPCHK(h, hipMemcpy(h->buf["dpt_pos" + std::to_string(i)], full.data(), full.size() * 4, hipMemcpyHostToDevice));
    }
    uv_tables(d.W, d.H, c.dpt_features / 2, aspect, tx, ty);
    HIPCHK(h, hipMemcpy(h->buf["dpt_posx"], tx.data(), tx.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->buf["dpt_posy"], ty.data(), ty.size() * 4, hipMemcpyHostToDevice));
    if (c.enable_gs) {
      uv_tables(d.W, d.H, c.gs_dim / 2, aspect, tx, ty);
      HIPCHK(h, hipMemcpy(h->buf["gs_posx"], tx.data(), tx.size() * 4, hipMemcpyHostToDevice));
      HIPCHK(h, hipMemcpy(h->buf["gs_posy"], ty.data(), ty.size() * 4, hipMemcpyHostToDevice));
    }
  }
  // the composed ConvTranspose -> layer_rn GEMMs of the DPT heads (weight-derived only: rebuilt after a weight change, not per shape)
  if (!h->tconv_valid) {
    for (auto& kv : h->tconv) free_tconv(kv.second);
  for (auto& kv : h->up1comp) free_up1comp(kv.second);
    h->tconv.clear();
    std::vector<std::string> heads;
    if (c.enable_pts) heads.push_back("pts_head.");
    if (c.enable_depth) heads.push_back("depth_head.");
    if (c.enable_norm) heads.push_back("norm_head.");
    if (c.dpt_features == 256)
      for (const std::string& p : heads)
        for (int i = 0; i < 2; ++i) {
          const Weight* wct = W(h, p + "resize_layers." + std::to_string(i) + ".weight");
          const Weight* bct = W(h, p + "resize_layers." + std::to_string(i) + ".bias");
          const Weight* wrn = W(h, p + "scratch.layer" + std::to_string(i + 1) + "_rn.weight");
          const int k = i == 0 ? 4 : 2, oc = c.dpt_out_channels[i];
          if (!wct || !bct || !wrn || wct->host.empty() || bct->host.empty() || wrn->host.empty() || oc % 64) continue;
          if (wct->host.size() != (size_t)oc * oc * k * k || bct->host.size() != (size_t)oc || wrn->host.size() != (size_t)256 * oc * 9) continue;
          HIPCHK(h, build_tconv(c.head_dtype, k, oc, oc, 256, wct->host.data(), bct->host.data(), wrn->host.data(), h->tconv[p + std::to_string(i)], nullptr));
        }
    h->up1comp.clear();
    if (c.head_dtype == WM_DT_F16)
      for (const std::string& p : heads) {
        const Weight* w1 = W(h, p + "scratch.output_conv1.weight");
        const Weight* wo = W(h, p + "scratch.refinenet1.out_conv.weight");
        const Weight* bo = W(h, p + "scratch.refinenet1.out_conv.bias");
        const int F_ = c.dpt_features;
        if (!w1 || !wo || !bo || w1->host.empty() || wo->host.empty() || bo->host.empty() || F_ % 64 || w1->shape.size() != 4) continue;
        const int Co = (int)w1->shape[0];
        if ((Co != 128 && Co != 64 && Co != 32) || w1->host.size() != (size_t)Co * F_ * 9 || wo->host.size() != (size_t)F_ * F_ || bo->host.size() != (size_t)F_) continue;
        HIPCHK(h, build_up1comp(c.head_dtype, F_, Co, w1->host.data(), wo->host.data(), bo->host.data(), h->up1comp[p], nullptr));
      }
    h->tconv_valid = true;
  }
  // camera init token broadcast [nt][12]
  if (c.enable_cam) {
    const Weight* it = W(h, "cam_head.init_token");
    std::vector<float> v((size_t)d.nt * 12, 0.f);
    if (it && !it->host.empty())
      for (int i = 0; i < d.nt; ++i)
        for (int k = 0; k < 9; ++k) v[(size_t)i * 12 + k] = it->host[k];
    HIPCHK(h, hipMemcpy(h->buf["cam_init"], v.data(), v.size() * 4, hipMemcpyHostToDevice));
  }
  h->plan_n = d.n; h->plan_nt = d.nt; h->plan_H = d.H; h->plan_W = d.W;
  return WM_OK;
}

// ---------------------------------------------------------------- profiling helpers
struct ProfScope {
  wm_handle* h; int kind; hipStream_t s; EvPair* e = nullptr;
  ProfScope(wm_handle* h_, int k, hipStream_t s_) : h(h_), kind(k), s(s_) {
    if (!h->prof) return;
    auto& v = h->ev[kind];
    if (h->ev_used[kind] == v.size()) {
      EvPair p;
      (void)hipEventCreate(&p.a); (void)hipEventCreate(&p.b);
      v.push_back(p);
    }
    e = &v[h->ev_used[kind]++];
    (void)hipEventRecord(e->a, s);
  }
  ~ProfScope() { if (e) (void)hipEventRecord(e->b, s); }
};

// ---------------------------------------------------------------- collective
wm_status comm_allgather(wm_handle* h, const void* send, void* recv, size_t bytes, hipStream_t s) {
  ProfScope ps(h, 12, s);   // timing kind 12: the collective, on the queue it runs on (the compute queue unless WM_COMM_OVERLAP=1)
  Comm& cm = h->comm;
  if (cm.kind == 1) {
    if (wm_tuning[WM_TUNE_COMM_P2P] == 1) {
      // Direct all-gather (opt-in, tuning comm_p2p = 1 / bench.py --gather p2p): every rank sends its chunk to every peer and
      // receives every peer's chunk as ONE group of point-to-point operations, so each of the 7 xGMI links of a GPU carries one
      // chunk in each direction at once (~0.29 ms per layer link-bound at C4, against ~2.1 ms if the all-gather
      // runs as a ring).  Not the default: no multi-GPU node was available to any build round, and RCCL may already pick a
      // direct algorithm for ncclAllGather on a fully connected node — the first 8-GPU run has to say.
      if (ncclGroupStart() != ncclSuccess) return fail(h, WM_ERR_COMM, "ncclGroupStart failed");
      for (int r = 0; r < cm.world; ++r) {
        if (r == cm.rank) continue;
        if (ncclSend(send, bytes, ncclInt8, r, cm.nccl, s) != ncclSuccess) return fail(h, WM_ERR_COMM, "ncclSend failed");
        if (ncclRecv((char*)recv + (size_t)r * bytes, bytes, ncclInt8, r, cm.nccl, s) != ncclSuccess) return fail(h, WM_ERR_COMM, "ncclRecv failed");
      }
      if (ncclGroupEnd() != ncclSuccess) return fail(h, WM_ERR_COMM, "ncclGroupEnd failed");
      HIPCHK(h, hipMemcpyAsync((char*)recv + (size_t)cm.rank * bytes, send, bytes, hipMemcpyDeviceToDevice, s));
      return WM_OK;
    }
    if (ncclAllGather(send, recv, bytes, ncclInt8, cm.nccl, s) != ncclSuccess) return fail(h, WM_ERR_COMM, "ncclAllGather failed");
    return WM_OK;
  }
  if (cm.kind == 2) {
    wm_local_group* g = cm.grp;
    g->recv[cm.rank] = recv;
    HIPCHK(h, hipStreamSynchronize(s));
    pthread_barrier_wait(&g->bar);
    for (int r = 0; r < g->world; ++r)
      HIPCHK(h, hipMemcpyAsync((char*)g->recv[r] + (size_t)cm.rank * bytes, send, bytes, hipMemcpyDeviceToDevice, s));
    HIPCHK(h, hipStreamSynchronize(s));
    pthread_barrier_wait(&g->bar);
    return WM_OK;
  }
  return fail(h, WM_ERR_COMM, "sharded forward without a communicator");
}

// ---------------------------------------------------------------- building blocks
struct Ctx {
  wm_handle* h; Dims d; hipStream_t s; int bdt, hdt;
  int attn_site = 0;   // attention call sites passed so far in this forward (index of the site's sticky fallback hints, ATT_HINT)
};

#define LCHK(c, e)                                                                                            \
  do {                                                                                                        \
    hipError_t _e = (e);                                                                                      \
    if (_e != hipSuccess)                                                                                     \
      return fail((c).h, WM_ERR_HIP, std::string(hipGetErrorString(_e)) + " at " + __FILE__ + ":" + std::to_string(__LINE__)); \
  } while (0)

wm_status gemm(Ctx& c, int dt, int epi, const void* A, int lda, const void* Wp, int ldw, void* C, int ldc, const float* bias,
               const float* gamma, int M, int N, int K, WmGemmArgs* extra = nullptr, int prof_kind = -1, bool* ln_fused = nullptr) {
  WmGemmArgs a;
  if (extra) a = *extra; else memset(&a, 0, sizeof(a));
  a.A = A; a.W = Wp; a.C = C; a.bias = bias; a.gamma = gamma;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.dtype = dt; a.epi = epi;
  // timing kinds by kernel instantiation: the three epilogues that carry the transformer blocks, the rest under 2
  ProfScope ps(c.h, prof_kind >= 0 ? prof_kind : epi == WM_EPI_QKV ? 5 : epi == WM_EPI_RESID ? 6 : epi == WM_EPI_GELU_T16 ? 7 : 2, c.s);
  const bool fuse = a.ln_out != nullptr && wm_gemm_fuses_ln(a);
  if (!fuse) a.ln_out = nullptr;
  if (ln_fused) *ln_fused = fuse;
  LCHK(c, wm_launch_gemm(a, c.s));
  if (fuse) LCHK(c, wm_launch_gemm_ln_fallback(a, c.s));   // bands whose rendezvous timed out (normally none: every block exits at entry)
  return WM_OK;
}

wm_status layernorm(Ctx& c, const float* x, int ld_in, void* y, int ld_out, const float* w, const float* b, int D, float eps,
                    int groups, int rpg, int in_group, int in_off, int out_group, int out_off, int out_f32, int dt) {
  WmLnArgs a;
  a.x = x; a.y = y; a.w = w; a.b = b; a.D = D; a.ld_in = ld_in; a.ld_out = ld_out; a.eps = eps;
  a.groups = groups; a.rows_per_group = rpg; a.in_group = in_group; a.in_off = in_off; a.out_group = out_group; a.out_off = out_off;
  a.out_f32 = out_f32; a.dtype = dt;
  LCHK(c, wm_launch_layernorm(a, c.s));
  return WM_OK;
}

// Block.forward (block.py:72-93) on the fp32 residual stream X [M][D]; seq_len = attention span.
// tap_half: when non-null, the block's output (the value its last GEMM stores to X) is also written to tap_half[row * 2D + col]
// by that GEMM's epilogue — the cat([frame_out, global_out], -1) of visual_transformer.py:337-339 without a copy pass.
// next_norm1: name prefix of the block that follows on the same stream X ("" = none): its norm1 is then computed by this block's fc2
// epilogue when the shape allows (wm_gemm_fuses_ln) and *ln1_ready tells the caller to pass ln1_done to that block.
wm_status backbone_block(Ctx& c, const std::string& p, float* X, int M, int seq_len, int heads, float eps, bool qk_norm,
                         bool rope, int tokens_per_view, int patch_start, bool is_global, float* tap_half = nullptr,
                         bool ln1_done = false, const std::string& next_norm1 = std::string(), bool* ln1_ready = nullptr) {
  wm_handle* h = c.h;
  const Dims& d = c.d;
  const int D = d.D, dt = c.bdt;
  uint16_t* A16 = B<uint16_t>(h, "A16");
  char* QKV16 = B<char>(h, "QKV16");
  const size_t hsz = (size_t)M * D * 2;  // bytes of one of Q/K/V
  void* Q16 = QKV16; void* K16 = QKV16 + hsz; void* V16 = QKV16 + 2 * hsz;
  void* O16 = B<void>(h, "O16");
  void* H16 = B<void>(h, "H16");
  wm_status st;
  if (ln1_ready) *ln1_ready = false;
  if (!ln1_done) {
    st = layernorm(c, X, D, A16, D, F(h, p + "norm1.weight"), F(h, p + "norm1.bias"), D, eps, 1, M, 0, 0, 0, 0, 0, dt);
    if (st) return st;
  }
  // the residual GEMMs carry the LayerNorm that follows them when the launch allows it (gemm.hip epilogue_resid_ln)
  auto fuse_args = [&](WmGemmArgs& ex, const float* w, const float* b) {
    ex.ln_out = A16; ex.ln_ld = D; ex.ln_w = w; ex.ln_b = b; ex.ln_eps = eps;
    ex.ln_stats = B<float>(h, "LN_STATS");
    int* sync = B<int>(h, "LN_SYNC");
    ex.ln_sync = sync; ex.ln_fallback = sync + ((size_t)d.Mx / 16 + 2) * 2;
  };
  {  // QKV projection with q/k-norm + RoPE + head-major relayout fused into the epilogue (attention.py:50-56)
    WmGemmArgs ex;
    memset(&ex, 0, sizeof(ex));
    WmQkvArgs& a = ex.qkv;
    a.q = Q16; a.k = K16; a.v = V16;
    if (qk_norm) {
      a.qn_w = F(h, p + "attn.q_norm.weight"); a.qn_b = F(h, p + "attn.q_norm.bias");
      a.kn_w = F(h, p + "attn.k_norm.weight"); a.kn_b = F(h, p + "attn.k_norm.bias");
    }
    if (rope) { a.rope_cos = B<float>(h, "rope_cos"); a.rope_sin = B<float>(h, "rope_sin"); }
    a.M = M; a.H = heads; a.head_stride = M; a.tokens_per_view = tokens_per_view; a.patch_start = patch_start; a.grid_w = d.gw;
    a.q_scale = 0.125f * 1.4426950408889634f; a.dtype = dt;  // 1/sqrt(64) * log2(e): the attention kernel works in base 2
    st = gemm(c, dt, WM_EPI_QKV, A16, D, W16(h, p + "attn.qkv.weight"), D, nullptr, 0, F(h, p + "attn.qkv.bias"), nullptr, M, 3 * D, D, &ex);
    if (st) return st;
  }
  {
    WmAttnArgs a;
    memset(&a, 0, sizeof(a));
    a.Q = Q16; a.O = O16; a.H = heads; a.q_rows = M; a.q_head_stride = M; a.dtype = dt;
    const bool force_gather = wm_tuning[WM_TUNE_FORCE_GATHER] > 0;  // 1-rank test of the collective path (tests/test_gpu_sharded.py)
    // Opt-in (WM_COMM_OVERLAP=1 / tuning comm_overlap = 1; default: the gather on the compute queue): the overlapped form is
    // exercised by 8 in-process ranks on one GPU (tests/test_gpu_fullsize.py), but RCCL has not run it on real links yet
    // (no multi-GPU node was available to any round): the simpler event-free path is the default until one 8-GPU run of both,
    // compared bit for bit, is on record.  The second queue is safe here: between fork and join the compute queue runs only
    // the attention kernels, which contain no packed-fp32 instruction (the hazard described at the DPT heads below needs one;
    // tests/test_kernel_resources_cpu.py disassembles them)
    static const bool overlap_env = [] { const char* e = wm_env("WM_COMM_OVERLAP"); return e && atoi(e) != 0; }();
    a.part_o = B<float>(h, "ATT_PO"); a.part_ml = B<float>(h, "ATT_ML"); a.max_splits = WM_ATTN_MAX_SPLITS;
    a.unit_flags = B<int>(h, "ATT_FLAGS");
    const size_t hint_n = wm_attention_max_blocks((int)d.Mx, d.P < d.Td ? d.P : d.Td, d.heads);
    const int site = c.attn_site++;
    int* const hint0 = B<int>(h, "ATT_HINT") + (size_t)site * 3 * hint_n;
    int* const stat0 = B<int>(h, "ATT_STAT") + (size_t)site * 3;
    a.unit_hint = hint0; a.unit_stat = stat0;
    bool general_only = false;
    if (site * 3 < h->att_stat_n) {
      // host policy from the (asynchronously mirrored, possibly one or two forwards old) counter: a quarter of the site's fast-kernel
      // units on the general kernel since the last look -> general kernel only for the next WM_ATTN_HINT_TTL calls
      const int cur = h->att_stat_host[site * 3], delta = cur - h->att_seen[site * 3];
      h->att_seen[site * 3] = cur;
      const long units = (long)((seq_len + 511) / 512) * (M / (seq_len > 0 ? seq_len : 1)) * heads;
      if (delta > 0 && (long)delta * 4 >= units) h->att_general_ttl[site * 3] = WM_ATTN_HINT_TTL;
      if (h->att_general_ttl[site * 3] > 0) { --h->att_general_ttl[site * 3]; general_only = true; }
    }
    const bool sharded = is_global && (d.world > 1 || (force_gather && h->comm.kind != 0));
    const int ntpc = (M + 63) / 64;  // key tiles per rank chunk
    const bool overlap = wm_tuning[WM_TUNE_COMM_OVERLAP] >= 0 ? wm_tuning[WM_TUNE_COMM_OVERLAP] != 0 : overlap_env;
    // every launch argument set of the overlapped form is checked BEFORE the fork: a rank that returned between the fork and
    // the collective would leave its peers waiting in the all-gather
    bool overlap_ok = sharded && overlap && d.world > 1 && ntpc >= 16;
    if (overlap_ok) {
      WmAttnArgs t = a;
      t.force_partial = 1; t.K = K16; t.V = V16; t.seq_len = M; t.kv_head_stride = M; t.kv_chunks = 1;
      overlap_ok = wm_attention_variant(t) >= 0;
      t.kv_chunks = d.world - 1 > 1 ? d.world - 1 : 2; t.kv_rows_per_chunk = M; t.kv_chunk_stride = (long long)(2 * hsz / 2);
      overlap_ok = overlap_ok && wm_attention_variant(t) >= 0;
    }
    if (overlap_ok) {
      // ---- K/V all-gather UNDER the attention over this rank's own keys: the collective runs on the handle's
      // communication queue as soon as the QKV epilogue has written K|V; the compute queue meanwhile attends the local chunk
      // (1 / world of the keys: about the gather's own duration at 8 ranks), then the remote chunks — the gathered buffer's
      // ranges [0, rank) and (rank, world) — and one combine pass over all partial slots.  Softmax partials make the order
      // of the three key ranges irrelevant (attention.hip attn_combine_kernel).
      if (!h->cstream) {
        LCHK(c, hipStreamCreateWithFlags(&h->cstream, hipStreamNonBlocking));
        LCHK(c, hipEventCreateWithFlags(&h->cfork, hipEventDisableTiming));
        LCHK(c, hipEventCreateWithFlags(&h->cjoin, hipEventDisableTiming));
      }
      char* KVG = B<char>(h, "KVG");
      const int rank = h->comm.rank, world = d.world;
      LCHK(c, hipEventRecord(h->cfork, c.s));
      // slice counts: minimise rounds x tiles per block over the resident slots, per launch (uniform slices only)
      static const int ncu = [] { hipDeviceProp_t pr; int dv = 0; (void)hipGetDevice(&dv); return hipGetDeviceProperties(&pr, dv) == hipSuccess ? pr.multiProcessorCount : 256; }();
      // Slice count of one piecewise launch over `ntiles` key tiles: the unit size and residency are those of the kernel THAT launch
      // takes (with one view per rank the local launch is a 256-row-unit kernel and the remote ones attn_v4 with 512-row units at
      // one block per CU), so each launch asks wm_attention_geometry with its own arguments
      auto pick = [&](int chunks, int smax) {
        WmAttnArgs g = a;
        g.force_partial = 1; g.seq_len = M; g.kv_head_stride = M;
        if (chunks <= 1) { g.K = K16; g.V = V16; g.kv_chunks = 1; }
        else { g.K = B<char>(h, "KVG"); g.V = g.K; g.kv_chunks = chunks; g.kv_rows_per_chunk = M; g.kv_chunk_stride = (long long)(2 * hsz / 2); }
        int unit_rows = 256, per_cu = 2;
        wm_attention_geometry(g, &unit_rows, &per_cu);
        const int ntiles = (chunks < 1 ? 1 : chunks) * ntpc;
        const long slots = (long)per_cu * ncu, units = (long)((M + unit_rows - 1) / unit_rows) * heads;
        int best = 1; long best_cost = -1;
        for (int S = 1; S <= smax && S * 8 <= ntiles; ++S) {
          const long cost = ((units * S + slots - 1) / slots) * ((ntiles + S - 1) / S) + 6L * S;  // + partial write / read per slice
          if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = S; }
        }
        return best;
      };
      const int nb = rank, nc = world - 1 - rank;  // remote chunks before / after the own one
      const int sa = pick(1, 2);
      const int sb = nb > 0 ? pick(nb, nc > 0 ? 3 : 6) : 0, sc = nc > 0 ? pick(nc, WM_ATTN_MAX_SPLITS - sa - sb) : 0;
      a.force_partial = 1;
      // (1) local keys, while the gather is in flight
      a.K = K16; a.V = V16; a.seq_len = M; a.kv_head_stride = M; a.kv_chunks = 1; a.kv_chunk_stride = 0; a.kv_rows_per_chunk = 0;
      a.kv_splits = sa; a.part_slot0 = 0;
      {
        ProfScope ps(h, 0, c.s);
        LCHK(c, wm_launch_attention(a, c.s));
      }
      LCHK(c, hipStreamWaitEvent(h->cstream, h->cfork, 0));
      st = comm_allgather(h, K16, KVG, 2 * hsz, h->cstream);
      if (st) return st;
      LCHK(c, hipEventRecord(h->cjoin, h->cstream));
      LCHK(c, hipStreamWaitEvent(c.s, h->cjoin, 0));
      // (2) remote keys: gathered chunks [0, rank) and (rank, world)
      ProfScope ps(h, 0, c.s);
      a.seq_len = M; a.kv_head_stride = M; a.kv_chunk_stride = (long long)(2 * hsz / 2); a.kv_rows_per_chunk = M;
      if (nb > 0) {
        a.K = KVG; a.V = KVG + hsz; a.kv_chunks = nb; a.kv_splits = sb; a.part_slot0 = sa; a.unit_hint = hint0 + hint_n; a.unit_stat = stat0 + 1;
        LCHK(c, wm_launch_attention(a, c.s));
      }
      if (nc > 0) {
        char* base = KVG + (size_t)(rank + 1) * 2 * hsz;
        a.K = base; a.V = base + hsz; a.kv_chunks = nc; a.kv_splits = sc; a.part_slot0 = sa + sb; a.unit_hint = hint0 + 2 * hint_n; a.unit_stat = stat0 + 2;
        LCHK(c, wm_launch_attention(a, c.s));
      }
      LCHK(c, wm_launch_attention_combine(a, sa + sb + sc, c.s));
    } else {
      if (sharded) {
        // K and V are adjacent: one all-gather of [K|V] per layer -> [world][2][H][M][64]
        char* KVG = B<char>(h, "KVG");
        st = comm_allgather(h, K16, KVG, 2 * hsz, c.s);
        if (st) return st;
        a.K = KVG; a.V = KVG + hsz; a.seq_len = M; a.kv_head_stride = M; a.kv_chunks = d.world;
        a.kv_chunk_stride = (long long)(2 * hsz / 2); a.kv_rows_per_chunk = M;
      } else {
        a.K = K16; a.V = V16; a.seq_len = seq_len; a.kv_head_stride = M; a.kv_chunks = 1; a.kv_chunk_stride = 0; a.kv_rows_per_chunk = 0;
      }
      if (general_only) { a.unit_flags = nullptr; a.unit_hint = nullptr; a.unit_stat = nullptr; }
      ProfScope ps(h, is_global ? 0 : 1, c.s);
      LCHK(c, wm_launch_attention(a, c.s));
    }
  }
  {
    WmGemmArgs ex;
    memset(&ex, 0, sizeof(ex));
    fuse_args(ex, F(h, p + "norm2.weight"), F(h, p + "norm2.bias"));
    bool fused = false;
    st = gemm(c, dt, WM_EPI_RESID, O16, D, W16(h, p + "attn.proj.weight"), D, X, D, F(h, p + "attn.proj.bias"), F(h, p + "ls1.gamma"), M, D, D, &ex, -1, &fused);
    if (st) return st;
    if (!fused) {
      st = layernorm(c, X, D, A16, D, F(h, p + "norm2.weight"), F(h, p + "norm2.bias"), D, eps, 1, M, 0, 0, 0, 0, 0, dt);
      if (st) return st;
    }
  }
  const int Hd = c.h->cfg.mlp_ratio * D;
  st = gemm(c, dt, WM_EPI_GELU_T16, A16, D, W16(h, p + "mlp.fc1.weight"), D, H16, Hd, F(h, p + "mlp.fc1.bias"), nullptr, M, Hd, D);
  if (st) return st;
  WmGemmArgs ex;
  memset(&ex, 0, sizeof(ex));
  ex.C2 = tap_half; ex.ldc2 = 2 * D;
  bool fused = false;
  if (!next_norm1.empty()) fuse_args(ex, F(h, next_norm1 + "norm1.weight"), F(h, next_norm1 + "norm1.bias"));
  st = gemm(c, dt, WM_EPI_RESID, H16, Hd, W16(h, p + "mlp.fc2.weight"), Hd, X, D, F(h, p + "mlp.fc2.bias"), F(h, p + "ls2.gamma"), M, D, Hd, &ex, -1, &fused);
  if (ln1_ready) *ln1_ready = fused;
  return st;
}

wm_status linear32(Ctx& c, const float* X, int ldx, const std::string& name, float* Y, int ldy, int M, int pre, int post,
                   const float* gamma = nullptr, int accumulate = 0) {
  const Weight* w = W(c.h, name + ".weight");
  if (!w || !w->f32) return fail(c.h, WM_ERR_STATE, "missing fp32 weight " + name);
  const int N = (int)w->shape[0];
  int K = 1;
  for (size_t i = 1; i < w->shape.size(); ++i) K *= (int)w->shape[i];
  LCHK(c, wm_launch_linear_f32(X, w->f32, F(c.h, name + ".bias"), Y, M, N, K, ldx, ldy, pre, post, gamma, accumulate, c.s));
  return WM_OK;
}

// CameraHead.forward (camera_head.py:58-104) on all nt views, fp32
wm_status camera_head(Ctx& c, float* out_params) {
  wm_handle* h = c.h;
  const Dims& d = c.d;
  const wm_config& cf = h->cfg;
  const int D2 = 2 * d.D, S = d.nt;
  const std::string ch = "cam_head.";
  float* tok = B<float>(h, "cam_tok");
  wm_status st;
  // token_norm on tap3[:, :, 0]
  float* tok_local = d.world > 1 ? B<float>(h, "cam_tok_local") : tok;
  st = layernorm(c, B<float>(h, "tap3"), D2, tok_local, D2, F(h, ch + "token_norm.weight"), F(h, ch + "token_norm.bias"), D2, 1e-5f,
                 d.n, 1, d.P, 0, 1, 0, 1, 0);
  if (st) return st;
  if (d.world > 1) {
    st = comm_allgather(h, tok_local, tok, (size_t)d.n * D2 * 4, c.s);
    if (st) return st;
  }
  float *e = B<float>(h, "cam_e"), *mod = B<float>(h, "cam_mod"), *hh = B<float>(h, "cam_h"), *a = B<float>(h, "cam_a");
  float *qkv = B<float>(h, "cam_qkv"), *o = B<float>(h, "cam_o"), *f = B<float>(h, "cam_f");
  float *pred = B<float>(h, "cam_pred"), *delta = B<float>(h, "cam_delta"), *init = B<float>(h, "cam_init");
  LCHK(c, hipMemsetAsync(pred, 0, (size_t)S * 12 * 4, c.s));
  LCHK(c, hipMemsetAsync(delta, 0, (size_t)S * 12 * 4, c.s));
  for (int step = 0; step < cf.cam_steps; ++step) {
    st = linear32(c, step == 0 ? init : pred, 12, ch + "param_embed", e, D2, S, 0, 0);
    if (st) return st;
    st = linear32(c, e, D2, ch + "adapt_norm_gen.1", mod, 3 * D2, S, 1, 0);
    if (st) return st;
    LCHK(c, wm_launch_adaln(tok, mod, hh, S, D2, 1e-6f, c.s));
    for (int b = 0; b < cf.cam_trunk_depth; ++b) {
      const std::string p = ch + "refine_net." + std::to_string(b) + ".";
      st = layernorm(c, hh, D2, a, D2, F(h, p + "norm1.weight"), F(h, p + "norm1.bias"), D2, 1e-5f, 1, S, 0, 0, 0, 0, 1, 0);
      if (st) return st;
      st = linear32(c, a, D2, p + "attn.qkv", qkv, 3 * D2, S, 0, 0);
      if (st) return st;
      LCHK(c, wm_launch_small_attention(qkv, o, S, cf.cam_heads, D2 / cf.cam_heads, c.s));
      st = linear32(c, o, D2, p + "attn.proj", hh, D2, S, 0, 0, F(h, p + "ls1.gamma"), 1);
      if (st) return st;
      st = layernorm(c, hh, D2, a, D2, F(h, p + "norm2.weight"), F(h, p + "norm2.bias"), D2, 1e-5f, 1, S, 0, 0, 0, 0, 1, 0);
      if (st) return st;
      st = linear32(c, a, D2, p + "mlp.fc1", f, 4 * D2, S, 0, 2);
      if (st) return st;
      st = linear32(c, f, 4 * D2, p + "mlp.fc2", hh, D2, S, 0, 0, F(h, p + "ls2.gamma"), 1);
      if (st) return st;
    }
    st = layernorm(c, hh, D2, a, D2, F(h, ch + "out_norm.weight"), F(h, ch + "out_norm.bias"), D2, 1e-5f, 1, S, 0, 0, 0, 0, 1, 0);
    if (st) return st;
    st = linear32(c, a, D2, ch + "param_predictor.fc1", f, D2 / 2, S, 0, 2);
    if (st) return st;
    st = linear32(c, f, D2 / 2, ch + "param_predictor.fc2", delta, 12, S, 0, 0);
    if (st) return st;
    LCHK(c, wm_launch_cam_update(pred, delta, out_params, S, step == 0, c.s));
  }
  return WM_OK;
}

// ---- token-conv: Conv2d(3x3, pad 1, no bias) o ConvTranspose2d(kernel = stride = k) composed at the token resolution (WmGemmArgs::tc_k)
// For output phase (a, b) of token (i, j) the 3x3 taps land in at most two token rows and two token columns:
//   rn[(k i + a, k j + b)] = sum over neighbours (di, dj) of M[a, b, di, dj] p[i + di][j + dj]  +  (sum over taps inside the image of W_rn[tap]) b_ct
//   M[a, b, di, dj] = sum over taps (dy, dx) with floor((a + dy) / k) = di, floor((b + dx) / k) = dj of W_rn[tap] W_ct[(a + dy) mod k][(b + dx) mod k]
// 36 (phase, neighbour) matrices for k = 4, 16 for k = 2, instead of k^2 x 9 tap products per token: 4.4x / 2.7x fewer flops than ConvTranspose
// GEMM + 3x3 conv, no ConvTranspose output tensor, one rounding of the combined weight instead of two operand roundings.
// wct: torch ConvTranspose2d weight [Cin][Cm][k][k], bct [Cm], wrn: torch Conv2d weight [F][Cm][3][3] (host fp32).  F must be 256.
static hipError_t build_tconv(int dt, int k, int Cin, int Cm, int F_, const float* wct, const float* bct, const float* wrn, TconvPack& out, hipStream_t s) {
  if (F_ != 256 || (k != 2 && k != 4) || Cin % 64 || Cm % 4) return hipErrorInvalidValue;
  free_tconv(out);
  out.k = k; out.cin = Cin;
  const int k2 = k * k;
  std::vector<float> a_tap((size_t)9 * F_ * Cm), b_ph((size_t)k2 * Cin * Cm);
  for (int f = 0; f < F_; ++f)
    for (int c = 0; c < Cm; ++c)
      for (int t = 0; t < 9; ++t) a_tap[((size_t)t * F_ + f) * Cm + c] = wrn[((size_t)f * Cm + c) * 9 + t];
  for (int ci = 0; ci < Cin; ++ci)
    for (int c = 0; c < Cm; ++c)
      for (int ph = 0; ph < k2; ++ph) b_ph[((size_t)ph * Cin + ci) * Cm + c] = wct[((size_t)ci * Cm + c) * k2 + ph];
  float *d_a = nullptr, *d_b = nullptr, *d_bct = nullptr, *d_m = nullptr;
  hipError_t e;
#define TC(x) do { e = (x); if (e != hipSuccess) goto done; } while (0)
  TC(hipMalloc((void**)&d_a, a_tap.size() * 4));
  TC(hipMalloc((void**)&d_b, b_ph.size() * 4));
  TC(hipMalloc((void**)&d_bct, (size_t)Cm * 4));
  TC(hipMalloc((void**)&d_m, (size_t)F_ * Cin * 4));
  TC(hipMalloc(&out.w16, (size_t)k2 * 256 * 4 * Cin * 2));
  TC(hipMalloc((void**)&out.bias_rep, (size_t)k2 * 256 * 4));
  TC(hipMalloc((void**)&out.bmiss, (size_t)9 * 256 * 4));
  TC(hipMemcpyAsync(d_a, a_tap.data(), a_tap.size() * 4, hipMemcpyHostToDevice, s));
  TC(hipMemcpyAsync(d_b, b_ph.data(), b_ph.size() * 4, hipMemcpyHostToDevice, s));
  TC(hipMemcpyAsync(d_bct, bct, (size_t)Cm * 4, hipMemcpyHostToDevice, s));
  TC(hipMemsetAsync(out.w16, 0, (size_t)k2 * 256 * 4 * Cin * 2, s));
  for (int a = 0; a < k; ++a)
    for (int b = 0; b < k; ++b) {
      const int ph = a * k + b;
      unsigned word = 0;
      int idx = 0;
      for (int di = -1; di <= 1; ++di)
        for (int dj = -1; dj <= 1; ++dj) {
          bool first = true;
          for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
              const int ya = a + dy, xb = b + dx;
              const int fi = ya < 0 ? -1 : ya / k, fj = xb < 0 ? -1 : xb / k;   // floor
              if (fi != di || fj != dj) continue;
              const int pa = ((ya % k) + k) % k, pb = ((xb % k) + k) % k;
              TC(wm_launch_linear_f32(d_a + (size_t)((dy + 1) * 3 + dx + 1) * F_ * Cm, d_b + (size_t)(pa * k + pb) * Cin * Cm, nullptr, d_m, F_, Cin, Cm, Cm,
                                      Cin, 0, 0, nullptr, first ? 0 : 1, s));
              first = false;
            }
          if (first) continue;   // no tap of this phase lands in that neighbour
          TC(wm_launch_f32_to_16_2d(d_m, Cin, (uint16_t*)out.w16 + (size_t)ph * 256 * 4 * Cin + (size_t)idx * Cin, 4 * Cin, F_, Cin, dt, s));
          word |= (unsigned)((di + 1) | ((dj + 1) << 2)) << (4 * idx);
          ++idx;
        }
      out.list[ph] = word | ((unsigned)idx << 16);
    }
  for (int t = 0; t < 9; ++t)   // bmiss[t] = W_rn[t] b_ct ; bias = their sum
    TC(wm_launch_linear_f32(d_bct, d_a + (size_t)t * F_ * Cm, nullptr, out.bmiss + (size_t)t * 256, 1, F_, Cm, Cm, F_, 0, 0, nullptr, 0, s));
  for (int t = 0; t < 9; ++t)
    TC(wm_launch_linear_f32(d_bct, d_a + (size_t)t * F_ * Cm, nullptr, out.bias_rep, 1, F_, Cm, Cm, F_, 0, 0, nullptr, t ? 1 : 0, s));
  for (int ph = 1; ph < k2; ++ph) TC(hipMemcpyAsync(out.bias_rep + (size_t)ph * 256, out.bias_rep, 256 * 4, hipMemcpyDeviceToDevice, s));
  TC(hipStreamSynchronize(s));
#undef TC
done:
  if (d_a) (void)hipFree(d_a);
  if (d_b) (void)hipFree(d_b);
  if (d_bct) (void)hipFree(d_bct);
  if (d_m) (void)hipFree(d_m);
  if (e != hipSuccess) free_tconv(out);
  return e;
}
// woc1: torch Conv2d weight [Co][F][3][3]; wout: [F][F] (1x1); bout [F] (host fp32)
static hipError_t build_up1comp(int dt, int F_, int Co, const float* woc1, const float* wout, const float* bout, Up1Comp& out, hipStream_t s) {
  free_up1comp(out);
  out.co = Co;
  std::vector<float> a_tap((size_t)9 * Co * F_), wt((size_t)F_ * F_);
  for (int co = 0; co < Co; ++co)
    for (int c = 0; c < F_; ++c)
      for (int t = 0; t < 9; ++t) a_tap[((size_t)t * Co + co) * F_ + c] = woc1[((size_t)co * F_ + c) * 9 + t];
  for (int c = 0; c < F_; ++c)
    for (int ci = 0; ci < F_; ++ci) wt[(size_t)ci * F_ + c] = wout[(size_t)c * F_ + ci];   // W_out^T: rows ci, K = c
  float *d_a = nullptr, *d_w = nullptr, *d_b = nullptr, *d_m = nullptr;
  hipError_t e;
#define TC(x) do { e = (x); if (e != hipSuccess) goto done; } while (0)
  TC(hipMalloc((void**)&d_a, a_tap.size() * 4));
  TC(hipMalloc((void**)&d_w, wt.size() * 4));
  TC(hipMalloc((void**)&d_b, (size_t)F_ * 4));
  TC(hipMalloc((void**)&d_m, (size_t)9 * Co * F_ * 4));
  TC(hipMalloc(&out.w16, (size_t)9 * Co * F_ * 2));
  TC(hipMalloc((void**)&out.bias, (size_t)9 * Co * 4));
  TC(hipMemcpyAsync(d_a, a_tap.data(), a_tap.size() * 4, hipMemcpyHostToDevice, s));
  TC(hipMemcpyAsync(d_w, wt.data(), wt.size() * 4, hipMemcpyHostToDevice, s));
  TC(hipMemcpyAsync(d_b, bout, (size_t)F_ * 4, hipMemcpyHostToDevice, s));
  // all nine taps at once: rows (tap, co) of a_tap against W_out^T -> [9 Co][F]; the bias rows the same against b_out
  TC(wm_launch_linear_f32(d_a, d_w, nullptr, d_m, 9 * Co, F_, F_, F_, F_, 0, 0, nullptr, 0, s));
  TC(wm_launch_f32_to_16_2d(d_m, F_, out.w16, F_, 9 * Co, F_, dt, s));
  TC(wm_launch_linear_f32(d_b, d_a, nullptr, out.bias, 1, 9 * Co, F_, F_, 9 * Co, 0, 0, nullptr, 0, s));
  TC(hipStreamSynchronize(s));
#undef TC
done:
  if (d_a) (void)hipFree(d_a);
  if (d_w) (void)hipFree(d_w);
  if (d_b) (void)hipFree(d_b);
  if (d_m) (void)hipFree(d_m);
  if (e != hipSuccess) free_up1comp(out);
  return e;
}
// rn = tconv(tokens): tokens16 [n][gh][gw][Cin] (16-bit) -> out fp32 [n][k gh][k gw][256]
static hipError_t launch_tconv(const TconvPack& t, int dt, const void* tokens16, float* out, int n, int gh, int gw, const void* zero16, hipStream_t s) {
  WmGemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = tokens16; g.W = t.w16; g.C = out; g.bias = t.bias_rep; g.M = n * gh * gw; g.N = t.k * t.k * 256; g.K = 4 * t.cin; g.lda = t.cin; g.ldw = 4 * t.cin; g.ldc = 256;
  g.dtype = dt; g.epi = WM_EPI_CONV; g.cv_h = gh; g.cv_w = gw; g.cv_cin = t.cin; g.cv_zero = zero16; g.tc_k = t.k;
  memcpy(g.tc_list, t.list, sizeof(g.tc_list));
  hipError_t e = wm_launch_gemm(g, s);
  if (e != hipSuccess) return e;
  return wm_launch_tconv_border(out, t.bmiss, n, t.k * gh, t.k * gw, 256, s);
}

// up_hs > 0: x is [N][up_hs][up_ws][Cin] and the conv runs on its align_corners bilinear resize to (Hi, Wi)
// (+ position tables), fused into the 3x3 halo kernel's input staging.
wm_status conv(Ctx& c, const float* x, const std::string& wname, bool bias, const float* resid, bool resid_relu, const float* resid2,
               float* y, int N, int Hi, int Wi, int ks, int stride, int pad, bool relu_in, int up_hs = 0, int up_ws = 0,
               const float* up_addx = nullptr, const float* up_addy = nullptr, bool* out16 = nullptr, bool relu_out = false, bool in16 = false) {
  const Weight* w = W(c.h, wname + ".weight");
  if (!w || !w->w16) return fail(c.h, WM_ERR_STATE, "missing conv weight " + wname);
  WmConvArgs a;
  memset(&a, 0, sizeof(a));
  a.up_hs = up_hs; a.up_ws = up_ws; a.up_addx = up_addx; a.up_addy = up_addy;
  a.x = x; a.w = w->w16; a.bias = bias ? F(c.h, wname + ".bias") : nullptr; a.resid = resid; a.resid2 = resid2; a.y = y;
  a.N = N; a.Hi = Hi; a.Wi = Wi; a.Cin = (int)w->shape[1]; a.Cout = (int)w->shape[0]; a.ksize = ks; a.stride = stride; a.pad = pad;
  a.Ho = (Hi + 2 * pad - ks) / stride + 1; a.Wo = (Wi + 2 * pad - ks) / stride + 1;
  a.relu_in = relu_in; a.resid_relu = resid_relu; a.relu_out = relu_out ? 1 : 0; a.in16 = in16 ? 1 : 0; a.dtype = c.hdt;
  if (out16) {  // the caller can take y as a 16-bit tensor (its only consumer rounds it to the operand type anyway): granted when the kernel can
    static const int o16_env = [] { const char* e = wm_env("WM_OUTCONV_GEMM"); return e ? atoi(e) : 1; }();
    *out16 = o16_env != 0 && wm_conv3x3_out16_ok(a);
    a.out16 = *out16 ? 1 : 0;
  }
  // timing kinds by kernel instantiation: the F -> F 3x3 convs of the two large pyramid levels, output_conv1 with its fused
  // resize; everything else (small levels, 1x1, stride 2) under 3
  const bool pyr = ks == 3 && stride == 1 && up_hs == 0 && a.Cin == a.Cout && a.Cin >= 128;
  const int kind = up_hs > 0 ? 10 : pyr && Hi == 4 * c.d.gh ? 8 : pyr && Hi == 2 * c.d.gh ? 9 : 3;
  ProfScope ps(c.h, kind, c.s);
  if (wm_tuning[WM_TUNE_CONV_GEMM] == 1 && in16 && ks == 3 && stride == 1 && pad == 1 && up_hs == 0 && a.Cin % 64 == 0 && a.Cout % 8 == 0 &&
      (long)N * Hi * Wi >= 4096) {
    // a 16-bit NHWC input: the conv IS the ping-pong GEMM over (pixels) x (tap, channel), the A pieces DMA-ed from the shifted pixels
    WmGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = x; g.W = w->w16; g.C = y; g.bias = a.bias; g.M = N * Hi * Wi; g.N = a.Cout; g.K = 9 * a.Cin; g.lda = a.Cin; g.ldw = 9 * a.Cin; g.ldc = a.Cout;
    g.dtype = c.hdt; g.epi = WM_EPI_CONV; g.cv_h = Hi; g.cv_w = Wi; g.cv_cin = a.Cin; g.cv_zero = B<uint16_t>(c.h, "ZERO256");
    g.cv_resid = resid; g.cv_resid2 = resid2; g.cv_resid_relu = resid_relu ? 1 : 0; g.out16 = a.out16; g.relu = a.relu_out;
    LCHK(c, wm_launch_gemm(g, c.s));
    return WM_OK;
  }
  LCHK(c, wm_launch_conv(a, c.s));
  return WM_OK;
}

// ResidualConvUnit (dense_head.py:435-455): y = conv2(relu(conv1(relu x))) + relu(x) (+ extra)
// y16: when non-null the caller can take y as a 16-bit tensor; *y16 tells whether it got one
// conv1's output has ONE consumer, conv2's input staging, which applies ReLU and rounds to the operand type: where the kernel can, conv1
// writes exactly that — relu(conv1) as a 16-bit tensor — and conv2 stages it unconverted (bit-identical values; a quarter of the
// bytes written, a quarter read: 268 MB less traffic per RCU at 148^2 x 8 views).  WM_RCU_MID16=0: the fp32 intermediate (A/B).
wm_status rcu(Ctx& c, const std::string& p, const float* x, const float* extra, float* tmp, float* y, int N, int Hh, int Ww, bool* y16 = nullptr) {
  static const bool mid16_env = [] { const char* e = wm_env("WM_RCU_MID16"); return !e || atoi(e) != 0; }();
  const bool want = wm_tuning[WM_TUNE_RCU_MID16] >= 0 ? wm_tuning[WM_TUNE_RCU_MID16] != 0 : mid16_env;
  bool mid16 = false;
  wm_status st = conv(c, x, p + "conv1", true, nullptr, false, nullptr, tmp, N, Hh, Ww, 3, 1, 1, true, 0, 0, nullptr, nullptr, want ? &mid16 : nullptr, want);
  if (st) return st;
  // (relu_out was requested together with out16; if the kernel could not grant out16 the fp32 tmp holds relu(conv1), and relu is idempotent)
  return conv(c, tmp, p + "conv2", true, x, true, extra, y, N, Hh, Ww, 3, 1, 1, !mid16, 0, 0, nullptr, nullptr, y16, false, mid16);
}

// FeatureFusionBlock.out_conv (1x1, dense_head.py:496) on x2 [N*H*W][F]: a plain GEMM when x2 came as a 16-bit tensor (the ping-pong
// kernel, fp32 NHWC out), the generic conv kernel on the fp32 tensor otherwise
// y16: the output as a 16-bit tensor of the operand type (its only consumer is a GEMM: the tap GEMM of upconv.hip); needs x2_is_16
wm_status out_conv(Ctx& c, const std::string& name, const float* x2, bool x2_is_16, float* y, int N, int Hh, int Ww, int F_, bool y16 = false) {
  if (!x2_is_16) return y16 ? fail(c.h, WM_ERR_STATE, "out_conv: 16-bit output needs the 16-bit input") : conv(c, x2, name, true, nullptr, false, nullptr, y, N, Hh, Ww, 1, 1, 0, false);
  const Weight* w = W(c.h, name + ".weight");
  if (!w || !w->w16 || (int)w->shape[0] != F_ || (int)w->shape[1] != F_) return fail(c.h, WM_ERR_STATE, "missing conv weight " + name);
  return gemm(c, c.hdt, y16 ? WM_EPI_T16 : WM_EPI_F32, x2, F_, w->w16, F_, y, F_, F(c.h, name + ".bias"), nullptr, N * Hh * Ww, F_, F_, nullptr, 3);
}

// DPTHead (dense_head.py:107-295) for views [v0, v0+n) of this rank
wm_status dpt_head(Ctx& c, const std::string& p, int F_, int out_dim, int act, bool is_gs, float* out_attr, float* out_conf,
                   const float* img, const wm_outputs* out, int first_view, int slot) {
  const std::string sx = "_h" + std::to_string(slot);
  auto HB = [&](const char* n) { return (float*)c.h->buf.at(std::string(n) + sx); };
  wm_handle* h = c.h;
  const Dims& d = c.d;
  const wm_config& cf = h->cfg;
  const int D2 = 2 * d.D, hw = d.hw, gh = d.gh, gw = d.gw;
  const int32_t* oc = cf.dpt_out_channels;
  const std::string sc = p + "scratch.";
  float* S0 = HB("dpt_s0"); float* S1 = HB("dpt_s1"); float* S2 = HB("dpt_s2"); float* S3 = HB("dpt_s3");
  wm_status st;
  for (int v0 = 0; v0 < d.n; v0 += d.chunk) {
    const int n = std::min(d.chunk, d.n - v0);
    float* feats[4] = {HB("dpt_f0"), HB("dpt_f1"), HB("dpt_f2"), HB("dpt_f3")};
    bool tconv_done[4] = {false, false, false, false};
    for (int i = 0; i < 4; ++i) {
      const float* tap = B<float>(h, ("tap" + std::to_string(i)).c_str()) + (size_t)v0 * d.P * D2;
      void* T16 = HB("dpt_T16");
      st = layernorm(c, tap, D2, T16, D2, F(h, p + "norm.weight"), F(h, p + "norm.bias"), D2, 1e-5f, n, hw, d.P, d.psi, hw, 0, 0, c.hdt);
      if (st) return st;
      const std::string pj = p + "projects." + std::to_string(i);
      WmGemmArgs ex;
      memset(&ex, 0, sizeof(ex));
      ex.rows_per_group = hw; ex.out_group = hw; ex.out_off = 0; ex.add = B<float>(h, ("dpt_pos" + std::to_string(i)).c_str());
      if (i < 2) {  // feeds a k==stride ConvTranspose (GEMM): 16-bit output
        ex.out16 = 1;
        void* P16 = HB("dpt_P16");
        st = gemm(c, c.hdt, WM_EPI_ROWMAP_ADD, T16, D2, W16(h, pj + ".weight"), D2, P16, oc[i], F(h, pj + ".bias"), nullptr, n * hw, oc[i], D2, &ex);
        if (st) return st;
        auto tc = h->tconv.find(p + std::to_string(i));
        tconv_done[i] = wm_tuning[WM_TUNE_TCONV] != 0 && !is_gs && F_ == 256 && tc != h->tconv.end() && tc->second.w16 != nullptr;
        if (tconv_done[i]) {   // ConvTranspose and layer{i+1}_rn as one GEMM at the token resolution, straight into rn[i]
          ProfScope ps(h, i == 0 ? 8 : 3, c.s);
          LCHK(c, launch_tconv(tc->second, c.hdt, P16, HB(i == 0 ? "dpt_rn1" : "dpt_rn2"), n, gh, gw, B<uint16_t>(h, "ZERO256"), c.s));
          continue;
        }
        const int k = i == 0 ? 4 : 2;
        const std::string rs = p + "resize_layers." + std::to_string(i);
        WmGemmArgs ct;
        memset(&ct, 0, sizeof(ct));
        ct.ct_k = k; ct.ct_cout = oc[i]; ct.ct_gh = gh; ct.ct_gw = gw;
        st = gemm(c, c.hdt, WM_EPI_CONVT, P16, oc[i], W16(h, rs + ".weight"), oc[i], feats[i], 0, F(h, rs + ".bias"), nullptr, n * hw, k * k * oc[i], oc[i], &ct);
        if (st) return st;
      } else {
        float* dst = i == 2 ? feats[2] : HB("dpt_f3in");
        st = gemm(c, c.hdt, WM_EPI_ROWMAP_ADD, T16, D2, W16(h, pj + ".weight"), D2, dst, oc[i], F(h, pj + ".bias"), nullptr, n * hw, oc[i], D2, &ex);
        if (st) return st;
        if (i == 3) {
          st = conv(c, dst, p + "resize_layers.3", true, nullptr, false, nullptr, feats[3], n, gh, gw, 3, 2, 1, false);
          if (st) return st;
        }
      }
    }
    const int Hs[4] = {4 * gh, 2 * gh, gh, d.gh2}, Ws[4] = {4 * gw, 2 * gw, gw, d.gw2};
    float* rn[4] = {HB("dpt_rn1"), HB("dpt_rn2"), HB("dpt_rn3"), HB("dpt_rn4")};
    for (int i = 0; i < 4; ++i) {
      if (tconv_done[i]) continue;
      st = conv(c, feats[i], sc + "layer" + std::to_string(i + 1) + "_rn", false, nullptr, false, nullptr, rn[i], n, Hs[i], Ws[i], 3, 1, 1, false);
      if (st) return st;
    }
    // refinenet4: RCU2(rn4) -> resize to level 3 -> out_conv
    bool x16 = false;
    st = rcu(c, sc + "refinenet4.resConfUnit2.", rn[3], nullptr, S0, S1, n, Hs[3], Ws[3], &x16);
    if (st) return st;
    // out_conv (1x1) and the align_corners bilinear resize are both linear and the interpolation weights sum
    // to 1, so out_conv(resize(x)) == resize(out_conv(x)): run the 1x1 at the LOW resolution (4x fewer FLOPs)
    st = out_conv(c, sc + "refinenet4.out_conv", S1, x16, S0, n, Hs[3], Ws[3], F_);
    if (st) return st;
    LCHK(c, wm_launch_bilinear(S0, S2, n, Hs[3], Ws[3], Hs[2], Ws[2], F_, nullptr, nullptr, c.s));
    float* cur = S2;  // output of the previous fusion block at level L
    // the two big resizes feed only a 3x3 conv: fuse them into that conv's input staging when the halo kernel applies
    const bool fuse_on = wm_tuning[WM_TUNE_CONV_FUSE_UP] != 0 && wm_env("WM_CONV_GENERIC") == nullptr;
    const bool fuse_up1 = fuse_on && F_ % 64 == 0 && 4 * Hs[0] * Ws[0] >= 256;
    const bool fuse_up2 = fuse_on && !is_gs && (F_ / 2) % 64 == 0;
    // output_conv1 on the resized tensor as nine 1x1 products at the LOW resolution + a bilinear gather (upconv.hip): a quarter of the MFMA work
    const Weight* w_oc1 = W(h, sc + "output_conv1.weight");
    const bool gather_on = wm_tuning[WM_TUNE_UP1_GATHER] != 0 && c.hdt == WM_DT_F16 && w_oc1 && w_oc1->tap_major && F_ % 64 == 0 &&
                           (int)w_oc1->shape[1] == F_ && ((int)w_oc1->shape[0] == 128 || (int)w_oc1->shape[0] == 64 || (int)w_oc1->shape[0] == 32) &&
                           (unsigned long long)n * Hs[0] * Ws[0] * 9ull * (unsigned long long)w_oc1->shape[0] * 2ull < (1ull << 32);   // the gather's 32-bit offsets
    bool up1_gather = false;
    const Up1Comp* comp = nullptr;
    {
      auto it = h->up1comp.find(p);
      if (gather_on && wm_tuning[WM_TUNE_UP1_COMP] != 0 && it != h->up1comp.end() && it->second.w16 && it->second.co == (int)w_oc1->shape[0]) comp = &it->second;
    }
    for (int L = 2; L >= 0; --L) {  // refinenet3 (level 2), refinenet2 (level 1), refinenet1 (level 0)
      const std::string rp = sc + "refinenet" + std::to_string(L + 1) + ".";
      float* others[3];
      int k = 0;
      for (float* b : {S0, S1, S2, S3}) if (b != cur) others[k++] = b;
      // x = cur + RCU1(rn[L])
      st = rcu(c, rp + "resConfUnit1.", rn[L], cur, others[0], others[1], n, Hs[L], Ws[L]);
      if (st) return st;
      // x = RCU2(x)
      st = rcu(c, rp + "resConfUnit2.", others[1], nullptr, others[0], others[2], n, Hs[L], Ws[L], &x16);
      if (st) return st;
      const int Ho = L > 0 ? Hs[L - 1] : 2 * Hs[0], Wo = L > 0 ? Ws[L - 1] : 2 * Ws[0];
      up1_gather = L == 0 && gather_on && x16;
      if (up1_gather && comp != nullptr) { cur = others[2]; break; }   // out_conv lives inside the tap matrices: the tap GEMM reads x2 itself
      st = out_conv(c, rp + "out_conv", others[2], x16, others[0], n, Hs[L], Ws[L], F_, up1_gather);
      if (st) return st;
      if (L == 0 && (fuse_up1 || up1_gather)) { cur = others[0]; break; }  // the last resize is fused into output_conv1 (its input staging, or the tap form)
      LCHK(c, wm_launch_bilinear(others[0], others[1], n, Hs[L], Ws[L], Ho, Wo, F_, nullptr, nullptr, c.s));
      cur = others[1];
    }
    const int H8 = 8 * gh, W8 = 8 * gw;
    float* others[3];
    {
      int k = 0;
      for (float* b : {S0, S1, S2, S3}) if (b != cur) others[k++] = b;
    }
    if (up1_gather) {
      const int Co = (int)w_oc1->shape[0];
      const size_t numel = (size_t)Co * 9 * F_;
      void* Y1 = others[1];
      if (comp) st = gemm(c, c.hdt, WM_EPI_T16, cur, F_, comp->w16, F_, Y1, 9 * Co, comp->bias, nullptr, n * Hs[0] * Ws[0], 9 * Co, F_, nullptr, 10);
      else st = gemm(c, c.hdt, WM_EPI_T16, cur, F_, (const uint16_t*)w_oc1->w16 + numel, F_, Y1, 9 * Co, nullptr, nullptr, n * Hs[0] * Ws[0], 9 * Co, F_, nullptr, 10);
      if (!st) {
        ProfScope ps(h, 10, c.s);
        LCHK(c, wm_launch_upconv_gather(Y1, F(h, sc + "output_conv1.bias"), others[0], n, Hs[0], Ws[0], H8, W8, Co, c.s));
      }
    } else if (fuse_up1) st = conv(c, cur, sc + "output_conv1", true, nullptr, false, nullptr, others[0], n, H8, W8, 3, 1, 1, false, Hs[0], Ws[0]);
    else st = conv(c, cur, sc + "output_conv1", true, nullptr, false, nullptr, others[0], n, H8, W8, 3, 1, 1, false);
    if (st) return st;
    const int Ho = gh * cf.patch_size, Wo = gw * cf.patch_size;
    float* fused = others[1];
    const float* posx = B<float>(h, is_gs ? "gs_posx" : "dpt_posx");
    const float* posy = B<float>(h, is_gs ? "gs_posy" : "dpt_posy");
    // output_conv2[0] un-fused (measured: 16-bit LDS-tiled resize 260 us + DMA-fed 32-channel conv 255 us vs 600-630 us for the
    // fused-resize kernel at 8 views, tools/bench_up_conv_n32.py): resize into `fused` as 16-bit, conv reads it by LDS-DMA
    bool tail_done = false;
    static const int up2_env = [] { const char* e = wm_env("WM_UP2_UNFUSED"); return e ? atoi(e) : 1; }();
    const Weight* w_oc2 = W(h, sc + "output_conv2.0.weight");
    const bool up2_unfused = up2_env && !is_gs && (F_ / 2) % 64 == 0 && F_ / 2 <= 128 && w_oc2 && w_oc2->shape[0] == 32 && w_oc2->w16 != nullptr;
    if (up2_unfused) {
      LCHK(c, wm_launch_bilinear16(others[0], fused, n, H8, W8, Ho, Wo, F_ / 2, posx, posy, c.hdt, c.s));
      WmConvN32Args a;
      memset(&a, 0, sizeof(a));
      a.x = (const uint16_t*)fused; a.w = (const uint16_t*)w_oc2->w16; a.bias = F(h, sc + "output_conv2.0.bias"); a.y = others[2];
      a.zero = B<uint16_t>(h, "ZERO256"); a.N = n; a.H = Ho; a.W = Wo; a.Cin = F_ / 2; a.relu_out = 0; a.dtype = c.hdt;
      // ... with the head's tail (ReLU, 1x1 conv 32 -> C, activations) in its epilogue: the 32-channel tensor is never stored
      a.tail_w = F(h, sc + "output_conv2.2.weight"); a.tail_b = F(h, sc + "output_conv2.2.bias"); a.tail_C = out_dim; a.tail_act = act;
      a.tail_attr = out_attr + (size_t)v0 * Ho * Wo * (out_dim - 1); a.tail_conf = out_conf + (size_t)v0 * Ho * Wo;
      ProfScope ps(h, 11, c.s);
      LCHK(c, wm_launch_conv3x3_n32_in16(a, c.s));
      tail_done = true;
    } else if (fuse_up2) {  // (the GS branch also needs the resized tensor itself: input_merger accumulates into it)
      st = conv(c, others[0], sc + "output_conv2.0", true, nullptr, false, nullptr, others[2], n, Ho, Wo, 3, 1, 1, false, H8, W8, posx, posy);
    } else {
      LCHK(c, wm_launch_bilinear(others[0], fused, n, H8, W8, Ho, Wo, F_ / 2, posx, posy, c.s));
      st = conv(c, fused, sc + "output_conv2.0", true, nullptr, false, nullptr, others[2], n, Ho, Wo, 3, 1, 1, false);
    }
    if (st) return st;
    const size_t npix = (size_t)n * Ho * Wo, voff = (size_t)v0 * Ho * Wo;
    if (!tail_done)
      LCHK(c, wm_launch_dpt_tail(others[2], F(h, sc + "output_conv2.2.weight"), F(h, sc + "output_conv2.2.bias"),
                                 out_attr + voff * (out_dim - 1), out_conf + voff, npix, out_dim, act, c.s));
    if (is_gs && out->splat_means) {
      // dense_head.py:232-244: fused += ReLU(conv7x7(img)); then GaussianSplatRenderer.gs_head
      // (rasterization.py:149-153): conv3x3 (no bias) -> ReLU -> conv1x1 -> 12 raw parameters per pixel
      const float* img_c = img + (size_t)v0 * 3 * Ho * Wo;
      void* col = B<void>(h, "gs_im2col");
      LCHK(c, wm_launch_im2col7(img_c, col, n, Ho, Wo, 192, c.hdt, c.s));
      WmGemmArgs ex;
      memset(&ex, 0, sizeof(ex));
      ex.rows_per_group = (int)npix; ex.accumulate = 1; ex.relu = 1;
      st = gemm(c, c.hdt, WM_EPI_ROWMAP_ADD, col, 192, W16(h, p + "input_merger.0.weight"), 192, fused, F_ / 2,
                F(h, p + "input_merger.0.bias"), nullptr, (int)npix, F_ / 2, 192, &ex);
      if (st) return st;
      float* X = others[0];   // [n][H][W][gs_dim]
      float* gp = others[2];  // [n*H*W][12] (y32 is dead after the tail)
      st = conv(c, fused, "gs_renderer.gs_head.0", false, nullptr, false, nullptr, X, n, Ho, Wo, 3, 1, 1, false);
      if (st) return st;
      st = conv(c, X, "gs_renderer.gs_head.2", true, nullptr, false, nullptr, gp, n, Ho, Wo, 1, 1, 0, true);
      if (st) return st;
      LCHK(c, wm_launch_gs_splat(gp, img_c, out_attr + voff, B<float>(h, "cam_params") + (size_t)(first_view + v0) * 9,
                                 out->splat_means + voff * 3, out->splat_quats + voff * 4, out->splat_scales + voff * 3,
                                 out->splat_opacities + voff, out->splat_sh + voff * 3, out->splat_weights + voff, n, Ho, Wo, c.s));
    }
  }
  return WM_OK;
}

wm_status forward_impl(wm_handle* h, const float* img, int n, int first_view, int nt, int H, int W_, const float* pose7,
                       const float* depth, const float* ray4, const int32_t* flags, const wm_outputs* out, hipStream_t s) {
  if (!h || !img || !out) return WM_ERR_INVALID;
  if (!h->finalized) return fail(h, WM_ERR_STATE, "weights not finalized");
  const wm_config& cf = h->cfg;
  if (n <= 0 || nt < n || nt % n) return fail(h, WM_ERR_INVALID, "n_total must be a positive multiple of n_local");
  if (H % cf.patch_size || W_ % cf.patch_size) return fail(h, WM_ERR_INVALID, "H and W must be multiples of patch_size");
  if (nt / n > 1 && (h->comm.kind == 0 || h->comm.world != nt / n)) return fail(h, WM_ERR_COMM, "communicator world size mismatch");
  HIPCHK(h, hipSetDevice(h->device));
  Ctx c{h, make_dims(h, n, nt, H, W_), s, cf.backbone_dtype, cf.head_dtype};
  const Dims& d = c.d;
  // no allocation, no host-side table building and no synchronisation in the forward: the workspace must exist
  if (!(h->plan_n == d.n && h->plan_nt == d.nt && h->plan_H == d.H && h->plan_W == d.W))
    return fail(h, WM_ERR_STATE, "no workspace for this shape: call wm_reserve(h, n_local, n_total, H, W) first (and again after loading weights)");
  wm_status st = WM_OK;
  for (int k = 0; k < wm_handle::NKIND; ++k) h->ev_used[k] = 0;
  ProfScope whole(h, 4, s);
  HIPCHK(h, hipMemsetAsync(B<char>(h, "ZERO256"), 0, 256, s));
  const int D = d.D;
  const std::string v = "visual_geometry_transformer.", dn = v + "patch_embed.";
  float* Xd = B<float>(h, "Xd");
  float* Xv = B<float>(h, "Xv");
  void* A16 = B<void>(h, "A16");

  // ---- a3/a4: normalise + patchify + DINOv2 encoder (vision_transformer.py:209-266)
  LCHK(c, wm_launch_im2col(img, A16, n, 3, H, W_, cf.patch_size, d.kpad_patch, 1, c.bdt, s));
  {
    WmGemmArgs ex;
    memset(&ex, 0, sizeof(ex));
    ex.rows_per_group = d.hw; ex.out_group = d.Td; ex.out_off = 1 + d.R; ex.add = B<float>(h, "dino_pos") + D;
    st = gemm(c, c.bdt, WM_EPI_ROWMAP_ADD, A16, d.kpad_patch, W16(h, dn + "patch_embed.proj.weight"), d.kpad_patch, Xd, D,
              F(h, dn + "patch_embed.proj.bias"), nullptr, n * d.hw, D, d.kpad_patch, &ex);
    if (st) return st;
  }
  LCHK(c, wm_launch_dino_tokens(nullptr, F(h, dn + "cls_token"), F(h, dn + "register_tokens"), B<float>(h, "dino_pos"), Xd, n, d.hw, d.R, D, s));
  {
    bool ln1 = false;
    for (int i = 0; i < cf.dino_depth; ++i) {
      const std::string nxt = i + 1 < cf.dino_depth ? dn + "blocks." + std::to_string(i + 1) + "." : std::string();
      st = backbone_block(c, dn + "blocks." + std::to_string(i) + ".", Xd, d.Md, d.Td, cf.dino_heads, 1e-6f, false, false, d.Td, 0, false, nullptr, ln1, nxt, &ln1);
      if (st) return st;
    }
  }
  // final LN, patch tokens only, straight into the multi-view token buffer
  st = layernorm(c, Xd, D, Xv, D, F(h, dn + "norm.weight"), F(h, dn + "norm.bias"), D, 1e-6f, n, d.hw, d.Td, 1 + d.R, d.P, d.psi, 1, 0);
  if (st) return st;

  // ---- a5: special + prior tokens (visual_transformer.py:285-295,343-371)
  const float* pose_tok = nullptr; const float* ray_tok = nullptr;
  if (cf.enable_cond) {
    float* pin = B<float>(h, "pr_in"); float* ph = B<float>(h, "pr_h");
    if (flags && flags[0] == 1 && pose7) {
      LCHK(c, hipMemcpy2DAsync(pin, 8 * 4, pose7, 7 * 4, 7 * 4, n, hipMemcpyDeviceToDevice, s));
      st = linear32(c, pin, 8, v + "pose_embed.0", ph, D, n, 0, 1);
      if (st) return st;
      st = linear32(c, ph, D, v + "pose_embed.2", B<float>(h, "pose_tok"), D, n, 0, 0);
      if (st) return st;
      pose_tok = B<float>(h, "pose_tok");
    }
    if (flags && flags[2] == 1 && ray4) {
      st = linear32(c, ray4, 4, v + "ray_embed.0", ph, D, n, 0, 1);
      if (st) return st;
      st = linear32(c, ph, D, v + "ray_embed.2", B<float>(h, "ray_tok"), D, n, 0, 0);
      if (st) return st;
      ray_tok = B<float>(h, "ray_tok");
    }
    if (flags && flags[1] == 1 && depth) {  // PatchEmbed_Mlp (patch_embed.py:79-93): unshuffle -> fc1 -> GELU -> fc2, added to patches
      LCHK(c, wm_launch_im2col(depth, A16, n, 1, H, W_, cf.patch_size, d.kpad_depth, 0, c.bdt, s));
      void* H16 = B<void>(h, "H16");
      st = gemm(c, c.bdt, WM_EPI_GELU_T16, A16, d.kpad_depth, W16(h, v + "depth_embed.proj.2.fc1.weight"), d.kpad_depth, H16, 4 * D,
                F(h, v + "depth_embed.proj.2.fc1.bias"), nullptr, n * d.hw, 4 * D, d.kpad_depth);
      if (st) return st;
      WmGemmArgs ex;
      memset(&ex, 0, sizeof(ex));
      ex.rows_per_group = d.hw; ex.out_group = d.P; ex.out_off = d.psi; ex.accumulate = 1;
      st = gemm(c, c.bdt, WM_EPI_ROWMAP_ADD, H16, 4 * D, W16(h, v + "depth_embed.proj.2.fc2.weight"), 4 * D, Xv, D,
                F(h, v + "depth_embed.proj.2.fc2.bias"), nullptr, n * d.hw, D, 4 * D, &ex);
      if (st) return st;
    }
  }
  LCHK(c, wm_launch_vgt_special(Xv, F(h, v + "cam_token"), F(h, v + "reg_token"), pose_tok, ray_tok, n, d.P, d.R, D, cf.enable_cond, first_view, s));

  // ---- a7-a10: 24 x (frame block, global block) + taps (visual_transformer.py:309-339)
  int tap_i = 0;
  bool vgt_ln1 = false;   // A16 already holds the coming block's norm1(X): the previous block's fc2 epilogue wrote it
  for (int i = 0; i < cf.depth; ++i) {
    const bool is_tap = tap_i < 4 && i == cf.intermediate_idxs[tap_i];
    float* tap = is_tap ? B<float>(h, ("tap" + std::to_string(tap_i)).c_str()) : nullptr;
    // (frame and global blocks alternate on the same token buffer: each block's fc2 epilogue computes the next block's norm1)
    const std::string gname = v + "global_blocks." + std::to_string(i) + ".";
    const std::string fnext = i + 1 < cf.depth ? v + "frame_blocks." + std::to_string(i + 1) + "." : std::string();
    st = backbone_block(c, v + "frame_blocks." + std::to_string(i) + ".", Xv, d.Mv, d.P, cf.num_heads, 1e-5f, true, true, d.P, d.psi, false, tap,
                        vgt_ln1, gname, &vgt_ln1);
    if (st) return st;
    st = backbone_block(c, gname, Xv, d.Mv, d.Mv, cf.num_heads, 1e-5f, true, true, d.P, d.psi, true, tap ? tap + D : nullptr, vgt_ln1, fnext, &vgt_ln1);
    if (st) return st;
    if (is_tap) {
      if (out->taps[tap_i]) LCHK(c, hipMemcpyAsync(out->taps[tap_i], tap, (size_t)d.Mv * 2 * D * 4, hipMemcpyDeviceToDevice, s));
      ++tap_i;
    }
  }

  // fallback statistics of the fast attention kernels -> pinned host mirror (read by the NEXT forwards' launch policy; no synchronisation)
  if (h->att_stat_host) LCHK(c, hipMemcpyAsync(h->att_stat_host, B<int>(h, "ATT_STAT"), (size_t)h->att_stat_n * 4, hipMemcpyDeviceToHost, s));

  // The camera head and the DPT heads are independent of each other: each runs on one of the handle's own queues, forked from and
  // joined back to the caller's stream (-1.15 ms per forward at 8 x 518^2, -5 ms at 32 views: the HBM-bound camera head and the under-filled small DPT
  // levels run beside the MFMA-bound convs).  WM_HEADS_CONCURRENT=0 (tuning heads_concurrent = 0) keeps one queue.
  // History: round 1 saw sparse wrong lanes with several queues active and fenced this off; round 2 traced it to packed-fp32 VALU
  // instructions (compiler-generated v_pk_mul_f32 / v_pk_fma_f32, op_sel forms) dropping one half's result in 16-lane groups when a
  // kernel of another kind shares the SIMD from another queue — reproduced without torch on the ROCm 7.2 runtime with the library's own
  // kernels (tools/micro/splat_hazard_repro.cpp, profiles/r02_multiqueue_hazard.md), never with one queue, never without packed fp32.
  // No kernel outside the GELU instantiations of gemm.hip contains a packed-fp32 instruction any more (-fno-slp-vectorize,
  // WM_NO_PACKED_FP32 in wm_common.h; held by tests/test_kernel_resources_cpu.py, which disassembles every object), and those
  // GELU GEMMs of the backbone have finished before this fork.  tools/stress_concurrent_heads.py: 4 900 concurrent forwards (C2, C3, C5 flag set) bit-identical
  // to the serial one.
  static const bool conc_env = [] { const char* e = wm_env("WM_HEADS_CONCURRENT"); return !e || atoi(e) != 0; }();
  const bool serial = !(wm_tuning[WM_TUNE_HEADS_CONC] >= 0 ? wm_tuning[WM_TUNE_HEADS_CONC] != 0 : conc_env) || h->prof;
  if (!h->hfork) LCHK(c, hipEventCreateWithFlags(&h->hfork, hipEventDisableTiming));
  if (!serial) LCHK(c, hipEventRecord(h->hfork, s));
  // ---- a11-a12: camera head.  864 MB of fp32 weights streamed 4 times for <= 64 rows: HBM-bound (1.3 ms at 8 views), so it runs
  // on its own queue BESIDE the MFMA-bound DPT heads (which do not depend on it, except the Gaussian head's splat assembly).
  bool cam_async = false;
  if (cf.enable_cam && out->camera_params) {
    Ctx cc = c;
    if (!serial) {
      if (!h->camstream) {
        LCHK(c, hipStreamCreateWithFlags(&h->camstream, hipStreamNonBlocking));
        LCHK(c, hipEventCreateWithFlags(&h->camjoin, hipEventDisableTiming));
      }
      cc.s = h->camstream;
      LCHK(c, hipStreamWaitEvent(cc.s, h->hfork, 0));
      cam_async = true;
    }
    st = camera_head(cc, B<float>(h, "cam_params"));
    if (st) return st;
    LCHK(c, hipMemcpyAsync(out->camera_params, B<float>(h, "cam_params"), (size_t)nt * 9 * 4, hipMemcpyDeviceToDevice, cc.s));
    if (out->camera_poses && out->camera_intrs)
      LCHK(c, wm_launch_cam_matrices(B<float>(h, "cam_params"), out->camera_poses, out->camera_intrs, nt, H, W_, cc.s));
    if (cam_async) LCHK(c, hipEventRecord(h->camjoin, cc.s));
  }
  // ---- a13: DPT heads (worldmirror.py:74-98): independent of each other; one per queue only with WM_HEADS_CONCURRENT=1 (see above).
  {
    struct HeadJob { const char* p; int F; int od; int act; bool gs; float* attr; float* conf; };
    std::vector<HeadJob> jobs;
    if (cf.enable_depth && out->depth && out->depth_conf) jobs.push_back({"depth_head.", cf.dpt_features, 2, WM_ACT_EXP, false, out->depth, out->depth_conf});
    if (cf.enable_pts && out->pts3d && out->pts3d_conf) jobs.push_back({"pts_head.", cf.dpt_features, 4, WM_ACT_INV_LOG, false, out->pts3d, out->pts3d_conf});
    if (cf.enable_norm && out->normals && out->normals_conf) jobs.push_back({"norm_head.", cf.dpt_features, 4, WM_ACT_NORM, false, out->normals, out->normals_conf});
    if (cf.enable_gs && out->gs_depth && out->gs_depth_conf) {
      if (out->splat_means && !(out->splat_quats && out->splat_scales && out->splat_opacities && out->splat_sh && out->splat_weights))
        return fail(h, WM_ERR_INVALID, "splat outputs must be given together");
      if (out->splat_means && !(cf.enable_cam && out->camera_params)) return fail(h, WM_ERR_INVALID, "splats need the camera head");
      jobs.push_back({"gs_head.", cf.gs_dim, 2, WM_ACT_EXP, true, out->gs_depth, out->gs_depth_conf});
    }
    // Round 4: the LAST head runs on the caller's stream itself and only the others fork.  HIP multiplexes streams onto four hardware
    // queues: with every head on a stream of its own (three heads + the camera head + the caller's idle stream) two DPT heads shared
    // a hardware queue and ran one after the other — the kernel trace of a timed forward showed one head done after 7.5 ms and the other
    // two, serialised, after 12.3 ms (profiles/r04_head_phase_queues.md).  Side heads are enqueued first, the caller's stream joins them
    // only behind its own head's launches.
    const size_t main_k = jobs.empty() || wm_tuning[WM_TUNE_HEADS_MAIN] == 0 ? (size_t)-1 : jobs.size() - 1;   // (tuning heads_main = 0: every head forks, round 3's form — A/B)
    auto run_head = [&](size_t k, hipStream_t hs) -> wm_status {
      Ctx hc = c;
      hc.s = hs;
      if (jobs[k].gs && cam_async) LCHK(c, hipStreamWaitEvent(hs, h->camjoin, 0));  // the splats are unprojected with the predicted cameras
      return dpt_head(hc, jobs[k].p, jobs[k].F, jobs[k].od, jobs[k].act, jobs[k].gs, jobs[k].attr, jobs[k].conf, img, out, first_view, (int)k);
    };
    const bool fork_heads = !serial && (jobs.size() > 1 || cam_async);
    for (size_t k = 0; k < jobs.size(); ++k) {
      if (fork_heads && k == main_k) continue;
      hipStream_t hs = s;
      if (fork_heads) {
        if (!h->hstream[k]) LCHK(c, hipStreamCreateWithFlags(&h->hstream[k], hipStreamNonBlocking));
        if (!h->hjoin[k]) LCHK(c, hipEventCreateWithFlags(&h->hjoin[k], hipEventDisableTiming));
        hs = h->hstream[k];
        LCHK(c, hipStreamWaitEvent(hs, h->hfork, 0));
      }
      st = run_head(k, hs);
      if (st) return st;
      if (hs != s) LCHK(c, hipEventRecord(h->hjoin[k], hs));
    }
    if (fork_heads && !jobs.empty()) {
      if (main_k != (size_t)-1) {
        st = run_head(main_k, s);
        if (st) return st;
      }
      for (size_t k = 0; k < jobs.size(); ++k)
        if (k != main_k) LCHK(c, hipStreamWaitEvent(s, h->hjoin[k], 0));
    }
  }
  if (cam_async) LCHK(c, hipStreamWaitEvent(s, h->camjoin, 0));
  return WM_OK;
}

}  // namespace

extern "C" wm_status wm_forward(wm_handle* h, const float* img, int N, int H, int W, const float* pose7, const float* depth,
                                const float* ray4, const int32_t cond_flags[3], const wm_outputs* out, void* stream) {
  return forward_impl(h, img, N, 0, N, H, W, pose7, depth, ray4, cond_flags, out, (hipStream_t)stream);
}

extern "C" wm_status wm_forward_sharded(wm_handle* h, const float* img, int n_local, int first_view, int n_total, int H, int W,
                                        const float* pose7, const float* depth, const float* ray4, const int32_t cond_flags[3],
                                        const wm_outputs* out, void* stream) {
  return forward_impl(h, img, n_local, first_view, n_total, H, W, pose7, depth, ray4, cond_flags, out, (hipStream_t)stream);
}

extern "C" wm_status wm_set_workspace(wm_handle* h, void* device_ptr, size_t bytes) {
  if (!h) return WM_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipDeviceSynchronize());
  if (h->arena && h->arena_owned) HIPCHK(h, hipFree(h->arena));
  h->arena = (char*)device_ptr;
  h->arena_bytes = device_ptr ? bytes : 0;
  h->arena_owned = device_ptr == nullptr;  // NULL hands the arena back to the library
  h->plan_n = -1;
  h->buf.clear();
  return WM_OK;
}

extern "C" wm_status wm_reserve(wm_handle* h, int n_local, int n_total, int H, int W) {
  if (!h) return WM_ERR_INVALID;
  if (!h->finalized) return fail(h, WM_ERR_STATE, "weights not finalized");
  if (n_local <= 0 || n_total < n_local || n_total % n_local) return fail(h, WM_ERR_INVALID, "n_total must be a positive multiple of n_local");
  if (H <= 0 || W <= 0 || H % h->cfg.patch_size || W % h->cfg.patch_size) return fail(h, WM_ERR_INVALID, "H and W must be multiples of patch_size");
  HIPCHK(h, hipSetDevice(h->device));
  return plan(h, make_dims(h, n_local, n_total, H, W));
}

// ====================================================================================== communicator
extern "C" wm_status wm_rccl_unique_id(uint8_t id[WM_RCCL_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) <= WM_RCCL_ID_BYTES, "id buffer too small");
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) return WM_ERR_COMM;
  memset(id, 0, WM_RCCL_ID_BYTES);
  memcpy(id, &u, sizeof(u));
  return WM_OK;
}

extern "C" wm_status wm_comm_init_rccl(wm_handle* h, const uint8_t id[WM_RCCL_ID_BYTES], int rank, int world) {
  if (!h || !id || world < 1 || rank < 0 || rank >= world) return WM_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  if (ncclCommInitRank(&h->comm.nccl, world, u, rank) != ncclSuccess) return fail(h, WM_ERR_COMM, "ncclCommInitRank failed");
  h->comm.kind = 1; h->comm.rank = rank; h->comm.world = world;
  return WM_OK;
}

extern "C" wm_local_group* wm_local_group_create(int world) {
  wm_local_group* g = new wm_local_group();
  g->world = world;
  g->recv.assign(world, nullptr);
  pthread_barrier_init(&g->bar, nullptr, world);
  return g;
}
extern "C" void wm_local_group_destroy(wm_local_group* g) {
  if (!g) return;
  pthread_barrier_destroy(&g->bar);
  delete g;
}
extern "C" wm_status wm_comm_init_local(wm_handle* h, wm_local_group* g, int rank) {
  if (!h || !g || rank < 0 || rank >= g->world) return WM_ERR_INVALID;
  h->comm.kind = 2; h->comm.rank = rank; h->comm.world = g->world; h->comm.grp = g;
  return WM_OK;
}

extern "C" wm_status wm_allgather(wm_handle* h, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
  if (!h || !send || !recv || bytes_per_rank == 0) return WM_ERR_INVALID;
  return comm_allgather(h, send, recv, bytes_per_rank, (hipStream_t)stream);
}

// ====================================================================================== profiling
extern "C" wm_status wm_profile_enable(wm_handle* h, int on) {
  if (!h) return WM_ERR_INVALID;
  h->prof = on != 0;
  return WM_OK;
}
extern "C" wm_status wm_profile_read(wm_handle* h, int kind, double* total_ms, int64_t* launches) {
  if (!h || kind < 0 || kind >= wm_handle::NKIND) return WM_ERR_INVALID;
  double tot = 0;
  for (size_t i = 0; i < h->ev_used[kind]; ++i) {
    float ms = 0;
    HIPCHK(h, hipEventSynchronize(h->ev[kind][i].b));
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev[kind][i].a, h->ev[kind][i].b));
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = (int64_t)h->ev_used[kind];
  return WM_OK;
}

// ====================================================================================== operator-level entry points
extern "C" wm_status wm_op_gemm(int dtype, int epi, const void* A, const void* Wp, void* C, const float* bias, const float* gamma,
                                int M, int N, int K, void* stream) {
  WmGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.W = Wp; a.C = C; a.bias = bias; a.gamma = gamma; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ldc = N;
  if (wm_tuning[WM_TUNE_OP_LDPAD] > 0) a.lda = a.ldw = K + wm_tuning[WM_TUNE_OP_LDPAD];  // wm_op_gemm only: operand row pitch (elements)
  a.dtype = dtype; a.epi = epi;
  return wm_launch_gemm(a, (hipStream_t)stream) == hipSuccess ? WM_OK : WM_ERR_HIP;
}
extern "C" wm_status wm_op_gemm_resid_ln(int dtype, const void* A, const void* Wp, float* X, const float* bias, const float* gamma, const float* ln_w,
                                         const float* ln_b, float ln_eps, void* ln_out, float* stats, int* sync, int M, int N, int K, int* fused_out,
                                         void* stream) {
  WmGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.W = Wp; a.C = X; a.bias = bias; a.gamma = gamma; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ldc = N;
  a.dtype = dtype; a.epi = WM_EPI_RESID;
  a.ln_out = ln_out; a.ln_ld = N; a.ln_w = ln_w; a.ln_b = ln_b; a.ln_eps = ln_eps; a.ln_stats = stats; a.ln_sync = sync;
  a.ln_fallback = sync + ((size_t)M / 16 + 2) * 2;
  const bool fuse = wm_gemm_fuses_ln(a);
  if (fused_out) *fused_out = fuse ? 1 : 0;
  if (!fuse) a.ln_out = nullptr;
  if (wm_launch_gemm(a, (hipStream_t)stream) != hipSuccess) return WM_ERR_HIP;
  if (fuse && wm_launch_gemm_ln_fallback(a, (hipStream_t)stream) != hipSuccess) return WM_ERR_HIP;
  return WM_OK;
}
extern "C" wm_status wm_op_gemm_qkv(int dtype, const void* A, const void* Wp, const float* bias, void* q, void* k, void* v,
                                    const float* qn_w, const float* qn_b, const float* kn_w, const float* kn_b, const float* rope_cos,
                                    const float* rope_sin, int M, int H, int K, int tokens_per_view, int patch_start, int grid_w,
                                    float q_scale, void* stream) {
  WmGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.W = Wp; a.bias = bias; a.M = M; a.N = 3 * H * 64; a.K = K; a.lda = K; a.ldw = K; a.dtype = dtype; a.epi = WM_EPI_QKV;
  a.qkv.q = q; a.qkv.k = k; a.qkv.v = v; a.qkv.qn_w = qn_w; a.qkv.qn_b = qn_b; a.qkv.kn_w = kn_w; a.qkv.kn_b = kn_b;
  a.qkv.rope_cos = rope_cos; a.qkv.rope_sin = rope_sin; a.qkv.M = M; a.qkv.H = H; a.qkv.head_stride = M;
  a.qkv.tokens_per_view = tokens_per_view; a.qkv.patch_start = patch_start; a.qkv.grid_w = grid_w; a.qkv.q_scale = q_scale; a.qkv.dtype = dtype;
  return wm_launch_gemm(a, (hipStream_t)stream) == hipSuccess ? WM_OK : WM_ERR_HIP;
}
extern "C" wm_status wm_op_attention(int dtype, const void* Q, const void* K, const void* V, void* O, int H, int q_rows, int seq_len,
                                     int kv_chunks, int kv_rows_per_chunk, void* stream) {
  WmAttnArgs a;
  memset(&a, 0, sizeof(a));
  a.Q = Q; a.K = K; a.V = V; a.O = O; a.H = H; a.q_rows = q_rows; a.seq_len = seq_len; a.q_head_stride = q_rows;
  a.kv_chunks = kv_chunks; a.dtype = dtype;
  if (kv_chunks > 1) {
    a.kv_head_stride = kv_rows_per_chunk; a.kv_rows_per_chunk = kv_rows_per_chunk; a.kv_chunk_stride = (long long)H * kv_rows_per_chunk * 64;
  } else {
    a.kv_head_stride = q_rows;
  }
  return wm_launch_attention(a, (hipStream_t)stream) == hipSuccess ? WM_OK : WM_ERR_HIP;
}
extern "C" wm_status wm_op_attention_split(int dtype, const void* Q, const void* K, const void* V, void* O, int H, int q_rows, int seq_len,
                                           int kv_chunks, int kv_rows_per_chunk, int kv_splits, float* part_o, float* part_ml, void* stream) {
  WmAttnArgs a;
  memset(&a, 0, sizeof(a));
  a.Q = Q; a.K = K; a.V = V; a.O = O; a.H = H; a.q_rows = q_rows; a.seq_len = seq_len; a.q_head_stride = q_rows;
  a.kv_chunks = kv_chunks; a.dtype = dtype; a.kv_splits = kv_splits; a.max_splits = WM_ATTN_MAX_SPLITS; a.part_o = part_o; a.part_ml = part_ml;
  if (kv_chunks > 1) {
    a.kv_head_stride = kv_rows_per_chunk; a.kv_rows_per_chunk = kv_rows_per_chunk; a.kv_chunk_stride = (long long)H * kv_rows_per_chunk * 64;
  } else {
    a.kv_head_stride = q_rows;
  }
  return wm_launch_attention(a, (hipStream_t)stream) == hipSuccess ? WM_OK : WM_ERR_HIP;
}
extern "C" wm_status wm_op_attention_ex(int dtype, const void* Q, const void* K, const void* V, void* O, int H, int q_rows, int seq_len,
                                        int kv_chunks, int kv_rows_per_chunk, int kv_splits, float* part_o, float* part_ml, int* unit_flags,
                                        void* stream) {
  WmAttnArgs a;
  memset(&a, 0, sizeof(a));
  a.Q = Q; a.K = K; a.V = V; a.O = O; a.H = H; a.q_rows = q_rows; a.seq_len = seq_len; a.q_head_stride = q_rows;
  a.kv_chunks = kv_chunks; a.dtype = dtype; a.kv_splits = kv_splits; a.max_splits = WM_ATTN_MAX_SPLITS; a.part_o = part_o; a.part_ml = part_ml;
  a.unit_flags = unit_flags;
  // the second half of the caller's flag buffer holds the sticky hints (wm_op_attention_flag_count counts both halves): a caller that
  // reuses one zero-initialised buffer across calls gets the forward's behaviour, one that clears it per call the hint-free one
  if (unit_flags) {
    const size_t nb = wm_attention_max_blocks(q_rows, seq_len, H);
    a.unit_hint = unit_flags + nb;
    a.unit_stat = unit_flags + 2 * nb;
    // the forward's host policy (backbone_block) for ONE buffer at a time, when the tuning key attn_op_policy is 1 (tools/bench_attn_v4.py)
    const bool policy = wm_tuning[WM_TUNE_ATTN_OP_POLICY] == 1;
    static int* mirror = nullptr;
    static const int* key = nullptr;
    static int seen = 0, ttl = 0;
    if (!mirror && hipHostMalloc((void**)&mirror, 4, hipHostMallocDefault) != hipSuccess) return WM_ERR_HIP;
    if (key != unit_flags) { key = unit_flags; seen = 0; ttl = 0; *mirror = 0; }
    const int cur = *mirror, delta = cur - seen;
    seen = cur;
    const long units = (long)((seq_len + 511) / 512) * (q_rows / (seq_len > 0 ? seq_len : 1)) * H;
    if (delta < 0) ttl = 0;                                   // the caller cleared its buffer
    else if (delta > 0 && (long)delta * 4 >= units) ttl = WM_ATTN_HINT_TTL;
    const bool general_only = policy && ttl > 0;
    if (general_only) --ttl;
    int* stat = a.unit_stat;
    if (general_only) { a.unit_flags = nullptr; a.unit_hint = nullptr; a.unit_stat = nullptr; }
    if (kv_chunks > 1) {
      a.kv_head_stride = kv_rows_per_chunk; a.kv_rows_per_chunk = kv_rows_per_chunk; a.kv_chunk_stride = (long long)H * kv_rows_per_chunk * 64;
    } else {
      a.kv_head_stride = q_rows;
    }
    if (wm_launch_attention(a, (hipStream_t)stream) != hipSuccess) return WM_ERR_HIP;
    return hipMemcpyAsync(mirror, stat, 4, hipMemcpyDeviceToHost, (hipStream_t)stream) == hipSuccess ? WM_OK : WM_ERR_HIP;
  }
  if (kv_chunks > 1) {
    a.kv_head_stride = kv_rows_per_chunk; a.kv_rows_per_chunk = kv_rows_per_chunk; a.kv_chunk_stride = (long long)H * kv_rows_per_chunk * 64;
  } else {
    a.kv_head_stride = q_rows;
  }
  return wm_launch_attention(a, (hipStream_t)stream) == hipSuccess ? WM_OK : WM_ERR_HIP;
}
// ints of the caller's buffer: [flags | sticky hints | 1 fallback counter (+ pad)]
extern "C" size_t wm_op_attention_flag_count(int q_rows, int seq_len, int H) { return 2 * wm_attention_max_blocks(q_rows, seq_len, H) + 4; }
// resize (align_corners bilinear + position tables) to 16 bits, then the 32-channel 3x3 conv on it: the unfused form of
// wm_op_conv3x3_up for Cout == 32.  up16: Hi * Wi * N * Cin 16-bit elements + 16 B of scratch (zeroed here).
extern "C" wm_status wm_op_up_conv_n32(int dtype, const float* x, const void* w16, const float* bias, float* y, int N, int Hs, int Ws, int Hi,
                                       int Wi, int Cin, const float* addx, const float* addy, int relu_out, void* up16, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (Cin % 64 || !up16) return WM_ERR_INVALID;
  const size_t n16 = (size_t)N * Hi * Wi * Cin;
  uint16_t* zero = (uint16_t*)up16 + ((n16 + 7) & ~(size_t)7);
  if (hipMemsetAsync(zero, 0, 16, s) != hipSuccess) return WM_ERR_HIP;
  if (wm_launch_bilinear16(x, up16, N, Hs, Ws, Hi, Wi, Cin, addx, addy, dtype, s) != hipSuccess) return WM_ERR_HIP;
  WmConvN32Args a;
  memset(&a, 0, sizeof(a));
  a.x = (const uint16_t*)up16; a.w = (const uint16_t*)w16; a.bias = bias; a.y = y; a.zero = zero;
  a.N = N; a.H = Hi; a.W = Wi; a.Cin = Cin; a.relu_out = relu_out; a.dtype = dtype;
  return wm_launch_conv3x3_n32_in16(a, s) == hipSuccess ? WM_OK : WM_ERR_HIP;
}
// splat assembly of prepare_splats (rasterization.py:389-498) as an operator: used by tools/micro/splat_hazard_repro.cpp
extern "C" wm_status wm_op_gs_splat(const float* gp, const float* img, const float* depth, const float* cam, float* means, float* quats,
                                    float* scales, float* opac, float* sh, float* wts, int N, int H, int W, void* stream) {
  return wm_launch_gs_splat(gp, img, depth, cam, means, quats, scales, opac, sh, wts, N, H, W, (hipStream_t)stream) == hipSuccess ? WM_OK : WM_ERR_HIP;
}
extern "C" size_t wm_prune_gs_workspace_bytes(size_t n) { return wm_prune_workspace_bytes(n); }
extern "C" wm_status wm_prune_gs(const float* means, const float* quats, const float* scales, const float* opacities, const float* sh,
                                 const float* weights, int n, float voxel_size, float* out_means, float* out_quats, float* out_scales,
                                 float* out_opacities, float* out_sh, int* n_voxels, void* workspace, size_t workspace_bytes, void* stream) {
  if (n < 0 || !(voxel_size > 0.f) || !n_voxels) return WM_ERR_INVALID;
  if (n > 0 && (!means || !quats || !scales || !sh || !weights || !out_means || !out_quats || !out_scales || !out_opacities || !out_sh || !workspace))
    return WM_ERR_INVALID;
  const hipError_t e = wm_launch_prune_gs(means, quats, scales, opacities, sh, weights, n, voxel_size, out_means, out_quats, out_scales,
                                          out_opacities, out_sh, n_voxels, workspace, workspace_bytes, (hipStream_t)stream);
  return e == hipSuccess ? WM_OK : e == hipErrorInvalidValue ? WM_ERR_INVALID : WM_ERR_HIP;
}
extern "C" size_t wm_rasterize_workspace_bytes(int n_gaussians, int n_cameras, int width, int height, size_t max_isects) {
  return wm_raster_workspace_bytes(n_gaussians, n_cameras, width, height, max_isects);
}
extern "C" wm_status wm_rasterize_splats(const float* means, const float* quats, const float* scales, const float* opacities,
                                         const float* colors, int colors_are_sh0, int n_gaussians, const float* viewmats, const float* Ks,
                                         int n_cameras, int width, int height, float* out_rgb, float* out_depth, float* out_alpha,
                                         int* radii_out, void* workspace, size_t workspace_bytes, size_t max_isects,
                                         unsigned long long* n_isects, void* stream) {
  if (!means || !quats || !scales || !opacities || !colors || !viewmats || !Ks || !out_rgb || !out_depth || !out_alpha || !workspace)
    return WM_ERR_INVALID;
  WmRasterArgs a;
  memset(&a, 0, sizeof(a));
  a.means = means; a.quats = quats; a.scales = scales; a.opacities = opacities; a.colors = colors; a.is_sh = colors_are_sh0;
  a.N = n_gaussians; a.viewmats = viewmats; a.Ks = Ks; a.C = n_cameras; a.width = width; a.height = height;
  a.out_rgb = out_rgb; a.out_depth = out_depth; a.out_alpha = out_alpha; a.radii_out = radii_out;
  a.workspace = workspace; a.workspace_bytes = workspace_bytes; a.max_isects = max_isects;
  unsigned long long n = 0;
  const hipError_t e = wm_launch_rasterize(a, (hipStream_t)stream, &n);
  if (n_isects) *n_isects = n;
  if (e == hipErrorInvalidValue) return WM_ERR_INVALID;
  if (e != hipSuccess) return WM_ERR_HIP;
  return n > max_isects ? WM_ERR_STATE : WM_OK;  // WM_ERR_STATE: the workspace is too small for *n_isects pairs, nothing was rendered
}
extern "C" wm_status wm_op_layernorm(const float* x, void* y, const float* w, const float* b, int rows, int D, float eps, int out_f32,
                                     int dtype, void* stream) {
  WmLnArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.y = y; a.w = w; a.b = b; a.D = D; a.ld_in = D; a.ld_out = D; a.eps = eps; a.groups = 1; a.rows_per_group = rows;
  a.out_f32 = out_f32; a.dtype = dtype;
  return wm_launch_layernorm(a, (hipStream_t)stream) == hipSuccess ? WM_OK : WM_ERR_HIP;
}
extern "C" wm_status wm_op_qkv_post(int dtype, const float* qkv, void* q, void* k, void* v, const float* qn_w, const float* qn_b,
                                    const float* kn_w, const float* kn_b, const float* rope_cos, const float* rope_sin, int M, int H,
                                    int tokens_per_view, int patch_start, int grid_w, float q_scale, void* stream) {
  WmQkvArgs a;
  memset(&a, 0, sizeof(a));
  a.qkv = qkv; a.q = q; a.k = k; a.v = v; a.qn_w = qn_w; a.qn_b = qn_b; a.kn_w = kn_w; a.kn_b = kn_b; a.rope_cos = rope_cos; a.rope_sin = rope_sin;
  a.M = M; a.H = H; a.head_stride = M; a.tokens_per_view = tokens_per_view; a.patch_start = patch_start; a.grid_w = grid_w; a.q_scale = q_scale;
  a.dtype = dtype;
  return wm_launch_qkv_post(a, (hipStream_t)stream) == hipSuccess ? WM_OK : WM_ERR_HIP;
}
extern "C" wm_status wm_op_conv(int dtype, const float* x, const void* w16, const float* bias, const float* resid, const float* resid2,
                                float* y, int N, int Hi, int Wi, int Cin, int Cout, int ksize, int stride, int pad, int relu_in,
                                int resid_relu, void* stream) {
  WmConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.w = w16; a.bias = bias; a.resid = resid; a.resid2 = resid2; a.y = y; a.N = N; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Cout = Cout;
  a.ksize = ksize; a.stride = stride; a.pad = pad; a.Ho = (Hi + 2 * pad - ksize) / stride + 1; a.Wo = (Wi + 2 * pad - ksize) / stride + 1;
  a.relu_in = relu_in; a.resid_relu = resid_relu; a.dtype = dtype;
  return wm_launch_conv(a, (hipStream_t)stream) == hipSuccess ? WM_OK : WM_ERR_HIP;
}
// wm_op_conv with the 16-bit tensor forms of the register-staged 3x3 kernel: x is 16-bit NHWC when in16 (relu_in must be 0), y is 16-bit NHWC
// when out16 (refused with WM_ERR_INVALID when the launch would take a kernel without that form: wm_conv3x3_out16_ok)
extern "C" wm_status wm_op_conv_ex(int dtype, const void* x, int in16, const void* w16, const float* bias, const float* resid, const float* resid2,
                                   void* y, int out16, int N, int Hi, int Wi, int Cin, int Cout, int relu_in, int resid_relu, int relu_out, void* stream) {
  WmConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = (const float*)x; a.w = w16; a.bias = bias; a.resid = resid; a.resid2 = resid2; a.y = (float*)y; a.N = N; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Cout = Cout;
  a.ksize = 3; a.stride = 1; a.pad = 1; a.Ho = Hi; a.Wo = Wi; a.relu_in = relu_in; a.resid_relu = resid_relu; a.relu_out = relu_out; a.dtype = dtype;
  a.in16 = in16; a.out16 = out16;
  if ((in16 || out16) && !wm_conv3x3_out16_ok(a)) return WM_ERR_INVALID;
  return wm_launch_conv(a, (hipStream_t)stream) == hipSuccess ? WM_OK : WM_ERR_HIP;
}
// Conv2d(C, Co, 3, padding=1)(F.interpolate(x, (Ho, Wo), bilinear, align_corners=True)) in the tap form of upconv.hip on a 16-bit (f16) NHWC x:
// wt16: scratch of 9 Co C 16-bit elements (the tap-major weight copy), y16: scratch of N Hi Wi 9 Co 16-bit elements
extern "C" wm_status wm_op_upconv3x3_tap(int dtype, const void* x16, const void* w16, const float* bias, float* out, int N, int Hi, int Wi, int Ho,
                                         int Wo, int C, int Co, void* wt16, void* y16, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (dtype != WM_DT_F16 || !x16 || !w16 || !out || !wt16 || !y16 || C % 64) return WM_ERR_INVALID;
  if (wm_launch_repack_tap_major(w16, wt16, Co, C, s) != hipSuccess) return WM_ERR_HIP;
  WmGemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = x16; g.W = wt16; g.C = y16; g.M = N * Hi * Wi; g.N = 9 * Co; g.K = C; g.lda = C; g.ldw = C; g.ldc = 9 * Co; g.dtype = dtype; g.epi = WM_EPI_T16;
  if (wm_launch_gemm(g, s) != hipSuccess) return WM_ERR_HIP;
  return wm_launch_upconv_gather(y16, bias, out, N, Hi, Wi, Ho, Wo, Co, s) == hipSuccess ? WM_OK : WM_ERR_HIP;
}
// Conv2d(Cm, 256, 3, padding=1, bias=False)(ConvTranspose2d(Cin, Cm, k, stride=k)(tokens)) composed at the token resolution (build_tconv):
// tokens16 [N][gh][gw][Cin] 16-bit device; wct [Cin][Cm][k][k], bct [Cm], wrn [256][Cm][3][3]: HOST fp32 in torch's layouts; out fp32 [N][k gh][k gw][256]
extern "C" wm_status wm_op_tconv(int dtype, const void* tokens16, const float* wct, const float* bct, const float* wrn, float* out, int N, int gh, int gw,
                                 int k, int Cin, int Cm, const void* zero16, void* stream) {
  if (!tokens16 || !wct || !bct || !wrn || !out || !zero16) return WM_ERR_INVALID;
  TconvPack t;
  hipStream_t s = (hipStream_t)stream;
  if (build_tconv(dtype, k, Cin, Cm, 256, wct, bct, wrn, t, s) != hipSuccess) return WM_ERR_HIP;
  const hipError_t e = launch_tconv(t, dtype, tokens16, out, N, gh, gw, zero16, s);
  (void)hipStreamSynchronize(s);
  free_tconv(t);
  return e == hipSuccess ? WM_OK : WM_ERR_HIP;
}
extern "C" wm_status wm_op_upconv_gather(const void* y16, const float* bias, float* out, int N, int Hi, int Wi, int Ho, int Wo, int Co, void* stream) {
  return wm_launch_upconv_gather(y16, bias, out, N, Hi, Wi, Ho, Wo, Co, (hipStream_t)stream) == hipSuccess ? WM_OK : WM_ERR_HIP;
}
extern "C" wm_status wm_op_conv3x3_gemm16(int dtype, const void* x16, const void* w16, const float* bias, const float* resid, int resid_relu,
                                          const float* resid2, void* y, int out16, int relu_out, int N, int H, int W, int Cin, int Cout,
                                          const void* zero16, void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin % 64 || (Cout & 7) || !zero16 || !x16 || !w16 || !y) return WM_ERR_INVALID;
  WmGemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = x16; g.W = w16; g.C = y; g.bias = bias; g.M = N * H * W; g.N = Cout; g.K = 9 * Cin; g.lda = Cin; g.ldw = 9 * Cin; g.ldc = Cout;
  g.dtype = dtype; g.epi = WM_EPI_CONV; g.cv_h = H; g.cv_w = W; g.cv_cin = Cin; g.cv_zero = zero16;
  g.cv_resid = resid; g.cv_resid2 = resid2; g.cv_resid_relu = resid_relu; g.out16 = out16; g.relu = relu_out;
  return wm_launch_gemm(g, (hipStream_t)stream) == hipSuccess ? WM_OK : WM_ERR_HIP;
}
extern "C" wm_status wm_op_conv3x3_up(int dtype, const float* x, const void* w16, const float* bias, float* y, int N, int Hs, int Ws, int Hi, int Wi,
                                      int Cin, int Cout, const float* addx, const float* addy, void* stream) {
  WmConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.w = w16; a.bias = bias; a.y = y; a.N = N; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Cout = Cout; a.ksize = 3; a.stride = 1; a.pad = 1;
  a.Ho = Hi; a.Wo = Wi; a.dtype = dtype; a.up_hs = Hs; a.up_ws = Ws; a.up_addx = addx; a.up_addy = addy;
  return wm_launch_conv(a, (hipStream_t)stream) == hipSuccess ? WM_OK : WM_ERR_HIP;
}
// ---------------------------------------------------------------- image ingest
// Pillow's precompute_coeffs + normalize_coeffs_8bpc (libImaging/Resample.c) for the BICUBIC filter, in double
// precision with contraction off so that every rounding matches Pillow's C build (and oracle/ingest_ref.py).
#pragma clang fp contract(off)
static double wm_bicubic(double x) {
  const double a = -0.5;
  if (x < 0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}
static int wm_resample_coeffs(int in_size, int out_size, std::vector<int>& bounds, std::vector<int>& kk) {
#pragma clang fp contract(off)
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * filterscale;
  const int ksize = (int)std::ceil(support) * 2 + 1;
  bounds.assign((size_t)out_size * 2, 0);
  kk.assign((size_t)out_size * ksize, 0);
  std::vector<double> w((size_t)ksize);
  const double ss = 1.0 / filterscale;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      w[x] = wm_bicubic((x + xmin - center + 0.5) * ss);
      ww += w[x];
    }
    for (int x = 0; x < xmax; ++x) {
      double v = w[x];
      if (ww != 0.0) v /= ww;
      kk[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (double)(1 << 22)) : (int)(0.5 + v * (double)(1 << 22));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  return ksize;
}
// inference_utils.py:70-83; Python's round() is half-to-even = nearbyint in the default rounding mode
static void wm_ingest_dims(int H, int W, int mode, int target, int* sw, int* sh, int* hf, int* wf, int* ry0, int* rx0) {
#pragma clang fp contract(off)
  if (mode == 1) {  // pad
    if (W >= H) { *sw = target; *sh = (int)std::nearbyint((double)H * ((double)target / (double)W) / 14.0) * 14; }
    else { *sh = target; *sw = (int)std::nearbyint((double)W * ((double)target / (double)H) / 14.0) * 14; }
    *hf = target > *sh ? target : *sh; *wf = target > *sw ? target : *sw;           // pad up to the square; never crops
    *ry0 = -((*hf - *sh) / 2); *rx0 = -((*wf - *sw) / 2);
  } else {          // crop
    *sw = target; *sh = (int)std::nearbyint((double)H * ((double)target / (double)W) / 14.0) * 14;
    *wf = *sw; *hf = *sh > target ? target : *sh;
    *ry0 = *sh > target ? (*sh - target) / 2 : 0; *rx0 = 0;
  }
}
extern "C" wm_status wm_preprocess_image_size(int H, int W, int mode, int output_size, int* out_h, int* out_w) {
  if (H <= 0 || W <= 0 || (mode != 0 && mode != 1) || output_size <= 0 || !out_h || !out_w) return WM_ERR_INVALID;
  int sw, sh, ry0, rx0;
  wm_ingest_dims(H, W, mode, output_size, &sw, &sh, out_h, out_w, &ry0, &rx0);
  return (sw > 0 && sh > 0) ? WM_OK : WM_ERR_INVALID;
}
extern "C" size_t wm_preprocess_image_workspace_bytes(int H, int W, int mode, int output_size) {
  int sw, sh, hf, wf, ry0, rx0;
  if (H <= 0 || W <= 0) return 0;
  wm_ingest_dims(H, W, mode, output_size, &sw, &sh, &hf, &wf, &ry0, &rx0);
  if (sw <= 0 || sh <= 0) return 0;
  auto ks = [](int in, int out) { const double sc = (double)in / out; return (int)std::ceil(2.0 * (sc < 1.0 ? 1.0 : sc)) * 2 + 1; };
  const size_t tabs = ((size_t)sw * (2 + ks(W, sw)) + (size_t)sh * (2 + ks(H, sh))) * sizeof(int);
  return (size_t)H * sw * 3 + 256 + tabs + 256;
}
extern "C" wm_status wm_preprocess_image(const unsigned char* rgb, int H, int W, int mode, int output_size, float* out, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  if (!rgb || !out || !workspace || H <= 0 || W <= 0 || (mode != 0 && mode != 1)) return WM_ERR_INVALID;
  if (workspace_bytes < wm_preprocess_image_workspace_bytes(H, W, mode, output_size)) return WM_ERR_INVALID;
  int sw, sh, hf, wf, ry0, rx0;
  wm_ingest_dims(H, W, mode, output_size, &sw, &sh, &hf, &wf, &ry0, &rx0);
  if (sw <= 0 || sh <= 0) return WM_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  std::vector<int> bh, kh, bv, kv;
  const int ksh = sw != W ? wm_resample_coeffs(W, sw, bh, kh) : 0;
  const int ksv = sh != H ? wm_resample_coeffs(H, sh, bv, kv) : 0;
  char* ws = (char*)workspace;
  unsigned char* tmp = (unsigned char*)ws;
  size_t off = ((size_t)H * sw * 3 + 255) / 256 * 256;
  int* d_bh = (int*)(ws + off); off += bh.size() * 4;
  int* d_kh = (int*)(ws + off); off += kh.size() * 4;
  int* d_bv = (int*)(ws + off); off += bv.size() * 4;
  int* d_kv = (int*)(ws + off); off += kv.size() * 4;
  // the tables are small (a few hundred KB at most); pageable copies are synchronous with respect to the host, which
  // keeps the vectors alive long enough
  if (ksh) {
    if (hipMemcpyAsync(d_bh, bh.data(), bh.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess) return WM_ERR_HIP;
    if (hipMemcpyAsync(d_kh, kh.data(), kh.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess) return WM_ERR_HIP;
  }
  if (ksv) {
    if (hipMemcpyAsync(d_bv, bv.data(), bv.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess) return WM_ERR_HIP;
    if (hipMemcpyAsync(d_kv, kv.data(), kv.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess) return WM_ERR_HIP;
  }
  if (hipStreamSynchronize(s) != hipSuccess) return WM_ERR_HIP;
  const unsigned char* hsrc = rgb;
  if (ksh) {
    if (wm_launch_resample_h(rgb, tmp, H, W, sw, d_bh, d_kh, ksh, s) != hipSuccess) return WM_ERR_HIP;
    hsrc = tmp;
  }
  if (wm_launch_resample_v_tensor(hsrc, out, H, sw, sh, hf, wf, ry0, rx0, ksv ? d_bv : nullptr, ksv ? d_kv : nullptr, ksv, s) != hipSuccess)
    return WM_ERR_HIP;
  return WM_OK;
}

extern "C" size_t wm_confidence_mask_workspace_bytes(size_t n) { return wm_confidence_mask_workspace(n); }
extern "C" wm_status wm_confidence_mask(const float* conf, size_t n, float conf_threshold_percent, unsigned char* mask, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  if (!conf || !mask || !workspace || workspace_bytes < wm_confidence_mask_workspace(n)) return WM_ERR_INVALID;
  if (n == 0) return WM_OK;
  // infer.py:44-48: keep the top ceil(N (100 - p) / 100), at least one; p <= 0 keeps everything
  double keep = (double)n;
  if (conf_threshold_percent > 0.f) keep = std::ceil((double)n * (100.0 - (double)conf_threshold_percent) / 100.0);
  size_t K = keep < 1.0 ? 1 : (size_t)keep;
  if (K > n) K = n;
  return wm_launch_confidence_mask(conf, n, (unsigned int)K, mask, workspace, (hipStream_t)stream) == hipSuccess ? WM_OK : WM_ERR_HIP;
}
extern "C" wm_status wm_depth_to_world(const float* depth, const float* extrinsic, const float* intrinsic, float* world, float* cam,
                                       unsigned char* mask, int B, int H, int W, float eps, void* stream) {
  if (!depth || !extrinsic || !intrinsic || B < 0 || H < 0 || W < 0) return WM_ERR_INVALID;
  return wm_launch_depth_to_world(depth, extrinsic, intrinsic, world, cam, mask, B, H, W, eps, (hipStream_t)stream) == hipSuccess ? WM_OK : WM_ERR_HIP;
}
extern "C" wm_status wm_op_bilinear(const float* in, float* out, int N, int Hi, int Wi, int Ho, int Wo, int C, void* stream) {
  return wm_launch_bilinear(in, out, N, Hi, Wi, Ho, Wo, C, nullptr, nullptr, (hipStream_t)stream) == hipSuccess ? WM_OK : WM_ERR_HIP;
}
extern "C" wm_status wm_op_linear_f32(const float* X, const float* Wp, const float* b, float* Y, int M, int N, int K, int ldx, int pre_act,
                                      int post_act, void* stream) {
  return wm_launch_linear_f32(X, Wp, b, Y, M, N, K, ldx, N, pre_act, post_act, nullptr, 0, (hipStream_t)stream) == hipSuccess ? WM_OK : WM_ERR_HIP;
}
