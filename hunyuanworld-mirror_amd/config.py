"""Architecture description of the WorldMirror forward path and its parameter list.

The reference exposes only the top-level ctor kwargs (src/models/models/worldmirror.py:17-34);
everything else is a default buried in its sub-modules.  ``WMConfig`` gathers those defaults
(reference file:line in the comments) so that one object describes every shape the HIP
library, the weight generator and the CPU oracle need.  ``param_spec`` lists every tensor of the
reference ``state_dict`` (names verified against the instantiated reference model).
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field, asdict
from typing import List, Tuple


@dataclass
class WMConfig:
    # --- reference ctor kwargs (worldmirror.py:17-34)
    img_size: int = 518
    patch_size: int = 14
    embed_dim: int = 1024
    gs_dim: int = 256
    enable_cond: bool = True
    enable_cam: bool = True
    enable_pts: bool = True
    enable_depth: bool = True
    enable_norm: bool = True
    enable_gs: bool = False
    # --- VisualGeometryTransformer defaults (visual_transformer.py:48-70)
    depth: int = 24
    num_heads: int = 16
    mlp_ratio: int = 4
    num_register_tokens: int = 4
    intermediate_idxs: Tuple[int, ...] = (4, 11, 17, 23)
    rope_freq: float = 100.0
    # --- DINOv2 encoder (vision_transformer.py:364-375 vit_large)
    dino_depth: int = 24
    dino_heads: int = 16
    # --- CameraHead (camera_head.py:16-27)
    cam_trunk_depth: int = 4
    cam_heads: int = 16
    cam_steps: int = 4
    # --- DPTHead (dense_head.py:34-46)
    dpt_features: int = 256
    dpt_out_channels: Tuple[int, ...] = (256, 512, 1024, 1024)

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads

    @property
    def patch_start_idx(self) -> int:
        # visual_transformer.py:100-103
        return 1 + self.num_register_tokens + (2 if self.enable_cond else 0)

    @property
    def pos_grid(self) -> int:
        return self.img_size // self.patch_size

    def to_dict(self):
        d = asdict(self)
        d["intermediate_idxs"] = list(self.intermediate_idxs)
        d["dpt_out_channels"] = list(self.dpt_out_channels)
        return d

    @staticmethod
    def tiny(**kw) -> "WMConfig":
        """Scaled-down architecture used for committed golden fixtures:
        every code path of the full model at 1/8 width, head_dim kept at 64."""
        base = dict(
            img_size=70, patch_size=14, embed_dim=128, gs_dim=64, depth=4, num_heads=2,
            intermediate_idxs=(0, 1, 2, 3), dino_depth=2, dino_heads=2,
            cam_trunk_depth=2, cam_heads=4, dpt_features=64, dpt_out_channels=(64, 128, 256, 256),
        )
        base.update(kw)
        return WMConfig(**base)


def _block(spec, p, D, hidden, qk_norm_dim=None):
    spec[p + "norm1.weight"] = (D,)
    spec[p + "norm1.bias"] = (D,)
    spec[p + "attn.qkv.weight"] = (3 * D, D)
    spec[p + "attn.qkv.bias"] = (3 * D,)
    if qk_norm_dim:
        spec[p + "attn.q_norm.weight"] = (qk_norm_dim,)
        spec[p + "attn.q_norm.bias"] = (qk_norm_dim,)
        spec[p + "attn.k_norm.weight"] = (qk_norm_dim,)
        spec[p + "attn.k_norm.bias"] = (qk_norm_dim,)
    spec[p + "attn.proj.weight"] = (D, D)
    spec[p + "attn.proj.bias"] = (D,)
    spec[p + "ls1.gamma"] = (D,)
    spec[p + "norm2.weight"] = (D,)
    spec[p + "norm2.bias"] = (D,)
    spec[p + "mlp.fc1.weight"] = (hidden, D)
    spec[p + "mlp.fc1.bias"] = (hidden,)
    spec[p + "mlp.fc2.weight"] = (D, hidden)
    spec[p + "mlp.fc2.bias"] = (D,)
    spec[p + "ls2.gamma"] = (D,)


def _dpt(spec, p, dim_in, F, oc, out_dim, is_gs):
    spec[p + "norm.weight"] = (dim_in,)
    spec[p + "norm.bias"] = (dim_in,)
    for i, c in enumerate(oc):
        spec[p + f"projects.{i}.weight"] = (c, dim_in, 1, 1)
        spec[p + f"projects.{i}.bias"] = (c,)
    spec[p + "resize_layers.0.weight"] = (oc[0], oc[0], 4, 4)
    spec[p + "resize_layers.0.bias"] = (oc[0],)
    spec[p + "resize_layers.1.weight"] = (oc[1], oc[1], 2, 2)
    spec[p + "resize_layers.1.bias"] = (oc[1],)
    spec[p + "resize_layers.3.weight"] = (oc[3], oc[3], 3, 3)
    spec[p + "resize_layers.3.bias"] = (oc[3],)
    for i, c in enumerate(oc):
        spec[p + f"scratch.layer{i + 1}_rn.weight"] = (F, c, 3, 3)
    for r in (1, 2, 3, 4):
        q = p + f"scratch.refinenet{r}."
        spec[q + "out_conv.weight"] = (F, F, 1, 1)
        spec[q + "out_conv.bias"] = (F,)
        units = ("resConfUnit1", "resConfUnit2") if r != 4 else ("resConfUnit2",)
        for u in units:
            for c in ("conv1", "conv2"):
                spec[q + f"{u}.{c}.weight"] = (F, F, 3, 3)
                spec[q + f"{u}.{c}.bias"] = (F,)
    spec[p + "scratch.output_conv1.weight"] = (F // 2, F, 3, 3)
    spec[p + "scratch.output_conv1.bias"] = (F // 2,)
    spec[p + "scratch.output_conv2.0.weight"] = (32, F // 2, 3, 3)
    spec[p + "scratch.output_conv2.0.bias"] = (32,)
    spec[p + "scratch.output_conv2.2.weight"] = (out_dim, 32, 1, 1)
    spec[p + "scratch.output_conv2.2.bias"] = (out_dim,)
    if is_gs:
        spec[p + "input_merger.0.weight"] = (F // 2, 3, 7, 7)
        spec[p + "input_merger.0.bias"] = (F // 2,)


def param_spec(cfg: WMConfig) -> "OrderedDict[str, tuple]":
    """name -> shape for every tensor of the reference state_dict."""
    D = cfg.embed_dim
    H4 = cfg.mlp_ratio * D
    g = cfg.pos_grid
    R = cfg.num_register_tokens
    spec: "OrderedDict[str, tuple]" = OrderedDict()
    v = "visual_geometry_transformer."
    spec[v + "cam_token"] = (1, 2, 1, D)
    spec[v + "reg_token"] = (1, 2, R, D)
    d = v + "patch_embed."
    spec[d + "cls_token"] = (1, 1, D)
    spec[d + "pos_embed"] = (1, 1 + g * g, D)
    spec[d + "register_tokens"] = (1, R, D)
    spec[d + "mask_token"] = (1, D)
    spec[d + "patch_embed.proj.weight"] = (D, 3, cfg.patch_size, cfg.patch_size)
    spec[d + "patch_embed.proj.bias"] = (D,)
    for i in range(cfg.dino_depth):
        _block(spec, d + f"blocks.{i}.", D, H4)
    spec[d + "norm.weight"] = (D,)
    spec[d + "norm.bias"] = (D,)
    if cfg.enable_cond:
        spec[v + "pose_embed.0.weight"] = (D, 7)
        spec[v + "pose_embed.0.bias"] = (D,)
        spec[v + "pose_embed.2.weight"] = (D, D)
        spec[v + "pose_embed.2.bias"] = (D,)
        pp = cfg.patch_size * cfg.patch_size
        spec[v + "depth_embed.proj.2.fc1.weight"] = (4 * D, pp)
        spec[v + "depth_embed.proj.2.fc1.bias"] = (4 * D,)
        spec[v + "depth_embed.proj.2.fc2.weight"] = (D, 4 * D)
        spec[v + "depth_embed.proj.2.fc2.bias"] = (D,)
        spec[v + "ray_embed.0.weight"] = (D, 4)
        spec[v + "ray_embed.0.bias"] = (D,)
        spec[v + "ray_embed.2.weight"] = (D, D)
        spec[v + "ray_embed.2.bias"] = (D,)
    for i in range(cfg.depth):
        _block(spec, v + f"frame_blocks.{i}.", D, H4, cfg.head_dim)
    for i in range(cfg.depth):
        _block(spec, v + f"global_blocks.{i}.", D, H4, cfg.head_dim)
    D2 = 2 * D
    if cfg.enable_cam:
        c = "cam_head."
        for i in range(cfg.cam_trunk_depth):
            _block(spec, c + f"refine_net.{i}.", D2, 4 * D2)
        spec[c + "token_norm.weight"] = (D2,)
        spec[c + "token_norm.bias"] = (D2,)
        spec[c + "out_norm.weight"] = (D2,)
        spec[c + "out_norm.bias"] = (D2,)
        spec[c + "init_token"] = (1, 1, 9)
        spec[c + "param_embed.weight"] = (D2, 9)
        spec[c + "param_embed.bias"] = (D2,)
        spec[c + "adapt_norm_gen.1.weight"] = (3 * D2, D2)
        spec[c + "adapt_norm_gen.1.bias"] = (3 * D2,)
        spec[c + "param_predictor.fc1.weight"] = (D2 // 2, D2)
        spec[c + "param_predictor.fc1.bias"] = (D2 // 2,)
        spec[c + "param_predictor.fc2.weight"] = (9, D2 // 2)
        spec[c + "param_predictor.fc2.bias"] = (9,)
    oc = tuple(cfg.dpt_out_channels)
    if cfg.enable_pts:
        _dpt(spec, "pts_head.", D2, cfg.dpt_features, oc, 4, False)
    if cfg.enable_depth:
        _dpt(spec, "depth_head.", D2, cfg.dpt_features, oc, 2, False)
    if cfg.enable_norm:
        _dpt(spec, "norm_head.", D2, cfg.dpt_features, oc, 4, False)
    if cfg.enable_gs:
        _dpt(spec, "gs_head.", D2, cfg.gs_dim, oc, 2, True)
        spec["gs_renderer.gs_head.0.weight"] = (cfg.gs_dim, cfg.gs_dim // 2, 3, 3)
        spec["gs_renderer.gs_head.2.weight"] = (12, cfg.gs_dim, 1, 1)
        spec["gs_renderer.gs_head.2.bias"] = (12,)
    return spec
