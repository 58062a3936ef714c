"""Deterministic, name-keyed synthetic weights.

No checkpoint exists offline, so parity and benchmarks run on synthetic weights that both the
reference (in the build container) and this package can regenerate bit-identically from the
parameter *name* alone: every tensor is drawn from a numpy Philox stream keyed on crc32(name).

The scales are chosen to make parity tests *sensitive* rather than to mimic the reference's
init (reference init: vision_transformer.py:328-333 trunc_normal 0.02, LayerScale 0.01 in
visual_transformer.py:65 — with gamma 0.01 every block is almost a no-op and kernel bugs hide):
  * Linear / conv weights  N(0, 1/fan_in)
  * LayerNorm weight       1 + 0.1 N(0,1);  biases 0.05 N(0,1)
  * LayerScale gamma       0.3 + 0.05 N(0,1)
  * learned tokens / pos   0.2 N(0,1)
  * the few output layers that feed exp/expm1 or the fov ReLU are damped so that outputs stay in
    a realistic range (random init gives fov = 0 -> focal = inf).
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterator, Tuple

import numpy as np

from .config import WMConfig, param_spec

_TOKENS = ("cam_token", "reg_token", "cls_token", "pos_embed", "register_tokens",
           "mask_token", "init_token")


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[zlib.crc32(name.encode()), seed]))


def _make_param_refinit(name: str, shape: Tuple[int, ...], n: np.ndarray) -> np.ndarray:
    """Statistics of the reference's own initialisation.  DINOv2 encoder: Linear trunc_normal(0.02) with zero bias
    (init_weights_vit_timm, vision_transformer.py:328-333), LayerScale 1.0 (visual_transformer.py:125), pos_embed
    0.02, tokens 1e-6.  Everything else keeps torch's defaults: Linear / Conv weight AND bias ~ U(+-1/sqrt(fan_in))
    (drawn here as a normal of the same variance 1/(3 fan_in)), LayerNorm 1/0, LayerScale 0.01 in the multi-view
    blocks and the camera trunk (visual_transformer.py:65, camera_head.py:24), cam/reg tokens sigma 1e-6 (:247-248).
    Only the fov bias is kept positive (random init gives fov = 0 -> focal = inf)."""
    leaf = name.rsplit(".", 1)[-1]
    dino = ".patch_embed." in name
    if leaf == "pos_embed":
        return (0.02 * np.clip(n, -2, 2)).astype(np.float32)
    if leaf in _TOKENS:
        return (1e-6 * n).astype(np.float32)
    if leaf == "gamma":
        return np.full(shape, 1.0 if dino else 0.01, np.float32)
    is_norm = any(t in name for t in (".norm1.", ".norm2.", ".norm.", "q_norm.", "k_norm.", "token_norm.", "out_norm."))
    if is_norm:
        return np.ones(shape, np.float32) if leaf == "weight" else np.zeros(shape, np.float32)
    if leaf == "bias":
        if dino and "patch_embed.proj" not in name:
            return np.zeros(shape, np.float32)
        # fan_in is not recoverable from the bias shape alone: use the layer's known input widths by family
        fan = _bias_fan_in(name, shape)
        b = (n * np.float32(1.0 / np.sqrt(3.0 * fan))).astype(np.float32)
        if name == "cam_head.param_predictor.fc2.bias":
            b[7:9] = 0.3
        return b
    if dino and len(shape) == 2:
        return (0.02 * np.clip(n, -2, 2)).astype(np.float32)
    fan_in = shape[0] if ("resize_layers.0." in name or "resize_layers.1." in name) else int(np.prod(shape[1:]))
    return (n * np.float32(1.0 / np.sqrt(3.0 * fan_in))).astype(np.float32)


_FAN_CACHE: Dict[str, int] = {}


def _bias_fan_in(name: str, shape) -> int:
    """fan_in of the layer a bias belongs to (from the weight's shape in any spec that contains it)."""
    if not _FAN_CACHE:
        for cfg in (WMConfig(enable_gs=True), WMConfig.tiny(enable_gs=True)):
            for k, sh in param_spec(cfg).items():
                if k.endswith(".weight") and len(sh) >= 2:
                    fan = sh[0] if ("resize_layers.0." in k or "resize_layers.1." in k) else int(np.prod(sh[1:]))
                    _FAN_CACHE[k[:-7] + "|" + str(sh[0] if not ("resize_layers.0." in k or "resize_layers.1." in k) else sh[1])] = fan
    return _FAN_CACHE.get(name[:-5] + "|" + str(shape[0]), 64)


def make_param(name: str, shape: Tuple[int, ...], seed: int = 0, preset: str = "sensitive") -> np.ndarray:
    """fp32 ndarray for one reference parameter name.  preset: "sensitive" (default, see module docstring)
    or "refinit" (the reference's own init statistics)."""
    rng = _rng(name, seed)
    leaf = name.rsplit(".", 1)[-1]
    n = rng.standard_normal(shape, dtype=np.float32)
    if preset == "refinit":
        return _make_param_refinit(name, shape, n)
    if leaf in _TOKENS:
        return 0.2 * n
    if leaf == "gamma":
        return (0.3 + 0.05 * n).astype(np.float32)
    is_norm = any(t in name for t in (".norm1.", ".norm2.", ".norm.", "q_norm.", "k_norm.",
                                      "token_norm.", "out_norm."))
    if is_norm:
        return (1.0 + 0.1 * n).astype(np.float32) if leaf == "weight" else (0.05 * n).astype(np.float32)
    if leaf == "bias":
        b = (0.05 * n).astype(np.float32)
        if name == "cam_head.param_predictor.fc2.bias":
            b[7:9] = 0.3  # fov stays positive through the ReLU: 4 iterations accumulate ~1.2 rad
        if name == "gs_renderer.gs_head.2.bias":
            b[4:7] -= 4.0  # log-scales, reference init -7 (rasterization.py:133-137)
        return b
    # Linear [out, in] / Conv [out, in, kh, kw] / ConvTranspose [in, out, kh, kw]
    if "resize_layers.0." in name or "resize_layers.1." in name:
        fan_in = shape[0]  # transposed conv, k == stride: one tap per output pixel
    else:
        fan_in = int(np.prod(shape[1:]))
    w = n * np.float32(1.0 / np.sqrt(fan_in))
    if name.endswith("param_predictor.fc2.weight"):
        w *= np.float32(0.1)
    if name.endswith("scratch.output_conv2.2.weight") or name == "gs_renderer.gs_head.2.weight":
        w *= np.float32(0.3)
    return w.astype(np.float32)


def iter_params(cfg: WMConfig, seed: int = 0, preset: str = "sensitive") -> Iterator[Tuple[str, np.ndarray]]:
    for name, shape in param_spec(cfg).items():
        yield name, make_param(name, shape, seed, preset)


def make_state_dict(cfg: WMConfig, seed: int = 0, preset: str = "sensitive") -> Dict[str, np.ndarray]:
    return dict(iter_params(cfg, seed, preset))
