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


def make_param(name: str, shape: Tuple[int, ...], seed: int = 0) -> np.ndarray:
    """fp32 ndarray for one reference parameter name."""
    rng = _rng(name, seed)
    leaf = name.rsplit(".", 1)[-1]
    n = rng.standard_normal(shape, dtype=np.float32)
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


def iter_params(cfg: WMConfig, seed: int = 0) -> Iterator[Tuple[str, np.ndarray]]:
    for name, shape in param_spec(cfg).items():
        yield name, make_param(name, shape, seed)


def make_state_dict(cfg: WMConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    return dict(iter_params(cfg, seed))
