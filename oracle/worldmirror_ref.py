"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A plain fp32 restatement (torch-CPU functional ops, no nn.Module, no reference import) of the
WorldMirror forward pass of zubair-irshad/HunyuanWorld-Mirror, i.e. SURVEY.md §8(a) rows a1-a15.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Pinning: the reference ships no golden vectors or tests for this path, so this
restatement is pinned against *outputs of the reference itself* run in the build container:
oracle/gen_golden.py imports /root/reference, loads the same name-keyed synthetic weights and
writes tests/golden/*.npz; tests/test_oracle_golden.py checks this file against them.

Every function cites the reference lines it restates (paths relative to the reference root).
``P`` is a dict name -> fp32 torch tensor using the reference's state_dict names.

Rounding emulation (``forward(..., emulate=("bf16", "f16"))``): the same restatement, but every operand of a
contraction that the HIP build feeds to a 16-bit MFMA is rounded to that type first, at exactly the build's rounding
points (DESIGN.md "Numerics"): backbone = LayerNorm output, q / k / v after q-k-norm + RoPE + the folded softmax scale,
the softmax numerator P (base 2, against an INTEGER running max, so the rounding does not depend on the order keys
are visited in), the attention output, GELU(fc1), the patch matrix, all Linear weights; DPT heads = every conv / GEMM
input activation and weight (fp32 activations in between, the refinenet 1x1 ``out_conv`` applied before the bilinear
resize as the build does).  Accumulation, residual stream, LayerNorm, softmax sums, the camera head and the last 1x1
conv of each head stay fp32.  What remains between the HIP build and this mode is fp32 summation order and the
hardware exp2 / rcp — i.e. kernel error, separated from the recipe's rounding error.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

LOG2E = 1.4426950408889634
_EMU = None  # (backbone torch dtype, head torch dtype) while forward(..., emulate=...) runs, else None
_DT = {"bf16": torch.bfloat16, "f16": torch.float16, "fp16": torch.float16}


def _rb(x: Tensor) -> Tensor:
    """round to the backbone's 16-bit operand type (emulation mode only)"""
    return x if _EMU is None else x.to(_EMU[0]).float()


def _rh(x: Tensor) -> Tensor:
    """round to the DPT heads' 16-bit operand type (emulation mode only); f16 saturates at +-65504 as the build's staging does"""
    if _EMU is None:
        return x
    if _EMU[1] == torch.float16:
        x = x.clamp(-65504.0, 65504.0)
    return x.to(_EMU[1]).float()


RESNET_MEAN = (0.485, 0.456, 0.406)   # src/models/models/visual_transformer.py:16
RESNET_STD = (0.229, 0.224, 0.225)    # :17


# ----------------------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------------------
def layer_norm(x: Tensor, w: Optional[Tensor], b: Optional[Tensor], eps: float) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    y = (x - mu) / torch.sqrt(var + eps)
    if w is not None:
        y = y * w + b
    return y


def gelu_erf(x: Tensor) -> Tensor:
    # nn.GELU() default = exact erf form (src/models/layers/mlp.py:17)
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def linear(x: Tensor, P, name: str) -> Tensor:
    return x @ P[name + ".weight"].t() + P[name + ".bias"]


def linear16(x: Tensor, P, name: str, r) -> Tensor:
    """a Linear the build runs on 16-bit MFMA: operands through ``r`` (identity outside emulation), fp32 accumulate + bias"""
    return r(x) @ r(P[name + ".weight"]).t() + P[name + ".bias"]


def rope_tables(max_pos: int, dim: int, base: float) -> Tuple[Tensor, Tensor]:
    """src/models/layers/rope.py:80-111 — dim is the per-axis width (head_dim/2)."""
    expo = torch.arange(0, dim, 2, dtype=torch.float32) / dim
    inv_freq = 1.0 / (base ** expo)
    ang = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv_freq[None, :]
    ang = torch.cat([ang, ang], -1)
    return ang.cos(), ang.sin()


def rope_1d(x: Tensor, pos: Tensor, cos_t: Tensor, sin_t: Tensor) -> Tensor:
    """rope.py:114-146. x [B,H,L,d], pos [B,L] int."""
    c = cos_t[pos][:, None]
    s = sin_t[pos][:, None]
    h = x.shape[-1] // 2
    rot = torch.cat([-x[..., h:], x[..., :h]], -1)
    return x * c + rot * s


def rope_2d(x: Tensor, pos: Tensor, base: float) -> Tensor:
    """rope.py:148-181. x [B,H,L,hd]; pos [B,L,2] (y,x)."""
    half = x.shape[-1] // 2
    cos_t, sin_t = rope_tables(int(pos.max()) + 1, half, base)
    v = rope_1d(x[..., :half], pos[..., 0], cos_t, sin_t)
    h = rope_1d(x[..., half:], pos[..., 1], cos_t, sin_t)
    return torch.cat([v, h], -1)


def attention(x: Tensor, P, p: str, heads: int, pos: Optional[Tensor], qk_norm: bool,
              rope_base: float, kv_gather=None, r=None) -> Tensor:
    """src/models/layers/attention.py:48-69 (softmax(q k^T / sqrt(hd)) v, no mask).
    kv_gather (view-sharded evaluation): maps this rank's K or V [B,H,L,hd] to the
    keys/values of ALL ranks concatenated on L; queries stay local."""
    B, L, C = x.shape
    hd = C // heads
    emu = _EMU is not None and r is not None  # r is None for the fp32 camera trunk
    r = r if r is not None else (lambda t: t)
    qkv = linear16(x, P, p + "qkv", r).reshape(B, L, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    if qk_norm:  # attention.py:41-42,52 — LayerNorm(head_dim), eps 1e-5, affine
        q = layer_norm(q, P[p + "q_norm.weight"], P[p + "q_norm.bias"], 1e-5)
        k = layer_norm(k, P[p + "k_norm.weight"], P[p + "k_norm.bias"], 1e-5)
    if pos is not None:
        q = rope_2d(q, pos, rope_base)
        k = rope_2d(k, pos, rope_base)
    scale = hd ** -0.5
    if emu:  # the build folds scale * log2(e) into q before rounding it and runs the softmax in base 2
        q, k, v = r(q * (scale * LOG2E)), r(k), r(v)
    if kv_gather is not None:
        k, v = kv_gather(k), kv_gather(v)
    o = torch.empty_like(q)
    step = 2048  # row blocks keep the score matrix small; result is identical
    for s in range(0, L, step):
        if emu:
            a = q[:, :, s:s + step] @ k.transpose(-1, -2)
            pm = torch.exp2(a - torch.ceil(a.max(-1, keepdim=True)[0]))   # integer max: P's rounding is order-independent
            o[:, :, s:s + step] = (r(pm) @ v) / pm.sum(-1, keepdim=True)  # fp32 row sums of the un-rounded P
        else:
            a = (q[:, :, s:s + step] * scale) @ k.transpose(-1, -2)
            o[:, :, s:s + step] = torch.softmax(a, -1) @ v
    o = o.transpose(1, 2).reshape(B, L, C)
    return linear16(o, P, p + "proj", r)


def block(x: Tensor, P, p: str, heads: int, eps: float, pos=None, qk_norm=False,
          rope_base: float = 100.0, kv_gather=None, r=None) -> Tensor:
    """src/models/layers/block.py:72-93 eval branch; LayerScale layer_scale.py:16-17.
    r: operand rounding of the 16-bit GEMMs / attention (emulation mode; None = fp32, e.g. the camera trunk)."""
    rr = r if r is not None else (lambda t: t)
    h = layer_norm(x, P[p + "norm1.weight"], P[p + "norm1.bias"], eps)
    x = x + attention(h, P, p + "attn.", heads, pos, qk_norm, rope_base, kv_gather, r) * P[p + "ls1.gamma"]
    h = layer_norm(x, P[p + "norm2.weight"], P[p + "norm2.bias"], eps)
    h = linear16(gelu_erf(linear16(h, P, p + "mlp.fc1", rr)), P, p + "mlp.fc2", rr)
    return x + h * P[p + "ls2.gamma"]


# ----------------------------------------------------------------------------------------------
# a4: DINOv2 encoder
# ----------------------------------------------------------------------------------------------
def dino_pos_embed(P, p: str, gh: int, gw: int) -> Tensor:
    """src/models/layers/vision_transformer.py:175-207 with interpolate_offset=0.0,
    antialias=True (visual_transformer.py:117-120,152-160). Returns [1, 1+gh*gw, D]."""
    pe = P[p + "pos_embed"]
    n = pe.shape[1] - 1
    m = int(math.isqrt(n))
    if gh * gw == n and gh == gw:
        return pe
    D = pe.shape[-1]
    grid = pe[:, 1:].reshape(1, m, m, D).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(gh, gw), mode="bicubic", antialias=True)
    grid = grid.permute(0, 2, 3, 1).reshape(1, gh * gw, D)
    return torch.cat([pe[:, :1], grid], 1)


def dino_encode(P, p: str, img: Tensor, cfg) -> Tensor:
    """vision_transformer.py:209-221,247-266 -> x_norm_patchtokens [N, gh*gw, D]."""
    N, _, H, W = img.shape
    ps = cfg.patch_size
    assert H % ps == 0 and W % ps == 0  # patch_embed.py:67-68
    gh, gw = H // ps, W // ps
    x = F.conv2d(_rb(img), _rb(P[p + "patch_embed.proj.weight"]), P[p + "patch_embed.proj.bias"], stride=ps)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([P[p + "cls_token"].expand(N, -1, -1), x], 1)
    x = x + dino_pos_embed(P, p, gh, gw)
    x = torch.cat([x[:, :1], P[p + "register_tokens"].expand(N, -1, -1), x[:, 1:]], 1)
    for i in range(cfg.dino_depth):
        x = block(x, P, p + f"blocks.{i}.", cfg.dino_heads, 1e-6, r=_rb)
    x = layer_norm(x, P[p + "norm.weight"], P[p + "norm.bias"], 1e-6)
    return x[:, 1 + cfg.num_register_tokens:]


# ----------------------------------------------------------------------------------------------
# a2: priors
# ----------------------------------------------------------------------------------------------
def rotmat_to_quat_xyzw(R: Tensor) -> Tensor:
    """src/models/utils/rotation.py:41-97,114-126 (best-conditioned candidate, w >= 0)."""
    m = R.reshape(R.shape[:-2] + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.unbind(-1)
    t = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22,
                     1 - m00 + m11 - m22, 1 - m00 - m11 + m22], -1)
    q_abs = torch.where(t > 0, torch.sqrt(t.clamp(min=0)), torch.zeros_like(t))
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], -1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], -1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], -1)], -2)
    cand = cand / (2.0 * q_abs[..., None].clamp(min=0.1))
    idx = q_abs.argmax(-1)
    out = torch.gather(cand, -2, idx[..., None, None].expand(idx.shape + (1, 4))).squeeze(-2)
    out = out[..., [1, 2, 3, 0]]
    return torch.where(out[..., 3:4] < 0, -out, out)


def normalize_poses(ext: Tensor, padding: float = 0.1) -> Tensor:
    """src/models/utils/priors.py:4-105. ext [B,S,3,4] c2w."""
    ext = torch.nan_to_num(ext.clone(), nan=0.0, posinf=1e6, neginf=-1e6)
    out = ext.clone()
    for b in range(ext.shape[0]):
        pos = ext[b, :, :3, 3]
        if pos.shape[0] > 10:
            lo = torch.quantile(pos, 0.05, dim=0)
            hi = torch.quantile(pos, 0.95, dim=0)
        else:
            lo = pos.min(0)[0]
            hi = pos.max(0)[0]
        rng = torch.maximum(hi - lo, torch.maximum(torch.tensor(1e-6), hi.abs() * 1e-6))
        scale = rng.max().clamp(1e-6, 1e6)
        centre = (lo + hi) / 2
        out[b, :, :3, 3] = ((pos - centre) / (scale / (1 - 2 * padding)) + 0.5).clamp(0, 1)
    return out


def normalize_depth(depth: Tensor, eps: float = 1e-6) -> Tensor:
    """priors.py:108-167. depth [B,S,H,W]."""
    B, S, H, W = depth.shape
    d = torch.nan_to_num(depth.reshape(B * S, H, W), nan=0.0, posinf=1e6, neginf=0.0)
    outs = []
    for i in range(B * S):
        flat = d[i].flatten()
        use = flat[flat > 0] if (flat > 0).any() else flat
        if use.numel() > 100:
            lo, hi = torch.quantile(use, 0.01), torch.quantile(use, 0.99)
        else:
            lo, hi = use.min(), use.max()
        if hi == lo:
            hi = lo + 1.0
        e = max(eps, float((hi - lo).abs()) * eps)
        outs.append(((d[i] - lo) / (hi - lo + e)).clamp(0, 1))
    return torch.stack(outs).reshape(B, S, H, W)


def extract_priors(views: Dict[str, Tensor]):
    """src/models/models/worldmirror.py:218-251 -> (depths, rays, poses)."""
    h, w = views["img"].shape[-2:]
    depths = rays = poses = None
    if "camera_pose" in views:
        ext = normalize_poses(views["camera_pose"][:, :, :3].float())
        q = rotmat_to_quat_xyzw(ext[..., :3, :3])
        poses = torch.cat([ext[..., :3, 3], q], -1).float()   # camera_utils.py:25-35
    if "depthmap" in views:
        depths = normalize_depth(views["depthmap"].float())
    if "camera_intrinsics" in views:
        K = views["camera_intrinsics"][:, :, :3, :3].float()
        rays = torch.stack([K[..., 0, 0] / w, K[..., 1, 1] / h, K[..., 0, 2] / w, K[..., 1, 2] / h], -1)
    return depths, rays, poses


# ----------------------------------------------------------------------------------------------
# a3-a10: backbone
# ----------------------------------------------------------------------------------------------
def special_tokens(tok: Tensor, S: int, first_view: int = 0) -> Tensor:
    """visual_transformer.py:397-416: slot 0 -> (global) view 0, slot 1 -> views 1.. ; [S,X,D] (B=1)."""
    idx = [0 if first_view + i == 0 else 1 for i in range(S)]
    return tok[0, idx]


def backbone(P, img: Tensor, cfg, priors=None, cond_flags=(0, 0, 0),
             collect: Optional[dict] = None, shard=None) -> Tuple[List[Tensor], int]:
    """visual_transformer.py:250-341. img [1,S,3,H,W] in [0,1] -> 4 x [1,S,P,2D].
    shard = (first_view, kv_gather): img/priors hold only this rank's views."""
    first_view, kv_gather = shard if shard is not None else (0, None)
    v = "visual_geometry_transformer."
    B, S, C, H, W = img.shape
    assert B == 1
    if C != 3:
        raise ValueError(f"Expected 3 input channels, got {C}")
    mean = torch.tensor(RESNET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(RESNET_STD).view(1, 3, 1, 1)
    x = (img[0] - mean) / std
    patches = dino_encode(P, v + "patch_embed.", x, cfg)
    if collect is not None:
        collect["dino"] = patches
    D = patches.shape[-1]
    cam = special_tokens(P[v + "cam_token"], S, first_view)
    reg = special_tokens(P[v + "reg_token"], S, first_view)
    gh, gw = H // cfg.patch_size, W // cfg.patch_size
    if cfg.enable_cond:  # :343-371
        depths, rays, poses = priors if priors is not None else (None, None, None)
        zero = torch.zeros(S, 1, D)
        if cond_flags[0] == 1 and poses is not None:
            h = F.silu(linear(poses.reshape(S, 7), P, v + "pose_embed.0"))
            pose_t = linear(h, P, v + "pose_embed.2")[:, None]
        else:
            pose_t = zero
        if cond_flags[1] == 1 and depths is not None:
            ps = cfg.patch_size
            d = F.pixel_unshuffle(depths.reshape(S, 1, H, W), ps).permute(0, 2, 3, 1)  # patch_embed.py:79-93
            d = linear16(gelu_erf(linear16(d, P, v + "depth_embed.proj.2.fc1", _rb)), P, v + "depth_embed.proj.2.fc2", _rb)
            patches = patches + d.reshape(S, gh * gw, D)
        if cond_flags[2] == 1 and rays is not None:
            h = F.silu(linear(rays.reshape(S, 4), P, v + "ray_embed.0"))
            ray_t = linear(h, P, v + "ray_embed.2")[:, None]
        else:
            ray_t = zero
        tok = torch.cat([cam, reg, pose_t, ray_t, patches], 1)
    else:
        tok = torch.cat([cam, reg, patches], 1)
    psi = cfg.patch_start_idx
    Pn = tok.shape[1]
    yy, xx = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")  # rope.py:33-53
    pos = torch.stack([yy.flatten(), xx.flatten()], -1) + 1                      # :302-303
    pos = torch.cat([torch.zeros(psi, 2, dtype=pos.dtype), pos], 0)              # :304-306
    pos_f = pos[None].expand(S, -1, -1)
    pos_g = pos_f.reshape(1, S * Pn, 2)
    if collect is not None:
        collect["tokens0"] = tok
    taps = []
    for i in range(cfg.depth):  # :309-339
        tok = block(tok.reshape(S, Pn, D), P, v + f"frame_blocks.{i}.", cfg.num_heads, 1e-5,
                    pos_f, True, cfg.rope_freq, r=_rb)
        frame_out = tok
        tok = block(tok.reshape(1, S * Pn, D), P, v + f"global_blocks.{i}.", cfg.num_heads, 1e-5,
                    pos_g, True, cfg.rope_freq, kv_gather, r=_rb)
        if i in cfg.intermediate_idxs:
            taps.append(torch.cat([frame_out.reshape(1, S, Pn, D), tok.reshape(1, S, Pn, D)], -1))
    return taps, psi


# ----------------------------------------------------------------------------------------------
# a11-a12: camera head
# ----------------------------------------------------------------------------------------------
def camera_head(P, taps: Sequence[Tensor], cfg, tok_gather=None) -> Tensor:
    """src/models/heads/camera_head.py:58-104 -> last iterate [1,S,9].
    tok_gather (sharded evaluation): local camera tokens [1,S_local,2D] -> all views [1,S,2D]."""
    c = "cam_head."
    tok = layer_norm(taps[-1][:, :, 0], P[c + "token_norm.weight"], P[c + "token_norm.bias"], 1e-5)
    if tok_gather is not None:
        tok = tok_gather(tok)
    B, S, D2 = tok.shape
    pred = None
    for _ in range(cfg.cam_steps):
        inp = P[c + "init_token"].expand(B, S, -1) if pred is None else pred
        e = linear(inp, P, c + "param_embed")
        mod = linear(F.silu(e), P, c + "adapt_norm_gen.1")
        shift, scale, gate = mod.chunk(3, -1)
        h = gate * (layer_norm(tok, None, None, 1e-6) * (1 + scale) + shift) + tok
        for i in range(cfg.cam_trunk_depth):
            h = block(h, P, c + f"refine_net.{i}.", cfg.cam_heads, 1e-5)
        h = layer_norm(h, P[c + "out_norm.weight"], P[c + "out_norm.bias"], 1e-5)
        delta = linear(gelu_erf(linear(h, P, c + "param_predictor.fc1")), P, c + "param_predictor.fc2")
        pred = delta if pred is None else pred + delta
    return torch.cat([pred[..., :7], F.relu(pred[..., 7:])], -1)  # :116-147 (t,quat linear; fov relu)


def quat_to_rotmat(q: Tensor) -> Tensor:
    """src/models/utils/rotation.py:8-38 (xyzw, two_s = 2/|q|^2)."""
    i, j, k, r = q.unbind(-1)
    s = 2.0 / (q * q).sum(-1)
    o = torch.stack([1 - s * (j * j + k * k), s * (i * j - k * r), s * (i * k + j * r),
                     s * (i * j + k * r), 1 - s * (i * i + k * k), s * (j * k - i * r),
                     s * (i * k - j * r), s * (j * k + i * r), 1 - s * (i * i + j * j)], -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def camera_matrices(vec: Tensor, H: int, W: int) -> Tuple[Tensor, Tensor]:
    """camera_utils.py:46-75 + worldmirror.py:165-175 -> (c2w [..,4,4], K [..,3,3])."""
    R = quat_to_rotmat(vec[..., 3:7])
    ext = torch.cat([R, vec[..., 0:3, None]], -1)
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0]).expand(ext.shape[:-2] + (1, 4))
    c2w = torch.linalg.inv(torch.cat([ext, bottom], -2))
    K = torch.zeros(vec.shape[:-1] + (3, 3))
    K[..., 1, 1] = H * 0.5 / torch.tan(vec[..., 7] * 0.5)
    K[..., 0, 0] = W * 0.5 / torch.tan(vec[..., 8] * 0.5)
    K[..., 0, 2] = W * 0.5
    K[..., 1, 2] = H * 0.5
    K[..., 2, 2] = 1.0
    return c2w, K


# ----------------------------------------------------------------------------------------------
# a13: DPT head
# ----------------------------------------------------------------------------------------------
def uv_pos_embed(h: int, w: int, C: int, aspect: float) -> Tensor:
    """src/models/utils/grid.py:4-90 via dense_head.py:253-263 -> [C,h,w] fp32 (x 0.1 applied by caller)."""
    diag = (aspect ** 2 + 1.0) ** 0.5
    sx, sy = aspect / diag, 1.0 / diag
    u = torch.linspace(-sx * (w - 1) / w, sx * (w - 1) / w, w, dtype=torch.float32)
    v = torch.linspace(-sy * (h - 1) / h, sy * (h - 1) / h, h, dtype=torch.float32)
    uu, vv = torch.meshgrid(u, v, indexing="xy")          # [h,w]
    om = torch.arange(C // 4, dtype=torch.float64) / (C / 4.0)
    om = 1.0 / (100.0 ** om)
    ox = uu.reshape(-1).double()[:, None] * om[None]
    oy = vv.reshape(-1).double()[:, None] * om[None]
    emb = torch.cat([ox.sin(), ox.cos(), oy.sin(), oy.cos()], 1).float()
    return emb.reshape(h, w, C).permute(2, 0, 1)


def rcu(x: Tensor, P, p: str) -> Tensor:
    """dense_head.py:435-455 with nn.ReLU(inplace=True): the skip is relu(x)."""
    r = F.relu(x)
    y = F.conv2d(_rh(r), _rh(P[p + "conv1.weight"]), P[p + "conv1.bias"], padding=1)
    y = F.conv2d(_rh(F.relu(y)), _rh(P[p + "conv2.weight"]), P[p + "conv2.bias"], padding=1)
    return y + r


def fusion(P, p: str, x: Tensor, skip: Optional[Tensor], size) -> Tensor:
    """dense_head.py:509-538."""
    if skip is not None:
        x = x + rcu(skip, P, p + "resConfUnit1.")
    x = rcu(x, P, p + "resConfUnit2.")
    if size is None:
        size = (x.shape[-2] * 2, x.shape[-1] * 2)
    if _EMU is not None:  # the build applies the 1x1 at the low resolution (both maps are linear, the weights sum to 1)
        x = F.conv2d(_rh(x), _rh(P[p + "out_conv.weight"]), P[p + "out_conv.bias"])
        return F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=True)
    x = F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=True)
    return F.conv2d(x, P[p + "out_conv.weight"], P[p + "out_conv.bias"])


def activate(out: Tensor, act: str) -> Tuple[Tensor, Tensor]:
    """dense_head.py:297-344: out [n,C,H,W] -> (attr [n,H,W,C-1], conf [n,H,W])."""
    f = out.permute(0, 2, 3, 1)
    a, c = f[..., :-1], f[..., -1]
    if act == "inv_log":
        a = torch.sign(a) * torch.expm1(a.abs())
    elif act == "exp":
        a = a.exp()
    elif act == "norm":
        a = a / a.norm(dim=-1, keepdim=True)
    else:
        raise ValueError(act)
    return a, 1 + c.exp()


def dpt_head(P, p: str, taps: Sequence[Tensor], img: Tensor, psi: int, cfg, act: str,
             is_gs: bool = False, collect: Optional[dict] = None):
    """dense_head.py:107-295 (frame chunking :131-162 is result-neutral)."""
    _, S, _, H, W = img.shape
    ps = cfg.patch_size
    gh, gw = H // ps, W // ps
    aspect = W / H
    feats = []
    for i, t in enumerate(taps):
        x = t[0, :, psi:]
        x = layer_norm(x, P[p + "norm.weight"], P[p + "norm.bias"], 1e-5)
        x = x.permute(0, 2, 1).reshape(S, -1, gh, gw)
        x = F.conv2d(_rh(x), _rh(P[p + f"projects.{i}.weight"]), P[p + f"projects.{i}.bias"])
        x = x + 0.1 * uv_pos_embed(gh, gw, x.shape[1], aspect)
        if i == 0:
            x = F.conv_transpose2d(_rh(x), _rh(P[p + "resize_layers.0.weight"]), P[p + "resize_layers.0.bias"], stride=4)
        elif i == 1:
            x = F.conv_transpose2d(_rh(x), _rh(P[p + "resize_layers.1.weight"]), P[p + "resize_layers.1.bias"], stride=2)
        elif i == 3:
            x = F.conv2d(_rh(x), _rh(P[p + "resize_layers.3.weight"]), P[p + "resize_layers.3.bias"], stride=2, padding=1)
        feats.append(x)
    s = p + "scratch."
    rn = [F.conv2d(_rh(f), _rh(P[s + f"layer{i + 1}_rn.weight"]), None, padding=1) for i, f in enumerate(feats)]
    o = fusion(P, s + "refinenet4.", rn[3], None, rn[2].shape[2:])
    o = fusion(P, s + "refinenet3.", o, rn[2], rn[1].shape[2:])
    o = fusion(P, s + "refinenet2.", o, rn[1], rn[0].shape[2:])
    o = fusion(P, s + "refinenet1.", o, rn[0], None)
    o = F.conv2d(_rh(o), _rh(P[s + "output_conv1.weight"]), P[s + "output_conv1.bias"], padding=1)
    o = F.interpolate(o, size=(gh * ps, gw * ps), mode="bilinear", align_corners=True)
    o = o + 0.1 * uv_pos_embed(o.shape[-2], o.shape[-1], o.shape[1], aspect)
    if collect is not None:
        collect[p + "fused"] = o
    y = F.conv2d(_rh(o), _rh(P[s + "output_conv2.0.weight"]), P[s + "output_conv2.0.bias"], padding=1)
    y = F.conv2d(F.relu(y), P[s + "output_conv2.2.weight"], P[s + "output_conv2.2.bias"])
    attr, conf = activate(y, act)
    attr, conf = attr[None], conf[None]
    if is_gs:  # dense_head.py:232-244
        f = o + F.relu(F.conv2d(_rh(img[0]), _rh(P[p + "input_merger.0.weight"]), P[p + "input_merger.0.bias"], padding=3))
        return f[None], attr, conf
    return attr, conf


# ----------------------------------------------------------------------------------------------
# a15: Gaussian-splat parameters (rasterisation stubbed, as the reference discards its result)
# ----------------------------------------------------------------------------------------------
SH_C0 = 0.28209479177387814  # src/models/utils/sh_utils.py


def gs_splats(P, gs_feat: Tensor, img: Tensor, cam_params: Tensor, gs_depth: Tensor) -> Dict[str, Tensor]:
    """src/models/models/rasterization.py:149-153,389-498 with position_from='gsdepth+predcamera'.
    Returns the pre-prune splats (B=1): means [M,3], quats [M,4], scales [M,3], opacities [M],
    sh [M,1,3], weights [M]."""
    _, S, _, H, W = img.shape
    x = F.conv2d(_rh(gs_feat[0]), _rh(P["gs_renderer.gs_head.0.weight"]), None, padding=1)
    x = F.conv2d(_rh(F.relu(x)), _rh(P["gs_renderer.gs_head.2.weight"]), P["gs_renderer.gs_head.2.bias"])
    g = x.permute(0, 2, 3, 1).reshape(S * H * W, -1)
    quats, scales, opac, rsh, wts = torch.split(g, [4, 3, 1, 3, 1], -1)
    out = {}
    out["quats"] = quats / (quats.norm(dim=-1, keepdim=True) + 1e-8)       # act_gs.py:13-14
    out["scales"] = scales.exp().clamp_max(0.3)                           # act_gs.py:10-11, :431
    out["opacities"] = opac.reshape(-1).sigmoid()
    rgb = img[0].permute(0, 2, 3, 1).reshape(S * H * W, 3)
    out["sh"] = ((rgb - 0.5) / SH_C0 + rsh)[:, None, :]                    # :436-443
    out["weights"] = wts.reshape(-1).sigmoid()
    # means: unproject gs_depth with the predicted camera (:469-484, geometry.py:57-89)
    vec = cam_params.reshape(S, 9)
    R = quat_to_rotmat(vec[:, 3:7])
    t = vec[:, 0:3]
    Rc2w = R.transpose(1, 2)                      # closed-form SE3 inverse
    tc2w = -(Rc2w @ t[:, :, None])[:, :, 0]
    fy = H * 0.5 / torch.tan(vec[:, 7] * 0.5)
    fx = W * 0.5 / torch.tan(vec[:, 8] * 0.5)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    d = gs_depth.reshape(S, H, W)
    xc = (xs[None] - W * 0.5) * d / fx[:, None, None]
    yc = (ys[None] - H * 0.5) * d / fy[:, None, None]
    cam = torch.stack([xc, yc, d], -1)
    world = torch.einsum("shwi,sji->shwj", cam, Rc2w) + tc2w[:, None, None, :]
    out["means"] = world.reshape(S * H * W, 3)
    return out


def prune_gs(sp: Dict[str, Tensor], voxel: float = 0.002) -> Dict[str, Tensor]:
    """rasterization.py:301-387 weighted voxel merge (B=1)."""
    vi = (sp["means"] / voxel).floor().long()
    vi = vi - vi.min(0)[0]
    dims = vi.max(0)[0] + 1
    flat = vi[:, 0] * dims[1] * dims[2] + vi[:, 1] * dims[2] + vi[:, 2]
    uniq, inv = torch.unique(flat, return_inverse=True)
    K = uniq.numel()
    w = sp["weights"]
    wsum = torch.zeros(K).scatter_add_(0, inv, w).clamp(min=1e-8)

    def wavg(x):
        acc = torch.zeros((K,) + x.shape[1:])
        acc.index_add_(0, inv, x * w.reshape((-1,) + (1,) * (x.dim() - 1)))
        return acc
    out = {}
    out["means"] = wavg(sp["means"]) / wsum[:, None]
    out["sh"] = wavg(sp["sh"]) / wsum[:, None, None]
    out["opacities"] = torch.zeros(K).scatter_add_(0, inv, w * w) / wsum
    out["scales"] = wavg(sp["scales"]) / wsum[:, None]
    q = wavg(sp["quats"])
    out["quats"] = q / q.norm(dim=1, keepdim=True).clamp(min=1e-8)
    return out


# ----------------------------------------------------------------------------------------------
# a1, a14: top level
# ----------------------------------------------------------------------------------------------
HEAD_ACT = {"pts_head.": "inv_log", "depth_head.": "exp", "norm_head.": "norm", "gs_head.": "exp"}


def forward(P: Dict[str, Tensor], views: Dict[str, Tensor], cond_flags=(0, 0, 0), cfg=None,
            collect: Optional[dict] = None, prune: bool = True, emulate=None) -> Dict[str, Tensor]:
    """src/models/models/worldmirror.py:120-216.
    emulate = (backbone dtype, head dtype) as "bf16" / "f16": round where the HIP build rounds (module docstring)."""
    global _EMU
    if emulate is not None:
        _EMU = (_DT[emulate[0]], _DT[emulate[1]])
        try:
            return forward(P, views, cond_flags, cfg, collect, prune, None)
        finally:
            _EMU = None
    img = views["img"].float()
    priors = extract_priors(views) if sum(cond_flags) > 0 else None
    taps, psi = backbone(P, img, cfg, priors, cond_flags, collect)
    if collect is not None:
        collect["taps"] = taps
    H, W = img.shape[-2:]
    out: Dict[str, Tensor] = {}
    if cfg.enable_cam:
        cp = camera_head(P, taps, cfg)
        out["camera_params"] = cp
        out["camera_poses"], out["camera_intrs"] = camera_matrices(cp, H, W)
    if cfg.enable_depth:
        a, c = dpt_head(P, "depth_head.", taps, img, psi, cfg, "exp", collect=collect)
        out["depth"], out["depth_conf"] = a, c
    if cfg.enable_pts:
        a, c = dpt_head(P, "pts_head.", taps, img, psi, cfg, "inv_log", collect=collect)
        out["pts3d"], out["pts3d_conf"] = a, c
    if cfg.enable_norm:
        a, c = dpt_head(P, "norm_head.", taps, img, psi, cfg, "norm", collect=collect)
        out["normals"], out["normals_conf"] = a, c
    if cfg.enable_gs:
        f, a, c = dpt_head(P, "gs_head.", taps, img, psi, cfg, "exp", is_gs=True, collect=collect)
        out["gs_depth"], out["gs_depth_conf"] = a, c
        sp = gs_splats(P, f, img, out["camera_params"], a)
        if collect is not None:
            collect["splats_raw"] = sp
        out["splats"] = prune_gs(sp) if prune else sp
    return out
