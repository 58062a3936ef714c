"""Host-side mirror of the reference's model API (src/models/models/worldmirror.py:16-251).

``WorldMirror`` keeps the reference's constructor kwargs, ``from_pretrained(dir)``, ``.to()``,
``.eval()`` and ``__call__(views, cond_flags)`` so that the reference's infer.py / app.py only need
their import line changed.  Everything between the input dict and the output dict
runs in libwm_hip.so (hand-written HIP for gfx950) through the C ABI of include/wm_hip.h; torch is
used for device memory, the current stream and the torch.distributed rendezvous only.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from .config import WMConfig, param_spec
from .weights import iter_params


# ------------------------------------------------------------------------------------------------
# a2: prior normalisation (worldmirror.py:218-251, utils/priors.py, utils/rotation.py) — tiny,
# cross-view statistics; computed once on the full view set before sharding.
# ------------------------------------------------------------------------------------------------
def _rotmat_to_quat(R: torch.Tensor) -> torch.Tensor:  # rotation.py:41-97 (xyzw, w >= 0)
    m = R.reshape(R.shape[:-2] + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.unbind(-1)
    t = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], -1)
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


def _normalize_poses(ext: torch.Tensor, padding: float = 0.1) -> torch.Tensor:  # priors.py:4-105
    ext = torch.nan_to_num(ext.clone(), nan=0.0, posinf=1e6, neginf=-1e6)
    out = ext.clone()
    for b in range(ext.shape[0]):
        pos = ext[b, :, :3, 3]
        if pos.shape[0] > 10:
            lo, hi = torch.quantile(pos, 0.05, dim=0), torch.quantile(pos, 0.95, dim=0)
        else:
            lo, hi = pos.min(0)[0], pos.max(0)[0]
        rng = torch.maximum(hi - lo, torch.maximum(torch.full_like(hi, 1e-6), hi.abs() * 1e-6))
        scale = rng.max().clamp(1e-6, 1e6)
        out[b, :, :3, 3] = ((pos - (lo + hi) / 2) / (scale / (1 - 2 * padding)) + 0.5).clamp(0, 1)
    return out


def _normalize_depth(depth: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:  # priors.py:108-167
    B, S, H, W = depth.shape
    d = torch.nan_to_num(depth.reshape(B * S, H, W), nan=0.0, posinf=1e6, neginf=0.0)
    outs = []
    for i in range(B * S):
        flat = d[i].flatten()
        use = flat[flat > 0] if bool((flat > 0).any()) else flat
        if use.numel() > 100:
            lo, hi = torch.quantile(use, 0.01), torch.quantile(use, 0.99)
        else:
            lo, hi = use.min(), use.max()
        if bool(hi == lo):
            hi = lo + 1.0
        e = max(eps, float((hi - lo).abs()) * eps)
        outs.append(((d[i] - lo) / (hi - lo + e)).clamp(0, 1))
    return torch.stack(outs).reshape(B, S, H, W)


def extract_priors(views: Dict[str, torch.Tensor]):
    """(depths [B,S,H,W], rays [B,S,4], poses [B,S,7]) exactly as worldmirror.py:218-251."""
    h, w = views["img"].shape[-2:]
    depths = rays = poses = None
    if "camera_pose" in views:
        ext = _normalize_poses(views["camera_pose"][:, :, :3].float())
        poses = torch.cat([ext[..., :3, 3], _rotmat_to_quat(ext[..., :3, :3])], -1).float()
    if "depthmap" in views:
        depths = _normalize_depth(views["depthmap"].float())
    if "camera_intrinsics" in views:
        K = views["camera_intrinsics"][:, :, :3, :3].float()
        rays = torch.stack([K[..., 0, 0] / w, K[..., 1, 1] / h, K[..., 0, 2] / w, K[..., 1, 2] / h], -1)
    return depths, rays, poses


def shard_inputs(views: Dict[str, torch.Tensor], cond_flags, rank: int, world: int, patch_size: int = 14, batch: int = 0):
    """The host-side sharding rules of the view-sharded forward, device-agnostic: validate the input dict as the
    reference does (visual_transformer.py:272-273, patch_embed.py:67-68), normalise the priors over ALL views
    (worldmirror.py:134-141: cross-view statistics come before sharding), then cut the contiguous block of views
    [rank * S / world, (rank + 1) * S / world) that this rank owns.  Returns a dict with the local tensors
    (img [n,3,H,W], pose [n,7] | None, ray [n,4] | None, depth [n,H,W] | None; fp32, contiguous, on the inputs' device) and
    n, first_view, S, H, W, flags.  Used by WorldMirror.forward and by tests/test_sharding_cpu.py (gloo, world 2)."""
    imgs = views["img"]
    if imgs.dim() != 5:
        raise ValueError("views['img'] must be [B, S, 3, H, W]")
    if imgs.shape[0] != 1:   # one batch element at a time (WorldMirror.forward loops; the reference folds B into B*S, visual_transformer.py:271-277)
        views = {k: (v[batch:batch + 1] if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == imgs.shape[0] else v) for k, v in views.items()}
        imgs = views["img"]
    _, S, ch, H, W = imgs.shape
    if ch != 3:
        raise ValueError(f"Expected 3 input channels, got {ch}")  # visual_transformer.py:272-273
    assert H % patch_size == 0, f"Input image height {H} is not a multiple of patch height {patch_size}"  # patch_embed.py:67-68
    assert W % patch_size == 0, f"Input image width {W} is not a multiple of patch width: {patch_size}"
    flags = [int(x) for x in cond_flags]
    depths = rays = poses = None
    if sum(flags) > 0:  # worldmirror.py:134-141
        depths, rays, poses = extract_priors(views)
    if S % world:
        raise ValueError(f"{S} views do not shard evenly over {world} ranks")
    n = S // world
    v0 = rank * n

    def local(t, shape):
        return None if t is None else t[0, v0:v0 + n].to(torch.float32).reshape(shape).contiguous()
    return {"img": local(imgs, (n, 3, H, W)), "pose": local(poses, (n, 7)), "ray": local(rays, (n, 4)), "depth": local(depths, (n, H, W)),
            "n": n, "first_view": v0, "S": S, "H": H, "W": W, "flags": flags}


def prune_gs(splats: Dict[str, torch.Tensor], voxel_size: float = 0.002) -> Dict[str, List[torch.Tensor]]:
    """Weighted voxel merge of the per-pixel splats (rasterization.py:301-387) through ``wm_prune_gs``
    (splat_prune.hip: voxel keys, stable radix sort, one thread per voxel summing in index order).  Same input dict
    ([B, n, ...] tensors incl. ``weights``) and output dict (lists over B) as the reference method."""
    import ctypes as C
    L = _lib.lib()
    out = {k: [] for k in ("means", "sh", "opacities", "scales", "quats")}
    for i in range(splats["means"].shape[0]):
        t = {k: splats[k][i].detach().to(torch.float32).contiguous() for k in ("means", "quats", "scales", "opacities", "sh", "weights")}
        dev = t["means"].device
        if dev.type != "cuda":
            raise RuntimeError("prune_gs runs in libwm_hip.so on the GPU: move the splats to a HIP device")
        n = int(t["means"].shape[0])
        nsh = int(t["sh"].shape[1]) if t["sh"].dim() == 3 else 1
        if nsh != 1:
            raise NotImplementedError("degree-0 SH only (the reference merges sh[:, 0] and leaves higher bands at zero)")
        o = {"means": torch.empty((n, 3), device=dev), "quats": torch.empty((n, 4), device=dev), "scales": torch.empty((n, 3), device=dev),
             "opacities": torch.empty((n,), device=dev), "sh": torch.empty((n, 1, 3), device=dev)}
        ws = torch.empty(max(int(L.wm_prune_gs_workspace_bytes(n)), 256), device=dev, dtype=torch.uint8)
        K = C.c_int(0)
        p = lambda x: C.c_void_p(x.data_ptr())
        st = L.wm_prune_gs(p(t["means"]), p(t["quats"]), p(t["scales"]), p(t["opacities"]), p(t["sh"]), p(t["weights"]), n, float(voxel_size),
                           p(o["means"]), p(o["quats"]), p(o["scales"]), p(o["opacities"]), p(o["sh"]), C.byref(K), p(ws), ws.numel(),
                           C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if st != 0:
            raise RuntimeError(f"wm_prune_gs failed with status {st}")
        for k in out:
            out[k].append(o[k][:K.value])
    return out


class _GSRenderer:
    """``model.gs_renderer`` as the callers use it (infer.py:264 -> render_interpolated_video, src/utils/render_utils.py:121):
    the object carrying ``.rasterizer`` (rasterization.py:133).  The splat head itself runs inside ``wm_forward``; the
    forward pass does not rasterise (the reference discards that result, rasterization.py:243-246)."""

    def __init__(self):
        from .rasterization import Rasterizer
        self.rasterizer = Rasterizer()
        self.enable_prune, self.voxel_size, self.sh_degree = True, 0.002, 0

    def prune_gs(self, splats, voxel_size: float = 0.002):  # rasterization.py:301 (render_utils.py:206 calls it on the renderer)
        return prune_gs(splats, voxel_size)


class WorldMirror:
    """Drop-in for the reference WorldMirror (ctor kwargs: worldmirror.py:17-34)."""

    def __init__(self, img_size=518, patch_size=14, embed_dim=1024, gs_dim=256, enable_cond=True, enable_cam=True,
                 enable_pts=True, enable_depth=True, enable_norm=True, enable_gs=True,
                 patch_embed="dinov2_vitl14_reg", fixed_patch_embed=False, sampling_strategy="uniform",
                 dpt_gradient_checkpoint=False, condition_strategy=("token", "pow3r", "token"),
                 enable_interpolation=False, max_resolution=2044, *, dtype="bf16", head_dtype="f16",
                 arch: Optional[WMConfig] = None):
        # accepted and ignored, as they are inert at inference in the reference
        del fixed_patch_embed, sampling_strategy, dpt_gradient_checkpoint, enable_interpolation, max_resolution
        if list(condition_strategy) != ["token", "pow3r", "token"]:
            raise NotImplementedError("only condition_strategy=['token','pow3r','token'] exists in the reference")
        if arch is None:
            if patch_embed != "dinov2_vitl14_reg":
                raise NotImplementedError("reference checkpoints use patch_embed='dinov2_vitl14_reg'")
            arch = WMConfig(img_size=img_size, patch_size=patch_size, embed_dim=embed_dim, gs_dim=gs_dim,
                            enable_cond=enable_cond, enable_cam=enable_cam, enable_pts=enable_pts,
                            enable_depth=enable_depth, enable_norm=enable_norm, enable_gs=enable_gs)
        self.cfg = arch
        self.dtype, self.head_dtype = dtype, head_dtype
        self.enable_cam, self.enable_pts = arch.enable_cam, arch.enable_pts
        self.enable_depth, self.enable_norm, self.enable_gs = arch.enable_depth, arch.enable_norm, arch.enable_gs
        self.gs_renderer = _GSRenderer()
        self._host_weights: Dict[str, np.ndarray] = {}
        self._handle = None
        self._device: Optional[torch.device] = None
        self._comm = None  # (rank, world)
        self.missing_weights, self.missing_weight_names = 0, []
        self._reserved = None  # (n_local, n_total, H, W) the library workspace is laid out for
        self._workspace = None  # caller-owned arena (use_workspace)
        self.training = False
        self.return_taps = False
        self.enable_prune = True  # GaussianSplatRenderer(enable_prune=True), worldmirror.py:113

    # ------------------------------------------------------------------ weights
    @classmethod
    def from_pretrained(cls, path: str, **kw):
        """Local directory with config.json + model.safetensors (huggingface_hub mixin layout)."""
        if not os.path.isdir(path):
            raise FileNotFoundError(f"{path}: only local checkpoints are supported (no network)")
        with open(os.path.join(path, "config.json")) as f:
            cfg = json.load(f)
        # PyTorchModelHubMixin passes only the keys the constructor's signature names (worldmirror.py:13,16): metadata the hub adds
        # to config.json ("model_type", library versions, ...) must not reach the constructor
        import inspect
        known = set(inspect.signature(cls.__init__).parameters) - {"self"}
        m = cls(**{**{k: v for k, v in cfg.items() if k in known}, **kw})
        from safetensors import safe_open
        sd = {}
        with safe_open(os.path.join(path, "model.safetensors"), framework="pt", device="cpu") as f:  # "pt": bf16 / f16 checkpoints too
            for k in f.keys():
                sd[k] = f.get_tensor(k)
        missing, _ = m.load_state_dict(sd, strict=False)  # PyTorchModelHubMixin loads non-strictly: missing -> init value, unexpected -> ignored
        if missing:
            import warnings
            warnings.warn(f"{len(missing)} parameter(s) missing from {path}/model.safetensors keep their init values "
                          f"(LayerNorm 1/0, LayerScale 1.0 DINO / 0.01 elsewhere; randomly-initialised tensors of the reference become 0): "
                          + ", ".join(missing[:8]) + (" ..." if len(missing) > 8 else ""))
        return m

    def load_state_dict(self, sd, strict: bool = False):
        spec = param_spec(self.cfg)
        missing = [k for k in spec if k not in sd]
        unexpected = [k for k in sd if k not in spec]
        if strict and (missing or unexpected):
            raise RuntimeError(f"missing {missing[:5]} unexpected {unexpected[:5]}")
        for k, v in sd.items():
            if k in spec:
                a = v.detach().cpu().float().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, np.float32)
                if tuple(a.shape) != tuple(spec[k]):
                    raise RuntimeError(f"shape mismatch for {k}: {a.shape} vs {spec[k]}")
                self._host_weights[k] = np.ascontiguousarray(a)
        if self._handle is not None:
            self._upload()
        return missing, unexpected

    def init_synthetic_weights(self, seed: int = 0, preset: str = "sensitive"):
        """Deterministic name-keyed weights (weights.py) — what parity tests and bench.py use."""
        if self._handle is not None:
            self._upload(iter_params(self.cfg, seed, preset))
        else:
            self._host_weights = dict(iter_params(self.cfg, seed, preset))
        return self

    def share_weights_from(self, other: "WorldMirror"):
        """Use ``other``'s device weights (same config, same device) without copying: several handles — one per stream or
        per in-process rank — on one GPU (wm_share_weights).  ``other`` must outlive this model."""
        if self._handle is None or other._handle is None:
            raise RuntimeError("both models must be on the device (.to('cuda')) first")
        if _lib.lib().wm_share_weights(self._handle, other._handle) != 0:
            raise RuntimeError(f"wm_share_weights: {self._err()}")
        self._weights_owner = other  # keep-alive
        self.missing_weights, self.missing_weight_names = other.missing_weights, list(other.missing_weight_names)
        self._reserved = None
        return self

    def _err(self) -> str:
        return _lib.lib().wm_last_error(self._handle).decode()

    def _upload(self, it=None):
        L = _lib.lib()
        items = it if it is not None else self._host_weights.items()
        for k, a in items:
            a = np.ascontiguousarray(a, dtype=np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            st = L.wm_set_weight(self._handle, k.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim)
            if st != 0:
                raise RuntimeError(f"wm_set_weight({k}): {self._err()}")
        miss = C.c_int(0)
        if L.wm_finalize_weights(self._handle, C.byref(miss)) != 0:
            raise RuntimeError(f"wm_finalize_weights: {self._err()}")
        self.missing_weights = miss.value
        self.missing_weight_names = [L.wm_missing_name(self._handle, i).decode() for i in range(miss.value)]
        self._reserved = None  # weight-derived workspace tables must be rebuilt

    # ------------------------------------------------------------------ nn.Module-like surface
    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("the MI355X build has no CPU path: move the model to a 'cuda' (HIP) device")
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        L = _lib.lib()
        if self._handle is not None:
            L.wm_destroy(self._handle)
        cfg = _lib.make_config(self.cfg, self.dtype, self.head_dtype)
        h = C.c_void_p()
        st = L.wm_create(C.byref(cfg), idx, C.byref(h))
        self._handle = h
        if st != 0:
            raise RuntimeError(f"wm_create: {self._err()}")
        self._device = torch.device("cuda", idx)
        self._reserved, self._workspace = None, None
        if self._host_weights:
            self._upload()
        return self

    def reserve(self, n_local: int, n_total: int, H: int, W: int):
        """Prepare the workspace for a shape (wm_reserve): the forward itself never allocates.  forward() calls it on a
        shape change; call it yourself to keep the first forward of a shape free of allocation too."""
        key = (int(n_local), int(n_total), int(H), int(W))
        if self._reserved != key:
            if _lib.lib().wm_reserve(self._handle, *key) != 0:
                raise RuntimeError(f"wm_reserve{key}: {self._err()}")
            self._reserved = key
        return self

    def workspace_bytes(self, n_local: int, n_total: int, H: int, W: int) -> int:
        return int(_lib.lib().wm_workspace_bytes(self._handle, n_local, n_total, H, W))

    def use_workspace(self, buf: Optional[torch.Tensor]):
        """Caller-owned arena (a uint8 device tensor, kept alive here); None returns ownership to the library."""
        L = _lib.lib()
        if buf is None:
            st = L.wm_set_workspace(self._handle, None, 0)
        else:
            if buf.device != self._device or buf.dtype != torch.uint8 or not buf.is_contiguous():
                raise ValueError("workspace must be a contiguous uint8 tensor on the model's device")
            st = L.wm_set_workspace(self._handle, C.c_void_p(buf.data_ptr()), buf.numel())
        if st != 0:
            raise RuntimeError(f"wm_set_workspace: {self._err()}")
        self._workspace, self._reserved = buf, None
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def eval(self):
        self.training = False
        return self

    def requires_grad_(self, flag=False):
        return self

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().wm_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    # ------------------------------------------------------------------ multi-GPU
    def shard(self, group=None):
        """View-shard across the ranks of a torch.distributed group (one process per GPU): creates
        the handle's own RCCL communicator from an id broadcast through ``group``."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        L = _lib.lib()
        buf = (C.c_uint8 * 128)()
        if rank == 0 and L.wm_rccl_unique_id(buf) != 0:
            raise RuntimeError("wm_rccl_unique_id failed")
        dev = self._device if dist.get_backend(group) == "nccl" else torch.device("cpu")
        t = torch.tensor(list(buf), dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ids = (C.c_uint8 * 128)(*t.cpu().tolist())
        if L.wm_comm_init_rccl(self._handle, ids, rank, world) != 0:
            raise RuntimeError(f"wm_comm_init_rccl: {self._err()}")
        self._comm = (rank, world)
        return self

    def shard_local(self, group_ptr, rank: int, world: int):
        """In-process rank group (tests): see wm_comm_init_local."""
        if _lib.lib().wm_comm_init_local(self._handle, group_ptr, rank) != 0:
            raise RuntimeError(f"wm_comm_init_local: {self._err()}")
        self._comm = (rank, world)
        return self

    # ------------------------------------------------------------------ forward
    def __call__(self, views: Dict[str, torch.Tensor], cond_flags: List[int] = [0, 0, 0]):
        return self.forward(views, cond_flags)

    @torch.no_grad()
    def forward(self, views: Dict[str, torch.Tensor], cond_flags: List[int] = [0, 0, 0]):
        if self._handle is None:
            raise RuntimeError("call .to('cuda') first: the forward pass runs in libwm_hip.so on the GPU")
        B = int(views["img"].shape[0]) if views["img"].dim() == 5 else 1
        if B == 1:
            return self._forward_one(views, cond_flags, 0)
        # B > 1 (visual_transformer.py:271-277 folds the batch into B*S; frame attention is per view and global attention per
        # batch element, so the elements are independent): one forward per element, outputs stacked along dim 0
        outs = [self._forward_one(views, cond_flags, b) for b in range(B)]
        res: Dict[str, torch.Tensor] = {}
        for k, v0 in outs[0].items():
            if isinstance(v0, torch.Tensor):
                res[k] = torch.cat([o[k] for o in outs], 0)
            elif k == "taps":
                res[k] = [torch.cat([o[k][i] for o in outs], 0) for i in range(len(v0))]
            elif k == "splats_raw" or (k == "splats" and isinstance(next(iter(v0.values())), torch.Tensor)):
                res[k] = {kk: torch.cat([o[k][kk] for o in outs], 0) for kk in v0}
            else:   # pruned splats: lists over the batch (rasterization.py:301-387)
                res[k] = {kk: [x for o in outs for x in o[k][kk]] for kk in v0}
        return res

    def _gather_splats(self, raw: Dict[str, torch.Tensor], world: int):
        """All ranks' raw splats, rank-major = global view order (contiguous view blocks per rank): [1, world * M, ...]."""
        L = _lib.lib()
        stream = C.c_void_p(torch.cuda.current_stream(self._device).cuda_stream)
        out = {}
        for k, t in raw.items():
            t = t.contiguous()
            g = torch.empty((world,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            if L.wm_allgather(self._handle, _lib.ptr(t), _lib.ptr(g), t.numel() * t.element_size(), stream) != 0:
                raise RuntimeError(f"wm_allgather: {self._err()}")
            out[k] = g.reshape((1, world * t.shape[1]) + tuple(t.shape[2:]))
        return out

    def _forward_one(self, views: Dict[str, torch.Tensor], cond_flags, batch: int):
        L = _lib.lib()
        dev = self._device
        rank, world = self._comm if self._comm else (0, 1)
        sh = shard_inputs(views, cond_flags, rank, world, self.cfg.patch_size, batch)
        n, v0, S, H, W, flags = sh["n"], sh["first_view"], sh["S"], sh["H"], sh["W"], sh["flags"]
        ps = self.cfg.patch_size
        img_l, pose_l, ray_l, depth_l = [None if sh[k] is None else sh[k].to(dev) for k in ("img", "pose", "ray", "depth")]

        o = _lib.wm_outputs()
        res: Dict[str, torch.Tensor] = {}

        def alloc(key, field, shape):
            t = torch.empty(shape, dtype=torch.float32, device=dev)
            res[key] = t
            setattr(o, field, t.data_ptr())
        if self.cfg.enable_cam:
            alloc("camera_params", "camera_params", (1, S, 9))
            alloc("camera_poses", "camera_poses", (1, S, 4, 4))
            alloc("camera_intrs", "camera_intrs", (1, S, 3, 3))
        if self.cfg.enable_depth:
            alloc("depth", "depth", (1, n, H, W, 1))
            alloc("depth_conf", "depth_conf", (1, n, H, W))
        if self.cfg.enable_pts:
            alloc("pts3d", "pts3d", (1, n, H, W, 3))
            alloc("pts3d_conf", "pts3d_conf", (1, n, H, W))
        if self.cfg.enable_norm:
            alloc("normals", "normals", (1, n, H, W, 3))
            alloc("normals_conf", "normals_conf", (1, n, H, W))
        if self.cfg.enable_gs:
            alloc("gs_depth", "gs_depth", (1, n, H, W, 1))
            alloc("gs_depth_conf", "gs_depth_conf", (1, n, H, W))
            for key, ch in (("means", 3), ("quats", 4), ("scales", 3), ("opacities", 0), ("sh", 3), ("weights", 0)):
                alloc("_splat_" + key, "splat_" + key, (1, n, H, W, ch) if ch else (1, n, H, W))
        taps = None
        if self.return_taps:
            P = self.cfg.patch_start_idx + (H // ps) * (W // ps)
            taps = [torch.empty((1, n, P, 2 * self.cfg.embed_dim), dtype=torch.float32, device=dev) for _ in range(4)]
            for i in range(4):
                o.taps[i] = taps[i].data_ptr()
        fl = (C.c_int32 * 3)(*flags)
        self.reserve(n, S, H, W)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if world == 1:
            st = L.wm_forward(self._handle, _lib.ptr(img_l), n, H, W, _lib.ptr(pose_l), _lib.ptr(depth_l), _lib.ptr(ray_l),
                              fl, C.byref(o), stream)
        else:
            st = L.wm_forward_sharded(self._handle, _lib.ptr(img_l), n, v0, S, H, W, _lib.ptr(pose_l), _lib.ptr(depth_l),
                                      _lib.ptr(ray_l), fl, C.byref(o), stream)
        if st != 0:
            raise RuntimeError(f"wm_forward failed ({st}): {self._err()}")
        if taps is not None:
            res["taps"] = taps
        if self.cfg.enable_gs:
            raw = {k: res.pop("_splat_" + k) for k in ("means", "quats", "scales", "opacities", "sh", "weights")}
            M = n * H * W
            raw = {"means": raw["means"].reshape(1, M, 3), "quats": raw["quats"].reshape(1, M, 4),
                   "scales": raw["scales"].reshape(1, M, 3), "opacities": raw["opacities"].reshape(1, M),
                   "sh": raw["sh"].reshape(1, M, 1, 3), "weights": raw["weights"].reshape(1, M)}
            res["splats_raw"] = raw
            if self.enable_prune:
                # prune_gs merges voxels over ALL views (rasterization.py:301-387): the one cross-view step behind the forward.
                # A sharded forward gathers the ranks' raw splats (global view order) and every rank merges the full set, so
                # preds["splats"] is the reference's on every rank (not world separately merged sets)
                res["splats"] = prune_gs(self._gather_splats(raw, world) if world > 1 else raw)
            else:
                res["splats"] = raw
        self._keepalive = (img_l, pose_l, ray_l, depth_l)
        return res

    # ------------------------------------------------------------------ profiling hooks (bench.py)
    def profile(self, on: bool):
        _lib.lib().wm_profile_enable(self._handle, 1 if on else 0)

    def profile_read(self, kind: int):
        ms, n = C.c_double(0), C.c_int64(0)
        if _lib.lib().wm_profile_read(self._handle, kind, C.byref(ms), C.byref(n)) != 0:
            raise RuntimeError(self._err())
        return ms.value, n.value
