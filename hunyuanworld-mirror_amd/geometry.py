"""Post-path geometry on the GPU: drop-in for the reference helper the callers run right after
the forward pass (reference: src/models/utils/geometry.py:57-89; infer.py:303, app.py:151)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib


def depth_to_world_coords_points(depth_map: Optional[torch.Tensor], extrinsic: torch.Tensor, intrinsic: torch.Tensor,
                                 eps: float = 1e-8) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor], Optional[torch.Tensor]]:
    """Same signature and return triple as the reference: (world [B,H,W,3], camera [B,H,W,3], mask [B,H,W] bool);
    ``extrinsic`` is camera-to-world.  Tensors must live on a HIP device (there is no CPU path)."""
    if depth_map is None:
        return None, None, None
    if depth_map.device.type != "cuda":
        raise RuntimeError("depth_to_world_coords_points runs in libwm_hip.so: tensors must be on the GPU")
    B, H, W = depth_map.shape
    d = depth_map.contiguous().float()
    e = extrinsic.to(d.device).contiguous().float()
    k = intrinsic.to(d.device).contiguous().float()
    if e.shape != (B, 4, 4) or k.shape != (B, 3, 3):
        raise ValueError(f"extrinsic {tuple(e.shape)} / intrinsic {tuple(k.shape)} do not match depth {tuple(d.shape)}")
    world = torch.empty(B, H, W, 3, device=d.device)
    cam = torch.empty(B, H, W, 3, device=d.device)
    mask = torch.empty(B, H, W, device=d.device, dtype=torch.uint8)
    s = C.c_void_p(torch.cuda.current_stream(d.device).cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    if _lib.lib().wm_depth_to_world(p(d), p(e), p(k), p(world), p(cam), p(mask), B, H, W, C.c_float(eps), s) != 0:
        raise RuntimeError("wm_depth_to_world failed")
    return world, cam, mask.bool()


def create_confidence_mask(confidence: torch.Tensor, conf_threshold_percent: float = 30.0) -> torch.Tensor:
    """Drop-in for infer.py:25-59: flat bool mask keeping the top (100 - p) % confidences (conf <= 1e-5 counts as
    -inf).  Exact radix select on the GPU; ties at the threshold value go to the lowest indices."""
    if confidence.device.type != "cuda":
        raise RuntimeError("create_confidence_mask runs in libwm_hip.so: the tensor must be on the GPU")
    c = confidence.contiguous().float().flatten()
    n = c.numel()
    mask = torch.empty(n, device=c.device, dtype=torch.uint8)
    if n == 0:
        return mask.bool()
    L = _lib.lib()
    wsb = L.wm_confidence_mask_workspace_bytes(n)
    ws = torch.empty(wsb, device=c.device, dtype=torch.uint8)
    s = C.c_void_p(torch.cuda.current_stream(c.device).cuda_stream)
    if L.wm_confidence_mask(C.c_void_p(c.data_ptr()), n, C.c_float(conf_threshold_percent), C.c_void_p(mask.data_ptr()),
                            C.c_void_p(ws.data_ptr()), wsb, s) != 0:
        raise RuntimeError("wm_confidence_mask failed")
    return mask.bool()

