"""Image ingest: drop-in for the reference's ``load_and_preprocess_images``
(src/utils/inference_utils.py:14-149).  Decoding, alpha compositing and RGB conversion stay with Pillow on the host, as
in the reference; everything after the decode — Pillow-exact bicubic resize, /255, centre crop / white padding — runs
in libwm_hip.so on the uint8 image, so 3 bytes per pixel cross PCIe instead of 12 and no fp32 image is touched on the
CPU."""
from __future__ import annotations

import ctypes as C
from typing import List

import torch

from . import _lib


def preprocess_rgb(img_u8: torch.Tensor, preprocessing_mode: str = "crop", output_size: int = 518) -> torch.Tensor:
    """One decoded image, uint8 [H, W, 3] on the GPU -> float32 [3, H', W'] (inference_utils.py:67-108)."""
    if preprocessing_mode not in ("crop", "pad"):
        raise ValueError("preprocessing_mode must be either 'crop' or 'pad'")  # inference_utils.py:46-47
    if img_u8.device.type != "cuda":
        raise RuntimeError("preprocess_rgb runs in libwm_hip.so: the image must be on the GPU")
    if img_u8.dtype != torch.uint8 or img_u8.dim() != 3 or img_u8.shape[2] != 3:
        raise ValueError("expected a uint8 [H, W, 3] RGB image")
    img_u8 = img_u8.contiguous()
    H, W = int(img_u8.shape[0]), int(img_u8.shape[1])
    mode = 1 if preprocessing_mode == "pad" else 0
    L = _lib.lib()
    oh, ow = C.c_int32(), C.c_int32()
    if L.wm_preprocess_image_size(H, W, mode, output_size, C.byref(oh), C.byref(ow)) != 0:
        raise ValueError(f"image of {W}x{H} cannot be brought to {output_size}")
    out = torch.empty(3, oh.value, ow.value, device=img_u8.device)
    wsb = L.wm_preprocess_image_workspace_bytes(H, W, mode, output_size)
    ws = torch.empty(wsb, device=img_u8.device, dtype=torch.uint8)
    s = C.c_void_p(torch.cuda.current_stream(img_u8.device).cuda_stream)
    if L.wm_preprocess_image(C.c_void_p(img_u8.data_ptr()), H, W, mode, output_size, C.c_void_p(out.data_ptr()),
                             C.c_void_p(ws.data_ptr()), wsb, s) != 0:
        raise RuntimeError("wm_preprocess_image failed")
    return out


def load_and_preprocess_images(image_file_paths: List[str], preprocessing_mode: str = "crop", output_size: int = 518,
                               device: str = "cuda:0") -> torch.Tensor:
    """Same arguments and result layout as the reference ([1, N, 3, H, W] in [0, 1]); the tensor lives on ``device``."""
    if len(image_file_paths) == 0:
        raise ValueError("At least 1 image is required")  # inference_utils.py:43-44
    if preprocessing_mode not in ["crop", "pad"]:
        raise ValueError("preprocessing_mode must be either 'crop' or 'pad'")
    import numpy as np
    from PIL import Image
    tensors, shapes = [], set()
    for path in image_file_paths:
        im = Image.open(path)
        if im.mode == "RGBA":  # inference_utils.py:58-63
            white = Image.new("RGBA", im.size, (255, 255, 255, 255))
            im = Image.alpha_composite(white, im)
        im = im.convert("RGB")
        u8 = torch.from_numpy(np.asarray(im).copy()).to(device)
        t = preprocess_rgb(u8, preprocessing_mode, output_size)
        shapes.add((t.shape[1], t.shape[2]))
        tensors.append(t)
    if len(shapes) > 1:  # inference_utils.py:113-134: centre every image in the largest frame, white border
        print(f"Warning: Found images with different shapes: {shapes}")
        mh, mw = max(s[0] for s in shapes), max(s[1] for s in shapes)
        padded = []
        for t in tensors:
            ph, pw = mh - t.shape[1], mw - t.shape[2]
            if ph > 0 or pw > 0:
                top, left = ph // 2, pw // 2
                t = torch.nn.functional.pad(t, (left, pw - left, top, ph - top), mode="constant", value=1.0)
            padded.append(t)
        tensors = padded
    return torch.stack(tensors).unsqueeze(0)
