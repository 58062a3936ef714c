"""Generate tests/golden/raster_*.npz by running the REFERENCE's own gsplat torch implementation (build container only):
gsplat/cuda/_torch_impl.py functions for covariance, projection, SH colours and tile intersection.  The compositing stage
has no runnable reference here (CUDA op + nerfacc missing) — see oracle/raster_ref.py's header.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_raster.py
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path[:0] = ["/root/reference", "/root/reference/submodules/gsplat"]
from gsplat.cuda._torch_impl import (  # noqa: E402
    _fully_fused_projection, _isect_offset_encode, _isect_tiles, _quat_scale_to_covar_preci, _spherical_harmonics)

GOLD = os.path.join(ROOT, "tests", "golden")


def scene(seed, N, C, W, H):
    rng = np.random.Generator(np.random.Philox(key=[seed, 99]))
    means = np.concatenate([rng.uniform(-1.6, 1.6, (N, 2)), rng.uniform(0.6, 5.0, (N, 1))], 1).astype(np.float32)
    means[: N // 20, 2] = rng.uniform(-1.0, 0.02, N // 20)          # some behind / at the near plane
    quats = rng.standard_normal((N, 4)).astype(np.float32)            # wxyz, not normalised (gsplat normalises)
    scales = np.exp(rng.uniform(-4.5, -1.2, (N, 3))).astype(np.float32)
    scales[: N // 10] *= 6.0                                          # a few large splats covering many tiles
    opac = rng.uniform(0.02, 1.0, N).astype(np.float32)
    sh = rng.uniform(-2.0, 2.0, (N, 1, 3)).astype(np.float32)         # degree-0 SH (colours may clamp at 0)
    viewmats = np.tile(np.eye(4, dtype=np.float32), (C, 1, 1))
    for c in range(C):
        a = 0.15 * c
        viewmats[c, :3, :3] = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]], np.float32)
        viewmats[c, :3, 3] = [0.1 * c, -0.05 * c, 0.2 * c]
    Ks = np.tile(np.array([[0.9 * W, 0, W / 2 + 1.5], [0, 1.1 * W, H / 2 - 0.7], [0, 0, 1]], np.float32), (C, 1, 1))
    return dict(means=means, quats=quats, scales=scales, opacities=opac, sh=sh, viewmats=viewmats, Ks=Ks)


def run(name, seed, N, C, W, H):
    s = scene(seed, N, C, W, H)
    t = {k: torch.from_numpy(v) for k, v in s.items()}
    covars, _ = _quat_scale_to_covar_preci(t["quats"], t["scales"], True, False, triu=False)
    radii, means2d, depths, conics, _ = _fully_fused_projection(t["means"], covars, t["viewmats"], t["Ks"], W, H)
    tw, th = math.ceil(W / 16.0), math.ceil(H / 16.0)
    tiles_per_gauss, isect_ids, flatten_ids = _isect_tiles(means2d, radii, depths, 16, tw, th)
    offsets = _isect_offset_encode(isect_ids, C, tw, th)
    dirs = t["means"][None] - torch.inverse(t["viewmats"])[:, None, :3, 3]
    shs = torch.broadcast_to(t["sh"][None], (C, N, 1, 3))
    colors = torch.clamp_min(_spherical_harmonics(0, dirs, shs) + 0.5, 0.0)
    out = {f"in_{k}": v for k, v in s.items()}
    out.update(width=np.array(W), height=np.array(H), ref_covars=covars.numpy(), ref_radii=radii.numpy(), ref_means2d=means2d.numpy(),
               ref_depths=depths.numpy(), ref_conics=conics.numpy(), ref_colors=colors.numpy(), ref_tiles_per_gauss=tiles_per_gauss.numpy(),
               ref_isect_ids=isect_ids.numpy(), ref_flatten_ids=flatten_ids.numpy(), ref_offsets=offsets.numpy())
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "visible", int((radii > 0).all(-1).sum()), "isects", int(isect_ids.numel()), os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    run("raster_600g_2c_80x56", 7, 600, 2, 80, 56)
    run("raster_1500g_3c_100x70", 8, 1500, 3, 100, 70)
