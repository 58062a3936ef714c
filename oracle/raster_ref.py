"""TEST INFRASTRUCTURE ONLY (imported by tests/ and gen scripts, never by the product path).

CPU restatement (numpy, fp32) of the 3D-Gaussian-splat rasteriser forward the reference calls through
`Rasterizer.rasterize_splats` (src/models/models/rasterization.py:29-66): gsplat.rasterization(...,
packed=True, rasterize_mode="classic", camera_model="pinhole", render_mode="RGB+ED", sh_degree=0).
gsplat is the un-built submodule at submodules/gsplat (version in submodules/gsplat/gsplat/version.py);
its CUDA extension cannot be built here (no nvcc), so the stages are pinned as follows:

  * quat/scale -> covariance, world -> camera, perspective EWA projection, radii, conics, SH(0) colours and the
    tile-intersection lists follow gsplat/cuda/_torch_impl.py (_quat_scale_to_covar_preci :45-75, _world_to_cam
    :250-283, _persp_proj :78-133, _fully_fused_projection :286-375, _isect_tiles :378-474,
    _isect_offset_encode :477-503, _eval_sh_bases_fast :720) and gsplat/rendering.py:853-923 — these ARE pinned:
    oracle/gen_golden_raster.py runs those reference functions on CPU and tests/golden/raster_*.npz holds their outputs.
  * alpha compositing follows gsplat/cuda/csrc/RasterizeToPixels3DGSFwd.cu:118-184 and the expected-depth
    normalisation rendering.py:984-992.  PARITY UNPINNED for this stage: the reference's torch path for it
    (_rasterize_to_pixels, _torch_impl.py:607) needs the CUDA op rasterize_to_indices_in_range and nerfacc, neither
    available; it is restated from the CUDA source and cross-checked by a second, tiling-free formulation
    (`composite_bruteforce`: every pixel sorts ALL Gaussians itself).
"""
from __future__ import annotations

import math

import numpy as np

SH_C0 = 0.28209479177387814
ALPHA_THRESHOLD = 1.0 / 255.0
TILE = 16
F = np.float32


def quat_scale_to_covar(quats, scales):
    """_torch_impl.py:11-29,45-61 (quats wxyz, normalised here)."""
    q = quats.astype(F)
    q = q / np.maximum(np.linalg.norm(q, axis=-1, keepdims=True), F(1e-12))
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                  2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                  2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3).astype(F)
    M = R * scales.astype(F)[:, None, :]
    return np.einsum("nij,nkj->nik", M, M).astype(F)


def project(means, quats, scales, viewmats, Ks, width, height, eps2d=0.3, near_plane=0.01, far_plane=1e10):
    """-> radii [C,N,2] int32, means2d [C,N,2], depths [C,N], conics [C,N,3]   (_torch_impl.py:78-133,250-375)."""
    means = means.astype(F); viewmats = viewmats.astype(F); Ks = Ks.astype(F)
    covars = quat_scale_to_covar(quats, scales)
    R, t = viewmats[:, :3, :3], viewmats[:, :3, 3]
    mc = np.einsum("cij,nj->cni", R, means) + t[:, None, :]
    cc = np.einsum("cij,njk,clk->cnil", R, covars, R).astype(F)
    tx, ty, tz = mc[..., 0], mc[..., 1], mc[..., 2]
    tz2 = tz * tz
    fx, fy, cx, cy = Ks[:, 0, 0, None], Ks[:, 1, 1, None], Ks[:, 0, 2, None], Ks[:, 1, 2, None]
    tfx, tfy = F(0.5) * F(width) / fx, F(0.5) * F(height) / fy
    lxp, lxn = (F(width) - cx) / fx + F(0.3) * tfx, cx / fx + F(0.3) * tfx
    lyp, lyn = (F(height) - cy) / fy + F(0.3) * tfy, cy / fy + F(0.3) * tfy
    with np.errstate(divide="ignore", invalid="ignore"):
        txc = tz * np.clip(tx / tz, -lxn, lxp)
        tyc = tz * np.clip(ty / tz, -lyn, lyp)
        O = np.zeros_like(tz)
        J = np.stack([fx / tz, O, -fx * txc / tz2, O, fy / tz, -fy * tyc / tz2], -1).reshape(tz.shape + (2, 3)).astype(F)
        cov2d = np.einsum("cnij,cnjk,cnlk->cnil", J, cc, J).astype(F)
        m2 = np.einsum("cij,cnj->cni", Ks[:, :2, :3], mc) / tz[..., None]
    cov2d = cov2d + np.eye(2, dtype=F) * F(eps2d)
    det = cov2d[..., 0, 0] * cov2d[..., 1, 1] - cov2d[..., 0, 1] * cov2d[..., 1, 0]
    det = np.maximum(det, F(1e-10))
    conics = np.stack([cov2d[..., 1, 1] / det, -(cov2d[..., 0, 1] + cov2d[..., 1, 0]) / F(2.0) / det, cov2d[..., 0, 0] / det], -1).astype(F)
    rx = np.ceil(F(3.33) * np.sqrt(cov2d[..., 0, 0]))
    ry = np.ceil(F(3.33) * np.sqrt(cov2d[..., 1, 1]))
    radius = np.stack([rx, ry], -1)
    valid = (det > 0) & (tz > near_plane) & (tz < far_plane)
    radius[~valid] = 0.0
    inside = (m2[..., 0] + radius[..., 0] > 0) & (m2[..., 0] - radius[..., 0] < width) & \
             (m2[..., 1] + radius[..., 1] > 0) & (m2[..., 1] - radius[..., 1] < height)
    radius[~inside] = 0.0
    radius = np.nan_to_num(radius, nan=0.0, posinf=0.0, neginf=0.0)
    return radius.astype(np.int32), m2.astype(F), tz.astype(F), conics


def sh0_colors(sh_dc):
    """rendering.py:919-923 with sh_degree = 0: colour = clamp_min(C0 * sh[:, 0] + 0.5, 0)  (no view dependence)."""
    return np.maximum(F(SH_C0) * sh_dc.astype(F) + F(0.5), F(0.0)).astype(F)


def tile_rects(means2d, radii, width, height):
    """Per (camera, Gaussian) tile rectangle [x0, y0, x1, y1) and count (_torch_impl.py:404-415)."""
    tw, th = math.ceil(width / TILE), math.ceil(height / TILE)
    tm, tr = means2d / F(TILE), radii.astype(F) / F(TILE)
    lo = np.floor(tm - tr).astype(np.int64)
    hi = np.ceil(tm + tr).astype(np.int64)
    lo[..., 0] = np.clip(lo[..., 0], 0, tw); lo[..., 1] = np.clip(lo[..., 1], 0, th)
    hi[..., 0] = np.clip(hi[..., 0], 0, tw); hi[..., 1] = np.clip(hi[..., 1], 0, th)
    cnt = (hi - lo).prod(-1) * (radii > 0).all(-1)
    return lo, hi, cnt.astype(np.int64), tw, th


def isect_tiles(means2d, radii, depths, width, height):
    """Sorted intersection list: keys (((camera << tile_n_bits) | tile) << 32 | depth bits), tile_n_bits =
    bit_length(tiles per image), and flattened (camera * N + g) ids, plus per-(camera, tile) offsets
    (_torch_impl.py:378-503).  Vectorised, same ordering (stable in the emission order)."""
    C, N = depths.shape
    lo, hi, cnt, tw, th = tile_rects(means2d, radii, width, height)
    nbits = int(tw * th).bit_length()
    keys, vals = [], []
    dbits = depths.astype(F).view(np.uint32).astype(np.uint64)
    for c in range(C):
        for g in np.nonzero(cnt[c])[0]:
            ys, xs = np.meshgrid(np.arange(lo[c, g, 1], hi[c, g, 1]), np.arange(lo[c, g, 0], hi[c, g, 0]), indexing="ij")
            tid = (ys * tw + xs).reshape(-1).astype(np.uint64)
            keys.append((((np.uint64(c) << np.uint64(nbits)) | tid) << np.uint64(32)) | dbits[c, g])
            vals.append(np.full(tid.shape, c * N + g, np.int64))
    if not keys:
        return np.zeros(0, np.uint64), np.zeros(0, np.int64), np.zeros(C * tw * th + 1, np.int64), tw, th
    keys, vals = np.concatenate(keys), np.concatenate(vals)
    order = np.argsort(keys, kind="stable")
    keys, vals = keys[order], vals[order]
    tile_keys = ((np.arange(C, dtype=np.uint64)[:, None] << np.uint64(nbits)) | np.arange(tw * th, dtype=np.uint64)[None, :]).reshape(-1)
    offs = np.searchsorted(keys >> np.uint64(32), tile_keys, side="left")
    return keys, vals, np.concatenate([offs, [len(keys)]]).astype(np.int64), tw, th


def composite(means2d, conics, opacities, colors, depths, vals, offs, C, N, width, height, tw, th):
    """Front-to-back alpha blending per 16 x 16 tile (RasterizeToPixels3DGSFwd.cu:118-184), channels = RGB + depth.
    -> rgb [C,H,W,3], expected depth [C,H,W,1] (rendering.py:984-992), alpha [C,H,W,1]."""
    out = np.zeros((C, height, width, 4), F)
    alpha_out = np.zeros((C, height, width, 1), F)
    m2 = means2d.reshape(C * N, 2); cn = conics.reshape(C * N, 3); dp = depths.reshape(C * N)
    for c in range(C):
        for ty in range(th):
            for tx in range(tw):
                t = (c * th + ty) * tw + tx
                ids = vals[offs[t]:offs[t + 1]]
                y0, x0 = ty * TILE, tx * TILE
                y1, x1 = min(y0 + TILE, height), min(x0 + TILE, width)
                py, px = np.meshgrid(np.arange(y0, y1, dtype=F) + F(0.5), np.arange(x0, x1, dtype=F) + F(0.5), indexing="ij")
                T = np.ones(py.shape, F)
                done = np.zeros(py.shape, bool)
                acc = np.zeros(py.shape + (4,), F)
                for g in ids:
                    if done.all():
                        break
                    gi = g % N
                    dx, dy = m2[g, 0] - px, m2[g, 1] - py
                    sigma = F(0.5) * (cn[g, 0] * dx * dx + cn[g, 2] * dy * dy) + cn[g, 1] * dx * dy
                    alpha = np.minimum(F(0.999), opacities[gi] * np.exp(-sigma, dtype=F))
                    use = ~done & ~((sigma < 0) | (alpha < F(ALPHA_THRESHOLD)))
                    nT = T * (F(1.0) - alpha)
                    stop = use & (nT <= F(1e-4))
                    done |= stop
                    use &= ~stop
                    vis = np.where(use, alpha * T, F(0.0)).astype(F)
                    col = np.concatenate([colors[gi], dp[g:g + 1]]).astype(F)
                    acc += vis[..., None] * col
                    T = np.where(use, nT, T).astype(F)
                out[c, y0:y1, x0:x1] = acc
                alpha_out[c, y0:y1, x0:x1, 0] = F(1.0) - T
    ed = out[..., 3:4] / np.maximum(alpha_out, F(1e-10))
    return out[..., :3].copy(), ed.astype(F), alpha_out


def composite_bruteforce(means2d, conics, opacities, colors, depths, radii, width, height):
    """Tiling-free cross-check: every pixel walks ALL Gaussians with radii > 0 of its camera in depth order."""
    C, N = depths.shape
    rgb = np.zeros((C, height, width, 3), F); ed = np.zeros((C, height, width, 1), F); al = np.zeros((C, height, width, 1), F)
    py, px = np.meshgrid(np.arange(height, dtype=F) + F(0.5), np.arange(width, dtype=F) + F(0.5), indexing="ij")
    for c in range(C):
        vis_g = np.nonzero((radii[c] > 0).all(-1))[0]
        order = vis_g[np.argsort(depths[c, vis_g].view(np.uint32), kind="stable")]
        T = np.ones((height, width), F); done = np.zeros((height, width), bool); acc = np.zeros((height, width, 4), F)
        for g in order:
            dx, dy = means2d[c, g, 0] - px, means2d[c, g, 1] - py
            sigma = F(0.5) * (conics[c, g, 0] * dx * dx + conics[c, g, 2] * dy * dy) + conics[c, g, 1] * dx * dy
            alpha = np.minimum(F(0.999), opacities[g] * np.exp(-sigma, dtype=F))
            use = ~done & ~((sigma < 0) | (alpha < F(ALPHA_THRESHOLD)))
            nT = T * (F(1.0) - alpha)
            stop = use & (nT <= F(1e-4))
            done |= stop
            use &= ~stop
            vis = np.where(use, alpha * T, F(0.0)).astype(F)
            acc += vis[..., None] * np.concatenate([colors[g], depths[c, g:g + 1]]).astype(F)
            T = np.where(use, nT, T).astype(F)
        rgb[c] = acc[..., :3]; al[c, ..., 0] = F(1.0) - T
        ed[c] = acc[..., 3:4] / np.maximum(al[c], F(1e-10))
    return rgb, ed, al


def rasterize(means, quats, scales, opacities, sh_dc, viewmats, Ks, width, height):
    """The whole forward: -> (rgb [C,H,W,3], expected depth [C,H,W,1], alpha [C,H,W,1], meta)."""
    radii, m2, depths, conics = project(means, quats, scales, viewmats, Ks, width, height)
    colors = sh0_colors(sh_dc)
    keys, vals, offs, tw, th = isect_tiles(m2, radii, depths, width, height)
    C, N = depths.shape
    rgb, ed, al = composite(m2, conics, opacities.astype(F), colors, depths, vals, offs, C, N, width, height, tw, th)
    return rgb, ed, al, {"radii": radii, "means2d": m2, "depths": depths, "conics": conics, "colors": colors,
                         "isect_keys": keys, "flatten_ids": vals, "offsets": offs}
