"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

Usage (from the repo root, /root/reference present):
    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [--full]

This is the only file that imports /root/reference.  It never travels as a dependency: tests,
smoke() and bench.py read the committed .npz fixtures (inputs + expected outputs) instead.
Weights are not stored; both sides regenerate them from hunyuanworld_mirror_amd.weights
(name-keyed Philox streams), so a fixture is {seeded inputs, reference outputs}.

Reference construction follows SURVEY.md App. B: the fork's WorldMirror.__init__ passes two
kwargs its VisualGeometryTransformer rejects, so a 2-kwarg shim subclass is installed first.
"""
from __future__ import annotations

import argparse
import os
import sys
from functools import partial

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path[:0] = ["/root/reference", "/root/reference/submodules/gsplat"]

from hunyuanworld_mirror_amd.config import WMConfig, param_spec  # noqa: E402
from hunyuanworld_mirror_amd.weights import make_param  # noqa: E402

import src.models.models.worldmirror as wm  # noqa: E402
from src.models.heads.camera_head import CameraHead  # noqa: E402
from src.models.heads.dense_head import DPTHead  # noqa: E402
from src.models.layers.block import NestedTensorBlock  # noqa: E402
from src.models.layers.attention import MemEffAttention  # noqa: E402
from src.models.layers.vision_transformer import DinoVisionTransformer  # noqa: E402
from src.models.models.rasterization import GaussianSplatRenderer  # noqa: E402


class _VGT(wm.VisualGeometryTransformer):
    def __init__(self, *a, enable_interpolation=False, max_resolution=2044, **k):
        super().__init__(*a, **k)


wm.VisualGeometryTransformer = _VGT
GOLD = os.path.join(ROOT, "tests", "golden")


def build_reference(cfg: WMConfig, preset: str = "sensitive"):
    """Reference model for ``cfg``; the full config uses the stock ctor, scaled-down configs swap
    scaled-down sub-modules into a stock shell (SURVEY §8c 'Scaled-down oracle')."""
    full = cfg.embed_dim == 1024 and cfg.depth == 24
    if full:
        m = wm.WorldMirror(enable_gs=cfg.enable_gs)
    else:
        m = wm.WorldMirror(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=16,
                           patch_embed="conv", enable_gs=False, enable_cam=False, enable_pts=False,
                           enable_depth=False, enable_norm=False)
        vgt = _VGT(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim,
                   depth=cfg.depth, num_heads=cfg.num_heads, patch_embed="conv", enable_cond=cfg.enable_cond,
                   intermediate_idxs=list(cfg.intermediate_idxs))
        vgt.patch_embed = DinoVisionTransformer(
            img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.dino_depth,
            num_heads=cfg.dino_heads, mlp_ratio=4, num_register_tokens=cfg.num_register_tokens, init_values=1.0,
            interpolate_antialias=True, interpolate_offset=0.0, block_chunks=0,
            block_fn=partial(NestedTensorBlock, attn_class=MemEffAttention))
        m.visual_geometry_transformer = vgt
        D2 = 2 * cfg.embed_dim
        oc = list(cfg.dpt_out_channels)
        m.enable_cam = m.enable_pts = m.enable_depth = m.enable_norm = True
        m.cam_head = CameraHead(dim_in=D2, trunk_depth=cfg.cam_trunk_depth, num_heads=cfg.cam_heads)
        m.pts_head = DPTHead(dim_in=D2, output_dim=4, activation="inv_log+expp1", features=cfg.dpt_features, out_channels=oc)
        m.depth_head = DPTHead(dim_in=D2, output_dim=2, activation="exp+expp1", features=cfg.dpt_features, out_channels=oc)
        m.norm_head = DPTHead(dim_in=D2, output_dim=4, activation="norm+expp1", features=cfg.dpt_features, out_channels=oc)
        if cfg.enable_gs:
            m.gs_head = DPTHead(dim_in=D2, output_dim=2, activation="exp+expp1", features=cfg.gs_dim,
                                out_channels=oc, is_gsdpt=True)
            m.gs_renderer = GaussianSplatRenderer(feature_dim=cfg.gs_dim, sh_degree=0, predict_offset=False,
                                                  predict_residual_sh=True, enable_prune=True, voxel_size=0.002,
                                                  using_gtcamera_splat=True, render_novel_views=True)
    m.eval()
    # names/shapes must equal the build's param_spec exactly
    sd = m.state_dict()
    spec = param_spec(cfg)
    ref_keys = {k: tuple(v.shape) for k, v in sd.items()}
    assert ref_keys == dict(spec), (
        sorted(set(ref_keys) ^ set(spec))[:20],
        [(k, ref_keys[k], spec[k]) for k in ref_keys if k in spec and ref_keys[k] != spec[k]][:10])
    with torch.no_grad():
        for k, v in sd.items():
            v.copy_(torch.from_numpy(make_param(k, tuple(v.shape), 0, preset)))
    return m


def make_inputs(seed: int, S: int, H: int, W: int, priors: bool):
    rng = np.random.Generator(np.random.Philox(key=[seed, 77]))
    views = {"img": rng.random((1, S, 3, H, W), dtype=np.float32)}
    if priors:
        pose = np.zeros((1, S, 4, 4), np.float32)
        for i in range(S):
            q = rng.standard_normal(4)
            q /= np.linalg.norm(q)
            x, y, z, w = q
            R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                          [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                          [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
            pose[0, i, :3, :3] = R
            pose[0, i, :3, 3] = rng.standard_normal(3) * 2.0
            pose[0, i, 3, 3] = 1
        views["camera_pose"] = pose
        K = np.zeros((1, S, 3, 3), np.float32)
        K[..., 0, 0] = W * (0.8 + 0.4 * rng.random(S))
        K[..., 1, 1] = H * (0.8 + 0.4 * rng.random(S))
        K[..., 0, 2] = W / 2 + rng.standard_normal(S)
        K[..., 1, 2] = H / 2 + rng.standard_normal(S)
        K[..., 2, 2] = 1
        views["camera_intrinsics"] = K
        d = (0.5 + 4.0 * rng.random((1, S, H, W), dtype=np.float32)).astype(np.float32)
        d[rng.random((1, S, H, W)) < 0.05] = 0.0
        views["depthmap"] = d
    return views


def make_depth_518(seed: int, S: int, H: int, W: int):
    """Depth prior of a benchmark-size fixture, regenerated from its seed like the image (tests/conftest.py draws it the same way):
    depths in [0.5, 4.5), 5 % of the pixels invalid (0)."""
    g = torch.Generator().manual_seed(seed)
    d = 0.5 + 4.0 * torch.rand(1, S, H, W, generator=g)
    d[torch.rand(1, S, H, W, generator=g) < 0.05] = 0.0
    return d.numpy()


def make_inputs_518(seed: int, S: int, H: int, W: int, priors: bool, depth: bool = False):
    """Benchmark-size inputs.  The image is NOT stored in the fixture (25 MB at 8 views): it is regenerated from the
    seed exactly as bench.py draws it (torch.rand on a seeded CPU generator) and checked against a stored fp64 sum.
    Priors (C3 flag set: camera pose + intrinsics) are small and stored."""
    g = torch.Generator().manual_seed(seed)
    views = {"img": torch.rand(1, S, 3, H, W, generator=g).numpy()}
    if priors:
        small = make_inputs(seed, S, 14, 14, True)
        views["camera_pose"] = small["camera_pose"]
        K = small["camera_intrinsics"] * np.float32(1.0)
        K[..., 0, :] *= W / 14.0
        K[..., 1, :] *= H / 14.0
        views["camera_intrinsics"] = K.astype(np.float32)
    if depth:
        views["depthmap"] = make_depth_518(seed + 7, S, H, W)
    return views


def run_case(m, cfg, name, seed, S, H, W, flags, sub=1, keep_taps=True, preset="sensitive", splat_stride=1, regen_img=False):
    if regen_img:
        views_np = make_inputs_518(seed, S, H, W, priors=sum(flags) > 0, depth=bool(flags[1]))
    else:
        views_np = make_inputs(seed, S, H, W, priors=sum(flags) > 0)
    views = {k: torch.from_numpy(v.copy()) for k, v in views_np.items()}
    store = {f"in_{k}": v for k, v in views_np.items() if not (regen_img and k in ("img", "depthmap"))}
    if regen_img:
        spec = {"kind": "torch_rand", "seed": seed, "shape": [1, S, 3, H, W]}
        if "depthmap" in views_np:
            spec["depth_seed"] = seed + 7
            store["sum_in_depthmap"] = np.array(views_np["depthmap"].astype(np.float64).sum())
        store["regen_img"] = np.array(__import__("json").dumps(spec))
        store["sum_in_img"] = np.array(views_np["img"].astype(np.float64).sum())
    store["cond_flags"] = np.array(flags, np.int64)
    store["cfg_json"] = np.array(__import__("json").dumps(cfg.to_dict()))
    store["weights_preset"] = np.array(preset)
    with torch.no_grad():
        if sum(flags) > 0:
            pri = m.extract_priors(views)
            for nm, pr_ in zip(("prior_depths", "prior_rays", "prior_poses"), pri):
                if pr_ is None:
                    continue
                if regen_img and nm == "prior_depths":  # benchmark size: every 8th pixel + the fp64 sum of the whole map
                    store["prior_depths_sub8"] = np.ascontiguousarray(pr_.numpy()[..., ::8, ::8])
                    store["sum_prior_depths"] = np.array(pr_.double().sum().item())
                else:
                    store[nm] = pr_.numpy()
            taps, psi = m.visual_geometry_transformer(views["img"], pri, cond_flags=flags)
        else:
            taps, psi = m.visual_geometry_transformer(views["img"])
        if cfg.enable_gs:
            preds = {}
            cam_seq = m.cam_head(taps)
            preds["camera_params"] = cam_seq[-1]
            f, d, c = m.gs_head(taps, images=views["img"], patch_start_idx=psi)
            preds["gs_depth"], preds["gs_depth_conf"] = d, c
            from einops import rearrange
            gp = m.gs_renderer.gs_head(rearrange(f, "b s c h w -> (b s) c h w"))
            sp = m.gs_renderer.prepare_splats(views, preds, views["img"], gp, S, 0)
            if splat_stride == 1:
                store["gs_feat"] = f.numpy()
            for k in ("means", "quats", "scales", "opacities", "sh", "weights"):
                store[f"splats_raw_{k}"] = np.ascontiguousarray(sp[k][0].numpy()[::splat_stride])  # per-pixel splats, every splat_stride-th
                store[f"sum_splats_raw_{k}"] = np.array(sp[k][0].double().sum().item())
            store["splat_stride"] = np.array(splat_stride)
            if splat_stride == 1:  # the voxel-merged set is only kept for the small fixtures (it is as large as the per-pixel one)
                pr = m.gs_renderer.prune_gs(sp)
                for k in ("means", "quats", "scales", "opacities", "sh"):
                    store[f"splats_{k}"] = pr[k][0].numpy()
            preds_np = {k: v.numpy() for k, v in preds.items()}
        else:
            preds = m._gen_all_preds(taps, views["img"], psi, views)
            preds_np = {k: v.numpy() for k, v in preds.items()}
    assert psi == cfg.patch_start_idx
    if keep_taps:
        for i, t in enumerate(taps):
            store[f"tap{i}"] = t.numpy()
    for k, v in preds_np.items():
        store[f"sum_{k}"] = np.array(np.nan_to_num(v.astype(np.float64), posinf=0, neginf=0).sum())
        if sub > 1 and v.ndim >= 4 and v.shape[2] == H:
            v = v[:, :, ::sub, ::sub]
        store[f"out_{k}"] = np.ascontiguousarray(v)
    store["subsample"] = np.array(sub)
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **store)
    print(name, {k: (v.shape, float(np.abs(v).mean())) for k, v in preds_np.items() if k != "splats"},
          os.path.getsize(path) // 1024, "KiB")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also run the full-size 2x224 case (config C1)")
    ap.add_argument("--only-full", action="store_true")
    ap.add_argument("--refinit", action="store_true", help="goldens with the reference's own init statistics")
    ap.add_argument("--full-priors", action="store_true", help="full architecture, 2 x 224^2, pose + intrinsics priors (the C3 flag set)")
    ap.add_argument("--full-nonsquare", action="store_true", help="full architecture, 3 views of 154 x 210 (non-square: pos-embed resample), pose + depth + intrinsics priors")
    ap.add_argument("--full-518", action="store_true", help="full architecture at the benchmarked size: BASELINE C2 inputs of bench.py (8 x 518^2, seed 1234, no priors) and 4 x 518^2 with pose + intrinsics priors, both presets for C2")
    ap.add_argument("--cases", default="", help="comma-separated subset of the --full-518 cases")
    ap.add_argument("--full-gs", action="store_true", help="full architecture with the 3D-Gaussian head (BASELINE config 5 path, rasterisation not run), 2 x 224^2")
    a = ap.parse_args()
    torch.manual_seed(0)
    if a.full_priors:
        cfg = WMConfig()
        m = build_reference(cfg)
        run_case(m, cfg, "full_2v_224_pose_ray", 12, 2, 224, 224, [1, 0, 1], sub=4, keep_taps=False)
        return
    if a.full_nonsquare:
        cfg = WMConfig()
        m = build_reference(cfg)
        run_case(m, cfg, "full_3v_154x210_allpriors", 14, 3, 154, 210, [1, 1, 1], sub=2, keep_taps=False)
        return
    if a.full_518:
        cfg = WMConfig()
        want = set(a.cases.split(",")) if a.cases else None
        todo = [c for c in ("full_8v_518_noprior", "full_4v_518_pose_ray", "full_2v_518_allpriors") if want is None or c in want]
        if todo:
            m = build_reference(cfg)
            if "full_8v_518_noprior" in todo:
                run_case(m, cfg, "full_8v_518_noprior", 1234, 8, 518, 518, [0, 0, 0], sub=8, keep_taps=False, regen_img=True)
            if "full_4v_518_pose_ray" in todo:
                run_case(m, cfg, "full_4v_518_pose_ray", 4321, 4, 518, 518, [1, 0, 1], sub=8, keep_taps=False, regen_img=True)
            if "full_2v_518_allpriors" in todo:  # depth prior at the benchmarked size (PatchEmbed_Mlp on 37 x 37 patches of the normalised depth)
                run_case(m, cfg, "full_2v_518_allpriors", 777, 2, 518, 518, [1, 1, 1], sub=8, keep_taps=False, regen_img=True)
            del m
        if want is None or "refinit_full_8v_518_noprior" in want:
            m = build_reference(cfg, "refinit")
            run_case(m, cfg, "refinit_full_8v_518_noprior", 1234, 8, 518, 518, [0, 0, 0], sub=8, keep_taps=False, preset="refinit", regen_img=True)
        return
    if a.full_gs:
        cfg = WMConfig(enable_gs=True)
        m = build_reference(cfg)
        run_case(m, cfg, "full_gs_2v_224", 13, 2, 224, 224, [0, 0, 0], sub=4, keep_taps=False, splat_stride=16)
        return
    if a.refinit:
        cfg = WMConfig.tiny()
        m = build_reference(cfg, "refinit")
        run_case(m, cfg, "refinit_tiny_3v_70x56_pose_ray", 21, 3, 70, 56, [1, 0, 1], preset="refinit")
        cfg = WMConfig()
        m = build_reference(cfg, "refinit")
        run_case(m, cfg, "refinit_full_2v_224_noprior", 22, 2, 224, 224, [0, 0, 0], sub=4, keep_taps=False, preset="refinit")
        return
    if not a.only_full:
        cfg = WMConfig.tiny()
        m = build_reference(cfg)
        run_case(m, cfg, "tiny_3v_70x56_pose_ray", 1, 3, 70, 56, [1, 0, 1])
        run_case(m, cfg, "tiny_2v_70x70_noprior", 2, 2, 70, 70, [0, 0, 0])
        run_case(m, cfg, "tiny_12v_56x70_allpriors", 3, 12, 56, 70, [1, 1, 1])
        run_case(m, cfg, "tiny_1v_70x70_depth", 4, 1, 70, 70, [0, 1, 0])
        cfg = WMConfig.tiny(enable_gs=True)
        m = build_reference(cfg)
        run_case(m, cfg, "tiny_gs_2v_70x70", 5, 2, 70, 70, [0, 0, 0])
    if a.full or a.only_full:
        cfg = WMConfig()
        m = build_reference(cfg)
        run_case(m, cfg, "full_2v_224_noprior", 11, 2, 224, 224, [0, 0, 0], sub=4, keep_taps=False)


if __name__ == "__main__":
    main()
