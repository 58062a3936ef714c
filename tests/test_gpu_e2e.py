"""GPU: end-to-end parity of WorldMirror.forward (HIP, through the C ABI) against
  (a) the committed golden fixtures = outputs of the reference itself (CPU fp32), and
  (b) the CPU oracle on the same seeded inputs.

Tolerances (north_star): point-map relative L2 < 1e-3 vs the reference fp32 path; the backbone runs
bf16 MFMA with fp32 accumulate / residual stream (the reference's own GPU recipe), DPT heads f16 MFMA
with fp32 activations, camera head fp32.  camera_params is the most bf16-sensitive output
and is reported separately.
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, load_golden, rel_l2

pytestmark = pytest.mark.gpu

# Accuracy bounds live in tests/test_gpu_emulated.py, where they are DERIVED per fixture from the rounding-emulated oracle
# (GPU vs emulated <= 1.25 F + 5e-5, GPU vs reference <= 1.15 R + 1e-4; refinit preset: the north-star 1e-3 outright) — the
# hand-tuned per-output tolerances of round 1 are gone.  This file checks the plumbing of the same fixtures: shapes, finiteness,
# taps, full-resolution checksums, and a gross-error gate that no recipe noise can reach.
GROSS = 2e-2
DENSE = ("pts3d", "depth", "normals", "pts3d_conf", "depth_conf", "normals_conf", "camera_params", "camera_poses", "camera_intrs")


def _model(cfg, **kw):
    from hunyuanworld_mirror_amd import WorldMirror
    return WorldMirror(arch=cfg, **kw).init_synthetic_weights().to("cuda:0")


def _run(m, views, flags):
    tv = {k: torch.from_numpy(v).cuda() for k, v in views.items()}
    out = m(tv, flags)
    torch.cuda.synchronize()
    return out


_MODELS = {}


def _cached_model(cfg, **kw):
    key = json.dumps(cfg.to_dict(), sort_keys=True) + json.dumps(kw, sort_keys=True)
    if key not in _MODELS:
        _MODELS.clear()
        _MODELS[key] = _model(cfg, **kw)
    return _MODELS[key]


@pytest.mark.parametrize("name", ["tiny_2v_70x70_noprior", "tiny_3v_70x56_pose_ray", "tiny_12v_56x70_allpriors",
                                  "tiny_1v_70x70_depth"])
def test_tiny_golden(name):
    cfg, views, flags, outs, z = load_golden(name)
    m = _cached_model(cfg)
    m.return_taps = True
    got = _run(m, views, flags)
    errs = {}
    for i in range(4):
        errs[f"tap{i}"] = rel_l2(got["taps"][i].cpu().numpy(), z[f"tap{i}"])
    for k, v in outs.items():
        g = got[k].cpu().numpy()
        assert g.shape == v.shape, (k, g.shape, v.shape)
        assert np.isfinite(g).all(), k
        errs[k] = rel_l2(g, v)
    print(name, {k: f"{e:.2e}" for k, e in errs.items()})
    for i in range(4):
        assert errs[f"tap{i}"] < 2e-2
    for k in DENSE:
        assert errs[k] < GROSS, (k, errs[k])


def test_tiny_golden_f16_backbone():
    """dtype='f16' (BASELINE config 5's dtype: 3 more mantissa bits in every backbone operand) must be closer to the reference than
    bf16 on the dense outputs — a relation between two measured errors, not a tuned number."""
    cfg, views, flags, outs, z = load_golden("tiny_3v_70x56_pose_ray")
    got16 = _run(_model(cfg, dtype="f16"), views, flags)
    gotbf = _run(_cached_model(cfg), views, flags)
    for k in ("pts3d", "depth", "normals"):
        e16, ebf = rel_l2(got16[k].cpu().numpy(), outs[k]), rel_l2(gotbf[k].cpu().numpy(), outs[k])
        print("\nf16 vs bf16 backbone", k, f"{e16:.2e} {ebf:.2e}")
        assert e16 < 0.75 * ebf, (k, e16, ebf)


@pytest.mark.parametrize("name", ["full_2v_224_noprior", "full_2v_224_pose_ray", "full_3v_154x210_allpriors"])
def test_full_arch_2x224_golden(name):
    """The full 1.23 B-parameter architecture against the reference's outputs: 2 x 224^2 BASELINE config C1 (no priors) and
    the C3 flag set (camera-pose + intrinsics priors on); 3 x 154 x 210 (non-square) with pose + depth + intrinsics priors."""
    cfg, views, flags, outs, z = load_golden(name)
    m = _cached_model(cfg)
    got = _run(m, views, flags)
    sub = int(z["subsample"])
    errs = {}
    for k, v in outs.items():
        g = got[k].cpu().numpy()
        if g.ndim >= 4 and g.shape[2] == views["img"].shape[-2]:
            g = g[:, :, ::sub, ::sub]
        assert g.shape == v.shape, k
        errs[k] = rel_l2(g, v)
    print(name, {k: f"{e:.2e}" for k, e in errs.items()})
    for k in DENSE:
        assert errs[k] < GROSS, (k, errs[k])
    # checksum-of-everything property at full resolution (not just the subsampled pixels)
    for k in ("pts3d", "depth", "normals"):
        s = float(got[k].double().sum())
        ref = float(z["sum_" + k])
        # signed sums: the components of unit normals cancel heavily, so their checksum is the loosest
        assert abs(s - ref) / abs(ref) < (5e-3 if k == "normals" else 2e-3), (k, s, ref)


@pytest.mark.parametrize("name", ["refinit_tiny_3v_70x56_pose_ray", "refinit_full_2v_224_noprior"])
def test_refinit_weights_bf16_meets_north_star(name):
    """With the reference's own init statistics (weights.py preset "refinit": DINO trunc_normal 0.02 / LayerScale 1,
    torch-default U(+-1/sqrt(fan_in)) elsewhere, LayerScale 0.01 in the multi-view blocks — what the SURVEY's
    3.0e-4 emulation used) the bf16 recipe meets the north-star tolerance: pts3d / depth / normals rel-L2 < 1e-3
    against the reference's fp32 CPU output.  Measured r01: full 2x224 pts3d 3.2e-4, depth 4.6e-5, normals 4.3e-4;
    tiny pts3d 1.6e-4.  Camera parameters (cam token ~1e-6 + 0.01-scaled bf16 increments) are looser: < 2e-2."""
    from conftest import golden_preset
    cfg, views, flags, outs, z = load_golden(name)
    from hunyuanworld_mirror_amd import WorldMirror
    m = WorldMirror(arch=cfg).to("cuda:0").init_synthetic_weights(preset=golden_preset(z))
    got = _run(m, views, flags)
    sub = int(z["subsample"])
    errs = {}
    for k, v in outs.items():
        g = got[k].cpu().numpy()
        if sub > 1 and g.ndim >= 4 and g.shape[2] == views["img"].shape[-2]:
            g = g[:, :, ::sub, ::sub]
        errs[k] = rel_l2(g, v)
    print(name, {k: f"{e:.2e}" for k, e in errs.items()})
    assert errs["pts3d"] < 1e-3 and errs["depth"] < 1e-3 and errs["normals"] < 1e-3, errs
    # camera outputs: 2 x the measured values (r02: camera_params 4.3e-3 / 5.8e-3, camera_poses 7.0e-3 / 1.2e-2 on the two fixtures;
    # the rounding-emulated oracle itself sits at 3.4e-3 / 5.1e-3 and 4.9e-3 / 1.2e-2: tests/test_gpu_emulated.py)
    assert errs["camera_params"] < 1.2e-2 and errs["camera_poses"] < 2.4e-2, errs


def test_tiny_gs_branch_golden():
    """BASELINE config 5 path (3D-Gaussian head, rasterisation stubbed): gs_depth, the per-pixel splats of
    prepare_splats (rasterization.py:389-498) and the voxel-pruned splats (:301-387) vs the reference."""
    cfg, views, flags, outs, z = load_golden("tiny_gs_2v_70x70")
    m = _model(cfg)
    got = _run(m, views, flags)
    errs = {k: rel_l2(got[k].cpu().numpy(), outs[k]) for k in ("gs_depth", "gs_depth_conf", "camera_params")}
    for k in ("means", "quats", "scales", "opacities", "sh", "weights"):
        errs["raw_" + k] = rel_l2(got["splats_raw"][k][0].cpu().numpy(), z["splats_raw_" + k])
    # voxel pruning sorts by voxel id (0.002 units): the order is not stable under 1e-3 perturbations of the
    # means, so prune_gs is checked on the reference's own raw splats, where it must reproduce the reference
    from hunyuanworld_mirror_amd.worldmirror import prune_gs
    raw = {k: torch.from_numpy(z["splats_raw_" + k])[None].cuda() for k in ("means", "quats", "scales", "opacities", "sh", "weights")}
    pr = prune_gs(raw)
    for k in ("means", "quats", "scales", "opacities", "sh"):
        a = pr[k][0].cpu().numpy()
        assert a.shape == z["splats_" + k].shape, (k, a.shape)
        errs["pruned_" + k] = rel_l2(a, z["splats_" + k])
        assert got["splats"][k][0].shape[1:] == a.shape[1:]
    print("gs", {k: f"{e:.2e}" for k, e in errs.items()})
    assert errs["gs_depth"] < 2e-3 and errs["gs_depth_conf"] < 1e-3
    for k, e in errs.items():
        assert e < 1e-2, (k, e)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_full_arch_gs_branch_golden(dtype):
    """BASELINE config 5's path at the full 1.26 B-parameter architecture (3D-Gaussian head on, rasterisation stubbed as the
    reference's result is discarded, rasterization.py:243-246), 2 x 224^2, against the reference's outputs."""
    cfg, views, flags, outs, z = load_golden("full_gs_2v_224")
    m = _model(cfg, dtype=dtype)  # config 5 quotes fp16; bf16 is the default recipe
    got = _run(m, views, flags)
    sub, st = int(z["subsample"]), int(z["splat_stride"])
    errs = {}
    for k in ("gs_depth", "gs_depth_conf", "camera_params"):
        g = got[k].cpu().numpy()
        if g.ndim >= 4 and g.shape[2] == 224:
            g = g[:, :, ::sub, ::sub]
        assert g.shape == outs[k].shape, k
        errs[k] = rel_l2(g, outs[k])
    for k in ("means", "quats", "scales", "opacities", "sh", "weights"):
        raw = got["splats_raw"][k][0]
        errs["raw_" + k] = rel_l2(raw.cpu().numpy()[::st], z["splats_raw_" + k])
        ref = float(z["sum_splats_raw_" + k])  # fp64 checksum over ALL per-pixel splats (not only the stored 16th)
        errs["sum_" + k] = abs(float(raw.double().sum()) - ref) / max(abs(ref), 1.0)
    print("full gs", dtype, {k: f"{e:.2e}" for k, e in errs.items()})
    # means = unprojection of gs_depth through the PREDICTED camera (rasterization.py:469-484): they inherit the camera
    # head's error (focal / translation), the other attributes come straight from the head
    for k in ("quats", "scales", "opacities", "sh", "weights"):
        assert errs["sum_" + k] < 5e-3, (k, errs["sum_" + k])
    assert errs["sum_means"] < 2e-2
    assert errs["gs_depth"] < 2e-3 and errs["gs_depth_conf"] < 1e-3 and errs["camera_params"] < 5e-3
    for k, e in errs.items():
        assert e < (2e-2 if "means" in k else 1e-2), (k, e)


def test_errors_mirror_reference():
    from hunyuanworld_mirror_amd import WorldMirror, WMConfig
    cfg = WMConfig.tiny()
    m = _cached_model(cfg)
    with pytest.raises(ValueError):  # visual_transformer.py:272-273
        m({"img": torch.rand(1, 2, 4, 70, 70).cuda()})
    with pytest.raises(AssertionError):  # patch_embed.py:67-68
        m({"img": torch.rand(1, 2, 3, 72, 70).cuda()})
    with pytest.raises(RuntimeError):
        WorldMirror(arch=cfg).init_synthetic_weights()({"img": torch.rand(1, 1, 3, 70, 70)})


def test_from_pretrained_then_forward_equals_synthetic(tmp_path):
    """The callers' load path (infer.py:95-96: WorldMirror.from_pretrained(dir).to(device); eval): a checkpoint directory
    written from the synthetic weights must give bit-identical outputs to the directly initialised model."""
    from safetensors.torch import save_file
    from hunyuanworld_mirror_amd import WorldMirror
    from hunyuanworld_mirror_amd.weights import iter_params
    cfg, views, flags, outs, z = load_golden("tiny_3v_70x56_pose_ray")
    d = tmp_path / "ckpt"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(dict(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, enable_gs=False)))
    save_file({k: torch.from_numpy(v.copy()) for k, v in iter_params(cfg)}, str(d / "model.safetensors"))
    m = WorldMirror.from_pretrained(str(d), arch=cfg).to("cuda:0")
    m.eval()
    got = _run(m, views, flags)
    ref = _run(_cached_model(cfg), views, flags)
    for k in ("pts3d", "depth", "normals", "camera_params", "camera_poses"):
        assert torch.equal(got[k], ref[k]), k
        assert rel_l2(got[k].cpu().numpy(), outs[k]) < GROSS
