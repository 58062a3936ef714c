"""GPU: kernel error separated from recipe (rounding) error.

The HIP build rounds GEMM / attention / conv operands to 16 bits (bf16 backbone, f16 heads: the reference's own GPU
recipe rounds at least as often).  `oracle/worldmirror_ref.py forward(..., emulate=(bdt, hdt))` is the fp32
restatement with exactly those roundings inserted, so

    err(GPU, emulated oracle)  = kernel error (fp32 summation order, hardware exp2 / rcp, indexing bugs)   -> asserted <= 5e-4
    err(emulated oracle, golden) = what the recipe itself costs against the reference's fp32 CPU path       -> measured, printed
    err(GPU, golden)           <= 1.25 x err(emulated oracle, golden) + 5e-4                                 -> asserted

on every fixture and both weight presets.  The north-star bound pts3d < 1e-3 against the reference is asserted wherever
the recipe itself meets it (err(emulated, golden) < 5e-4: the "refinit" preset in bf16, every fixture with dtype f16).
Benchmark-size (518 x 518) emulated outputs are precomputed by oracle/gen_emulated.py (tests/golden/emu_*.npz; the CPU
oracle needs minutes there); set WM_EMU_LIVE=1 to recompute them on the spot.
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, golden_preset, load_golden, rel_l2, torch_weights

pytestmark = pytest.mark.gpu

KEYS = ("pts3d", "depth", "normals", "pts3d_conf", "depth_conf", "normals_conf", "camera_params", "camera_poses")
EMU_TOL = 5e-4   # kernel error bound (GPU vs emulated oracle), every output, every fixture
NORTH_STAR = 1e-3


def _gpu(cfg, views, flags, preset, dtype="bf16", head_dtype="f16"):
    from hunyuanworld_mirror_amd import WorldMirror
    m = WorldMirror(arch=cfg, dtype=dtype, head_dtype=head_dtype).to("cuda:0").init_synthetic_weights(preset=preset)
    out = m({k: torch.from_numpy(v).cuda() for k, v in views.items()}, flags)
    torch.cuda.synchronize()
    res = {k: v.cpu().numpy() for k, v in out.items() if isinstance(v, torch.Tensor)}
    del m
    return res


def _emulated(cfg, views, flags, preset, dtype="bf16", head_dtype="f16"):
    from oracle import worldmirror_ref as R
    P = torch_weights(cfg, preset)
    with torch.no_grad():
        o = R.forward(P, {k: torch.from_numpy(v) for k, v in views.items()}, flags, cfg, emulate=(dtype, head_dtype), prune=False)
    return {k: v.numpy() for k, v in o.items() if isinstance(v, torch.Tensor)}


def _sub(a, sub, H):
    return a[:, :, ::sub, ::sub] if sub > 1 and a.ndim >= 4 and a.shape[2] == H else a


def _check(name, got, emu, outs, sub, H, expect_north_star=None):
    rows = {}
    if emu is None:  # no emulated output available (518-px fixture not generated yet): reference only, round-1 bounds
        for k in ("pts3d", "depth", "normals", "camera_params"):
            e = rel_l2(_sub(got[k], sub, H), outs[k])
            print(f"  {name} {k}: GPU vs reference {e:.2e} (emulated output missing)")
            assert e < (NORTH_STAR if expect_north_star and k != "camera_params" else 5e-3), (name, k, e)
        return rows
    for k in KEYS:
        if k not in outs or k not in got:
            continue
        g, e, ref = _sub(got[k], sub, H), _sub(emu[k], sub, H) if emu[k].shape != outs[k].shape else emu[k], outs[k]
        assert g.shape == ref.shape == e.shape, (k, g.shape, e.shape, ref.shape)
        assert np.isfinite(g).all(), k
        rows[k] = (rel_l2(g, e), rel_l2(e, ref), rel_l2(g, ref))
    print("\n" + name + "  (GPU vs emulated | emulated vs reference | GPU vs reference)")
    for k, (a, b, c) in rows.items():
        print(f"  {k:14s} {a:.2e} | {b:.2e} | {c:.2e}")
    for k, (a, b, c) in rows.items():
        assert a <= EMU_TOL, (name, k, "GPU vs emulated oracle", a)
        assert c <= 1.25 * b + EMU_TOL, (name, k, "GPU vs reference beyond the recipe's own error", c, b)
    if expect_north_star is None:
        expect_north_star = rows["pts3d"][1] < 5e-4 if "pts3d" in rows else False
    if expect_north_star and "pts3d" in rows:
        for k in ("pts3d", "depth", "normals"):
            assert rows[k][2] < NORTH_STAR, (name, k, rows[k])
    return rows


TINY = ["tiny_2v_70x70_noprior", "tiny_3v_70x56_pose_ray", "tiny_12v_56x70_allpriors", "tiny_1v_70x70_depth",
        "refinit_tiny_3v_70x56_pose_ray"]
FULL = ["full_2v_224_noprior", "full_2v_224_pose_ray", "full_3v_154x210_allpriors", "refinit_full_2v_224_noprior"]


@pytest.mark.parametrize("name", TINY + FULL)
def test_gpu_vs_emulated_oracle(name):
    cfg, views, flags, outs, z = load_golden(name)
    preset = golden_preset(z)
    got = _gpu(cfg, views, flags, preset)
    emu = _emulated(cfg, views, flags, preset)
    _check(name, got, emu, outs, int(z["subsample"]), views["img"].shape[-2],
           expect_north_star=True if preset == "refinit" else None)


@pytest.mark.parametrize("name", ["tiny_3v_70x56_pose_ray", "full_2v_224_noprior"])
def test_gpu_vs_emulated_oracle_f16_backbone(name):
    """dtype='f16' (BASELINE config 5's dtype; same MFMA rate and bytes as bf16): 11 mantissa bits instead of 8."""
    cfg, views, flags, outs, z = load_golden(name)
    got = _gpu(cfg, views, flags, "sensitive", dtype="f16")
    emu = _emulated(cfg, views, flags, "sensitive", dtype="f16")
    _check(name + " [f16]", got, emu, outs, int(z["subsample"]), views["img"].shape[-2])


def test_gpu_vs_emulated_oracle_bf16_heads():
    """head_dtype='bf16' (the range-safe choice for checkpoints whose DPT activations may exceed f16's 65504)."""
    cfg, views, flags, outs, z = load_golden("tiny_3v_70x56_pose_ray")
    got = _gpu(cfg, views, flags, "sensitive", head_dtype="bf16")
    emu = _emulated(cfg, views, flags, "sensitive", head_dtype="bf16")
    _check("tiny_3v_70x56_pose_ray [bf16 heads]", got, emu, outs, 1, views["img"].shape[-2])


def _emu_518(name, cfg, views, flags, preset, sub):
    path = os.path.join(GOLD, "emu_" + name + ".npz")
    if os.environ.get("WM_EMU_LIVE") or not os.path.exists(path):
        if not os.environ.get("WM_EMU_LIVE"):
            return None
        e = _emulated(cfg, views, flags, preset)
        return {k: _sub(v, sub, views["img"].shape[-2]) for k, v in e.items()}
    z = dict(np.load(path, allow_pickle=False))
    assert str(z["emulate"]) == "bf16,f16" and str(z["weights_preset"]) == preset
    return {k[4:]: v for k, v in z.items() if k.startswith("out_")}


@pytest.mark.parametrize("name", ["full_8v_518_noprior", "full_4v_518_pose_ray", "refinit_full_8v_518_noprior"])
def test_518_golden_and_emulated(name):
    """The benchmarked size.  BASELINE C2's own inputs (bench.py: 8 x 518 x 518, seed 1234, no priors) on both weight presets
    and 4 x 518 x 518 with camera-pose + intrinsics priors (the C3 flag set), against the reference's outputs
    (oracle/gen_golden.py --full-518: every 8th pixel + fp64 checksums of the full tensors) and the emulated oracle.
    Only at 518 x 518 is pos_embed used verbatim (vision_transformer.py:179-180), are the DPT levels 148 / 296 / 518
    (dense_head.py:217-251) with 518 not a multiple of the conv tile, and do the 32 x 8-pixel conv tile, the fused 148 -> 296
    resize, the DMA-fed 32-channel conv with the tail in its epilogue and the attention tail split run end to end."""
    if not os.path.exists(os.path.join(GOLD, name + ".npz")):
        pytest.skip("fixture missing")
    cfg, views, flags, outs, z = load_golden(name)
    preset, sub, H = golden_preset(z), int(z["subsample"]), views["img"].shape[-2]
    got = _gpu(cfg, views, flags, preset)
    emu = _emu_518(name, cfg, views, flags, preset, sub)
    _check(name, got, emu, outs, sub, H, expect_north_star=True if preset == "refinit" else None)
    # checksum over ALL pixels (not only the stored 1/64th): bounded by the recipe error of the stored sample
    for k in ("pts3d", "depth", "normals", "pts3d_conf", "depth_conf", "normals_conf"):
        s, ref = float(got[k].astype(np.float64).sum()), float(z["sum_" + k])
        lim = 5e-3 if k == "normals" else 2e-3  # signed sums: the components of unit normals cancel heavily
        assert abs(s - ref) / abs(ref) < lim, (k, s, ref)
