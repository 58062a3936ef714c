"""GPU: kernel error separated from recipe (rounding) error.

The HIP build rounds GEMM / attention / conv operands to 16 bits (bf16 backbone, f16 heads: the reference's own GPU
recipe rounds at least as often).  `oracle/worldmirror_ref.py forward(..., emulate=(bdt, hdt))` is the fp32
restatement with exactly those roundings inserted.  Three numbers per fixture and output:

    R = err(emulated oracle, reference)      what the recipe costs against the reference's fp32 CPU path
    F = err(emulated oracle, emulated oracle on an input perturbed by 1e-7 relative)
                                             the SELF-DECORRELATION FLOOR of rounded arithmetic: a quantiser turns a
                                             perturbation d << ulp into sqrt(d * ulp), so after a few of the ~400
                                             successive roundings of a forward ANY fp32-level difference (summation
                                             order, exp2 / rcp implementation) has grown to a fixed fraction of R.
                                             Measured here on the CPU alone: F ~ 0.45 R (fp32 itself moves by 1e-6).
    G = err(GPU, emulated oracle)

and the assertions are   G <= 1.25 F + 5e-5   (the build is one more realisation of the same rounded computation — measured
G / F = 0.93 ... 1.04 on the dense outputs of every fixture; a kernel error above the floor shows up here)   and
err(GPU, reference) <= 1.15 R + 1e-4   (no systematic error on top of the recipe's).  On the "refinit" preset (the reference's own init statistics, damped: LayerScale 0.01) the floor is
below 5e-4 and G <= 5e-4 and the north-star pts3d / depth / normals < 1e-3 vs the reference are asserted outright; on the
sensitivity-maximising preset R itself is ~3e-3 in bf16 — and so is the REFERENCE'S OWN bf16-autocast recipe
(3.3e-3, oracle/validate_emulation.py, profiles/r02_emulation_validation.md): no 16-bit recipe meets 1e-3 there.
Kernel error proper (same rounded operands in, one op) is pinned at op level: tests/test_gpu_ops.py.
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
EMU_TOL = 5e-4   # GPU vs emulated oracle where the decorrelation floor allows it (refinit preset)
NORTH_STAR = 1e-3
PERTURB = 1e-7   # relative input perturbation that defines the self-decorrelation floor


def _gpu(cfg, views, flags, preset, dtype="bf16", head_dtype="f16"):
    from hunyuanworld_mirror_amd import WorldMirror
    m = WorldMirror(arch=cfg, dtype=dtype, head_dtype=head_dtype).to("cuda:0").init_synthetic_weights(preset=preset)
    out = m({k: torch.from_numpy(v).cuda() for k, v in views.items()}, flags)
    torch.cuda.synchronize()
    res = {k: v.cpu().numpy() for k, v in out.items() if isinstance(v, torch.Tensor)}
    del m
    return res


def perturbed(views):
    """the same views with the image multiplied by (1 + 1e-7 N(0,1)): an fp32-rounding-sized change"""
    g = torch.Generator().manual_seed(0)
    img = torch.from_numpy(views["img"])
    out = dict(views)
    out["img"] = (img * (1 + PERTURB * torch.randn(img.shape, generator=g))).numpy()
    return out


def _emulated(cfg, views, flags, preset, dtype="bf16", head_dtype="f16"):
    from oracle import worldmirror_ref as R
    P = torch_weights(cfg, preset)
    with torch.no_grad():
        o = R.forward(P, {k: torch.from_numpy(v) for k, v in views.items()}, flags, cfg, emulate=(dtype, head_dtype), prune=False)
    return {k: v.numpy() for k, v in o.items() if isinstance(v, torch.Tensor)}


def _sub(a, sub, H):
    return a[:, :, ::sub, ::sub] if sub > 1 and a.ndim >= 4 and a.shape[2] == H else a


def _check(name, got, emu, pert, outs, sub, H, refinit=False):
    """got / emu / pert: GPU, emulated oracle, emulated oracle on the perturbed input.  Returns the table."""
    rows = {}
    if emu is None:  # no emulated output available (518-px fixture not generated): reference only, round-1 bounds
        for k in ("pts3d", "depth", "normals", "camera_params"):
            e = rel_l2(_sub(got[k], sub, H), outs[k])
            print(f"  {name} {k}: GPU vs reference {e:.2e} (emulated output missing)")
            assert e < (NORTH_STAR if refinit and k != "camera_params" else 5e-3), (name, k, e)
        return rows

    def fit(a, ref):
        return _sub(a, sub, H) if a.shape != ref.shape else a
    for k in KEYS:
        if k not in outs or k not in got:
            continue
        ref = outs[k]
        g, e, q = _sub(got[k], sub, H), fit(emu[k], ref), fit(pert[k], ref)
        assert g.shape == ref.shape == e.shape == q.shape, (k, g.shape, e.shape, q.shape, ref.shape)
        assert np.isfinite(g).all(), k
        rows[k] = (rel_l2(g, e), rel_l2(q, e), rel_l2(e, ref), rel_l2(g, ref))
    print("\n" + name + "   G = GPU vs emulated | F = emulated vs emulated(1e-7-perturbed input) | R = emulated vs reference | GPU vs reference")
    for k, (G, F, R, C) in rows.items():
        print(f"  {k:14s} {G:.2e} | {F:.2e} | {R:.2e} | {C:.2e}")
    for k, (G, F, R, C) in rows.items():
        if k.startswith("camera"):
            # 9 / 16 numbers per view: ONE draw of the rounding noise each for G, F, R, C, with no averaging over pixels (measured
            # G / F between 0.9 and 2.2 across fixtures) — bounded loosely here; the camera head itself runs in exact fp32 MFMA
            assert G <= 3.0 * F + 5e-4 and C <= 2.0 * R + 5e-4, (name, k, G, F, R, C)
            continue
        # dense outputs average the noise over >= 10^4 values: measured G / F = 0.93 ... 1.04 on every fixture
        assert G <= 1.25 * F + 5e-5, (name, k, "GPU vs emulated oracle above the self-decorrelation floor", G, F)
        assert C <= 1.15 * R + 1e-4, (name, k, "GPU vs reference beyond the recipe's own error", C, R)
    if refinit:  # damped weights: the floor is low enough for absolute bounds
        for k in ("pts3d", "depth", "normals"):
            assert rows[k][0] <= EMU_TOL, (name, k, "GPU vs emulated oracle", rows[k][0])
            assert rows[k][3] < NORTH_STAR, (name, k, "north-star tolerance vs the reference", rows[k][3])
    elif "pts3d" in rows and rows["pts3d"][2] < 5e-4:  # any other case whose recipe error is that small (f16 backbone is not: ~1e-3)
        assert rows["pts3d"][3] < NORTH_STAR
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
    pert = _emulated(cfg, perturbed(views), flags, preset)
    _check(name, got, emu, pert, outs, int(z["subsample"]), views["img"].shape[-2], refinit=preset == "refinit")


@pytest.mark.parametrize("name", ["tiny_3v_70x56_pose_ray", "full_2v_224_noprior"])
def test_gpu_vs_emulated_oracle_f16_backbone(name):
    """dtype='f16' (BASELINE config 5's dtype; same MFMA rate and bytes as bf16): 11 mantissa bits instead of 8."""
    cfg, views, flags, outs, z = load_golden(name)
    got = _gpu(cfg, views, flags, "sensitive", dtype="f16")
    emu = _emulated(cfg, views, flags, "sensitive", dtype="f16")
    pert = _emulated(cfg, perturbed(views), flags, "sensitive", dtype="f16")
    _check(name + " [f16]", got, emu, pert, outs, int(z["subsample"]), views["img"].shape[-2])


def test_gpu_vs_emulated_oracle_bf16_heads():
    """head_dtype='bf16' (the range-safe choice for checkpoints whose DPT activations may exceed f16's 65504)."""
    cfg, views, flags, outs, z = load_golden("tiny_3v_70x56_pose_ray")
    got = _gpu(cfg, views, flags, "sensitive", head_dtype="bf16")
    emu = _emulated(cfg, views, flags, "sensitive", head_dtype="bf16")
    pert = _emulated(cfg, perturbed(views), flags, "sensitive", head_dtype="bf16")
    _check("tiny_3v_70x56_pose_ray [bf16 heads]", got, emu, pert, outs, 1, views["img"].shape[-2])


def _emu_518(name, cfg, views, flags, preset, sub, dtype="bf16"):
    """(emulated, emulated on the perturbed input) at every sub-th pixel, from tests/golden/emu_<name>.npz (emu_f16_<name>.npz for an
    f16 backbone) or live"""
    path = os.path.join(GOLD, ("emu_" if dtype == "bf16" else "emu_" + dtype + "_") + name + ".npz")
    H = views["img"].shape[-2]
    if os.environ.get("WM_EMU_LIVE"):
        e, q = _emulated(cfg, views, flags, preset, dtype=dtype), _emulated(cfg, perturbed(views), flags, preset, dtype=dtype)
        return {k: _sub(v, sub, H) for k, v in e.items()}, {k: _sub(v, sub, H) for k, v in q.items()}
    if not os.path.exists(path):
        return None, None
    z = dict(np.load(path, allow_pickle=False))
    if "perturb" not in z:
        return None, None
    assert str(z["emulate"]) == dtype + ",f16" and str(z["weights_preset"]) == preset and float(z["perturb"]) == PERTURB
    return ({k[4:]: v for k, v in z.items() if k.startswith("out_")}, {k[5:]: v for k, v in z.items() if k.startswith("pert_")})


@pytest.mark.parametrize("name", ["full_8v_518_noprior", "full_4v_518_pose_ray", "refinit_full_8v_518_noprior", "full_2v_518_allpriors"])
def test_518_golden_and_emulated(name):
    """The benchmarked size.  BASELINE C2's own inputs (bench.py: 8 x 518 x 518, seed 1234, no priors) on both weight presets
    and 4 x 518 x 518 with camera-pose + intrinsics priors (the C3 flag set), against the reference's outputs
    (oracle/gen_golden.py --full-518: every 8th pixel + fp64 checksums of the full tensors) and the emulated oracle.
    Only at 518 x 518 is pos_embed used verbatim (vision_transformer.py:179-180), are the DPT levels 148 / 296 / 518
    (dense_head.py:217-251) with 518 not a multiple of the conv tile, and do the 32 x 8-pixel conv tile, the fused 148 -> 296
    resize, the DMA-fed 32-channel conv with the tail in its epilogue and the attention tail split run end to end.
    full_2v_518_allpriors adds the depth prior at this size (PatchEmbed_Mlp over 37 x 37 patches of the normalised depth map)."""
    if not os.path.exists(os.path.join(GOLD, name + ".npz")):
        pytest.skip("fixture missing")
    cfg, views, flags, outs, z = load_golden(name)
    preset, sub, H = golden_preset(z), int(z["subsample"]), views["img"].shape[-2]
    got = _gpu(cfg, views, flags, preset)
    emu, pert = _emu_518(name, cfg, views, flags, preset, sub)
    _check(name, got, emu, pert, outs, sub, H, refinit=preset == "refinit")
    # checksum over ALL pixels (not only the stored 1/64th): bounded by the recipe error of the stored sample
    for k in ("pts3d", "depth", "normals", "pts3d_conf", "depth_conf", "normals_conf"):
        s, ref = float(got[k].astype(np.float64).sum()), float(z["sum_" + k])
        lim = 5e-3 if k == "normals" else 2e-3  # signed sums: the components of unit normals cancel heavily
        assert abs(s - ref) / abs(ref) < lim, (k, s, ref)


def test_518_f16_backbone_all_priors():
    """BASELINE config 5's dtype (f16 backbone operands) at the benchmarked size, all three priors on: the general attention kernel
    (the pipelined no-max kernel is bf16-only), f16 GEMM epilogues and the depth-prior encoder at 518 x 518, against the reference
    and the f16-emulating oracle (oracle/gen_emulated.py --f16)."""
    name = "full_2v_518_allpriors"
    if not os.path.exists(os.path.join(GOLD, name + ".npz")):
        pytest.skip("fixture missing")
    cfg, views, flags, outs, z = load_golden(name)
    sub, H = int(z["subsample"]), views["img"].shape[-2]
    got = _gpu(cfg, views, flags, "sensitive", dtype="f16")
    emu, pert = _emu_518(name, cfg, views, flags, "sensitive", sub, dtype="f16")
    _check(name + " [f16]", got, emu, pert, outs, sub, H)
