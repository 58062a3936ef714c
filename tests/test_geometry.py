"""Post-path geometry: depth_to_world_coords_points.

CPU: the numpy oracle against outputs of the reference function itself (tests/golden/geometry_depth_to_world.npz,
written by oracle/gen_golden_geometry.py).  GPU: the HIP kernel through the C ABI and through the Python mirror
against the same golden and against the oracle at full size.  Tolerances: camera points bit-exact (same fp32
expression order), world points rel-L2 < 1e-6 (3-term dot product: summation order / FMA contraction may differ from
torch's bmm), mask exact."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, rel_l2
from oracle import geometry_ref as G


def _gold():
    return np.load(os.path.join(GOLD, "geometry_depth_to_world.npz"))


def test_oracle_matches_reference_geometry():
    z = _gold()
    world, cam, mask = G.depth_to_world_coords_points(z["depth"], z["extrinsic"], z["intrinsic"])
    assert np.array_equal(mask, z["mask"])
    assert np.array_equal(cam, z["cam"])
    assert rel_l2(world, z["world"]) < 1e-6
    assert G.depth_to_world_coords_points(z["depth"], z["extrinsic"], z["intrinsic"])[0].shape == z["world"].shape


@pytest.mark.gpu
def test_gpu_depth_to_world_golden():
    from hunyuanworld_mirror_amd import depth_to_world_coords_points
    z = _gold()
    dev = torch.device("cuda:0")
    world, cam, mask = depth_to_world_coords_points(torch.from_numpy(z["depth"]).to(dev), torch.from_numpy(z["extrinsic"]).to(dev),
                                                   torch.from_numpy(z["intrinsic"]).to(dev))
    assert mask.dtype == torch.bool and np.array_equal(mask.cpu().numpy(), z["mask"])
    assert np.array_equal(cam.cpu().numpy(), z["cam"]), "camera points follow the reference's fp32 expression order exactly"
    assert rel_l2(world.cpu().numpy(), z["world"]) < 1e-6
    assert depth_to_world_coords_points(None, None, None) == (None, None, None)  # geometry.py:72-73
    with pytest.raises(RuntimeError):
        depth_to_world_coords_points(torch.from_numpy(z["depth"]), torch.from_numpy(z["extrinsic"]), torch.from_numpy(z["intrinsic"]))


@pytest.mark.gpu
def test_gpu_depth_to_world_full_size_and_bandwidth():
    """8 x 518^2 (BASELINE C2's output size): equals the oracle; prints the achieved HBM rate (29 B per pixel)."""
    from hunyuanworld_mirror_amd import depth_to_world_coords_points
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    B, H, W = 8, 518, 518
    depth = torch.rand(B, H, W, generator=g) * 4
    depth[depth < 0.2] = 0
    ext = torch.eye(4).repeat(B, 1, 1)
    ext[:, :3, :3] = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))[0]
    ext[:, :3, 3] = torch.randn(B, 3, generator=g)
    K = torch.tensor([[500.0, 0, 259], [0, 510.0, 258.5], [0, 0, 1]]).repeat(B, 1, 1)
    d, e, k = depth.to(dev), ext.to(dev), K.to(dev)
    world, cam, mask = depth_to_world_coords_points(d, e, k)
    torch.cuda.synchronize()
    ow, oc, om = G.depth_to_world_coords_points(depth.numpy(), ext.numpy(), K.numpy())
    assert np.array_equal(mask.cpu().numpy(), om)
    assert np.array_equal(cam.cpu().numpy(), oc)
    assert rel_l2(world.cpu().numpy(), ow) < 1e-6
    # kernel alone, through the C ABI, outputs preallocated
    import ctypes as C
    from hunyuanworld_mirror_amd import _lib
    L = _lib.lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    m8 = torch.empty(B, H, W, device=dev, dtype=torch.uint8)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        L.wm_depth_to_world(p(d), p(e), p(k), p(world), p(cam), p(m8), B, H, W, C.c_float(1e-8), s)
    e0.record()
    for _ in range(50):
        L.wm_depth_to_world(p(d), p(e), p(k), p(world), p(cam), p(m8), B, H, W, C.c_float(1e-8), s)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    gbs = B * H * W * 29 / (us * 1e-6) / 1e9
    print(f"depth_to_world 8x518^2: {us:.1f} us, {gbs:.0f} GB/s of 8000 (algorithmic 29 B/pixel)")
    # odd pixel count -> scalar path
    w1, c1, k1 = depth_to_world_coords_points(d[:1, :5, :7].contiguous(), e[:1], k[:1])
    o1 = G.depth_to_world_coords_points(depth[:1, :5, :7].numpy(), ext[:1].numpy(), K[:1].numpy())
    assert np.array_equal(c1.cpu().numpy(), o1[1]) and rel_l2(w1.cpu().numpy(), o1[0]) < 1e-6


# ------------------------------------------------------------------ create_confidence_mask (infer.py:25-59)
def _gold_mask():
    return np.load(os.path.join(GOLD, "geometry_confidence_mask.npz"))


def test_oracle_confidence_mask_matches_reference():
    z = _gold_mask()
    for pct in (30.0, 0.0, 55.5, 99.99):
        got = G.create_confidence_mask(z["conf"], pct)
        assert got.shape == z[f"mask_{pct}"].shape
        assert np.array_equal(got, z[f"mask_{pct}"]), pct


@pytest.mark.gpu
def test_gpu_confidence_mask_golden_and_edges():
    from hunyuanworld_mirror_amd import create_confidence_mask
    dev = torch.device("cuda:0")
    z = _gold_mask()
    conf = torch.from_numpy(z["conf"]).to(dev)
    for pct in (30.0, 0.0, 55.5, 99.99):
        m = create_confidence_mask(conf, pct)
        assert m.dtype == torch.bool and m.shape == (conf.numel(),)
        assert np.array_equal(m.cpu().numpy(), z[f"mask_{pct}"]), pct
    # ties at the threshold, everything invalid, a single element, negative values, +inf and NaN: equal to the oracle
    g = torch.Generator().manual_seed(0)
    cases = {
        "ties": torch.randint(0, 7, (10007,), generator=g).float(),
        "all_invalid": torch.zeros(5000),
        "single": torch.tensor([0.5]),
        "mixed": torch.cat([torch.randn(3000, generator=g), torch.tensor([float("inf"), float("nan"), 1e-5, 2e-5])]),
    }
    for name, c in cases.items():
        for pct in (30.0, 0.0, 80.0):
            got = create_confidence_mask(c.to(dev), pct).cpu().numpy()
            want = G.create_confidence_mask(c.numpy(), pct)
            if name == "mixed":  # numpy's argsort puts NaN last, torch.topk (and the kernel) first: compare counts + ordering only
                n = c.numel(); k = max(1, int(np.ceil(n * (100.0 - pct) / 100.0))) if pct > 0 else n
                assert got.sum() == k
                cm = c.clone(); cm[cm <= 1e-5] = -float("inf"); cm[torch.isnan(cm)] = float("inf")
                assert cm.numpy()[got].min() >= cm.numpy()[~got].max() if (~got).any() else True
            else:
                assert np.array_equal(got, want), (name, pct)


@pytest.mark.gpu
def test_gpu_confidence_mask_full_size():
    """8 x 518^2 confidences (BASELINE C2 output size): equals the oracle (sort-based), count exact; prints the time."""
    from hunyuanworld_mirror_amd import create_confidence_mask
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    conf = 1.0 + torch.rand(8, 518, 518, generator=g) * 20
    conf[torch.rand(8, 518, 518, generator=g) < 0.05] = 0
    c = conf.to(dev)
    m = create_confidence_mask(c, 30.0)
    torch.cuda.synchronize()
    want = G.create_confidence_mask(conf.numpy(), 30.0)
    assert np.array_equal(m.cpu().numpy(), want)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        create_confidence_mask(c, 30.0)
    e1.record(); torch.cuda.synchronize()
    n = conf.numel()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"confidence mask {n} elements: {us:.0f} us incl. workspace allocation ({6 * 4 * n / (us * 1e-6) / 1e9:.0f} GB/s over its 6 read passes)")


# ------------------------------------------------------------------------------------------------ prune_gs (voxel merge)
@pytest.mark.gpu
@pytest.mark.parametrize("n,spread", [(200000, 0.08), (5000, 3.0), (1, 1.0), (4096, 0.0005)])
def test_gpu_prune_gs_matches_oracle(n, spread):
    """wm_prune_gs (splat_prune.hip) against the CPU oracle (oracle/worldmirror_ref.prune_gs, itself pinned to the
    reference's prune_gs by the tiny_gs golden): same voxel set and order (torch.unique), sums in index order -> the
    merged attributes agree to the last bits; negative coordinates, heavy collisions (spread 0.08 / voxel 0.002: ~3
    splats per voxel), one voxel holding everything, a single splat."""
    import torch
    from hunyuanworld_mirror_amd.worldmirror import prune_gs
    from oracle import worldmirror_ref as WR
    g = torch.Generator().manual_seed(n)
    sp = {"means": (torch.rand(n, 3, generator=g) - 0.5) * spread + torch.tensor([0.3, -0.7, 1.1]),
          "quats": torch.randn(n, 4, generator=g), "scales": torch.rand(n, 3, generator=g) * 0.05,
          "opacities": torch.rand(n, generator=g), "sh": torch.randn(n, 1, 3, generator=g), "weights": torch.rand(n, generator=g) + 1e-3}
    ref = WR.prune_gs(sp)
    got = prune_gs({k: v[None].cuda() for k, v in sp.items()})
    K = ref["means"].shape[0]
    print(f"prune_gs n={n}: {K} voxels")
    for k in ("means", "quats", "scales", "opacities", "sh"):
        a = got[k][0].cpu()
        assert a.shape == ref[k].shape, (k, a.shape, ref[k].shape)
        tol = 1e-5 if k == "quats" else 2e-6   # quats: torch.norm's own reduction order in the normalisation
        assert torch.allclose(a, ref[k], rtol=tol, atol=tol / 10), (k, float((a - ref[k]).abs().max()))
    if n > 1000:
        again = prune_gs({k: v[None].cuda() for k, v in sp.items()})
        assert all(torch.equal(again[k][0], got[k][0]) for k in got)   # no float atomics: bit-exact run to run
