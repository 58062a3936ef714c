"""GPU: the drop-in boundary's ownership and loading rules (include/wm_hip.h).

* wm_forward never allocates: a forward without wm_reserve for its shape is refused; the Python mirror reserves on a
  shape change; a caller-owned workspace of wm_workspace_bytes() works and gives identical results, a smaller one is refused.
* weights re-loaded on a live handle take effect (the weight-derived workspace tables are rebuilt).
* strict=False loading (PyTorchModelHubMixin, worldmirror.py:13,16): a tensor missing from the checkpoint keeps the
  init value of the reference's constructor where that is deterministic (LayerNorm 1 / 0, LayerScale 1.0 DINO / 0.01).
* several handles on one device may share one copy of the weights.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu


def _views(views):
    return {k: torch.from_numpy(v).cuda() for k, v in views.items()}


def test_forward_refuses_without_reserve_and_never_allocates():
    from hunyuanworld_mirror_amd import WorldMirror, _lib
    cfg, views, flags, outs, z = load_golden("tiny_2v_70x70_noprior")
    m = WorldMirror(arch=cfg).init_synthetic_weights().to("cuda:0")
    L = _lib.lib()
    img = torch.from_numpy(views["img"])[0].cuda().contiguous()
    n, _, H, W = img.shape
    o = _lib.wm_outputs()
    keep = []
    for f, shape in (("camera_params", (n, 9)), ("pts3d", (n, H, W, 3)), ("pts3d_conf", (n, H, W))):
        t = torch.empty(shape, device="cuda")
        keep.append(t)
        setattr(o, f, t.data_ptr())
    fl = (C.c_int32 * 3)(0, 0, 0)
    st = L.wm_forward(m._handle, _lib.ptr(img), n, H, W, None, None, None, fl, C.byref(o), None)
    assert st != 0 and "wm_reserve" in m._err()           # no workspace yet: refused, nothing allocated behind the caller's back
    assert L.wm_reserve(m._handle, n, n, H, W) == 0
    assert L.wm_forward(m._handle, _lib.ptr(img), n, H, W, None, None, None, fl, C.byref(o), None) == 0  # first launches load the code objects
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        assert L.wm_forward(m._handle, _lib.ptr(img), n, H, W, None, None, None, fl, C.byref(o), None) == 0
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0           # device memory untouched by the forward
    assert L.wm_forward(m._handle, _lib.ptr(img), n, H + 14, W, None, None, None, fl, C.byref(o), None) != 0  # other shape: refused


def test_caller_owned_workspace():
    from hunyuanworld_mirror_amd import WorldMirror
    cfg, views, flags, outs, z = load_golden("tiny_3v_70x56_pose_ray")
    tv = _views(views)
    m = WorldMirror(arch=cfg).init_synthetic_weights().to("cuda:0")
    ref = {k: v.clone() for k, v in m(tv, flags).items()}
    n, H, W = tv["img"].shape[1], tv["img"].shape[3], tv["img"].shape[4]
    need = m.workspace_bytes(n, n, H, W)
    assert need > 0
    m.use_workspace(torch.empty(need - 256, dtype=torch.uint8, device="cuda"))
    with pytest.raises(RuntimeError, match="too small"):
        m(tv, flags)
    m.use_workspace(torch.empty(need, dtype=torch.uint8, device="cuda"))
    got = m(tv, flags)
    torch.cuda.synchronize()
    for k in ("pts3d", "depth", "normals", "camera_params"):
        assert torch.equal(got[k], ref[k]), k
    m.use_workspace(None)                                   # back to the library's own arena
    got = m(tv, flags)
    torch.cuda.synchronize()
    assert torch.equal(got["pts3d"], ref["pts3d"])


def test_reloading_weights_on_a_live_handle_takes_effect():
    """plan() used to cache the pos_embed / init_token derived tables on the shape alone."""
    from hunyuanworld_mirror_amd import WorldMirror
    cfg, views, flags, outs, z = load_golden("tiny_3v_70x56_pose_ray")   # 70 x 56: pos_embed is resampled on the host
    tv = _views(views)
    m = WorldMirror(arch=cfg).to("cuda:0").init_synthetic_weights(seed=5)
    a = {k: v.clone() for k, v in m(tv, flags).items()}
    m.init_synthetic_weights(seed=0)                        # same handle, same shape, other weights
    b = m(tv, flags)
    fresh = WorldMirror(arch=cfg).to("cuda:0").init_synthetic_weights(seed=0)(tv, flags)
    torch.cuda.synchronize()
    for k in ("pts3d", "depth", "camera_params"):
        assert torch.equal(b[k], fresh[k]), k
        assert not torch.equal(a[k], b[k]), k
        assert rel_l2(b[k].cpu().numpy(), outs[k]) < 6e-3


def test_missing_tensors_keep_reference_init_values():
    from hunyuanworld_mirror_amd import WorldMirror
    from hunyuanworld_mirror_amd.weights import iter_params
    cfg, views, flags, outs, z = load_golden("tiny_2v_70x70_noprior")
    tv = _views(views)
    full = dict(iter_params(cfg))
    drop = [k for k in full if ".norm1." in k or k.endswith("ls2.gamma") or "token_norm" in k]
    assert len(drop) > 20
    part = {k: v for k, v in full.items() if k not in drop}
    m = WorldMirror(arch=cfg)
    missing, unexpected = m.load_state_dict(part)
    assert sorted(missing) == sorted(drop) and not unexpected
    m.to("cuda:0")
    assert m.missing_weights == len(drop) and sorted(m.missing_weight_names) == sorted(drop)
    got = m(tv, flags)
    # the same model with the init values written out explicitly
    expl = dict(part)
    for k in drop:
        if k.endswith(".gamma"):
            expl[k] = np.full(full[k].shape, 1.0 if ".patch_embed." in k else 0.01, np.float32)
        elif k.endswith(".weight"):
            expl[k] = np.ones(full[k].shape, np.float32)
        else:
            expl[k] = np.zeros(full[k].shape, np.float32)
    m2 = WorldMirror(arch=cfg)
    m2.load_state_dict(expl)
    ref = m2.to("cuda:0")(tv, flags)
    torch.cuda.synchronize()
    assert m2.missing_weights == 0
    for k in ("pts3d", "depth", "normals", "camera_params"):
        assert torch.isfinite(got[k]).all()
        assert torch.equal(got[k], ref[k]), k


def test_shared_weights_between_handles():
    from hunyuanworld_mirror_amd import WorldMirror
    cfg, views, flags, outs, z = load_golden("tiny_3v_70x56_pose_ray")
    tv = _views(views)
    a = WorldMirror(arch=cfg).to("cuda:0").init_synthetic_weights()
    free0 = torch.cuda.mem_get_info()[0]
    b = WorldMirror(arch=cfg).to("cuda:0").share_weights_from(a)
    ra, rb = a(tv, flags), b(tv, flags)
    torch.cuda.synchronize()
    for k in ("pts3d", "depth", "camera_params"):
        assert torch.equal(ra[k], rb[k]), k
    del b
    ra2 = a(tv, flags)                                      # the owner is intact after a sharer is destroyed
    torch.cuda.synchronize()
    assert torch.equal(ra2["pts3d"], ra["pts3d"])


def test_batch_of_two_equals_two_forwards():
    """The reference takes [B, S, 3, H, W] (visual_transformer.py:271-277 folds B into B*S; frame attention is per view, global
    attention per batch element, so elements are independent): B = 2 must equal the two single-element forwards, stacked."""
    from hunyuanworld_mirror_amd import WorldMirror
    cfg, views, flags, outs, z = load_golden("tiny_3v_70x56_pose_ray")
    m = WorldMirror(arch=cfg).init_synthetic_weights().to("cuda:0")
    tv = {k: torch.from_numpy(v).cuda() for k, v in views.items()}
    g = torch.Generator().manual_seed(1)
    tv2 = {k: v.clone() for k, v in tv.items()}
    tv2["img"] = torch.rand(tv["img"].shape, generator=g).cuda()
    both = {k: torch.cat([tv[k], tv2[k]], 0) for k in tv}
    a, b = m(tv, flags), m(tv2, flags)
    ab = m(both, flags)
    for k in ("pts3d", "depth", "normals", "camera_params", "camera_poses", "pts3d_conf"):
        assert ab[k].shape[0] == 2 and torch.equal(ab[k][0:1], a[k]) and torch.equal(ab[k][1:2], b[k]), k
    if "splats" in ab:
        assert len(ab["splats"]["means"]) == 2 and torch.equal(ab["splats"]["means"][1], b["splats"]["means"][0])
