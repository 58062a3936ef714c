"""CPU, world_size 2, gloo: the view-sharded evaluation of the path equals the unsharded one.

Each rank runs the oracle on its own views; the only exchanges are the per-global-layer all-gather
of K/V and the camera-token all-gather, carried here by torch.distributed (gloo).  The per-rank inputs
are cut by the PRODUCT's own host code — hunyuanworld_mirror_amd.worldmirror.shard_inputs, the function
WorldMirror.forward calls before wm_forward_sharded — so this pins both the rules (token slot 0 belongs to GLOBAL view 0,
priors are normalised over all views before sharding, K/V gather order is irrelevant to softmax) and the code that applies
them.  (shard_inputs is pure torch; the HIP library is not loaded here.)
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden, rel_l2


def _worker(rank, world, port, name, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from conftest import load_golden, torch_weights
    from oracle import worldmirror_ref as R
    cfg, views, flags, outs, z = load_golden(name)
    P = torch_weights(cfg)
    tv = {k: torch.from_numpy(v) for k, v in views.items()}
    from hunyuanworld_mirror_amd.worldmirror import shard_inputs
    sh = shard_inputs(tv, flags, rank, world, cfg.patch_size)          # the product's slicing + prior normalisation
    n, v0 = sh["n"], sh["first_view"]
    assert v0 == rank * (tv["img"].shape[1] // world) and sh["img"].shape[0] == n
    # back to the oracle's (depths, rays, poses) tuple with a leading batch axis
    priors = None
    if sum(flags):
        priors = tuple(None if sh[k] is None else sh[k][None] for k in ("depth", "ray", "pose"))

    def gather_seq(t):  # [1,H,L,hd] -> [1,H,world*L,hd]
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t.contiguous())
        return torch.cat(parts, 2)

    def gather_tok(t):  # [1,n,2D] -> [1,S,2D]
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t.contiguous())
        return torch.cat(parts, 1)

    with torch.no_grad():
        img = sh["img"][None]
        taps, psi = R.backbone(P, img, cfg, priors, flags, shard=(v0, gather_seq))
        cam = R.camera_head(P, taps, cfg, tok_gather=gather_tok)
        pts, conf = R.dpt_head(P, "pts_head.", taps, img, psi, cfg, "inv_log")
    q.put((rank, taps[3].numpy(), cam.numpy(), pts.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["tiny_2v_70x70_noprior", "tiny_12v_56x70_allpriors"])
def test_sharded_oracle_equals_unsharded(name):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cfg, views, flags, outs, z = load_golden(name)
    tap3 = np.concatenate([r[1] for r in res], 1)
    pts = np.concatenate([r[3] for r in res], 1)
    assert rel_l2(tap3, z["tap3"]) < 1e-5
    assert rel_l2(pts, outs["pts3d"]) < 1e-5
    for r in res:  # every rank holds the camera parameters of ALL views
        assert rel_l2(r[2], outs["camera_params"]) < 1e-5
