"""Experiment: the C2 workload (8 x 518^2) as W in-process virtual ranks of 8 / W views each — W handles sharing one copy of the weights,
one host thread and one HIP stream per rank, the library's in-process K/V gather (wm_local_group) — against the single-handle forward.
Two queues de-phase the kernels: one rank's HBM-bound epilogues and partly filled last rounds run beside the other's MFMA phases.
usage: python tools/bench_virtual_ranks.py [views=8]"""
import ctypes as C, json, sys, threading, time
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import WorldMirror, WMConfig, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = torch.Generator().manual_seed(1234)
img = torch.rand(1, S, 3, 518, 518, generator=g).to(dev)
owner = WorldMirror(arch=WMConfig(), dtype="bf16").to(dev).init_synthetic_weights()
owner.reserve(S, S, 518, 518)

def timed(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

nf = 10 if S <= 8 else 3
res = {"single_handle": [round(timed(lambda: owner({"img": img}), nf), 2) for _ in range(3)]}
ref = owner({"img": img}); torch.cuda.synchronize()
for world in (2, 4):
    if S % world: continue
    per = S // world
    grp = C.c_void_p(L.wm_local_group_create(world))
    models = [WorldMirror(arch=WMConfig()).to(dev).share_weights_from(owner).shard_local(grp, r, world) for r in range(world)]
    for mm in models: mm.reserve(per, S, 518, 518)
    streams = [torch.cuda.Stream() for _ in range(world)]
    outs = [None] * world
    def forward_all():
        errs = []
        def run(r):
            try:
                torch.cuda.set_device(0)
                with torch.cuda.stream(streams[r]):
                    outs[r] = models[r]({"img": img})
            except Exception as e:
                errs.append(e)
        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in th: t.start()
        for t in th: t.join()
        assert not errs, errs
    res[f"{world}_virtual_ranks"] = [round(timed(forward_all, nf), 2) for _ in range(3)]
    forward_all(); torch.cuda.synchronize()
    pts = torch.cat([outs[r]["pts3d"] for r in range(world)], 1)
    res[f"{world}_ranks_pts3d_rel_vs_single"] = float((pts - ref["pts3d"]).norm() / ref["pts3d"].norm())
    del models
    L.wm_local_group_destroy(grp)
print(json.dumps({"views": S, "ms_per_forward": res}), flush=True)
