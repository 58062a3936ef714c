"""Same-box A/B of the head phase: camera head + DPT heads on their own queues (default) against one queue (tuning heads_concurrent = 0);
C2 and C3 workloads, interleaved rounds of 10 / 3 forwards.  usage: python tools/bench_heads_concurrency.py"""
import json, sys, time
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import WorldMirror, WMConfig, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
m = WorldMirror(arch=WMConfig(), dtype="bf16").to(dev).init_synthetic_weights()
g = torch.Generator().manual_seed(1234)
for label, S, nf in (("C2 8 views", 8, 10), ("C3-size 32 views (no priors)", 32, 3)):
    v = {"img": torch.rand(1, S, 3, 518, 518, generator=g).to(dev)}
    m.reserve(S, S, 518, 518)
    res = {}
    for rep in range(3):
        for mode in (1, 0):
            L.wm_set_tuning(b"heads_concurrent", mode)
            m(v); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(nf): m(v)
            torch.cuda.synchronize()
            res.setdefault("own_queues" if mode else "one_queue", []).append(round((time.perf_counter() - t0) / nf * 1e3, 2))
    L.wm_set_tuning(b"heads_concurrent", -1)
    print(json.dumps({"workload": label, "ms_per_forward": res}), flush=True)
