"""Same-box, same-process A/B of the whole forward under wm_set_tuning arms, interleaved rounds (guides rule 24).
usage: python tools/ab_forward.py [views] arm [arm ...]     arm = name:key=val,key=val   (an arm with no keys = the defaults)
e.g.   python tools/ab_forward.py 8 default: r03gemm:gemm_pp=2,gemm_sched=0"""
import json, sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import WorldMirror, WMConfig, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
args = sys.argv[1:]
S = int(args.pop(0)) if args and args[0].isdigit() else 8
arms = []
for a in args:
    name, _, kv = a.partition(":")
    arms.append((name, [(k.encode(), int(v)) for k, v in (x.split("=") for x in kv.split(",") if x)]))
keys = sorted({k for _, kvs in arms for k, _ in kvs})
import os
GS, DT = os.environ.get("AB_GS") == "1", os.environ.get("AB_DTYPE", "bf16")   # (AB_GS=1 AB_DTYPE=f16: BASELINE C5's flag set)
m = WorldMirror(arch=WMConfig(enable_gs=True) if GS else WMConfig(), dtype=DT).to(dev).init_synthetic_weights()
if GS:
    m.enable_prune = False
g = torch.Generator().manual_seed(1234)
v = {"img": torch.rand(1, S, 3, 518, 518, generator=g).to(dev)}
m.reserve(S, S, 518, 518)
nf = 10 if S <= 8 else 3
res = {n: [] for n, _ in arms}
for rep in range(5):
    for name, kvs in arms:
        for k in keys: L.wm_set_tuning(k, -1)
        for k, val in kvs: assert L.wm_set_tuning(k, val) == 0
        m(v); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nf): m(v)
        torch.cuda.synchronize()
        res[name].append(round((time.perf_counter() - t0) / nf * 1e3, 3))
for k in keys: L.wm_set_tuning(k, -1)
print(json.dumps({"views": S, "ms_per_forward": res, "median": {n: float(np.median(x)) for n, x in res.items()}}), flush=True)
