"""Concurrent heads against the serial forward, many times, bit for bit — in ONE process, switching with the tuning key
(heads_concurrent = 1 | 0), on the three workloads that matter: C2 (8 x 518^2 bf16, 4 heads), the C5 flag set on one rank
(f16 + Gaussian head, VIEWS5 views) and C3 (32 views, priors).  usage: python tools/stress_concurrent_heads.py [N2=150] [N5=20] [N3=10]"""
import ctypes as C, hashlib, json, sys, time
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import WorldMirror, WMConfig, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
N2, N5, N3 = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 150), (2, 20), (3, 10)))


def digest(out):
    h = hashlib.sha256()
    for k in sorted(out):
        v = out[k]
        if isinstance(v, torch.Tensor):
            h.update(k.encode()); h.update(v.detach().contiguous().cpu().numpy().tobytes())
        elif isinstance(v, dict):
            for kk in sorted(v):
                vv = v[kk]
                vv = vv[0] if isinstance(vv, (list, tuple)) else vv
                if isinstance(vv, torch.Tensor):
                    h.update(kk.encode()); h.update(vv.detach().contiguous().cpu().numpy().tobytes())
    return h.hexdigest()


def run(label, m, views, flags, n):
    m.reserve(views["img"].shape[1], views["img"].shape[1], views["img"].shape[-2], views["img"].shape[-1])
    assert L.wm_set_tuning(b"heads_concurrent", 0) == 0
    ref = digest(m(views, flags)); torch.cuda.synchronize()
    assert digest(m(views, flags)) == ref, "serial forward not deterministic"
    assert L.wm_set_tuning(b"heads_concurrent", 1) == 0
    bad, t0 = 0, time.time()
    for i in range(n):
        d = digest(m(views, flags))
        bad += d != ref
    L.wm_set_tuning(b"heads_concurrent", -1)
    print(json.dumps({"workload": label, "concurrent_forwards": n, "mismatching": bad, "seconds": round(time.time() - t0, 1)}), flush=True)
    return bad


g = torch.Generator().manual_seed(1234)
total = 0
m = WorldMirror(arch=WMConfig(), dtype="bf16").to(dev).init_synthetic_weights()
total += run("C2: 8 x 518^2 bf16, 4 heads", m, {"img": torch.rand(1, 8, 3, 518, 518, generator=g).to(dev)}, [0, 0, 0], N2)
v3 = {"img": torch.rand(1, 32, 3, 518, 518, generator=g).to(dev)}
pose = torch.eye(4).repeat(1, 32, 1, 1); pose[0, :, 0, 3] = 0.1 * torch.arange(32)
K = torch.zeros(1, 32, 3, 3); K[..., 0, 0] = 518; K[..., 1, 1] = 518; K[..., 0, 2] = 259; K[..., 1, 2] = 259; K[..., 2, 2] = 1
v3["camera_pose"] = pose.to(dev); v3["camera_intrinsics"] = K.to(dev)
total += run("C3: 32 x 518^2 bf16, pose + intrinsics priors", m, v3, [1, 0, 1], N3)
del m, v3
cfg5 = WMConfig(enable_gs=True)
m5 = WorldMirror(arch=cfg5, dtype="f16").to(dev).init_synthetic_weights()
m5.enable_prune = False
total += run("C5 flag set on one rank: 8 x 518^2 f16 + Gaussian head", m5, {"img": torch.rand(1, 8, 3, 518, 518, generator=g).to(dev)}, [0, 0, 0], N5)
print(json.dumps({"total_mismatching": total}))
sys.exit(1 if total else 0)
