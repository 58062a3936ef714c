import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Returns (cfg, views(np), cond_flags, expected outputs(np), raw npz dict)."""
    from hunyuanworld_mirror_amd.config import WMConfig
    z = dict(np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False))
    d = json.loads(str(z["cfg_json"]))
    d["intermediate_idxs"] = tuple(d["intermediate_idxs"])
    d["dpt_out_channels"] = tuple(d["dpt_out_channels"])
    cfg = WMConfig(**d)
    views = {k[3:]: v for k, v in z.items() if k.startswith("in_")}
    if "regen_img" in z:  # benchmark-size fixtures: the image is drawn from its seed as bench.py draws it (oracle/gen_golden.py make_inputs_518)
        import torch
        g = json.loads(str(z["regen_img"]))
        assert g["kind"] == "torch_rand"
        gen = torch.Generator().manual_seed(int(g["seed"]))
        img = torch.rand(*g["shape"], generator=gen).numpy()
        assert abs(float(img.astype(np.float64).sum()) - float(z["sum_in_img"])) < 1e-6 * img.size, "regenerated image differs from the fixture's"
        views["img"] = img
        if "depth_seed" in g:  # depth prior drawn as oracle/gen_golden.py make_depth_518 draws it
            gd = torch.Generator().manual_seed(int(g["depth_seed"]))
            S, H, W = g["shape"][1], g["shape"][3], g["shape"][4]
            dep = 0.5 + 4.0 * torch.rand(1, S, H, W, generator=gd)
            dep[torch.rand(1, S, H, W, generator=gd) < 0.05] = 0.0
            dep = dep.numpy()
            assert abs(float(dep.astype(np.float64).sum()) - float(z["sum_in_depthmap"])) < 1e-6 * dep.size, "regenerated depth prior differs from the fixture's"
            views["depthmap"] = dep
    outs = {k[4:]: v for k, v in z.items() if k.startswith("out_")}
    return cfg, views, [int(x) for x in z["cond_flags"]], outs, z


def golden_preset(z):
    return str(z["weights_preset"]) if "weights_preset" in z else "sensitive"


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


_WCACHE = {}


def torch_weights(cfg, preset="sensitive"):
    """Synthetic name-keyed weights as torch fp32 tensors (cached per config)."""
    import torch
    from hunyuanworld_mirror_amd.weights import iter_params
    key = json.dumps(cfg.to_dict(), sort_keys=True) + preset
    if key not in _WCACHE:
        _WCACHE.clear()
        _WCACHE[key] = {k: torch.from_numpy(v) for k, v in iter_params(cfg, 0, preset)}
    return _WCACHE[key]
