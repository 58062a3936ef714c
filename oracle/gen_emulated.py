"""Precompute the rounding-emulated oracle outputs for the benchmark-size fixtures (tests/golden/emu_<name>.npz).

    python oracle/gen_emulated.py [name ...]

Uses only this repository's own code (oracle/worldmirror_ref.py in emulate=("bf16", "f16") mode + the name-keyed synthetic
weights): no reference import, so it can run anywhere; it is precomputed only because the CPU oracle needs minutes at
8 x 518 x 518.  tests/test_gpu_emulated.py compares the HIP build with these (kernel error <= 5e-4) and with the
reference's own outputs in tests/golden/<name>.npz; tests/test_oracle_golden.py re-derives a small one to check that
the stored file is what this script produces.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import GOLD, golden_preset, load_golden, torch_weights  # noqa: E402
from oracle import worldmirror_ref as R  # noqa: E402

DEFAULT = ["full_8v_518_noprior", "full_4v_518_pose_ray", "refinit_full_8v_518_noprior"]


def run(name: str, emulate=("bf16", "f16")):
    cfg, views, flags, outs, z = load_golden(name)
    preset, sub, H = golden_preset(z), int(z["subsample"]), views["img"].shape[-2]
    P = torch_weights(cfg, preset)
    t0 = time.time()
    with torch.no_grad():
        o = R.forward(P, {k: torch.from_numpy(v) for k, v in views.items()}, flags, cfg, emulate=emulate, prune=False)
    store = {"emulate": np.array(",".join(emulate)), "weights_preset": np.array(preset), "subsample": np.array(sub)}
    for k, v in o.items():
        if not isinstance(v, torch.Tensor):
            continue
        v = v.numpy()
        store["sum_" + k] = np.array(np.nan_to_num(v.astype(np.float64), posinf=0, neginf=0).sum())
        if sub > 1 and v.ndim >= 4 and v.shape[2] == H:
            v = v[:, :, ::sub, ::sub]
        store["out_" + k] = np.ascontiguousarray(v)
    path = os.path.join(GOLD, "emu_" + name + ".npz")
    np.savez_compressed(path, **store)
    err = {k: float(np.linalg.norm(store["out_" + k].astype(np.float64) - outs[k]) / np.linalg.norm(outs[k]))
           for k in ("pts3d", "depth", "normals", "camera_params") if k in outs}
    print(name, f"{time.time() - t0:.0f} s", "emulated vs reference:", {k: f"{e:.2e}" for k, e in err.items()}, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    for n in (sys.argv[1:] or DEFAULT):
        run(n)
