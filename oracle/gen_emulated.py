"""Precompute the rounding-emulated oracle outputs for the benchmark-size fixtures (tests/golden/emu_<name>.npz).

    python oracle/gen_emulated.py [--f16] [name ...]

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
from test_gpu_emulated import PERTURB, perturbed  # noqa: E402  (one definition of the perturbation for script and test)

DEFAULT = ["full_8v_518_noprior", "full_4v_518_pose_ray", "refinit_full_8v_518_noprior", "full_2v_518_allpriors"]


def run(name: str, emulate=("bf16", "f16")):
    cfg, views, flags, outs, z = load_golden(name)
    preset, sub, H = golden_preset(z), int(z["subsample"]), views["img"].shape[-2]
    P = torch_weights(cfg, preset)
    t0 = time.time()
    store = {"emulate": np.array(",".join(emulate)), "weights_preset": np.array(preset), "subsample": np.array(sub), "perturb": np.array(PERTURB)}
    # "out_": the fixture's input; "pert_": the image multiplied by (1 + 1e-7 N(0,1)) — the self-decorrelation floor of
    # rounded arithmetic (tests/test_gpu_emulated.py)
    for tag, vw in (("out_", views), ("pert_", perturbed(views))):
        with torch.no_grad():
            o = R.forward(P, {k: torch.from_numpy(v) for k, v in vw.items()}, flags, cfg, emulate=emulate, prune=False)
        for k, v in o.items():
            if not isinstance(v, torch.Tensor):
                continue
            v = v.numpy()
            if tag == "out_":
                store["sum_" + k] = np.array(np.nan_to_num(v.astype(np.float64), posinf=0, neginf=0).sum())
            if sub > 1 and v.ndim >= 4 and v.shape[2] == H:
                v = v[:, :, ::sub, ::sub]
            store[tag + k] = np.ascontiguousarray(v)
        del o
    path = os.path.join(GOLD, ("emu_" if emulate[0] == "bf16" else "emu_" + emulate[0] + "_") + name + ".npz")
    np.savez_compressed(path, **store)
    def rl(a, b):
        return float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
    ks = [k for k in ("pts3d", "depth", "normals", "camera_params") if k in outs]
    print(name, f"{time.time() - t0:.0f} s", "emulated vs reference:", {k: f"{rl(store['out_' + k], outs[k]):.2e}" for k in ks},
          "floor (emulated vs perturbed):", {k: f"{rl(store['pert_' + k], store['out_' + k]):.2e}" for k in ks}, os.path.getsize(path) // 1024, "KiB", flush=True)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--f16"]
    f16 = "--f16" in sys.argv[1:]   # backbone operands in f16 (BASELINE config 5's dtype): tests/golden/emu_f16_<name>.npz
    for n in (args or DEFAULT):
        run(n, ("f16", "f16") if f16 else ("bf16", "f16"))
