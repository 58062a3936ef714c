"""Build container only (imports /root/reference): where does the recipe's rounding error come from, and is the
rounding-emulated oracle a fair stand-in for the reference's own reduced-precision recipe?

For a fixture, three things are compared with the reference's fp32 CPU outputs (the committed golden):
  (a) the REFERENCE ITSELF under bf16 autocast — its own GPU recipe (visual_transformer.py:275,309: backbone under
      torch.amp.autocast(bf16), heads with autocast disabled, worldmirror.py:146), emulated on the CPU by mapping the
      'cuda' autocast contexts to 'cpu' ones (the same emulation SURVEY §7 used for its 3.0e-4 figure);
  (b) this repository's oracle in emulate=("bf16", "f16") mode = what the HIP build computes, up to fp32 summation order;
  (c) for reference, (a) vs (b).
Output: one markdown table row per fixture and output (profiles/r02_emulation_validation.md is this script's stdout).

    PYTHONDONTWRITEBYTECODE=1 python oracle/validate_emulation.py [fixture ...]
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import golden_preset, load_golden, rel_l2, torch_weights  # noqa: E402
import oracle.gen_golden as G  # noqa: E402  (imports the reference, installs the 2-kwarg shim)
from oracle import worldmirror_ref as R  # noqa: E402

_real_autocast = torch.amp.autocast


class _cpu_autocast(_real_autocast):
    """torch.amp.autocast('cuda', ...) -> the same context for the CPU backend"""

    def __init__(self, device_type, *a, **k):
        super().__init__("cpu" if device_type == "cuda" else device_type, *a, **k)


def reference_autocast(cfg, views_np, flags, preset):
    m = G.build_reference(cfg, preset)
    views = {k: torch.from_numpy(v.copy()) for k, v in views_np.items()}
    torch.amp.autocast = _cpu_autocast
    try:
        with torch.no_grad():
            if sum(flags) > 0:
                pri = m.extract_priors(views)
                taps, psi = m.visual_geometry_transformer(views["img"], pri, cond_flags=flags)
            else:
                taps, psi = m.visual_geometry_transformer(views["img"])
            taps = [t.float() for t in taps]
            with torch.amp.autocast("cuda", enabled=False):  # worldmirror.py:146
                preds = m._gen_all_preds(taps, views["img"], psi, views)
    finally:
        torch.amp.autocast = _real_autocast
    return {k: v.float().numpy() for k, v in preds.items() if isinstance(v, torch.Tensor)}


def main():
    names = sys.argv[1:] or ["tiny_3v_70x56_pose_ray", "refinit_tiny_3v_70x56_pose_ray", "full_2v_224_noprior", "refinit_full_2v_224_noprior"]
    print("| fixture (weights preset) | output | reference bf16-autocast vs reference fp32 | emulated oracle (bf16, f16) vs reference fp32 | autocast vs emulated |")
    print("|---|---|---|---|---|")
    for name in names:
        cfg, views, flags, outs, z = load_golden(name)
        preset, sub, H = golden_preset(z), int(z["subsample"]), views["img"].shape[-2]
        ac = reference_autocast(cfg, views, flags, preset)
        with torch.no_grad():
            em = R.forward(torch_weights(cfg, preset), {k: torch.from_numpy(v) for k, v in views.items()}, flags, cfg,
                           emulate=("bf16", "f16"), prune=False)
        em = {k: v.numpy() for k, v in em.items() if isinstance(v, torch.Tensor)}
        for k in ("pts3d", "depth", "normals", "camera_params"):
            def s(a):
                return a[:, :, ::sub, ::sub] if sub > 1 and a.ndim >= 4 and a.shape[2] == H else a
            a, e, ref = s(ac[k]), s(em[k]), outs[k]
            print(f"| {name} ({preset}) | {k} | {rel_l2(a, ref):.2e} | {rel_l2(e, ref):.2e} | {rel_l2(a, e):.2e} |", flush=True)


if __name__ == "__main__":
    main()
