"""bench.py's bookkeeping, checked without a GPU: every timing kind has algorithmic FLOPs, the per-kernel rows add up to their
classes, the totals are BASELINE.md's (42 TFLOP per 8-view forward at 518 x 518, 7.94 TFLOP per cross-view attention launch at
32 views), and the kinds named here are the kinds include/wm_hip.h documents."""
import importlib.util
import os
import re

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def test_flop_model_rows_add_up_to_their_classes():
    from hunyuanworld_mirror_amd import WMConfig
    b = _bench()
    cfg = WMConfig()
    for n_local, n_total in ((8, 8), (32, 32), (8, 64)):
        fl = b.flop_model(cfg, n_local, n_total, 518, 518)
        for name in b.KINDS.values():
            assert name in fl and fl[name] >= 0, name
        assert abs(sum(v for k, v in fl.items() if k.startswith("gemm_")) - fl["gemm"]) < 1e-6 * fl["gemm"]
        assert abs(sum(v for k, v in fl.items() if k.startswith("dpt_") and k != "dpt_conv") - fl["dpt_conv"]) < 1e-6 * fl["dpt_conv"]
        parts = fl["gemm"] + fl["dpt_conv"] + fl["global_attention"] + fl["frame_dino_attention"]
        assert parts <= fl["total"] <= parts * 1.01
    c2 = b.flop_model(cfg, 8, 8, 518, 518)
    assert abs(c2["total"] / 1e12 - 42.0) < 0.5
    c3 = b.flop_model(cfg, 32, 32, 518, 518)
    assert abs(c3["global_attention"] / 24 / 1e12 - 7.94) < 0.02     # per launch at 32 views
    c4 = b.flop_model(cfg, 8, 64, 518, 518)
    assert abs(c4["global_attention"] / c2["global_attention"] - 8.0) < 1e-9   # 8 x the keys per rank at 8 ranks


def test_timing_kinds_match_the_header():
    b = _bench()
    hdr = open(os.path.join(ROOT, "include", "wm_hip.h")).read()
    doc = hdr[hdr.index("/* kind:"):hdr.index("wm_status wm_profile_enable")]
    documented = {int(x) for x in re.findall(r"(\d+)(?: / (\d+))? =", doc) for x in x if x}
    assert set(b.KINDS) | {4} <= documented, (sorted(b.KINDS), sorted(documented))
    assert b.PEAK_TFLOPS == 2500.0
