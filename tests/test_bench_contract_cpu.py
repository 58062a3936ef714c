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


def test_two_pass_gather_decision_logic():
    """bench.two_pass_overlap (N > 1): both modes are timed, one forward of each is compared, and the overlapped mode is headlined only
    when it is faster AND agrees with the serial one within the bound; a disagreement or a non-finite output falls back to the serial
    gather whatever the timing says.  Pure host logic: fake forwards on the CPU."""
    import torch
    b = _bench()

    def run(ms0, ms1, delta, nan=False):
        state = {"mode": None, "set": []}
        base = {"pts3d": torch.arange(12.0).reshape(1, 1, 2, 2, 3) + 1, "depth": torch.ones(1, 1, 2, 2, 1), "camera_params": torch.ones(1, 1, 9)}

        def set_overlap(v):
            state["mode"] = v
            state["set"].append(v)

        def fwd():
            o = {k: v.clone() for k, v in base.items()}
            if state["mode"] == 1:
                o["pts3d"] = o["pts3d"] * (1.0 + delta)
                if nan:
                    o["depth"][0, 0, 0, 0, 0] = float("nan")
            return o
        pick, ms, info = b.two_pass_overlap(set_overlap, fwd, lambda: (ms0 if state["mode"] == 0 else ms1), lambda x: x)
        assert state["set"][:2] == [0, 1] and state["set"][-1] == pick      # serial first; the tuning is left at the choice
        assert ms == (ms0, ms1)[pick] and info["chosen_comm_overlap"] == pick
        return pick, info

    assert run(10.0, 9.0, 1e-4)[0] == 1                 # faster and within the bound
    assert run(10.0, 11.0, 1e-4)[0] == 0                # slower
    pick, info = run(10.0, 5.0, 1e-2)                   # faster but disagrees: not headlined
    assert pick == 0 and not info["comparison_passed"]
    pick, info = run(10.0, 5.0, 0.0, nan=True)          # non-finite output
    assert pick == 0 and not info["comparison_passed"]
