#!/usr/bin/env python
"""Headline benchmark: views/sec of the WorldMirror forward pass on MI355X (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (synthetic, random-init weights of the full 1.23 B-parameter architecture):
  N = 1 : BASELINE config C2 — 8 views x 518 x 518, bf16, no priors, camera + depth + point + normal heads.
  N > 1 : 8 views per GPU, view-sharded, K/V of every global-attention layer all-gathered over RCCL
          (N = 8 is BASELINE config C4: 64 views) -> "scaling": "weak" (views per GPU fixed).
A step = one full forward of the whole job; value = total views / step time (inputs resident in HBM).
The JSON line also carries `roofline` (dominant kernel class, HIP-event timed on the launch stream) and,
at N = 1, `cpu_baseline` (the CPU oracle timed on the host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = 2500.0   # dense bf16/f16 MFMA, guides/MI355X_MICROARCH.md "Chip-level parameters"
# timing kinds of wm_profile_read (include/wm_hip.h): one per kernel instantiation that matters; "gemm" in the class table is
# the sum of the four gemm_* rows
KINDS = {0: "global_attention", 1: "frame_dino_attention", 5: "gemm_qkv", 6: "gemm_proj_fc2", 7: "gemm_fc1", 2: "gemm_other",
         8: "dpt_conv3x3_level4x", 9: "dpt_conv3x3_level2x", 10: "dpt_output_conv1_up", 11: "dpt_output_conv2_tail", 3: "dpt_conv_other"}


def flop_model(cfg, n_local, n_total, H, W, heads=3):
    """Algorithmic FLOPs per rank (2*MAC), per kernel class — BASELINE.md §3 formulas."""
    ps, D = cfg.patch_size, cfg.embed_dim
    hw = (H // ps) * (W // ps)
    Td, P = 1 + cfg.num_register_tokens + hw, cfg.patch_start_idx + hw
    lin_tok = 2 * D * (3 * D + D + 2 * cfg.mlp_ratio * D)
    toks = cfg.dino_depth * n_local * Td + 2 * cfg.depth * n_local * P
    gemm = toks * lin_tok
    g_other = 2 * n_local * hw * (3 * ps * ps) * D        # patch embedding (the DPT projections are counted under dpt_conv)
    gemm += g_other
    attn_local = cfg.dino_depth * n_local * 4 * Td * Td * D + cfg.depth * n_local * 4 * P * P * D
    attn_global = cfg.depth * 4 * (n_local * P) * (n_total * P) * D
    # DPT head: per view, scaled from the 518-px figure of BASELINE.md (298.6 GF) by pixel count
    dpt = heads * 298.6e9 * (H * W) / (518.0 * 518.0) * n_local if D == 1024 else 0.0
    F = cfg.dpt_features
    gh, gw = H // ps, W // ps
    c33 = lambda px, ci, co: 2.0 * px * ci * co * 9
    # per head: layer_rn + 2 RCUs (4 convs) on each of the two large levels; output_conv1 at 8x; output_conv2[0] + the 1x1 tail at full size
    d4 = heads * n_local * 5 * c33(16 * gh * gw, F, F)
    d2 = heads * n_local * 5 * c33(4 * gh * gw, F, F)
    d_up = heads * n_local * c33(64 * gh * gw, F, F // 2)
    d_tail = heads * n_local * (c33(H * W, F // 2, 32) + 2.0 * H * W * 32 * 4)
    return {"dpt_conv3x3_level4x": d4, "dpt_conv3x3_level2x": d2, "dpt_output_conv1_up": d_up, "dpt_output_conv2_tail": d_tail,
            "dpt_conv_other": max(dpt - d4 - d2 - d_up - d_tail, 0.0),
            "gemm": gemm, "gemm_qkv": toks * 2 * D * 3 * D, "gemm_proj_fc2": toks * 2 * D * (D + cfg.mlp_ratio * D),
            "gemm_fc1": toks * 2 * D * cfg.mlp_ratio * D, "gemm_other": g_other, "frame_dino_attention": attn_local, "global_attention": attn_global, "dpt_conv": dpt,
            "total": gemm + attn_local + attn_global + dpt + 1.6e9 * n_total}


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def host_cores():
    """CPU share actually usable: affinity mask, capped at the GPU box's per-GPU share (16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("WM_BENCH_CORES", "16"))))


def cpu_baseline(cfg, H, W, budget_views=8):
    """CPU oracle (oracle/worldmirror_ref.py, a port of the reference's fp32 path) on the host cores, on the SAME workload
    as the GPU line at N = 1 (BASELINE C2: all 8 views, so the cross-view attention over 11 008 keys is included);
    `--cpu-views 1` gives the cheaper 1-view sample (about 8 s instead of about 70 s on 16 cores)."""
    from hunyuanworld_mirror_amd.weights import iter_params
    from oracle import worldmirror_ref as R
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: generating weights, {cores} threads")
    P = {k: torch.from_numpy(v) for k, v in iter_params(cfg)}
    log("cpu_baseline: running the oracle")
    g = torch.Generator().manual_seed(1234)
    img = torch.rand(1, budget_views, 3, H, W, generator=g)
    t0 = time.time()
    with torch.no_grad():
        R.forward(P, {"img": img}, (0, 0, 0), cfg)
    dt = time.time() - t0
    del P
    return {"value": budget_views / dt, "unit": "views/s", "cores": cores, "kind": "port",
            "sample": f"{budget_views} view(s) x {H}x{W}, full architecture, fp32 torch-CPU oracle, {dt:.1f} s"}


def fixture_parity(m, dev):
    """Parity of THE BUILD THAT WAS JUST TIMED against the reference itself, so that the headline number and its tolerance travel
    together: the committed fixture tests/golden/refinit_full_8v_518_noprior.npz holds the REFERENCE's fp32 outputs (written by
    oracle/gen_golden.py from /root/reference, every 8th pixel + fp64 checksums) for BASELINE C2's own input (seed 1234) on the
    "refinit" weight preset (the reference's init statistics).  The model's weights are swapped for that preset on the live handle,
    one forward is run, and rel-L2 against the reference is reported for the dense outputs (north_star: point maps < 1e-3).  Data
    only is read here (no oracle, no reference code)."""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "refinit_full_8v_518_noprior.npz")
    if not os.path.exists(path):
        return None
    z = np.load(path, allow_pickle=False)
    g = json.loads(str(z["regen_img"]))
    img = torch.rand(*g["shape"], generator=torch.Generator().manual_seed(int(g["seed"])))
    if abs(float(img.double().sum()) - float(z["sum_in_img"])) > 1e-6 * img.numel():
        return {"error": "regenerated input differs from the fixture's"}
    log("parity leg: refinit weights")
    m.init_synthetic_weights(preset="refinit")
    out = m({"img": img.to(dev)}, [0, 0, 0])
    torch.cuda.synchronize(dev)
    sub = int(z["subsample"])
    res = {}
    for k in ("pts3d", "depth", "normals"):
        got = out[k].cpu().numpy()[:, :, ::sub, ::sub].astype(np.float64)
        ref = z["out_" + k].astype(np.float64)
        res[k] = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    return {"fixture": "tests/golden/refinit_full_8v_518_noprior.npz (reference fp32 outputs on this workload's input, refinit weights)",
            "rel_l2_vs_reference": {k: float(f"{v:.3e}") for k, v in res.items()}, "tolerance": 1e-3,
            "within_tolerance": all(v < 1e-3 for v in res.values())}


def two_pass_overlap(set_overlap, forward, timed, allreduce_max, profile_once=None, bound=2.5e-3,
                     keys=("pts3d", "depth", "normals", "camera_params")):
    """N > 1: let the run decide by itself whether the K|V all-gather of the global layers goes on the compute queue (comm_overlap 0:
    24 exposed gathers, one attention launch per layer) or on the communication queue under the attention over the local keys
    (comm_overlap 1: three partial launches + a combine pass per layer).  No multi-GPU node was available to any build round, so
    neither form has a measured time on real links: both are timed here, one forward of each is compared on every rank (bound: the
    key-partition-order floor of the rounded arithmetic, tests/test_gpu_fullsize.py::test_c4_shapes_eight_virtual_ranks), and the
    faster one is used for the headline ONLY if the comparison passed on every rank; otherwise the serial gather.
    set_overlap(v): tuning comm_overlap; forward() -> outputs of one forward; timed() -> this rank's ms per step of the timed region;
    allreduce_max(x) -> max over ranks; profile_once() -> dict of per-rank event times of one forward in the current mode."""
    res = {}
    for ov in (0, 1):
        set_overlap(ov)
        forward()                                   # first forward in this mode (queues / events are created lazily)
        ms = float(allreduce_max(timed()))
        out = forward()
        res[ov] = {"ms": ms, "out": {k: out[k].float().clone() for k in keys if k in out},
                   "profile": profile_once() if profile_once else None}
        # on record (stderr) before the next mode is attempted: the overlapped form has never run on real links
        log(f"two-pass gather decision: comm_overlap {ov}: {ms:.3f} ms/step (max over ranks)")
    err, finite = 0.0, True
    for k, a in res[0]["out"].items():
        b = res[1]["out"][k]
        finite = finite and bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all())
        err = max(err, float((a.double() - b.double()).norm() / b.double().norm().clamp(min=1e-30)))
    err = float(allreduce_max(err))
    ok = float(allreduce_max(0.0 if finite else 1.0)) == 0.0 and err < bound
    pick = 1 if (ok and res[1]["ms"] < res[0]["ms"]) else 0
    set_overlap(pick)
    info = {"ms_per_step": {"comm_overlap_0": round(res[0]["ms"], 3), "comm_overlap_1": round(res[1]["ms"], 3)},
            "max_rel_l2_between_modes_over_ranks": float(f"{err:.3e}"), "bound": bound, "comparison_passed": ok, "chosen_comm_overlap": pick,
            "per_rank_events": {"comm_overlap_0": res[0]["profile"], "comm_overlap_1": res[1]["profile"]}}
    return pick, res[pick]["ms"], info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--views-per-gpu", type=int, default=8)
    ap.add_argument("--size", type=int, default=518)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--priors", action="store_true",
                    help="camera-pose + intrinsics priors on (BASELINE config C3: --views-per-gpu 32 --priors), cond_flags [1, 0, 1]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-views", type=int, default=8, help="views of the CPU-baseline sample (8 = the C2 workload itself)")
    ap.add_argument("--no-north-star", action="store_true", help="skip the C3 (32 views, priors) leg that follows the timed region at N = 1")
    ap.add_argument("--no-parity", action="store_true", help="skip the committed-fixture parity leg that follows the timed region at N = 1")
    ap.add_argument("--gather", default="allgather", choices=["allgather", "p2p"],
                    help="N > 1: the K|V exchange as ncclAllGather (default) or as one group of point-to-point sends / receives (direct over the xGMI mesh; opt-in, never run on hardware)")
    ap.add_argument("--tiny", action="store_true", help="scaled-down architecture (plumbing check only)")
    ap.add_argument("--gs", action="store_true", help="3D-Gaussian head on (BASELINE config C5's flag set with --dtype f16; rasterisation not run, voxel merge off)")
    a = ap.parse_args()

    from hunyuanworld_mirror_amd import WorldMirror, WMConfig
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    cfg = WMConfig.tiny() if a.tiny else WMConfig(enable_gs=True) if a.gs else WMConfig()
    H = W = a.size if not a.tiny else 70
    n_local, n_total = a.views_per_gpu, a.views_per_gpu * world
    torch.set_num_threads(host_cores())
    log("generating + uploading weights")
    m = WorldMirror(arch=cfg, dtype=a.dtype).to(dev).init_synthetic_weights()
    if a.gs:
        m.enable_prune = False   # the cross-view voxel merge is a caller's step (rasterization.py:301), not part of the forward's heads
    log("weights ready")
    if world > 1:
        m.shard()
        if a.gather == "p2p":
            from hunyuanworld_mirror_amd import _lib as _wl0
            assert _wl0.lib().wm_set_tuning(b"comm_p2p", 1) == 0
    def make_views(nv, priors):
        g = torch.Generator().manual_seed(1234)
        vw = {"img": torch.rand(1, nv, 3, H, W, generator=g).to(dev)}
        if not priors:
            return vw, [0, 0, 0]
        # SURVEY §8d synthetic priors: identity rotations, x = 0.1 i, fx = fy = W, principal point at the centre
        pose = torch.eye(4).repeat(1, nv, 1, 1)
        pose[0, :, 0, 3] = 0.1 * torch.arange(nv)
        K = torch.zeros(1, nv, 3, 3)
        K[..., 0, 0] = W; K[..., 1, 1] = H; K[..., 0, 2] = W / 2; K[..., 1, 2] = H / 2; K[..., 2, 2] = 1
        vw["camera_pose"] = pose.to(dev)
        vw["camera_intrinsics"] = K.to(dev)
        return vw, [1, 0, 1]

    views, flags = make_views(n_total, a.priors)
    m.reserve(n_local, n_total, H, W)  # workspace + tables: the forward itself never allocates

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for i in range(a.warmup):
        m(views, flags)
        torch.cuda.synchronize(dev)
        log(f"warmup {i} done")

    def timed():
        """EXACTLY a.steps forwards between two (barrier + synchronize) brackets; this rank's ms per step"""
        sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            m(views, flags)
        sync()
        return (time.perf_counter() - t0) / a.steps * 1e3

    overlap_info = None
    if world > 1:
        from hunyuanworld_mirror_amd import _lib as _wl

        def allmax(x):
            t = torch.tensor([float(x)], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        def prof_once():
            m.profile(True)
            m(views, flags)
            torch.cuda.synchronize(dev)
            ga, _ = m.profile_read(0); cm, cn = m.profile_read(12); wh, _ = m.profile_read(4)
            m.profile(False)
            mine = {"rank": rank, "global_attention_ms": round(ga, 3), "allgather_ms_on_its_queue": round(cm, 3), "allgather_calls": cn, "forward_ms_events": round(wh, 3)}
            allr = [None] * world
            dist.all_gather_object(allr, mine)
            return allr

        def fwd():
            o = m(views, flags)
            torch.cuda.synchronize(dev)
            return o
        pick, ms_step, overlap_info = two_pass_overlap(lambda v: _wl.lib().wm_set_tuning(b"comm_overlap", v), fwd, timed, allmax, prof_once)
        log(f"two-pass gather decision: {overlap_info['ms_per_step']} -> comm_overlap {pick} (comparison passed: {overlap_info['comparison_passed']})")
    else:
        ms_step = timed()
    log(f"timed region done: {ms_step:.2f} ms/step")

    # per-kernel-class timing: HIP events recorded by the library on the launch stream (one extra step)
    m.profile(True)
    m(views, flags)
    torch.cuda.synchronize(dev)
    fl = flop_model(cfg, n_local, n_total, H, W, heads=4 if a.gs else 3)
    classes = {}
    for k, name in KINDS.items():
        ms, n = m.profile_read(k)
        if n:
            classes[name] = {"ms_total": round(ms, 3), "launches": n, "avg_ms": round(ms / n, 4),
                             "tflops": round(fl[name] / (ms * 1e-3) / 1e12, 1) if ms > 0 else None}
    whole_ms, _ = m.profile_read(4)
    comm_ms, comm_n = m.profile_read(12)
    m.profile(False)
    # N > 1: what the first real multi-GPU run needs to explain itself — per rank the cross-view attention time and the time of
    # the collectives on the queue they ran on (exposed time when the gather ran on the compute queue: gather_decision.chosen_comm_overlap 0)
    multi = None
    if world > 1:
        mine = {"rank": rank, "global_attention_ms": classes.get("global_attention", {}).get("ms_total"), "allgather_ms": round(comm_ms, 3),
                "allgather_calls": comm_n, "forward_ms_events": round(whole_ms, 3)}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        multi = {"per_rank": allr, "gather_decision": overlap_info, "rccl_world": m._comm[1] if m._comm else 1,
                 "allgather_bytes_per_rank_per_layer": 2 * 16 * n_local * (cfg.patch_start_idx + (H // cfg.patch_size) * (W // cfg.patch_size)) * 64 * 2,
                 "note": "allgather_ms = sum over the K|V gathers of the 24 global layers + the camera-token gather, HIP events on the queue the collective ran on; "
                         "with the gather on the compute queue it is exposed time.  No scaling efficiency is computed here."}
    kernels = dict(classes)                 # per kernel instantiation
    for cls, pre in (("gemm", "gemm_"), ("dpt_conv", "dpt_")):   # the classes as a whole (r01's rows), beside their kernels
        gk = [k for k in classes if k.startswith(pre)]
        if gk:
            gms, gn = sum(classes[k]["ms_total"] for k in gk), sum(classes[k]["launches"] for k in gk)
            classes = {k: v for k, v in classes.items() if not k.startswith(pre)}
            classes[cls] = {"ms_total": round(gms, 3), "launches": gn, "avg_ms": round(gms / gn, 4),
                            "tflops": round(fl[cls] / (gms * 1e-3) / 1e12, 1)}

    # North-star leg (N = 1 only, after and outside the timed region): BASELINE C3 = 32 views x 518 x 518, camera-pose +
    # intrinsics priors, same weights; 1 warm-up + 3 timed steps + 1 HIP-event step for the cross-view attention class.
    north = None
    if world == 1 and not a.no_north_star and not a.tiny and not (n_local == 32 and a.priors):
        v3, f3 = make_views(32, True)
        m.reserve(32, 32, H, W)
        m(v3, f3)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(3):
            m(v3, f3)
        torch.cuda.synchronize(dev)
        ms3 = (time.perf_counter() - t1) / 3 * 1e3
        m.profile(True)
        m(v3, f3)
        torch.cuda.synchronize(dev)
        ga_ms, ga_n = m.profile_read(0)
        m.profile(False)
        fl3 = flop_model(cfg, 32, 32, H, W)
        north = {"config": "C3: 32-view 518x518 bf16, camera-pose + intrinsics priors, 1 GPU", "steps": 3, "ms_per_step": round(ms3, 3),
                 "views_per_s": round(32 / (ms3 * 1e-3), 2), "whole_forward_tflops": round(fl3["total"] / (ms3 * 1e-3) / 1e12, 1),
                 "global_attention": {"ms_total": round(ga_ms, 3), "launches": ga_n, "tflop_total": round(fl3["global_attention"] / 1e12, 2),
                                      "tflops": round(fl3["global_attention"] / (ga_ms * 1e-3) / 1e12, 1),
                                      "frac": round(fl3["global_attention"] / (ga_ms * 1e-3) / 1e12 / PEAK_TFLOPS, 4),
                                      "target_frac": 0.6,
                                      "ceiling_frac": 0.56, "ceiling_source": "profiles/r03_attention_ceiling.md (register-only loop of the hd = 64 instruction mix: 38.5 cycles per MFMA at the 1.66 GHz the board's power limit allows)"}}
        log(f"north-star leg: {ms3:.1f} ms/step, cross-view attention {north['global_attention']['tflops']} TF/s")
        del v3

    if rank == 0:
        # the dominant KERNEL (largest total time among the instantiations timed separately); gemm_other mixes shapes, so it
        # is never the roofline row
        dom = max((k for k in kernels if not k.endswith("_other")), key=lambda k: kernels[k]["ms_total"])
        ach = kernels[dom]["tflops"]
        # HBM traffic per launch of the dominant kernel: PMC counters cannot be read from inside this process, so the
        # value is the committed rocprofv3 --pmc measurement of THIS workload (profiles/r01_traffic_n1.json), else null
        traffic = None
        tpath = next((p for p in (os.path.join(ROOT, "profiles", f) for f in ("r04_traffic_n1.json", "r03_traffic_n1.json", "r02_traffic_n1.json", "r01_traffic_n1.json")) if os.path.exists(p)), "")
        if world == 1 and n_local == 8 and H == 518 and not a.tiny and tpath:
            traffic = json.load(open(tpath))["kernels"].get(dom, {}).get("hbm_bytes_per_launch")
        roof = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_TFLOPS, 4), "traffic": traffic,
                "avg_launch_ms": kernels[dom]["avg_ms"], "flops_per_launch": fl[dom] / kernels[dom]["launches"],
                "kernels": kernels, "classes": classes,
                "kernels_note": "per-kernel times come from ONE serial, HIP-event-timed forward run after the timed region; the timed region "
                                "itself runs the camera head and the DPT heads on their own queues, so ms_per_step < forward_ms_events.  TFLOP/s are the "
                                "REFERENCE's algorithmic flops over time: the DPT heads run two algebraically reduced forms (ConvTranspose composed with "
                                "layer_rn at the token resolution; output_conv1 as nine low-resolution 1x1 products + a gather), so dpt_conv executes "
                                "fewer MFMA flops than the class's figure counts", "forward_ms_events": round(whole_ms, 3),
                "per_gpu_algorithmic_tflop": round(fl["total"] / 1e12, 2),
                "whole_forward_tflops": round(fl["total"] / (ms_step * 1e-3) / 1e12, 1),
                "whole_forward_frac": round(fl["total"] / (ms_step * 1e-3) / 1e12 / PEAK_TFLOPS, 4)}
        line = {"metric": "views/sec", "value": round(n_total / (ms_step * 1e-3), 3), "unit": "views/s", "n_gpus": world,
                "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
                "head_dtype": "f16 MFMA operands, fp32 accumulate / residuals / activations (the reference's DPT heads are fp32: worldmirror.py:146); camera head fp32",
                "data": "synthetic",
                "config": {"workload": f"{n_total}-view {H}x{W} {a.dtype}, {'camera-pose + intrinsics priors' if a.priors else 'no priors'}, camera+depth+pointmap+normal" + ("+gaussian" if a.gs else "") + " heads, "
                                       f"{n_local} views/GPU" + (", tiny arch" if a.tiny else ", full 1.23B-param arch"),
                           "views_per_gpu": n_local, "global_views": n_total, "parallelism": f"view-shard x{world}",
                           "collective": (f"RCCL {'grouped send/recv (direct)' if a.gather == 'p2p' else 'all-gather'} of K|V per global layer, world {m._comm[1]}" if m._comm else "none (1 GPU)")},
                "roofline": roof}
        if north is not None:
            line["north_star"] = north
        if multi is not None:
            line["multi_gpu"] = multi
        if world == 1 and not a.no_parity and not a.tiny and not a.gs and H == 518 and a.dtype == "bf16":
            line["parity"] = fixture_parity(m, dev)
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, H, W, a.cpu_views)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
