"""Where a backbone GEMM launch spends its time.

  stamps  (needs `make -C hunyuanworld-mirror_amd/csrc stamps`, WM_HIP_LIB=hunyuanworld-mirror_amd/libwm_hip_stamps.so):
          wave 0 of every block of gemm_pp2_kernel stamps s_memrealtime (100 MHz) at entry / K-loop start / K-loop end / exit
          (own stores drained) and the shader cycles of the loop, plus the CU it ran on.  Per launch: prologue, loop, epilogue
          in us (median, p90), in-kernel clock, per-CU gap between one block's exit and the next block's entry, and the
          launch as the blocks saw it (first entry -> last exit) beside the HIP-event time.  Run times of this build are not
          quoted anywhere else.
  yard    (product library): the same shapes, interleaved in one process on one box: torch.matmul bf16 (the vendor GEMM,
          no epilogue), our kernel with the plain 16-bit epilogue, with the fp32 epilogue, and with the epilogue the forward
          uses.  Median / min over rounds.

  exp     (stamps build, which also carries the -DWM_GEMM_PP_DEBUG timing variants; their results are wrong by construction):
          the K loop with its LDS-DMA removed (gemm_pp 16), with every block reading the same operand rows (17: all L2 hits), and
          as shipped, on both tile heights: cycles per K-tile and in-kernel clock.

  sched   (product library): the row-band schedule (wm_launch_gemm) against the full-height grid, interleaved, with the fused epilogues.

usage: python tools/gemm_timeline.py stamps|yard|exp|sched [M ...]"""
import ctypes as C, json, sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
mode = sys.argv[1]
Ms = [int(x) for x in sys.argv[2:]] or [11008, 44032]
# name, epilogue of the forward, N, K
SHAPES = [("qkv", 6, 3072, 1024), ("proj", 3, 1024, 1024), ("fc1", 2, 4096, 1024), ("fc2", 3, 1024, 4096)]
H = 16


def operands(M, N, K):
    g = torch.Generator(device="cpu").manual_seed(7)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev); gamma = (torch.randn(N, generator=g) * 0.01).to(dev)
    return A, W, bias, gamma


def runner(M, N, K, epi, A, W, bias, gamma):
    """returns a closure launching the GEMM once with epilogue `epi`"""
    if epi == 6:
        q = torch.empty(3, H, M, 64, device=dev, dtype=torch.int16)
        nw = torch.ones(64, device=dev); nb = torch.zeros(64, device=dev)
        pos = torch.arange(64, device=dev, dtype=torch.float32)[:, None] * (100.0 ** (-torch.arange(16, device=dev, dtype=torch.float32) / 16.0))[None]
        rc, rs = torch.cos(pos).contiguous(), torch.sin(pos).contiguous()
        keep = (q, nw, nb, rc, rs)
        def f():
            assert L.wm_op_gemm_qkv(0, p(A), p(W), p(bias), p(q[0]), p(q[1]), p(q[2]), p(nw), p(nb), p(nw), p(nb), p(rc), p(rs), M, H, K,
                                    1376, 7, 37, C.c_float(0.125 * 1.4426950408889634), s) == 0
        f.keep = keep
        return f
    Cc = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (0, 3) else torch.int16)
    def f():
        assert L.wm_op_gemm(0, epi, p(A), p(W), p(Cc), p(bias), p(gamma), M, N, K, s) == 0
    f.keep = Cc
    return f


def ev_time(f, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def heat(f, secs):
    t0 = time.time()
    while time.time() - t0 < secs:
        for _ in range(16): f()
        torch.cuda.synchronize()


def read_stamps():
    buf = (C.c_ulonglong * (8192 * 8))()
    assert L.wm_debug_gemm_stamps(buf, 8192) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8)
    return a[a[:, 7] == 1]


if mode == "sched":
    for M in Ms:
        for name, epi, N, K in SHAPES:
            A, W, bias, gamma = operands(M, N, K)
            f = runner(M, N, K, epi, A, W, bias, gamma)
            arms = (("v2_full_height", 2, 0), ("v2_scheduled", 2, -1), ("v3_full_height", 4, 0), ("v3_scheduled", 4, -1))
            res = {k: [] for k, _, _ in arms}
            heat(f, 0.5)
            for _ in range(7):
                for k, pp, v in arms:
                    assert L.wm_set_tuning(b"gemm_sched", v) == 0 and L.wm_set_tuning(b"gemm_pp", pp) == 0
                    f(); res[k].append(ev_time(f, 20))
            L.wm_set_tuning(b"gemm_sched", -1); L.wm_set_tuning(b"gemm_pp", -1)
            fl = 2.0 * M * N * K
            print(json.dumps({"M": M, "gemm": name, **{k: {"us_median": round(float(np.median(v)), 1), "tflops": round(fl / float(np.median(v)) / 1e6, 1)} for k, v in res.items()}}), flush=True)
elif mode == "exp":
    assert hasattr(L, "wm_debug_gemm_stamps"), "not the stamps build"
    for M in Ms:
        for name, N, K in (("fc1", 4096, 1024), ("fc2", 1024, 4096)):
            A, W, bias, gamma = operands(M, N, K)
            f = runner(M, N, K, 0, A, W, bias, gamma)
            for cfg, rows in ((4, 256), (5, 192)):
                for pp, label in ((2, "v2_shipped"), (4, "v3_half_the_barriers"), (17, "v2_same_rows_all_l2_hits"), (101, "v2_no_dma"), (102, "v2_no_frag_reads"), (104, "v2_no_barriers"), (108, "v2_no_setprio"),
                                  (103, "v2_no_dma_no_reads"), (105, "v2_no_dma_no_barriers"), (112, "v2_no_barriers_no_setprio"), (107, "v2_no_dma_reads_barriers"), (115, "v2_mfma_only")):
                    assert L.wm_set_tuning(b"gemm_cfg", cfg) == 0 and L.wm_set_tuning(b"gemm_pp", pp) == 0 and L.wm_set_tuning(b"gemm_sched", 0) == 0
                    heat(f, 0.7)
                    us = ev_time(f, 20)
                    assert L.wm_debug_gemm_stamps_clear() == 0
                    f(); torch.cuda.synchronize()
                    a = read_stamps()
                    loop = (a[:, 2] - a[:, 1]).astype(np.float64)
                    cyc = a[:, 4].astype(np.float64)
                    nk = K // 64
                    ideal = 2 * (rows // 32) * 4 * 2 * 16            # 2 waves/SIMD x SM (= rows/32) x SN (4) x 2 k-halves x 16 cycles
                    print(json.dumps({"M": M, "gemm": name, "N": N, "K": K, "tile_rows": rows, "variant": label, "launch_us": round(us, 1),
                                      "loop_us_p50": round(float(np.median(loop)) / 100.0, 2),
                                      "cycles_per_ktile_p50": round(float(np.median(cyc)) / nk, 1), "mfma_issue_cycles_per_ktile": ideal,
                                      "mfma_issue_frac": round(ideal / (float(np.median(cyc)) / nk), 3),
                                      "clock_mhz_p50": round(float(np.median(cyc / np.maximum(loop, 1.0) * 100.0)), 0)}), flush=True)
            L.wm_set_tuning(b"gemm_cfg", -1); L.wm_set_tuning(b"gemm_pp", -1); L.wm_set_tuning(b"gemm_sched", -1)
elif mode == "stamps":
    assert hasattr(L, "wm_debug_gemm_stamps"), "not the stamps build: set WM_HIP_LIB to libwm_hip_stamps.so"
    pct = lambda a, q: round(float(np.percentile(a, q)) / 100.0, 2)
    for M in Ms:
        for name, epi, N, K in SHAPES:
            A, W, bias, gamma = operands(M, N, K)
            for label, e in ((name, epi), (name + "_t16", 1)):
                f = runner(M, N, K, e, A, W, bias, gamma)
                heat(f, 1.5)
                us_back_to_back = ev_time(f, 20)
                assert L.wm_debug_gemm_stamps_clear() == 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); f(); e1.record(); torch.cuda.synchronize()
                us_event = e0.elapsed_time(e1) * 1e3
                a = read_stamps()
                t = a[:, :4].astype(np.float64)
                t0 = t[:, 0].min()
                pro, loop, epi_t = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
                clk = a[:, 4].astype(np.float64) / np.maximum(loop, 1.0) * 100.0
                cu = ((a[:, 5] >> 32) << 16) | (a[:, 5] & 0xFF00)   # XCC id | HW_ID's se / sh / cu fields: one value per CU
                gaps, per_cu = [], []
                for c in np.unique(cu):
                    tt = t[cu == c]
                    tt = tt[np.argsort(tt[:, 0])]
                    per_cu.append(len(tt))
                    gaps += list(tt[1:, 0] - tt[:-1, 3])
                nk = K // 64
                row = {"M": M, "gemm": label, "N": N, "K": K, "blocks": int(len(a)), "cus_used": int(len(np.unique(cu))),
                       "blocks_per_cu_min_max": [int(min(per_cu)), int(max(per_cu))],
                       "launch_us_event_single": round(us_event, 1), "launch_us_event_back_to_back": round(us_back_to_back, 1),
                       "span_us_first_entry_to_last_exit": round(float(t[:, 3].max() - t0) / 100.0, 1),
                       "prologue_us_p50_p90": [pct(pro, 50), pct(pro, 90)], "loop_us_p50_p90": [pct(loop, 50), pct(loop, 90)],
                       "epilogue_us_p50_p90": [pct(epi_t, 50), pct(epi_t, 90)],
                       "block_us_p50": pct(t[:, 3] - t[:, 0], 50),
                       "gap_same_cu_us_p50_p90": [pct(gaps, 50), pct(gaps, 90)] if gaps else None,
                       "entry_us_p50_p90_max": [pct(t[:, 0] - t0, 50), pct(t[:, 0] - t0, 90), pct(t[:, 0] - t0, 100)],
                       "exit_us_p10_p50_max": [pct(t[:, 3] - t0, 10), pct(t[:, 3] - t0, 50), pct(t[:, 3] - t0, 100)],
                       "in_kernel_clock_mhz_p50": round(float(np.median(clk)), 0),
                       "loop_cycles_per_ktile_p50": round(float(np.median(a[:, 4])) / nk, 1)}
                print(json.dumps(row), flush=True)
                del f
else:
    for M in Ms:
        for name, epi, N, K in SHAPES:
            A, W, bias, gamma = operands(M, N, K)
            Wt = W.t()
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            cands = {"torch_matmul_bf16": lambda: torch.matmul(A, Wt, out=out),
                     "ours_t16": runner(M, N, K, 1, A, W, bias, gamma),
                     "ours_f32": runner(M, N, K, 0, A, W, bias, gamma),
                     "ours_fused_" + name: runner(M, N, K, epi, A, W, bias, gamma)}
            for f in cands.values():
                heat(f, 0.3)
            res = {k: [] for k in cands}
            for _ in range(7):
                for k, f in cands.items():
                    res[k].append(ev_time(f, 20))
            fl = 2.0 * M * N * K
            print(json.dumps({"M": M, "gemm": name, "N": N, "K": K,
                              **{k: {"us_median": round(float(np.median(v)), 1), "us_min": round(min(v), 1),
                                     "tflops_median": round(fl / float(np.median(v)) / 1e6, 1)} for k, v in res.items()}}), flush=True)
