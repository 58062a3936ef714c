"""In-kernel clock and cycles per MFMA of the cross-view attention kernels (guides/MI355X_MICROARCH.md, DVFS give-back item 6).

Needs the diagnostic build:  make -C hunyuanworld-mirror_amd/csrc stamps   and   WM_HIP_LIB=hunyuanworld-mirror_amd/libwm_hip_stamps.so
In that build wave 0 of every block stamps s_memtime / s_memrealtime before and after its tile loop into a buffer of its own
(no output depends on it).  Here: >= 2 s of back-to-back launches on random gaussian data, then one more launch whose stamps are
read: in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, cycles per MFMA = d(s_memtime) / (MFMAs a wave issues in the
loop), medians over the whole (unsplit) blocks.  Run times of this build are not quoted.
usage: WM_HIP_LIB=... python tools/attn_stamps.py [bf16|f16] [views...]"""
import ctypes as C, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
assert hasattr(L, "wm_debug_attn_stamps"), "not the stamps build: set WM_HIP_LIB to libwm_hip_stamps.so"
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
args = sys.argv[1:]
dt_name = args.pop(0) if args and args[0] in ("bf16", "f16") else "bf16"
dt = 0 if dt_name == "bf16" else 1
tdt = torch.bfloat16 if dt == 0 else torch.float16
views = [int(x) for x in args] or [8, 32]
H = 16
for nv in views:
    M = nv * 1376
    Ls = 1376 if os.environ.get("FRAMES") == "1" else M   # FRAMES=1: nv per-frame sequences of 1376 rows (21.5 key tiles) instead of one of nv * 1376
    g = torch.Generator(device="cpu").manual_seed(1)
    q = (torch.randn(H, M, 64, generator=g) * 0.125 * 1.4427 * 1.5).to(tdt).to(dev)
    k = (torch.randn(1, H, M, 64, generator=g) * 1.5).to(tdt).to(dev)
    v = torch.randn(1, H, M, 64, generator=g).to(tdt).to(dev)
    o = torch.empty(M, H * 64, device=dev, dtype=torch.int16)
    po = torch.empty(8, M, H * 64, device=dev); pml = torch.empty(8, H, M, 2, device=dev)
    flags = torch.zeros((int(L.wm_op_attention_flag_count(M, Ls, H)),), device=dev, dtype=torch.int32)
    fl = 4.0 * M * Ls * 64 * H
    def run():
        assert L.wm_op_attention_ex(dt, p(q), p(k), p(v), p(o), H, M, Ls, 1, 0, 0, p(po), p(pml), p(flags), s) == 0
    for qb, fn, rows, mf_tile in ((7, "wm_debug_attn3_stamps", 256, 32), (8, "wm_debug_attn_stamps", 512, 64)):
        if (qb == 7 and dt != 0) or os.environ.get("STAMP_QB", str(qb)) != str(qb):
            continue
        assert L.wm_set_tuning(b"attn_qb", qb) == 0
        t0 = time.time(); n = 0
        while time.time() - t0 < 2.5:
            for _ in range(8): run()
            torch.cuda.synchronize(); n += 8
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if qb == 8 and hasattr(L, "wm_debug_attn_stamps_clear"): assert L.wm_debug_attn_stamps_clear() == 0
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        units = ((Ls + rows - 1) // rows) * (M // Ls) * H
        nb = 8192 if (qb == 8 and hasattr(L, "wm_debug_attn_stamps_clear")) else min(units, 8192)   # (cleared buffer: every block of the launch, splits included)
        buf = (C.c_ulonglong * (nb * 4))()
        assert getattr(L, fn)(buf, nb) == 0
        a = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 4).astype(np.float64)
        live = a[:, 2] > 0
        a = a[live]
        whole = a[a[:, 2] == ((Ls + 63) // 64)]          # blocks that walked every key tile
        cyc, rt, nt = whole[:, 0], whole[:, 1], whole[:, 2]
        clk = cyc / rt * 100.0                   # MHz
        cpm = cyc / (nt * mf_tile)
        extra = {}
        if qb == 8 and hasattr(L, "wm_debug_attn_stamps2"):   # kernel entry / exit of the blocks that write O themselves
            b2 = (C.c_ulonglong * (nb * 2))()
            assert L.wm_debug_attn_stamps2(b2, nb) == 0
            e = np.frombuffer(b2, dtype=np.uint64).reshape(nb, 2).astype(np.float64)[live]
            fin = e[:, 1] > 0
            t0 = e[fin, 0].min()
            extra = {"entry_to_loop_start_us_median": round(float(np.median(a[fin, 3] - e[fin, 0])) / 100.0, 2),
                     "loop_end_to_exit_us_median": round(float(np.median(e[fin, 1] - (a[fin, 3] + a[fin, 1]))) / 100.0, 2),
                     "first_entry_to_last_exit_us": round(float(e[fin, 1].max() - t0) / 100.0, 1),
                     "entry_us_p50_p90_max": [round(float(np.percentile(e[fin, 0] - t0, q)) / 100.0, 1) for q in (50, 90, 100)]}
        print(json.dumps({**extra, "views": nv, "seq_len": Ls, "dtype": dt_name, "attn_qb": qb, "whole_blocks": int(len(whole)), "launch_us_stamp_build": round(ms * 1e3, 1),
                          "tflops_stamp_build": round(fl / ms / 1e9), "in_kernel_clock_mhz_median": round(float(np.median(clk)), 1),
                          "in_kernel_clock_mhz_p10_p90": [round(float(np.percentile(clk, 10)), 1), round(float(np.percentile(clk, 90)), 1)],
                          "cycles_per_mfma_median": round(float(np.median(cpm)), 2), "cycles_per_mfma_p10_p90": [round(float(np.percentile(cpm, 10)), 2), round(float(np.percentile(cpm, 90)), 2)],
                          "loop_us_median": round(float(np.median(rt)) / 100.0, 1),
                          # the launch as the blocks saw it (100 MHz realtime counter): first loop start -> last loop end, and the loop starts' spread
                          "blocks": int(len(a)), "tiles_walked_counts": {str(int(k)): int((a[:, 2] == k).sum()) for k in np.unique(a[:, 2])},
                          "span_us_first_start_to_last_end": round(float((a[:, 3] + a[:, 1]).max() - a[:, 3].min()) / 100.0, 1),
                          "loop_start_us_p50_p90_max": [round(float(np.percentile(a[:, 3] - a[:, 3].min(), q)) / 100.0, 1) for q in (50, 90, 100)]}), flush=True)
    L.wm_set_tuning(b"attn_qb", -1)
