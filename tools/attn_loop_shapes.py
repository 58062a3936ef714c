"""A/B rows for the MFMA shape inside the cross-view attention loop (guides DVFS give-back items 6 and 7):
tools/micro/attn_loop_shapes.hip run back to back for a few seconds per variant, interleaved rounds, random gaussian operands.
Per variant: wall TF/s (MFMA flops), in-kernel clock (s_memtime / s_memrealtime), cycles per 32x32x16-MFMA-worth of flops, board W.
build first: hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize -shared -fPIC -o tools/micro/libattn_loop_shapes.so tools/micro/attn_loop_shapes.hip
usage: python tools/attn_loop_shapes.py [seconds per phase]"""
import ctypes as C, glob, json, os, sys, threading, time
import numpy as np
import torch
SEC = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", os.environ.get("LOOP_LIB", "libattn_loop_shapes.so")))
dev = torch.device("cuda:0")
p = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
BLOCKS, ITERS = 256, 4000                       # one 256-thread block per CU (one wave per SIMD), 4000 steps of 32 gaps
g = torch.Generator(device="cpu").manual_seed(3)
ab = (torch.randn(4096 * 8, generator=g) * 0.6).to(torch.bfloat16).to(dev)   # operand fragments: scores of std ~ 0.6^2 * 4 ~ 1.4 log2 units
sink = torch.zeros(BLOCKS * 256, device=dev)
stamps = torch.zeros(BLOCKS * 2, device=dev, dtype=torch.int64)
FLOP = BLOCKS * 4 * ITERS * 32 * 32768.0        # waves x gaps x flops of one 32x32x16 MFMA

def nodes():
    out = {}
    for hw in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*'):
        for f in ('power1_average', 'power1_input', 'freq1_input'):
            pth = os.path.join(hw, f)
            if os.path.exists(pth): out.setdefault(hw.split('/')[4], {})[f] = pth
    return out
NODES = nodes()
class Sampler(threading.Thread):
    def __init__(self): super().__init__(daemon=True); self.stop = False; self.rows = []
    def run(self):
        while not self.stop:
            r = {}
            for c, d in NODES.items():
                for f, pth in d.items():
                    try: r[(c, f)] = int(open(pth).read().strip())
                    except Exception: pass
            self.rows.append(r); time.sleep(0.05)

def phase(shape):
    def launch(): assert L.attn_loop_launch(shape, 1, BLOCKS, ITERS, p(ab), p(sink), p(stamps), s) == 0
    for _ in range(2): launch()
    torch.cuda.synchronize()
    sm = Sampler(); sm.start()
    t0 = time.perf_counter(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < SEC:
        for _ in range(4): launch()
        n += 4; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    sm.stop = True; sm.join()
    ms = e0.elapsed_time(e1) / n
    st = stamps.cpu().numpy().reshape(BLOCKS, 2).astype(np.float64)   # the last launch's stamps
    clk = st[:, 0] / st[:, 1] * 100.0
    cpg = st[:, 0] / (ITERS * 32)
    rows = sm.rows[len(sm.rows) // 4:]
    pw = {}
    for c in NODES:
        v = [r.get((c, 'power1_input'), r.get((c, 'power1_average'))) for r in rows]
        v = [x for x in v if x is not None]
        if v: pw[c] = sum(v) / len(v) / 1e6
    mine = max(pw, key=pw.get) if pw else None
    fq = [r[(mine, 'freq1_input')] / 1e6 for r in rows if mine and (mine, 'freq1_input') in r]
    return {"shape": ["32x32x16 + softmax mix", "16x16x32 (two per gap) + softmax mix", "32x32x16 bare MFMA loop", "16x16x32 bare MFMA loop (two per gap)"][shape], "ms": round(ms, 3), "tflops": round(FLOP / ms / 1e9), "in_kernel_clock_mhz_median": round(float(np.median(clk)), 1),
            "cycles_per_gap_median": round(float(np.median(cpg)), 2), "board_W": round(pw[mine], 1) if mine else None, "hwmon_sclk_mhz": round(sum(fq) / len(fq), 1) if fq else None}

SHAPES = [int(x) for x in os.environ.get("SHAPES", "0,1,2,3").split(",")]
for rnd in range(int(os.environ.get("ROUNDS", "2"))):
    for shape in SHAPES:
        print(json.dumps({"round": rnd, **phase(shape)}), flush=True)
