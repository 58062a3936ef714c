"""Board power and shader clock while one kernel runs back to back (sysfs hwmon of the card; rocm-smi as a fallback).

Why: the PMC runs of the pipelined cross-view attention show the matrix pipes 71 % busy at an average 1.66 GHz, the GEMMs 34-38 %
busy at 2.35-2.5 GHz.  If the attention kernel sits at the board's power limit, its throughput is set by energy per flop, not by
issue slots — this probe records power / clock per kernel so that the claim is a measurement.
usage: python tools/power_probe.py [seconds per phase]   -> JSON lines"""
import ctypes as C, glob, json, os, subprocess, sys, threading, time
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
SEC = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0

def sysfs_nodes():
    out = {}
    for hw in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*'):
        for f in ('power1_average', 'power1_input', 'freq1_input', 'freq2_input', 'temp1_input', 'power1_cap'):
            pth = os.path.join(hw, f)
            if os.path.exists(pth):
                out.setdefault(hw, {})[f] = pth
    return out
NODES = sysfs_nodes()

def read_all():
    r = {}
    for hw, d in NODES.items():
        for f, pth in d.items():
            try:
                r[f"{hw.split('/')[4]}:{f}"] = int(open(pth).read().strip())   # /sys/class/drm/cardN/...
            except Exception:
                pass
    return r

def smi():
    try:
        o = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=20).stdout
        return json.loads(o)
    except Exception as e:
        return {"error": str(e)}

class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True); self.stop = False; self.rows = []
    def run(self):
        while not self.stop:
            self.rows.append(read_all()); time.sleep(0.05)

def phase(name, launch, flops):
    for _ in range(3): launch()
    torch.cuda.synchronize()
    sm = Sampler(); sm.start()
    t0 = time.perf_counter(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    mid = None
    while time.perf_counter() - t0 < SEC:
        for _ in range(8): launch()
        n += 8
        torch.cuda.synchronize()
        if mid is None and time.perf_counter() - t0 > SEC / 2 and not NODES:
            mid = smi()
    e1.record(); torch.cuda.synchronize()
    sm.stop = True; sm.join()
    ms = e0.elapsed_time(e1) / max(n, 1)
    keys = sorted({k for r in sm.rows for k in r})
    agg = {}
    rows = sm.rows[len(sm.rows) // 4:]          # skip the ramp
    for k in keys:
        v = [r[k] for r in rows if k in r]
        if v: agg[k] = {"mean": sum(v) / len(v), "min": min(v), "max": max(v)}
    # the box's host shows every card of the node: ours is the one drawing the most power during the phase
    cards = sorted({k.split(':')[0] for k in agg})
    pw = {c: agg.get(f"{c}:power1_input", agg.get(f"{c}:power1_average", {"mean": 0}))["mean"] for c in cards}
    mine = max(pw, key=pw.get) if pw else None
    own = {k.split(':')[1]: {a: round(b / 1e6, 1) for a, b in v.items()} for k, v in agg.items() if mine and k.startswith(mine + ":")}
    print(json.dumps({"phase": name, "ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 1) if flops else None, "samples": len(rows),
                      "card": mine, "W_or_MHz": own, "other_cards_mean_W": {c: round(v / 1e6) for c, v in pw.items() if c != mine}, "rocm_smi": mid}), flush=True)

print(json.dumps({"nodes": {k: list(v) for k, v in NODES.items()}, "smi_idle": smi() if not NODES else None, "idle": read_all()}), flush=True)

def attn_case(M, qb):
    H = 16
    g = torch.Generator(device="cpu").manual_seed(1)
    q = (torch.randn(H, M, 64, generator=g) * 0.125 * 1.4427 * 1.5).to(torch.bfloat16).to(dev)
    k = (torch.randn(1, H, M, 64, generator=g) * 1.5).to(torch.bfloat16).to(dev)
    v = torch.randn(1, H, M, 64, generator=g).to(torch.bfloat16).to(dev)
    o = torch.empty(M, H * 64, device=dev, dtype=torch.int16)
    po = torch.empty(8, M, H * 64, device=dev); pml = torch.empty(8, H, M, 2, device=dev)
    flags = torch.zeros((int(L.wm_op_attention_flag_count(M, M, H)),), device=dev, dtype=torch.int32)
    def run():
        assert L.wm_set_tuning(b"attn_qb", qb) == 0
        assert L.wm_op_attention_ex(0, p(q), p(k), p(v), p(o), H, M, M, 1, 0, 0, p(po), p(pml), p(flags), s) == 0
    return run, 4.0 * M * M * 64 * H, (q, k, v, o, po, pml, flags)

def gemm_case(M, N, K, epi):
    g = torch.Generator(device="cpu").manual_seed(2)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.03).to(torch.bfloat16).to(dev)
    Cc = torch.empty(M, N, device=dev, dtype=torch.int16)
    bias = torch.zeros(N, device=dev)
    def run():
        assert L.wm_op_gemm(0, epi, p(A), p(W), p(Cc), p(bias), None, M, N, K, s) == 0
    return run, 2.0 * M * N * K, (A, W, Cc, bias)

def tuned(mk, **kv):
    def make():
        run, fl, keep = mk()
        def run2():
            for k, v in kv.items(): L.wm_set_tuning(k.encode(), v)
            run()
            for k in kv: L.wm_set_tuning(k.encode(), -1)
        return run2, fl, keep
    return make

PHASES = [("attention_v3_32v", lambda: attn_case(32 * 1376, 7)), ("attention_general_32v", lambda: attn_case(32 * 1376, 3)),
          ("attention_v3_8v", lambda: attn_case(8 * 1376, 7)), ("gemm_fc1_8v_gelu", lambda: gemm_case(8 * 1376, 4096, 1024, 2)),
          ("gemm_fc1_32v_gelu", lambda: gemm_case(32 * 1376, 4096, 1024, 2))]
if os.environ.get("MFMA_SHAPES"):   # energy per flop of the two MFMA shapes: the lock-step GEMM kernels exist in both
    PHASES = [("gemm_fc1_32v pingpong 16x16x32", lambda: gemm_case(32 * 1376, 4096, 1024, 1)),
              ("gemm_fc1_32v lockstep 32x32x16", tuned(lambda: gemm_case(32 * 1376, 4096, 1024, 1), gemm_pp=0, gemm_mfma16=0, gemm_cfg=4)),
              ("gemm_fc1_32v lockstep 16x16x32", tuned(lambda: gemm_case(32 * 1376, 4096, 1024, 1), gemm_pp=0, gemm_mfma16=2, gemm_cfg=4)),
              ("gemm_4096^3 pingpong 16x16x32", lambda: gemm_case(4096, 4096, 4096, 1)),
              ("gemm_4096^3 lockstep 32x32x16", tuned(lambda: gemm_case(4096, 4096, 4096, 1), gemm_pp=0, gemm_mfma16=0, gemm_cfg=4)),
              ("gemm_4096^3 lockstep 16x16x32", tuned(lambda: gemm_case(4096, 4096, 4096, 1), gemm_pp=0, gemm_mfma16=2, gemm_cfg=4))]
for name, mk in PHASES:
    run, fl, keep = mk()
    phase(name, run, fl)
    del keep
    time.sleep(1.0)
L.wm_set_tuning(b"attn_qb", -1)
