"""GPU: where does the per-tile time of the ping-pong GEMM go?  K sweep (slope = K-tile cost, intercept = launch +
prologue + epilogue) per epilogue, plus a no-epilogue debug variant (needs a -DWM_GEMM_PP_DEBUG build for `noepi`)."""
import ctypes as C, sys, json, math
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def tune(k, v): assert L.wm_set_tuning(k.encode(), v) == 0
DBG = len(sys.argv) > 1 and sys.argv[1] == "dbg"
def timeit(epi, A, W, Cc, bias, gamma, M, N, K, n=10):
    for _ in range(2): L.wm_op_gemm(0, epi, p(A), p(W), p(Cc), p(bias), p(gamma), M, N, K, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): L.wm_op_gemm(0, epi, p(A), p(W), p(Cc), p(bias), p(gamma), M, N, K, s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us
for M in (11008, 44032):
    for cfg, bm in ((4, 256), (5, 192)):
        for N in (1024, 4096):
            tiles = math.ceil(M / bm) * (N // 256); rounds = math.ceil(tiles / 256)
            for label, epi, pp in (("f32", 0, 1), ("t16", 1, 1), ("gelu", 2, 1), ("resid", 3, 1)) + ((("noepi", 0, 15),) if DBG else ()):
                row = {}
                for K in (512, 1024, 2048, 4096):
                    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
                    bias = torch.randn(N, device=dev); gamma = torch.randn(N, device=dev)
                    Cc = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (0, 3) else torch.int16)
                    tune("gemm_cfg", cfg); tune("gemm_pp", pp)
                    us = min(timeit(epi, A, W, Cc, bias, gamma, M, N, K) for _ in range(2))
                    row[K] = round(us / rounds, 2)
                slope = (row[4096] - row[1024]) / 48.0
                print(json.dumps({"M": M, "bm": bm, "N": N, "epi": label, "tiles": tiles, "rounds": rounds, "us_per_round": row,
                                  "us_per_ktile": round(slope, 3), "intercept_us": round(row[1024] - 16 * slope, 2)}), flush=True)
