"""Same-box A/B of the residual GEMMs (proj: K = 1024, fc2: K = 4096; N = 1024) with and without the old-C-tile prefetch of the last two
K-tiles (tuning resid_prefetch).  Interleaved rounds, median us per launch."""
import ctypes as C, json, math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import importlib
L = importlib.import_module("hunyuanworld_mirror_amd._lib").lib()
dev = torch.device("cuda:0")
p = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for M in (11008, 44032):
    for K in (1024, 4096):
        N = 1024
        A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
        b = torch.randn(N, device=dev); g = torch.randn(N, device=dev) * 0.01; X = torch.randn(M, N, device=dev)
        res = {0: [], 1: []}
        for rnd in range(7):
            for pf in (0, 1):
                L.wm_set_tuning(b"resid_prefetch", pf)
                for _ in range(3): L.wm_op_gemm(0, 3, p(A), p(W), p(X), p(b), p(g), M, N, K, s)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): L.wm_op_gemm(0, 3, p(A), p(W), p(X), p(b), p(g), M, N, K, s)
                e1.record(); torch.cuda.synchronize()
                res[pf].append(e0.elapsed_time(e1) / 20 * 1e3)
        L.wm_set_tuning(b"resid_prefetch", -1)
        med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
        print(json.dumps({"M": M, "K": K, "us_no_prefetch": round(med[0], 2), "us_prefetch": round(med[1], 2)}), flush=True)
