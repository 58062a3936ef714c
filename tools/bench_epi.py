"""Cost of the fused GEMM epilogues at M = 11008: plain 16-bit store vs GELU vs the QKV epilogue (LN + RoPE + relayout)."""
import ctypes as C, sys, json
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = 11008
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
res = {}
for name, N, K in (("qkv", 3072, 1024), ("fc1", 4096, 1024)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev); o16 = torch.zeros(M, N, device=dev, dtype=torch.int16)
    for rep in range(2):
        res.setdefault(name + "_t16", []).append(round(timeit(lambda: L.wm_op_gemm(0, 1, p(A), p(W), p(o16), p(bias), None, M, N, K, s))))
        if name == "fc1":
            res.setdefault("fc1_gelu", []).append(round(timeit(lambda: L.wm_op_gemm(0, 2, p(A), p(W), p(o16), p(bias), None, M, N, K, s))))
        else:
            H = 16
            q = torch.zeros(H, M, 64, device=dev, dtype=torch.int16); k = torch.zeros_like(q); v = torch.zeros_like(q)
            nw = torch.ones(64, device=dev); nb = torch.zeros(64, device=dev)
            cos = torch.ones(38 * 16, device=dev); sin = torch.zeros(38 * 16, device=dev)
            f = lambda: L.wm_op_gemm_qkv(0, p(A), p(W), p(bias), p(q), p(k), p(v), p(nw), p(nb), p(nw), p(nb), p(cos), p(sin), M, H, K, 1376, 7, 37, C.c_float(0.18), s)
            res.setdefault("qkv_fused", []).append(round(timeit(f)))
print(json.dumps(res))
