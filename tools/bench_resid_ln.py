"""Residual GEMM with the following LayerNorm fused into its epilogue (wm_op_gemm_resid_ln) against the residual GEMM + the LayerNorm kernel,
interleaved on one box; proj (K = 1024) and fc2 (K = 4096) at M = 11008."""
import ctypes as C, json, sys
import numpy as np
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N = 11008, 1024
assert L.wm_set_tuning(b"ln_fuse", 1) == 0   # the fused epilogue is opt-in
for K in (1024, 4096):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev); gamma = torch.randn(N, device=dev) * 0.01; lw = torch.randn(N, device=dev); lb = torch.randn(N, device=dev)
    X = torch.randn(M, N, device=dev); out16 = torch.empty(M, N, device=dev, dtype=torch.int16)
    stats = torch.empty(M * 8, device=dev); sync = torch.zeros(3 * (M // 16 + 2), device=dev, dtype=torch.int32)
    fused = C.c_int(0)
    def f_fused():
        assert L.wm_op_gemm_resid_ln(0, p(A), p(W), p(X), p(bias), p(gamma), p(lw), p(lb), C.c_float(1e-5), p(out16), p(stats), p(sync), M, N, K, C.byref(fused), s) == 0
    def f_split():
        assert L.wm_op_gemm(0, 3, p(A), p(W), p(X), p(bias), p(gamma), M, N, K, s) == 0
        assert L.wm_op_layernorm(p(X), p(out16), p(lw), p(lb), M, N, C.c_float(1e-5), 0, 0, s) == 0
    res = {"fused": [], "gemm_then_layernorm": []}
    for _ in range(7):
        for k, f in (("fused", f_fused), ("gemm_then_layernorm", f_split)):
            for _ in range(3): f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / 20 * 1e3)
    print(json.dumps({"M": M, "K": K, "fused_taken": fused.value, **{k: round(float(np.median(v)), 1) for k, v in res.items()}}), flush=True)
