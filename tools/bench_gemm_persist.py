"""(Historical: needs the build of commit 03ab536.)  A/B of the persistent ping-pong GEMM (gemm_persist = 1) against one block per tile (0) on the backbone shapes at 8 and
32 views; interleaved rounds in one process.  usage: python tools/bench_gemm_persist.py"""
import ctypes as C, sys, json, math
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (name, M, N, K, epi) in [("fc1_8v", 11008, 4096, 1024, 2), ("qkv_as_f32_8v", 11008, 3072, 1024, 0), ("fc2_8v", 11008, 1024, 4096, 3), ("proj_8v", 11008, 1024, 1024, 3),
                             ("fc1_32v", 44032, 4096, 1024, 2), ("fc2_32v", 44032, 1024, 4096, 3), ("proj_32v", 44032, 1024, 1024, 3), ("qkv_as_t16_32v", 44032, 3072, 1024, 1)]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    b = torch.randn(N, device=dev); g = torch.randn(N, device=dev) * 0.01
    o = torch.zeros(M, N, device=dev) if epi in (0, 3) else torch.zeros(M, N, device=dev, dtype=torch.int16)
    fl = 2.0 * M * N * K
    res = {}
    for rep in range(3):
        for persist in (1, 0):
            L.wm_set_tuning(b"gemm_persist", persist)
            run = lambda: L.wm_op_gemm(0, epi, p(A), p(W), p(o), p(b), p(g) if epi == 3 else None, M, N, K, s)
            for _ in range(3): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            res.setdefault("persistent" if persist else "block_per_tile", []).append([round(us, 1), round(fl / us / 1e6)])
    L.wm_set_tuning(b"gemm_persist", -1)
    print(json.dumps({"case": name, "M_N_K": [M, N, K], "us_tflops": res}), flush=True)
