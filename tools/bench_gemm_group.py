"""Supertile height (tuning gemm_group = row bands walked together per column tile; default 4) on the backbone GEMM shapes at 8 and
32 views: time in-process, interleaved.  The L2 -> fabric read traffic it trades (PMC FETCH_SIZE) is measured separately under
rocprofv3 with GROUPS=<g> (one value per run).  usage: python tools/bench_gemm_group.py"""
import ctypes as C, os, sys, json, math
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
groups = [int(x) for x in os.environ.get("GROUPS", "1,2,4,6,8,12,16").split(",")]
for (name, M, N, K, epi) in [("fc1_8v", 11008, 4096, 1024, 2), ("qkv_8v", 11008, 3072, 1024, 1), ("fc2_8v", 11008, 1024, 4096, 3),
                             ("fc1_32v", 44032, 4096, 1024, 2), ("qkv_32v", 44032, 3072, 1024, 1), ("fc2_32v", 44032, 1024, 4096, 3)]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    b = torch.randn(N, device=dev); g = torch.randn(N, device=dev) * 0.01
    o = torch.zeros(M, N, device=dev) if epi in (0, 3) else torch.zeros(M, N, device=dev, dtype=torch.int16)
    res = {}
    for rep in range(int(os.environ.get("REPS", "2"))):
        for G in groups:
            L.wm_set_tuning(b"gemm_group", G)
            run = lambda: L.wm_op_gemm(0, epi, p(A), p(W), p(o), p(b), p(g) if epi == 3 else None, M, N, K, s)
            for _ in range(2): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(f"G{G}", []).append(round(e0.elapsed_time(e1) / 10 * 1e3, 1))
    L.wm_set_tuning(b"gemm_group", -1)
    print(json.dumps({"case": name, "M_N_K": [M, N, K], "us": res}), flush=True)
