import ctypes as C, sys, json
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def tune(k, v): assert L.wm_set_tuning(k.encode(), v) == 0
for (M, N, K) in [(11008, 1024, 4096), (8192, 8192, 8192), (4096,4096,4096), (44032, 1024, 4096)]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    Cc = torch.zeros(M, N, device=dev)
    res = {}
    for rep in range(2):
        for label, pp in (("ppv1", 3), ("ppv2", 1), ("noDMA", 11), ("sameTile", 14)):
            tune("gemm_cfg", 4); tune("gemm_pp", pp)
            for _ in range(2): L.wm_op_gemm(0, 0, p(A), p(W), p(Cc), None, None, M, N, K, s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): L.wm_op_gemm(0, 0, p(A), p(W), p(Cc), None, None, M, N, K, s)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(label, []).append(round(2 * M * N * K / (e0.elapsed_time(e1) / 5) / 1e9))
    print(json.dumps({"shape": [M, N, K], "tflops": res}), flush=True)
