"""Cross-view attention as one rank of an 8-GPU run sees it: 8 local views of queries against 8 gathered K/V chunks."""
import ctypes as C, sys, json
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
H, M = 16, 8 * 1376
def tune(k, v): assert L.wm_set_tuning(k.encode(), v) == 0
for qb, chunks in ((3, 1), (4, 1), (3, 8), (4, 8), (3, 4), (4, 4)):
    tune("attn_qb", qb)
    q = (torch.randn(H, M, 64, device=dev) * 0.125).to(torch.bfloat16)
    k = torch.randn(chunks, H, M, 64, device=dev).to(torch.bfloat16); v = torch.randn(chunks, H, M, 64, device=dev).to(torch.bfloat16)
    o = torch.empty(M, H * 64, device=dev, dtype=torch.int16)
    f = lambda: L.wm_op_attention(0, p(q), p(k), p(v), p(o), H, M, M, chunks, M if chunks > 1 else 0, s)
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 4.0 * M * M * chunks * 64 * H
    print(json.dumps({"variant": qb, "kv_chunks": chunks, "ms": round(ms, 3), "tflops": round(fl / ms / 1e9)}), flush=True)
