"""Split-KV on the multi-GPU rank shape: 8 local views of queries against 8 gathered K/V chunks."""
import ctypes as C, sys, json
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
H, M = 16, 8 * 1376
for chunks in (8, 4, 2):
    q = (torch.randn(H, M, 64, device=dev) * 0.125).to(torch.bfloat16)
    k = torch.randn(chunks, H, M, 64, device=dev).to(torch.bfloat16); v = torch.randn(chunks, H, M, 64, device=dev).to(torch.bfloat16)
    o = torch.empty(M, H * 64, device=dev, dtype=torch.int16)
    po = torch.empty(8, M, H * 64, device=dev); pml = torch.empty(8, H, M, 2, device=dev)
    res = {}
    for rep in range(2):
        for label, sp, tail in (("unsplit", 1, -1), ("uniform4", 4, -1), ("auto", 0, -1), ("tail_split", 0, 1)):
            assert L.wm_set_tuning(b"attn_tail", tail) == 0
            f = lambda: L.wm_op_attention_split(0, p(q), p(k), p(v), p(o), H, M, M, chunks, M, sp, p(po), p(pml), s)
            for _ in range(2): f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): f()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            res.setdefault(label, []).append(round(4.0 * M * M * chunks * 64 * H / ms / 1e9))
    L.wm_set_tuning(b"attn_tail", -1)
    print(json.dumps({"kv_chunks": chunks, "tflops": res}), flush=True)
