"""In-process A/B of the attention kernel variants (wm_set_tuning attn_qb): 3 = production (lazy max, LDS-DMA), 10 = eager
row max, 6 = software-pipelined half-tile kernel.  Prints rel. error vs fp32 softmax, then timings.
(The -18 / -19 / -35 % ablation numbers in DESIGN.md were taken with temporary debug variants of the pre-lazy kernel.)"""
import ctypes as C, sys, json
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def tune(k, v): assert L.wm_set_tuning(k.encode(), v) == 0
import math
def ref_check(qb, H=4, M=2752):
    g = torch.Generator().manual_seed(1)
    q = (torch.randn(H, M, 64, generator=g) * 0.125 * 1.5 * 1.4427).to(torch.bfloat16).to(dev); k = (torch.randn(H, M, 64, generator=g) * 1.5).to(torch.bfloat16).to(dev); v = torch.randn(H, M, 64, generator=g).to(torch.bfloat16).to(dev)
    o = torch.empty(M, H * 64, device=dev, dtype=torch.int16)
    tune("attn_qb", qb)
    L.wm_op_attention(0, p(q), p(k), p(v), p(o), H, M, M, 1, 0, s); torch.cuda.synchronize()
    a = torch.softmax((q.float() @ k.float().transpose(-1, -2)) * math.log(2.0), -1) @ v.float()
    got = o.view(torch.bfloat16).float().reshape(M, H, 64).transpose(0, 1)
    return float((got - a).norm() / a.norm())
for qb in (3, 10, 6): print("relerr qb", qb, ref_check(qb), ref_check(qb, 2, 1000), flush=True)
for name, H, M, Ls in [("frame_8x1376", 16, 8 * 1376, 1376), ("global_8v", 16, 8 * 1376, 8 * 1376), ("global_32v", 16, 32 * 1376, 32 * 1376)]:
    q = (torch.randn(H, M, 64, device=dev) * 0.125).to(torch.bfloat16); k = torch.randn(H, M, 64, device=dev).to(torch.bfloat16)
    v = torch.randn(H, M, 64, device=dev).to(torch.bfloat16); o = torch.empty(M, H * 64, device=dev, dtype=torch.int16)
    fl = 4.0 * M * Ls * 64 * H
    res = {}
    for rep in range(2):
        for label, qb in (("lazy_dma", 3), ("eager_max", 10), ("sw_pipelined", 6)):
            tune("attn_qb", qb)
            for _ in range(2): L.wm_op_attention(0, p(q), p(k), p(v), p(o), H, M, Ls, 1, 0, s)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10 if M < 20000 else 3
            e0.record()
            for _ in range(n): L.wm_op_attention(0, p(q), p(k), p(v), p(o), H, M, Ls, 1, 0, s)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(label, []).append(round(e0.elapsed_time(e1) / n * 1e3))
    print(json.dumps({"case": name, "us": res}), flush=True)
