import ctypes as C, sys, json
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def tune(k, v): assert L.wm_set_tuning(k.encode(), v) == 0
H, M = 16, 32 * 1376
q = (torch.randn(H, M, 64, device=dev) * 0.125).to(torch.bfloat16); k = torch.randn(H, M, 64, device=dev).to(torch.bfloat16)
v = torch.randn(H, M, 64, device=dev).to(torch.bfloat16); o = torch.empty(M, H * 64, device=dev, dtype=torch.int16)
po = torch.zeros(1, device=dev); pml = torch.zeros(64, device=dev)
for qb in (6, 5):
    tune("attn_qb", qb)
    pml.zero_()
    for _ in range(2): L.wm_op_attention_split(0, p(q), p(k), p(v), p(o), H, M, M, 1, 0, 1, p(po), p(pml), s)
    torch.cuda.synchronize()
    r = pml.cpu().reshape(8, 8)[:4]
    n = float(r[0, 4])
    print("qb", qb, "per-iteration cycles (memtime ticks) [dma/addr, halfA, halfB, end_iter] per wave:", [[round(float(x) / n) for x in row[:4]] for row in r], flush=True)
