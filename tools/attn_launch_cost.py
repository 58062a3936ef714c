"""What the passes around the fast cross-view attention kernel cost per call: the recompute pass (the general kernel launched over the
same grid, every block leaving at its flag) and the combine pass of the tail split.  Each variant is a process of its own (the
switch is read once): python tools/attn_launch_cost.py [views] with WM_ATTN_DEBUG_SKIP = 0 / 1 / 2 / 3.  Timing only."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
H = 16
for nv in [int(x) for x in sys.argv[1:]] or [8, 32]:
    M = nv * 1376
    g = torch.Generator(device="cpu").manual_seed(1)
    q = (torch.randn(H, M, 64, generator=g) * 0.125 * 1.4427 * 1.5).to(torch.bfloat16).to(dev)
    k = (torch.randn(1, H, M, 64, generator=g) * 1.5).to(torch.bfloat16).to(dev)
    v = torch.randn(1, H, M, 64, generator=g).to(torch.bfloat16).to(dev)
    o = torch.empty(M, H * 64, device=dev, dtype=torch.int16)
    po = torch.empty(8, M, H * 64, device=dev); pml = torch.empty(8, H, M, 2, device=dev)
    flags = torch.zeros((int(L.wm_op_attention_flag_count(M, M, H)),), device=dev, dtype=torch.int32)
    def run():
        assert L.wm_op_attention_ex(0, p(q), p(k), p(v), p(o), H, M, M, 1, 0, 0, p(po), p(pml), p(flags), s) == 0
    for _ in range(10): run()
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        n = 40 if nv <= 8 else 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): run()
        e1.record(); torch.cuda.synchronize()
        res.append(round(e0.elapsed_time(e1) / n * 1e3, 1))
    print(json.dumps({"views": nv, "skip": int(os.environ.get("WM_ATTN_DEBUG_SKIP", "0")), "us_per_call": res}), flush=True)
