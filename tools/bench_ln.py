"""LayerNorm launch times on the backbone / DPT shapes: the branch-free kernel at 1 and 2 rows per wave, the general kernel (fp32 out)."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for M, D in ((11008, 1024), (44032, 1024), (10952, 2048)):
    x = torch.randn(M, D, device=dev); w = torch.randn(D, device=dev); b = torch.randn(D, device=dev)
    o16 = torch.empty(M, D, device=dev, dtype=torch.int16); o32 = torch.empty(M, D, device=dev)
    for name, out, f32, rpw in (("16-bit out, 1 row per wave", o16, 0, 1), ("16-bit out, 2 rows per wave", o16, 0, 2), ("fp32 out (general kernel)", o32, 1, -1)):
        L.wm_set_tuning(b"ln_rpw", rpw)
        for _ in range(5): L.wm_op_layernorm(p(x), p(out), p(w), p(b), M, D, C.c_float(1e-5), f32, 0, s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): L.wm_op_layernorm(p(x), p(out), p(w), p(b), M, D, C.c_float(1e-5), f32, 0, s)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(M, D, name, round(us, 2), "us", round(M * D * (4 + (4 if f32 else 2)) / us / 1e6, 2), "TB/s")
