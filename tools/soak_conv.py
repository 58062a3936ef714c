"""Soak test of the register-staged 3x3 conv (plain and fused-resize) against the LDS-DMA kernel: many repetitions,
bit-exact comparison expected for plain inputs, < 1e-3 for the fused resize (different FMA contraction)."""
import ctypes as C, sys, math
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def tune(k, v): assert L.wm_set_tuning(k.encode(), v) == 0
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for (N, H, Cin, Cout) in ((8, 148, 256, 256), (8, 74, 512, 256), (8, 296, 256, 128), (8, 37, 1024, 256)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, H, H, Cin, generator=g).to(dev)
    w16 = (torch.randn(Cout, 3, 3, Cin, generator=g) / math.sqrt(Cin * 9)).half().view(torch.int16).to(dev)
    b = torch.randn(Cout, generator=g).to(dev); r1 = torch.randn(N, H, H, Cout, generator=g).to(dev)
    tune("conv_rs", 0)
    y0 = torch.empty(N, H, H, Cout, device=dev)
    L.wm_op_conv(1, p(x), p(w16), p(b), p(r1), None, p(y0), N, H, H, Cin, Cout, 3, 1, 1, 1, 1, s)
    tune("conv_rs", -1)
    nbad = 0
    for rep in range(REPS):
        y = torch.empty(N, H, H, Cout, device=dev)
        L.wm_op_conv(1, p(x), p(w16), p(b), p(r1), None, p(y), N, H, H, Cin, Cout, 3, 1, 1, 1, 1, s)
        torch.cuda.synchronize()
        nbad += int((y != y0).sum())
    print("plain", (N, H, Cin, Cout), "differing elements over", REPS, "reps:", nbad, flush=True)
for (N, Hs, Hi, Cin, Cout, pos) in ((2, 74, 148, 256, 128, False), (8, 148, 296, 256, 128, False), (2, 40, 70, 128, 128, True)):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(N, Hs, Hs, Cin, generator=g).to(dev)
    w16 = (torch.randn(Cout, 3, 3, Cin, generator=g) / math.sqrt(Cin * 9)).half().view(torch.int16).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    ax = torch.randn(Hi, Cin // 2, generator=g).to(dev) if pos else None
    ay = torch.randn(Hi, Cin // 2, generator=g).to(dev) if pos else None
    tune("conv_rs", 0)
    y0 = torch.empty(N, Hi, Hi, Cout, device=dev)
    L.wm_op_conv3x3_up(1, p(x), p(w16), p(b), p(y0), N, Hs, Hs, Hi, Hi, Cin, Cout, p(ax), p(ay), s)
    tune("conv_rs", -1)
    nbad = 0; mx = 0.0
    for rep in range(REPS):
        y = torch.empty(N, Hi, Hi, Cout, device=dev)
        L.wm_op_conv3x3_up(1, p(x), p(w16), p(b), p(y), N, Hs, Hs, Hi, Hi, Cin, Cout, p(ax), p(ay), s)
        torch.cuda.synchronize()
        d = (y - y0).abs()
        nbad += int((d > 2e-3).sum()); mx = max(mx, float(d.max()))
    print("fused resize", (N, Hs, Hi, Cin, Cout, pos), "elements off by > 2e-3 over", REPS, "reps:", nbad, "max abs diff", mx, flush=True)
