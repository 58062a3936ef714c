import ctypes as C, sys, math
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
for (Cin, Cout, Hs, Ws, Hi, Wi) in [(256, 128, 20, 16, 40, 32), (256, 128, 74, 74, 148, 148), (256, 128, 148, 148, 296, 296), (64, 64, 148, 148, 296, 296), (256,128,148,148,290,290)]:
    g = torch.Generator().manual_seed(1)
    N = 2
    x = torch.randn(N, Hs, Ws, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).half().float().to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    w16 = w.permute(0, 2, 3, 1).contiguous().half().view(torch.int16)
    y = torch.empty(N, Hi, Wi, Cout, device=dev); y2 = torch.empty_like(y); upb = torch.empty(N, Hi, Wi, Cin, device=dev)
    assert L.wm_op_conv3x3_up(1, p(x), p(w16), p(b), p(y), N, Hs, Ws, Hi, Wi, Cin, Cout, None, None, s) == 0
    assert L.wm_op_bilinear(p(x), p(upb), N, Hs, Ws, Hi, Wi, Cin, s) == 0
    assert L.wm_op_conv(1, p(upb), p(w16), p(b), None, None, p(y2), N, Hi, Wi, Cin, Cout, 3, 1, 1, 0, 0, s) == 0
    torch.cuda.synchronize()
    up = torch.nn.functional.interpolate(x.permute(0, 3, 1, 2), size=(Hi, Wi), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    ref = torch.nn.functional.conv2d(up.half().float().permute(0, 3, 1, 2), w, b, padding=1).permute(0, 2, 3, 1)
    d = (y - y2).abs()
    print((Cin, Cout, Hs, Hi), "fused-vs-torch", rel(y, ref), "unfused-vs-torch", rel(y2, ref), "fused-vs-unfused", rel(y, y2), "bilinear-vs-torch", rel(upb, up),
          "max diff at", [int(v) for v in torch.unravel_index(d.argmax(), d.shape)], float(d.max()))
