import ctypes as C, sys, math
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def tune(k, v): assert L.wm_set_tuning(k.encode(), v) == 0
N, Cin, Cout, Hs, Hi = 2, 256, 128, 74, 148
g = torch.Generator().manual_seed(1)
x = torch.randn(N, Hs, Hs, Cin, generator=g).to(dev)
w16 = (torch.randn(Cout, 3, 3, Cin, generator=g) / math.sqrt(Cin * 9)).half().view(torch.int16).to(dev)
b = torch.randn(Cout, generator=g).to(dev)
tune("conv_rs", 0)
y0 = torch.empty(N, Hi, Hi, Cout, device=dev)
L.wm_op_conv3x3_up(1, p(x), p(w16), p(b), p(y0), N, Hs, Hs, Hi, Hi, Cin, Cout, None, None, s)
tune("conv_rs", 7)
nb = []
for rep in range(10):
    y = torch.empty(N, Hi, Hi, Cout, device=dev)
    L.wm_op_conv3x3_up(1, p(x), p(w16), p(b), p(y), N, Hs, Hs, Hi, Hi, Cin, Cout, None, None, s)
    torch.cuda.synchronize()
    nb.append(int(((y - y0).abs() > 1e-3).any(-1).sum()))
print("bad pixel counts:", nb)
