import ctypes as C, sys, itertools, math, torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(Cin, Cout, Hh, Ww, N, use_bias, use_r1, use_r2, relu_in, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Hh, Ww, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).half().float().to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    r1 = torch.randn(N, Hh, Ww, Cout, generator=g).to(dev); r2 = torch.randn(N, Hh, Ww, Cout, generator=g).to(dev)
    w16 = w.permute(0, 2, 3, 1).contiguous().half()
    ys = []
    for rep in range(3):
        y = torch.full((N, Hh, Ww, Cout), float('nan'), device=dev)
        st = L.wm_op_conv(1, p(x), p(w16), p(b) if use_bias else None, p(r1) if use_r1 else None, p(r2) if use_r2 else None, p(y),
                          N, Hh, Ww, Cin, Cout, 3, 1, 1, relu_in, 1, s)
        assert st == 0
        torch.cuda.synchronize(); ys.append(y)
    xin = (torch.relu(x) if relu_in else x).half().float()
    ref = torch.nn.functional.conv2d(xin.permute(0, 3, 1, 2), w, b if use_bias else None, padding=1).permute(0, 2, 3, 1)
    if use_r1: ref = ref + torch.relu(r1)
    if use_r2: ref = ref + r2
    err = float((ys[0] - ref).norm() / ref.norm())
    det = bool(torch.equal(ys[0], ys[1]) and torch.equal(ys[1], ys[2]))
    return err, det
for shape in [(256, 256, 32, 32, 2), (1024, 256, 16, 16, 2), (256, 256, 64, 64, 2), (256, 128, 128, 128, 2)]:
    for ub, u1, u2, ri in itertools.product([0, 1], repeat=4):
        e, d = run(*shape, ub, u1, u2, ri)
        flag = "" if (e < 1e-5 and d) else "   <<<<<<"
        print(shape, "bias", ub, "r1", u1, "r2", u2, "relu", ri, f"err {e:.2e} det {d}{flag}")
