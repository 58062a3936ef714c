import ctypes as C, sys, math, torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
def mk(seed, Cin=256, Cout=256, Hh=74, Ww=74, N=8):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Hh, Ww, Cin, generator=g).to(dev)
    w16 = (torch.randn(Cout, 3, 3, Cin, generator=g) / math.sqrt(Cin * 9)).half().to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    y = torch.empty(N, Hh, Ww, Cout, device=dev)
    return dict(x=x, w=w16, b=b, y=y, N=N, H=Hh, W=Ww, Cin=Cin, Cout=Cout)
def conv(d, stream):
    st = L.wm_op_conv(1, p(d['x']), p(d['w']), p(d['b']), None, None, p(d['y']), d['N'], d['H'], d['W'], d['Cin'], d['Cout'], 3, 1, 1, 0, 0,
                      C.c_void_p(stream.cuda_stream))
    assert st == 0
def gemm(d, stream):
    st = L.wm_op_gemm(1, 0, p(d['A']), p(d['Wg']), p(d['Cg']), p(d['bg']), None, d['M'], d['Ng'], d['K'], C.c_void_p(stream.cuda_stream))
    assert st == 0
jobs = [mk(1), mk(2), mk(3)]
for j in jobs:
    j['M'], j['Ng'], j['K'] = 10952, 256, 2048
    j['A'] = torch.randn(j['M'], j['K'], device=dev).half(); j['Wg'] = (torch.randn(j['Ng'], j['K'], device=dev) / 45).half()
    j['bg'] = torch.randn(j['Ng'], device=dev); j['Cg'] = torch.empty(j['M'], j['Ng'], device=dev)
s0 = torch.cuda.current_stream()
ref = []
for j in jobs:
    conv(j, s0); gemm(j, s0); torch.cuda.synchronize(); ref.append((j['y'].clone(), j['Cg'].clone()))
streams = [torch.cuda.Stream() for _ in jobs]
bad = 0
for it in range(20):
    for j in jobs: j['y'].fill_(float('nan')); j['Cg'].fill_(float('nan'))
    torch.cuda.synchronize()
    for rep in range(3):
        for j, s in zip(jobs, streams):
            conv(j, s); gemm(j, s)
    torch.cuda.synchronize()
    for j, (ry, rc) in zip(jobs, ref):
        if not torch.equal(j['y'], ry): bad += 1; print("it", it, "conv mismatch", float((j['y'] - ry).abs().max()))
        if not torch.equal(j['Cg'], rc): bad += 1; print("it", it, "gemm mismatch", float((j['Cg'] - rc).abs().max()))
print("mismatches:", bad)
