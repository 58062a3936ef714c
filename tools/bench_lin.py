import ctypes as C, sys, torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0'); p = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
import itertools
for (M, N, K), mf in itertools.product([(8, 6144, 2048), (8, 8192, 2048), (8, 2048, 8192), (8, 2048, 2048), (13, 2048, 2048)], (0, 1)):
    L.wm_set_tuning(b'lin_mfma', mf)
    X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); Y = torch.empty(M, N, device=dev)
    for _ in range(3): L.wm_op_linear_f32(p(X), p(W), p(b), p(Y), M, N, K, K, 0, 0, s)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): L.wm_op_linear_f32(p(X), p(W), p(b), p(Y), M, N, K, K, 0, 0, s)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 20
    ok = float((Y - (X @ W.t() + b)).abs().max())
    print((M, N, K), "mfma" if mf else "valu", f"{ms*1e3:.1f} us  {N*K*4/ms/1e9:.2f} TB/s  maxerr {ok:.1e}")
