"""fp32 weight-streaming Linear of the camera head (small.hip linear_f32_mfma_kernel): the fp32-MFMA form (tuning lin_mfma != 0) against
the VALU kernel (0), for 8 / 32 / 64 rows.  Each shape rotates over a pool of distinct weight tensors larger than the 256 MB memory-side
cache, as in the forward, where 870 MB of weights stream through per refinement iteration.  usage: python tools/bench_lin.py"""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0'); p = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K) in [(8, 6144, 2048), (8, 8192, 2048), (8, 2048, 8192), (8, 2048, 2048), (32, 8192, 2048), (32, 2048, 8192), (64, 6144, 2048), (13, 2048, 2048)]:
    npool = max(2, int(700e6 / (N * K * 4)))
    X = torch.randn(M, K, device=dev); Ws = [torch.randn(N, K, device=dev) for _ in range(npool)]; b = torch.randn(N, device=dev)
    Y = torch.empty(M, N, device=dev)
    ref = X @ Ws[0].t() + b
    outs = {}
    for mf in (0, 4):
        L.wm_set_tuning(b'lin_mfma', mf)
        for i in range(3): L.wm_op_linear_f32(p(X), p(Ws[i % npool]), p(b), p(Y), M, N, K, K, 0, 0, s)
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3 * npool
        e0.record()
        for i in range(reps): L.wm_op_linear_f32(p(X), p(Ws[i % npool]), p(b), p(Y), M, N, K, K, 0, 0, s)
        e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / reps
        L.wm_op_linear_f32(p(X), p(Ws[0]), p(b), p(Y), M, N, K, K, 0, 0, s); torch.cuda.synchronize()
        outs[mf] = Y.clone()
        err = float((Y - ref).abs().max())
        print((M, N, K), f"lin_mfma={mf:2d}  {ms*1e3:6.1f} us  {N*K*4/ms/1e9:.2f} TB/s  maxerr {err:.1e}  bit-identical to lin_mfma=4: {bool(torch.equal(outs[mf], outs[4])) if 4 in outs else None}", flush=True)
L.wm_set_tuning(b'lin_mfma', -1)
