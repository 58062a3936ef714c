"""GPU: A/B of GEMM tile configurations on the four backbone GEMM shapes (interleaved in one process), with a correctness
check of every epilogue for the candidate configs.  usage: python tools/gemm_cfg_ab.py 4:1 5:1 7:0 8:0   (cfg:pp)"""
import ctypes as C, sys, json, math
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def tune(k, v): assert L.wm_set_tuning(k.encode(), v) == 0
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
cands = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [(4, 1), (5, 1)]
GROUPS = (1, 4)  # gemm_group: row bands per L2 supertile
for cfg, pp in cands:
    tune("gemm_cfg", cfg); tune("gemm_pp", pp)
    for (M, N, K) in [(256, 256, 64), (1000, 384, 640), (2752, 1024, 1024), (777, 4096, 1024), (300, 64, 192)]:
        g = torch.Generator().manual_seed(M + N + K)
        A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev); W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev)
        bias = torch.randn(N, generator=g).to(dev); gamma = torch.randn(N, generator=g).to(dev)
        ref = A.float() @ W.float().t() + bias
        out = torch.full((M, N), float('nan'), device=dev)
        assert L.wm_op_gemm(0, 0, p(A), p(W), p(out), p(bias), None, M, N, K, s) == 0
        o16 = torch.zeros(M, N, device=dev, dtype=torch.int16)
        assert L.wm_op_gemm(0, 2, p(A), p(W), p(o16), p(bias), None, M, N, K, s) == 0
        X0 = torch.randn(M, N, generator=g).to(dev); X = X0.clone()
        assert L.wm_op_gemm(0, 3, p(A), p(W), p(X), p(bias), p(gamma), M, N, K, s) == 0
        e = (rel(out, ref), rel(o16.view(torch.bfloat16).float(), torch.nn.functional.gelu(ref)), rel(X, X0 + gamma * ref))
        print(f"cfg{cfg} pp{pp} {M}x{N}x{K}: f32 {e[0]:.1e} gelu16 {e[1]:.1e} resid {e[2]:.1e} {'ok' if e[0] < 2e-5 and e[1] < 6e-3 and e[2] < 2e-5 else 'FAIL'}", flush=True)
SH = [("qkvT16", 1, 3072, 1024), ("proj", 3, 1024, 1024), ("fc1", 2, 4096, 1024), ("fc2", 3, 1024, 4096)]
for M in (11008, 44032):
    for name, epi, N, K in SH:
        A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev); gamma = torch.randn(N, device=dev)
        Cc = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (0, 3) else torch.int16)
        res = {}
        for rep in range(2):
            for cfg, pp in cands:
              for gb in GROUPS:
                tune("gemm_cfg", cfg); tune("gemm_pp", pp); tune("gemm_group", gb)
                for _ in range(2): L.wm_op_gemm(0, epi, p(A), p(W), p(Cc), p(bias), p(gamma), M, N, K, s)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): L.wm_op_gemm(0, epi, p(A), p(W), p(Cc), p(bias), p(gamma), M, N, K, s)
                e1.record(); torch.cuda.synchronize()
                res.setdefault(f"cfg{cfg}pp{pp}g{gb}", []).append(round(2 * M * N * K / (e0.elapsed_time(e1) / 10) / 1e9))
        print(json.dumps({"M": M, "gemm": name, "tflops": res}), flush=True)
