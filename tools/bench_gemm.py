"""GPU micro-benchmark of wm_op_gemm on the backbone's real shapes, per tile config (WM_GEMM_CFG)."""
import ctypes as C, os, subprocess, sys, json
import torch
sys.path.insert(0, '.')
SHAPES = [("qkv", 0, 3072, 1024), ("proj", 3, 1024, 1024), ("fc1", 2, 4096, 1024), ("fc2", 3, 1024, 4096)]
def child(cfg, M):
    from hunyuanworld_mirror_amd import _lib
    L = _lib.lib(); dev = torch.device('cuda:0')
    p = lambda t: C.c_void_p(t.data_ptr())
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}
    for name, epi, N, K in SHAPES:
        A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev); gamma = torch.randn(N, device=dev)
        Cc = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (0, 3) else torch.int16)
        for _ in range(3): L.wm_op_gemm(0, epi, p(A), p(W), p(Cc), p(bias), p(gamma), M, N, K, s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): L.wm_op_gemm(0, epi, p(A), p(W), p(Cc), p(bias), p(gamma), M, N, K, s)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        out[name] = (round(ms * 1e3, 1), round(2 * M * N * K / ms / 1e9, 1))
    print(json.dumps({"cfg": cfg, "M": M, "us_tflops": out}))
if __name__ == "__main__":
    if len(sys.argv) > 2:
        child(int(sys.argv[1]), int(sys.argv[2]))
    else:
        for M in (11008, 44032):
            for cfg in (4, 5, -1):
                env = dict(os.environ)
                if cfg >= 0: env["WM_GEMM_CFG"] = str(cfg)
                subprocess.run([sys.executable, __file__, str(cfg), str(M)], env=env)
