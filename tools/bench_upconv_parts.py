"""The two launches of the tap-form output_conv1 (upconv.hip) timed apart: the low-resolution tap GEMM (M = N Hi Wi, N = 9 Co, K = C) and the gather."""
import ctypes as C, json, math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, importlib
L = importlib.import_module("hunyuanworld_mirror_amd._lib").lib()
dev = torch.device("cuda:0")
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, n=5, reps=9):
    for _ in range(3): assert fn() == 0
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[len(ts) // 2]
for (N, Hi, Wi, Ho, Wo, Cin, Co) in [(8, 148, 148, 296, 296, 256, 128), (8, 296, 296, 518, 518, 128, 32)]:
    x = torch.randn(N, Hi, Wi, Cin, device=dev).half()
    wt = (torch.randn(9 * Co, Cin, device=dev) / math.sqrt(9 * Cin)).half(); b = torch.randn(Co, device=dev)
    out = torch.empty(N, Ho, Wo, Co, device=dev)
    y16 = torch.empty(N * Hi * Wi, 9 * Co, dtype=torch.float16, device=dev)
    M = N * Hi * Wi
    t_g = timeit(lambda: L.wm_op_gemm(1, 1, p(x), p(wt), p(y16), None, None, M, 9 * Co, Cin, s))
    cfgs = {}
    for cfg in (0, 1, 4, 5):
        L.wm_set_tuning(b"gemm_cfg", cfg)
        cfgs[cfg] = round(timeit(lambda: L.wm_op_gemm(1, 1, p(x), p(wt), p(y16), None, None, M, 9 * Co, Cin, s)), 1)
    L.wm_set_tuning(b"gemm_cfg", -1)
    print(json.dumps({"tap_gemm_us_by_cfg": cfgs}), flush=True)
    t_mm = timeit(lambda: (torch.matmul(x.view(M, Cin), wt.t(), out=y16), 0)[1])
    t_s = timeit(lambda: L.wm_op_upconv_gather(p(y16), p(b), p(out), N, Hi, Wi, Ho, Wo, Co, s))
    print(json.dumps({"shape": [N, Hi, Wi, Ho, Wo, Cin, Co], "tap_gemm_us": round(t_g, 1), "torch_matmul_us": round(t_mm, 1), "gather_us": round(t_s, 1),
                      "gemm_tflops": round(2.0 * M * 9 * Co * Cin / t_g / 1e6, 1), "gather_GBps_out_plus_y": round((M * 9 * Co * 2 + N * Ho * Wo * Co * 4) / t_s / 1e3, 1)}), flush=True)
