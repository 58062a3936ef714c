"""The under-filled generic conv launches (resize_layers[3]: 3x3 stride 2, 1024 -> 1024 at 37^2 -> 19^2; out_conv 1x1 at the small
levels): 128-pixel tiles (WM_CONV_BM=128) against 64-pixel tiles (64); one process per setting (the env is read once).
usage: python tools/bench_conv_s2.py"""
import ctypes as C, os, subprocess, sys, json, math
import torch
sys.path.insert(0, '.')
def child():
    from hunyuanworld_mirror_amd import _lib
    L = _lib.lib(); dev = torch.device('cuda:0')
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for (N, H, W, Cin, Cout, ks, st) in [(8, 37, 37, 1024, 1024, 3, 2), (8, 37, 37, 256, 256, 1, 1), (8, 19, 19, 256, 256, 1, 1), (32, 37, 37, 1024, 1024, 3, 2)]:
        pad = 1 if ks == 3 else 0
        Ho, Wo = (H + 2 * pad - ks) // st + 1, (W + 2 * pad - ks) // st + 1
        x = torch.randn(N, H, W, Cin, device=dev); w16 = (torch.randn(Cout, ks, ks, Cin, device=dev) / math.sqrt(ks * ks * Cin)).half().view(torch.int16)
        b = torch.randn(Cout, device=dev); y = torch.empty(N, Ho, Wo, Cout, device=dev)
        run = lambda: L.wm_op_conv(1, p(x), p(w16), p(b), None, None, p(y), N, H, W, Cin, Cout, ks, st, pad, 0, 0, s)
        for _ in range(3): assert run() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        fl = 2.0 * N * Ho * Wo * Cout * ks * ks * Cin
        print(json.dumps({"WM_CONV_BM": os.environ.get("WM_CONV_BM"), "shape": [N, H, W, Cin, Cout, ks, st], "us": round(us, 1), "tflops": round(fl / us / 1e6), "checksum": float(y.double().sum())}), flush=True)
if __name__ == "__main__":
    if len(sys.argv) > 1: child()
    else:
        for bm in ("128", "64"):
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, WM_CONV_BM=bm))
