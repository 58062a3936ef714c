"""Same-box A/B of the RCU's second conv (3x3, 256 -> 256, 16-bit NHWC input, relu(resid) + fusion add) in its two forms: the
register-staged halo kernel (wm_op_conv via a ... not exposed with in16: timed through the whole forward instead) and the ping-pong
GEMM form (wm_op_conv3x3_gemm16).  Prints one JSON line per shape: median us and TFLOP/s of the GEMM form."""
import ctypes as C, json, math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import importlib
_libm = importlib.import_module("hunyuanworld_mirror_amd._lib")
L = _libm.lib()
dev = torch.device("cuda:0")
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (N, H, W, Cin, Cout) in [(8, 148, 148, 256, 256), (8, 74, 74, 256, 256)]:
    x = torch.randn(N, H, W, Cin, device=dev).half()
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) / math.sqrt(9 * Cin)).half()
    b = torch.randn(Cout, device=dev); r1 = torch.randn(N, H, W, Cout, device=dev); r2 = torch.randn(N, H, W, Cout, device=dev)
    y = torch.empty(N, H, W, Cout, device=dev); y16 = torch.empty(N, H, W, Cout, device=dev, dtype=torch.float16)
    zero = torch.zeros(128, dtype=torch.int16, device=dev)
    xf = torch.randn(N, H, W, Cin, device=dev)
    for name, fn in (("gemm_resid_f32", lambda: L.wm_op_conv3x3_gemm16(1, p(x), p(w), p(b), p(r1), 1, p(r2), p(y), 0, 0, N, H, W, Cin, Cout, p(zero), s)),
                     ("gemm_relu_o16", lambda: L.wm_op_conv3x3_gemm16(1, p(x), p(w), p(b), None, 0, None, p(y16), 1, 1, N, H, W, Cin, Cout, p(zero), s)),
                     ("halo_f32in_relu_o16", lambda: L.wm_op_conv_ex(1, p(xf), 0, p(w), p(b), None, None, p(y16), 1, N, H, W, Cin, Cout, 1, 0, 1, s)),
                     ("halo_in16_relu_o16", lambda: L.wm_op_conv_ex(1, p(x), 1, p(w), p(b), None, None, p(y16), 1, N, H, W, Cin, Cout, 0, 0, 1, s)),
                     ("halo_in16_resid_f32", lambda: L.wm_op_conv_ex(1, p(x), 1, p(w), p(b), p(r1), p(r2), p(y), 0, N, H, W, Cin, Cout, 0, 1, 0, s)),
                     ("halo_in16_resid_o16", lambda: L.wm_op_conv_ex(1, p(x), 1, p(w), p(b), p(r1), None, p(y16), 1, N, H, W, Cin, Cout, 0, 1, 0, s)),
                     ("halo_f32in_plain_f32", lambda: L.wm_op_conv_ex(1, p(xf), 0, p(w), None, None, None, p(y), 0, N, H, W, Cin, Cout, 0, 0, 0, s)),
                     ("halo_f32in_resid", lambda: L.wm_op_conv(1, p(xf), p(w), p(b), p(r1), p(r2), p(y), N, H, W, Cin, Cout, 3, 1, 1, 1, 1, s))):
        for _ in range(3):
            assert fn() == 0
        ts = []
        for _ in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5 * 1e3)
        ts.sort()
        us = ts[len(ts) // 2]
        fl = 2.0 * N * H * W * Cout * 9 * Cin
        print(json.dumps({"shape": [N, H, W, Cin, Cout], "form": name, "us": round(us, 1), "tflops": round(fl / us / 1e6, 1)}), flush=True)

# output_conv1 behind the last resize (148^2 -> 296^2, 256 -> 128): the fused-resize halo kernel vs the tap form (upconv.hip)
N, Hi, Wi, Ho, Wo, Cin, Co = 8, 148, 148, 296, 296, 256, 128
x = torch.randn(N, Hi, Wi, Cin, device=dev).half(); xf = x.float()
w = (torch.randn(Co, 3, 3, Cin, device=dev) / math.sqrt(9 * Cin)).half(); b = torch.randn(Co, device=dev)
out = torch.empty(N, Ho, Wo, Co, device=dev); wt = torch.empty(9 * Co * Cin, dtype=torch.int16, device=dev)
y16 = torch.empty(N * Hi * Wi * 9 * Co, dtype=torch.int16, device=dev)
for name, fn in (("up1_tap_form", lambda: L.wm_op_upconv3x3_tap(1, p(x), p(w), p(b), p(out), N, Hi, Wi, Ho, Wo, Cin, Co, p(wt), p(y16), s)),
                 ("up1_fused_halo", lambda: L.wm_op_conv3x3_up(1, p(xf), p(w), p(b), p(out), N, Hi, Wi, Ho, Wo, Cin, Co, None, None, s))):
    for _ in range(3):
        assert fn() == 0
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5 * 1e3)
    ts.sort()
    print(json.dumps({"shape": [N, Hi, Wi, Ho, Wo, Cin, Co], "form": name, "us": round(ts[len(ts) // 2], 1)}), flush=True)
