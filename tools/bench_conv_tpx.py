"""GPU micro-benchmark: 16 x 16 vs 32 x 8 pixel tiles of conv3x3_rs_kernel on the DPT shapes (conv_tpx tuning key)."""
import ctypes as C, sys, json, math
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def tune(k, v): assert L.wm_set_tuning(k.encode(), v) == 0
for (N, H, W, Cin, Cout) in [(8, 148, 148, 256, 256), (8, 74, 74, 256, 256), (8, 37, 37, 256, 256), (8, 296, 296, 256, 128), (32, 148, 148, 256, 256), (4, 148, 148, 256, 256)]:
    x = torch.randn(N, H, W, Cin, device=dev); w16 = (torch.randn(Cout, 3, 3, Cin, device=dev) / math.sqrt(9 * Cin)).half().view(torch.int16)
    b = torch.randn(Cout, device=dev); r1 = torch.randn(N, H, W, Cout, device=dev); y = torch.empty(N, H, W, Cout, device=dev)
    fl = 2.0 * N * H * W * Cout * 9 * Cin
    res = {}; outs = {}
    for rep in range(2):
        for label, v in (("16x16", 16), ("32x8", 32), ("auto", -1)):
            tune("conv_tpx", v)
            for _ in range(2): L.wm_op_conv(1, p(x), p(w16), p(b), p(r1), None, p(y), N, H, W, Cin, Cout, 3, 1, 1, 1, 1, s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): L.wm_op_conv(1, p(x), p(w16), p(b), p(r1), None, p(y), N, H, W, Cin, Cout, 3, 1, 1, 1, 1, s)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(label, []).append(round(fl / (e0.elapsed_time(e1) / 10) / 1e9))
            outs[label] = y.clone()
    tune("conv_tpx", -1)
    print(json.dumps({"shape": [N, H, W, Cin, Cout], "tflops": res, "bit_equal": bool(torch.equal(outs["16x16"], outs["32x8"]))}), flush=True)
