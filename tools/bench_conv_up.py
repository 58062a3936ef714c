"""A/B of the narrow conv variant on output_conv2[0] (128 -> 32 at 518^2, resize 296 -> 518 fused)."""
import ctypes as C, sys, json, math
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def tune(k, v): assert L.wm_set_tuning(k.encode(), v) == 0
N, Hs, Ws, Hi, Wi, Cin, Cout = 8, 296, 296, 518, 518, 128, 32
x = torch.randn(N, Hs, Ws, Cin, device=dev); w16 = (torch.randn(Cout, 3, 3, Cin, device=dev) / math.sqrt(9 * Cin)).half().view(torch.int16)
b = torch.randn(Cout, device=dev); y = torch.empty(N, Hi, Wi, Cout, device=dev)
ax = torch.randn(Wi, Cin // 2, device=dev); ay = torch.randn(Hi, Cin // 2, device=dev)
res = {}; outs = {}
for rep in range(2):
    for label, pp in (("per_tap", 0), ("narrow", -1)):
        tune("conv_narrow", pp)
        for _ in range(2): L.wm_op_conv3x3_up(1, p(x), p(w16), p(b), p(y), N, Hs, Ws, Hi, Wi, Cin, Cout, p(ax), p(ay), s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): L.wm_op_conv3x3_up(1, p(x), p(w16), p(b), p(y), N, Hs, Ws, Hi, Wi, Cin, Cout, p(ax), p(ay), s)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(label, []).append(round(e0.elapsed_time(e1) / 5 * 1e3))
        outs[label] = y.clone()
print(json.dumps({"us": res, "max_abs_diff": float((outs["per_tap"] - outs["narrow"]).abs().max())}))
