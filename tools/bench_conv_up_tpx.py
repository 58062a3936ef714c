"""A/B of the pixel-tile shape for the fused-resize conv (148 -> 296, 256 -> 128 channels, output_conv1 of the DPT head)."""
import ctypes as C, sys, json, math
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def tune(k, v): assert L.wm_set_tuning(k.encode(), v) == 0
N, Hs, Ws, Hi, Wi, Cin, Cout = 8, 148, 148, 296, 296, 256, 128
x = torch.randn(N, Hs, Ws, Cin, device=dev); w16 = (torch.randn(Cout, 3, 3, Cin, device=dev) / math.sqrt(9 * Cin)).half().view(torch.int16)
b = torch.randn(Cout, device=dev); y = torch.empty(N, Hi, Wi, Cout, device=dev)
res = {}; outs = {}
for rep in range(3):
    for label, v in (("16x16", 16), ("32x8", 32)):
        tune("conv_tpx", v)
        for _ in range(2): L.wm_op_conv3x3_up(1, p(x), p(w16), p(b), p(y), N, Hs, Ws, Hi, Wi, Cin, Cout, None, None, s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): L.wm_op_conv3x3_up(1, p(x), p(w16), p(b), p(y), N, Hs, Ws, Hi, Wi, Cin, Cout, None, None, s)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(label, []).append(round(e0.elapsed_time(e1) / 5 * 1e3))
        outs[label] = y.clone()
tune("conv_tpx", -1)
d = (outs["16x16"] - outs["32x8"]).abs()
idx = torch.nonzero(d > 0)
print(json.dumps({"us": res, "bit_equal": bool(torch.equal(outs["16x16"], outs["32x8"])), "max_abs_diff": float(d.max()), "n_diff": int((d > 0).sum()),
                  "first_diffs": idx[:6].tolist(), "out_absmax": float(outs["16x16"].abs().max())}))
# fp32 reference of the fused op: bilinear (align_corners) then conv on f16-rounded operands
xr = torch.nn.functional.interpolate(x.permute(0, 3, 1, 2), size=(Hi, Wi), mode="bilinear", align_corners=True)
ref = torch.nn.functional.conv2d(xr.half().float(), w16.view(torch.float16).float().permute(0, 3, 1, 2), b, padding=1).permute(0, 2, 3, 1)
for k in outs: print(k, "rel err vs torch", float((outs[k] - ref).norm() / ref.norm()), "max abs", float((outs[k] - ref).abs().max()))
