"""GPU: output_conv2[0] of the DPT head (resize 296 -> 518 + position tables, 3x3 conv 128 -> 32) — fused-resize conv (wm_op_conv3x3_up)
vs 16-bit resize pass + DMA-fed conv (wm_op_up_conv_n32); both against torch fp32 on the same 16-bit-rounded operands."""
import ctypes as C, sys, json, math
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
cases = [(8, 296, 296, 518, 518, 128)] if len(sys.argv) < 2 else [(2, 20, 16, 35, 28, 64), (3, 40, 32, 70, 56, 128), (8, 296, 296, 518, 518, 128)]
for (N, Hs, Ws, Hi, Wi, Cin) in cases:
    Cout = 32
    x = torch.randn(N, Hs, Ws, Cin, device=dev); w = (torch.randn(Cout, 3, 3, Cin, device=dev) / math.sqrt(9 * Cin)).half()
    b = torch.randn(Cout, device=dev); ax = torch.randn(Wi, Cin // 2, device=dev); ay = torch.randn(Hi, Cin // 2, device=dev)
    y1 = torch.empty(N, Hi, Wi, Cout, device=dev); y2 = torch.full((N, Hi, Wi, Cout), float("nan"), device=dev)
    up16 = torch.empty(N * Hi * Wi * Cin + 64, device=dev, dtype=torch.int16)
    f1 = lambda: L.wm_op_conv3x3_up(1, p(x), p(w.view(torch.int16)), p(b), p(y1), N, Hs, Ws, Hi, Wi, Cin, Cout, p(ax), p(ay), s)
    f2 = lambda: L.wm_op_up_conv_n32(1, p(x), p(w.view(torch.int16)), p(b), p(y2), N, Hs, Ws, Hi, Wi, Cin, p(ax), p(ay), 0, p(up16), s)
    res = {}
    for rep in range(3):
        for label, f in (("fused", f1), ("resize16+dma_conv", f2)):
            for _ in range(2): assert f() == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(label, []).append(round(e0.elapsed_time(e1) / 5 * 1e3))
    xr = torch.nn.functional.interpolate(x.permute(0, 3, 1, 2), size=(Hi, Wi), mode="bilinear", align_corners=True)
    pos = torch.cat([ax.t()[:, None, :].expand(Cin // 2, Hi, Wi), ay.t()[:, :, None].expand(Cin // 2, Hi, Wi)], 0)
    ref = torch.nn.functional.conv2d((xr + pos[None]).half().float(), w.float().permute(0, 3, 1, 2), b, padding=1).permute(0, 2, 3, 1)
    e1_, e2_ = float((y1 - ref).norm() / ref.norm()), float((y2 - ref).norm() / ref.norm())
    y2b = y2.clone(); f2(); torch.cuda.synchronize()
    print(json.dumps({"shape": [N, Hs, Ws, Hi, Wi, Cin], "us": res, "rel_err_fused": e1_, "rel_err_unfused": e2_,
                      "unfused_vs_fused": float((y2 - y1).norm() / y1.norm()), "deterministic": bool(torch.equal(y2, y2b)), "finite": bool(torch.isfinite(y2).all())}), flush=True)
