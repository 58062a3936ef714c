"""Where does the 3x3 conv's time go?  Needs a library built with `make -C hunyuanworld-mirror_amd/csrc EXTRA=-DWM_CONV_TIMING_EXPERIMENT`
(the switches exist only in that build; results of the switched runs are wrong by design).  WM_CONV_DBG bits: 1 = no halo refill in
the loop, 2 = no epilogue, 4 = no weight refill.  usage: python tools/conv_timing_experiment.py"""
import ctypes as C, sys, json, math, os
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (N, H, W, Cin, Cout, resid) in [(8, 148, 148, 256, 256, 0), (8, 148, 148, 256, 256, 2), (8, 74, 74, 256, 256, 2)]:
    x = torch.randn(N, H, W, Cin, device=dev); w16 = (torch.randn(Cout, 3, 3, Cin, device=dev) / math.sqrt(9 * Cin)).half().view(torch.int16)
    b = torch.randn(Cout, device=dev); r1 = torch.randn(N, H, W, Cout, device=dev); r2 = torch.randn(N, H, W, Cout, device=dev); y = torch.empty(N, H, W, Cout, device=dev)
    fl = 2.0 * N * H * W * Cout * 9 * Cin
    res = {}
    for rep in range(2):
        for dbg in (0, 1, 2, 4, 3, 7):
            os.environ["WM_CONV_DBG"] = str(dbg)
            run = lambda: L.wm_op_conv(1, p(x), p(w16), p(b), p(r1) if resid else None, p(r2) if resid > 1 else None, p(y), N, H, W, Cin, Cout, 3, 1, 1, 1, 1 if resid else 0, s)
            for _ in range(2): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(f"dbg{dbg}", []).append(round(e0.elapsed_time(e1) / 10 * 1e3, 1))
    print(json.dumps({"shape": [N, H, W, Cin, Cout], "residuals": resid, "GF": round(fl / 1e9, 1), "us": res}), flush=True)
