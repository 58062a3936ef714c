"""GPU micro-benchmark of the LayerNorm kernel on the backbone shapes (run twice: WM_LN_RPW=1 and default)."""
import ctypes as C, sys, json, os
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for rows, D in ((11008, 1024), (10992, 1024), (44032, 1024), (10952, 2048)):
    x = torch.randn(rows, D, device=dev); y = torch.empty(rows, D, device=dev, dtype=torch.int16)
    w = torch.randn(D, device=dev); b = torch.randn(D, device=dev)
    f = lambda: L.wm_op_layernorm(p(x), p(y), p(w), p(b), rows, D, C.c_float(1e-5), 0, 0, s)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    ref = torch.nn.functional.layer_norm(x, (D,), w, b, 1e-5)
    err = float((y.view(torch.bfloat16).float() - ref).norm() / ref.norm())
    print(json.dumps({"rpw": os.environ.get("WM_LN_RPW", "2"), "rows": rows, "D": D, "us": round(us, 2), "TB/s": round(rows * D * 6 / us / 1e6, 2), "rel_err": err}), flush=True)
