"""GPU micro-benchmark: attention kernel variants (attn_qb tuning key) on the per-frame and cross-view shapes."""
import ctypes as C, sys, json
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
cases = [("frame_8x1376", 16, 8 * 1376, 1376), ("dino_8x1374", 16, 8 * 1374, 1374), ("global_8v", 16, 8 * 1376, 8 * 1376), ("frame_32x1376", 16, 32 * 1376, 1376)]
variants = [int(x) for x in sys.argv[1:]] or [3, 5, 4]
for name, H, M, Ls in cases:
    q = (torch.randn(H, M, 64, device=dev) * 0.125).to(torch.bfloat16); k = torch.randn(H, M, 64, device=dev).to(torch.bfloat16)
    v = torch.randn(H, M, 64, device=dev).to(torch.bfloat16); o = torch.empty(M, H * 64, device=dev, dtype=torch.int16)
    fl = 4.0 * M * Ls * 64 * H
    res = {}; outs = {}
    for rep in range(int(__import__('os').environ.get('REPS', '2'))):
        for qb in variants:
            assert L.wm_set_tuning(b"attn_qb", qb) == 0
            for _ in range(2): L.wm_op_attention(0, p(q), p(k), p(v), p(o), H, M, Ls, 1, 0, s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): L.wm_op_attention(0, p(q), p(k), p(v), p(o), H, M, Ls, 1, 0, s)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            res.setdefault(f"qb{qb}", []).append([round(ms * 1e3, 1), round(fl / ms / 1e9)])
            outs[qb] = o.clone()
    L.wm_set_tuning(b"attn_qb", -1)
    diff = {f"qb{qb}": int((outs[qb] != outs[variants[0]]).sum()) for qb in variants[1:]}
    print(json.dumps({"case": name, "us_tflops": res, "elements_differing_from_first": diff}), flush=True)
