"""GPU micro-benchmark: unsplit attention (kv_splits = 1) vs the automatic tail split (kv_splits = 0 with a workspace) on the
path's real shapes (bf16, random data)."""
import ctypes as C, sys, json
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
cases = [("frame_8x1376", 16, 8 * 1376, 1376), ("dino_8x1374", 16, 8 * 1374, 1374), ("global_8v", 16, 8 * 1376, 8 * 1376),
         ("global_4v", 16, 4 * 1376, 4 * 1376), ("global_16v", 16, 16 * 1376, 16 * 1376), ("global_32v", 16, 32 * 1376, 32 * 1376),
         ("frame_32x1376", 16, 32 * 1376, 1376)]
if len(sys.argv) > 1: cases = [c for c in cases if c[0] in sys.argv[1:]]
for name, H, M, Ls in cases:
    q = (torch.randn(H, M, 64, device=dev) * 0.125).to(torch.bfloat16); k = torch.randn(H, M, 64, device=dev).to(torch.bfloat16)
    v = torch.randn(H, M, 64, device=dev).to(torch.bfloat16); o = torch.empty(M, H * 64, device=dev, dtype=torch.int16)
    po = torch.empty(8, M, H * 64, device=dev); pml = torch.empty(8, H, M, 2, device=dev)
    fl = 4.0 * M * Ls * 64 * H
    res = {}
    for rep in range(2):
        for label, sp, tail in (("unsplit", 1, -1), ("tail_auto", 0, -1)) + tuple((f"tail{S}", 0, S) for S in (2, 3, 4, 5, 6, 8)):
            assert L.wm_set_tuning(b"attn_tail", tail) == 0
            for _ in range(2): L.wm_op_attention_split(0, p(q), p(k), p(v), p(o), H, M, Ls, 1, 0, sp, p(po), p(pml), s)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10 if M < 20000 else 3
            e0.record()
            for _ in range(n): L.wm_op_attention_split(0, p(q), p(k), p(v), p(o), H, M, Ls, 1, 0, sp, p(po), p(pml), s)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            res.setdefault(label, []).append([round(ms * 1e3, 1), round(fl / ms / 1e9)])
    print(json.dumps({"case": name, "us_tflops": res}), flush=True)
