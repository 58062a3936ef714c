"""GPU micro-benchmark + correctness of the software-pipelined no-max attention kernel (attn_qb = 7, attention_v3.hip) against the
production kernel (attn_qb = 3) on the cross-view shapes: 8 / 16 / 32 views on one GPU and 8 local views x 8 gathered chunks.
Interleaved rounds in one process (guides rule 24); random gaussian data (rule 25).  usage: python tools/bench_attn_v3.py [variants...]"""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
variants = [int(x) for x in sys.argv[1:]] or [3, 7]
cases = [("global_8v", 16, 8 * 1376, 1, 0), ("global_16v", 16, 16 * 1376, 1, 0), ("global_32v", 16, 32 * 1376, 1, 0), ("sharded_8v_x8chunks", 16, 8 * 1376, 8, 8 * 1376),
         ("frame_8x1376", 16, 8 * 1376, 1, -1376), ("dino_8x1374", 16, 8 * 1374, 1, -1374), ("frame_32x1376", 16, 32 * 1376, 1, -1376)]  # Lc < 0: per-frame sequences of -Lc rows
if os.environ.get("CASES"):
    cases = [c for c in cases if c[0] in os.environ["CASES"].split(",")]
for name, H, M, chunks, Lc in cases:
    Ls = M
    if Lc < 0:
        Ls, Lc = -Lc, 0
    g = torch.Generator(device="cpu").manual_seed(1)
    q = (torch.randn(H, M, 64, generator=g) * 0.125 * 1.4427 * 1.5).to(torch.bfloat16).to(dev)
    nk = chunks if chunks > 1 else 1
    k = (torch.randn(nk, H, Lc if chunks > 1 else M, 64, generator=g) * 1.5).to(torch.bfloat16).to(dev)
    v = torch.randn(nk, H, Lc if chunks > 1 else M, 64, generator=g).to(torch.bfloat16).to(dev)
    o = torch.empty(M, H * 64, device=dev, dtype=torch.int16)
    po = torch.empty(8, M, H * 64, device=dev); pml = torch.empty(8, H, M, 2, device=dev)
    flags = torch.full((int(L.wm_op_attention_flag_count(M, Ls, H)),), 7, device=dev, dtype=torch.int32)
    keys = Ls if chunks == 1 else chunks * Lc
    fl = 4.0 * M * keys * 64 * H
    res, outs = {}, {}
    def run():
        assert L.wm_op_attention_ex(0, p(q), p(k), p(v), p(o), H, M, Ls, chunks, Lc, 0, p(po), p(pml), p(flags), s) == 0
    for rep in range(int(os.environ.get('REPS', '3'))):
        for qb in variants:
            assert L.wm_set_tuning(b"attn_qb", qb) == 0
            for _ in range(2): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10 if M < 30000 else 4
            e0.record()
            for _ in range(n): run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            res.setdefault(f"qb{qb}", []).append([round(ms * 1e3, 1), round(fl / ms / 1e9)])
            outs[qb] = o.clone()
            if qb == 7: nflag = int((flags[:] == 1).sum())
    L.wm_set_tuning(b"attn_qb", -1)
    diff = {f"qb{qb}": round(float((outs[qb] != outs[variants[0]]).float().mean()), 5) for qb in variants[1:]}
    rel = {f"qb{qb}": float((outs[qb].view(torch.bfloat16).float() - outs[variants[0]].view(torch.bfloat16).float()).norm() / outs[variants[0]].view(torch.bfloat16).float().norm()) for qb in variants[1:]}
    print(json.dumps({"case": name, "us_tflops": res, "fraction_differing_from_first": diff, "rel_l2_vs_first": rel, "flagged_blocks": nflag if 7 in variants else None}), flush=True)
