"""GPU micro-benchmark + correctness of attn_v4_kernel (attn_qb = 8: one wave per SIMD, 128 query rows per wave, attention_v4.hip)
against the round-2 kernels (3 = general kernel with the integer running max, 7 = attn_v3, bf16 only) on the cross-view shapes.
Interleaved rounds in one process (guides rule 24), random gaussian data (rule 25); every variant is also compared with an fp32
torch reference on one head.  usage: python tools/bench_attn_v4.py [bf16|f16] [variants...]   env: CASES, REPS, SPIKE=1"""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
args = sys.argv[1:]
dt_name = args.pop(0) if args and args[0] in ("bf16", "f16") else "bf16"
dt = 0 if dt_name == "bf16" else 1
tdt = torch.bfloat16 if dt == 0 else torch.float16
variants = [int(x) for x in args] or ([3, 7, 8] if dt == 0 else [3, 8])
cases = [("global_8v", 16, 8 * 1376, 1, 0), ("global_16v", 16, 16 * 1376, 1, 0), ("global_32v", 16, 32 * 1376, 1, 0),
         ("sharded_8v_x8chunks", 16, 8 * 1376, 8, 8 * 1376), ("global_2v", 16, 2 * 1376, 1, 0), ("global_7v", 16, 7 * 1376, 1, 0),
         ("sharded_3v_x3chunks", 16, 3 * 1376, 3, 3 * 1376),   # ragged chunks: 64.5 key tiles each
         ("frame_8x1376", 16, 8 * 1376, 1, -1376), ("dino_8x1374", 16, 8 * 1374, 1, -1374), ("frame_32x1376", 16, 32 * 1376, 1, -1376)]  # Lc < 0: per-frame sequences of -Lc rows
if os.environ.get("CASES"):
    cases = [c for c in cases if c[0] in os.environ["CASES"].split(",")]
spike = os.environ.get("SPIKE", "0") == "1"
assert L.wm_set_tuning(b"attn_op_policy", 1) == 0   # the forward's fallback policy (sticky hints + general-kernel-only mode) on the op entry
for name, H, M, chunks, Lc in cases:
    Ls = M
    if Lc < 0:
        Ls, Lc = -Lc, 0
    g = torch.Generator(device="cpu").manual_seed(1)
    q32 = torch.randn(H, M, 64, generator=g) * 0.125 * 1.4427 * 1.5
    nk = chunks if chunks > 1 else 1
    k32 = torch.randn(nk, H, Lc if chunks > 1 else M, 64, generator=g) * 1.5
    if spike:  # rows whose maximum arrives late and far above the first tile's: forces the f16 raise path / the bf16 range check
        k32[0, :, 5000:5003] *= 6.0
        k32[-1, :, -70:-60] *= 9.0
    q = q32.to(tdt).to(dev); k = k32.to(tdt).to(dev)
    v = torch.randn(nk, H, Lc if chunks > 1 else M, 64, generator=g).to(tdt).to(dev)
    o = torch.empty(M, H * 64, device=dev, dtype=torch.int16)
    po = torch.empty(8, M, H * 64, device=dev); pml = torch.empty(8, H, M, 2, device=dev)
    nfl = int(L.wm_op_attention_flag_count(M, Ls, H)) - 4   # [flags | sticky hints | counter + pad]
    flags = torch.zeros((nfl + 4,), device=dev, dtype=torch.int32)
    keys = Ls if chunks == 1 else chunks * Lc
    fl = 4.0 * M * keys * 64 * H
    # fp32 reference on head 3, a slice of query rows of the first sequence (scores are in log2 units: P = 2^S)
    hd = 3; rows = torch.arange(0, Ls, max(1, Ls // 2048), device=dev)
    kk = k[:, hd].reshape(-1, 64).float()[:keys]; vv = v[:, hd].reshape(-1, 64).float()[:keys]
    S = q[hd][rows].float() @ kk.T
    P = torch.exp2(S - S.max(dim=1, keepdim=True).values)
    ref = (P / P.sum(dim=1, keepdim=True)) @ vv
    del S, P
    res, outs, err = {}, {}, {}
    def run():
        assert L.wm_op_attention_ex(dt, p(q), p(k), p(v), p(o), H, M, Ls, chunks, Lc, 0, p(po), p(pml), p(flags), s) == 0
    nflag = {}
    for rep in range(int(os.environ.get('REPS', '3'))):
        for qb in variants:
            assert L.wm_set_tuning(b"attn_qb", qb) == 0
            o.zero_(); flags.zero_(); flags[: nfl // 2] = 7   # (hints belong to one kernel's grid: start every variant without)
            for _ in range(2): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10 if M < 30000 else 4
            if spike:   # the host half of the fallback policy reads a counter mirrored by the PREVIOUS calls: let them finish (a server syncs per request)
                for _ in range(3): run(); torch.cuda.synchronize()
                ms = 0.0
                for _ in range(n):
                    e0.record(); run(); e1.record(); torch.cuda.synchronize()
                    ms += e0.elapsed_time(e1) / n
            else:
                e0.record()
                for _ in range(n): run()
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / n
            res.setdefault(f"qb{qb}", []).append([round(ms * 1e3, 1), round(fl / ms / 1e9)])
            outs[qb] = o.clone()
            if qb in (7, 8):
                fl_ = flags[: nfl // 2]; fl_ = fl_[fl_ != 7]
                nflag[f"qb{qb}"] = {int(a): int(b) for a, b in zip(*torch.unique(fl_[fl_ != 0], return_counts=True))}
            got = o.view(tdt).float().view(M, H, 64)[rows, hd]
            err[f"qb{qb}"] = float((got - ref).norm() / ref.norm())
    L.wm_set_tuning(b"attn_qb", -1)
    f0 = outs[variants[0]].view(tdt).float()
    diff = {f"qb{qb}": round(float((outs[qb] != outs[variants[0]]).float().mean()), 5) for qb in variants[1:]}
    rel = {f"qb{qb}": float((outs[qb].view(tdt).float() - f0).norm() / f0.norm()) for qb in variants[1:]}
    print(json.dumps({"case": name, "dtype": dt_name, "spike": spike, "us_tflops": res, "rel_l2_vs_fp32_head3": err, "fraction_differing_from_first": diff,
                      "rel_l2_vs_first": rel, "flagged_blocks": nflag}), flush=True)
