"""Multi-queue stress: is anything corrupted when several HIP queues are active?
  part A: the 3-stream conv -> conv -> conv -> bilinear chains of tools/dbg_chain2.py, ROUNDS rounds, every word checked;
  part B: the full model at BASELINE C2 (8 x 518 x 518) with the three DPT heads on three streams (WM_HEADS_CONCURRENT=1), FWD
          forwards: every output must be bit-identical to the single-queue forward of the same process-independent inputs.
usage: python tools/stress_multiqueue.py [ROUNDS=100] [FWD=12]"""
import ctypes as C, hashlib, json, math, os, subprocess, sys
import torch
sys.path.insert(0, '.')


def part_a(rounds, nq=3):
    from hunyuanworld_mirror_amd import _lib
    L = _lib.lib(); dev = torch.device('cuda:0')
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    Cc, Hh, N = 128, 296, 4
    def mk(seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(N, Hh, Hh, Cc, generator=g).to(dev)
        ws = [(torch.randn(Cc, 3, 3, Cc, generator=g) / math.sqrt(Cc * 9)).half().to(dev) for _ in range(3)]
        b = torch.randn(Cc, generator=g).to(dev)
        return dict(x=x, ws=ws, b=b, bufs=[torch.empty(N, Hh, Hh, Cc, device=dev) for _ in range(2)], up=torch.empty(N, 518, 518, Cc, device=dev))
    def chain(d, st):
        s = C.c_void_p(st.cuda_stream)
        assert L.wm_op_conv(1, p(d['x']), p(d['ws'][0]), p(d['b']), None, None, p(d['bufs'][0]), N, Hh, Hh, Cc, Cc, 3, 1, 1, 1, 0, s) == 0
        assert L.wm_op_conv(1, p(d['bufs'][0]), p(d['ws'][1]), p(d['b']), None, None, p(d['bufs'][1]), N, Hh, Hh, Cc, Cc, 3, 1, 1, 1, 0, s) == 0
        assert L.wm_op_conv(1, p(d['bufs'][1]), p(d['ws'][2]), p(d['b']), None, None, p(d['bufs'][0]), N, Hh, Hh, Cc, Cc, 3, 1, 1, 1, 0, s) == 0
        assert L.wm_op_bilinear(p(d['bufs'][0]), p(d['up']), N, Hh, Hh, 518, 518, Cc, s) == 0
    jobs = [mk(i + 1) for i in range(nq)]
    s0 = torch.cuda.current_stream()
    ref = []
    for j in jobs:
        chain(j, s0); torch.cuda.synchronize()
        ref.append((j['bufs'][1].clone(), j['bufs'][0].clone(), j['up'].clone()))
    streams = [torch.cuda.Stream() for _ in jobs]
    bad = 0
    for it in range(rounds):
        torch.cuda.synchronize()
        for j, s in zip(jobs, streams): chain(j, s)
        torch.cuda.synchronize()
        for ji, j in enumerate(jobs):
            for nm, cur, r in (("conv2", j['bufs'][1], ref[ji][0]), ("conv3", j['bufs'][0], ref[ji][1]), ("bilinear", j['up'], ref[ji][2])):
                n = int((cur != r).sum())
                if n:
                    bad += n
                    print(f"round {it} job {ji} {nm}: {n} wrong words", flush=True)
    return bad


def model_digests(fwd):
    from hunyuanworld_mirror_amd import WorldMirror, WMConfig
    m = WorldMirror(arch=WMConfig()).to("cuda:0").init_synthetic_weights()
    g = torch.Generator().manual_seed(1234)
    img = torch.rand(1, 8, 3, 518, 518, generator=g).cuda()
    out = []
    for _ in range(fwd):
        o = m({"img": img})
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for k in ("pts3d", "pts3d_conf", "depth", "depth_conf", "normals", "normals_conf", "camera_params"):
            h.update(o[k].cpu().numpy().tobytes())
        out.append(h.hexdigest())
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--model":
        print(json.dumps(model_digests(int(sys.argv[2]))))
        sys.exit(0)
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    fwd = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    nq = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    import hunyuanworld_mirror_amd  # noqa
    print(json.dumps({"part": "A", "hip_runtime": torch.version.hip, "rounds": rounds, "streams": nq, "wrong_words": part_a(rounds, nq)}), flush=True)
    if fwd <= 0:
        sys.exit(0)
    res = {}
    for mode in ("serial", "concurrent"):
        env = dict(os.environ)
        env.pop("WM_HEADS_CONCURRENT", None)
        if mode == "concurrent":
            env["WM_HEADS_CONCURRENT"] = "1"
        r = subprocess.run([sys.executable, __file__, "--model", str(fwd if mode == "concurrent" else 2)], env=env, capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            print(r.stderr[-2000:]); sys.exit(1)
        res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    same = all(d == res["serial"][0] for d in res["serial"] + res["concurrent"])
    print(json.dumps({"part": "B", "forwards_concurrent": len(res["concurrent"]), "distinct_digests": len(set(res["serial"] + res["concurrent"])), "all_bit_identical_to_single_queue": same}), flush=True)
