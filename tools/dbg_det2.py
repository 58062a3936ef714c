import os, sys, subprocess, torch
sys.path.insert(0, '.')
if len(sys.argv) > 1:
    from hunyuanworld_mirror_amd import WorldMirror, WMConfig
    m = WorldMirror(arch=WMConfig()).to("cuda:0").init_synthetic_weights()
    g = torch.Generator().manual_seed(1234)
    img = torch.rand(1, 8, 3, 518, 518, generator=g).cuda()
    for i in range(2): o = m({"img": img}); torch.cuda.synchronize()
    torch.save({k: v.cpu() for k, v in o.items() if k in ("depth", "pts3d", "normals")}, sys.argv[1])
else:
    subprocess.run([sys.executable, __file__, "/tmp/ser.pt"], env=dict(os.environ, WM_HEADS_SERIAL="1"))
    subprocess.run([sys.executable, __file__, "/tmp/con.pt"])
    a, b = torch.load("/tmp/ser.pt"), torch.load("/tmp/con.pt")
    for k in a:
        d = (a[k] - b[k]).abs().amax(-1)[0]   # [8, 518, 518]
        bad = d > 1e-6
        print(k, "bad frac", float(bad.float().mean()), "per view", [round(float(bad[v].float().mean()), 4) for v in range(8)])
        if bad.any():
            v = int(bad.flatten(1).sum(1).argmax())
            rows = bad[v].any(1).nonzero().flatten(); cols = bad[v].any(0).nonzero().flatten()
            print("   view", v, "rows", int(rows.min()), int(rows.max()), "cols", int(cols.min()), int(cols.max()), "maxdiff", float(d.max()))
            # coarse 8x8 occupancy map of bad pixels
            occ = torch.nn.functional.avg_pool2d(bad[v].float()[None, None, :512, :512], 64)[0, 0]
            print("   occupancy(8x8):", [[round(float(x), 2) for x in r] for r in occ])
