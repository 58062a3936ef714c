import os, sys, torch
os.environ["WM_DBG_SPLAT"] = "1"
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import WorldMirror, WMConfig
nv = 32
m = WorldMirror(arch=WMConfig(enable_gs=True), dtype="f16").to("cuda:0").init_synthetic_weights()
g = torch.Generator().manual_seed(555)
img = torch.rand(1, nv, 3, 518, 518, generator=g).cuda()
m.enable_prune = False
HW = 518 * 518
for it in range(24):
    o = m({"img": img}); torch.cuda.synchronize()
    cp = o["camera_params"][0]                      # [nv, 9] final
    sp = o["splats"]
    read = torch.cat([sp["opacities"][0][:, None], sp["weights"][0][:, None], sp["scales"][0][:, 0:1], sp["quats"][0], sp["scales"][0][:, 1:3]], 1)  # [M, 9] as read
    exp = cp[:, None, :].expand(nv, HW, 9).reshape(-1, 9)
    bad = (read != exp)
    nb = int(bad.any(1).sum())
    tc = sp["sh"][0][:, 0, :]
    tcn = int((tc != tc.reshape(nv, HW, 3)[:, :1, :].expand(nv, HW, 3).reshape(-1, 3)).any(1).sum())  # tc must be constant per view
    print("iter", it, "splats whose kernel read a camera vector != the final one:", nb, "| splats whose tc differs from their view's first pixel:", tcn, flush=True)
    if tcn:
        d = (tc != tc.reshape(nv, HW, 3)[:, :1, :].expand(nv, HW, 3).reshape(-1, 3))
        idx2 = d.any(1).nonzero().flatten()
        for j in idx2[:3].tolist():
            print("   tc splat", j, "view", j // HW, "pixel", j % HW, "tc", tc[j].tolist(), "view's first", tc[(j // HW) * HW].tolist(), "read", [round(x, 5) for x in read[j].tolist()])
    if nb:
        idx = bad.any(1).nonzero().flatten()
        for j in idx[:3].tolist() + idx[-2:].tolist():
            print("   splat", j, "view", j // HW, "pixel", j % HW, "read", [round(x, 5) for x in read[j].tolist()], "final", [round(x, 5) for x in cp[j // HW].tolist()], "cols", bad[j].nonzero().flatten().tolist())
        views = torch.unique(idx // HW).tolist()
        print("   views affected", views, "pixel ranges", [(int(idx[(idx // HW) == v].min() % HW), int(idx[(idx // HW) == v].max() % HW)) for v in views[:6]])
