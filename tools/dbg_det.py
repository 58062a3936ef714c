import sys, torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import WorldMirror, WMConfig
m = WorldMirror(arch=WMConfig()).to("cuda:0").init_synthetic_weights()
g = torch.Generator().manual_seed(1234)
img = torch.rand(1, 8, 3, 518, 518, generator=g).cuda()
outs = []
for i in range(8):
    o = m({"img": img}); torch.cuda.synchronize()
    outs.append({k: v.clone() for k, v in o.items()})
for i in range(1, 8):
    print("run", i - 1, "vs", i, {k: (bool(torch.equal(outs[i][k], outs[i - 1][k])), float((outs[i][k] - outs[i - 1][k]).abs().max())) for k in ("depth", "pts3d", "normals", "camera_params")})
