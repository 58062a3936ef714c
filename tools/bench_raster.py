"""GPU micro-benchmark of wm_rasterize_splats at the reference's own size: one splat per pixel of V views at 518 x 518
(prepare_splats, rasterization.py:389-498), rendered back into V views."""
import sys, json, time
import torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import Rasterizer
dev = torch.device('cuda:0')
for V in (2, 8):
    g = torch.Generator().manual_seed(5)
    N, W, H = V * 518 * 518, 518, 518
    means = torch.cat([torch.rand(N, 2, generator=g) * 3 - 1.5, torch.rand(N, 1, generator=g) * 2 + 1.5], 1).to(dev)
    quats = torch.randn(N, 4, generator=g).to(dev); scales = torch.exp(torch.rand(N, 3, generator=g) * 1.5 - 6.5).to(dev)
    opac = torch.rand(N, generator=g).to(dev); sh = (torch.rand(N, 1, 3, generator=g) * 2 - 1).to(dev)
    c2w = torch.eye(4).repeat(V, 1, 1); c2w[:, 0, 3] = torch.linspace(-0.3, 0.3, V)
    K = torch.tensor([[500.0, 0, 259], [0, 500.0, 259], [0, 0, 1]]).repeat(V, 1, 1)
    rz = Rasterizer()
    args = (means, quats, scales, opac, sh, c2w.to(dev), K.to(dev), W, H)
    for _ in range(2): rz.rasterize_splats(*args, sh_degree=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n): rz.rasterize_splats(*args, sh_degree=0)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
    print(json.dumps({"views": V, "gaussians": N, "pairs": rz.last_n_isects, "ms_per_call": round(ms, 2),
                      "views_per_s": round(V / ms * 1e3, 1), "Mpairs_per_s": round(rz.last_n_isects / ms / 1e3, 1)}), flush=True)
