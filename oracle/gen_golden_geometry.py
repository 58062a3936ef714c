"""Generate tests/golden/geometry_depth_to_world.npz by running the REFERENCE function (build container only).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_geometry.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
from src.models.utils.geometry import depth_to_world_coords_points  # noqa: E402

rng = np.random.Generator(np.random.Philox(key=[2024, 5]))
B, H, W = 3, 37, 52
depth = (0.2 + 5.0 * rng.random((B, H, W), dtype=np.float32)).astype(np.float32)
depth[rng.random((B, H, W)) < 0.07] = 0.0          # invalid pixels
depth[0, 0, 0] = 5e-9                               # below eps
ext = np.zeros((B, 4, 4), np.float32)
for i in range(B):
    q = rng.standard_normal(4); q /= np.linalg.norm(q)
    x, y, z, w = q
    ext[i, :3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
    ext[i, :3, 3] = rng.standard_normal(3) * 3
    ext[i, 3, 3] = 1
K = np.zeros((B, 3, 3), np.float32)
K[:, 0, 0] = W * (0.8 + 0.4 * rng.random(B)); K[:, 1, 1] = H * (0.8 + 0.4 * rng.random(B))
K[:, 0, 2] = W / 2 + rng.standard_normal(B); K[:, 1, 2] = H / 2 + rng.standard_normal(B); K[:, 2, 2] = 1
with torch.no_grad():
    world, cam, mask = depth_to_world_coords_points(torch.from_numpy(depth), torch.from_numpy(ext), torch.from_numpy(K))
out = os.path.join(ROOT, "tests", "golden", "geometry_depth_to_world.npz")
np.savez_compressed(out, depth=depth, extrinsic=ext, intrinsic=K, world=world.numpy(), cam=cam.numpy(), mask=mask.numpy())
print("wrote", out, os.path.getsize(out) // 1024, "KiB")

# ---- create_confidence_mask (infer.py:25-59); infer.py itself imports heavy optional packages, so the function's
# source text is executed from the file (build container only) instead of importing the module
import ast, types  # noqa: E402
src = open("/root/reference/infer.py").read()
tree = ast.parse(src)
fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "create_confidence_mask"][0]
mod = types.ModuleType("ref_infer_part")
mod.__dict__.update({"torch": torch, "np": np})
exec(compile(ast.Module(body=[fn], type_ignores=[]), "/root/reference/infer.py", "exec"), mod.__dict__)
conf = (1.0 + 9.0 * rng.random((2, 61, 47), dtype=np.float32)).astype(np.float32)   # continuous: no ties at the threshold
conf[rng.random(conf.shape) < 0.1] = 0.0                                                # invalid (<= 1e-5 -> -inf), 10 % < the 30 % dropped
store = {"conf": conf}
for pct in (30.0, 0.0, 55.5, 99.99):
    with torch.no_grad():
        m = mod.create_confidence_mask(torch.from_numpy(conf), pct)
    store[f"mask_{pct}"] = m.numpy()
out2 = os.path.join(ROOT, "tests", "golden", "geometry_confidence_mask.npz")
np.savez_compressed(out2, **store)
print("wrote", out2, os.path.getsize(out2) // 1024, "KiB", {k: int(v.sum()) for k, v in store.items() if k != "conf"})

