"""Ingest kernel timing: one 1920 x 1080 RGB frame -> 518 x 294 (crop mode), on the GPU vs Pillow + numpy on the host."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import preprocess_rgb
img = np.random.default_rng(0).integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
d = torch.from_numpy(img).cuda()
for _ in range(3): preprocess_rgb(d)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): preprocess_rgb(d)
torch.cuda.synchronize()
gpu = (time.perf_counter() - t0) / 20 * 1e3
try:
    from PIL import Image
    t0 = time.perf_counter()
    for _ in range(5):
        r = np.asarray(Image.fromarray(img, "RGB").resize((518, 294), Image.Resampling.BICUBIC)).transpose(2, 0, 1).astype(np.float32) / 255.0
    cpu = (time.perf_counter() - t0) / 5 * 1e3
except ImportError:
    cpu = float("nan")
print(f"1920x1080 -> 518x294: GPU {gpu:.3f} ms per image (incl. output + workspace allocation and the coefficient-table upload), Pillow on the host {cpu:.2f} ms")
