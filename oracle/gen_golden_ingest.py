"""Generate tests/golden/ingest_*.npz with Pillow (the library the reference calls at inference_utils.py:86), in the
build container:  python oracle/gen_golden_ingest.py
Each fixture: a small synthetic uint8 image, the target size, and Pillow's own BICUBIC resize of it."""
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.Generator(np.random.Philox(key=[7, 7]))
store = {}
for i, (h, w, ow, oh) in enumerate([(60, 80, 37, 29), (50, 70, 112, 84), (97, 131, 70, 56), (33, 33, 66, 70), (64, 48, 64, 20)]):
    # smooth structure + noise so that both interpolation and clipping paths are exercised
    yy, xx = np.mgrid[0:h, 0:w]
    base = 127 + 120 * np.sin(xx / 7.0 + i) * np.cos(yy / 5.0)
    img = np.clip(base[..., None] + rng.normal(0, 40, (h, w, 3)), 0, 255).astype(np.uint8)
    img[::9, ::7] = 255; img[4::11, 3::5] = 0
    out = np.asarray(Image.fromarray(img, "RGB").resize((ow, oh), Image.Resampling.BICUBIC))
    store[f"in{i}"] = img; store[f"size{i}"] = np.array([ow, oh]); store[f"out{i}"] = out
p = os.path.join(ROOT, "tests", "golden", "ingest_pillow_bicubic.npz")
np.savez_compressed(p, **store)
import PIL
print("wrote", p, os.path.getsize(p) // 1024, "KiB with Pillow", PIL.__version__)
