"""Image ingest: load_and_preprocess_images from the decoded uint8 image on.

CPU: the oracle's restatement of Pillow's 8-bit BICUBIC resample against Pillow's own outputs (committed fixtures, and
live against the installed Pillow when it is importable).  GPU: the HIP path through the C ABI must equal the oracle
BIT FOR BIT (integer resize, exact /255), for both preprocessing modes, with and without the centre crop / white pad."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD
from oracle import ingest_ref as R


def _gold():
    return np.load(os.path.join(GOLD, "ingest_pillow_bicubic.npz"))


def test_oracle_resize_equals_pillow_fixtures():
    z = _gold()
    i = 0
    while f"in{i}" in z:
        ow, oh = (int(v) for v in z[f"size{i}"])
        assert np.array_equal(R.resize_bicubic_u8(z[f"in{i}"], ow, oh), z[f"out{i}"]), i
        i += 1
    assert i == 5


def test_oracle_resize_equals_pillow_live():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    for (h, w, ow, oh) in [(41, 63, 518, 336), (300, 200, 70, 98), (20, 20, 20, 20), (90, 160, 160, 90)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img, "RGB").resize((ow, oh), Image.Resampling.BICUBIC))
        assert np.array_equal(R.resize_bicubic_u8(img, ow, oh), ref), (h, w, ow, oh)


def test_oracle_target_sizes():
    assert R.target_size(1920, 1080) == (518, 294)          # 16:9 -> 518 x 294 (21 patch rows)
    assert R.target_size(1080, 1920) == (518, 924)          # portrait: cropped to 518 afterwards
    assert R.target_size(1080, 1920, "pad") == (294, 518)
    assert R.target_size(518, 518, "pad") == (518, 518)
    assert R.preprocess_rgb(np.zeros((192, 108, 3), np.uint8)).shape == (3, 518, 518)
    assert R.preprocess_rgb(np.zeros((108, 192, 3), np.uint8), "pad").shape == (3, 518, 518)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,mode", [(97, 131, "crop"), (131, 97, "crop"), (97, 131, "pad"), (131, 97, "pad"), (518, 518, "crop"),
                                      (1080, 1920, "crop"), (600, 400, "pad"), (37, 1000, "crop")])
def test_gpu_preprocess_equals_oracle_exactly(h, w, mode):
    from hunyuanworld_mirror_amd import preprocess_rgb
    rng = np.random.default_rng(h * 7 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    img[::5, ::3] = 255
    want = R.preprocess_rgb(img, mode)
    got = preprocess_rgb(torch.from_numpy(img).cuda(), mode).cpu().numpy()
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.array_equal(got, want), f"max diff {np.abs(got - want).max()}"


@pytest.mark.gpu
def test_gpu_pillow_fixtures_through_the_kernels():
    """The committed Pillow outputs themselves, through the device resize (crop-mode geometry: width 70, 5 patch rows)."""
    from hunyuanworld_mirror_amd import preprocess_rgb
    z = _gold()
    img = z["in2"]                      # 97 x 131 -> Pillow resized it to 70 x 56
    got = preprocess_rgb(torch.from_numpy(img).cuda(), "crop", output_size=70).cpu().numpy()
    want = z["out2"].transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.gpu
def test_gpu_load_and_preprocess_images_files(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    from hunyuanworld_mirror_amd import load_and_preprocess_images
    rng = np.random.default_rng(11)
    paths, want = [], []
    for i, (h, w) in enumerate([(120, 160), (160, 120)]):   # landscape + portrait: different heights -> white-padded to the taller
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        p = str(tmp_path / f"im{i}.png")
        Image.fromarray(img, "RGB").save(p)
        paths.append(p)
        want.append(R.preprocess_rgb(img, "crop"))
    rgba = rng.integers(0, 256, (50, 70, 4), dtype=np.uint8)
    p = str(tmp_path / "im_rgba.png")
    Image.fromarray(rgba, "RGBA").save(p)
    out = load_and_preprocess_images(paths, "crop")
    assert out.shape == (1, 2, 3, 518, 518) and out.device.type == "cuda"
    mh = max(t.shape[1] for t in want)
    for i, t in enumerate(want):
        ph = mh - t.shape[1]
        top = ph // 2
        t = np.pad(t, ((0, 0), (top, ph - top), (0, 0)), constant_values=1.0)
        assert np.array_equal(out[0, i].cpu().numpy(), t), i
    one = load_and_preprocess_images([p], "pad")            # RGBA is composited onto white first (inference_utils.py:58-63)
    white = Image.new("RGBA", (70, 50), (255, 255, 255, 255))
    comp = np.asarray(Image.alpha_composite(white, Image.fromarray(rgba, "RGBA")).convert("RGB"))
    assert np.array_equal(one[0, 0].cpu().numpy(), R.preprocess_rgb(comp, "pad"))
    with pytest.raises(ValueError):
        load_and_preprocess_images([])
    with pytest.raises(ValueError):
        load_and_preprocess_images(paths, "stretch")
