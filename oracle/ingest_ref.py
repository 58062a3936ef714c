"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's image ingest
(src/utils/inference_utils.py:14-149 load_and_preprocess_images), from the decoded uint8 RGB image on.

The resize the reference calls (inference_utils.py:86, PIL ``Image.resize(..., Image.Resampling.BICUBIC)``) lives in a
third-party dependency, Pillow (12.2.0 in this image; libImaging/Resample.c).  Its published algorithm is restated
here in numpy integer arithmetic — separable two-pass convolution, support scaled by the down-sampling factor,
double-precision Keys bicubic (a = -0.5) coefficients normalised per output pixel and quantised to 22 fractional bits,
uint8 intermediate after the horizontal pass — and pinned bit-for-bit against Pillow itself
(tests/test_ingest.py::test_oracle_resize_equals_pillow and tests/golden/ingest_*.npz written by
oracle/gen_golden_ingest.py).  The reference function as a whole cannot be imported here (torchvision is absent), so
the glue around the resize (target size, centre crop, white padding, /255) is restated from its source lines.
Only tests/ may import this module."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc: per output index (first input index, count, int32 weights)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size)
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """One 8-bit pass along `axis` (0 = vertical, 1 = horizontal) of an [H][W][C] uint8 image."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.uint8)
    for i in range(bounds.shape[0]):
        x0, n = bounds[i]
        acc = np.tensordot(kk[i, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bicubic_u8(img, out_w, out_h):
    """PIL Image.resize((out_w, out_h), BICUBIC) on an [H][W][3] uint8 array: horizontal pass, then vertical."""
    h, w = img.shape[:2]
    out = img
    if out_w != w:
        b, k = resample_coeffs(w, out_w)
        out = _pass(out, b, k, 1)
    if out_h != h:
        b, k = resample_coeffs(h, out_h)
        out = _pass(out, b, k, 0)
    return out


def target_size(w, h, mode="crop", output_size=518):
    """inference_utils.py:70-83 (Python round = half to even)."""
    if mode == "pad":
        if w >= h:
            return output_size, round(h * (output_size / w) / 14) * 14
        return round(w * (output_size / h) / 14) * 14, output_size
    return output_size, round(h * (output_size / w) / 14) * 14


def preprocess_rgb(img, mode="crop", output_size=518):
    """One decoded RGB uint8 image -> float32 [3][H'][W'] in [0, 1] (inference_utils.py:67-108)."""
    h, w = img.shape[:2]
    sw, sh = target_size(w, h, mode, output_size)
    r = resize_bicubic_u8(img, sw, sh)
    t = np.ascontiguousarray(r.transpose(2, 0, 1)).astype(np.float32) / np.float32(255.0)      # ToTensor (:87)
    if mode == "crop" and sh > output_size:                                                     # :90-92
        y0 = (sh - output_size) // 2
        t = t[:, y0:y0 + output_size, :]
    if mode == "pad":                                                                           # :95-108
        ph, pw = output_size - t.shape[1], output_size - t.shape[2]
        if ph > 0 or pw > 0:
            top, left = ph // 2, pw // 2
            t = np.pad(t, ((0, 0), (top, ph - top), (left, pw - left)), constant_values=1.0)
    return t
