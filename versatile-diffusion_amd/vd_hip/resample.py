"""Host-side tables for the on-device CLIP image pre-processing (`vd_clip_preprocess_f16`).

The device kernels only do integer multiply-adds; everything that Pillow / transformers compute in double or float32
on the host is tabulated here with the same operation order (python floats are IEEE doubles), so the result is
bit-exact with the reference's host path (lib/model_zoo/clip.py:88-94 there):
  * `pil_bicubic_taps(in, out)`: Pillow ImagingResample's 8-bit coefficients for one axis (Keys bicubic a = -0.5, support
    2 * max(in/out, 1), taps normalised and quantised to 22 fractional bits, round half away from zero);
  * `resize_output_size(h, w)`: transformers' shortest-edge rule, long edge = int(size * long / short);
  * `clip_norm_table()`: (level * (1/255) in double -> float32, minus mean, divided by std in float32) per uint8 level.
"""
import functools
import math

import numpy as np

PRECISION_BITS = 22
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
RESCALE_FACTOR = 0.00392156862745098


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@functools.lru_cache(maxsize=64)
def pil_bicubic_taps(in_size, out_size):
    """-> (bounds int32 [out, 2] = (first input index, taps), kk int32 [out, ksize], ksize)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    inv = 1.0 / filterscale
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    one = float(1 << PRECISION_BITS)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * inv) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * one) if v < 0 else int(0.5 + v * one)
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def resize_output_size(h, w, size):
    short, long = (w, h) if w <= h else (h, w)
    new_long = int(size * long / short)
    return (new_long, size) if w <= h else (size, new_long)


@functools.lru_cache(maxsize=4)
def clip_norm_table():
    lv = (np.arange(256, dtype=np.float64) * RESCALE_FACTOR).astype(np.float32)
    mean, std = np.array(CLIP_MEAN, dtype=np.float32), np.array(CLIP_STD, dtype=np.float32)
    return np.ascontiguousarray(((lv[:, None] - mean[None, :]) / std[None, :]).astype(np.float32))
