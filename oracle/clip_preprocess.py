"""CPU oracle of the CLIP image pre-processing the reference runs on the host.  TEST INFRASTRUCTURE ONLY.

Path restated (reference lib/model_zoo/clip.py:88-94): tensor [3,H,W] in [0,1] -> torchvision `ToPILImage`
(`pic.mul(255).byte()`: truncation) -> HuggingFace `CLIPProcessor` image branch = PIL bicubic resize of the shortest
edge to 224 (long edge = int(224 * long / short)), centre crop 224x224, rescale by 1/255, normalise by the CLIP
mean / std (float32) -> `.half()` when the encoder runs fp16.

The arithmetic lives in third-party code that is not under /root/reference: Pillow (`ImagingResample`, 8-bit path:
coefficients in double, quantised to 22 fractional bits, horizontal pass then vertical pass with rounding and clipping
to uint8 after each) and transformers' CLIP image processor (reference pins transformers==4.24.0; 5.15 installed here,
Pillow 12.2.0).  This file restates that published algorithm in numpy; `oracle/gen_golden_clip_pre.py` pins it against
the installed Pillow / transformers on seeded images and stores golden vectors in tests/golden/clip_pre.npz.

Only tests/ may import this module."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2  # Pillow Resample.c: 8-bit samples, 2 bits of headroom
CLIP_MEAN = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32)
CLIP_STD = np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32)
RESCALE = 0.00392156862745098  # transformers: rescale_factor = 1 / 255 (python float)


def to_uint8(img01):
    """torchvision ToPILImage on a float tensor: mul(255).byte() -- truncation toward zero, computed in the tensor's
    dtype (fp32 in the reference: `images` arrive as fp32 from the app)."""
    x = np.asarray(img01, dtype=np.float32) * np.float32(255.0)
    return x.astype(np.uint8)  # values are in [0, 255]: the C cast truncates


def bicubic_filter(x, a=-0.5):
    """Pillow Resample.c bicubic_filter (Keys, a = -0.5), support 2."""
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size, support=2.0):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the full-image box: per output sample the first input
    index, the tap count, and int32 taps with PRECISION_BITS fractional bits."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            # normalize_coeffs_8bpc: round half away from zero in double
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resample_pass(img, bounds, kk, axis):
    """One separable pass over a uint8 image [H, W, C] along `axis` (1 = horizontal, 0 = vertical)."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], dtype=np.uint8)
    for o in range(bounds.shape[0]):
        x0, n = int(bounds[o, 0]), int(bounds[o, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for t in range(n):
            acc += src[x0 + t] * int(kk[o, t])
        out[o] = _clip8(acc)
    return np.moveaxis(out, 0, axis)


def resize_output_size(h, w, size=224):
    """transformers get_resize_output_image_size(size=int, default_to_square=False): short edge -> size,
    long edge -> int(size * long / short)."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def pil_resize_bicubic(img_u8, out_h, out_w):
    """Pillow Image.resize(resample=BICUBIC) on an RGB uint8 image [H, W, 3]: horizontal pass, then vertical."""
    h, w = img_u8.shape[:2]
    out = img_u8
    if out_w != w:
        b, k = precompute_coeffs(w, out_w)
        out = resample_pass(out, b, k, axis=1)
    if out_h != h:
        b, k = precompute_coeffs(h, out_h)
        out = resample_pass(out, b, k, axis=0)
    return out


def center_crop(img, size=224):
    """transformers center_crop for images at least `size` large: top = (h - size) // 2, left = (w - size) // 2."""
    h, w = img.shape[:2]
    t, l = (h - size) // 2, (w - size) // 2
    return img[t:t + size, l:l + size]


def normalize(img_u8):
    """transformers image_transforms.rescale / normalize as the PIL backend calls them: the rescale runs in float64 and
    is rounded to float32, the normalisation (x - mean) / std runs in float32 -> [3, H, W]."""
    x = (img_u8.astype(np.float64) * RESCALE).astype(np.float32)
    x = (x - CLIP_MEAN) / CLIP_STD
    return np.ascontiguousarray(x.transpose(2, 0, 1)).astype(np.float32)


def clip_preprocess(img01_chw, size=224):
    """[3, H, W] float in [0, 1] -> (uint8 [size, size, 3] after resize + crop, float32 pixel_values [3, size, size])."""
    u8 = to_uint8(np.asarray(img01_chw).transpose(1, 2, 0))
    h, w = u8.shape[:2]
    nh, nw = resize_output_size(h, w, size)
    u8 = center_crop(pil_resize_bicubic(u8, nh, nw), size)
    return u8, normalize(u8)


def normalize_table():
    """The 256 x 3 float32 values normalize() can produce (one per uint8 level and channel): what a device kernel
    needs to reproduce it bit-exactly without depending on its own division."""
    lv = (np.arange(256, dtype=np.float64) * RESCALE).astype(np.float32)
    return ((lv[:, None] - CLIP_MEAN[None, :]) / CLIP_STD[None, :]).astype(np.float32)
