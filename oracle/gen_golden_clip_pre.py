"""Pin oracle/clip_preprocess.py against the installed Pillow + transformers CLIP image processor and write
tests/golden/clip_pre.npz.  Run in the CPU container:  python oracle/gen_golden_clip_pre.py

Inputs are regenerated from seeds by the tests (smooth random fields, so they are not stored); the fixture holds the
uint8 image after resize + centre crop for each case (full for the first, an 80x80 corner + sha256 for the others) and
a strip of the float32 pixel_values."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import clip_preprocess as CP  # noqa: E402

CASES = [(0, 512, 512), (1, 300, 420), (2, 640, 333), (3, 224, 224), (4, 97, 160)]  # (seed, H, W)


def make_image(seed, h, w):
    """Deterministic smooth-ish field in [0, 1], [3, H, W] float32 (low-res noise upsampled + fine noise)."""
    rs = np.random.RandomState(seed)
    low = rs.rand(3, (h + 15) // 16 + 1, (w + 15) // 16 + 1).astype(np.float32)
    up = np.kron(low, np.ones((16, 16), dtype=np.float32))[:, :h, :w]
    img = 0.8 * up + 0.2 * rs.rand(3, h, w).astype(np.float32)
    return np.clip(img, 0.0, 1.0).astype(np.float32)


def reference_path(img01):
    """What the reference does (clip.py:88-94) with the installed third-party packages."""
    from PIL import Image
    from transformers import CLIPImageProcessor
    u8 = (img01 * np.float32(255.0)).astype(np.uint8).transpose(1, 2, 0)  # ToPILImage: mul(255).byte()
    pil = Image.fromarray(u8, mode="RGB")
    proc = CLIPImageProcessor()
    pv = proc(images=[pil], return_tensors="np")["pixel_values"][0]
    # the uint8 image the processor saw after resize + crop, recovered through Pillow directly
    nh, nw = CP.resize_output_size(u8.shape[0], u8.shape[1])
    res = np.asarray(pil.resize((nw, nh), resample=Image.BICUBIC))
    return CP.center_crop(res), pv.astype(np.float32)


def main():
    import PIL
    import transformers
    out, meta = {}, {"pillow": PIL.__version__, "transformers": transformers.__version__, "cases": CASES}
    for seed, h, w in CASES:
        img = make_image(seed, h, w)
        ref_u8, ref_pv = reference_path(img)
        ora_u8, ora_pv = CP.clip_preprocess(img)
        assert np.array_equal(ref_u8, ora_u8), "oracle differs from Pillow at case %s" % ((seed, h, w),)
        assert np.array_equal(ref_pv, ora_pv), "oracle differs from the HF processor at case %s: max %g" % (
            (seed, h, w), np.abs(ref_pv - ora_pv).max())
        key = "c%d" % seed
        out[key + "_u8"] = ref_u8 if seed == 0 else ref_u8[:80, :80]
        out[key + "_pv"] = ref_pv[:, 100:104, :]
        meta[key + "_sha256"] = hashlib.sha256(ref_u8.tobytes()).hexdigest()
        print("case", (seed, h, w), "bit-exact vs Pillow + transformers")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "clip_pre.npz"), **out)
    with open(os.path.join(ROOT, "tests", "golden", "clip_pre_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
