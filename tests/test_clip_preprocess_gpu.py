"""vd_clip_preprocess_f16 (through the C ABI) vs the oracle: bit-exact, every input kind."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("seed,h,w", [(0, 512, 512), (1, 300, 420), (2, 640, 333), (3, 224, 224), (4, 97, 160), (5, 768, 768)])
def test_preprocess_bit_exact_f32(dev, seed, h, w):
    from oracle import clip_preprocess as CP
    from oracle.gen_golden_clip_pre import make_image
    from vd_hip import ops
    imgs = np.stack([make_image(seed, h, w), make_image(seed + 100, h, w)])
    out = ops.clip_preprocess(torch.from_numpy(imgs).to(dev)).cpu().numpy()
    for b in range(2):
        _, pv = CP.clip_preprocess(imgs[b])
        assert np.array_equal(out[b], pv.astype(np.float16)), "case %s image %d: %d values differ" % (
            (seed, h, w), b, int((out[b] != pv.astype(np.float16)).sum()))


def test_preprocess_u8_and_f16_inputs(dev):
    from oracle import clip_preprocess as CP
    from vd_hip import ops
    rs = np.random.RandomState(9)
    u8 = rs.randint(0, 256, size=(1, 3, 333, 500)).astype(np.uint8)
    out = ops.clip_preprocess(torch.from_numpy(u8).to(dev)).cpu().numpy()[0]
    img = u8[0].transpose(1, 2, 0)
    nh, nw = CP.resize_output_size(333, 500)
    ref = CP.normalize(CP.center_crop(CP.pil_resize_bicubic(img, nh, nw)))
    assert np.array_equal(out, ref.astype(np.float16))
    # fp16 tensors are quantised in fp16, like `pic.mul(255).byte()` on a half tensor
    xh = torch.rand((1, 3, 256, 256), generator=torch.Generator().manual_seed(1)).half()
    lv = (xh * 255).float().numpy().astype(np.uint8)[0].transpose(1, 2, 0)
    ref = CP.normalize(CP.center_crop(CP.pil_resize_bicubic(lv, 224, 224)))
    out = ops.clip_preprocess(xh.to(dev)).cpu().numpy()[0]
    assert np.array_equal(out, ref.astype(np.float16))


def test_encoder_preprocess_uses_device_path(dev):
    from lib.model_zoo.clip import CLIPImageContextEncoder
    from oracle import clip_preprocess as CP
    from oracle.gen_golden_clip_pre import make_image
    cfg = {"text": {"vocab_size": 99, "hidden_size": 64, "intermediate_size": 128, "num_hidden_layers": 1, "num_attention_heads": 2,
                    "max_position_embeddings": 16},
           "vision": {"hidden_size": 64, "intermediate_size": 128, "num_hidden_layers": 1, "num_attention_heads": 2,
                      "image_size": 224, "patch_size": 14}, "projection_dim": 32}
    enc = CLIPImageContextEncoder(config=cfg, fp16=True).half().to(dev)
    img = make_image(3, 400, 520)
    pv = enc.preprocess(torch.from_numpy(img)[None])
    assert pv.shape == (1, 3, 224, 224) and pv.dtype == torch.float16
    assert np.array_equal(pv.cpu().numpy()[0], CP.clip_preprocess(img)[1].astype(np.float16))


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_image_to_u8_matches_topilimage_arithmetic(dev, dtype):
    """ToPILImage on a float tensor is `pic.mul(255).byte()` in the tensor's dtype, then CHW -> HWC."""
    from vd_hip import ops
    x = torch.rand((2, 3, 64, 96), generator=torch.Generator().manual_seed(4)).to(dtype)
    x[0, :, 0, :4] = torch.tensor([0.0, 1.0, 0.5, 254.999 / 255]).to(dtype)
    ref = x.mul(255).byte().permute(0, 2, 3, 1).contiguous()
    out = ops.image_to_u8(x.to(dev)).cpu()
    assert torch.equal(out, ref)
