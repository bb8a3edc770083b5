"""AutoencoderKL (KL-f8) inference path with the reference's surface (lib/model_zoo/autokl.py:14-49 there):
`encode(x)` -> sampled latent, `decode(z)` -> image in [0,1]; NCHW in / NCHW out, HIP kernels in between.
Training-only pieces of the reference (loss, discriminator, optimisers; autokl.py:57-140) are out of scope."""
import torch
import torch.nn as nn

from vd_hip import ops

from .autokl_modules import Decoder, Encoder
from .common.get_model import register
from .distributions import DiagonalGaussianDistribution
from .hip_layers import Conv2d


def _img16(x):
    if not x.is_cuda:
        raise RuntimeError("AutoencoderKL needs GPU tensors: this package has no CPU path")
    return x.to(torch.float16).contiguous()


@register("autoencoderkl")
class AutoencoderKL(nn.Module):
    def __init__(self, ddconfig, lossconfig, embed_dim):
        super().__init__()
        assert lossconfig is None, "the LPIPS/discriminator loss is training-only and not part of this package"
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        assert ddconfig["double_z"]
        self.quant_conv = Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim

    def _posterior(self, x):
        h = self.encoder(_img16(x), in_scale=2.0, in_shift=-1.0)  # x*2-1 folded into the first conv's gather
        return DiagonalGaussianDistribution(self.quant_conv(h))

    @torch.no_grad()
    def encode(self, x, out_posterior=False, noise=None):
        post = self._posterior(x)
        return post if out_posterior else post.sample(noise).to(x.dtype)

    @torch.no_grad()
    def encode_scaled(self, x, scale, noise=None):
        """scale * encode(x) with the latent scale folded into the sampling kernel (VD_v2_0.vae_encode)."""
        return self._posterior(x).sample(noise, scale=scale).to(x.dtype)

    @torch.no_grad()
    def decode(self, z):
        return self.decode_scaled(z, 1.0)

    @torch.no_grad()
    def decode_scaled(self, z, inv_scale):
        """decode(inv_scale * z); the scale rides in the first 1x1 conv's gather, clamp((x+1)/2) in the final
        NHWC->NCHW store."""
        h = self.post_quant_conv(_img16(z), in_layout="nchw", in_scale=float(inv_scale))
        dec = self.decoder(h)
        return ops.nhwc_to_nchw(dec, scale=0.5, shift=0.5, clamp01=True).to(z.dtype)
