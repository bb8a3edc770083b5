"""DiagonalGaussianDistribution of the KL-VAE posterior (reference lib/model_zoo/distributions.py:24-37).
Parameters are kept channels-last [B, H, W, 2*zc] as the encoder produces them; sampling is one fused kernel
(clamp, exp, FMA with the noise, latent scale) that writes the NCHW latent."""
import torch

from vd_hip import ops


class DiagonalGaussianDistribution(object):
    def __init__(self, moments_nhwc, deterministic=False):
        self.moments = moments_nhwc
        self.deterministic = deterministic
        B, H, W, C2 = moments_nhwc.shape
        self.shape = (B, C2 // 2, H, W)

    @property
    def parameters(self):
        """NCHW moments [B, 2*zc, H, W] like the reference attribute."""
        return ops.nhwc_to_nchw(self.moments)

    def sample(self, noise=None, scale=1.0):
        B, zc, H, W = self.shape
        if self.deterministic:
            noise = None
        elif noise is None:
            # the reference draws the posterior noise with the CPU generator in fp32 (distributions.py:36)
            noise = torch.randn(self.shape)
        if noise is not None:
            noise = noise.to(device=self.moments.device, dtype=torch.float16).contiguous()
        return ops.diag_gaussian_sample(self.moments, noise, B, zc, H, W, scale)

    def mode(self, scale=1.0):
        B, zc, H, W = self.shape
        return ops.diag_gaussian_sample(self.moments, None, B, zc, H, W, scale)
