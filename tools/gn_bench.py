"""GroupNorm+SiLU timings at the UNet's shapes (dev tool). VD_GN_FUSED=0 selects the two-kernel path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
import torch
from vd_hip import ops
dev = torch.device("cuda:0")


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (B, H, W, C) in [(8, 64, 64, 320), (8, 32, 32, 640), (8, 16, 16, 1280), (8, 8, 8, 1280), (8, 64, 64, 640), (8, 64, 64, 960),
                     (8, 32, 32, 1280), (8, 32, 32, 1920), (8, 16, 16, 2560), (8, 32, 32, 320), (8, 16, 16, 640)]:
    x = torch.randn(B, H, W, C, device=dev, dtype=torch.float16)
    g = torch.randn(C, device=dev, dtype=torch.float16)
    b = torch.randn(C, device=dev, dtype=torch.float16)
    y = torch.empty_like(x)
    us = timeit(lambda: ops.groupnorm_silu(x, g, b, out=y))
    mb = x.numel() * 2 / 1e6
    print("GN B=%d %dx%d C=%d: %7.1f us  (%.0f MB -> %.2f TB/s at read+write once)" % (B, H, W, C, us, mb, 2 * mb / us / 1e6))
