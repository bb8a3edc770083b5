import os, sys
sys.path.insert(0, "/root/repo/versatile-diffusion_amd")
import torch
from vd_hip import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
H, D, N = 8, 40, 4096
C = H * D
for B in (2, 4, 6, 8, 10, 12, 16, 24, 32):
    qkv = torch.randn(B, N, 3 * C, device=dev, dtype=torch.float16)
    ms = timeit(lambda: ops.attention(qkv[..., :C], qkv[..., C:2*C], qkv[..., 2*C:], H))
    print("B=%2d blocks=%4d: %7.1f us  %6.1f TF/s  per 256 blocks %.1f us" % (B, B * H * N // 512, ms * 1e3, 4.0 * B * N * N * C / ms / 1e9, ms * 1e3 / (B * H * N / 512 / 256)))
