"""conv3x3_halo_kernel at the 64x64-level shapes, a few timing numbers for the VD_HALO_ABL experiments (dev tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
import torch
from vd_hip import ops
from vd_hip.pack import pack_conv_weight
dev = torch.device("cuda:0")
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for (B, H, W, Ci, Co) in ((8, 64, 64, 320, 320), (8, 64, 64, 640, 320), (8, 32, 32, 640, 640), (8, 16, 16, 1280, 1280)):
    x = torch.randn(B, H, W, Ci, device=dev, dtype=torch.float16)
    w = pack_conv_weight(torch.randn(Co, Ci, 3, 3, device=dev, dtype=torch.float16) * 0.02)
    b = torch.randn(Co, device=dev, dtype=torch.float16)
    res = torch.randn(B, H, W, Co, device=dev, dtype=torch.float16)
    f = lambda: ops.conv2d_nhwc(x, w, b, ksize=3, pad=1, res=res)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    warm = e0.elapsed_time(e1) / 20 * 1e3
    cold = 0.0
    for _ in range(6):
        flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        cold += e0.elapsed_time(e1) / 6 * 1e3
    print("B%d %dx%d %d->%d: warm %.1f us cold %.1f us" % (B, H, W, Ci, Co, warm, cold), flush=True)
