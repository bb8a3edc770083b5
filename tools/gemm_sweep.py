"""Sweep split-K factors for the UNet's GEMM / conv shapes under the tile config forced by VD_GEMM_TILE (dev tool).

usage: VD_GEMM_TILE=<0|1|2> python tools/gemm_sweep.py      (unset = planner's own tile choice)
prints: shape, then time per split factor (us); 'p' marks the planner's own choice (split_k=0).
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
import torch
from vd_hip import ops
from vd_hip.pack import pack_conv_weight

dev = torch.device("cuda:0")
SHAPES = [  # (B, H, W, Cin, Cout, ksize)
    (8, 16, 16, 1280, 1280, 3), (8, 64, 64, 320, 320, 3), (8, 32, 32, 640, 640, 3), (8, 8, 8, 1280, 1280, 3),
    (8, 16, 16, 2560, 1280, 3), (8, 64, 64, 640, 320, 3), (8, 32, 32, 1280, 640, 3), (8, 32, 32, 1920, 640, 3),
    (8, 64, 64, 960, 320, 3), (8, 16, 16, 1920, 1280, 3), (8, 32, 32, 960, 640, 3), (8, 8, 8, 2560, 1280, 3),
    (8, 16, 16, 640, 1280, 3), (8, 32, 32, 320, 640, 3),
    (8, 16, 16, 5120, 1280, 1), (8, 16, 16, 1280, 1280, 1), (8, 32, 32, 2560, 640, 1), (8, 64, 64, 1280, 320, 1),
    (8, 32, 32, 640, 640, 1), (8, 64, 64, 320, 320, 1), (8, 8, 8, 5120, 1280, 1),
    (8, 64, 64, 320, 960, 1), (8, 64, 64, 320, 640, 1), (8, 64, 64, 640, 640, 3),
    (4, 512, 512, 128, 128, 3), (4, 256, 256, 256, 256, 3), (4, 128, 128, 512, 512, 3), (4, 512, 512, 256, 128, 3),
    (1, 64, 64, 4096, 4096, 1), (1, 64, 128, 8192, 8192, 1),
]
SPLITS = [int(v) for v in os.environ.get("VD_SWEEP_SPLITS", "0,1,2,3,4,5,6,8,10,12,16").split(",")]


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print("tile override:", os.environ.get("VD_GEMM_TILE", "planner"))
print("%-40s" % "shape (M N K)", " ".join("%7s" % ("p" if s == 0 else "s%d" % s) for s in SPLITS))
FILT = os.environ.get("VD_SWEEP_FILTER")  # e.g. N=320
for (B, H, W, Ci, Co, ks) in SHAPES:
    if FILT and ("N=%d" % Co) != FILT:
        continue
    x = torch.randn(B, H, W, Ci, device=dev, dtype=torch.float16)
    wt = torch.randn(Co, Ci, ks, ks, device=dev, dtype=torch.float16) * 0.02
    w = pack_conv_weight(wt) if ks == 3 else wt.reshape(Co, Ci).contiguous()
    b = torch.randn(Co, device=dev, dtype=torch.float16)
    M, K = B * H * W, Ci * ks * ks
    row = []
    for s in SPLITS:
        if s > 1 and K // 64 // s < 4:
            row.append("      -")
            continue
        try:
            if ks == 3:
                us = timeit(lambda: ops.conv2d_nhwc(x, w, b, ksize=3, pad=1, split_k=s))
            else:
                us = timeit(lambda: ops.gemm(x.view(M, Ci), w, bias=b, split_k=s))
            row.append("%7.1f" % us)
        except Exception as e:  # noqa
            row.append("    err")
    fl = 2.0 * M * Co * K
    best = min(float(v) for v in row if v.strip() not in ("-", "err"))
    print("%-40s" % ("conv%d M=%d N=%d K=%d" % (ks, M, Co, K)), " ".join(row), "  best %.0f TF/s" % (fl / best / 1e6))
