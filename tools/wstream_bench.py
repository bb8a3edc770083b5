"""vd_conv3x3_wstream_f16 in isolation: time per launch as a function of the chunk count per block (split factor) -> the
per-chunk steady-state time of the main loop and the fixed cost of a launch.

    python tools/wstream_bench.py [variant]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
os.environ.setdefault("VD_QUIET", "1")
import torch
from vd_hip import ops
from vd_hip.loader import lib
from vd_hip.pack import pack_conv_weight, pack_conv_weight_stream

dev = torch.device("cuda:0")
var = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lib().vd_conv3x3_wstream_set_variant(var, 256)
g = torch.Generator(device=dev).manual_seed(0)
for B, C, N in ((8, 1280, 1280), (8, 2560, 1280), (16, 1280, 1280)):
    x = torch.randn((B, 8, 8, C), device=dev, generator=g).half()
    wt = (torch.randn((N, C, 3, 3), device=dev, generator=g) * 0.03).half()
    b = torch.zeros((N,), device=dev).half()
    wp, wsm = pack_conv_weight(wt), pack_conv_weight_stream(wt)
    nch = C // 64
    for split in (1, 2, 4, 5, 10, 20):
        if split > nch:
            continue
        for _ in range(3):
            ops.conv2d_nhwc(x, wp, b, ksize=3, pad=1, w_stream=wsm, split_k=split)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.conv2d_nhwc(x, wp, b, ksize=3, pad=1, w_stream=wsm, split_k=split)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        tiles = (B // 2) * (N // 256)
        print("B=%d C=%d N=%d split=%2d: %7.1f us per launch (+reduce)  blocks=%3d chunks/block=%2d -> %.2f us per chunk" % (
            B, C, N, split, us, tiles * split, nch // split, us / (nch // split)))
