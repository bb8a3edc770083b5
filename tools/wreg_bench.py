"""conv3x3_wreg_kernel against conv3x3_halo_kernel in isolation (same box, warm weights): the 3x3 shapes of the UNet forward.

    python tools/wreg_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
os.environ.setdefault("VD_QUIET", "1")
import torch
from vd_hip import ops
from vd_hip.pack import pack_conv_weight, pack_conv_weight_stream

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
ops.WREG = True
for B, H, C, N, ups in ((8, 64, 320, 320, 0), (8, 64, 640, 320, 0), (8, 64, 960, 320, 0), (8, 32, 640, 640, 0), (8, 32, 640, 640, 1),
                        (8, 16, 1280, 1280, 0), (8, 16, 2560, 1280, 0), (8, 16, 1280, 1280, 1)):
    x = torch.randn((B, H, H, C), device=dev, generator=g).half()
    wt = (torch.randn((N, C, 3, 3), device=dev, generator=g) * 0.03).half()
    b = torch.zeros((N,), device=dev).half()
    Hv = H << ups
    rv = torch.randn((B, N), device=dev, generator=g).half()
    wp, wsm = pack_conv_weight(wt), pack_conv_weight_stream(wt)
    res = {}
    for name, kw in (("halo", {}), ("wreg", {"w_stream": wsm})):
        f = lambda: ops.conv2d_nhwc(x, wp, b, ksize=3, pad=1, ups=ups, rowvec=rv, rows_per_batch=Hv * Hv, want_stats=True, **kw)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) * 1e3 / 20
    gf = 2.0 * B * Hv * Hv * N * 9 * C / 1e9
    print("B=%d %dx%d C=%d N=%d ups=%d: halo %.1f us (%.0f TF/s)  wreg %.1f us (%.0f TF/s)" % (
        B, Hv, Hv, C, N, ups, res["halo"], gf / res["halo"] * 1e-3 * 1e3, res["wreg"], gf / res["wreg"] * 1e-3 * 1e3))
