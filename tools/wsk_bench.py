"""8x8-level 3x3 convolution: the whole-K kernel (conv_wsk_kernel.h) against the split kernel + reduce launch, back to back in one
process (VD_WSK is read per call), eager launches timed with HIP events and the same pair replayed from a HIP graph."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
os.environ.setdefault("VD_QUIET", "1")
import torch
from vd_hip import ops
from vd_hip.pack import pack_conv_weight, pack_conv_weight_stream
dev = torch.device("cuda:0")


def graph_time(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    torch.cuda.current_stream().wait_stream(s)
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for (B, cin, co) in ((8, 1280, 1280), (8, 2560, 1280), (4, 1280, 1280), (16, 1280, 1280)):
    gen = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, 8, 8, cin, device=dev, dtype=torch.float16, generator=gen)
    wt = torch.randn(co, cin, 3, 3, device=dev, dtype=torch.float16, generator=gen) * 0.02
    b = torch.randn(co, device=dev, dtype=torch.float16, generator=gen)
    wp, wsm = pack_conv_weight(wt), pack_conv_weight_stream(wt)
    fn = lambda: ops.conv2d_nhwc(x, wp, b, ksize=3, pad=1, w_stream=wsm, want_stats=True)
    res = {}
    for mode in ("0", "1"):
        os.environ["VD_WSK"] = mode
        res[mode] = graph_time(fn)
    fl = 2.0 * B * 64 * co * 9 * cin
    print("B=%-2d %4d->%4d  split+reduce %6.1f us (%4.0f TF/s)   whole-K %6.1f us (%4.0f TF/s)" % (
        B, cin, co, res["0"], fl / res["0"] / 1e6, res["1"], fl / res["1"] / 1e6))
