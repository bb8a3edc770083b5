"""Attention / GEMM micro-benchmarks at the UNet's shapes (dev tool; run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
import torch
from vd_hip import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def attn_cases():
    for (B, H, Nq, Nk, D) in [(8, 8, 4096, 4096, 40), (8, 8, 9216, 9216, 40), (4, 8, 4096, 4096, 40), (16, 8, 4096, 4096, 40), (8, 8, 1024, 1024, 80), (8, 8, 256, 256, 160), (8, 8, 4096, 77, 40),
                              (8, 8, 1024, 77, 80), (8, 12, 77, 77, 64), (8, 16, 257, 257, 64)]:
        C = H * D
        qkv = torch.randn(B, Nq, 3 * C, device=dev, dtype=torch.float16)
        k = torch.randn(B, Nk, C, device=dev, dtype=torch.float16)
        v = torch.randn(B, Nk, C, device=dev, dtype=torch.float16)
        q = qkv[..., :C]
        if Nq == Nk:
            k, v = qkv[..., C:2 * C], qkv[..., 2 * C:]
        ms = timeit(lambda: ops.attention(q, k, v, H))
        fl = 4.0 * B * Nq * Nk * C
        print("attn B=%d H=%d Nq=%d Nk=%d D=%d: %8.1f us  %7.1f TF/s" % (B, H, Nq, Nk, D, ms * 1e3, fl / ms / 1e9))


def gemm_cases():
    from vd_hip.pack import pack_conv_weight
    shapes = [  # (B,H,W,Cin,Cout,ks)  convs
        (8, 64, 64, 320, 320, 3), (8, 32, 32, 640, 640, 3), (8, 16, 16, 1280, 1280, 3), (8, 8, 8, 1280, 1280, 3),
        (8, 16, 16, 2560, 1280, 3), (8, 64, 64, 640, 320, 3), (8, 32, 32, 1280, 640, 3)]
    for (B, H, W, Ci, Co, ks) in shapes:
        x = torch.randn(B, H, W, Ci, device=dev, dtype=torch.float16)
        w = pack_conv_weight(torch.randn(Co, Ci, ks, ks, device=dev, dtype=torch.float16) * 0.02)
        b = torch.randn(Co, device=dev, dtype=torch.float16)
        ms = timeit(lambda: ops.conv2d_nhwc(x, w, b, ksize=ks, pad=ks // 2))
        fl = 2.0 * B * H * W * Co * Ci * ks * ks
        print("conv%d B=%d %dx%d %d->%d: %8.1f us  %7.1f TF/s" % (ks, B, H, W, Ci, Co, ms * 1e3, fl / ms / 1e9))
    lin = [(32768, 320, 320), (32768, 960, 320), (32768, 2560, 320), (32768, 320, 1280), (8192, 640, 640), (8192, 5120, 640),
           (8192, 640, 2560), (2048, 1280, 1280), (2048, 10240, 1280), (2048, 1280, 5120), (8192, 8192, 8192), (4096, 4096, 4096)]
    from vd_hip.pack import pack_geglu
    for (M, N, K) in lin:
        a = torch.randn(M, K, device=dev, dtype=torch.float16)
        w = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.02
        b = torch.randn(N, device=dev, dtype=torch.float16)
        geglu = N in (2560, 5120, 10240)
        if geglu:
            wp, bp = pack_geglu(w, b)
            ms = timeit(lambda: ops.gemm(a, wp, bias=bp, act=ops.ACT_GEGLU))
        else:
            r = torch.randn(M, N, device=dev, dtype=torch.float16)
            ms = timeit(lambda: ops.gemm(a, w, bias=b, res=r))
        print("gemm M=%d N=%d K=%d%s: %8.1f us  %7.1f TF/s" % (M, N, K, " geglu" if geglu else "", ms * 1e3, 2.0 * M * N * K / ms / 1e9))


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("attn", "all"):
        attn_cases()
    if which in ("gemm", "all"):
        gemm_cases()
