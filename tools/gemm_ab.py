"""A few GEMM problems timed back to back (dev tool: A/B of two library builds via VD_HIP_LIB; warm caches, so only effects
that do not depend on the surrounding forward show)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
import torch
from vd_hip import ops
dev = torch.device("cuda:0")
CASES = [(32768, 960, 320, "b"), (8192, 1920, 640, "b"), (2048, 3840, 1280, "b"), (2048, 1280, 5120, "r"), (32768, 320, 1280, "r"),
         (32768, 320, 320, "r"), (8192, 640, 2560, "r"), (32768, 2560, 320, "g")]
if os.environ.get("VD_AB_KSWEEP"):   # one split-K problem at growing K: fixed per-launch cost vs per-K-tile cost
    CASES = [(2048, 1280, k, "s") for k in (1280, 2560, 5120, 10240, 20480)]
if os.environ.get("VD_AB_CASE"):
    CASES = [CASES[int(os.environ["VD_AB_CASE"])]]
for (M, N, K, kind) in CASES:
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn(M, K, device=dev, dtype=torch.float16, generator=g)
    w = torch.randn(N, K, device=dev, dtype=torch.float16, generator=g) * 0.02
    b = torch.randn(N, device=dev, dtype=torch.float16, generator=g)
    r = torch.randn(M, N, device=dev, dtype=torch.float16, generator=g) if kind == "r" else None
    out = torch.empty(M, N // 2 if kind == "g" else N, device=dev, dtype=torch.float16)
    fn = (lambda: ops.gemm(a, w, bias=b, act=ops.ACT_GEGLU, out=out)) if kind == "g" else (lambda: ops.gemm(a, w, bias=b, res=r, out=out))
    if kind == "s":
        fn = lambda: ops.gemm(a, w, bias=b, out=out, split_k=3)
    best = 1e9
    for rep in range(3):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    print("M=%-6d N=%-5d K=%-5d %s  %7.1f us  %6.0f TF/s" % (M, N, K, kind, best, 2.0 * M * N * K / best / 1e6))
