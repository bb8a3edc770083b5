import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
import torch
from vd_hip import ops
from attn_bench import timeit
dev = torch.device("cuda:0")
for (M, N, K) in [(32768, 320, 320), (32768, 960, 320), (8192, 640, 640), (2048, 1280, 1280), (32768, 320, 1280)]:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.02
    b = torch.randn(N, device=dev, dtype=torch.float16)
    r = torch.randn(M, N, device=dev, dtype=torch.float16)
    t_pure = timeit(lambda: ops.gemm(a, w), 50)
    t_bias = timeit(lambda: ops.gemm(a, w, bias=b), 50)
    t_full = timeit(lambda: ops.gemm(a, w, bias=b, res=r), 50)
    t_ns = timeit(lambda: ops.gemm(a, w, bias=b, res=r, split_k=1), 50)
    print("M=%d N=%d K=%d tile=%s: pure %.1f us, +bias %.1f, +bias+res %.1f, nosplit %.1f" % (M, N, K, os.environ.get("VD_GEMM_TILE", "auto"), t_pure * 1e3, t_bias * 1e3, t_full * 1e3, t_ns * 1e3))
# empty-kernel launch floor
x = torch.zeros(256, device=dev, dtype=torch.float16)
print("axpby tiny: %.1f us" % (timeit(lambda: ops.axpby(x, x, 1.0, 1.0), 200) * 1e3))
