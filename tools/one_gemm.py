"""Run one GEMM shape a few times (target for rocprofv3 --pmc)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
import torch
from vd_hip import ops
dev = torch.device("cuda:0")
M, N, K = [int(v) for v in sys.argv[1:4]]
a = torch.randn(M, K, device=dev, dtype=torch.float16)
w = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.02
for _ in range(5):
    ops.gemm(a, w)
torch.cuda.synchronize()
