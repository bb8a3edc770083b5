"""vd_ff_geglu_f16 at the UNet's 64x64-level shape a few times (target for rocprofv3 --pmc)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
import torch
from vd_hip import ops
dev = torch.device("cuda:0")
C, M = 320, 32768
x = (torch.randn(M, C, device=dev) * 1.2).half()
wp, bp = (torch.randn(8 * C, C, device=dev) * 0.05).half(), (torch.randn(8 * C, device=dev) * 0.2).half()
w2, b2 = (torch.randn(C, 4 * C, device=dev) * 0.03).half(), (torch.randn(C, device=dev) * 0.2).half()
for _ in range(5):
    ops.ff_geglu(x, wp, bp, w2, b2, x, 1e-5)
torch.cuda.synchronize()
