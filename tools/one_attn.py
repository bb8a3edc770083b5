import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
import torch
from vd_hip import ops
dev = torch.device("cuda:0")
B, H, N, D = [int(v) for v in sys.argv[1:5]]
C = H * D
qkv = torch.randn(B, N, 3 * C, device=dev, dtype=torch.float16)
for _ in range(5):
    ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], H)
torch.cuda.synchronize()
