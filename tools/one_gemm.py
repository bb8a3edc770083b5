"""Run one GEMM shape a few times (target for rocprofv3 --pmc).  usage: one_gemm.py M N K [geglu|bias]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
import torch
from vd_hip import ops
from vd_hip.pack import pack_geglu
dev = torch.device("cuda:0")
M, N, K = [int(v) for v in sys.argv[1:4]]
mode = sys.argv[4] if len(sys.argv) > 4 else "pure"
a = torch.randn(M, K, device=dev, dtype=torch.float16)
w = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.02
b = torch.randn(N, device=dev, dtype=torch.float16)
if mode == "geglu":
    w, b = pack_geglu(w, b)
for _ in range(5):
    if mode == "geglu":
        ops.gemm(a, w, bias=b, act=ops.ACT_GEGLU)
    elif mode == "bias":
        ops.gemm(a, w, bias=b)
    else:
        ops.gemm(a, w)
torch.cuda.synchronize()
