"""One 3x3 conv shape a few times (target for rocprofv3 --pmc). usage: one_conv.py B H W Cin Cout"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
import torch
from vd_hip import ops
from vd_hip.pack import pack_conv_weight
dev = torch.device("cuda:0")
B, H, W, Ci, Co = [int(v) for v in sys.argv[1:6]]
x = torch.randn(B, H, W, Ci, device=dev, dtype=torch.float16)
w = pack_conv_weight(torch.randn(Co, Ci, 3, 3, device=dev, dtype=torch.float16) * 0.02)
b = torch.randn(Co, device=dev, dtype=torch.float16)
for _ in range(5):
    ops.conv2d_nhwc(x, w, b, ksize=3, pad=1)
torch.cuda.synchronize()
