"""vd_ff_geglu_f16 against torch fp32 and timed against the three-launch chain at the UNet's 64x64-level shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
import torch
import torch.nn.functional as F
from vd_hip import ops
from vd_hip.pack import pack_geglu
from lib.model_zoo.hip_layers import fold_layernorm
dev = torch.device("cuda:0")
C = 320
torch.manual_seed(0)
w1, b1 = (torch.randn(8 * C, C, device=dev) * 0.05).half(), (torch.randn(8 * C, device=dev) * 0.2).half()
w2, b2 = (torch.randn(C, 4 * C, device=dev) * 0.03).half(), (torch.randn(C, device=dev) * 0.2).half()
ln = torch.nn.LayerNorm(C, eps=1e-5).to(dev)
with torch.no_grad():
    ln.weight.copy_(1.0 + 0.2 * torch.randn(C, device=dev)); ln.bias.copy_(0.1 * torch.randn(C, device=dev))
wf, bf, _ = fold_layernorm(w1, b1, ln)
wp, bp = pack_geglu(wf, bf)
cs = wp.float().sum(1).contiguous()
for M in (128, 1000, 4136, 32768):
    x = (torch.randn(M, C, device=dev) * 1.2 + 0.3).half()
    xn = F.layer_norm(x.float(), (C,), ln.weight.float(), ln.bias.float(), 1e-5)
    v, g = (xn @ w1.float().t() + b1.float()).chunk(2, dim=-1)
    ref = x.float() + (v * F.gelu(g)) @ w2.float().t() + b2.float()
    out = ops.ff_geglu(x, wp, bp, w2, b2, x, 1e-5)
    chain = ops.linear(ops.linear(x, wp, bp, act=ops.ACT_GEGLU, colsum=cs, ln_eps=1e-5), w2, b2, res=x)
    torch.cuda.synchronize()
    e = lambda a: float((a.float() - ref).norm() / ref.norm())
    print("M=%6d fused rel-L2 %.2e  chain rel-L2 %.2e  finite=%s" % (M, e(out), e(chain), bool(torch.isfinite(out).all())), flush=True)
M = 32768
x = (torch.randn(M, C, device=dev) * 1.2 + 0.3).half()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
gf = (2.0 * M * C * 8 * C + 2.0 * M * 4 * C * C) / 1e9
for rep in range(2):
    tf = timeit(lambda: ops.ff_geglu(x, wp, bp, w2, b2, x, 1e-5))
    tc = timeit(lambda: ops.linear(ops.linear(x, wp, bp, act=ops.ACT_GEGLU, colsum=cs, ln_eps=1e-5), w2, b2, res=x))
    print("M=32768: fused %.1f us (%.0f TF/s)   chain (row_stats + GEGLU GEMM + out GEMM) %.1f us (%.0f TF/s)" % (tf, gf / tf * 1e3, tc, gf / tc * 1e3), flush=True)
