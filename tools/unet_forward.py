"""N UNet forwards at the bench shape (CFG batch 8, 64x64 latent, L=77); target for rocprofv3 --pmc."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
os.environ.setdefault("VD_QUIET", "1")
import torch
import bench
dev = torch.device("cuda:0")
net = bench.build_model(dev)
B = 4
x = torch.randn(2 * B, 4, 64, 64, device=dev, dtype=torch.float16)
t = torch.full((2 * B,), 501, device=dev, dtype=torch.long)
c = torch.randn(2 * B, 77, 768, device=dev, dtype=torch.float16) * 0.5
ci = {"type": "text", "c": c, "kv_cache": {}}
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    net.apply_model({"type": "image", "x": x}, t, ci)
torch.cuda.synchronize()
if len(sys.argv) > 2 and sys.argv[2] == "time":
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            net.apply_model({"type": "image", "x": x}, t, ci)
        e1.record()
        torch.cuda.synchronize()
        print("forward ms: %.3f" % (e0.elapsed_time(e1) / 10))
if len(sys.argv) > 2 and sys.argv[2] == "sustain":
    for n in (10, 50, 100):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            net.apply_model({"type": "image", "x": x}, t, ci)
        e1.record()
        torch.cuda.synchronize()
        print("n=%d forward ms: %.3f" % (n, e0.elapsed_time(e1) / n))
