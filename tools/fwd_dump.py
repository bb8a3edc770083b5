"""One UNet forward at the bench shape with seeded inputs / weights, output saved to argv[1] (dev tool: A/B of kernel
builds and switches for CORRECTNESS: python tools/fwd_dump.py a.pt; VD_LN_FOLD=0 python tools/fwd_dump.py b.pt;
python tools/fwd_dump.py --cmp a.pt b.pt)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
os.environ.setdefault("VD_QUIET", "1")
import torch
if sys.argv[1] == "--cmp":
    a, b = torch.load(sys.argv[2]).float(), torch.load(sys.argv[3]).float()
    print("%s vs %s: rel-L2 %.3e  max|a| %.3f" % (sys.argv[2], sys.argv[3], float((a - b).norm() / b.norm()), float(a.abs().max())))
    sys.exit(0)
import bench
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = bench.build_model(dev)
B = int(os.environ.get("VD_FWD_BATCH", "4"))
side = int(os.environ.get("VD_FWD_SIDE", "64"))
g = torch.Generator(device="cpu").manual_seed(5)
x = torch.randn(2 * B, 4, side, side, generator=g).half().to(dev)
t = torch.full((2 * B,), 501, device=dev, dtype=torch.long)
c = (torch.randn(2 * B, 77, 768, generator=g) * 0.5).half().to(dev)
if os.environ.get("VD_FWD_OVERRIDE"):
    from vd_hip import ops
    ops.gemm_set_override(int(os.environ["VD_FWD_OVERRIDE"]))
with torch.no_grad():
    out = net.apply_model({"type": "image", "x": x}, t, {"type": "text", "c": c, "kv_cache": {}})
torch.cuda.synchronize()
torch.save(out.cpu(), sys.argv[1])
print(sys.argv[1], "finite", bool(torch.isfinite(out).all()), "norm %.4f" % float(out.float().norm()))
