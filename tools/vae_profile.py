"""Per-shape kernel table of one kl-f8 decode (B=4, 64x64 latent -> 512x512) (dev tool; run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
os.environ.setdefault("VD_QUIET", "1")
import torch
import bench
from vd_hip import ops
dev = torch.device("cuda:0")
net = bench.build_model(dev)
z = torch.randn(4, 4, 64, 64, device=dev, dtype=torch.float16)
for _ in range(2):
    net.vae_decode(z, which="image")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); net.vae_decode(z, which="image"); e1.record(); torch.cuda.synchronize()
print("decode B=4: %.2f ms" % e0.elapsed_time(e1))
ops.PROFILE_SHAPES = True
agg = {}
ops.profile_begin()
net.vae_decode(z, which="image")
for name, fl, by, ms in ops.profile_end():
    a = agg.setdefault(name, [0, 0.0, 0.0]); a[0] += 1; a[1] += fl; a[2] += ms
tot = sum(a[2] for a in agg.values())
print("instrumented total %.3f ms" % tot)
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][2])[:25]:
    print("%7.3f ms %5.1f%% n=%3d avg=%8.1f us %7.1f TF/s  %s" % (a[2], 100 * a[2] / tot, a[0], 1e3 * a[2] / a[0], a[1] / a[2] / 1e9 if a[2] else 0, name))
