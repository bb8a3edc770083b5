"""Per-shape kernel table of one UNet forward at the benchmark shape (dev tool; run on the GPU box)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
os.environ.setdefault("VD_QUIET", "1")
import torch
import bench
from vd_hip import ops

dev = torch.device("cuda:0")
net = bench.build_model(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
MODE = sys.argv[2] if len(sys.argv) > 2 else "t2i"   # per-GPU shares of BASELINE configs[1..4]: t2i | i2v (image context 257) | dual
#   (text 77 + image 257, mixed 0.5 / 0.5) | triple (96x96 latents: text 77 + two masked images 257 + 257 -> 514)
SIDE = 96 if MODE == "triple" else 64
x = torch.randn(2 * B, 4, SIDE, SIDE, device=dev, dtype=torch.float16)
t = torch.full((2 * B,), 501, device=dev, dtype=torch.long)
c = torch.randn(2 * B, 77, 768, device=dev, dtype=torch.float16) * 0.5
ci = torch.randn(2 * B, 514 if MODE == "triple" else 257, 768, device=dev, dtype=torch.float16) * 0.5
_kv = [{}, {}]


def forward():
    if MODE in ("dual", "triple"):
        return net.apply_model_multicontext({"type": "image", "x": x}, t, [{"type": "text", "c": c, "ratio": 0.5, "kv_cache": _kv[0]},
                                                                           {"type": "image", "c": ci, "ratio": 0.5, "kv_cache": _kv[1]}])
    if MODE == "i2v":
        return net.apply_model({"type": "image", "x": x}, t, {"type": "image", "c": ci, "kv_cache": _kv[1]})
    return net.apply_model({"type": "image", "x": x}, t, {"type": "text", "c": c})


for _ in range(2):
    forward()
ops.PROFILE_SHAPES = True
agg = {}
reps = 5
for _ in range(reps):
    ops.profile_begin()
    forward()
    for name, fl, by, ms in ops.profile_end():
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1; a[1] += fl; a[2] += ms
rows = sorted(agg.items(), key=lambda kv: -kv[1][2])
tot = sum(a[2] for _, a in rows) / reps
print("total %.3f ms" % tot)
for name, a in rows:
    n = a[0] // reps
    print("%7.3f ms %5.1f%% n=%3d avg=%7.1f us %7.1f TF/s  %s" % (a[2] / reps, 100 * a[2] / reps / tot, n, 1e3 * a[2] / a[0], a[1] / a[2] / 1e9 if a[2] else 0, name))
