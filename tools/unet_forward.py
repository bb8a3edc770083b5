"""N UNet forwards at the bench shape (CFG batch 8, 64x64 latent, L=77); target for rocprofv3 --pmc.

    python tools/unet_forward.py N                 N eager forwards
    python tools/unet_forward.py N time            + eager timing (host-bound below ~14 ms: ~35 us of Python per launch)
    python tools/unet_forward.py N graph [L]       one forward captured into a HIP graph and replayed: device time per
                                                   forward (the number that matters; L = context length, default 77)
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
os.environ.setdefault("VD_QUIET", "1")
import torch
import bench
from vd_hip import ops
dev = torch.device("cuda:0")
net = bench.build_model(dev)
B = int(os.environ.get("VD_FWD_BATCH", "4"))
L = int(sys.argv[3]) if len(sys.argv) > 3 else 77
side = int(os.environ.get("VD_FWD_SIDE", "64"))
x = torch.randn(2 * B, 4, side, side, device=dev, dtype=torch.float16)
t = torch.full((2 * B,), 501, device=dev, dtype=torch.long)
c = torch.randn(2 * B, L, 768, device=dev, dtype=torch.float16) * 0.5
ci = {"type": "text", "c": c, "kv_cache": {}}
mode = sys.argv[2] if len(sys.argv) > 2 else ""
if os.environ.get("VD_FWD_TUNE"):   # "M,N,K,ks,cls,cfg,nsplit;..." pins single problems to an instantiation (A/B of planner rules)
    from vd_hip.loader import lib
    from vd_hip import tune
    tune.ensure_loaded()
    for ent in os.environ["VD_FWD_TUNE"].split(";"):
        assert lib().vd_gemm_tune_set(*[int(v) for v in ent.split(",")]) == 0
with torch.no_grad():
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
        net.apply_model({"type": "image", "x": x}, t, ci)
    torch.cuda.synchronize()
    if mode == "time":
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                net.apply_model({"type": "image", "x": x}, t, ci)
            e1.record()
            torch.cuda.synchronize()
            print("eager forward ms: %.3f" % (e0.elapsed_time(e1) / 10))
    if mode == "graph":
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                out = net.apply_model({"type": "image", "x": x}, t, ci)
        torch.cuda.current_stream().wait_stream(s)
        ops.drop_workspaces(s.cuda_stream)
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            print("graph forward ms: %.3f" % (e0.elapsed_time(e1) / 20))
