"""UNet forward at the bench shape replayed from a HIP graph, once per conv3x3_halo setting, in ONE process (box-to-box
spread is larger than the differences): device ms per forward and the output's distance from setting 0 (every 3x3 conv on
gemm_f16_kernel).

    python tools/halo_forward.py [settings, default "0 2 1 3 5 7 8 -1"]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
os.environ.setdefault("VD_QUIET", "1")
import torch
import bench
from vd_hip import ops
from vd_hip.loader import lib

dev = torch.device("cuda:0")
net = bench.build_model(dev)
B = int(os.environ.get("VD_FWD_BATCH", "4"))
side = int(os.environ.get("VD_FWD_SIDE", "64"))
L = 77
torch.manual_seed(0)
x = torch.randn(2 * B, 4, side, side, device=dev, dtype=torch.float16)
t = torch.full((2 * B,), 501, device=dev, dtype=torch.long)
c = torch.randn(2 * B, L, 768, device=dev, dtype=torch.float16) * 0.5
settings = [int(v) for v in (sys.argv[1].split() if len(sys.argv) > 1 else "0 2 1 3 5 7 8 -1".split())]
base = None
with torch.no_grad():
    for rnd in range(2):
        for sset in settings:
            assert lib().vd_conv_halo_set_variant(sset) == 0
            ci = {"type": "text", "c": c, "kv_cache": {}}
            for _ in range(2):
                out = net.apply_model({"type": "image", "x": x}, t, ci)
            torch.cuda.synchronize()
            ref = out.float().clone()
            if base is None:
                base = ref
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                with torch.cuda.graph(g, stream=s):
                    out = net.apply_model({"type": "image", "x": x}, t, ci)
            torch.cuda.current_stream().wait_stream(s)
            ops.drop_workspaces(s.cuda_stream)
            best = 1e9
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(20):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20)
            err = float((out.float() - base).norm() / base.norm())
            print("round %d setting %2d: graph forward %.3f ms   rel-L2 vs setting %d: %.2e   finite=%s" %
                  (rnd, sset, best, settings[0], err, bool(torch.isfinite(out).all())), flush=True)
            del g
lib().vd_conv_halo_set_variant(-1)
