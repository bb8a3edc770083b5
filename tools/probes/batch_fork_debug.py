"""Development probe: run-to-run identity of the forward with the half-batch branches forced on (VD_BATCH_FORK), and where two
runs differ (which samples = which branch)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from vdtest_util import full_vd_cfg, synth_into
from lib.model_zoo import get_model, vd

dev = torch.device("cuda:0")
net = get_model()(full_vd_cfg(with_vae=False), verbose=False)
synth_into(net, 7)
net = net.half()
net.to(dev)
g = torch.Generator().manual_seed(21)
side = int(os.environ.get("SIDE", "32"))
x = torch.randn((4, 4, side, side), generator=g).half().to(dev)
c = (torch.randn((4, 77, 768), generator=g) * 0.5).half().to(dev)
t = torch.tensor([741, 741, 301, 301], device=dev)
fwd = lambda: net.apply_model({"type": "image", "x": x}, t, {"type": "text", "c": c}).float()
for mode in ("0", "1", "0", "1"):
    vd.BATCH_FORK = mode
    outs = []
    for i in range(4):
        outs.append(fwd()); torch.cuda.synchronize()
    for i in range(1, 4):
        d = (outs[i] - outs[0]).abs()
        print("fork %s run %d vs run 0: max diff %.3e; per sample %s" % (mode, i, d.max().item(), [round(d[b].max().item(), 6) for b in range(4)]))
    if mode == "0":
        ref = outs[0]
    else:
        d = (outs[3] - ref).abs()
        print("fork 1 vs fork 0: rel_l2 %.3e per sample max %s" % (((outs[3] - ref).norm() / ref.norm()).item(), [round(d[b].max().item(), 6) for b in range(4)]))
