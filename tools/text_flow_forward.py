"""Time the text-latent (0-D) flow forward of vd_four_flow at CFG batch 8 (image context L=257); dev tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
os.environ.setdefault("VD_QUIET", "1")
import torch
from lib.cfg_helper import CfgDict, model_cfg_bank
from lib.model_zoo import get_model
from vd_hip import ops

dev = torch.device("cuda:0")
bank = model_cfg_bank()
cfg = CfgDict(type="vd_v2_0", args=CfgDict(
    vae_cfg_list=[], ctx_cfg_list=[["image", "ctx-image-placeholder"], ["text", "ctx-text-placeholder"]],
    diffuser_cfg_list=[["image", bank("openai_unet_2d_v1")],
                       ["text", bank("openai_unet_0d_v1_dc")]],
    global_layer_ptr="image", latent_scale_factor={"image": 0.18215}, beta_linear_start=0.00085,
    beta_linear_end=0.012, timesteps=1000, use_ema=False))
with torch.device(dev):
    net = get_model()(cfg, verbose=False)
for p in net.parameters():
    if p.dim() > 1:
        torch.nn.init.normal_(p, std=0.02)
net = net.half()
net.to(dev)
B = 8
x = torch.randn(B, 768, device=dev, dtype=torch.float16)
t = torch.full((B,), 501, device=dev, dtype=torch.long)
c = torch.randn(B, 257, 768, device=dev, dtype=torch.float16) * 0.5
ci = {"type": "image", "c": c, "kv_cache": {}}
for _ in range(3):
    net.apply_model({"type": "text", "x": x}, t, ci)
ops.profile_begin()
net.apply_model({"type": "text", "x": x}, t, ci)
rec = ops.profile_end()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    net.apply_model({"type": "text", "x": x}, t, ci)
e1.record()
torch.cuda.synchronize()
nparam = sum(p.numel() for p in net.diffuser["text"].parameters())
ms = e0.elapsed_time(e1) / 10
print("text-latent forward (B=%d): %.3f ms; weights %.2f GB -> %.2f TB/s effective weight stream" % (B, ms, nparam * 2 / 1e9, nparam * 2 / ms / 1e9))
ops.PROFILE_SHAPES = True
ops.profile_begin()
net.apply_model({"type": "text", "x": x}, t, ci)
rec = ops.profile_end()
agg = {}
for name, fl, by, ms_k in rec:
    a = agg.setdefault(name, [0, 0.0, 0.0])
    a[0] += 1; a[1] += ms_k; a[2] += by
tot = sum(v[1] for v in agg.values())
print("instrumented total %.3f ms over %d launches" % (tot, len(rec)))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %7.3f ms n=%3d avg %7.1f us  %6.2f TB/s  %s" % (v[1], v[0], v[1] / v[0] * 1e3, v[2] / max(v[1], 1e-9) / 1e9, k))

# graph replay (what the DDIM loop runs): GPU time without the per-launch host cost
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        out = net.apply_model({"type": "text", "x": x}, t, ci)
torch.cuda.current_stream().wait_stream(s)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    g.replay()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("graph replay: %.3f ms per forward -> %.2f TB/s effective weight stream" % (ms, nparam * 2 / ms / 1e9))
