"""dev: where does the device generator go during DDIMSampler.sample (eta = 0, graph path)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("VD_QUIET", "1")
import torch
dev = torch.device("cuda:0")
def st():
    s = torch.cuda.get_rng_state(dev)
    return int.from_bytes(bytes(s[:8].tolist()), "little"), int.from_bytes(bytes(s[8:16].tolist()), "little")
torch.manual_seed(77); print("seeded", st())
x = torch.randn([2, 4, 16, 16], device=dev, dtype=torch.float16); print("after randn latent", st())
for i in range(6):
    torch.randn_like(x)
print("after 6 randn_like", st())
print(torch.randn(8, device=dev)[:3].tolist())
# graph capture effect
torch.manual_seed(77); print("seeded", st())
g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
y = torch.zeros(8, device=dev)
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        y.add_(1)
torch.cuda.current_stream().wait_stream(s); print("after capture", st())
g.replay(); print("after replay", st())
from vdtest_util import load_gold, meta, synth_into, tiny_vd_cfg
from lib.model_zoo import get_model
from lib.model_zoo.ddim import DDIMSampler
m = meta(); net = get_model()(tiny_vd_cfg(m), verbose=False); synth_into(net, m["seed"]); net = net.half(); net.to(dev)
gold = load_gold("ddim_tiny.npz")
T = lambda a: torch.from_numpy(a).to(dev).half()
ct = {"type": "text", "conditioning": T(gold["c_text"]), "unconditional_conditioning": T(gold["u_text"]), "unconditional_guidance_scale": 7.5}
for eta in (0.0, 0.5):
    torch.manual_seed(77); print("eta", eta, "seeded", st())
    DDIMSampler(net).sample(steps=6, shape=[2, 4, 16, 16], x_info={"type": "image"}, c_info=dict(ct), eta=eta, verbose=False)
    print("after sample", st()); print(torch.randn(8, device=dev)[:3].tolist())
