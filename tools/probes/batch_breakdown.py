"""Development probe (round 6): where one bench batch (BASELINE configs[1]: 4 images, 50 guided DDIM steps, decode) spends its time
outside the 50 replayed UNet steps -- host-side phases of vd_sample_sharded timed with synchronize() markers.

    python tools/probes/batch_breakdown.py [workload]
"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench as B

wlname = sys.argv[1] if len(sys.argv) > 1 else "t2i"
dev = torch.device("cuda:0")
net = B.build_model(dev)
from lib.model_zoo.ddim import DDIMSampler
sampler = DDIMSampler(net)
wl = B.WORKLOADS[wlname]
per_gpu = B.default_per_gpu(wl)
_, ctxs, images = B.workload_inputs(wl, per_gpu, 1, 0, dev)
marks = []


def timed(obj, name, label):
    f = getattr(obj, name)

    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); marks.append((label, 1e3 * (time.perf_counter() - t0)))
        return r
    setattr(obj, name, w)


timed(net, "vae_decode", "vae_decode")
if hasattr(net, "vae_encode"):
    timed(net, "vae_encode", "vae_encode")
timed(sampler, "sample", "sampler.sample")
timed(sampler, "sample_multicontext", "sampler.sample_multicontext")
from lib.model_zoo import sharded
_draw = sharded.draw_initial_latent
def draw(*a, **k):
    t0 = time.perf_counter(); r = _draw(*a, **k); marks.append(("draw x_T (host)", 1e3 * (time.perf_counter() - t0))); return r
sharded.draw_initial_latent = draw
# inside the sampler: the loop itself
timed(sampler, "_loop_static", "  sampler._loop_static")
for rep in range(4):
    del marks[:]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    B.one_batch(net, sampler, wl, ctxs, per_gpu, 50, rep, images=images)
    torch.cuda.synchronize(); tot = 1e3 * (time.perf_counter() - t0)
    print("batch %d: %.2f ms total; %s" % (rep, tot, "; ".join("%s %.2f" % m for m in marks)))
