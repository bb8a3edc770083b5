"""One-off parity evidence too slow for the GPU suite (parity soft spot: "the only 50-step guided loop is at a
32x32 latent"): BASELINE configs[1]'s OWN geometry end to end -- full-width vd_four_flow_v1-0 UNet (synthetic weights, seed 7,
the suite's `full` fixture), 64x64x4 latent, L = 77 text context, CFG 7.5, all 50 DDIM steps, B = 1 (CFG batch 2) -- HIP
path (graph-replayed sampler loop through the C ABI) against the fp32 CPU oracle (oracle/vd_oracle.py), with the error
trajectory every 10 steps.  The oracle's 50 forwards take ~4 minutes on 16 host threads.

    python tools/probes/parity_50step_fullres.py [out.txt]            on the GPU box
    python tools/probes/parity_50step_fullres.py --dual --steps 20 [out.txt]   BASELINE configs[3]'s geometry: text (L = 77) +
                                                   image (L = 257) context mixing 0.5 / 0.5 through sample_multicontext
    python tools/probes/parity_50step_fullres.py --dry                oracle side only, 2 steps at 16x16 (CPU container)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "versatile-diffusion_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("VD_QUIET", "1")
import numpy as np
import torch

from vdtest_util import full_vd_cfg, rel_l2, synth_into
from oracle import vd_oracle as O

DRY = "--dry" in sys.argv
DUAL = "--dual" in sys.argv
_args = [a for a in sys.argv[1:]]
_steps = int(_args[_args.index("--steps") + 1]) if "--steps" in _args else 50
OUT = next((a for i, a in enumerate(_args) if not a.startswith("--") and (i == 0 or _args[i - 1] != "--steps")), None)
STEPS, SIDE, EVERY = (2, 16, 1) if DRY else (_steps, 64, 10 if _steps >= 50 else 5)
LATENT_TOL = 1e-2   # north star: <= 1e-2 rel-L2 vs reference latents


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    if OUT:
        with open(OUT, "a") as f:
            f.write(s + "\n")


def main():
    from lib.model_zoo import get_model
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    net = get_model()(full_vd_cfg(with_vae=False), verbose=False)
    sd = synth_into(net, 7)
    g = torch.Generator().manual_seed(2024)
    xT = torch.randn((1, 4, SIDE, SIDE), generator=g)
    c = torch.randn((1, 77, 768), generator=g) * 0.5
    u = torch.randn((1, 77, 768), generator=g) * 0.5
    ci = torch.randn((1, 257, 768), generator=g) * 0.5
    say("full-resolution loop parity (%s): steps %d, latent %dx%d, B = 1 (CFG batch 2), guidance 7.5, eta 0" % (
        "dual context: text 77 + image 257, ratios 0.5 / 0.5" if DUAL else "text context", STEPS, SIDE, SIDE))

    z = inter = None
    if not DRY:
        from lib.model_zoo.ddim import DDIMSampler
        from vd_hip.loader import lib_digest
        dev = torch.device("cuda:0")
        net = net.half()
        net.to(dev)
        sampler = DDIMSampler(net)
        t0 = time.time()
        h = lambda t: t.half().to(dev)
        ct = {"type": "text", "conditioning": h(c), "unconditional_conditioning": h(u), "unconditional_guidance_scale": 7.5}
        if DUAL:
            ct["ratio"] = 0.5
            cim = {"type": "image", "conditioning": h(ci), "unconditional_conditioning": h(torch.zeros_like(ci)),
                   "unconditional_guidance_scale": 7.5, "ratio": 0.5}
            z, inter = sampler.sample_multicontext(steps=STEPS, shape=[1, 4, SIDE, SIDE], x_info={"type": "image", "xt": h(xT)},
                                                   c_info_list=[ct, cim], eta=0., verbose=False, log_every_t=EVERY)
        else:
            z, inter = sampler.sample(steps=STEPS, shape=[1, 4, SIDE, SIDE], x_info={"type": "image", "xt": h(xT)},
                                      c_info=ct, eta=0., verbose=False, log_every_t=EVERY)
        torch.cuda.synchronize()
        say("HIP path: library %s, graph loop %s, %.2f s (first call: capture included), %d logged latents" % (
            lib_digest(), sampler.use_graph, time.time() - t0, len(inter["pred_xt"])))
        z = z.float().cpu()
        inter = [t.float().cpu() for t in inter["pred_xt"]]

    # the oracle's loop, step by step (O.ddim_sample's body), keeping the latents the sampler logs
    sched = O.ddim_schedule(sd["alphas_cumprod"], STEPS, 0.0)
    ts = sched["timesteps"]
    ctx = [{"type": "text", "conditioning": c, "unconditional_conditioning": u}]
    if DUAL:
        ctx = [{"type": "text", "conditioning": c, "unconditional_conditioning": u, "ratio": 0.5},
               {"type": "image", "conditioning": ci, "unconditional_conditioning": torch.zeros_like(ci), "ratio": 0.5}]
    x = xT
    logged = []
    t0 = time.time()
    with torch.no_grad():
        for i, step in enumerate(np.flip(ts)):
            index = ts.shape[0] - i - 1
            x, _ = O.p_sample_ddim(sd, O.unet_plan(), sched, x, ctx, index, step, 7.5, "image", "image")
            if index % EVERY == 0 or index == ts.shape[0] - 1:
                logged.append((index, x.clone()))
                if inter is not None:
                    k = len(logged) - 1
                    say("  after DDIM index %2d (t = %3d): rel-L2 %.3e   |x| %.3f   oracle %.0f s" % (
                        index, int(step), rel_l2(inter[k], x), float(x.norm()), time.time() - t0))
                else:
                    say("  oracle index %d t = %d |x| %.4f" % (index, int(step), float(x.norm())))
    if DRY:
        say("dry run ok")
        return
    assert len(inter) == len(logged), (len(inter), len(logged))
    err = rel_l2(z, x)
    say("final latent rel-L2 vs fp32 oracle after %d DDIM timesteps: %.3e (tolerance %.0e) -> %s" % (int(ts.shape[0]), err, LATENT_TOL, "PASS" if err < LATENT_TOL else "FAIL"))
    sys.exit(0 if err < LATENT_TOL else 1)


if __name__ == "__main__":
    main()
