"""TEST INFRASTRUCTURE: seeded RANDOM cases through the live REFERENCE (tiny VD_v2_0 of oracle/gen_golden.py, synthetic
weights): apply_model with a text / image context, apply_model_multicontext with random ratios, ragged geometries, and
guided / unguided / multi-context / partial-schedule DDIM loops with the start latent injected, KL-f8 encode / decode at
random image sizes, the text-latent (0-D) flow.  Writes an .npz with the
inputs and the reference outputs; the CPU test replays the inputs through the oracle.  Separate process (the reference
package is also called `lib`); needs /root/reference.      usage: python oracle/ref_live_cases.py out.npz seed n_cases"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden as G, refshim  # noqa: E402


def main():
    out_path, seed, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    ref = refshim.load_reference()
    with refshim.reference_cwd():   # the reference's `lib` package visible for its lazy registry imports
        torch.manual_seed(0)
        net = G.build_ref_vd(ref)
        net.device = "cpu"
        rng = np.random.RandomState(seed)
        out = {"n": np.int64(n)}
        for k in range(n):
            B = int(rng.randint(1, 4))
            H, W = int(rng.choice([8, 16, 24])), int(rng.choice([8, 16, 24]))
            Lt, Li = int(rng.randint(3, 90)), int(rng.randint(3, 40))
            x = G.seeded((B, 4, H, W), 100 + k)
            t = torch.from_numpy(rng.randint(0, 1000, size=B)).long()
            ct, ci = G.seeded((B, Lt, 128), 200 + k, 0.5), G.seeded((B, Li, 128), 300 + k, 0.5)
            r = float(rng.uniform(0.1, 0.9))
            with torch.no_grad():
                e_t = net.apply_model({"type": "image", "x": x}, t, {"type": "text", "c": ct})
                e_i = net.apply_model({"type": "image", "x": x}, t, {"type": "image", "c": ci})
                e_m = net.apply_model_multicontext({"type": "image", "x": x}, t, [
                    {"type": "text", "c": ct, "ratio": r}, {"type": "image", "c": ci, "ratio": 1.0 - r}])
            # DDIM: steps / scale / flavour at random; start latent (or forward-process noise) injected
            steps = int(rng.choice([4, 5, 6, 7, 10]))
            scale = float(rng.choice([1.0, 3.0, 7.5]))
            flavour = k % 3                           # 0 single text ctx, 1 two contexts, 2 partial schedule from x0
            ut = G.seeded((1, Lt, 128), 400 + k, 0.5).repeat(B, 1, 1)
            ui = torch.zeros_like(ci)
            xT = G.seeded((B, 4, H, W), 500 + k)
            sampler = ref.ddim.DDIMSampler(net)
            c_text = {"type": "text", "conditioning": ct, "unconditional_conditioning": ut, "unconditional_guidance_scale": scale}
            c_img = {"type": "image", "conditioning": ci, "unconditional_conditioning": ui, "unconditional_guidance_scale": scale}
            fwd = -1
            with G.injected_randn([xT]), torch.no_grad():
                if flavour == 0:
                    z, _ = sampler.sample(steps=steps, shape=[B, 4, H, W], x_info={"type": "image"}, c_info=c_text, eta=0.0, verbose=False)
                elif flavour == 1:
                    z, _ = sampler.sample_multicontext(steps=steps, shape=[B, 4, H, W], x_info={"type": "image"},
                                                       c_info_list=[dict(c_text, ratio=r), dict(c_img, ratio=1.0 - r)], eta=0.0, verbose=False)
                else:
                    fwd = int(rng.randint(1, steps))
                    z, _ = sampler.sample(steps=steps, shape=[B, 4, H, W], x_info={"type": "image", "x0": x, "x0_forward_timesteps": fwd},
                                          c_info=c_img, eta=0.0, verbose=False)    # q_sample noise = xT (injected)
            # KL-f8 VAE on a random image size (encode up to the moments, decode of a random latent through vae_decode's
            # 1/scale) and the text-latent (0-D) flow with either context type
            Hi, Wi = int(rng.choice([16, 32, 48])), int(rng.choice([16, 32, 48]))
            img = torch.rand((B, 3, Hi, Wi), generator=torch.Generator().manual_seed(600 + k))
            zl = G.seeded((B, 4, Hi // 2, Wi // 2), 700 + k)
            x0d = G.seeded((B, 128), 800 + k)
            with torch.no_grad():
                mom = net.vae["image"].encode(img, out_posterior=True).parameters
                dec = net.vae_decode(zl, which="image")
                e0_i = net.apply_model({"type": "text", "x": x0d}, t, {"type": "image", "c": ci})
                e0_t = net.apply_model({"type": "text", "x": x0d}, t, {"type": "text", "c": ct})
            for name, v in (("img", img), ("zl", zl), ("mom", mom), ("dec", dec), ("x0d", x0d), ("e0_i", e0_i), ("e0_t", e0_t)):
                out["%d_%s" % (k, name)] = v.numpy()
            for name, v in (("x", x), ("t", t), ("ct", ct), ("ci", ci), ("e_t", e_t), ("e_i", e_i), ("e_m", e_m), ("ut", ut), ("xT", xT), ("z", z)):
                out["%d_%s" % (k, name)] = v.numpy()
            out["%d_meta" % k] = np.array([r, steps, scale, flavour, fwd], dtype=np.float64)
    np.savez_compressed(out_path, **out)


if __name__ == "__main__":
    main()
