"""Generate tests/golden/*.npz by running the REFERENCE's own modules (imported read-only from /root/reference)
in this CPU container.  TEST INFRASTRUCTURE ONLY -- run by hand:

    python oracle/gen_golden.py

The reference ships no tests, golden vectors or fixtures for this path, so these files are
the pin: reference code + seeded synthetic weights (oracle/synth.py) + seeded inputs -> stored outputs.  The
fixtures carry the configs and seeds, so weights are regenerated on the GPU box instead of being shipped.
"""
import contextlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import refshim, synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

UNET2D_TINY = dict(in_channels=4, out_channels=4, model_channels=64, attention_resolutions=[1, 2],
                   num_res_blocks=[1, 1], channel_mult=[1, 2], num_heads=None, num_head_channels=64, context_dim=128,
                   use_checkpoint=True, parts=["global", "data", "context"])
UNET0D_TINY = dict(input_channels=128, model_channels=64, output_channels=128, num_noattn_blocks=[1, 1],
                   channel_mult=[1, 2], second_dim=[4, 4], with_attn=[True, True], num_heads=None,
                   num_head_channels=64, context_dim=128, use_checkpoint=True, parts=["data", "context"])
VAE_TINY = dict(embed_dim=4, lossconfig=None,
                ddconfig=dict(double_z=True, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=64,
                              ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[], dropout=0.0))
CLIP_TINY = dict(
    text_config=dict(vocab_size=120, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1,
                     max_position_embeddings=24, hidden_act="quick_gelu", projection_dim=64, eos_token_id=2,
                     bos_token_id=0, pad_token_id=1),
    vision_config=dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                       image_size=224, patch_size=14, hidden_act="quick_gelu", projection_dim=64),
    projection_dim=64)
SEED = 1234


def seeded(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


@contextlib.contextmanager
def injected_randn(queue):
    """The reference draws its latents/noise with torch.randn / randn_like; feed it ours instead."""
    real, real_like = torch.randn, torch.randn_like

    def fake(*size, **kw):
        shape = tuple(size[0]) if (len(size) == 1 and not isinstance(size[0], int)) else tuple(size)
        if queue and tuple(queue[0].shape) == shape:
            return queue.pop(0).clone().to(kw.get("dtype") or torch.float32)
        return real(*size, **kw)

    def fake_like(t, **kw):
        if queue and tuple(queue[0].shape) == tuple(t.shape):
            return queue.pop(0).clone().to(t.dtype)
        return real_like(t, **kw)

    torch.randn, torch.randn_like = fake, fake_like
    try:
        yield
    finally:
        torch.randn, torch.randn_like = real, real_like


def build_ref_vd(ref):
    E = ref.edict
    cfg = E(type="vd_v2_0", args=E(
        vae_cfg_list=[["image", E(type="autoencoderkl", args=E(VAE_TINY))]],
        ctx_cfg_list=[["image", "ctx-image-placeholder"], ["text", "ctx-text-placeholder"]],
        diffuser_cfg_list=[["image", E(type="openai_unet_2d_next", args=E(UNET2D_TINY))],
                           ["text", E(type="openai_unet_0d_next", args=E(UNET0D_TINY))]],
        global_layer_ptr="image", latent_scale_factor=E(image=0.18215),
        beta_linear_start=0.00085, beta_linear_end=0.012, timesteps=1000, use_ema=False))
    net = ref.get_model()(cfg, verbose=False)
    synth.load_synth_(net, SEED)
    net.eval()
    return net


def main():
    os.makedirs(GOLD, exist_ok=True)
    ref = refshim.load_reference()
    torch.manual_seed(0)
    net = build_ref_vd(ref)
    meta = dict(seed=SEED, unet2d=UNET2D_TINY, unet0d=UNET0D_TINY, vae=VAE_TINY, clip=CLIP_TINY)

    # ---- A. schedules -----------------------------------------------------------------------
    out = {k: getattr(net, k).numpy() for k in
           ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
            "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
            "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"]}
    sampler = ref.ddim.DDIMSampler(net)
    for steps in (50, 10, 5, 4):
        sampler.make_schedule(ddim_num_steps=steps, ddim_eta=0.0, verbose=False)
        out["ddim%d_timesteps" % steps] = np.asarray(sampler.ddim_timesteps)
        out["ddim%d_alphas" % steps] = np.asarray(sampler.ddim_alphas, dtype=np.float32)
        out["ddim%d_alphas_prev" % steps] = np.asarray(sampler.ddim_alphas_prev, dtype=np.float64)
        out["ddim%d_sigmas" % steps] = np.asarray(sampler.ddim_sigmas, dtype=np.float32)
        out["ddim%d_sqrt_one_minus_alphas" % steps] = np.asarray(sampler.ddim_sqrt_one_minus_alphas, dtype=np.float32)
    sampler.make_schedule(ddim_num_steps=10, ddim_eta=0.7, verbose=False)
    out["ddim10_eta07_sigmas"] = np.asarray(sampler.ddim_sigmas, dtype=np.float32)
    tt = torch.tensor([981, 1, 500, 21, 0, 999])
    out["temb_t"] = tt.numpy()
    out["temb_320"] = ref.diffusion_utils.timestep_embedding(tt, 320).numpy()
    out["temb_64"] = ref.diffusion_utils.timestep_embedding(tt, 64).numpy()
    np.savez_compressed(os.path.join(GOLD, "schedule.npz"), **out)

    # ---- B. tiny UNet forwards (VD_v2_0.apply_model / apply_model_multicontext) ----------------
    x = seeded((2, 4, 16, 16), 1)
    t = torch.tensor([981, 401])
    c_text = seeded((2, 77, 128), 2, 0.5)
    c_img = seeded((2, 20, 128), 3, 0.5)
    with torch.no_grad():
        e_text = net.apply_model({"type": "image", "x": x}, t, {"type": "text", "c": c_text})
        e_img = net.apply_model({"type": "image", "x": x}, t, {"type": "image", "c": c_img})
        e_mix = net.apply_model_multicontext({"type": "image", "x": x}, t, [
            {"type": "text", "c": c_text, "ratio": 0.4}, {"type": "image", "c": c_img, "ratio": 0.6}])
    np.savez_compressed(os.path.join(GOLD, "unet_tiny.npz"), x=x.numpy(), t=t.numpy(), c_text=c_text.numpy(), c_img=c_img.numpy(),
             eps_text=e_text.numpy(), eps_image=e_img.numpy(), eps_mix=e_mix.numpy(),
             state_keys=np.array(sorted(net.state_dict().keys())),
             state_shapes=np.array([json.dumps(list(net.state_dict()[k].shape)) for k in sorted(net.state_dict().keys())]))

    # ---- B2. text-latent (0-D) data flow: data blocks of diffuser['text'], context blocks of the context's type -------
    x0d = seeded((2, 128), 21)
    with torch.no_grad():
        e0_img = net.apply_model({"type": "text", "x": x0d}, t, {"type": "image", "c": c_img})
        e0_text = net.apply_model({"type": "text", "x": x0d}, t, {"type": "text", "c": c_text})
    # (apply_model_multicontext takes time_embed from diffuser[x_type] (vd.py:415-417); the 'text' diffuser is built
    #  without global layers, so the reference itself cannot run a multi-context text flow)
    np.savez_compressed(os.path.join(GOLD, "unet0d_tiny.npz"), x=x0d.numpy(), t=t.numpy(), c_text=c_text.numpy(),
                        c_img=c_img.numpy(), eps_image=e0_img.numpy(), eps_text=e0_text.numpy())

    # ---- C. DDIM loops ------------------------------------------------------------------------
    net.device = "cpu"
    sampler = ref.ddim.DDIMSampler(net)
    xT = seeded((2, 4, 16, 16), 4)
    u_text = seeded((1, 77, 128), 5, 0.5).repeat(2, 1, 1)
    u_img = torch.zeros_like(c_img)
    with injected_randn([xT]):
        z_t2i, inter = sampler.sample(steps=5, shape=[2, 4, 16, 16], x_info={"type": "image"},
                                      c_info={"type": "text", "conditioning": c_text, "unconditional_conditioning": u_text,
                                              "unconditional_guidance_scale": 7.5}, eta=0.0, verbose=False)
    with injected_randn([xT]):
        z_mc, _ = sampler.sample_multicontext(
            steps=4, shape=[2, 4, 16, 16], x_info={"type": "image"},
            c_info_list=[{"type": "text", "conditioning": c_text, "unconditional_conditioning": u_text,
                          "unconditional_guidance_scale": 5.0, "ratio": 0.4},
                         {"type": "image", "conditioning": c_img, "unconditional_conditioning": u_img,
                          "unconditional_guidance_scale": 5.0, "ratio": 0.6}], eta=0.0, verbose=False)
    # image-variation style start: q_sample(x0) at DDIM index 3 of 5, then 3 steps (ddim.py:97-103), guidance 1
    x0 = seeded((2, 4, 16, 16), 6)
    qn = seeded((2, 4, 16, 16), 7)
    with injected_randn([qn]):
        z_i2i, _ = sampler.sample(steps=5, shape=[2, 4, 16, 16],
                                  x_info={"type": "image", "x0": x0, "x0_forward_timesteps": 3},
                                  c_info={"type": "image", "conditioning": c_img, "unconditional_conditioning": u_img,
                                          "unconditional_guidance_scale": 1.0}, eta=0.0, verbose=False)
    np.savez_compressed(os.path.join(GOLD, "ddim_tiny.npz"), xT=xT.numpy(), c_text=c_text.numpy(), u_text=u_text.numpy(),
             c_img=c_img.numpy(), u_img=u_img.numpy(), z_t2i=z_t2i.numpy(), pred_x0_t2i=inter["pred_x0"][-1].numpy(),
             z_mc=z_mc.numpy(), x0=x0.numpy(), q_noise=qn.numpy(), z_i2i=z_i2i.numpy())

    # ---- D. tiny VAE ----------------------------------------------------------------------------
    img = torch.rand((2, 3, 32, 32), generator=torch.Generator().manual_seed(8))
    vae = net.vae["image"]
    with torch.no_grad():
        post = vae.encode(img, out_posterior=True)
        pn = seeded(tuple(post.mean.shape), 9)
        with injected_randn([pn]):
            z = net.vae_encode(img, which="image")
        dec = net.vae_decode(z, which="image")
        zlat = seeded((2, 4, 4, 4), 10)
        dec2 = net.vae_decode(zlat, which="image")
    np.savez_compressed(os.path.join(GOLD, "vae_tiny.npz"), img=img.numpy(), moments=post.parameters.numpy(), post_noise=pn.numpy(),
             z=z.numpy(), dec=dec.numpy(), zlat=zlat.numpy(), dec2=dec2.numpy())

    # ---- E. tiny CLIP through the reference's encoder classes -------------------------------------
    from transformers import CLIPConfig, CLIPModel
    torch.manual_seed(0)
    clip = CLIPModel(CLIPConfig(**CLIP_TINY)).eval()
    synth.load_synth_(clip, SEED, prefix="ctx.text.model.")
    ids = torch.randint(3, 119, (3, 24), generator=torch.Generator().manual_seed(11))
    eos = [23, 9, 15]
    for b, e in enumerate(eos):
        ids[b, e] = 119  # largest id marks EOS (HF 4.24 pooled = argmax(input_ids))
        ids[b, e + 1:] = 1
    px = seeded((3, 3, 224, 224), 12)  # regenerated from the seed by the tests, not stored

    class FakeTok:
        def __call__(self, text, **kw):
            return {"input_ids": ids}

    class FakeProc:
        def __call__(self, images=None, **kw):
            return {"pixel_values": px}

    tenc = object.__new__(ref.clip.CLIPTextContextEncoder)
    torch.nn.Module.__init__(tenc)
    tenc.tokenizer, tenc.model, tenc.max_length, tenc.fp16 = FakeTok(), clip, 24, False
    ienc = object.__new__(ref.clip.CLIPImageContextEncoder)
    torch.nn.Module.__init__(ienc)
    ienc.processor, ienc.model, ienc.fp16 = FakeProc(), clip, False
    # transformers >= 4.3x calls embeddings.forward(pixel_values, interpolate_pos_encoding=...): accept + ignore
    emb = clip.vision_model.embeddings
    import types
    orig_setattr = type(emb).__setattr__
    masks = (torch.rand((3, 1, 64, 64), generator=torch.Generator().manual_seed(13)) > 0.4).float()
    with torch.no_grad():
        z_text = tenc.encode(["a", "b", "c"])
        z_img = ienc.encode([None, None, None])
        # masked path: wrap the reference's 1-arg replacement so the newer HF call signature is accepted
        real_method_type = types.MethodType

        def tolerant_method(fn, obj):
            def wrapped(self, pixel_values, *a, **k):
                return fn(self, pixel_values)
            return real_method_type(wrapped, obj)

        types.MethodType = tolerant_method
        try:
            z_img_masked = ienc.encode(torch.zeros(3, 3, 8, 8), masks)
        finally:
            types.MethodType = real_method_type
    np.savez_compressed(os.path.join(GOLD, "clip_tiny.npz"), input_ids=ids.numpy(), px_seed=np.array(12), masks=masks.numpy(),
             z_text=z_text.numpy(), z_img=z_img.numpy(), z_img_masked=z_img_masked.numpy())

    with open(os.path.join(GOLD, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    for fn in sorted(os.listdir(GOLD)):
        print("%-18s %8d bytes" % (fn, os.path.getsize(os.path.join(GOLD, fn))))


if __name__ == "__main__":
    refshim.load_reference()
    with refshim.reference_cwd():  # the reference does call-time relative imports (vd.py:337)
        main()
