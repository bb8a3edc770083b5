"""Shared helpers for the parity tests (test infrastructure)."""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_gold(name):
    return {k: v for k, v in np.load(os.path.join(GOLD, name), allow_pickle=False).items()}


def meta():
    with open(os.path.join(GOLD, "meta.json")) as f:
        return json.load(f)


def rel_l2(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def tiny_vd_cfg(m=None):
    from lib.cfg_helper import CfgDict
    m = meta() if m is None else m
    return CfgDict(type="vd_v2_0", args=CfgDict(
        vae_cfg_list=[["image", CfgDict(type="autoencoderkl", args=m["vae"])]],
        ctx_cfg_list=[["image", "ctx-image-placeholder"], ["text", "ctx-text-placeholder"]],
        diffuser_cfg_list=[["image", CfgDict(type="openai_unet_2d_next", args=m["unet2d"])],
                           ["text", CfgDict(type="openai_unet_0d_next", args=m["unet0d"])]],
        global_layer_ptr="image", latent_scale_factor={"image": 0.18215}, beta_linear_start=0.00085,
        beta_linear_end=0.012, timesteps=1000, use_ema=False))


def full_vd_cfg(with_vae=True):
    """vd_four_flow_v1-0 with the CLIP / Optimus entries replaced by string placeholders (keeps the test light)."""
    from lib.cfg_helper import CfgDict, model_cfg_bank
    bank = model_cfg_bank()
    vae = [["image", bank("autokl_v1")]] if with_vae else []
    for _, c in vae:
        c.pop("pth", None)
    return CfgDict(type="vd_v2_0", args=CfgDict(
        vae_cfg_list=vae, ctx_cfg_list=[["image", "ctx-image-placeholder"], ["text", "ctx-text-placeholder"]],
        diffuser_cfg_list=[["image", bank("openai_unet_2d_v1")], ["text", bank("openai_unet_0d_v1_c")]],
        global_layer_ptr="image", latent_scale_factor={"image": 0.18215}, beta_linear_start=0.00085,
        beta_linear_end=0.012, timesteps=1000, use_ema=False))


def synth_into(net, seed):
    """Load the deterministic synthetic weights (oracle/synth.py) into a product model; returns the fp32 state dict
    the oracle consumes."""
    from oracle import synth
    shapes = synth.shapes_of(net)
    sd = synth.synth_state_dict(shapes, seed)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected
    full = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    full.update(sd)
    return full
