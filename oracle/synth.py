"""Deterministic synthetic weights keyed by state-dict name.  TEST INFRASTRUCTURE ONLY.

No pretrained checkpoints are reachable offline, so parity is pinned on seeded random weights.  The
reference zero-initialises the 2nd conv of every ResBlock, SpatialTransformer.proj_out and the output conv
(openaimodel.py:228-230, attention.py:249-253, openaimodel.py:2735): with the stock init the UNet outputs
exactly 0, so EVERY tensor is re-drawn here.  The draw depends only on (seed, key name, shape), hence the
same weights are regenerated bit-identically on the GPU box without shipping them.
"""
import hashlib
import math

import torch

_SCHEDULE_KEYS = ("betas", "alphas_cumprod", "sqrt_", "log_one_minus", "posterior_", "lvlb_weights")


def _gen(seed, name):
    h = hashlib.sha256(("%d:%s" % (seed, name)).encode()).digest()
    g = torch.Generator(device="cpu")
    g.manual_seed(int.from_bytes(h[:7], "little"))
    return g


def synth_tensor(name, shape, seed=0):
    shape = tuple(shape)
    g = _gen(seed, name)
    leaf = name.split(".")[-1]
    is_norm = any(t in name for t in (".norm", "norm1", "norm2", "norm3", "norm_out", "layer_norm", "layernorm",
                                      "layrnorm", "LayerNorm", "in_layers.0.", "out_layers.0.", "ln_"))
    if name.endswith("position_ids"):
        return torch.arange(shape[-1]).expand(shape).clone()
    if leaf == "logit_scale":
        return torch.tensor(2.6592)
    if len(shape) == 0:
        return torch.randn((), generator=g)
    if is_norm and len(shape) == 1:
        if leaf == "weight":
            return 1.0 + 0.2 * torch.randn(shape, generator=g)
        return 0.1 * torch.randn(shape, generator=g)
    if leaf == "bias" or len(shape) == 1:
        return 0.1 * torch.randn(shape, generator=g)
    if "embedding" in name:  # token / position / class embeddings
        return 0.5 * torch.randn(shape, generator=g)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return torch.randn(shape, generator=g) / math.sqrt(fan_in)


def synth_state_dict(shapes, seed=0, skip_schedule=True):
    """shapes: {name: shape}. Returns {name: fp32 tensor}; DDPM schedule buffers are left to the model."""
    out = {}
    for name in sorted(shapes):
        if skip_schedule and name.split(".")[0].startswith(_SCHEDULE_KEYS) and "." not in name:
            continue
        out[name] = synth_tensor(name, shapes[name], seed)
    return out


def shapes_of(module_or_sd):
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, "state_dict") else module_or_sd
    return {k: tuple(v.shape) for k, v in sd.items()}


def load_synth_(module, seed=0, prefix=""):
    """Overwrite every parameter/buffer of `module` (except the top-level schedule buffers)."""
    shapes = shapes_of(module)
    sd = synth_state_dict({prefix + k: v for k, v in shapes.items()}, seed)
    own = module.state_dict()
    with torch.no_grad():
        for k, v in own.items():
            full = prefix + k
            if full in sd:
                v.copy_(sd[full].to(v.dtype))
    return module
