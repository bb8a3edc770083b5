"""CPU fp32 oracle of the Versatile-Diffusion sampling path.  TEST INFRASTRUCTURE ONLY.

A functional restatement (state-dict in, tensors out; NCHW, torch fp32 on the host) of what the reference
computes on the path named by BASELINE.json: multi-flow UNet forward, CFG + DDIM loop, AutoencoderKL
encode/decode, CLIP text/image context encoders.  Every function cites the reference lines it follows
(paths relative to /root/reference).  It is pinned against the reference itself: oracle/gen_golden.py runs
the reference's own modules in this container and stores inputs/outputs under tests/golden/, and
tests/test_oracle_golden.py checks this file against those fixtures (and against the live reference when
/root/reference is present).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ------------------------------------------------------------------------------------------------
# schedules  (lib/model_zoo/diffusion_utils.py:8-59, lib/model_zoo/vd.py:127-185, ddim.py:23-56)
# ------------------------------------------------------------------------------------------------


def make_beta_schedule_linear(n_timestep=1000, linear_start=0.00085, linear_end=0.012):
    """diffusion_utils.py:8-12: linspace(sqrt(start), sqrt(end), n, f64) ** 2"""
    return (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2).numpy()


def register_schedule(timesteps=1000, linear_start=0.00085, linear_end=0.012):
    """vd.py:127-173 -- f64 numpy math, buffers stored as fp32 tensors."""
    betas = make_beta_schedule_linear(timesteps, linear_start, linear_end)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    f32 = lambda a: torch.tensor(a, dtype=torch.float32)
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    return {
        "betas": f32(betas), "alphas_cumprod": f32(ac), "alphas_cumprod_prev": f32(ac_prev),
        "sqrt_alphas_cumprod": f32(np.sqrt(ac)), "sqrt_one_minus_alphas_cumprod": f32(np.sqrt(1.0 - ac)),
        "log_one_minus_alphas_cumprod": f32(np.log(1.0 - ac)), "sqrt_recip_alphas_cumprod": f32(np.sqrt(1.0 / ac)),
        "sqrt_recipm1_alphas_cumprod": f32(np.sqrt(1.0 / ac - 1)), "posterior_variance": f32(post_var),
        "posterior_log_variance_clipped": f32(np.log(np.maximum(post_var, 1e-20))),
        "posterior_mean_coef1": f32(betas * np.sqrt(ac_prev) / (1.0 - ac)),
        "posterior_mean_coef2": f32((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)),
    }


def make_ddim_timesteps(num_ddim_timesteps, num_ddpm_timesteps=1000):
    """diffusion_utils.py:32-46 ('uniform'): arange(0, T, T // S) + 1"""
    c = num_ddpm_timesteps // num_ddim_timesteps
    return np.asarray(list(range(0, num_ddpm_timesteps, c))) + 1


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta):
    """diffusion_utils.py:48-59. alphacums: fp32 tensor [T]. Returns (sigmas, alphas, alphas_prev)."""
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return sigmas, alphas, alphas_prev


def ddim_schedule(alphas_cumprod, steps, eta=0.0):
    """DDIMSampler.make_schedule (ddim.py:23-56) reduced to what the sampling loop reads."""
    ts = make_ddim_timesteps(steps, alphas_cumprod.shape[0])
    sig, a, a_prev = make_ddim_sampling_parameters(alphas_cumprod.cpu(), ts, eta)
    a = torch.as_tensor(a, dtype=torch.float32)
    sig = torch.as_tensor(np.asarray(sig), dtype=torch.float32)
    return {"timesteps": ts, "alphas": a, "alphas_prev": np.asarray(a_prev, dtype=np.float64),
            "sigmas": sig, "sqrt_one_minus_alphas": torch.sqrt(1.0 - a)}


def timestep_embedding(timesteps, dim, max_period=10000):
    """diffusion_utils.py:131-151"""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


# ------------------------------------------------------------------------------------------------
# UNet building blocks
# ------------------------------------------------------------------------------------------------


def _gn(sd, p, x, eps):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _conv(sd, p, x, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def resblock(sd, p, x, emb):
    """ResBlock._forward, use_scale_shift_norm=False, no up/down (openaimodel.py:254-274); GN eps 1e-5."""
    h = _conv(sd, p + ".in_layers.2", F.silu(_gn(sd, p + ".in_layers.0", x, 1e-5)), padding=1)
    emb_out = _lin(sd, p + ".emb_layers.1", F.silu(emb))
    h = h + emb_out[:, :, None, None]
    h = _conv(sd, p + ".out_layers.3", F.silu(_gn(sd, p + ".out_layers.0", h, 1e-5)), padding=1)
    if (p + ".skip_connection.weight") in sd:
        x = _conv(sd, p + ".skip_connection", x)
    return x + h


def cross_attention(sd, p, x, context, heads):
    """CrossAttention.forward (attention.py:170-193); scale = dim_head ** -0.5 applied after QK^T."""
    context = x if context is None else context
    q = F.linear(x, sd[p + ".to_q.weight"])
    k = F.linear(context, sd[p + ".to_k.weight"])
    v = F.linear(context, sd[p + ".to_v.weight"])
    b, n, c = q.shape
    d = c // heads
    sp = lambda t: t.view(b, t.shape[1], heads, d).transpose(1, 2)
    sim = sp(q) @ sp(k).transpose(-1, -2) * (d ** -0.5)
    out = (sim.softmax(dim=-1) @ sp(v)).transpose(1, 2).reshape(b, n, c)
    return _lin(sd, p + ".to_out.0", out)


def basic_transformer_block(sd, p, x, context, heads):
    """BasicTransformerBlock._forward (attention.py:214-218), GEGLU FF (attention.py:37-64), LN eps 1e-5."""
    c = x.shape[-1]
    ln = lambda name, t: F.layer_norm(t, (c,), sd[p + "." + name + ".weight"], sd[p + "." + name + ".bias"], 1e-5)
    x = cross_attention(sd, p + ".attn1", ln("norm1", x), None, heads) + x
    x = cross_attention(sd, p + ".attn2", ln("norm2", x), context, heads) + x
    h, gate = _lin(sd, p + ".ff.net.0.proj", ln("norm3", x)).chunk(2, dim=-1)
    x = _lin(sd, p + ".ff.net.2", h * F.gelu(gate)) + x
    return x


def spatial_transformer(sd, p, x, context, heads):
    """SpatialTransformer.forward (attention.py:255-266); GN eps 1e-6, 1x1 proj in/out, depth 1."""
    b, c, hh, ww = x.shape
    h = _conv(sd, p + ".proj_in", _gn(sd, p + ".norm", x, 1e-6))
    h = h.flatten(2).transpose(1, 2)
    h = basic_transformer_block(sd, p + ".transformer_blocks.0", h, context, heads)
    h = h.transpose(1, 2).reshape(b, c, hh, ww)
    return _conv(sd, p + ".proj_out", h) + x


def unet_plan(in_channels=4, model_channels=320, out_channels=4, num_res_blocks=(2, 2, 2, 2),
              attention_resolutions=(4, 2, 1), channel_mult=(1, 2, 4, 4), num_heads=8, num_head_channels=None,
              **_ignored):
    """Structure of UNetModel2D_Next.__init__ (openaimodel.py:2575-2753) as data: the list of data blocks, the
    list of context blocks (heads per block) and the i/m/o layer orders that VD_v2_0.apply_model walks."""
    if isinstance(num_res_blocks, int):
        num_res_blocks = [num_res_blocks] * len(channel_mult)

    def heads_of(ch):
        return num_heads if num_head_channels is None else ch // num_head_channels

    data, ctx, order = [], [], []
    data.append(("conv_in", in_channels, model_channels)); order += ["d", "save_hidden_feature"]
    chans = [model_channels]
    ch, ds = model_channels, 1
    for level, mult in enumerate(channel_mult):
        for _ in range(num_res_blocks[level]):
            data.append(("res", ch, mult * model_channels)); order.append("d")
            ch = mult * model_channels
            if ds in attention_resolutions:
                ctx.append((ch, heads_of(ch))); order.append("c")
            chans.append(ch); order.append("save_hidden_feature")
        if level != len(channel_mult) - 1:
            data.append(("down", ch, ch)); order += ["d", "save_hidden_feature"]
            chans.append(ch)
            ds *= 2
    i_order, order = order, []
    data.append(("res", ch, ch)); order.append("d")
    ctx.append((ch, heads_of(ch))); order.append("c")
    data.append(("res", ch, ch)); order.append("d")
    m_order, order = order, []
    for level, mult in list(enumerate(channel_mult))[::-1]:
        for _ in range(num_res_blocks[level] + 1):
            order.append("load_hidden_feature")
            ich = chans.pop()
            data.append(("res", ch + ich, model_channels * mult)); order.append("d")
            ch = model_channels * mult
            if ds in attention_resolutions:
                ctx.append((ch, heads_of(ch))); order.append("c")
        if level != 0:
            data.append(("up", ch, ch)); order.append("d")
            ds //= 2
    data.append(("out", ch, out_channels)); order.append("d")
    return {"data": data, "ctx": ctx, "i_order": i_order, "m_order": m_order, "o_order": order,
            "model_channels": model_channels}


def fcblock(sd, p, x, emb):
    """FCBlock_MultiDim.forward / FCBlock._forward (openaimodel.py:2084-2141, 2295-2332): the [C, sdim, 1] tail of x is
    flattened to C*sdim channels of a 1x1 map; GN32 (eps 1e-5) -> SiLU -> 1x1 conv, + emb, GN32 -> SiLU -> 1x1 conv,
    + skip(x)."""
    shape = x.shape
    xf = x.reshape(shape[0], -1, 1, 1)
    h = _conv(sd, p + ".in_layers.2", F.silu(_gn(sd, p + ".in_layers.0", xf, 1e-5)))
    emb_out = _lin(sd, p + ".emb_layers.1", F.silu(emb))
    h = h + emb_out[:, :, None, None]
    h = _conv(sd, p + ".out_layers.3", F.silu(_gn(sd, p + ".out_layers.0", h, 1e-5)))
    if (p + ".skip_connection.weight") in sd:
        xf = _conv(sd, p + ".skip_connection", xf)
    return xf + h          # [B, Cout*sdim, 1, 1]; the caller views it as [B, Cout, sdim, 1]


def linear_multidim(sd, p, x, out_shape):
    """Linear_MultiDim.forward (openaimodel.py:2275-2293): flatten the trailing feature dims, nn.Linear, view back."""
    return _lin(sd, p, x.reshape(x.shape[0], -1)).view(x.shape[0], *out_shape)


def unet0d_plan(input_channels=768, model_channels=320, output_channels=768, num_noattn_blocks=(2, 2, 2, 2),
                channel_mult=(1, 2, 4, 8), second_dim=(4, 4, 4, 4), with_attn=(True, True, True, False), num_heads=8,
                num_head_channels=None, **_ignored):
    """Structure of UNetModel0D_Next.__init__ (openaimodel.py:2815-2966) as data; same shape of result as unet_plan.
    Data entries carry the [C, sdim, 1] output shape of the block."""
    def heads_of(ch):
        return num_heads if num_head_channels is None else ch // num_head_channels

    data, ctx, order = [], [], []
    cur = [model_channels, second_dim[0], 1]
    data.append(("lin_md", cur)); order += ["d", "save_hidden_feature"]
    chans = [cur]
    for level, (mult, sdim) in enumerate(zip(channel_mult, second_dim)):
        for _ in range(num_noattn_blocks[level]):
            cur = [mult * model_channels, sdim, 1]
            data.append(("fc", cur)); order.append("d")
            if with_attn[level]:
                ctx.append((cur[0], heads_of(cur[0]))); order.append("c")
            chans.append(cur); order.append("save_hidden_feature")
        if level != len(channel_mult) - 1:
            data.append(("lin_md", cur)); order += ["d", "save_hidden_feature"]
            chans.append(cur)
    i_order, order = order, []
    data.append(("fc", cur)); order.append("d")
    ctx.append((cur[0], heads_of(cur[0]))); order.append("c")
    data.append(("fc", cur)); order.append("d")
    m_order, order = order, []
    for level, (mult, sdim) in list(enumerate(zip(channel_mult, second_dim)))[::-1]:
        for _ in range(num_noattn_blocks[level] + 1):
            order.append("load_hidden_feature")
            chans.pop()
            cur = [mult * model_channels, sdim, 1]
            data.append(("fc", cur)); order.append("d")
            if with_attn[level]:
                ctx.append((cur[0], heads_of(cur[0]))); order.append("c")
        if level != 0:
            data.append(("lin_md", cur)); order.append("d")
    data.append(("out0d", [output_channels])); order.append("d")
    return {"data": data, "ctx": ctx, "i_order": i_order, "m_order": m_order, "o_order": order,
            "model_channels": model_channels}


def _data_block(sd, p, kind, h, emb):
    """TimestepEmbedSequential dispatch (openaimodel.py:78-86) for the data blocks of the 2D UNet."""
    if kind == "conv_in":
        return _conv(sd, p + ".0", h, padding=1)                      # openaimodel.py:2664
    if kind == "res":
        return resblock(sd, p + ".0", h, emb)
    if kind == "down":
        return _conv(sd, p + ".0.op", h, stride=2, padding=1)         # Downsample, openaimodel.py:133-159
    if kind == "up":
        h = F.interpolate(h, scale_factor=2, mode="nearest")          # Upsample, openaimodel.py:89-117
        return _conv(sd, p + ".0.conv", h, padding=1)
    if kind == "out":      # openaimodel.py:2732-2737: an nn.Sequential wrapped in the TimestepEmbedSequential -> keys .0.0 / .0.2
        return _conv(sd, p + ".0.2", F.silu(_gn(sd, p + ".0.0", h, 1e-5)), padding=1)
    raise ValueError(kind)


def time_embed(sd, p, t_emb):
    """openaimodel.py:2627-2633"""
    return _lin(sd, p + ".2", F.silu(_lin(sd, p + ".0", t_emb)))


def apply_model_multicontext(sd, plan, x, timesteps, contexts, x_type="image", global_ptr=None):
    """VD_v2_0.apply_model / apply_model_multicontext (vd.py:330-455) with mixing_type='attention'.

    contexts: list of (c_type, c [B, L, 768], ratio).  One entry with ratio 1 == apply_model.
    Data blocks come from diffuser[x_type], context blocks from diffuser[c_type], time_embed from the global
    layer pointer."""
    g = x_type if global_ptr is None else global_ptr
    emb = time_embed(sd, "diffuser.%s.time_embed" % g, timestep_embedding(timesteps, plan["model_channels"]))
    ratios = np.array([float(r) for _, _, r in contexts])
    ratios = ratios / ratios.sum()
    di, ci = iter(range(len(plan["data"]))), iter(range(len(plan["ctx"])))
    hs = []

    def run_d(h):
        i = next(di)
        kind = plan["data"][i][0]
        p = "diffuser.%s.data_blocks.%d" % (x_type, i)
        if kind == "lin_md":      # plain layer inside the TimestepEmbedSequential (openaimodel.py:78-86)
            return linear_multidim(sd, p + ".0", h, plan["data"][i][1])
        if kind == "fc":
            return fcblock(sd, p + ".0", h, emb).view(h.shape[0], *plan["data"][i][1])
        if kind == "out0d":
            return linear_multidim(sd, p + ".0.2", F.silu(_gn(sd, p + ".0.0", h, 1e-5)), plan["data"][i][1])
        return _data_block(sd, p, kind, h, emb)

    def run_c(h):
        j = next(ci)
        heads = plan["ctx"][j][1]
        out = None
        for (c_type, c, _), r in zip(contexts, ratios):
            hi = spatial_transformer(sd, "diffuser.%s.context_blocks.%d.0" % (c_type, j), h, c, heads)
            if len(contexts) > 1:
                hi = hi * float(r)
            out = hi if out is None else out + hi
        return out

    h = x
    for lt in plan["i_order"] + plan["m_order"]:
        if lt == "d":
            h = run_d(h)
        elif lt == "c":
            h = run_c(h)
        elif lt == "save_hidden_feature":
            hs.append(h)
    for lt in plan["o_order"]:
        if lt == "load_hidden_feature":
            h = torch.cat([h, hs.pop()], dim=1)
        elif lt == "d":
            h = run_d(h)
        elif lt == "c":
            h = run_c(h)
    return h


def apply_model(sd, plan, x, timesteps, c, x_type="image", c_type="text", global_ptr=None):
    return apply_model_multicontext(sd, plan, x, timesteps, [(c_type, c, 1.0)], x_type, global_ptr)


# ------------------------------------------------------------------------------------------------
# DDIM sampling  (lib/model_zoo/ddim.py:81-171, 196-298)
# ------------------------------------------------------------------------------------------------


def p_sample_ddim(sd, plan, sched, x, contexts, index, step, guidance_scale, x_type="image", global_ptr=None,
                  noise=None):
    """contexts: list of dicts {type, conditioning, unconditional_conditioning, ratio}."""
    b = x.shape[0]
    t = torch.full((b,), int(step), dtype=torch.long)
    if guidance_scale == 1.0:
        e_t = apply_model_multicontext(sd, plan, x, t, [(c["type"], c["conditioning"], c.get("ratio", 1.0))
                                                        for c in contexts], x_type, global_ptr)
    else:
        x_in, t_in = torch.cat([x] * 2), torch.cat([t] * 2)
        cs = [(c["type"], torch.cat([c["unconditional_conditioning"], c["conditioning"]]), c.get("ratio", 1.0))
              for c in contexts]
        e_u, e_c = apply_model_multicontext(sd, plan, x_in, t_in, cs, x_type, global_ptr).chunk(2)
        e_t = e_u + guidance_scale * (e_c - e_u)
    a_t = float(sched["alphas"][index])
    a_prev = float(sched["alphas_prev"][index])
    sigma = float(sched["sigmas"][index])
    s1m = float(sched["sqrt_one_minus_alphas"][index])
    pred_x0 = (x - s1m * e_t) / math.sqrt(a_t)
    dir_xt = math.sqrt(1.0 - a_prev - sigma ** 2) * e_t
    x_prev = math.sqrt(a_prev) * pred_x0 + dir_xt
    if noise is not None:
        x_prev = x_prev + sigma * noise
    return x_prev, pred_x0


def ddim_sample(sd, plan, alphas_cumprod, x_T, contexts, steps, guidance_scale, eta=0.0, x_type="image",
                global_ptr=None, forward_steps=None):
    """DDIMSampler.ddim_sampling[_multicontext] with an explicit initial latent (RNG is injected, never re-drawn).
    forward_steps: run only the first `forward_steps` DDIM timesteps (x_T then is q_sample(x0), ddim.py:97-103)."""
    sched = ddim_schedule(alphas_cumprod, steps, eta)
    ts = sched["timesteps"] if forward_steps is None else sched["timesteps"][:forward_steps]
    x = x_T
    pred_x0 = None
    for i, step in enumerate(np.flip(ts)):
        index = ts.shape[0] - i - 1
        x, pred_x0 = p_sample_ddim(sd, plan, sched, x, contexts, index, step, guidance_scale, x_type, global_ptr)
    return x, pred_x0


def q_sample(sched_buffers, x0, t, noise):
    """vd.py:221-224"""
    sa = sched_buffers["sqrt_alphas_cumprod"][t].view(-1, 1, 1, 1)
    sb = sched_buffers["sqrt_one_minus_alphas_cumprod"][t].view(-1, 1, 1, 1)
    return sa * x0 + sb * noise


# ------------------------------------------------------------------------------------------------
# AutoencoderKL  (lib/model_zoo/autokl.py:30-49, autokl_modules.py:38-202,368-568)
# ------------------------------------------------------------------------------------------------


def _vae_resnet(sd, p, x):
    """ResnetBlock.forward with temb=None (autokl_modules.py:118-141), GN eps 1e-6, swish."""
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x, 1e-6)), padding=1)
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h, 1e-6)), padding=1)
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x)
    return x + h


def _vae_attn(sd, p, x):
    """AttnBlock.forward (autokl_modules.py:176-202): single head, scale C^-0.5, softmax over keys."""
    h = _gn(sd, p + ".norm", x, 1e-6)
    q, k, v = _conv(sd, p + ".q", h), _conv(sd, p + ".k", h), _conv(sd, p + ".v", h)
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, hh * ww)
    h = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, p + ".proj_out", h)


def vae_decode(sd, p, z, ch_mult=(1, 2, 4, 4), num_res_blocks=2):
    """AutoencoderKL.decode (autokl.py:44-49) + Decoder.forward (autokl_modules.py:535-568)."""
    nres = len(ch_mult)
    h = _conv(sd, p + ".post_quant_conv", z)
    d = p + ".decoder"
    h = _conv(sd, d + ".conv_in", h, padding=1)
    h = _vae_resnet(sd, d + ".mid.block_1", h)
    h = _vae_attn(sd, d + ".mid.attn_1", h)
    h = _vae_resnet(sd, d + ".mid.block_2", h)
    for lvl in reversed(range(nres)):
        for blk in range(num_res_blocks + 1):
            h = _vae_resnet(sd, d + ".up.%d.block.%d" % (lvl, blk), h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, d + ".up.%d.upsample.conv" % lvl, h, padding=1)
    h = _conv(sd, d + ".conv_out", F.silu(_gn(sd, d + ".norm_out", h, 1e-6)), padding=1)
    return torch.clamp((h + 1) / 2, 0, 1)


def vae_encode_moments(sd, p, x, ch_mult=(1, 2, 4, 4), num_res_blocks=2):
    """AutoencoderKL.encode up to the moments (autokl.py:30-37) + Encoder.forward (autokl_modules.py:434-459)."""
    nres = len(ch_mult)
    e = p + ".encoder"
    h = _conv(sd, e + ".conv_in", x * 2 - 1, padding=1)
    for lvl in range(nres):
        for blk in range(num_res_blocks):
            h = _vae_resnet(sd, e + ".down.%d.block.%d" % (lvl, blk), h)
        if lvl != nres - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)   # asymmetric pad, autokl_modules.py:72-76
            h = _conv(sd, e + ".down.%d.downsample.conv" % lvl, h, stride=2)
    h = _vae_resnet(sd, e + ".mid.block_1", h)
    h = _vae_attn(sd, e + ".mid.attn_1", h)
    h = _vae_resnet(sd, e + ".mid.block_2", h)
    h = _conv(sd, e + ".conv_out", F.silu(_gn(sd, e + ".norm_out", h, 1e-6)), padding=1)
    return _conv(sd, p + ".quant_conv", h)


def diag_gaussian_sample(moments, noise):
    """DiagonalGaussianDistribution (distributions.py:24-37) with the noise injected."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return mean + torch.exp(0.5 * logvar) * noise


# ------------------------------------------------------------------------------------------------
# CLIP context encoders  (lib/model_zoo/clip.py:53-62, 88-143)
# The tower arithmetic lives in the third-party `transformers` package (pinned 4.24.0 by the reference's
# requirements.txt:12; 5.x installed here): CLIPModel = pre-LN transformer, quick-GELU MLP, learned absolute
# positions, causal mask in the text tower, pooled token = argmax(input_ids) (EOS has the largest id).
# Restated here from the published algorithm and pinned against the installed implementation by
# oracle/gen_golden.py through the reference's own call sequence.
# ------------------------------------------------------------------------------------------------


def _clip_layer(sd, p, x, heads, causal):
    c = x.shape[-1]
    ln = lambda name, t: F.layer_norm(t, (c,), sd[p + "." + name + ".weight"], sd[p + "." + name + ".bias"], 1e-5)
    h = ln("layer_norm1", x)
    b, n, _ = h.shape
    d = c // heads
    sp = lambda t: t.view(b, n, heads, d).transpose(1, 2)
    q = sp(_lin(sd, p + ".self_attn.q_proj", h)) * (d ** -0.5)
    k = sp(_lin(sd, p + ".self_attn.k_proj", h))
    v = sp(_lin(sd, p + ".self_attn.v_proj", h))
    s = q @ k.transpose(-1, -2)
    if causal:
        s = s + torch.full((n, n), float("-inf")).triu(1)
    a = (s.softmax(-1) @ v).transpose(1, 2).reshape(b, n, c)
    x = x + _lin(sd, p + ".self_attn.out_proj", a)
    h = _lin(sd, p + ".mlp.fc1", ln("layer_norm2", x))
    h = h * torch.sigmoid(1.702 * h)
    return x + _lin(sd, p + ".mlp.fc2", h)


def clip_text_context(sd, p, input_ids, heads, layers):
    """CLIPTextContextEncoder.encode after tokenisation (clip.py:57-61): project ALL hidden states and divide
    by the norm of the projected pooled (EOS) state."""
    t = p + ".text_model"
    L = input_ids.shape[1]
    x = sd[t + ".embeddings.token_embedding.weight"][input_ids] + sd[t + ".embeddings.position_embedding.weight"][:L]
    for i in range(layers):
        x = _clip_layer(sd, t + ".encoder.layers.%d" % i, x, heads, causal=True)
    c = x.shape[-1]
    x = F.layer_norm(x, (c,), sd[t + ".final_layer_norm.weight"], sd[t + ".final_layer_norm.bias"], 1e-5)
    pooled = x[torch.arange(x.shape[0]), input_ids.argmax(dim=-1)]
    z = F.linear(x, sd[p + ".text_projection.weight"])
    z_pooled = F.linear(pooled, sd[p + ".text_projection.weight"])
    return z / torch.norm(z_pooled.unsqueeze(1), dim=-1, keepdim=True)


def clip_vtoken_mask(masks, patch=14, size=224):
    """clip.py:104-122: bilinear resize to 224^2, per-patch mean, global mean in front. masks [B,1,H,W] in [0,1]."""
    masks = torch.clamp(masks, 0, 1).float()
    masks = F.interpolate(masks, [size, size], mode="bilinear")
    gscale = masks.mean(dim=[1, 2, 3], keepdim=True).flatten(2)
    vt = F.conv2d(masks, torch.ones(1, 1, patch, patch), stride=patch).flatten(2).transpose(1, 2) / (patch * patch)
    return torch.cat([gscale, vt], dim=1), bool(masks.sum() == masks.numel())


def clip_image_context(sd, p, pixel_values, heads, layers, vtoken_mask=None):
    """CLIPImageContextEncoder._encode / _encode_wmask after the CLIPProcessor (clip.py:95-100, 124-142):
    post_layernorm on ALL tokens, visual_projection, divide by the norm of the projected CLS token; with a mask the
    embeddings (pre-encoder) and the outputs are multiplied by [global mean | patch means]."""
    v = p + ".vision_model"
    w = sd[v + ".embeddings.patch_embedding.weight"]
    patch = w.shape[-1]
    pe = F.conv2d(pixel_values, w, stride=patch).flatten(2).transpose(1, 2)
    b = pe.shape[0]
    cls = sd[v + ".embeddings.class_embedding"].expand(b, 1, -1)
    x = torch.cat([cls, pe], dim=1) + sd[v + ".embeddings.position_embedding.weight"][None]
    if vtoken_mask is not None:
        x = x * vtoken_mask
    c = x.shape[-1]
    x = F.layer_norm(x, (c,), sd[v + ".pre_layrnorm.weight"], sd[v + ".pre_layrnorm.bias"], 1e-5)
    for i in range(layers):
        x = _clip_layer(sd, v + ".encoder.layers.%d" % i, x, heads, causal=False)
    x = F.layer_norm(x, (c,), sd[v + ".post_layernorm.weight"], sd[v + ".post_layernorm.bias"], 1e-5)
    z = F.linear(x, sd[p + ".visual_projection.weight"])
    z = z / torch.norm(z[:, 0:1], dim=-1, keepdim=True)
    if vtoken_mask is not None:
        z = z * vtoken_mask
    return z
