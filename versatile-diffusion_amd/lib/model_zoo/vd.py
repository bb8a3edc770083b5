"""VD_v2_0: the multi-flow container (VAEs, context encoders, diffusers + DDPM schedule) behind the reference's
model contract (lib/model_zoo/vd.py:41-455 there): `ctx_encode`, `vae_encode`, `vae_decode`, `apply_model`,
`apply_model_multicontext`, `q_sample`, the 12 schedule buffers, `.to()` semantics and the state-dict layout.

Execution differs from the reference: `apply_model*` converts nothing to NCHW in between -- the latent enters the
first conv straight from NCHW, every block runs channels-last fp16 on HIP kernels, skip connections are read in
place (no torch.cat), context mixing is folded into the proj_out epilogues, and the step-invariant context K/V
projections can be cached across DDIM steps (`c_info['kv_cache']`, set up by DDIMSampler).
"""
from functools import partial

import os

import numpy as np
import numpy.random as npr
import torch
import torch.nn as nn

from vd_hip import ops

from ..log_service import print_log
from .common.get_model import get_model, register
from .diffusion_utils import extract_into_tensor, make_beta_schedule, timestep_embedding

symbol = "vd"


class String_Reg_Buffer(nn.Module):
    """A config entry given as a plain string is kept as a byte buffer (reference vd.py:28-39)."""

    def __init__(self, output_string):
        super().__init__()
        self.register_buffer("output_string", torch.tensor(list(bytes(output_string, "utf8")), dtype=torch.uint8))

    @torch.no_grad()
    def forward(self, *args, **kwargs):
        return bytes(self.output_string.tolist()).decode()


CTX_FORK = os.environ.get("VD_CTX_FORK", "1") != "0"   # development switch: 0 = the context types of a block one after the other
# Round 6: the low-resolution levels of the UNet (rows per sample <= VD_BATCH_FORK_HW, default 256 = the 16x16 and 8x8 levels and the
# middle block) as TWO forked branches of half the batch each.  Measured (profiles/HISTORY.md R6.8): image variation at CFG batch 16
# +1.6 .. +2.5 % images/s (each half is the CFG-batch-8 problem the planner's rules were measured on, and the halves' latency-bound
# launches fill each other's gaps); text-to-image at CFG batch 8 -4 % (halves of 4 samples: split-K convolutions that each fill
# the chip cannot run side by side).  So: "auto" = batches of >= VD_BATCH_FORK_MIN (16) samples with ONE context type (the context
# types of a multi-context stage are forked already); 0 = never, 1 = every even batch with one context type.
BATCH_FORK = os.environ.get("VD_BATCH_FORK", "auto")
BATCH_FORK_HW = int(os.environ.get("VD_BATCH_FORK_HW", "256"))
BATCH_FORK_MIN = int(os.environ.get("VD_BATCH_FORK_MIN", "16"))
_SIDE = {}


def _side_streams(device, n, kind="ctx"):
    """Side streams of the forked context-type branches, one set per device for the whole process: the eager warm-up step and
    the captured step use the same streams, so the per-stream workspaces of vd_hip.ops (keyed by stream) are allocated once,
    outside any capture."""
    lst = _SIDE.setdefault((kind, device.index), [])
    while len(lst) < n:
        lst.append(torch.cuda.Stream(device=device))
    return lst[:n]


def run_unet(data_net, ctx_specs, x, emb_silu, mixing_type="attention", repeat=1, emb_rows=None):
    """Walk i/m/o orders of `data_net` (reference vd.py:352-378 / 429-453).

    ctx_specs: list of (context_blocks, context [B, L, Dc], ratio, kv_cache or None), one per context type.
    x: NCHW latent; returns NCHW eps in fp16.
    repeat > 1: the batch the reference would feed is `x` replicated `repeat` times ([x; x] of classifier-free guidance,
    ddim.py:144-149) with identical timesteps per replica, while contexts / emb carry the full repeat * B rows.  The data
    blocks in front of the first context block see identical inputs in every replica, so they run once on B rows and the
    result (and the skip tensors saved so far) is replicated where the contexts make the replicas diverge.
    emb_rows: {data block index: [Cout] fp16} complete first-conv bias vectors of the ResBlocks for the ONE timestep the whole
    batch shares (VD_v2_0.precompute_step_emb); emb_silu may then be None."""
    d_iter = iter(enumerate(data_net.data_blocks))
    if emb_rows is not None:
        emb_outs = {}
    else:
        emb_rows = {}
        emb_outs = data_net.precompute_emb(emb_silu) if hasattr(data_net, "precompute_emb") else {}
    c_iters = [iter(spec[0]) for spec in ctx_specs]
    ratios = np.array([float(spec[2]) for spec in ctx_specs], dtype=np.float64)
    ratios = ratios / ratios.sum()

    # the walk, resolved to modules once: ("d", index, block) / ("c", modules, specs, ratios) / ("save",) / ("load",)
    steps = []
    for ltype in list(data_net.i_order) + list(data_net.m_order) + list(data_net.o_order):
        if ltype == "d":
            di, blk = next(d_iter)
            steps.append(("d", di, blk))
        elif ltype == "c":
            modules = [next(it) for it in c_iters]
            if mixing_type == "layer":
                pick = int(npr.choice(len(modules), p=ratios))
                steps.append(("c", [modules[pick]], [ctx_specs[pick]], [1.0]))
            else:
                steps.append(("c", modules, list(ctx_specs), ratios))
        elif ltype == "save_hidden_feature":
            steps.append(("save",))
        elif ltype == "load_hidden_feature":
            steps.append(("load",))

    def context_kv(module, spec):
        _, c, _, cache = spec
        kv = None
        if cache is not None:
            kv = cache.get(id(module))
            stale = cache.get("_stale")
            if kv is None:
                kv = module[0].project_context(c)
                cache[id(module)] = kv
                cache.setdefault("_modules", {})[id(module)] = module   # lets the sampler refresh K/V without a forward
            elif stale and id(module) in stale:   # buffer a captured graph reads: new context, same storage
                kv.copy_(module[0].project_context(c))
                stale.discard(id(module))
        return kv

    def run_context(h, modules, specs, rs, sl=None):
        """sl = (b0, b1): h holds samples b0 .. b1 - 1 of the batch (a half-batch branch): contexts and K/V are sliced alike."""
        single = len(modules) == 1
        kvs = [context_kv(m, sp) for m, sp in zip(modules, specs)]
        if sl is not None:
            kvs = [None if kv is None else kv[sl[0]:sl[1]] for kv in kvs]
            specs = [(sp[0], None if sp[1] is None else sp[1][sl[0]:sl[1]]) + tuple(sp[2:]) for sp in specs]
        # h_out = sum_i r_i * ST_i(h) = sum_i r_i * proj_i + h   (sum r_i = 1): chained through the epilogue of each type's LAST
        # launch (alpha = r_i, res = the previous type's output).  Everything in front of that launch depends on h only, so the
        # types run as forked branches (side streams; inside the sampler's HIP graph: parallel branches): at the per-GPU batch of
        # the multi-context workloads one type's launches fill half the chip.  Only the last launches are ordered, by events.
        if single or not CTX_FORK or not h.is_cuda:
            out = None
            for module, spec, kv, r in zip(modules, specs, kvs, rs):
                out = module(h, None, spec[1], kv=kv, alpha=1.0 if single else float(r), res=out)
            return out
        main = torch.cuda.current_stream()
        sides = _side_streams(h.device, len(modules) - 1)
        fork = torch.cuda.Event()
        fork.record(main)
        outs, done = [], []
        for i, (module, spec, kv, r) in enumerate(zip(modules, specs, kvs, rs)):
            st = main if i == 0 else sides[i - 1]
            prev = outs[-1] if outs else None
            prev_done = done[-1] if done else None
            with torch.cuda.stream(st):
                if i > 0:
                    st.wait_event(fork)
                out = module(h, None, spec[1], kv=kv, alpha=float(r), res=prev,
                             sync=(None if prev_done is None else (lambda e=prev_done, s_=st: s_.wait_event(e))))
                ev = torch.cuda.Event()
                ev.record(st)
            if i > 0:
                out.record_stream(main)
                st_ = getattr(out, "_vd_stats", None)   # ChanStats of the block output, written by the branch's last launch
                if st_ is not None and torch.is_tensor(getattr(st_, "buf", None)):
                    st_.buf.record_stream(main)
            outs.append(out)
            done.append(ev)
        main.wait_event(done[-1])
        for ev in done[1:-1]:
            main.wait_event(ev)
        return outs[-1]

    h = x
    nb = x.shape[0]
    shared = repeat > 1
    # LayerNorm row statistics accumulated by producer epilogues live in one arena per forward, zeroed by ONE fill
    # (the arena object is per CALL -- two threads running forwards of one net must not hand out overlapping slices -- the module
    # remembers only how many rows the previous forward took)
    arena = ops.RowSumArena(data_net.__dict__.get("_vd_rowsum_need", 0))
    arena.begin(x.device)
    try:
        return _run_unet_body(steps, emb_outs, emb_rows, emb_silu, run_context, lambda ms, sps: [context_kv(m, sp) for m, sp in zip(ms, sps)],
                              h, nb, shared, repeat, data_net.__dict__.setdefault("_vd_fork_warm", set()))
    finally:
        arena.end()
        data_net.__dict__["_vd_rowsum_need"] = arena.need


def _fork_region(steps, hw0, thr):
    """[a, b) of `steps`: from the Downsample that brings the rows per sample to <= thr up to (not including) the Upsample that
    leaves that range again, or None.  The region must keep its skip tensors to itself (as many loads as saves, never below)."""
    from .openaimodel import Downsample, Upsample
    hw, a, b = hw0, None, None
    for i, st in enumerate(steps):
        if st[0] != "d":
            continue
        layers = list(st[2]) if isinstance(st[2], (nn.Sequential, list, tuple)) else [st[2]]
        if any(isinstance(l, Downsample) for l in layers):
            hw //= 4
            if a is None and hw <= thr:
                a = i
        elif any(isinstance(l, Upsample) for l in layers):
            if a is not None and b is None and hw <= thr and hw * 4 > thr:
                b = i
            hw *= 4
    if a is None or b is None or b <= a:
        return None
    depth = 0
    for st in steps[a:b]:
        depth += 1 if st[0] == "save" else (-1 if st[0] == "load" else 0)
        if depth < 0:
            return None
    return (a, b) if depth == 0 and any(st[0] == "c" for st in steps[a:b]) else None


def _run_unet_body(steps, emb_outs, emb_rows, emb_silu, run_context, prepare_context, h, nb, shared, repeat, warm):
    state = {"shared": shared}

    def walk(lo, hi, h, hs, sl=None):
        """steps[lo:hi] on h; sl = (b0, b1): h is that slice of the batch (per-sample inputs are sliced alike)."""
        rows = (lambda t: t) if sl is None else (lambda t: None if t is None else t[sl[0]:sl[1]])
        skip = None
        for st in steps[lo:hi]:
            if st[0] == "d":
                di, blk = st[1], st[2]
                eo = emb_outs.get(di)
                if state["shared"]:
                    h = blk(h, None if emb_silu is None else emb_silu[:nb], None, emb_out=None if eo is None else eo[:nb],
                            emb_bias=emb_rows.get(di))
                else:
                    h = blk(h, rows(emb_silu), None, skip=skip, emb_out=rows(eo), emb_bias=emb_rows.get(di))
                skip = None
            elif st[0] == "c":
                if state["shared"]:  # the replicas diverge here
                    rep = lambda t: ops.repeat_batch(t, repeat)   # per-channel statistics of the tensors travel along
                    h, hs[:] = rep(h), [rep(t) for t in hs]
                    state["shared"] = False
                h = run_context(h, st[1], st[2], st[3], sl)
            elif st[0] == "save":
                hs.append(h)
            elif st[0] == "load":
                skip = hs.pop()
        assert skip is None
        return h

    hs = []
    region = None
    if BATCH_FORK != "0" and h.is_cuda and h.dim() == 4 and all(len(st[1]) == 1 for st in steps if st[0] == "c"):
        region = _fork_region(steps, h.shape[-2] * h.shape[-1], BATCH_FORK_HW)
    if region is None:
        h = walk(0, len(steps), h, hs)
    else:
        a, b = region
        h = walk(0, a, h, hs)
        B = h.shape[0]
        if state["shared"] or B < 2 or B % 2 or (BATCH_FORK != "1" and B < BATCH_FORK_MIN):
            h = walk(a, len(steps), h, hs)
        else:
            # two half-batch branches: samples 0 .. B/2 - 1 on the current stream, the rest on a side stream (inside the sampler's
            # HIP graph: parallel branches).  Batch slices of channels-last tensors are contiguous views; the region's skip tensors
            # stay inside it; nothing in it is consumed with producer statistics outside (the Upsample conv behind it takes none).
            # (cached context K/V of the region's blocks are projected -- or refreshed -- HERE, in front of the fork: a branch that
            # projected them on first use would hand the other branch a tensor its stream has not waited for)
            for st in steps[a:b]:
                if st[0] == "c":
                    prepare_context(st[1], st[2])
            # (the FIRST forward of a geometry orders the side branch behind the whole main branch: weight packs are built lazily at
            # first use, and a pack built by one branch would reach the other unsynchronised; both halves have the same shapes, so
            # the main branch builds every pack the side branch needs)
            key = (B,) + tuple(h.shape[1:]) + tuple(id(st[1][0]) for st in steps[a:b] if st[0] == "c")[:1]
            first = key not in warm
            warm.add(key)
            main = torch.cuda.current_stream()
            side = _side_streams(h.device, 1, kind="batch")[0]
            fork = torch.cuda.Event()
            if not first:
                fork.record(main)
            h0 = walk(a, b, h[:B // 2], [], (0, B // 2))
            if first:
                fork.record(main)
            with torch.cuda.stream(side):
                side.wait_event(fork)
                h1 = walk(a, b, h[B // 2:], [], (B // 2, B))
                done = torch.cuda.Event()
                done.record(side)
            h1.record_stream(main)
            main.wait_event(done)
            h = torch.cat([h0, h1], 0)
            h = walk(b, len(steps), h, hs)
    assert not state["shared"], "run_unet(repeat > 1) needs a context block in the input or middle stage"
    return h if h.dim() == 2 else ops.nhwc_to_nchw(h)   # 0-D (text-latent) data flow ends in [B, D]


@register("vd_v2_0")
class VD_v2_0(nn.Module):
    def __init__(self, vae_cfg_list, ctx_cfg_list, diffuser_cfg_list, global_layer_ptr=None, parameterization="eps",
                 timesteps=1000, use_ema=False, beta_schedule="linear", beta_linear_start=1e-4, beta_linear_end=2e-2,
                 given_betas=None, cosine_s=8e-3, loss_type="l2", l_simple_weight=1., l_elbo_weight=0.,
                 v_posterior=0., learn_logvar=False, logvar_init=0, latent_scale_factor=None):
        super().__init__()
        assert parameterization in ["eps", "x0"], 'currently only supporting "eps" and "x0"'
        assert not use_ema, "EMA is a training feature (the reference itself cannot enable it, vd.py:84)"
        self.parameterization = parameterization
        print_log("Running in {} mode".format(parameterization))
        self.vae = self.get_model_list(vae_cfg_list)
        self.ctx = self.get_model_list(ctx_cfg_list)
        self.diffuser = self.get_model_list(diffuser_cfg_list)
        self.global_layer_ptr = global_layer_ptr
        assert self.check_diffuser(), "diffuser layers are not aligned!"
        self.use_ema = use_ema
        self.loss_type, self.l_simple_weight, self.l_elbo_weight = loss_type, l_simple_weight, l_elbo_weight
        self.v_posterior = v_posterior
        self.device = "cpu"
        self.register_schedule(given_betas=given_betas, beta_schedule=beta_schedule, timesteps=timesteps,
                               linear_start=beta_linear_start, linear_end=beta_linear_end, cosine_s=cosine_s)
        self.learn_logvar = learn_logvar
        self.logvar = torch.full(fill_value=logvar_init, size=(self.num_timesteps,))
        self.latent_scale_factor = {} if latent_scale_factor is None else dict(latent_scale_factor)
        self.parameter_group = {}
        for namei, diffuseri in self.diffuser.items():
            self.parameter_group.update({"diffuser_{}_{}".format(namei, pgni): pgi
                                         for pgni, pgi in diffuseri.parameter_group.items()})

    def to(self, device):
        # reference quirk kept on purpose: returns None and records the device (vd.py:114-116)
        self.device = device
        super().to(device)

    def get_model_list(self, cfg_list):
        net = nn.ModuleDict()
        for name, cfg in cfg_list:
            net[name] = String_Reg_Buffer(cfg) if isinstance(cfg, str) else get_model()(cfg)
        return net

    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        betas = given_betas if given_betas is not None else make_beta_schedule(
            beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        alphas = 1. - betas
        alphas_cumprod = np.cumprod(alphas, axis=0)
        alphas_cumprod_prev = np.append(1., alphas_cumprod[:-1])
        timesteps, = betas.shape
        self.num_timesteps = int(timesteps)
        self.linear_start, self.linear_end = linear_start, linear_end
        assert alphas_cumprod.shape[0] == self.num_timesteps, "alphas have to be defined for each timestep"
        to_torch = partial(torch.tensor, dtype=torch.float32)
        reg = lambda name, arr: self.register_buffer(name, to_torch(arr))
        reg("betas", betas)
        reg("alphas_cumprod", alphas_cumprod)
        reg("alphas_cumprod_prev", alphas_cumprod_prev)
        reg("sqrt_alphas_cumprod", np.sqrt(alphas_cumprod))
        reg("sqrt_one_minus_alphas_cumprod", np.sqrt(1. - alphas_cumprod))
        reg("log_one_minus_alphas_cumprod", np.log(1. - alphas_cumprod))
        reg("sqrt_recip_alphas_cumprod", np.sqrt(1. / alphas_cumprod))
        reg("sqrt_recipm1_alphas_cumprod", np.sqrt(1. / alphas_cumprod - 1))
        posterior_variance = (1 - self.v_posterior) * betas * (1. - alphas_cumprod_prev) / (1. - alphas_cumprod) \
            + self.v_posterior * betas
        reg("posterior_variance", posterior_variance)
        reg("posterior_log_variance_clipped", np.log(np.maximum(posterior_variance, 1e-20)))
        reg("posterior_mean_coef1", betas * np.sqrt(alphas_cumprod_prev) / (1. - alphas_cumprod))
        reg("posterior_mean_coef2", (1. - alphas_cumprod_prev) * np.sqrt(alphas) / (1. - alphas_cumprod))
        # host copies in float32: the sampler reads schedule scalars without a device sync
        self._host_schedule = {"alphas_cumprod": alphas_cumprod.astype(np.float32),
                               "sqrt_alphas_cumprod": np.sqrt(alphas_cumprod).astype(np.float32),
                               "sqrt_one_minus_alphas_cumprod": np.sqrt(1. - alphas_cumprod).astype(np.float32)}

    def host_schedule(self, name):
        """fp32 host copy of a schedule buffer (re-derived from the buffer if a checkpoint overwrote it)."""
        return self._host_schedule[name]

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        for name in list(self._host_schedule):
            if prefix + name in state_dict:
                self._host_schedule[name] = state_dict[prefix + name].detach().float().cpu().numpy()

    def check_diffuser(self):
        orders = [d.layer_order for d in self.diffuser.values()]
        return all(o == orders[0] for o in orders)

    # ---- q(x_t | x_0) ----------------------------------------------------------------------------
    def q_sample(self, x_start, t, noise=None):
        """sqrt(acp_t) x0 + sqrt(1 - acp_t) noise (reference vd.py:221-224); fp16 in / fp16 out on the device."""
        if noise is None:
            noise = torch.randn_like(x_start)
        sa = extract_into_tensor(self.sqrt_alphas_cumprod, t, (t.shape[0],)).float().contiguous()
        sb = extract_into_tensor(self.sqrt_one_minus_alphas_cumprod, t, (t.shape[0],)).float().contiguous()
        dt = x_start.dtype
        out = ops.q_sample(x_start.to(torch.float16).contiguous(), noise.to(torch.float16).contiguous(), sa.view(-1), sb.view(-1))
        return out.to(dt)

    # ---- VAE / context encoders --------------------------------------------------------------------
    @torch.no_grad()
    def vae_encode(self, x, which, **kwargs):
        scale = (self.latent_scale_factor or {}).get(which, None)
        if scale is not None and hasattr(self.vae[which], "encode_scaled"):
            return self.vae[which].encode_scaled(x, scale=scale, **kwargs)
        z = self.vae[which].encode(x, **kwargs)
        return z if scale is None else scale * z

    @torch.no_grad()
    def vae_decode(self, z, which, **kwargs):
        scale = (self.latent_scale_factor or {}).get(which, None)
        if scale is not None and hasattr(self.vae[which], "decode_scaled"):
            return self.vae[which].decode_scaled(z, inv_scale=1. / scale, **kwargs)
        if scale is not None:
            z = 1. / scale * z
        return self.vae[which].decode(z, **kwargs)

    @torch.no_grad()
    def ctx_encode(self, x, which, **kwargs):
        if which.find("vae_") == 0:
            return self.vae[which[4:]].encode(x, **kwargs)
        return self.ctx[which].encode(x, **kwargs)

    # ---- the UNet forward ----------------------------------------------------------------------------
    def _emb_silu(self, glayer_ptr, timesteps):
        net = self.diffuser[glayer_ptr]
        return net.time_embed.forward_silu(timestep_embedding(timesteps, net.model_channels))

    @torch.no_grad()
    def precompute_step_emb(self, x_type, timesteps, multicontext=False):
        """Everything of the UNet forward that depends on t alone -- timestep_embedding -> time_embed MLP -> SiLU -> every
        ResBlock's emb_layers projection (reference vd.py:339-349, openaimodel.py:2627-2633, :263) -- for ALL `timesteps` [S]
        at once: (fp16 [S, total], layout) of UNetModel2D_Next.precompute_emb_table, or None when the data flow has no such
        table (0-D text-latent flow).  A caller whose batch shares one timestep per step hands row i to apply_model* as
        x_info['emb_rows'] (DDIMSampler does, through a static buffer the step graph reads)."""
        net = self.diffuser[x_type]
        if not hasattr(net, "precompute_emb_table"):
            return None
        glayer_ptr = x_type if (multicontext or self.global_layer_ptr is None) else self.global_layer_ptr
        return net.precompute_emb_table(self._emb_silu(glayer_ptr, timesteps))

    @staticmethod
    def _prep(x):
        if not x.is_cuda:
            raise RuntimeError("VD_v2_0.apply_model needs the latent on the GPU: this package has no CPU path")
        return x.to(torch.float16).contiguous()

    @torch.no_grad()
    def apply_model(self, x_info, timesteps, c_info):
        x_type, x = x_info["type"], x_info["x"]
        c_type, c = c_info["type"], c_info["c"]
        glayer_ptr = x_type if self.global_layer_ptr is None else self.global_layer_ptr
        rows = x_info.get("emb_rows")   # extension key: the t-only part precomputed for a batch that shares its timestep
        emb = self._emb_silu(glayer_ptr, timesteps) if rows is None else None
        spec = (self.diffuser[c_type].context_blocks, self._prep(c), 1.0, c_info.get("kv_cache"))
        return run_unet(self.diffuser[x_type], [spec], self._prep(x), emb, repeat=int(x_info.get("repeat", 1)),
                        emb_rows=rows).to(x.dtype)

    @torch.no_grad()
    def apply_model_multicontext(self, x_info, timesteps, c_info_list, mixing_type="attention"):
        """c_info_list: [{type, c, ratio}, ...]; 'attention' mixing = ratio-weighted sum of the context blocks."""
        x_type, x = x_info["type"], x_info["x"]
        assert mixing_type in ("attention", "layer")
        rows = x_info.get("emb_rows")
        # reference takes time_embed from diffuser[x_type] here (vd.py:415-417)
        emb = self._emb_silu(x_type, timesteps) if rows is None else None
        specs = [(self.diffuser[ci["type"]].context_blocks, self._prep(ci["c"]), ci["ratio"], ci.get("kv_cache"))
                 for ci in c_info_list]
        return run_unet(self.diffuser[x_type], specs, self._prep(x), emb, mixing_type,
                        repeat=int(x_info.get("repeat", 1)), emb_rows=rows).to(x.dtype)
