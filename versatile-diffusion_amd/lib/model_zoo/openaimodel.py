"""UNet of the Versatile-Diffusion image flow on the HIP kernel library.

Registry names, constructor arguments, attribute names and state-dict keys follow the reference
(lib/model_zoo/openaimodel.py there: ResBlock :162-274, Upsample :89-117, Downsample :133-159,
TimestepEmbedSequential :72-86, UNetModel2D_Next :2575-2812, UNetModel0D_Next :2814-2975), so
`get_model()(model_cfg_bank()('openai_unet_2d_v1'))` and reference checkpoints work unchanged.  The execution is
new: channels-last fp16 activations, GroupNorm+SiLU in one pass, 3x3 convs as implicit MFMA GEMMs that read the
skip-connection concat / nearest-2x upsample in place and fold bias, the timestep-embedding broadcast and the
residual add into their epilogue.
"""
import copy
import os
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from vd_hip import ops

from .attention import SpatialTransformer
from .common.get_model import register
from .diffusion_utils import timestep_embedding  # noqa: F401  (re-exported like the reference does)
from .hip_layers import Conv2d, GroupNorm, Linear, PackCache, SiLU, _h

symbol = "openai"


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


class TimestepBlock(nn.Module):
    """Marker: forward(x, emb) takes the timestep embedding."""


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """Dispatches (x, emb, context) to its single child by type, like the reference; additionally threads the
    channels-last extras: `skip` (tensor concatenated on channels in place), the context K/V cache and the
    mixing epilogue."""

    def forward(self, x, emb, context=None, skip=None, emb_out=None, emb_bias=None, **ctx_kw):
        for layer in self:
            if isinstance(layer, ResBlock):
                x = layer(x, emb, skip=skip, emb_out=emb_out, emb_bias=emb_bias)
            elif isinstance(layer, TimestepBlock):
                x = layer(x, emb, skip=skip, emb_out=emb_out)
            elif isinstance(layer, SpatialTransformer):
                x = layer(x, context, **ctx_kw)
            elif isinstance(layer, OutputHead):
                x = layer(x)
            else:
                x = layer(x)
            skip = None
        return x


class TimeEmbed(nn.Sequential, PackCache):
    """Linear -> SiLU -> Linear (reference openaimodel.py:2627-2633).  `forward_silu` returns SiLU(emb), which is
    what every consumer (ResBlock.emb_layers) actually reads, with both activations fused in GEMM epilogues."""

    def __init__(self, model_channels, time_embed_dim):
        super().__init__(Linear(model_channels, time_embed_dim), SiLU(), Linear(time_embed_dim, time_embed_dim))

    def forward(self, t_emb):
        return self[2](self[0](t_emb, act=ops.ACT_SILU))

    def forward_silu(self, t_emb):
        return self[2](self[0](t_emb, act=ops.ACT_SILU), act=ops.ACT_SILU)


class Upsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        assert use_conv and dims == 2
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.conv = Conv2d(self.channels, self.out_channels, 3, padding=padding)

    def forward(self, x):
        return self.conv(x, ups=1, want_stats=True)  # nearest 2x folded into the conv's gather; feeds a ResBlock's GroupNorm


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        assert use_conv and dims == 2
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.op = Conv2d(self.channels, self.out_channels, 3, stride=2, padding=padding)

    def forward(self, x):
        return self.op(x, want_stats=True)


# development switch (VD_RES_FORK=1): the skip 1x1 convolution of a ResBlock on a side stream beside the block's main path.
# Measured (tools/gpu_r03_ab.sh): 11.22 vs 11.12 ms per graph-replayed forward -- parallel branches lose on this stack.
RES_FORK = os.environ.get("VD_RES_FORK", "0") == "1"
_side_streams = {}


def _side_stream(device):
    s = _side_streams.get(device.index)
    if s is None:
        s = torch.cuda.Stream(device=device)
        _side_streams[device.index] = s
    return s


class ResBlock(TimestepBlock, PackCache):
    """GN+SiLU -> conv3x3 (+bias +emb) -> GN+SiLU -> conv3x3 (+bias +skip(x)); use_scale_shift_norm=False."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False, use_scale_shift_norm=False,
                 dims=2, use_checkpoint=False, up=False, down=False):
        super().__init__()
        assert dims == 2 and not use_scale_shift_norm and not up and not down and not use_conv
        self.channels = channels
        self.emb_channels = emb_channels
        self.dropout = dropout
        self.out_channels = out_channels or channels
        self.use_checkpoint = use_checkpoint
        self.in_layers = nn.Sequential(GroupNorm(32, channels), SiLU(), Conv2d(channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(SiLU(), Linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(GroupNorm(32, self.out_channels), SiLU(), nn.Dropout(p=dropout),
                                        zero_module(Conv2d(self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = Conv2d(channels, self.out_channels, 1)

    def forward(self, x, emb_silu, skip=None, emb_out=None, emb_bias=None):
        """x [B,H,W,C0] (++ skip [B,H,W,C1] on channels); emb_silu = SiLU(time embedding) [B, emb_channels].
        emb_out: optional pre-computed `emb_layers[1].weight @ emb_silu` WITHOUT its bias (UNet-level batched GEMM,
        see UNetModel2D_Next.precompute_emb); the bias then rides in the conv's bias vector.
        emb_bias: [Cout] fp16 = conv bias + emb_layers bias + emb_layers weight @ SiLU(emb) for ONE timestep shared by the whole
        batch (UNetModel2D_Next.precompute_emb_table: the sampler computes it for all steps once per sample() call); the
        embedding then needs no row vector in the conv epilogue at all."""
        B, H, W, _ = x.shape
        bias1 = None
        if emb_bias is not None:
            emb_out, bias1 = None, emb_bias
        elif emb_out is None:
            emb_out = self.emb_layers[1](emb_silu)
        else:
            conv, lin = self.in_layers[2], self.emb_layers[1]
            bias1 = self._packed("b1", (conv.bias, lin.bias),
                                 lambda: (conv.bias.detach().float() + lin.bias.detach().float()).to(torch.float16).contiguous())
        fork = RES_FORK and x.is_cuda and not isinstance(self.skip_connection, nn.Identity)
        if fork:
            # the 1x1 skip convolution depends only on the block's input: it runs on a side stream beside GroupNorm -> conv ->
            # GroupNorm of the main path and is joined in front of the second conv, which consumes it as its residual.  Its
            # output is allocated on the MAIN stream (the caching allocator ties a block to its allocation stream).
            main = torch.cuda.current_stream()
            side = _side_stream(x.device)
            res = torch.empty((B, H, W, self.out_channels), dtype=torch.float16, device=x.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self.skip_connection(x, x1=skip, out=res)
        # every tensor between the layers of the data flow feeds a GroupNorm (the next block's, or a later skip concat): its
        # producer emits the per-channel statistics (want_stats), the norms read them instead of measuring their input
        h = self.in_layers[0](x, x1=skip, silu=True)
        # conv1 -> GroupNorm -> SiLU: the conv (or the kernel that sums its split-K slabs) emits the statistics the norm reads
        n2 = self.out_layers[0]
        h = self.in_layers[2](h, rowvec=emb_out, rows_per_batch=H * W if emb_out is not None else 0, bias=bias1, want_stats=True)
        h = n2(h, silu=True)
        if fork:
            main.wait_stream(side)
        elif isinstance(self.skip_connection, nn.Identity):
            assert skip is None
            res = x
        else:
            # skip_connection(cat(x, skip)) + conv2(h): the 1x1 convolution rides in the second conv as extra K (one-tap chunks
            # behind the 3x3 chunks of the halo kernel) where that kernel runs -- no separate GEMM, no residual round trip
            sc, c2 = self.skip_connection, self.out_layers[3]
            wsk, _ = sc._w()
            bsum = self._packed("b2s", (c2.bias, sc.bias),
                                lambda: (c2.bias.detach().float() + sc.bias.detach().float()).to(torch.float16).contiguous())
            wsk_stream = sc._w_stream_1x1() if (ops.WSTREAM and H == 8 and W == 8 and sc.in_channels % 64 == 0) else None
            out = c2(h, bias=bsum, skip=(x, skip, wsk, wsk_stream), want_stats=True)
            if out is not None:
                return out
            res = sc(x, x1=skip)
        return self.out_layers[3](h, res=res, want_stats=True)


class OutputHead(nn.Sequential):
    """normalization -> SiLU -> zero conv3x3 (reference openaimodel.py:2732-2737)."""

    def __init__(self, ch, out_channels):
        super().__init__(GroupNorm(32, ch), SiLU(), zero_module(Conv2d(ch, out_channels, 3, padding=1)))

    def forward(self, x):
        return self[2](self[0](x, silu=True))


class InputConv(Conv2d):
    """First data layer: 3x3 conv on the NCHW latent the sampler hands over (small-Cin im2col + MFMA GEMM)."""

    def forward(self, x):
        return super().forward(x, in_layout="nchw", want_stats=True)


@register("openai_unet_2d_next")
class UNetModel2D_Next(nn.Module, PackCache):
    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, context_dim,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, use_checkpoint=False, num_heads=8,
                 num_head_channels=None, parts=("global", "data", "context")):
        super().__init__()
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        if isinstance(num_res_blocks, int):
            self.num_res_blocks = len(channel_mult) * [num_res_blocks]
        else:
            if len(num_res_blocks) != len(channel_mult):
                raise ValueError("provide num_res_blocks either as an int (globally constant) or "
                                 "as a list/tuple (per-level) with the same length as channel_mult")
            self.num_res_blocks = list(num_res_blocks)
        self.attention_resolutions = attention_resolutions
        self.context_dim = context_dim
        self.dropout = dropout
        self.channel_mult = channel_mult
        self.conv_resample = conv_resample
        self.use_checkpoint = use_checkpoint
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        assert (num_heads is None) + (num_head_channels is None) == 1, \
            "One of num_heads or num_head_channels need to be set"
        self._init_parts(parts, model_channels)

        time_embed_dim = model_channels * 4
        res = partial(ResBlock, emb_channels=time_embed_dim, dropout=dropout, dims=2, use_checkpoint=use_checkpoint,
                      use_scale_shift_norm=False) if self.dlayer_included else (lambda **kw: None)
        xattn = partial(SpatialTransformer, context_dim=context_dim, disable_self_attn=False) \
            if self.clayer_included else (lambda **kw: None)

        self.add_data_layer(InputConv(in_channels, model_channels, 3, padding=1) if self.dlayer_included else None)
        self.layer_sequence_ordering.append("save_hidden_feature")
        input_block_chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(self.num_res_blocks[level]):
                self.add_data_layer(res(channels=ch, out_channels=mult * model_channels))
                ch = mult * model_channels
                if ds in attention_resolutions:
                    d_head, n_heads = self.get_d_head_n_heads(ch)
                    self.add_context_layer(xattn(in_channels=ch, d_head=d_head, n_heads=n_heads))
                input_block_chans.append(ch)
                self.layer_sequence_ordering.append("save_hidden_feature")
            if level != len(channel_mult) - 1:
                self.add_data_layer(Downsample(ch, use_conv=True, dims=2, out_channels=ch) if self.dlayer_included else None)
                input_block_chans.append(ch)
                self.layer_sequence_ordering.append("save_hidden_feature")
                ds *= 2
        self.i_order = copy.deepcopy(self.layer_sequence_ordering)
        self.layer_sequence_ordering = []

        self.add_data_layer(res(channels=ch))
        d_head, n_heads = self.get_d_head_n_heads(ch)
        self.add_context_layer(xattn(in_channels=ch, d_head=d_head, n_heads=n_heads))
        self.add_data_layer(res(channels=ch))
        self.m_order = copy.deepcopy(self.layer_sequence_ordering)
        self.layer_sequence_ordering = []

        for level, mult in list(enumerate(channel_mult))[::-1]:
            for _ in range(self.num_res_blocks[level] + 1):
                self.layer_sequence_ordering.append("load_hidden_feature")
                ich = input_block_chans.pop()
                self.add_data_layer(res(channels=ch + ich, out_channels=model_channels * mult))
                ch = model_channels * mult
                if ds in attention_resolutions:
                    d_head, n_heads = self.get_d_head_n_heads(ch)
                    self.add_context_layer(xattn(in_channels=ch, d_head=d_head, n_heads=n_heads))
            if level != 0:
                self.add_data_layer(Upsample(ch, conv_resample, dims=2, out_channels=ch) if self.dlayer_included else None)
                ds //= 2
        self.add_data_layer(OutputHead(ch, out_channels) if self.dlayer_included else None)
        self._finish_orders()

    # ---- shared bookkeeping (2D and 0D nets) -------------------------------------------------
    def _init_parts(self, parts, model_channels):
        self.parts = list(parts) if isinstance(parts, (list, tuple)) else [parts]
        self.glayer_included = "global" in self.parts
        self.dlayer_included = "data" in self.parts
        self.clayer_included = "context" in self.parts
        self.layer_sequence_ordering = []
        if self.glayer_included:
            self.time_embed = TimeEmbed(model_channels, model_channels * 4)
        if self.dlayer_included:
            self.data_blocks = nn.ModuleList([])
        if self.clayer_included:
            self.context_blocks = nn.ModuleList([])

    def _finish_orders(self):
        self.o_order = copy.deepcopy(self.layer_sequence_ordering)
        self.layer_order = copy.deepcopy(self.i_order + self.m_order + self.o_order)
        del self.layer_sequence_ordering
        self.parameter_group = {}
        if self.glayer_included:
            self.parameter_group["global"] = self.time_embed
        if self.dlayer_included:
            self.parameter_group["data"] = self.data_blocks
        if self.clayer_included:
            self.parameter_group["context"] = self.context_blocks

    def precompute_emb(self, emb_silu):
        """All ResBlocks' `emb_layers` projections of this step in one batched GEMM per distinct width (3 launches
        instead of 22; the reference runs one Linear per ResBlock, openaimodel.py:263).  Returns
        {data_block_index: [B, Cout] fp16} without the Linear bias (ResBlock folds it into its conv bias)."""
        res = [(i, blk[0]) for i, blk in enumerate(self.data_blocks) if isinstance(blk[0], ResBlock)]
        groups = {}
        for i, rb in res:
            groups.setdefault(rb.out_channels, []).append((i, rb))

        def build():
            return {c: torch.stack([_h(rb.emb_layers[1].weight) for _, rb in lst]).contiguous() for c, lst in groups.items()}

        stacked = self._packed("emb_w", tuple(rb.emb_layers[1].weight for _, rb in res), build)
        out = {}
        Bn, E = emb_silu.shape
        for c, lst in groups.items():
            nb = len(lst)
            o = ops.gemm(emb_silu, stacked[c], M=Bn, N=c, K=E, batch=nb, strides=(0, c * E, Bn * c, 0),
                         out_shape=(nb, Bn, c))
            for j, (i, _) in enumerate(lst):
                out[i] = o[j]
        return out

    def precompute_emb_table(self, emb_silu_steps):
        """emb_silu_steps [S, E] = SiLU(time embedding) of S timesteps -> (fp16 [S, total], {data block index: (offset, Cout)}):
        per timestep and ResBlock the COMPLETE bias vector of the block's first conv (conv bias + emb_layers bias +
        emb_layers weight @ emb; reference openaimodel.py:263-266 adds emb_out to h behind the conv).  Everything here depends
        on t only, so a sampler whose batch shares one timestep per step computes it once for all steps (three batched GEMMs
        with M = S) and the step graph carries no time-embedding launches.  None for nets without ResBlocks (0-D flow)."""
        res = [(i, blk[0]) for i, blk in enumerate(self.data_blocks) if isinstance(blk[0], ResBlock)]
        if not res:
            return None
        outs = self.precompute_emb(emb_silu_steps)
        cols, layout, off = [], {}, 0
        for i, rb in res:
            conv, lin = rb.in_layers[2], rb.emb_layers[1]
            b = conv.bias.detach().float() + lin.bias.detach().float()
            cols.append((outs[i].float() + b[None, :]).to(torch.float16))
            layout[i] = (off, rb.out_channels)
            off += rb.out_channels
        return torch.cat(cols, dim=1).contiguous(), layout

    def get_d_head_n_heads(self, ch):
        if self.num_head_channels is None:
            return ch // self.num_heads, self.num_heads
        return self.num_head_channels, ch // self.num_head_channels

    def add_data_layer(self, layer):
        if self.dlayer_included:
            layers = list(layer) if isinstance(layer, (list, tuple)) else [layer]
            self.data_blocks.append(TimestepEmbedSequential(*layers))
        self.layer_sequence_ordering.append("d")

    def add_context_layer(self, layer):
        if self.clayer_included:
            layers = list(layer) if isinstance(layer, (list, tuple)) else [layer]
            self.context_blocks.append(TimestepEmbedSequential(*layers))
        self.layer_sequence_ordering.append("c")

    def forward(self, x, timesteps, context):
        """Stand-alone forward (NCHW in / NCHW out).  The reference's own .forward walks i_order twice (a bug,
        openaimodel.py:2801) and is never used by VD_v2_0; this one walks o_order as VD_v2_0.apply_model does."""
        from .vd import run_unet
        emb = self.time_embed.forward_silu(timestep_embedding(timesteps, self.model_channels))
        return run_unet(self, [(self.context_blocks, context, 1.0, None)], x, emb)


def _md_perm(w, out_md, in_md):
    """Re-order a dense weight [prod(out_md), prod(in_md)] from the reference's flattening of a [C, sdim, 1] feature
    (index c * sdim + s) to this package's channels-last one (s * C + c) on whichever side is multi-dimensional."""
    o_all, i_all = w.shape
    if len(out_md) == 3:
        co, so = out_md[0], out_md[1]
        w = w.view(co, so, i_all).permute(1, 0, 2).reshape(o_all, i_all)
    if len(in_md) == 3:
        ci, si = in_md[0], in_md[1]
        w = w.view(o_all, ci, si).permute(0, 2, 1).reshape(o_all, i_all)
    return w.contiguous()


def _md_perm_vec(v, md):
    if v is None or len(md) != 3:
        return v
    return v.view(md[0], md[1]).t().reshape(-1).contiguous()


class Linear_MultiDim(nn.Linear, PackCache):
    """Dense layer between multi-dimensional features (reference openaimodel.py:2275-2293), parameters in the
    reference's shapes.  Activations here are channels-last: a [C, sdim, 1] feature is held as [B, sdim, 1, C], so the
    weight is re-ordered once (cached) and the layer is one `vd_gemm_f16` on [B, prod(in)]."""

    def __init__(self, in_features, out_features, *args, **kwargs):
        in_features = [in_features] if isinstance(in_features, int) else list(in_features)
        out_features = [out_features] if isinstance(out_features, int) else list(out_features)
        self.in_features_multidim = in_features
        self.out_features_multidim = out_features
        super().__init__(int(np.prod(in_features)), int(np.prod(out_features)), *args, **kwargs)

    def _w(self):
        return self._packed("w", (self.weight, self.bias), lambda: (
            _md_perm(_h(self.weight), self.out_features_multidim, self.in_features_multidim),
            _md_perm_vec(_h(self.bias), self.out_features_multidim)))

    def forward(self, x):
        w, b = self._w()
        B = x.shape[0]
        y = ops.linear(x.reshape(B, -1), w, b)
        om = self.out_features_multidim
        return y.view(B, om[1], om[2], om[0]) if len(om) == 3 else y


class FCBlock_MultiDim(TimestepBlock, PackCache):
    """FCBlock on a flattened [C, sdim, 1] feature (reference openaimodel.py:2084-2141, 2295-2332): GN32 -> SiLU ->
    dense, + emb, GN32 -> SiLU -> dense, + skip(x).  Parameters keep the reference's keys / shapes (1x1 convs over
    C*sdim "channels"); x is [B, sdim, 1, C] channels-last, optionally with a skip tensor concatenated on C."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_checkpoint=False):
        super().__init__()
        channels = [channels] if isinstance(channels, int) else list(channels)
        c_all = int(np.prod(channels))
        if out_channels is not None:
            out_channels = [out_channels] if isinstance(out_channels, int) else list(out_channels)
            o_all = int(np.prod(out_channels))
        else:
            out_channels, o_all = channels, c_all
        self.channels_multidim, self.out_channels_multidim = channels, out_channels
        self.channels, self.out_channels, self.emb_channels = c_all, o_all, emb_channels
        self.in_layers = nn.Sequential(nn.GroupNorm(32, c_all), nn.SiLU(), nn.Conv2d(c_all, o_all, 1, padding=0))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, o_all))
        self.out_layers = nn.Sequential(nn.GroupNorm(32, o_all), nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(nn.Conv2d(o_all, o_all, 1, padding=0)))
        self.skip_connection = nn.Identity() if o_all == c_all else nn.Conv2d(c_all, o_all, 1, padding=0)

    def _pk(self):
        cm, om = self.channels_multidim, self.out_channels_multidim
        assert len(cm) == 3 and len(om) == 3 and cm[2] == 1 and om[2] == 1, "FCBlock_MultiDim expects [C, sdim, 1] features"
        g1, c1, lin, g2, c2, sk = (self.in_layers[0], self.in_layers[2], self.emb_layers[1], self.out_layers[0],
                                   self.out_layers[3], self.skip_connection)
        tensors = [g1.weight, g1.bias, c1.weight, c1.bias, lin.weight, lin.bias, g2.weight, g2.bias, c2.weight, c2.bias]
        if not isinstance(sk, nn.Identity):
            tensors += [sk.weight, sk.bias]

        def build():
            d = dict(
                g1=(_md_perm_vec(_h(g1.weight), cm), _md_perm_vec(_h(g1.bias), cm)),
                w1=_md_perm(_h(c1.weight).view(self.out_channels, self.channels), om, cm), b1=_md_perm_vec(_h(c1.bias), om),
                we=_md_perm(_h(lin.weight), om, [self.emb_channels]), be=_md_perm_vec(_h(lin.bias), om),
                g2=(_md_perm_vec(_h(g2.weight), om), _md_perm_vec(_h(g2.bias), om)),
                w2=_md_perm(_h(c2.weight).view(self.out_channels, self.out_channels), om, om), b2=_md_perm_vec(_h(c2.bias), om))
            if not isinstance(sk, nn.Identity):
                d["ws"] = _md_perm(_h(sk.weight).view(self.out_channels, self.channels), om, cm)
                d["bs"] = _md_perm_vec(_h(sk.bias), om)
            return d
        return self._packed("fc", tuple(tensors), build)

    def forward(self, x, emb_silu, skip=None, emb_out=None):
        pk = self._pk()
        cm, om = self.channels_multidim, self.out_channels_multidim
        B, S = x.shape[0], x.shape[1]
        x3 = x.reshape(B, S, x.shape[-1])
        s3 = None if skip is None else skip.reshape(B, S, skip.shape[-1])
        assert x3.shape[-1] + (0 if s3 is None else s3.shape[-1]) == cm[0] and S == cm[1]
        g1w, g1b = pk["g1"]
        h = ops.groupnorm0d_silu(x3, g1w.view(S, cm[0]), g1b.view(S, cm[0]), x1=s3, groups=32, eps=1e-5, silu=True)
        e = ops.linear(emb_silu, pk["we"], pk["be"])                       # emb_layers: SiLU (already applied) -> Linear
        h = ops.linear(h.view(B, -1), pk["w1"], pk["b1"], res=e)             # dense + emb_out
        g2w, g2b = pk["g2"]
        h = ops.groupnorm0d_silu(h.view(B, om[1], om[0]), g2w.view(om[1], om[0]), g2b.view(om[1], om[0]), groups=32,
                                 eps=1e-5, silu=True)
        xc = x3 if s3 is None else torch.cat([x3, s3], dim=-1)               # the reference's torch.cat (vd.py:371), 10-40 KB
        if "ws" in pk:
            r = ops.linear(xc.reshape(B, -1), pk["ws"], pk["bs"])
        else:
            r = xc.reshape(B, -1).contiguous()
        y = ops.linear(h.view(B, -1), pk["w2"], pk["b2"], res=r)
        return y.view(B, om[1], om[2], om[0])


class OutputHead0D(nn.Sequential):
    """normalization -> SiLU -> zero Linear_MultiDim (reference openaimodel.py:2953-2958): GroupNorm over the C channels
    of the [C, sdim, 1] feature (per-channel affine: the regular GroupNorm kernel with HW = sdim)."""

    def __init__(self, cur, output_channels):
        super().__init__(GroupNorm(32, cur[0]), SiLU(), zero_module(Linear_MultiDim(cur, [output_channels], bias=True)))

    def forward(self, x):
        return self[2](self[0](x, silu=True))


@register("openai_unet_0d_next")
class UNetModel0D_Next(UNetModel2D_Next):
    """The 'text' diffuser of vd_four_flow: its *context* blocks (SpatialTransformers conditioned on CLIP text)
    are what the image flow uses for text-to-image (vd.py:345 in the reference); its data blocks (FCBlock_MultiDim /
    Linear_MultiDim over [C, sdim, 1] features, held channels-last as [B, sdim, 1, C]) denoise the 768-d text latent of the
    image-to-text / text-variation flows."""

    def __init__(self, input_channels, model_channels, output_channels, context_dim=788, num_noattn_blocks=(2, 2, 2, 2),
                 channel_mult=(1, 2, 4, 8), second_dim=(4, 4, 4, 4), with_attn=(True, True, True, False), num_heads=8,
                 num_head_channels=None, use_checkpoint=False, parts=("global", "data", "context")):
        nn.Module.__init__(self)
        self.input_channels = input_channels
        self.model_channels = model_channels
        self.output_channels = output_channels
        self.num_noattn_blocks = num_noattn_blocks
        self.channel_mult = channel_mult
        self.second_dim = second_dim
        self.with_attn = with_attn
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        self._init_parts(parts, model_channels)
        time_embed_dim = model_channels * 4
        fc = partial(FCBlock_MultiDim, dropout=0, use_checkpoint=use_checkpoint) if self.dlayer_included \
            else (lambda *a, **kw: None)
        lin = Linear_MultiDim if self.dlayer_included else (lambda *a, **kw: None)
        xattn = partial(SpatialTransformer, context_dim=context_dim, disable_self_attn=False) \
            if self.clayer_included else (lambda **kw: None)

        cur = [model_channels, second_dim[0], 1]
        self.add_data_layer(lin([input_channels], cur, bias=True))
        self.layer_sequence_ordering.append("save_hidden_feature")
        chans = [cur]
        for level, (mult, sdim) in enumerate(zip(channel_mult, second_dim)):
            for _ in range(num_noattn_blocks[level]):
                self.add_data_layer(fc(cur, time_embed_dim, out_channels=[mult * model_channels, sdim, 1]))
                cur = [mult * model_channels, sdim, 1]
                if with_attn[level]:
                    d_head, n_heads = self.get_d_head_n_heads(cur[0])
                    self.add_context_layer(xattn(in_channels=cur[0], d_head=d_head, n_heads=n_heads))
                chans.append(cur)
                self.layer_sequence_ordering.append("save_hidden_feature")
            if level != len(channel_mult) - 1:
                self.add_data_layer(lin(cur, cur, bias=True))
                chans.append(cur)
                self.layer_sequence_ordering.append("save_hidden_feature")
        self.i_order = copy.deepcopy(self.layer_sequence_ordering)
        self.layer_sequence_ordering = []

        self.add_data_layer(fc(cur, time_embed_dim))
        d_head, n_heads = self.get_d_head_n_heads(cur[0])
        self.add_context_layer(xattn(in_channels=cur[0], d_head=d_head, n_heads=n_heads))
        self.add_data_layer(fc(cur, time_embed_dim))
        self.m_order = copy.deepcopy(self.layer_sequence_ordering)
        self.layer_sequence_ordering = []

        for level, (mult, sdim) in list(enumerate(zip(channel_mult, second_dim)))[::-1]:
            for _ in range(num_noattn_blocks[level] + 1):
                self.layer_sequence_ordering.append("load_hidden_feature")
                extra = chans.pop()
                self.add_data_layer(fc([cur[0] + extra[0]] + cur[1:], time_embed_dim,
                                       out_channels=[mult * model_channels, sdim, 1]))
                cur = [mult * model_channels, sdim, 1]
                if with_attn[level]:
                    d_head, n_heads = self.get_d_head_n_heads(cur[0])
                    self.add_context_layer(xattn(in_channels=cur[0], d_head=d_head, n_heads=n_heads))
            if level != 0:
                self.add_data_layer(lin(cur, cur, bias=True))
        head = OutputHead0D(cur, output_channels) if self.dlayer_included else None
        self.add_data_layer(head)
        self._finish_orders()

    def forward(self, x, timesteps, context):
        raise NotImplementedError("the 0-D diffuser has no global layers of its own: call VD_v2_0.apply_model (x_type='text')")
