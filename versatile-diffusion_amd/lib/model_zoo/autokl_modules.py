"""Encoder / Decoder of the KL-f8 autoencoder on the HIP kernel library, channels-last fp16.

Module tree and state-dict keys follow the reference (lib/model_zoo/autokl_modules.py there: Normalize :38-39,
Upsample :42-57, Downsample :60-79, ResnetBlock :82-141, AttnBlock :150-202, Encoder :368-459, Decoder :462-568).
Convs are implicit MFMA GEMMs with the nearest-2x upsample / the asymmetric (0,1,0,1) stride-2 padding folded into
the gather; GroupNorm(eps 1e-6)+swish is one pass; the single-head mid attention (C=512, N=HW) is one fused q|k|v
projection + vd_attention_f16's wide single-head kernel (online softmax, no [N, N] scores; other widths: batched GEMMs with
fp32 logits and a row-softmax kernel)."""
import torch
import torch.nn as nn

from vd_hip import ops

from .hip_layers import Conv2d, GroupNorm, PackCache, _h


def Normalize(in_channels, num_groups=32):
    return GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


class Upsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        assert with_conv
        self.with_conv = with_conv
        self.conv = Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        return self.conv(x, ups=1, want_stats=True)   # feeds a ResnetBlock's GroupNorm: statistics from the conv's epilogue


class Downsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        assert with_conv
        self.with_conv = with_conv
        self.conv = Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(x, pad_hi=1, want_stats=True)  # F.pad(x, (0,1,0,1)) then a valid stride-2 conv


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=0):
        super().__init__()
        assert not conv_shortcut and temb_channels == 0
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.norm1 = Normalize(in_channels)
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if in_channels != out_channels:
            self.nin_shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x, temb=None):
        # every tensor between the blocks feeds a GroupNorm: its producer emits the per-channel statistics (csrc/gn_fused.hip)
        h = self.conv1(self.norm1(x, silu=True), want_stats=True)
        h = self.norm2(h, silu=True)
        res = self.nin_shortcut(x) if self.in_channels != self.out_channels else x
        return self.conv2(h, res=res, want_stats=True)


class AttnBlock(nn.Module, PackCache):
    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.k = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.v = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.proj_out = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x):
        B, H, W, C = x.shape
        N = H * W
        hn = self.norm(x, silu=False)
        if C in (128, 256, 512):
            # one wide head: q | k | v from ONE 1x1 projection, then vd_attention_f16's single-head kernel (head dim split over
            # the waves of a block, online softmax): no [N, N] score tensor -- the reference's two torch.bmm around a softmax
            # (autokl_modules.py:186-198 there) materialise 67 MB per sample at 512x512, 340 MB at 768x768
            wqkv, bqkv = self._packed("qkv", (self.q.weight, self.q.bias, self.k.weight, self.k.bias, self.v.weight, self.v.bias),
                                      lambda: (torch.cat([_h(m.weight).reshape(C, C) for m in (self.q, self.k, self.v)], 0).contiguous(),
                                               torch.cat([_h(m.bias) for m in (self.q, self.k, self.v)], 0).contiguous()))
            qkv = ops.linear(hn.view(B, N, C), wqkv, bqkv)
            o = ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], 1, scale=float(C) ** -0.5)
            return self.proj_out(o.view(B, H, W, C), res=x, want_stats=True)
        q = self.q(hn).view(B, N, C)
        k = self.k(hn).view(B, N, C)
        wv, bv = self._packed("v", (self.v.weight, self.v.bias),
                              lambda: (_h(self.v.weight).reshape(C, C).contiguous(), _h(self.v.bias)))
        # other widths: V^T[b] = Wv hn[b]^T + bv  -> [B, C, N], fp32 scores, row softmax, P.V GEMM
        vt = ops.gemm(wv, hn.view(B, N, C), bias=bv, bias_along_m=True, M=C, N=N, K=C, batch=B,
                      strides=(0, N * C, C * N, 0))
        s = ops.gemm(q, k, M=N, N=N, K=C, batch=B, strides=(N * C, N * C, N * N, 0), alpha=float(C) ** -0.5,
                     out_f32=True)
        p = ops.softmax_rows(s)
        o = ops.gemm(p, vt, M=N, N=C, K=N, batch=B, strides=(N * N, C * N, N * C, 0))
        return self.proj_out(o.view(B, H, W, C), res=x, want_stats=True)


def make_attn(in_channels, attn_type="vanilla"):
    assert attn_type == "vanilla", "only the vanilla AttnBlock is used by kl-f8"
    return AttnBlock(in_channels)


class Encoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.conv_in = Conv2d(in_channels, ch, kernel_size=3, stride=1, padding=1)
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in, block_out = ch * in_ch_mult[i_level], ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            down = nn.Module()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res = curr_res // 2
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = Conv2d(block_in, 2 * z_channels if double_z else z_channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x_nchw, in_scale=1.0, in_shift=0.0):
        """x_nchw fp16 image; the affine (x*2-1 in AutoencoderKL.encode) is applied inside the first gather."""
        h = self.conv_in(x_nchw, in_layout="nchw", in_scale=in_scale, in_shift=in_shift, want_stats=True)
        for i_level in range(self.num_resolutions):
            for i_block in range(self.num_res_blocks):
                h = self.down[i_level].block[i_block](h)
                if len(self.down[i_level].attn) > 0:
                    h = self.down[i_level].attn[i_block](h)
            if i_level != self.num_resolutions - 1:
                h = self.down[i_level].downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        return self.conv_out(self.norm_out(h, silu=True))


class Decoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", **ignorekwargs):
        super().__init__()
        assert not give_pre_end and not tanh_out
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        self.conv_in = Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res = curr_res * 2
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)

    def forward(self, z_nhwc):
        self.last_z_shape = z_nhwc.shape
        h = self.conv_in(z_nhwc, want_stats=True)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        for i_level in reversed(range(self.num_resolutions)):
            for i_block in range(self.num_res_blocks + 1):
                h = self.up[i_level].block[i_block](h)
                if len(self.up[i_level].attn) > 0:
                    h = self.up[i_level].attn[i_block](h)
            if i_level != 0:
                h = self.up[i_level].upsample(h)
        return self.conv_out(self.norm_out(h, silu=True))
