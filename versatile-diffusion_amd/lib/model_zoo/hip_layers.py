"""Channels-last layer primitives that run on the HIP kernel library (vd_hip).

Every class subclasses the torch module the reference uses in the same slot (nn.Conv2d, nn.Linear,
nn.GroupNorm, nn.LayerNorm) so parameter names, shapes, initialisation and state-dict keys are identical to
the reference checkpoints -- but `forward` never touches a torch arithmetic op: it re-lays the weights out once
(cached, keyed on the parameter's storage/version) and enqueues vd_hip kernels.  Activations are fp16
[B, H, W, C] (== (B*H*W, C) row-major) throughout; there is no CPU or eager fallback.
"""
import os

import torch
import torch.nn as nn

from vd_hip import ops, pack

# development switch: VD_LN_FOLD=0 runs every LayerNorm as its own kernel (A/B runs of the fold)
LN_FOLD = os.environ.get("VD_LN_FOLD", "1") != "0"


def _ver(t):
    return None if t is None else (t.data_ptr(), t._version, t.dtype, str(t.device), tuple(t.shape))


class PackCache:
    """Mixin: cache of kernel-layout fp16 copies of parameters, rebuilt when the parameter storage changes
    (load_state_dict, .half(), .to(device))."""

    def _packed(self, key, tensors, fn):
        cache = self.__dict__.setdefault("_vd_pack_cache", {})
        ver = tuple(_ver(t) for t in tensors)
        hit = cache.get(key)
        if hit is None or hit[0] != ver:
            with torch.no_grad():
                val = fn()
            cache[key] = (ver, val)
            return val
        return hit[1]


def _h(t):
    """fp16 contiguous device copy/view of a parameter (compute dtype of the path is fp16)."""
    return None if t is None else t.detach().to(torch.float16).contiguous()


def fold_layernorm(w, b, ln):
    """Fold `ln` (nn.LayerNorm in front of a projection) into the projection: LN(x) W^T + b =
    rstd (x W'^T - mean colsum) + b'  with  W' = gamma (*) W,  colsum[n] = sum_k W'[n, k],  b' = beta W^T + b.
    Returns (W' fp16, b' fp16, colsum fp32); vd_gemm_f16 (VD_EPI_LNFOLD) derives mean / rstd from the A tiles as they
    pass through LDS, so the LayerNorm costs no pass over memory (reference attention.py:214-218 runs nn.LayerNorm as
    its own op in front of attn1 / attn2 / ff).  colsum is taken over the fp16-ROUNDED folded weights -- exactly what the
    MFMAs accumulate -- so the mean term cancels to fp32 round-off."""
    g, be = ln.weight.detach().float(), ln.bias.detach().float()
    wf = w.float()
    wp = (wf * g[None, :]).to(torch.float16).contiguous()
    colsum = wp.float().sum(1).contiguous()
    bp = wf @ be
    if b is not None:
        bp = bp + b.float()
    return wp, bp.to(torch.float16).contiguous(), colsum


class Conv2d(nn.Conv2d, PackCache):
    """3x3 / 1x1 convolution as implicit GEMM on channels-last input (vd_gemm_f16).

    forward(x, x1=None, ups=0, pad_hi=None, **epilogue): x1 is an optional second tensor whose channels are
    concatenated after x's (never materialised); ups=1 fuses a nearest 2x upsample in front; epilogue keywords
    (rowvec/rows_per_batch/res/act/alpha) are fused into the GEMM epilogue.  want_stats=True: the output feeds a GroupNorm --
    its producer emits the per-channel statistics (ops.ChanStats on the returned tensor, csrc/gn_fused.hip)."""

    def _w(self):
        def build():
            w = _h(self.weight)
            cin = w.shape[1]
            if cin % 64 != 0:  # small-Cin path goes through vd_im2col_small_f16
                return pack.pack_conv_weight_small(w), _h(self.bias)
            return pack.pack_conv_weight(w), _h(self.bias)
        return self._packed("w", (self.weight, self.bias), build)

    def _w_stream(self):
        """MFMA-fragment-ordered copy of the 3x3 weights for the weight-streaming kernel of the 8x8 level."""
        return self._packed("w_stream", (self.weight,), lambda: pack.pack_conv_weight_stream(_h(self.weight)))

    def _w_stream_1x1(self):
        """Fragment-ordered copy of a 1x1 conv's weights (folded skip convolution of the weight-streaming kernel)."""
        return self._packed("w_stream_1x1", (self.weight,),
                            lambda: pack.pack_linear_weight_stream(_h(self.weight).reshape(self.out_channels, self.in_channels)))

    def forward(self, x, x1=None, ups=0, pad_hi=None, in_layout="nhwc", in_scale=1.0, in_shift=0.0, bias=None, **epi):
        w, b = self._w()
        if bias is not None:  # caller-supplied (pre-combined) bias vector
            b = bias
        k, s, p = self.kernel_size[0], self.stride[0], self.padding[0]
        cin = self.in_channels
        if cin % 64 != 0:
            assert x1 is None and ups == 0
            a, (B, Ho, Wo) = ops.im2col_small(x, layout=in_layout, ksize=k, stride=s, pad=p, pad_hi=pad_hi,
                                              in_scale=in_scale, in_shift=in_shift)
            out = ops.gemm(a, w, bias=b, stat_img_rows=Ho * Wo, **epi)
            st = ops.stats_of(out)
            out = out.view(B, Ho, Wo, self.out_channels)
            if st is not None:
                out._vd_stats = st
            return out
        assert in_layout == "nhwc" and in_scale == 1.0 and in_shift == 0.0
        if k == 3 and s == 1 and p == 1 and pad_hi is None and cin % 64 == 0 and x.shape[-1] % 64 == 0 and self.out_channels % 32 == 0:
            small = (ops.WSTREAM and ups == 0 and x.shape[1] == 8 and x.shape[2] == 8 and x.shape[0] % 2 == 0
                     and self.out_channels % 256 == 0)   # 8x8 level: the layer is its weight stream
            if small:
                epi = dict(epi, w_stream=self._w_stream())
        return ops.conv2d_nhwc(x, w, b, ksize=k, stride=s, pad=p, ups=ups, x1=x1, pad_hi=pad_hi, **epi)


class Linear(nn.Linear, PackCache):
    def _w(self):
        return self._packed("w", (self.weight, self.bias), lambda: (_h(self.weight), _h(self.bias)))

    def _w_stream(self):
        """The weight in MFMA-fragment order for vd_gemm_wstream_f16 (long-K projections applied to few rows)."""
        return self._packed("w_stream", (self.weight,), lambda: pack.pack_linear_weight_stream(_h(self.weight)))

    def forward(self, x, **epi):
        w, b = self._w()
        # long-K, small-M: the layer is its weight stream (FeedForward's output projection at the 16x16 / 8x8 levels: 5120 ->
        # 1280 against 2048 / 512 rows) -- fragments straight into registers instead of both operands through LDS
        if (ops.GEMM_WSTREAM and self.in_features >= 2048 and self.in_features % 64 == 0 and self.out_features % 256 == 0
                and not epi.get("want_stats") and epi.get("colsum") is None and epi.get("rowvec") is None):
            m = x.numel() // x.shape[-1]
            if m % 128 == 0 and m <= 4096:
                epi = dict(epi, w_stream=self._w_stream())
        return ops.linear(x, w, b, **epi)


class GroupNorm(nn.GroupNorm, PackCache):
    """GroupNorm over channels-last input, optionally over cat([x, x1]) and fused with SiLU."""

    def _w(self):
        return self._packed("w", (self.weight, self.bias), lambda: (_h(self.weight), _h(self.bias)))

    def forward(self, x, x1=None, silu=False):
        g, b = self._w()
        return ops.groupnorm_silu(x, g, b, x1=x1, groups=self.num_groups, eps=self.eps, silu=silu)


class LayerNorm(nn.LayerNorm, PackCache):
    def _w(self):
        return self._packed("w", (self.weight, self.bias), lambda: (_h(self.weight), _h(self.bias)))

    def forward(self, x):
        g, b = self._w()
        return ops.layernorm(x, g, b, self.eps)


class SiLU(nn.Module):
    """Placeholder keeping nn.Sequential indices identical to the reference; the activation itself is always
    fused into the neighbouring kernel (GroupNorm+SiLU, GEMM epilogue)."""

    def forward(self, x):
        raise RuntimeError("SiLU is fused into the adjacent HIP kernel; this module is never called directly")


class Identity(nn.Identity):
    pass
