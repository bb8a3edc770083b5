"""Transformer blocks of the VD UNet on the HIP kernel library, tokens-major (B*HW, C) end to end.

Same module tree / state-dict keys as the reference (lib/model_zoo/attention.py:37-64,152-266 there), different
execution: no NCHW<->(b, hw, c) transposes (the activation already is (B*HW, C)), q/k/v of the self-attention
in ONE GEMM, scores never materialised (vd_attention_f16), residual adds / GEGLU gating / bias fused into GEMM
epilogues, context K/V projections reusable across DDIM steps through an explicit cache.
"""
import torch
import torch.nn as nn

from vd_hip import ops, pack

from . import hip_layers
from .hip_layers import Conv2d, GroupNorm, LayerNorm, Linear, PackCache, _h, fold_layernorm


class GEGLU(nn.Module, PackCache):
    """proj: dim_in -> 2*dim_out, out = value * gelu(gate)  (reference attention.py:37-45), gating fused in the
    GEMM epilogue; the weight rows are re-interleaved once so value/gate tiles meet in one workgroup."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = Linear(dim_in, dim_out * 2)

    def forward(self, x, ln=None, ln_sums=None):
        """ln: the nn.LayerNorm the reference applies in front (x is then the UN-normalised input; folded, see
        hip_layers.fold_layernorm).  ln_sums: (sum, sum of squares) per row of x from the launch that stored x."""
        if ln is None:
            wp, bp = self._packed("geglu", (self.proj.weight, self.proj.bias),
                                  lambda: pack.pack_geglu(_h(self.proj.weight), _h(self.proj.bias)))
            return ops.linear(x, wp, bp, act=ops.ACT_GEGLU)

        wp, bp, cs = self.folded(ln)
        return ops.linear(x, wp, bp, act=ops.ACT_GEGLU, colsum=cs, ln_eps=ln.eps, ln_sums=ln_sums)

    def folded(self, ln):
        """(gamma-folded GEGLU-packed weight, beta-folded packed bias, fp32 row sums of the packed weight), cached."""
        def build():
            w, b, _ = fold_layernorm(_h(self.proj.weight), _h(self.proj.bias), ln)
            wp, bp = pack.pack_geglu(w, b)
            return wp, bp, wp.float().sum(1).contiguous()
        return self._packed("geglu_ln", (self.proj.weight, self.proj.bias, ln.weight, ln.bias), build)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, glu=True, dropout=0.0):
        super().__init__()
        assert glu, "only the gated (GEGLU) feed-forward is on the VD path"
        inner = int(dim * mult)
        dim_out = dim if dim_out is None else dim_out
        self.net = nn.Sequential(GEGLU(dim, inner), nn.Dropout(dropout), Linear(inner, dim_out))

    def chain_operands(self, ln):
        """(w1 packed, b1 packed, w2, b2) of the one-launch kernels, or None when this feed-forward does not fit them."""
        proj, out = self.net[0].proj, self.net[2]
        C = proj.in_features
        if out.bias is None or proj.bias is None or out.out_features != C or out.in_features != 4 * C:
            return None
        wp, bp, _ = self.net[0].folded(ln)
        w2, b2 = out._w()
        return wp, bp, w2, b2

    def forward(self, x, res=None, ln=None, ln_sums=None):
        """ln: the LayerNorm in front (folded); ln_sums: row statistics of x from its producer (ops.gemm(row_sums=...)).  Where the library has the one-launch kernel for this width (C = 320: the
        64x64 level, whose [M, 4C] GEGLU intermediate is 84 MB) LayerNorm, both projections, the gating and the residual
        run in vd_ff_geglu_f16; elsewhere GEGLU GEMM -> output GEMM."""
        C = x.shape[-1]
        proj, out = self.net[0].proj, self.net[2]
        if (ln is not None and res is not None and out.bias is not None and proj.bias is not None and proj.in_features == C
                and out.out_features == C and out.in_features == 4 * C and ops.ff_geglu_supported(C)):
            wp, bp, _ = self.net[0].folded(ln)
            w2, b2 = out._w()
            return ops.ff_geglu(x, wp, bp, w2, b2, res, ln.eps)
        return out(self.net[0](x, ln=ln, ln_sums=ln_sums), res=res)


class CrossAttention(nn.Module, PackCache):
    """softmax(q k^T d^-1/2) v with q from x and k, v from the context (x itself when context is None)."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner = dim_head * heads
        self.is_self = context_dim is None
        context_dim = query_dim if context_dim is None else context_dim
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.inner = inner
        self.to_q = Linear(query_dim, inner, bias=False)
        self.to_k = Linear(context_dim, inner, bias=False)
        self.to_v = Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(Linear(inner, query_dim), nn.Dropout(dropout))

    def _w_qkv(self):
        return self._packed("qkv", (self.to_q.weight, self.to_k.weight, self.to_v.weight),
                            lambda: torch.cat([_h(self.to_q.weight), _h(self.to_k.weight), _h(self.to_v.weight)], 0))

    def _w_kv(self):
        return self._packed("kv", (self.to_k.weight, self.to_v.weight),
                            lambda: torch.cat([_h(self.to_k.weight), _h(self.to_v.weight)], 0))

    def project_context(self, context):
        """[B, L, Dc] -> fused [B, L, 2*inner] (k | v); step-invariant for a fixed context."""
        return ops.linear(context, self._w_kv())

    def _w_qkv_ln(self, ln):
        return self._packed("qkv_ln", (self.to_q.weight, self.to_k.weight, self.to_v.weight, ln.weight, ln.bias),
                            lambda: fold_layernorm(self._w_qkv(), None, ln))

    def _w_q_ln(self, ln):
        return self._packed("q_ln", (self.to_q.weight, ln.weight, ln.bias),
                            lambda: fold_layernorm(_h(self.to_q.weight), None, ln))

    def forward(self, x, context=None, res=None, kv=None, ln=None, qkv=None, ln_sums=None, out_sums=None, defer_out=False):
        """ln_sums: row statistics of x for `ln` (from x's producer); out_sums: zeroed fp32 [rows, 2] the output projection
        accumulates the row statistics of ITS output into (for the LayerNorm in front of the next block part).
        defer_out: return the attention output `a` WITHOUT the output projection (the caller folds to_out + res into its next
        launch: ops.ff_chain); only honoured on the one-launch cross-attention path, else the projected tensor comes back as usual.
        x [B, N, C] -> to_out(attn) (+ res fused).  ln: the LayerNorm in front of the block's q (and self k / v)
        projections, folded into them -- x is then the un-normalised input.  qkv: the fused self-attention projection when
        the caller has already computed it (SpatialTransformer's chained entry kernel)."""
        c = self.inner
        if context is None and kv is None:
            if qkv is not None:
                pass
            elif ln is None:
                qkv = ops.linear(x, self._w_qkv())
            else:
                w, b, cs = self._w_qkv_ln(ln)
                qkv = ops.linear(x, w, b, colsum=cs, ln_eps=ln.eps, ln_sums=ln_sums)
            a = ops.attention(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], self.heads, scale=self.scale)
        else:
            if kv is None:
                kv = self.project_context(context)
            if ln is not None and x.dim() == 3 and x.is_contiguous() and ops.xattn_supported(self.heads, c // self.heads):
                # LayerNorm + to_q + attention over the (short) context in ONE launch: q never exists in memory
                w, b, cs = self._w_q_ln(ln)
                a = ops.xattn(x, w, b, cs, ln.eps, kv[..., :c], kv[..., c:], self.heads, scale=self.scale)
                if defer_out:
                    a._vd_deferred_out = True
                    return a
                return self.to_out[0](a, res=res, row_sums=out_sums)
            if ln is None:
                q = self.to_q(x)
            else:
                w, b, cs = self._w_q_ln(ln)
                q = ops.linear(x, w, b, colsum=cs, ln_eps=ln.eps)
            a = ops.attention(q, kv[..., :c], kv[..., c:], self.heads, scale=self.scale)
        return self.to_out[0](a, res=res, row_sums=out_sums)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, gated_ff=True, checkpoint=True,
                 disable_self_attn=False):
        super().__init__()
        assert not disable_self_attn
        self.attn1 = CrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.norm1 = LayerNorm(dim)
        self.norm2 = LayerNorm(dim)
        self.norm3 = LayerNorm(dim)
        self.checkpoint = checkpoint  # kept for config compatibility; inference never re-computes

    def forward(self, x, context=None, kv=None, qkv=None, ln1_sums=None, post=None, sync=None):
        """post = (wp [C, C], bp, alpha, res [B, N, C], stat_img_rows): SpatialTransformer.proj_out (+ skip / context mixing) to be
        folded into the block's last launch.  Returns (tensor, True) when it was -- the tensor is then proj_out's output -- else
        the block output (the caller runs proj_out).  sync: called right before a launch that reads post's `res` (see
        SpatialTransformer.forward)."""
        if hip_layers.LN_FOLD:  # the three LayerNorms ride in the q/k/v, q and GEGLU projections (VD_EPI_LNFOLD)
            x = self.attn1(x, res=x, ln=self.norm1, qkv=qkv, ln_sums=ln1_sums)
            C = x.shape[-1]
            fops = self.ff.chain_operands(self.norm3) if (x.is_contiguous() and ops.ff_geglu_supported(C)) else None
            if fops is not None and (ops.ff_chain_supported(C, "pre") or (post is not None and ops.ff_chain_supported(C, "post"))):
                # 64x64 level: attn2.to_out + x -> norm3 -> feed-forward -> + x [-> proj_out -> + x_in] in ONE launch
                w1p, b1p, w2, b2 = fops
                kw = {}
                xin = x
                if ops.ff_chain_supported(C, "pre"):
                    a = self.attn2(x, context=context, res=x, kv=kv, ln=self.norm2, defer_out=True)
                    if getattr(a, "_vd_deferred_out", False):
                        wo, bo = self.attn2.to_out[0]._w()
                        kw.update(a=a, wo=wo, bo=bo)
                    else:
                        xin = a   # the projected tensor came back (the unfused cross-attention path)
                else:
                    xin = self.attn2(x, context=context, res=x, kv=kv, ln=self.norm2)
                folded = post is not None and ops.ff_chain_supported(C, "post")
                if folded:
                    wp, bp, alpha, pres, hw = post
                    kw.update(wp=wp, bp=bp, alpha=alpha, res=pres, want_stats=True, stat_img_rows=hw)
                if kw:
                    if folded and sync is not None:
                        sync()
                    out = ops.ff_chain(xin, w1p, b1p, w2, b2, self.norm3.eps, **kw)
                    return (out, True) if folded else out
                return self.ff(xin, res=xin, ln=self.norm3)
            # norm3's statistics: accumulated by attn2's output projection where the feed-forward runs as separate GEMMs (the
            # one-launch feed-forward of the 64x64 level normalises in registers)
            C = x.shape[-1]
            s3 = None if ops.ff_geglu_supported(C) else ops.rowsum_take(x.numel() // C, x.device)
            x = self.attn2(x, context=context, res=x, kv=kv, ln=self.norm2, out_sums=s3)
            return self.ff(x, res=x, ln=self.norm3, ln_sums=getattr(x, "_vd_rowsums", None))
        x = self.attn1(self.norm1(x), res=x)
        x = self.attn2(self.norm2(x), context=context, res=x, kv=kv)
        x = self.ff(self.norm3(x), res=x)
        return x


class SpatialTransformer(nn.Module):
    """GroupNorm(eps 1e-6) -> 1x1 proj_in -> transformer block -> 1x1 proj_out -> + x_in, on [B, H, W, C]."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None, disable_self_attn=False):
        super().__init__()
        assert depth == 1
        self.in_channels = in_channels
        inner = n_heads * d_head
        self.norm = GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.proj_in = Conv2d(in_channels, inner, kernel_size=1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=context_dim,
                                   disable_self_attn=disable_self_attn) for _ in range(depth)])
        self.proj_out = Conv2d(inner, in_channels, kernel_size=1, stride=1, padding=0)
        with torch.no_grad():  # zero_module(proj_out), reference attention.py:249-253
            self.proj_out.weight.zero_()
            self.proj_out.bias.zero_()

    def project_context(self, context):
        return self.transformer_blocks[0].attn2.project_context(context)

    def forward(self, x, context=None, kv=None, alpha=1.0, res=None, sync=None):
        """Returns alpha * (proj_out(...) + bias) + res, res defaulting to x (the block's own skip).
        alpha/res implement VD's context mixing sum_i r_i * ST_i(x) without extra passes.
        sync: callable invoked right before the ONE launch that reads `res` (the last of the block): the caller runs the blocks
        of several context types on forked streams and `res` is the previous type's output (vd.run_unet)."""
        B, H, W, C = x.shape
        blk = self.transformer_blocks[0]
        inner = self.proj_in.out_channels
        if hip_layers.LN_FOLD and x.is_contiguous() and ops.st_chain_supported(B, H * W, C, inner):
            # GroupNorm (as a per-sample affine map) -> proj_in -> h and LayerNorm(h) -> q | k | v in one launch
            g, b = self.norm._w()
            # (centred fp16 map: (x - fp16(mean)) * scale + shift', no |mean| / sigma amplification of the fp16 rounding)
            sc, sh, ct = ops.groupnorm_affine(x, g, b, groups=self.norm.num_groups, eps=self.norm.eps, centered=True)
            w1, b1 = self.proj_in._w()
            w2, b2, _ = blk.attn1._w_qkv_ln(blk.norm1)
            h, qkv = ops.row320_chain(x.view(B, H * W, C), sc, sh, H * W, w1, b1, w2, b2, blk.norm1.eps, center=ct)
            pres = x if res is None else res
            post = None
            if pres.is_contiguous() and self.proj_out.out_channels == inner and self.proj_out.bias is not None:
                wp, bp = self.proj_out._w()
                post = (wp, bp, float(alpha), pres.view(B, H * W, C), H * W)
            h = blk(h, context=context, kv=kv, qkv=qkv, post=post, sync=sync)
            if isinstance(h, tuple):   # proj_out (+ skip, alpha, statistics) ran inside the block's last launch
                out = h[0]
                st = ops.stats_of(out)
                out = out.view(B, H, W, C)
                if st is not None:
                    out._vd_stats = st
                return out
            if sync is not None:
                sync()
            return self.proj_out(h.view(B, H, W, -1), alpha=alpha, res=pres, want_stats=True)
        # norm1's row statistics ride on proj_in's epilogue (ops.gemm(row_sums=...)) instead of a vd_row_stats_f16 launch
        s1 = ops.rowsum_take(B * H * W, x.device) if hip_layers.LN_FOLD else None
        h = self.proj_in(self.norm(x, silu=False), row_sums=s1)
        h = self.transformer_blocks[0](h.view(B, H * W, -1), context=context, kv=kv, ln1_sums=getattr(h, "_vd_rowsums", None))
        if sync is not None:
            sync()
        return self.proj_out(h.view(B, H, W, -1), alpha=alpha, res=x if res is None else res, want_stats=True)
