"""Per-kernel numerics: every HIP kernel against a plain PyTorch fp32 reference of the same op.

All calls go through the C ABI (ctypes -> libvd_hip.so).  Tolerances are stated per test; inputs are
fp16-rounded so the only differences are accumulation order / fp16 output rounding.
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = a.float()
    b = b.float()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def rnd(shape, dev, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).half().to(dev)


@pytest.fixture(scope="module")
def ops():
    from vd_hip import ops as o
    return o


def test_mfma_layout_probe(ops, dev):
    """The kernels assume: C/D row=(r&3)+8*(r>>2)+4*(lane>>5), col=lane&31; A/B k-slots pair 1:1."""
    a_k, c_row, c_col = ops.probe_mfma_layout(dev)
    lanes = torch.arange(64).view(64, 1)
    regs = torch.arange(16).view(1, 16)
    exp_row = (regs & 3) + 8 * (regs >> 2) + 4 * (lanes >> 5)
    exp_col = (lanes & 31).expand(64, 16)
    assert torch.equal(c_row.long(), exp_row.long()), "C/D row map differs:\n%s" % c_row
    assert torch.equal(c_col.long(), exp_col.long()), "C/D col map differs:\n%s" % c_col
    assert a_k.tolist() == list(range(16)), "A/B k-slot pairing is not the identity: %s" % a_k.tolist()


@pytest.mark.parametrize("B,S,c0,c1,silu", [(2, 4, 320, 0, True), (3, 4, 1280, 1280, True), (1, 4, 64, 64, False), (8, 4, 640, 320, True)])
def test_groupnorm0d(ops, dev, B, S, c0, c1, silu):
    """FCBlock's GroupNorm: statistics per (sample, channel group) over all S positions, affine per (s, c)."""
    x0 = rnd((B, S, c0), dev, 2.0, 40) + 0.3
    x1 = rnd((B, S, c1), dev, 1.0, 41) if c1 else None
    C = c0 + c1
    gamma = rnd((S, C), dev, 0.5, 42) + 1.0
    beta = rnd((S, C), dev, 0.5, 43)
    x = torch.cat([x0, x1], -1) if c1 else x0
    # reference formulation: flatten [C, S] (c-major) to C*S channels of a 1x1 map, 32 groups, affine per flat channel
    xf = x.float().permute(0, 2, 1).reshape(B, C * S, 1, 1)
    ref = F.group_norm(xf, 32, gamma.float().t().reshape(-1), beta.float().t().reshape(-1), 1e-5)
    ref = ref.view(B, C, S).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    out = ops.groupnorm0d_silu(x0, gamma, beta, x1=x1, groups=32, eps=1e-5, silu=silu)
    assert out.shape == (B, S, C)
    assert rel_l2(out, ref) < 2e-3


def test_lds_transpose_read_probe(ops, dev):
    """ds_read_b64_tr_b16 semantics the attention kernel's V operand is built on: in every 16-lane group, lane i supplies
    the address of 4 contiguous halfs = row (i>>2), column quad (i&3) of a [4][16] block (row stride free), and receives
    column i of that block (4 rows)."""
    stride = 64  # bytes between block rows (the V panels use 64-byte rows)
    lanes = torch.arange(64)
    g, i = lanes >> 4, lanes & 15
    addr = (g * 1024 + (i >> 2) * stride + (i & 3) * 8).to(torch.int32).to(dev)
    got = ops.probe_lds_tr16(addr).long()
    j = torch.arange(4).view(1, 4)
    exp = (g.view(64, 1) * 1024 + j * stride) // 2 + i.view(64, 1)   # element index of block[j][i]
    assert torch.equal(got, exp), "transpose-read gather differs:\n%s\nexpected\n%s" % (got[:20], exp[:20])


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 320, 320), (1000, 192, 128), (77, 768, 768), (4096, 64, 2560),
                                   (33, 8, 64), (512, 1280, 1280), (130, 4, 320), (100, 72, 16), (256, 128, 200)])
def test_gemm_plain(ops, dev, M, N, K):
    a = rnd((M, K), dev, 1.0, 1)
    w = rnd((N, K), dev, 0.05, 2)
    ref = a.float() @ w.float().t()
    out = ops.gemm(a, w)
    assert out.shape == (M, N)
    assert rel_l2(out, ref) < 2e-3


@pytest.mark.parametrize("M,N,K", [(300, 100, 64), (300, 102, 64), (64, 4, 128), (1000, 36, 320), (130, 90, 1280)])
def test_gemm_epilogue_unaligned_widths(ops, dev, M, N, K):
    """Output widths that are not multiples of 8 (and of 4): the epilogue's requests fall back from the 16-byte segment /
    8-byte bias-group forms to the element-wise tail -- bias, per-sample row vector and residual must still all arrive."""
    a = rnd((M, K), dev, 1.0, 81)
    w = rnd((N, K), dev, 0.05, 82)
    bias = rnd((N,), dev, 0.5, 83)
    res = rnd((M, N), dev, 1.0, 84)
    ref = a.float() @ w.float().t() + bias.float()
    assert rel_l2(ops.gemm(a, w, bias=bias), ref) < 2e-3
    assert rel_l2(ops.gemm(a, w, bias=bias, res=res), ref + res.float()) < 2e-3
    rpb = M // 2 if M % 2 == 0 else M
    rv = rnd((M // rpb, N), dev, 0.5, 85)
    out = ops.gemm(a, w, bias=bias, rowvec=rv, rows_per_batch=rpb, res=res)
    assert rel_l2(out, ref + rv.float().repeat_interleave(rpb, 0) + res.float()) < 2e-3
    assert rel_l2(ops.gemm(a, w, bias=bias, act=ops.ACT_SILU, res=res), F.silu(ref) + res.float()) < 2e-3


def test_gemm_asymmetric_identity(ops, dev):
    """A = I with an asymmetric W catches row/col swaps in the C write (guide rule: transpose-detecting)."""
    K = 128
    a = torch.eye(K, dtype=torch.float16, device=dev)
    w = (torch.arange(192 * K, device=dev).reshape(192, K) % 97).half() / 16
    out = ops.gemm(a, w)
    assert torch.equal(out, w.t().contiguous())


def _act_ref(act, v):
    if act == "quick_gelu":
        return v * torch.sigmoid(1.702 * v)
    if act == "silu":
        return F.silu(v)
    if act == "gelu_tanh":
        return F.gelu(v, approximate="tanh")
    return v


@pytest.mark.parametrize("act", ["none", "quick_gelu", "silu", "gelu_tanh"])
def test_gemm_epilogue(ops, dev, act):
    M, N, K = 512, 320, 192
    a = rnd((M, K), dev, 1.0, 3)
    w = rnd((N, K), dev, 0.05, 4)
    bias = rnd((N,), dev, 0.5, 5)
    res = rnd((M, N), dev, 1.0, 7)
    v = a.float() @ w.float().t() + bias.float()
    ref = _act_ref(act, v) * 0.7 + res.float()
    code = {"none": ops.ACT_NONE, "quick_gelu": ops.ACT_QUICK_GELU, "silu": ops.ACT_SILU, "gelu_tanh": ops.ACT_GELU_TANH}[act]
    out = ops.gemm(a, w, bias=bias, res=res, act=code, alpha=0.7)
    assert rel_l2(out, ref) < 2e-3
    # the same through the split-K reduce kernel (its own epilogue: epi8_request / epi8_finish)
    out = ops.gemm(a, w, bias=bias, res=res, act=code, alpha=0.7, split_k=3)
    assert rel_l2(out, ref) < 2e-3


@pytest.mark.parametrize("act", ["quick_gelu", "silu", "gelu_tanh"])
def test_fused_activation_extremes(ops, dev, act):
    """The fused activations are one body x * sigmoid(x (c1 + c3 x^2)) (gemm_kernel.h: apply_act): pre-activations from -80 to 80
    (exp overflows to inf on one side, underflows on the other) against torch, through the GEMM and the 3x3 conv epilogues."""
    from vd_hip.pack import pack_conv_weight
    code = {"quick_gelu": ops.ACT_QUICK_GELU, "silu": ops.ACT_SILU, "gelu_tanh": ops.ACT_GELU_TANH}[act]
    M, N, K = 256, 128, 64
    a = rnd((M, K), dev, 1.0, 31)
    w = rnd((N, K), dev, 0.02, 32)
    bias = torch.linspace(-80.0, 80.0, N, device=dev).to(torch.float16)
    v = a.float() @ w.float().t() + bias.float()
    out = ops.gemm(a, w, bias=bias, act=code)
    assert torch.isfinite(out.float()).all()
    assert (out.float() - _act_ref(act, v)).abs().max() < 0.06 and rel_l2(out, _act_ref(act, v)) < 2e-3
    x = rnd((2, 16, 16, 64), dev, 1.0, 33)
    wt = rnd((N, 64, 3, 3), dev, 0.01, 34)
    ref = _act_ref(act, _conv_ref(x, wt, bias, 1, 1, 0))
    out = ops.conv2d_nhwc(x, pack_conv_weight(wt), bias, ksize=3, pad=1, act=code)
    assert torch.isfinite(out.float()).all() and rel_l2(out, ref) < 2e-3


@pytest.mark.parametrize("M,N,rpb", [(512, 320, 128), (384, 1280, 64), (200, 72, 50)])
def test_gemm_rowvec_residual(ops, dev, M, N, rpb):
    """ResBlock epilogues: + bias + emb[batch] broadcast over the rows of a sample, + residual."""
    K = 192
    a = rnd((M, K), dev, 1.0, 3)
    w = rnd((N, K), dev, 0.05, 4)
    bias = rnd((N,), dev, 0.5, 5)
    rowvec = rnd((M // rpb, N), dev, 0.5, 6)
    res = rnd((M, N), dev, 1.0, 7)
    ref = a.float() @ w.float().t() + bias.float() + rowvec.float().repeat_interleave(rpb, 0) + res.float()
    out = ops.gemm(a, w, bias=bias, rowvec=rowvec, rows_per_batch=rpb, res=res)
    assert rel_l2(out, ref) < 2e-3
    from vd_hip import VdHipError
    with pytest.raises(VdHipError, match="rowvec"):
        ops.gemm(a, w, bias=bias, rowvec=rowvec, rows_per_batch=rpb, alpha=0.5)


@pytest.mark.parametrize("M,C", [(384, 320), (4096, 320), (8192 + 72, 64)])
def test_gemm_geglu(ops, dev, M, C):
    from vd_hip.pack import pack_geglu
    x = rnd((M, C), dev, 1.0, 8)
    w = rnd((8 * C, C), dev, 0.05, 9)
    b = rnd((8 * C,), dev, 0.2, 10)
    h = x.float() @ w.float().t() + b.float()
    val, gate = h.chunk(2, dim=-1)
    ref = val * F.gelu(gate)
    wp, bp = pack_geglu(w, b)
    out = ops.gemm(x, wp, bias=bp, act=ops.ACT_GEGLU)
    assert out.shape == (M, 4 * C)
    assert rel_l2(out, ref) < 2e-3


@pytest.mark.parametrize("M,K,N,offset", [(4096, 320, 960, 0.0), (1000, 640, 640, 0.5), (256, 1280, 3840, -2.0), (130, 320, 320, 8.0),
                                          (2048, 328, 192, 0.3)])
@pytest.mark.parametrize("inloop", [True, False])
def test_gemm_layernorm_fold(ops, dev, M, K, N, offset, inloop, monkeypatch):
    """VD_EPI_LNFOLD: LayerNorm(x) @ W^T + b with the LayerNorm folded into the projection vs nn.LayerNorm followed by the
    matmul in fp32; row statistics from the A fragments inside the K loop (ln_stats NULL, the product path) and from
    vd_row_stats_f16.  `offset` shifts the row means away from zero (the fold subtracts mean * colsum from the accumulator
    and the in-loop variance is E[x^2] - mean^2 in fp32: the cancellation must stay harmless)."""
    from lib.model_zoo.hip_layers import fold_layernorm
    monkeypatch.setattr(ops, "LN_INLOOP", inloop)
    x = rnd((M, K), dev, 1.5, 50) + offset
    w = rnd((N, K), dev, 0.05, 51)
    b = rnd((N,), dev, 0.3, 52)
    ln = torch.nn.LayerNorm(K, eps=1e-5).to(dev)
    with torch.no_grad():
        ln.weight.copy_(1.0 + 0.3 * torch.randn(K, generator=torch.Generator().manual_seed(53)).to(dev))
        ln.bias.copy_(0.2 * torch.randn(K, generator=torch.Generator().manual_seed(54)).to(dev))
    res = rnd((M, N), dev, 1.0, 55)
    ref = F.layer_norm(x.float(), (K,), ln.weight.float(), ln.bias.float(), 1e-5) @ w.float().t() + b.float() + res.float()
    wp, bp, cs = fold_layernorm(w, b, ln)
    out = ops.gemm(x, wp, bias=bp, res=res, colsum=cs, ln_eps=1e-5)
    assert rel_l2(out, ref) < 3e-3
    # without a bias on the projection (to_q / to_k / to_v): bias' = beta W^T alone
    wp, bp, cs = fold_layernorm(w, None, ln)
    out = ops.gemm(x, wp, bias=bp, colsum=cs, ln_eps=1e-5)
    assert rel_l2(out, ref - b.float() - res.float()) < 3e-3


@pytest.mark.parametrize("M,pre,post,offset", [(128, True, True, 0.0), (1000, True, False, 0.5), (4096, False, True, -1.0), (4096 + 40, True, True, 0.0),
                                               (8192, True, True, 2.0)])
def test_ff_chain(ops, dev, M, pre, post, offset, monkeypatch):
    """vd_ff_chain_f16 (round 5): attn2.to_out + residual -> LayerNorm -> GEGLU feed-forward -> + residual -> proj_out (alpha, +
    residual, per-channel statistics) in one launch (C = 320) against torch fp32 and against the library's own separate launches
    (vd_gemm_f16 -> vd_ff_geglu_f16 -> vd_gemm_f16); either projection alone as well."""
    from lib.model_zoo.hip_layers import fold_layernorm
    from vd_hip.pack import pack_geglu
    monkeypatch.setattr(ops, "GN_SUMS", True)   # (opt-in in the product: VD_GN_SUMS=1)
    C = 320
    assert ops.ff_chain_supported(C)
    x0 = rnd((M, C), dev, 1.2, 900) + offset
    a = rnd((M, C), dev, 1.0, 901)
    wo, bo = rnd((C, C), dev, 0.05, 902), rnd((C,), dev, 0.2, 903)
    w1, b1 = rnd((8 * C, C), dev, 0.05, 904), rnd((8 * C,), dev, 0.2, 905)
    w2, b2 = rnd((C, 4 * C), dev, 0.03, 906), rnd((C,), dev, 0.2, 907)
    wp, bp = rnd((C, C), dev, 0.05, 908), rnd((C,), dev, 0.2, 909)
    r = rnd((M, C), dev, 1.0, 910)
    alpha = 0.4
    ln = torch.nn.LayerNorm(C, eps=1e-5).to(dev)
    with torch.no_grad():
        ln.weight.copy_(1.0 + 0.2 * torch.randn(C, generator=torch.Generator().manual_seed(911)).to(dev))
        ln.bias.copy_(0.1 * torch.randn(C, generator=torch.Generator().manual_seed(912)).to(dev))
    x1 = (a.float() @ wo.float().t() + bo.float() + x0.float()) if pre else x0.float()
    x1h = x1.half().float()   # the kernel parks x1 in fp16
    xn = F.layer_norm(x1h, (C,), ln.weight.float(), ln.bias.float(), 1e-5)
    v, g = (xn @ w1.float().t() + b1.float()).chunk(2, dim=-1)
    y = x1h + (v * F.gelu(g)) @ w2.float().t() + b2.float()
    ref = alpha * (y.half().float() @ wp.float().t() + bp.float()) + r.float() if post else y
    wf, bf, _ = fold_layernorm(w1, b1, ln)
    w1p, b1p = pack_geglu(wf, bf)
    kw = {}
    if pre:
        kw.update(a=a, wo=wo, bo=bo)
    if post:
        kw.update(wp=wp, bp=bp, alpha=alpha, res=r, want_stats=True, stat_img_rows=M if M % 128 == 0 else 0)
    out = ops.ff_chain(x0, w1p, b1p, w2, b2, 1e-5, **kw)
    assert out.shape == ref.shape and rel_l2(out, ref) < 3e-3
    # the separate launches of the library
    xs = ops.gemm(a, wo, bias=bo, res=x0) if pre else x0
    ys = ops.ff_geglu(xs, w1p, b1p, w2, b2, xs, 1e-5)
    os_ = ops.gemm(ys, wp, bias=bp, alpha=alpha, res=r) if post else ys
    assert rel_l2(out, os_) < 3e-3
    st = ops.stats_of(out)
    if post and M % 128 == 0:
        assert st is not None and st.T == M // 128 and st.HW == M
        _stats_close(st, _chan_stats_ref(out.view(1, M, C), 1, M // 128), 128)
        assert (st.sums is not None) == (2 * M * C > ops.GN_FUSED_MAX)   # one pair per (sample, channel) for the table-free apply
        if st.sums is not None:
            o64 = out.double()
            assert torch.allclose(st.sums.view(C, 2)[:, 0].double() / 2.0 ** 32, o64.sum(0), rtol=1e-5, atol=1e-2)
            assert torch.allclose(st.sums.view(C, 2)[:, 1].double() / 2.0 ** 16, (o64 * o64).sum(0), rtol=1e-5, atol=1e-2)
    else:
        assert st is None


@pytest.mark.parametrize("M,N,K,res,conv", [(8192, 640, 640, True, False), (2048, 1280, 1280, True, False), (512, 1280, 1280, False, True),
                                           (8192 + 24, 640, 640, False, False), (2048, 1280, 320, True, False)])
def test_gemm_row_sums_feed_the_layernorm_fold(ops, dev, M, N, K, res, conv):
    """VdGemmDesc.row_sums (ABI 6): the epilogue of the launch that STORES x accumulates (sum, sum of squares) of every row of x;
    the LayerNorm-folded consumer reads them with VD_EPI_LN_SUMS instead of running vd_row_stats_f16.  Checked: the sums
    against torch on the stored fp16 values, and LayerNorm(x) @ W2^T through both statistics sources against torch fp32."""
    from lib.model_zoo.hip_layers import fold_layernorm
    a = rnd((M, K), dev, 1.0, 300)
    w1 = rnd((N, K), dev, 0.04, 301)
    b1 = rnd((N,), dev, 0.3, 302)
    r = rnd((M, N), dev, 1.0, 303) + 0.7 if res else None
    sums = ops.rowsum_take(M, a.device)
    if conv and M % 64 == 0:   # proj_in: a 1x1 convolution on an 8x8 grid
        x4 = ops.conv2d_nhwc(a.view(M // 64, 8, 8, K), w1, b1, ksize=1, stride=1, pad=0, row_sums=sums)
        got = getattr(x4, "_vd_rowsums", None)
        x = x4.view(M, N)
    else:
        x = ops.gemm(a, w1, bias=b1, res=r, row_sums=sums)
        got = getattr(x, "_vd_rowsums", None)
    assert got is not None, "the planned launch did not take row_sums"
    xf = x.float()
    ref_s = torch.stack([xf.sum(1), (xf * xf).sum(1)], 1)
    assert got.dtype == torch.int64   # fixed point (sum x 2^24, sum of squares x 2^16): integer adds commute, runs are bit-identical
    assert rel_l2(got[:, 0].double() / 2 ** 24, ref_s[:, 0]) < 1e-4 and rel_l2(got[:, 1].double() / 2 ** 16, ref_s[:, 1]) < 1e-5
    N2 = 1920
    w2 = rnd((N2, N), dev, 0.05, 304)
    ln = torch.nn.LayerNorm(N, eps=1e-5).to(dev)
    with torch.no_grad():
        ln.weight.copy_(1.0 + 0.3 * torch.randn(N, generator=torch.Generator().manual_seed(305)).to(dev))
        ln.bias.copy_(0.2 * torch.randn(N, generator=torch.Generator().manual_seed(306)).to(dev))
    ref = F.layer_norm(xf, (N,), ln.weight.float(), ln.bias.float(), 1e-5) @ w2.float().t()
    wp, bp, cs = fold_layernorm(w2, None, ln)
    y_sums = ops.gemm(x, wp, bias=bp, colsum=cs, ln_eps=1e-5, ln_sums=got)
    y_stat = ops.gemm(x, wp, bias=bp, colsum=cs, ln_eps=1e-5)
    assert rel_l2(y_sums, ref) < 3e-3 and rel_l2(y_stat, ref) < 3e-3
    assert rel_l2(y_sums, y_stat) < 1e-3


@pytest.mark.parametrize("M,C", [(1024, 320), (4096 + 40, 320)])
def test_gemm_layernorm_fold_geglu(ops, dev, M, C):
    from lib.model_zoo.hip_layers import fold_layernorm
    from vd_hip.pack import pack_geglu
    x = rnd((M, C), dev, 1.0, 56) + 0.4
    w = rnd((8 * C, C), dev, 0.05, 57)
    b = rnd((8 * C,), dev, 0.2, 58)
    ln = torch.nn.LayerNorm(C, eps=1e-5).to(dev)
    with torch.no_grad():
        ln.weight.copy_(1.0 + 0.3 * torch.randn(C, generator=torch.Generator().manual_seed(59)).to(dev))
        ln.bias.copy_(0.2 * torch.randn(C, generator=torch.Generator().manual_seed(60)).to(dev))
    h = F.layer_norm(x.float(), (C,), ln.weight.float(), ln.bias.float(), 1e-5) @ w.float().t() + b.float()
    val, gate = h.chunk(2, dim=-1)
    ref = val * F.gelu(gate)
    wf, bf, _ = fold_layernorm(w, b, ln)
    wp, bp = pack_geglu(wf, bf)
    out = ops.gemm(x, wp, bias=bp, act=ops.ACT_GEGLU, colsum=wp.float().sum(1).contiguous(), ln_eps=1e-5)
    assert rel_l2(out, ref) < 3e-3


@pytest.mark.parametrize("M,offset", [(128, 0.0), (1000, 0.5), (4096 + 40, -2.0), (32, 8.0)])
def test_ff_geglu_fused(ops, dev, M, offset):
    """vd_ff_geglu_f16 (LayerNorm -> GEGLU projection -> gating -> output projection -> + residual in one launch, C = 320)
    against torch fp32, and against the library's own three-launch chain; rows with a large common offset exercise the
    in-register LayerNorm (mean >> spread)."""
    from lib.model_zoo.hip_layers import fold_layernorm
    from vd_hip.pack import pack_geglu
    C = 320
    assert ops.ff_geglu_supported(C)
    x = rnd((M, C), dev, 1.2, 80) + offset
    w1, b1 = rnd((8 * C, C), dev, 0.05, 81), rnd((8 * C,), dev, 0.2, 82)
    w2, b2 = rnd((C, 4 * C), dev, 0.03, 83), rnd((C,), dev, 0.2, 84)
    ln = torch.nn.LayerNorm(C, eps=1e-5).to(dev)
    with torch.no_grad():
        ln.weight.copy_(1.0 + 0.2 * torch.randn(C, device=dev))
        ln.bias.copy_(0.1 * torch.randn(C, device=dev))
    xn = F.layer_norm(x.float(), (C,), ln.weight.float(), ln.bias.float(), 1e-5)
    v, g = (xn @ w1.float().t() + b1.float()).chunk(2, dim=-1)
    ref = x.float() + (v * F.gelu(g)) @ w2.float().t() + b2.float()
    wf, bf, cs = fold_layernorm(w1, b1, ln)
    wp, bp = pack_geglu(wf, bf)
    out = ops.ff_geglu(x, wp, bp, w2, b2, x, 1e-5)
    assert out.shape == x.shape and rel_l2(out, ref) < 3e-3
    h = ops.linear(x, wp, bp, act=ops.ACT_GEGLU, colsum=wp.float().sum(1).contiguous(), ln_eps=1e-5)
    chain = ops.linear(h, w2, b2, res=x)
    assert rel_l2(out, chain.float()) < 3e-3


@pytest.mark.parametrize("M,N,ln,with_res,offset", [(32768, 320, False, True, 0.0), (32768, 960, True, False, 0.4), (24576 + 40, 320, True, True, -3.0),
                                                     (32768, 320, False, False, 0.0), (8192 + 5, 960, False, True, 0.0),
                                                     (24576 + 77, 960, False, False, 0.0), (4096, 1920, True, False, 1.0)])
def test_gemm_row320(ops, dev, M, N, ln, with_res, offset, monkeypatch):
    """vd_gemm_row320_f16 (rows of x resident in registers, K = 320) against torch fp32 and against gemm_f16_kernel on the same
    operands; through ops.gemm's dispatch (plain, LayerNorm-folded, with residual, ragged last row block)."""
    from lib.model_zoo.hip_layers import fold_layernorm
    from vd_hip.loader import lib
    x = rnd((M, 320), dev, 1.1, 900) + offset
    w = rnd((N, 320), dev, 0.05, 901)
    b = rnd((N,), dev, 0.3, 902)
    res = rnd((M, N), dev, 1.0, 903) if with_res else None
    lnm = torch.nn.LayerNorm(320, eps=1e-5).to(dev)
    with torch.no_grad():
        lnm.weight.copy_(1.0 + 0.3 * torch.randn(320, generator=torch.Generator().manual_seed(904)).to(dev))
        lnm.bias.copy_(0.2 * torch.randn(320, generator=torch.Generator().manual_seed(905)).to(dev))
    xin = F.layer_norm(x.float(), (320,), lnm.weight.float(), lnm.bias.float(), 1e-5) if ln else x.float()
    ref = xin @ w.float().t() + b.float() + (res.float() if with_res else 0.0)
    assert lib().vd_gemm_row320_supported(M, N, 320) == 1 and lib().vd_gemm_row320_supported(M, 480, 320) == 0
    if ln:
        wp, bp, cs = fold_layernorm(w, b, lnm)
        kw = dict(colsum=cs, ln_eps=1e-5)
    else:
        wp, bp, kw = w, b, {}
    out = ops.gemm_row320(x, wp, bp, res, ln, 1e-5)
    assert out.shape == (M, N) and rel_l2(out, ref) < 3e-3
    # ops.gemm sends the LayerNorm-folded projections whose row blocks fill the chip here, everything else to gemm_f16_kernel
    ops.profile_begin()
    via = ops.gemm(x, wp, bias=bp, res=res, **kw)
    names = [r[0] for r in ops.profile_end()]
    assert any(n.startswith("rowgemm320") for n in names) == (ln and ((M + 127) // 128) * (N // 320) >= 192), names
    assert rel_l2(via, ref) < 3e-3
    monkeypatch.setattr(ops, "ROW320", False)
    old = ops.gemm(x, wp, bias=bp, res=res, **kw)
    assert rel_l2(out, old.float()) < 3e-3


@pytest.mark.parametrize("B,HW,offset", [(8, 4096, 0.0), (7, 4096, 0.7), (2, 9216, -2.0), (25, 1024, 0.0)])
def test_row320_chain(ops, dev, B, HW, offset):
    """vd_groupnorm_affine_f16 + vd_gemm_row320_chain_f16 (GroupNorm as an affine map -> proj_in -> h; LayerNorm(h) -> q|k|v in
    one launch) against torch fp32, and h / q|k|v against the library's separate launches."""
    from lib.model_zoo.hip_layers import fold_layernorm
    C = 320
    x = rnd((B, HW, C), dev, 1.3, 950) + offset + 0.5 * rnd((1, 1, C), dev, 1.0, 951)
    gam, bet = 1.0 + 0.2 * rnd((C,), dev, 1.0, 952), 0.1 * rnd((C,), dev, 1.0, 953)
    w1, b1 = rnd((C, C), dev, C ** -0.5, 954), rnd((C,), dev, 0.2, 955)
    w2 = rnd((3 * C, C), dev, C ** -0.5, 956)
    ln = torch.nn.LayerNorm(C, eps=1e-5).to(dev)
    with torch.no_grad():
        ln.weight.copy_(1.0 + 0.2 * torch.randn(C, generator=torch.Generator().manual_seed(957)).to(dev))
        ln.bias.copy_(0.1 * torch.randn(C, generator=torch.Generator().manual_seed(958)).to(dev))
    xn = F.group_norm(x.float().transpose(1, 2), 32, gam.float(), bet.float(), 1e-6).transpose(1, 2)
    h_ref = xn @ w1.float().t() + b1.float()
    y_ref = F.layer_norm(h_ref, (C,), ln.weight.float(), ln.bias.float(), 1e-5) @ w2.float().t()
    sc, sh = ops.groupnorm_affine(x, gam, bet, groups=32, eps=1e-6)
    mean = x.float().view(B, HW, 32, C // 32).mean((1, 3))
    var = x.float().view(B, HW, 32, C // 32).var((1, 3), unbiased=False)
    sc_ref = (var + 1e-6).rsqrt().repeat_interleave(C // 32, 1) * gam.float()
    assert rel_l2(sc, sc_ref) < 2e-3 and rel_l2(sh, bet.float() - mean.repeat_interleave(C // 32, 1) * sc_ref) < 3e-3
    w2p, b2p, cs = fold_layernorm(w2, None, ln)
    h, y = ops.row320_chain(x, sc, sh, HW, w1, b1, w2p, b2p, 1e-5)
    assert h.shape == (B, HW, C) and y.shape == (B, HW, 3 * C)
    assert rel_l2(h, h_ref) < 3e-3 and rel_l2(y, y_ref) < 4e-3
    hs = ops.linear(ops.groupnorm_silu(x, gam, bet, groups=32, eps=1e-6, silu=False), w1, b1)
    ys = ops.linear(hs, w2p, b2p, colsum=cs, ln_eps=1e-5)
    assert rel_l2(h, hs.float()) < 3e-3 and rel_l2(y, ys.float()) < 4e-3


@pytest.mark.parametrize("offset,sigma", [(0.5, 1.0), (24.0, 0.8), (-60.0, 0.5)])
def test_row320_chain_centered_map_survives_large_means(ops, dev, offset, sigma):
    """The chained SpatialTransformer entry applies the GroupNorm as an fp16 map.  With producer statistics the map is centred --
    (x - fp16(mean)) * scale + shift' -- so its error is 2^-11 of the normalised value; the plain x * scale + shift form loses
    |mean| / sigma * 2^-11 (shown here at |mean| / sigma = 30 and 120)."""
    from lib.model_zoo.hip_layers import fold_layernorm
    B, HW, C = 2, 4096, 320
    x = (rnd((B, HW, C), dev, sigma, 960).float() + offset + 0.3 * rnd((1, 1, C), dev, sigma, 961).float()).half()
    gam, bet = 1.0 + 0.2 * rnd((C,), dev, 1.0, 962), 0.1 * rnd((C,), dev, 1.0, 963)
    w1, b1 = rnd((C, C), dev, C ** -0.5, 964), rnd((C,), dev, 0.2, 965)
    w2 = rnd((3 * C, C), dev, C ** -0.5, 966)
    ln = torch.nn.LayerNorm(C, eps=1e-5).to(dev)
    xn = F.group_norm(x.float().transpose(1, 2), 32, gam.float(), bet.float(), 1e-6).transpose(1, 2)
    h_ref = xn @ w1.float().t() + b1.float()
    w2p, b2p, _ = fold_layernorm(w2, None, ln)
    x._vd_stats = ops.chan_stats(x, 256)   # what a producer attaches
    sc, sh, ct = ops.groupnorm_affine(x, gam, bet, groups=32, eps=1e-6, centered=True)
    assert ct is not None
    h, _ = ops.row320_chain(x, sc, sh, HW, w1, b1, w2p, b2p, 1e-5, center=ct)
    assert rel_l2(h, h_ref) < 3e-3
    sc0, sh0 = ops.groupnorm_affine(x, gam, bet, groups=32, eps=1e-6)
    h0, _ = ops.row320_chain(x, sc0, sh0, HW, w1, b1, w2p, b2p, 1e-5)
    if abs(offset) / sigma > 20:
        assert rel_l2(h0, h_ref) > 2 * rel_l2(h, h_ref)   # the plain form is what the centred one repairs


@pytest.mark.parametrize("B,H,D,Nq,Nk,offset", [
    (2, 8, 40, 1024, 77, 0.0), (8, 8, 40, 4096, 77, 0.3), (1, 8, 40, 200, 77, -1.5), (2, 8, 80, 256, 257, 0.0),
    (8, 8, 80, 1024, 77, 2.0), (2, 8, 160, 64, 514, 0.0), (3, 8, 160, 100, 77, 0.5), (8, 8, 160, 256, 77, 0.0),
    (1, 8, 40, 130, 64, 0.0), (2, 8, 80, 96, 128, 4.0)])
def test_xattn_fused(ops, dev, B, H, D, Nq, Nk, offset):
    """vd_xattn_f16 (LayerNorm -> to_q -> softmax(q k^T) v in one launch) against torch fp32 and against the library's own
    row_stats -> GEMM -> attention chain: contexts of 77 / 257 / 514 keys (514 = 8 tiles + 2), ragged query blocks, batch x
    query-block counts that are / are not a multiple of the 8 XCDs, rows with a common offset (in-loop statistics)."""
    from lib.model_zoo.hip_layers import fold_layernorm
    C = H * D
    assert ops.xattn_supported(H, D)
    x = rnd((B, Nq, C), dev, 1.1, 700) + offset
    wq = rnd((C, C), dev, C ** -0.5, 701)
    kv = rnd((B, Nk, 2 * C), dev, 1.0, 702)
    ln = torch.nn.LayerNorm(C, eps=1e-5).to(dev)
    with torch.no_grad():
        ln.weight.copy_(1.0 + 0.2 * torch.randn(C, device=dev))
        ln.bias.copy_(0.1 * torch.randn(C, device=dev))
    w, b, cs = fold_layernorm(wq, None, ln)
    out = ops.xattn(x, w, b, cs, 1e-5, kv[..., :C], kv[..., C:], H)
    xn = F.layer_norm(x.float(), (C,), ln.weight.float(), ln.bias.float(), 1e-5)
    q = (xn @ wq.float().t()).view(B, Nq, H, D).transpose(1, 2)
    k = kv[..., :C].float().view(B, Nk, H, D).transpose(1, 2)
    v = kv[..., C:].float().view(B, Nk, H, D).transpose(1, 2)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * D ** -0.5, -1) @ v).transpose(1, 2).reshape(B, Nq, C)
    assert out.shape == (B, Nq, C) and bool(torch.isfinite(out).all())
    assert rel_l2(out, ref) < 4e-3
    qc = ops.linear(x, w, b, colsum=cs, ln_eps=1e-5)
    chain = ops.attention(qc, kv[..., :C], kv[..., C:], H)
    assert rel_l2(out, chain.float()) < 4e-3


@pytest.mark.parametrize("rows,C,ld", [(1000, 320, 320), (4099, 640, 640), (77, 1280, 1280), (300, 768, 800), (33, 2048, 2048), (5, 64, 64)])
def test_row_stats(ops, dev, rows, C, ld):
    """vd_row_stats_f16 (statistics of the folded LayerNorm) vs torch fp32, incl. a padded leading dimension, rows that
    do not fill a block, and a large common offset (two-pass: no cancellation)."""
    x = rnd((rows, ld), dev, 1.5, 90) + 6.0
    st = ops.row_stats(x, C, rows, 1e-5, ldx=ld)
    xf = x[:, :C].float()
    mean = xf.mean(1)
    rstd = (xf.var(1, unbiased=False) + 1e-5).rsqrt()
    assert st.shape == (rows, 2) and st.dtype == torch.float32
    assert torch.allclose(st[:, 0], mean, rtol=1e-5, atol=1e-5) and torch.allclose(st[:, 1], rstd, rtol=1e-4, atol=0)


def test_gemm_every_tile_configuration(ops, dev):
    """Every instantiation of the GEMM template (vd_gemm_set_override) on a conv with concat + row vector, a plain GEMM
    with ragged M / N / K and bias + residual, a split-K problem, a LayerNorm-folded projection and a GEGLU projection
    (tiles whose width is not a multiple of 128 columns cannot pair value / gate columns: the planner's choice runs)."""
    from lib.model_zoo.hip_layers import fold_layernorm
    from vd_hip.pack import pack_geglu
    from vd_hip.loader import lib
    from vd_hip.pack import pack_conv_weight
    B, H, W, c0, c1, Co = 2, 16, 16, 128, 64, 384
    x0, x1 = rnd((B, H, W, c0), dev, 1.0, 61), rnd((B, H, W, c1), dev, 1.0, 62)
    wt = rnd((Co, c0 + c1, 3, 3), dev, 0.05, 63)
    bias, rv = rnd((Co,), dev, 0.5, 64), rnd((B, Co), dev, 0.5, 65)
    ref_conv = _conv_ref(torch.cat([x0, x1], -1), wt, bias, 1, 1, 0) + rv.float().view(B, 1, 1, Co)
    a, w2 = rnd((1000, 200), dev, 1.0, 66), rnd((328, 200), dev, 0.05, 67)
    b2, r2 = rnd((328,), dev, 0.5, 68), rnd((1000, 328), dev, 1.0, 69)
    ref_plain = a.float() @ w2.float().t() + b2.float() + r2.float()
    a3, w3 = rnd((256, 64 * 48), dev, 1.0, 70), rnd((320, 64 * 48), dev, 0.03, 71)
    ref_split = a3.float() @ w3.float().t()
    K = 640
    xl, wl = rnd((700, K), dev, 1.2, 72) + 0.5, rnd((448, K), dev, 0.05, 73)
    ln = torch.nn.LayerNorm(K, eps=1e-5).to(dev)
    ref_ln = F.layer_norm(xl.float(), (K,), ln.weight.float(), ln.bias.float(), 1e-5) @ wl.float().t()
    wlp, blp, cs = fold_layernorm(wl, None, ln)
    xg, wg, bg = rnd((600, 192), dev, 1.0, 74), rnd((1024, 192), dev, 0.05, 75), rnd((1024,), dev, 0.2, 76)
    vg, gg = (xg.float() @ wg.float().t() + bg.float()).chunk(2, dim=-1)
    ref_geglu = vg * F.gelu(gg)
    wgp, bgp = pack_geglu(wg, bg)
    n = lib().vd_gemm_num_configs()
    assert n >= 8
    try:
        built = 0
        for cfg in range(n):
            if lib().vd_gemm_set_override(cfg) != 0:   # slot of a removed development tile
                continue
            built += 1
            name = ops.gemm_kernel_name(cfg)
            out = ops.conv2d_nhwc(x0, pack_conv_weight(wt), bias, x1=x1, rowvec=rv, rows_per_batch=H * W)
            assert rel_l2(out, ref_conv) < 2e-3, name
            out = ops.gemm(a, w2, bias=b2, res=r2)
            assert rel_l2(out, ref_plain) < 2e-3, name
            out = ops.gemm(a3, w3, split_k=3)
            assert rel_l2(out, ref_split) < 2e-3, name
            out = ops.gemm(xl, wlp, bias=blp, colsum=cs, ln_eps=1e-5)
            assert rel_l2(out, ref_ln) < 3e-3, name
            out = ops.gemm(xg, wgp, bias=bgp, act=ops.ACT_GEGLU)
            assert rel_l2(out, ref_geglu) < 2e-3, name
        assert built >= 9
    finally:
        ops.gemm_set_override(-1)


@pytest.mark.parametrize("split", [2, 5, 16, 32])
def test_gemm_split_k(ops, dev, split):
    M, N, K = 192, 320, 64 * 40
    a = rnd((M, K), dev, 1.0, 11)
    w = rnd((N, K), dev, 0.03, 12)
    bias = rnd((N,), dev, 0.5, 13)
    res = rnd((M, N), dev, 1.0, 14)
    ref = a.float() @ w.float().t() + bias.float() + res.float()
    out = ops.gemm(a, w, bias=bias, res=res, split_k=split)
    assert rel_l2(out, ref) < 2e-3
    out2 = ops.gemm(a, w, bias=bias, res=res)  # heuristic split
    assert rel_l2(out2, ref) < 2e-3


@pytest.mark.parametrize("shape", [(192, 320, 64 * 40, 5), (1000, 328, 64 * 24, 3), (2048, 1280, 64 * 90, 3), (130, 70, 64 * 33, 32)])
def test_gemm_split_k_in_kernel_fixup_equals_reduce_kernel(ops, dev, shape):
    """Arrival-counter split-K (last block of a tile sums the slabs) against the two-kernel path: same partial sums, same
    order; the fused epilogue rounds to fp16 before the residual add where the reduce kernel stays in fp32, so the two agree
    to fp16 rounding, and the in-kernel path is bit-identical from launch to launch (slabs are summed in split order whichever
    block arrives last).  Ragged edge tiles, every 64-deep tile shape, repeated launches (the counters re-arm)."""
    from vd_hip.loader import lib
    M, N, K, split = shape
    a, w = rnd((M, K), dev, 1.0, 31), rnd((N, K), dev, 0.03, 32)
    bias, res = rnd((N,), dev, 0.5, 33), rnd((M, N), dev, 1.0, 34)
    ref = a.float() @ w.float().t() + bias.float() + res.float()
    try:
        for cfg in (-1, 0, 1, 2, 3, 4, 7, 13, 14, 15):
            ops.gemm_set_override(cfg)
            two = ops.gemm(a, w, bias=bias, res=res, split_k=split, fixup=False)
            assert rel_l2(two, ref) < 2e-3, cfg
            first = None
            for rep in range(4):
                one = ops.gemm(a, w, bias=bias, res=res, split_k=split, fixup=True)
                assert rel_l2(one, ref) < 2e-3 and rel_l2(one, two) < 1e-3, (cfg, rep)
                first = one if first is None else first
                assert torch.equal(one, first), (cfg, rep)
    finally:
        ops.gemm_set_override(-1)
    assert int(ops.sync_counters(dev).abs().sum()) == 0
    # batched
    Bt = 3
    ab, wb = rnd((Bt, 256, 64 * 16), dev, 1.0, 35), rnd((Bt, 192, 64 * 16), dev, 0.05, 36)
    kw = dict(M=256, N=192, K=64 * 16, batch=Bt, strides=(256 * 64 * 16, 192 * 64 * 16, 256 * 192, 0), split_k=4)
    assert rel_l2(ops.gemm(ab, wb, fixup=True, **kw), ops.gemm(ab, wb, fixup=False, **kw)) < 1e-3


def test_gemm_batched_f32_and_bias_along_m(ops, dev):
    Bt, M, N, K = 3, 200, 136, 128
    a = rnd((Bt, M, K), dev, 1.0, 15)
    w = rnd((Bt, N, K), dev, 0.1, 16)
    ref = torch.einsum("bmk,bnk->bmn", a.float(), w.float()) * 0.25
    out = ops.gemm(a, w, M=M, N=N, K=K, batch=Bt, strides=(M * K, N * K, M * N, 0), alpha=0.25, out_f32=True)
    assert out.dtype == torch.float32 and out.shape == (Bt, M, N)
    assert rel_l2(out, ref) < 1e-3
    # shared A (stride 0), bias along M  (V^T = Wv x^T + bv)
    wv = rnd((M, K), dev, 0.1, 17)
    bv = rnd((M,), dev, 0.5, 18)
    ref2 = torch.einsum("mk,bnk->bmn", wv.float(), w.float()) + bv.float().view(1, M, 1)
    out2 = ops.gemm(wv, w, bias=bv, bias_along_m=True, M=M, N=N, K=K, batch=Bt, strides=(0, N * K, M * N, 0))
    assert rel_l2(out2, ref2) < 2e-3


def _conv_ref(x_nhwc, w, b, stride, pad, ups, pad_hi=None):
    x = x_nhwc.float().permute(0, 3, 1, 2)
    if ups:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    if pad_hi is not None:
        x = F.pad(x, (pad, pad_hi, pad, pad_hi))
        y = F.conv2d(x, w.float(), b.float() if b is not None else None, stride=stride)
    else:
        y = F.conv2d(x, w.float(), b.float() if b is not None else None, stride=stride, padding=pad)
    return y.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("cfg", [
    dict(B=2, H=16, W=16, Cin=64, Cout=128, stride=1, pad=1, ups=0),
    dict(B=2, H=16, W=16, Cin=320, Cout=320, stride=1, pad=1, ups=0),
    dict(B=1, H=32, W=32, Cin=64, Cout=64, stride=2, pad=1, ups=0),
    dict(B=2, H=8, W=8, Cin=128, Cout=64, stride=1, pad=1, ups=1),
    dict(B=1, H=17, W=13, Cin=64, Cout=72, stride=1, pad=1, ups=0),
    dict(B=1, H=16, W=16, Cin=64, Cout=64, stride=2, pad=0, ups=0, pad_hi=1),
    dict(B=2, H=8, W=8, Cin=1280, Cout=1280, stride=1, pad=1, ups=0),
    dict(B=1, H=64, W=64, Cin=128, Cout=3, stride=1, pad=1, ups=0),
])
def test_conv3x3(ops, dev, cfg):
    from vd_hip.pack import pack_conv_weight
    B, H, W, Cin, Cout = cfg["B"], cfg["H"], cfg["W"], cfg["Cin"], cfg["Cout"]
    x = rnd((B, H, W, Cin), dev, 1.0, 20)
    w = rnd((Cout, Cin, 3, 3), dev, 0.03, 21)
    b = rnd((Cout,), dev, 0.3, 22)
    ref = _conv_ref(x, w, b, cfg["stride"], cfg["pad"], cfg["ups"], cfg.get("pad_hi"))
    out = ops.conv2d_nhwc(x, pack_conv_weight(w), b, ksize=3, stride=cfg["stride"], pad=cfg["pad"], ups=cfg["ups"],
                          pad_hi=cfg.get("pad_hi"))
    assert out.shape == ref.shape
    assert rel_l2(out, ref) < 2e-3


def test_conv3x3_concat_rowvec_residual(ops, dev):
    """ResBlock first conv on cat([h, skip]) + emb broadcast; second conv + residual."""
    from vd_hip.pack import pack_conv_weight
    B, H, W, C0, C1, Cout = 2, 16, 16, 128, 64, 128
    x0 = rnd((B, H, W, C0), dev, 1.0, 23)
    x1 = rnd((B, H, W, C1), dev, 1.0, 24)
    w = rnd((Cout, C0 + C1, 3, 3), dev, 0.03, 25)
    b = rnd((Cout,), dev, 0.3, 26)
    emb = rnd((B, Cout), dev, 0.5, 27)
    res = rnd((B, H, W, Cout), dev, 1.0, 28)
    ref = _conv_ref(torch.cat([x0, x1], -1), w, b, 1, 1, 0) + emb.float().view(B, 1, 1, Cout) + res.float()
    out = ops.conv2d_nhwc(x0, pack_conv_weight(w), b, x1=x1, rowvec=emb, rows_per_batch=H * W, res=res)
    assert rel_l2(out, ref) < 2e-3
    # 1x1 skip conv on the concatenation
    w1 = rnd((Cout, C0 + C1, 1, 1), dev, 0.05, 29)
    ref1 = _conv_ref(torch.cat([x0, x1], -1), w1, b, 1, 0, 0)
    out1 = ops.conv2d_nhwc(x0, pack_conv_weight(w1), b, x1=x1, ksize=1, pad=0)
    assert rel_l2(out1, ref1) < 2e-3


@pytest.mark.parametrize("case", [
    # B, H, W, c0, c1, Cout, ups, rowvec, residual
    (2, 32, 32, 128, 64, 320, 0, True, False),     # two-source concat + per-image row vector (ResBlock conv 1 on a skip concat)
    (1, 64, 64, 320, 0, 320, 0, False, True),      # residual (ResBlock conv 2); 5 channel chunks
    (8, 8, 8, 256, 128, 320, 0, True, True),       # four whole 8x8 images per patch
    (2, 16, 16, 128, 0, 128, 1, False, False),     # nearest-2x upsample in front, 16-wide patches
    (1, 96, 96, 64, 0, 160, 0, False, True),       # 768^2 latent geometry (96 = 3 patches of 32)
    (8, 16, 16, 640, 0, 1280, 0, True, False),     # split over channel chunks (fp32 slabs + reduce kernel)
    (2, 64, 32, 64, 0, 72, 0, False, False),       # N not a multiple of the column tile
    (8, 8, 8, 640, 0, 1280, 0, True, True),        # 8x8 level (small-M variant 10: two images x 32 columns per block, whole K)
    (2, 8, 8, 128, 64, 96, 0, False, False),       # ... with one patch and a concat
])
def test_conv3x3_halo_every_variant(ops, dev, case):
    """conv3x3_halo_kernel: every instantiation (vd_conv_halo_set_variant) and the planner's own choice against torch's fp32
    convolution; setting 0 runs the same problem on gemm_f16_kernel."""
    from vd_hip.loader import lib
    from vd_hip.pack import pack_conv_weight
    B, H, W, c0, c1, Co, ups, rv, rs = case
    x = rnd((B, H, W, c0), dev, 1.0, 100)
    x1 = rnd((B, H, W, c1), dev, 1.0, 101) if c1 else None
    wt = rnd((Co, c0 + c1, 3, 3), dev, 0.04, 102)
    b = rnd((Co,), dev, 0.3, 103)
    Hv, Wv = H << ups, W << ups
    rowvec = rnd((B, Co), dev, 0.5, 104) if rv else None
    res = rnd((B, Hv, Wv, Co), dev, 1.0, 105) if rs else None
    ref = _conv_ref(torch.cat([x, x1], -1) if c1 else x, wt, b, 1, 1, ups)
    if rv:
        ref = ref + rowvec.float().view(B, 1, 1, Co)
    if rs:
        ref = ref + res.float()
    kw = dict(ksize=3, pad=1, ups=ups, x1=x1)
    if rv:
        kw.update(rowvec=rowvec, rows_per_batch=Hv * Wv)
    if rs:
        kw.update(res=res)
    wp = pack_conv_weight(wt)
    try:
        for v in (-1, 0, 3, 6, 13):   # planner / off (gemm_f16_kernel) / the three instantiated variants (v - 1 = 2, 5, 12)
            assert lib().vd_conv_halo_set_variant(v) == 0
            out = ops.conv2d_nhwc(x, wp, b, **kw)
            assert out.shape == ref.shape and rel_l2(out, ref) < 2e-3, v
    finally:
        lib().vd_conv_halo_set_variant(-1)


@pytest.mark.parametrize("B,HW,c0,c1,silu,eps", [(2, 4096, 320, 0, True, 1e-5), (2, 256, 1280, 1280, True, 1e-5),
                                                 (3, 1024, 640, 320, True, 1e-5), (1, 64, 1280, 0, False, 1e-6),
                                                 (1, 16384, 128, 0, True, 1e-6), (2, 4096, 320, 0, False, 1e-6),
                                                 (1, 100, 256, 0, True, 1e-6), (1, 300, 1920, 0, True, 1e-5),
                                                 # 5 / 6 / 2 channels per group: 8 consecutive channels span 3+ groups
                                                 (1, 200, 160, 0, True, 1e-5), (2, 4096, 192, 0, True, 1e-5),
                                                 (2, 4096, 160, 0, False, 1e-5), (1, 512, 64, 0, False, 1e-6)])
def test_groupnorm(ops, dev, B, HW, c0, c1, silu, eps):
    x0 = rnd((B, HW, c0), dev, 2.0, 30) + 0.5
    x1 = rnd((B, HW, c1), dev, 1.0, 31) if c1 else None
    C = c0 + c1
    gamma = rnd((C,), dev, 0.5, 32) + 1.0
    beta = rnd((C,), dev, 0.5, 33)
    x = torch.cat([x0, x1], -1) if c1 else x0
    ref = F.group_norm(x.float().permute(0, 2, 1), 32, gamma.float(), beta.float(), eps).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    out = ops.groupnorm_silu(x0, gamma, beta, x1=x1, groups=32, eps=eps, silu=silu)
    assert out.shape == (B, HW, C)
    assert rel_l2(out, ref) < 2e-3


def test_single_launch_groupnorms_are_bit_identical_run_to_run(ops, dev):
    """gn_slab_kernel / gn0d_kernel fold their per-thread partials in LDS as 64-bit integers at a block-wide power-of-two scale (round 6;
    float atomics before: arrival order, so two runs could differ in the last bits) -- 30 runs each must agree exactly, also for
    tiny and for large-mean inputs (the scale follows the block's largest partial)."""
    for scale, shift in ((2.0, 0.5), (1e-3, 0.0), (0.05, 300.0)):
        x = rnd((4, 16, 1280), dev, scale, 50) + shift       # the 4x4 level of a 32x32-latent forward: slab kernel
        gamma, beta = rnd((1280,), dev, 0.5, 51) + 1.0, rnd((1280,), dev, 0.5, 52)
        outs = [ops.groupnorm_silu(x, gamma, beta, groups=32, eps=1e-5, silu=True) for _ in range(30)]
        assert all(torch.equal(o, outs[0]) for o in outs[1:])
        ref = F.silu(F.group_norm(x.float().permute(0, 2, 1), 32, gamma.float(), beta.float(), 1e-5).permute(0, 2, 1))
        assert rel_l2(outs[0], ref) < (2e-3 if shift < 100 else 2e-2)   # (fp16 input at mean 300: the input's own rounding dominates)
        x0 = rnd((3, 4, 1280), dev, scale, 53) + shift
        g2, b2 = rnd((4, 1280), dev, 0.5, 54) + 1.0, rnd((4, 1280), dev, 0.5, 55)
        outs = [ops.groupnorm0d_silu(x0, g2, b2, groups=32, eps=1e-5, silu=True) for _ in range(30)]
        assert all(torch.equal(o, outs[0]) for o in outs[1:])


@pytest.mark.parametrize("HW,C", [(4096, 320), (256, 1280), (64, 1280)])
def test_groupnorm_large_mean_small_spread(ops, dev, HW, C):
    """|mean| >> sigma with eps = 1e-6 (VAE / SpatialTransformer norms): a plain E[x^2] - mean^2 in fp32 loses the
    variance to cancellation here; the kernels' shifted sums must match torch's two-pass statistics.  Covers the
    partial + apply path (HW = 4096) and the single-launch slab path."""
    g = torch.Generator().manual_seed(70 + HW)
    x = (16.0 + 0.03 * torch.randn((2, HW, C), generator=g)).half().to(dev)   # fp16 spacing at 16 is 0.0156: the spread survives
    gamma = rnd((C,), dev, 0.5, 71) + 1.0
    beta = rnd((C,), dev, 0.5, 72)
    ref = F.group_norm(x.float().permute(0, 2, 1), 32, gamma.float(), beta.float(), 1e-6).permute(0, 2, 1)
    out = ops.groupnorm_silu(x, gamma, beta, groups=32, eps=1e-6, silu=False)
    assert rel_l2(out, ref) < 3e-3


@pytest.mark.parametrize("rows,C", [(1000, 320), (77, 768), (513, 1280), (257, 1024), (5, 640)])
def test_layernorm(ops, dev, rows, C):
    x = rnd((rows, C), dev, 2.0, 40) + 0.3
    g = rnd((C,), dev, 0.5, 41) + 1.0
    b = rnd((C,), dev, 0.5, 42)
    ref = F.layer_norm(x.float(), (C,), g.float(), b.float(), 1e-5)
    out = ops.layernorm(x, g, b, 1e-5)
    assert rel_l2(out, ref) < 2e-3


def _attn_ref(q, k, v, heads, scale, causal):
    B, Nq, C = q.shape
    Nk = k.shape[1]
    D = C // heads
    qf = q.float().view(B, Nq, heads, D).transpose(1, 2)
    kf = k.float().view(B, Nk, heads, D).transpose(1, 2)
    vf = v.float().view(B, Nk, heads, D).transpose(1, 2)
    s = qf @ kf.transpose(-1, -2) * scale
    if causal:
        mask = torch.triu(torch.ones(Nq, Nk, dtype=torch.bool, device=q.device), 1)
        s = s.masked_fill(mask, float("-inf"))
    p = s.softmax(-1)
    return (p @ vf).transpose(1, 2).reshape(B, Nq, C)


@pytest.mark.parametrize("B,H,Nq,Nk,D,causal", [
    (2, 8, 1024, 1024, 40, False), (1, 8, 4096, 4096, 40, False), (2, 8, 1024, 77, 40, False),
    (2, 8, 256, 256, 80, False), (2, 8, 1024, 257, 80, False), (2, 8, 64, 64, 160, False),
    (2, 8, 256, 77, 160, False), (3, 12, 77, 77, 64, True), (2, 16, 257, 257, 64, False),
    (1, 8, 200, 514, 40, False), (1, 3, 130, 70, 64, True)])
def test_attention(ops, dev, B, H, Nq, Nk, D, causal):
    C = H * D
    q = rnd((B, Nq, C), dev, 1.0, 50)
    k = rnd((B, Nk, C), dev, 1.0, 51)
    v = rnd((B, Nk, C), dev, 1.0, 52)
    scale = D ** -0.5
    ref = _attn_ref(q, k, v, H, scale, causal)
    out = ops.attention(q, k, v, H, scale=scale, causal=causal)
    assert rel_l2(out, ref) < 3e-3


@pytest.mark.parametrize("B,H,Nq,Nk,spike,gain", [(1, 8, 2048, 1024, None, 1.0), (1, 8, 2100, 1100, 700, 1.0), (1, 4, 2304, 1089, 1088, 1.0),
                                                   (2, 2, 2048, 1088, 3, 1.0), (1, 8, 9216, 9216, 5000, 1.0), (1, 2, 2048, 1025, 64, 1.0),
                                                   (1, 2, 2560, 1024, 40, 1.0), (1, 4, 2048, 2048, 1500, 3.0), (3, 3, 2048, 1030, None, 2.5)])
def test_attention_pipelined_long(ops, dev, B, H, Nq, Nk, spike, gain):
    """`attn_pipe_kernel` (D = 40, Nq >= 2048, Nk >= 1024: the 64x64 / 96x96-level self-attention; 64 queries per wave, software-
    pipelined at 32-key steps over a 4-slot ring, running max in Q's spare k-slot as an fp16 value, constant LDS columns -- round 6)
    against fp32: even / odd tile counts, a last tile of one or six keys, ragged query blocks (one row block of a wave empty), a
    block count that is not a multiple of 8, the 768x768 geometry, a spiked key in the first / a middle / the last tile (deferred
    rescale with scores and probabilities in flight) and logits of +-40 (the fp16-rounded max must keep P finite)."""
    D = 40
    C = H * D
    qkv = rnd((B, max(Nq, Nk), 3 * C), dev, 1.0, 70 + Nk)
    qkv[..., :2 * C] *= gain
    if spike is not None:
        qkv[:, spike, C:2 * C] *= 10.0 / gain
    q, k, v = qkv[:, :Nq, :C], qkv[:, :Nk, C:2 * C], qkv[:, :Nk, 2 * C:]
    ref = _attn_ref(q.contiguous(), k.contiguous(), v.contiguous(), H, D ** -0.5, False)
    out = ops.attention(q, k, v, H)
    assert bool(torch.isfinite(out).all())
    assert rel_l2(out, ref) < (3e-3 if gain == 1.0 else 5e-3)
    if gain == 1.0 and spike is None:
        out4 = ops.attention(q[:, :1024], k, v, H)   # Nq < 2048: the serial 4-wave kernel on the same keys
        assert rel_l2(out[:, :1024], out4) < 2e-3


@pytest.mark.parametrize("B,N,D", [(2, 1024, 512), (1, 4096, 512), (1, 1000, 512), (2, 333, 256), (3, 64, 128)])
def test_attention_one_wide_head(ops, dev, B, N, D):
    """vd_attention_f16 with one head of 128 / 256 / 512 channels (AutoencoderKL mid-block AttnBlock): head dim split over the
    four waves of a block, partial scores exchanged through LDS; ragged key / query counts; q / k / v as column slices of a
    fused projection; one spiked key forces the deferred rescale."""
    qkv = rnd((B, N, 3 * D), dev, 0.6, 90)
    qkv[:, N // 2, D:2 * D] *= 6.0
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    scale = D ** -0.5
    ref = torch.softmax((q.float() @ k.float().transpose(1, 2)) * scale, dim=-1) @ v.float()
    out = ops.attention(q, k, v, 1, scale=scale)
    assert out.shape == (B, N, D) and rel_l2(out, ref) < 3e-3


def test_attention_fused_qkv_views_and_spike(ops, dev):
    """Strided column views of one fused projection; a spiked key forces a late running-max jump."""
    B, N, H, D = 2, 512, 8, 40
    C = H * D
    qkv = rnd((B, N, 3 * C), dev, 1.0, 53)
    qkv[:, 300, C:2 * C] *= 12.0  # one key with huge norm -> max jumps in tile 4
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    ref = _attn_ref(q.contiguous(), k.contiguous(), v.contiguous(), H, D ** -0.5, False)
    out = ops.attention(q, k, v, H)
    assert rel_l2(out, ref) < 3e-3


@pytest.mark.parametrize("D,gain", [(40, 3.0), (80, 3.0), (64, 4.0), (160, 2.5)])
def test_attention_large_logits_and_descending_max(ops, dev, D, gain):
    """Logits of +-40 and more (natural units): the running max rides in the MFMA C operand and P = exp2(s - m) must stay
    finite in fp16 through the deferred-rescale window; keys are ordered so the row max first rises, then falls."""
    B, N, H = 1, 640, 4
    C = H * D
    q = rnd((B, N, C), dev, gain, 60)
    k = rnd((B, N, C), dev, gain, 61)
    ramp = torch.cat([torch.linspace(0.2, 2.0, N // 2), torch.linspace(2.0, 0.05, N - N // 2)]).to(dev)
    k = (k.float() * ramp[None, :, None]).half()
    v = rnd((B, N, C), dev, 1.0, 62)
    ref = _attn_ref(q, k, v, H, D ** -0.5, False)
    out = ops.attention(q, k, v, H)
    assert bool(torch.isfinite(out).all())
    assert rel_l2(out, ref) < 5e-3


def test_softmax_rows(ops, dev):
    s = torch.randn(300, 4096, device=dev) * 5
    ref = s.softmax(-1)
    out = ops.softmax_rows(s)
    assert rel_l2(out, ref) < 2e-3


def test_timestep_embedding(ops, dev):
    t = torch.tensor([981, 1, 500, 21], dtype=torch.int64, device=dev)
    dim = 320
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32, device=dev) / half)
    args = t[:, None].float() * freqs[None]
    ref = torch.cat([torch.cos(args), torch.sin(args)], -1)
    out = ops.timestep_embedding(t, dim)
    assert (out.float() - ref).abs().max().item() < 2e-3
    # golden values from the reference
    assert abs(out[0, 0].item() - 0.67995721) < 1e-3 and abs(out[0, half].item() - 0.73325181) < 1e-3


@pytest.mark.parametrize("guided", [True, False])
def test_cfg_ddim_step(ops, dev, guided):
    x = rnd((4, 4, 64, 64), dev, 1.0, 60)
    eps = rnd((8 if guided else 4, 4, 64, 64), dev, 1.0, 61)
    noise = rnd((4, 4, 64, 64), dev, 1.0, 62)
    a_t, a_prev, sigma, s = 0.5888, 0.7521, 0.1, 7.5
    e = eps.float()
    if guided:
        eu, ec = e.chunk(2)
        e = eu + s * (ec - eu)
    p0 = (x.float() - math.sqrt(1 - a_t) * e) / math.sqrt(a_t)
    ref = math.sqrt(a_prev) * p0 + math.sqrt(1 - a_prev - sigma ** 2) * e + sigma * noise.float()
    xp, px0 = ops.cfg_ddim_step(x, eps, guided=guided, guidance_scale=s, a_t=a_t, a_prev=a_prev, sigma=sigma,
                                sqrt_one_minus_at=math.sqrt(1 - a_t), noise=noise)
    assert rel_l2(xp, ref) < 1e-3 and rel_l2(px0, p0) < 1e-3


def test_q_sample_axpby_layouts(ops, dev):
    x0 = rnd((3, 4, 32, 32), dev, 1.0, 63)
    nz = rnd((3, 4, 32, 32), dev, 1.0, 64)
    sa = torch.tensor([0.9, 0.5, 0.1], device=dev)
    sb = torch.tensor([0.3, 0.8, 0.99], device=dev)
    ref = sa.view(3, 1, 1, 1) * x0.float() + sb.view(3, 1, 1, 1) * nz.float()
    assert rel_l2(ops.q_sample(x0, nz, sa, sb), ref) < 1e-3
    assert rel_l2(ops.axpby(x0, nz, 0.4, 0.6), 0.4 * x0.float() + 0.6 * nz.float()) < 1e-3
    y = rnd((1, 7), dev, 1.0, 65)
    assert rel_l2(ops.axpby(y, y, 2.0, 1.0), 3.0 * y.float()) < 1e-3
    x = rnd((2, 5, 20, 33), dev, 1.0, 66)
    nhwc = ops.nchw_to_nhwc(x)
    assert torch.equal(nhwc, x.permute(0, 2, 3, 1).contiguous())
    back = ops.nhwc_to_nchw(nhwc)
    assert torch.equal(back, x)
    img = ops.nhwc_to_nchw(nhwc, scale=0.5, shift=0.5, clamp01=True)
    assert rel_l2(img, torch.clamp((x.float() + 1) / 2, 0, 1)) < 1e-3


@pytest.mark.parametrize("layout,C,ks,stride,pad,pad_hi", [("nchw", 4, 3, 1, 1, None), ("nchw", 3, 3, 1, 1, None),
                                                           ("nhwc", 8, 1, 1, 0, None), ("nhwc", 4, 1, 1, 0, None)])
def test_im2col_small_conv(ops, dev, layout, C, ks, stride, pad, pad_hi):
    from vd_hip.pack import pack_conv_weight_small
    B, H, W, Cout = 2, 24, 20, 64
    x = rnd((B, C, H, W), dev, 1.0, 70)
    w = rnd((Cout, C, ks, ks), dev, 0.2, 71)
    b = rnd((Cout,), dev, 0.3, 72)
    ref = F.conv2d(x.float() * 2 - 1, w.float(), b.float(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    xin = x if layout == "nchw" else x.permute(0, 2, 3, 1).contiguous()
    a, (Bo, Ho, Wo) = ops.im2col_small(xin, layout=layout, ksize=ks, stride=stride, pad=pad, in_scale=2.0, in_shift=-1.0)
    out = ops.gemm(a, pack_conv_weight_small(w), bias=b).view(Bo, Ho, Wo, Cout)
    assert rel_l2(out, ref) < 2e-3


def test_diag_gaussian_sample(ops, dev):
    B, zc, H, W = 2, 4, 16, 16
    mom = rnd((B, H, W, 2 * zc), dev, 2.0, 73)
    nz = rnd((B, zc, H, W), dev, 1.0, 74)
    m = mom.float().permute(0, 3, 1, 2)
    mean, logvar = m.chunk(2, 1)
    ref = (mean + torch.exp(0.5 * logvar.clamp(-30, 20)) * nz.float()) * 0.18215
    out = ops.diag_gaussian_sample(mom, nz, B, zc, H, W, 0.18215)
    assert rel_l2(out, ref) < 2e-3


def test_clip_helpers(ops, dev):
    B, L, C, V = 2, 77, 768, 1000
    ids = torch.randint(0, V, (B, L), device=dev)
    tok = rnd((V, C), dev, 1.0, 80)
    pos = rnd((L, C), dev, 1.0, 81)
    assert rel_l2(ops.embed_tokens(ids, tok, pos), tok.float()[ids] + pos.float()) < 1e-3
    # vision embed
    P, G, Cv = 14, 4, 256
    px = rnd((B, 3, G * P, G * P), dev, 1.0, 82)
    wpe = rnd((Cv, 3, P, P), dev, 0.05, 83)
    from vd_hip.pack import pack_patch_weight
    a = ops.patchify(px, P)
    pe = ops.gemm(a, pack_patch_weight(wpe)).view(B, G * G, Cv)
    ref_pe = F.conv2d(px.float(), wpe.float(), stride=P).flatten(2).transpose(1, 2)
    assert rel_l2(pe, ref_pe) < 2e-3
    cls = rnd((Cv,), dev, 1.0, 84)
    posv = rnd((G * G + 1, Cv), dev, 1.0, 85)
    tsc = torch.rand(B, G * G + 1, device=dev)
    ref_e = (torch.cat([cls.float().expand(B, 1, Cv), pe.float()], 1) + posv.float()) * tsc[..., None]
    assert rel_l2(ops.clip_vision_embed(pe, cls, posv, tsc), ref_e) < 2e-3
    # pooled-norm scaling
    z = rnd((B, L, C), dev, 1.0, 86)
    idx = torch.tensor([5, 76], dtype=torch.int32, device=dev)
    ref_z = z.float() / z.float()[torch.arange(B), idx.long()].norm(dim=-1).view(B, 1, 1)
    out = ops.scale_by_row_norm_(z.clone(), pool_idx=idx)
    assert rel_l2(out, ref_z) < 2e-3
    rs = torch.rand(B, L, device=dev)
    out2 = ops.scale_by_row_norm_(z.clone(), row_scale=rs)
    ref2 = z.float() / z.float()[:, 0:1].norm(dim=-1, keepdim=True) * rs[..., None]
    assert rel_l2(out2, ref2) < 2e-3


def test_error_reporting(ops, dev):
    from vd_hip import VdHipError
    a = rnd((64, 68), dev)
    w = rnd((64, 68), dev)
    with pytest.raises(VdHipError, match="multiple of 8"):
        ops.gemm(a, w)
    with pytest.raises(VdHipError, match="GPU"):
        ops.layernorm(torch.zeros(4, 64, dtype=torch.float16), torch.zeros(64).half(), torch.zeros(64).half())


@pytest.mark.parametrize("B,H,W,dtype", [(2, 512, 512, torch.float16), (1, 768, 512, torch.float32), (1, 224, 224, torch.float32),
                                         (3, 100, 60, torch.float16)])
def test_mask_patch_weights(ops, dev, B, H, W, dtype):
    """vd_mask_patch_weights vs the reference's arithmetic (clip.py:104-122): clamp, F.interpolate(bilinear) to 224^2,
    conv2d with a ones kernel / 196, global mean in front."""
    g = torch.Generator().manual_seed(80 + H)
    m = (torch.rand((B, 1, H, W), generator=g) * 1.4 - 0.2).to(dtype)     # values outside [0,1] exercise the clamp
    m[0, :, : H // 3] = 1.0
    mm = torch.clamp(m.float(), 0, 1)
    mm = F.interpolate(mm, [224, 224], mode="bilinear")
    gs = mm.mean(dim=[1, 2, 3]).view(B, 1)
    vt = F.conv2d(mm, torch.ones(1, 1, 14, 14), stride=14).flatten(1) / 196.0
    ref = torch.cat([gs, vt], 1)
    out = ops.mask_patch_weights(m.to(dev))
    assert out.shape == (B, 257) and out.dtype == torch.float32
    assert (out.cpu() - ref).abs().max().item() < 2e-5
    ones = ops.mask_patch_weights(torch.ones((1, 1, 64, 64), device=dev))
    assert (ones - 1.0).abs().max().item() < 1e-6


def test_color_adjust(ops, dev):
    """'Simple' colour adjustment (app.py:373-379) vs the same formula in fp32."""
    g = torch.Generator().manual_seed(90)
    im = (torch.rand((3, 3, 128, 96), generator=g) * torch.tensor([0.5, 0.9, 0.3]).view(1, 3, 1, 1) + 0.05).half()
    cx = (torch.rand((1, 3, 128, 96), generator=g) * 0.6 + 0.3).half()
    def ref_of(x, c):
        xm, xs = x.float().view(3, -1).mean(-1)[:, None, None], x.float().view(3, -1).std(-1)[:, None, None]
        cm, cs = c.float().view(3, -1).mean(-1)[:, None, None], c.float().view(3, -1).std(-1)[:, None, None]
        return torch.clamp((x.float() - xm) / xs * cs + cm, 0, 1)
    ref = torch.stack([ref_of(i, cx[0]) for i in im])
    out = ops.color_adjust(im.to(dev), cx.to(dev))
    assert (out.float().cpu() - ref).abs().max().item() < 2e-3
    from lib.app_ops import color_adjust_simple
    lst = color_adjust_simple([i.to(dev) for i in im], cx.to(dev))
    assert isinstance(lst, list) and torch.equal(torch.stack(lst), out)
    per = ops.color_adjust(im.to(dev), im.flip(0).contiguous().to(dev))     # one reference image per output image
    assert (per.float().cpu() - torch.stack([ref_of(a, b) for a, b in zip(im, im.flip(0))])).abs().max().item() < 2e-3


def test_adjust_rank_vs_reference_fixture(ops, dev):
    """vd_adjust_rank_f16 behind lib.app_ops.adjust_rank (focus control, app.py:48-127) against the fixture the
    reference's own code produced, for semantic (lvl < 0.5) and style (lvl > 0.5) levels; tolerance = fp16 rounding of
    input / output + convergence of the subspace iteration."""
    import numpy as np
    from vdtest_util import load_gold
    from lib.app_ops import adjust_rank
    g = load_gold("adjust_rank.npz")
    x = torch.from_numpy(g["x"]).to(dev)
    ar = adjust_rank(max_drop_rank=[1, 5], q=20)
    for lvl, key in ((0.0, "y_00"), (0.3, "y_03"), (0.8, "y_08"), (1.0, "y_10")):
        y = ar(x, lvl)
        assert y.dtype == torch.float16 and y.shape == x.shape
        assert rel_l2(y.cpu(), torch.from_numpy(g[key])) < 2e-3, lvl
    assert ar(x, 0.5) is x


def test_adjust_rank_batched_and_257_tokens(ops, dev):
    """Batch of 2 different samples and L = 257 (disentanglement_noglobal = False feeds all tokens) against the exact-SVD
    form of the same reconstruction (oracle.adjust_rank.exact)."""
    from oracle import adjust_rank as A
    from lib.app_ops import adjust_rank
    gen = torch.Generator().manual_seed(3)
    xs = []
    for b in range(2):
        u, _ = torch.linalg.qr(torch.randn(257, 40, generator=gen, dtype=torch.float64))
        v, _ = torch.linalg.qr(torch.randn(768, 40, generator=gen, dtype=torch.float64))
        s = (5.0 + b) * 0.8 ** torch.arange(40, dtype=torch.float64)
        xs.append((u * s) @ v.T + 0.003 * torch.randn(257, 768, generator=gen, dtype=torch.float64) + 0.1 * b)
    x = torch.stack(xs).half()
    ar = adjust_rank()
    for lvl in (0.2, 0.9):
        y = ar(x.to(dev), lvl)
        assert rel_l2(y.cpu(), A.exact(x.float(), lvl)) < 2e-3, lvl


# ---- round 4: GroupNorm from producer-emitted per-channel statistics (csrc/gn_fused.hip) ---------------------------------

def _chan_stats_ref(x, B, T):
    """(mean, M2) per channel over T blocks of HW / T rows per sample, torch fp32/fp64 on the fp16 values."""
    C = x.shape[-1]
    xb = x.double().reshape(B * T, -1, C)
    mean = xb.mean(1)
    m2 = ((xb - mean[:, None, :]) ** 2).sum(1)
    return torch.stack([mean, m2], -1).float()


def _stats_close(st, ref, rows):
    """mean to fp32 accuracy, M2 to 1e-3 of its scale (sum over `rows` values of O(1) deviations)."""
    assert st.buf.shape == ref.shape
    dm = (st.buf[..., 0] - ref[..., 0]).abs().max().item()
    scale = ref[..., 1].abs().max().item() + 1e-6
    d2 = (st.buf[..., 1] - ref[..., 1]).abs().max().item() / scale
    assert dm < 2e-4 and d2 < 1e-3, (dm, d2)


@pytest.mark.parametrize("B,HW,C,R", [(2, 4096, 320, 256), (3, 1024, 640, 64), (2, 256, 1280, 256), (4, 64, 1280, 64),
                                      (1, 4096, 200, 128), (2, 300, 64, 100)])
def test_chan_stats(ops, dev, B, HW, C, R):
    x = rnd((B, HW, C), dev, 2.0, 200) + 0.7
    st = ops.chan_stats(x, R)
    assert st.T == HW // R and st.C == C and st.HW == HW
    _stats_close(st, _chan_stats_ref(x, B, HW // R), R)


@pytest.mark.parametrize("B,HW,c0,c1,R0,R1,silu,eps", [
    (2, 4096, 320, 0, 256, 0, True, 1e-5), (2, 4096, 640, 320, 128, 256, True, 1e-5), (3, 1024, 640, 640, 64, 256, True, 1e-5),
    (2, 256, 1280, 1280, 256, 64, True, 1e-5), (4, 64, 1280, 0, 64, 0, False, 1e-6), (2, 4096, 320, 0, 64, 0, False, 1e-6),
    (1, 1024, 1280, 640, 64, 64, True, 1e-5), (1, 16384, 128, 0, 256, 0, True, 1e-6), (2, 1024, 512, 0, 256, 0, True, 1e-6)])
def test_groupnorm_from_stats(ops, dev, B, HW, c0, c1, R0, R1, silu, eps):
    """vd_groupnorm_from_stats_f16 (partials of either source in its own block size) against torch's GroupNorm; also the
    table form (vd_gn_table_f32 + vd_gn_apply_table_f16) and the dispatch inside ops.groupnorm_silu."""
    x0 = rnd((B, HW, c0), dev, 2.0, 210) + 0.5
    x1 = rnd((B, HW, c1), dev, 1.0, 211) - 0.3 if c1 else None
    C = c0 + c1
    gamma = rnd((C,), dev, 0.5, 212) + 1.0
    beta = rnd((C,), dev, 0.5, 213)
    x = torch.cat([x0, x1], -1) if c1 else x0
    ref = F.group_norm(x.float().permute(0, 2, 1), 32, gamma.float(), beta.float(), eps).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    st0 = ops.chan_stats(x0, R0)
    st1 = ops.chan_stats(x1, R1) if c1 else None
    out = ops.groupnorm_from_stats(x0, gamma, beta, st0, x1=x1, st1=st1, groups=32, eps=eps, silu=silu)
    assert out.shape == (B, HW, C) and rel_l2(out, ref) < 2e-3
    table = ops.gn_table(st0, gamma, beta, st1=st1, B=B, groups=32, eps=eps)
    out2 = ops.gn_apply_table(x0, table, x1=x1, silu=silu)
    assert rel_l2(out2, ref) < 2e-3
    # dispatch: statistics riding on the tensors (one source measured on the fly when only the other carries them)
    x0._vd_stats = st0
    out3 = ops.groupnorm_silu(x0, gamma, beta, x1=x1, groups=32, eps=eps, silu=silu)
    assert rel_l2(out3, ref) < 2e-3
    if ops.GN_STATS and not c1:   # the dispatch took the statistics on the tensor: one of the two forms, same partials
        small = B * HW * C <= ops.GN_FUSED_MAX or ops.GN_FORM == "fused"
        assert torch.equal(out3, out if small else out2)
    # run-to-run identical (no atomics)
    assert torch.equal(out, ops.groupnorm_from_stats(x0, gamma, beta, st0, x1=x1, st1=st1, groups=32, eps=eps, silu=silu))


@pytest.mark.parametrize("HW,C,R", [(4096, 320, 256), (256, 1280, 64), (64, 1280, 64)])
def test_gn_fused_large_mean_small_spread(ops, dev, HW, C, R):
    """|mean| >> sigma with eps = 1e-6: shifted partial sums + Chan's combination must match torch's two-pass statistics
    (the producers' epilogues use the same shifted form, checked against chan_stats in test_gemm_out_stats)."""
    g = torch.Generator().manual_seed(170 + HW)
    x = (16.0 + 0.03 * torch.randn((2, HW, C), generator=g)).half().to(dev)
    gamma = rnd((C,), dev, 0.5, 171) + 1.0
    beta = rnd((C,), dev, 0.5, 172)
    ref = F.group_norm(x.float().permute(0, 2, 1), 32, gamma.float(), beta.float(), 1e-6).permute(0, 2, 1)
    out = ops.groupnorm_from_stats(x, gamma, beta, ops.chan_stats(x, R), groups=32, eps=1e-6, silu=False)
    assert rel_l2(out, ref) < 3e-3


@pytest.mark.parametrize("B,HW,c0,c1,mean,sigma", [(2, 4096, 320, 0, 0.0, 1.0), (2, 1024, 640, 320, 0.0, 1.0), (2, 4096, 320, 0, 16.0, 0.03),
                                                  (3, 1024, 1280, 640, -40.0, 0.05), (1, 9216, 320, 320, 3.0, 2.0)])
def test_gn_apply_sums(ops, dev, B, HW, c0, c1, mean, sigma):
    """vd_gn_apply_sums_f16 against torch's GroupNorm: the per-(sample, channel) sums are accumulated here the way the producers
    do it (partials of chan_stats -> fixed point in fp64), over a skip concat, and with |mean| >> sigma at eps = 1e-6."""
    g = torch.Generator().manual_seed(410 + HW + c1)
    mk = lambda c: (mean + sigma * torch.randn((B, HW, c), generator=g)).half().to(dev)
    x, x1 = mk(c0), (mk(c1) if c1 else None)

    def sums_of(t):
        st = ops.chan_stats(t, 128)   # (mean, M2) over blocks of 128 rows
        R = HW // st.T
        m = st.buf[..., 0].double().view(B, st.T, -1)
        m2 = st.buf[..., 1].double().view(B, st.T, -1)
        s = (R * m * 2.0 ** 32).round().long().sum(1)
        q = ((m2 + R * m * m) * 2.0 ** 16).round().long().sum(1)
        return torch.stack([s, q], -1).view(-1, 2).contiguous()

    C = c0 + c1
    gamma = rnd((C,), dev, 0.5, 411) + 1.0
    beta = rnd((C,), dev, 0.5, 412)
    cat = torch.cat([x, x1], -1) if c1 else x
    eps = 1e-6 if sigma < 0.1 else 1e-5
    ref = F.silu(F.group_norm(cat.float().permute(0, 2, 1), 32, gamma.float(), beta.float(), eps).permute(0, 2, 1))
    got = ops.gn_apply_sums(x, sums_of(x), gamma, beta, x1=x1, sums1=sums_of(x1) if c1 else None, groups=32, eps=eps, silu=True)
    assert rel_l2(got, ref) < 3e-3
    with pytest.raises(ops.VdHipError):
        ops.gn_apply_sums(x, sums_of(x)[:-1], gamma, beta, x1=x1, sums1=sums_of(x1) if c1 else None)


@pytest.mark.parametrize("case", [
    # B, H, W, c0, c1, Co, ksize, stride, ups, rowvec, residual
    (8, 64, 64, 320, 0, 320, 3, 1, 0, True, False),     # halo conv, one block per patch, epilogue statistics
    (8, 64, 64, 64, 64, 320, 3, 1, 0, False, True),
    (8, 32, 32, 128, 0, 640, 3, 1, 0, False, False),    # ... 4 column tiles, no split
    (2, 64, 64, 320, 0, 320, 3, 1, 0, True, False),     # few patches: split over chunks, statistics from the reduce kernel
    (2, 32, 32, 640, 640, 640, 3, 1, 0, True, False),   # halo conv split over chunks: statistics from the reduce kernel
    (4, 16, 16, 1280, 0, 1280, 3, 1, 0, False, True),
    (8, 8, 8, 1280, 0, 1280, 3, 1, 0, True, False),     # 8x8 level: gemm_f16_kernel + deep split-K + reduce
    (2, 16, 16, 640, 0, 640, 3, 1, 1, False, False),    # Upsample conv
    (2, 64, 64, 320, 0, 320, 3, 2, 0, False, False),    # Downsample (stride 2) on gemm_f16_kernel
    (2, 64, 64, 320, 0, 320, 1, 1, 0, False, True),     # SpatialTransformer.proj_out (1x1 + residual)
    (2, 16, 16, 1280, 0, 1280, 1, 1, 0, False, True),
    (8, 8, 8, 1280, 0, 1280, 1, 1, 0, False, True),     # tiles span several 8x8 images: one partial per image
    (1, 96, 96, 320, 0, 320, 3, 1, 0, True, False),     # 768x768 geometry
])
def test_gemm_out_stats(ops, dev, case, monkeypatch):
    """VdGemmDesc.out_stats of every producer of the UNet's data flow: the partials attached to the output describe the
    STORED fp16 tensor (chan_stats of it, same block size), and the output itself is unchanged by the request."""
    monkeypatch.setattr(ops, "GN_SUMS", True)   # (opt-in in the product: VD_GN_SUMS=1)
    from vd_hip.pack import pack_conv_weight
    B, H, W, c0, c1, Co, ks, stride, ups, rv, rs = case
    x = rnd((B, H, W, c0), dev, 1.0, 300)
    x1 = rnd((B, H, W, c1), dev, 1.0, 301) if c1 else None
    wt = rnd((Co, c0 + c1, ks, ks), dev, 0.04, 302)
    b = rnd((Co,), dev, 0.3, 303)
    Ho, Wo = (H << ups) // stride, (W << ups) // stride
    kw = dict(ksize=ks, stride=stride, pad=ks // 2, ups=ups, x1=x1)
    if rv:
        kw.update(rowvec=rnd((B, Co), dev, 0.5, 304), rows_per_batch=Ho * Wo)
    if rs:
        kw.update(res=rnd((B, Ho, Wo, Co), dev, 1.0, 305))
    wp = pack_conv_weight(wt) if ks == 3 else wt.view(Co, c0 + c1).contiguous()
    plain = ops.conv2d_nhwc(x, wp, b, **kw)
    out = ops.conv2d_nhwc(x, wp, b, want_stats=True, **kw)
    assert torch.equal(out, plain) or rel_l2(out, plain) < 1e-6   # (the reduce kernels sum the slabs in the same order)
    st = ops.stats_of(out)
    assert st is not None, "this producer should emit statistics"
    assert st.HW == Ho * Wo and st.C == Co and st.buf.shape == (B * st.T, Co, 2)
    R = Ho * Wo // st.T
    if ks == 3 and stride == 1 and Ho * Wo >= 256 and st.T == Ho * Wo // 256:
        # halo epilogue: a partial is a 256-pixel PATCH of an image (8 x 32 or 16 x 16 pixels), not 256 consecutive rows
        tw = 32 if Wo % 32 == 0 else 16
        th = 256 // tw
        o = out.view(B, Ho // th, th, Wo // tw, tw, Co).permute(0, 1, 3, 2, 4, 5).reshape(B * st.T, 256, Co)
        ref = torch.stack([o.double().mean(1), ((o.double() - o.double().mean(1, keepdim=True)) ** 2).sum(1)], -1).float()
    else:
        ref = _chan_stats_ref(out.view(B, Ho * Wo, Co), B, st.T)
    _stats_close(st, ref, R)
    # and a GroupNorm fed with them equals one that measures the tensor itself
    gamma = rnd((Co,), dev, 0.5, 306) + 1.0
    beta = rnd((Co,), dev, 0.5, 307)
    o3 = out.view(B, Ho * Wo, Co)
    refn = F.silu(F.group_norm(o3.float().permute(0, 2, 1), 32, gamma.float(), beta.float(), 1e-5).permute(0, 2, 1))
    got = ops.groupnorm_from_stats(o3, gamma, beta, st, groups=32, eps=1e-5, silu=True)
    assert rel_l2(got, refn) < 2e-3
    # tensors large enough for the table + apply pair also carry ONE fixed-point (sum, sum of squares) per (sample, channel)
    # (VdGemmDesc.stat_sums): exact up to the fp32 rounding of the partials, and the apply launch that folds them itself agrees
    assert (st.sums is not None) == (ops.GN_SUMS and 2 * B * Ho * Wo * Co > ops.GN_FUSED_MAX)
    if st.sums is not None:
        sm = st.sums.view(B, Co, 2).double()
        o64 = o3.double()
        assert torch.allclose(sm[..., 0] / 2.0 ** 32, o64.sum(1), rtol=1e-5, atol=1e-2)
        assert torch.allclose(sm[..., 1] / 2.0 ** 16, (o64 * o64).sum(1), rtol=1e-5, atol=1e-2)
        got = ops.gn_apply_sums(o3, st.sums, gamma, beta, groups=32, eps=1e-5, silu=True)
        assert rel_l2(got, refn) < 2e-3
        if B * Ho * Wo * Co > ops.GN_FUSED_MAX:   # the path the model takes for a tensor of this size
            assert torch.equal(ops.groupnorm_silu(out, gamma, beta, groups=32, eps=1e-5, silu=True).view_as(got), got)


@pytest.mark.parametrize("case", [
    # B, c0, c1, Co, rowvec, residual
    (8, 1280, 0, 1280, True, False),      # ResBlock in-conv at ds = 8 (bench shape)
    (8, 1280, 1280, 1280, False, True),   # output block: skip concat, residual from the 1x1 skip conv
    (2, 128, 0, 256, False, False),       # one image pair, one column slice
    (4, 64, 64, 512, True, True),
    (6, 192, 0, 256, False, False),       # 3 chunks: ragged split
    (2, 256, 0, 256, True, True),         # 4 chunks: the fewest the whole-K kernel takes
    (4, 192, 128, 512, False, True),      # 5 chunks over two sources
    (8, 2560, 0, 256, True, False),       # 40 chunks: the longest unrolled sequence
])
def test_conv3x3_wstream(ops, dev, case, monkeypatch):
    """vd_conv3x3_wstream_f16 (weights in MFMA-fragment order streamed into registers, halo in LDS, split over chunks +
    reduce) against torch's fp32 convolution and against the same problem on gemm_f16_kernel; every instance, several grid
    targets; statistics of the stored output."""
    from vd_hip.loader import lib
    from vd_hip.pack import pack_conv_weight, pack_conv_weight_stream
    B, c0, c1, Co, rv, rs = case
    x = rnd((B, 8, 8, c0), dev, 1.0, 400)
    x1 = rnd((B, 8, 8, c1), dev, 1.0, 401) if c1 else None
    wt = rnd((Co, c0 + c1, 3, 3), dev, 0.03, 402)
    b = rnd((Co,), dev, 0.3, 403)
    ref = _conv_ref(torch.cat([x, x1], -1) if c1 else x, wt, b, 1, 1, 0)
    kw = dict(ksize=3, pad=1, x1=x1)
    if rv:
        rowvec = rnd((B, Co), dev, 0.5, 404)
        kw.update(rowvec=rowvec, rows_per_batch=64)
        ref = ref + rowvec.float().view(B, 1, 1, Co)
    if rs:
        res = rnd((B, 8, 8, Co), dev, 1.0, 405)
        kw.update(res=res)
        ref = ref + res.float()
    wp, wsm = pack_conv_weight(wt), pack_conv_weight_stream(wt)
    old = ops.conv2d_nhwc(x, wp, b, **kw)
    assert rel_l2(old, ref) < 2e-3
    # round 5: the whole-K kernel (conv_wsk_kernel.h: no split, epilogue + statistics in the kernel) wherever the input has at
    # least 4 chunks (forced here for small grids too)
    monkeypatch.setenv("VD_WSK", "1")
    monkeypatch.setenv("VD_WSK_MIN_BLOCKS", "1")
    outk = ops.conv2d_nhwc(x, wp, b, w_stream=wsm, want_stats=True, **kw)
    assert outk.shape == ref.shape and rel_l2(outk, ref) < 2e-3
    stk = ops.stats_of(outk)
    assert stk is not None and stk.T == 1 and stk.HW == 64
    _stats_close(stk, _chan_stats_ref(outk.view(B, 64, Co), B, 1), 64)
    if kw.get("rowvec") is not None:   # one row vector shared by the whole batch (rows_per_batch = M)
        kw1 = dict(kw, rowvec=kw["rowvec"][:1].contiguous(), rows_per_batch=B * 64)
        o1 = ops.conv2d_nhwc(x, wp, b, w_stream=wsm, **kw1)
        assert rel_l2(o1, ref - kw["rowvec"].float().view(B, 1, 1, Co) + kw["rowvec"][:1].float().view(1, 1, 1, Co)) < 2e-3
    monkeypatch.setenv("VD_WSK", "0")   # the split kernel + reduce launch
    try:
        for var in range(4):
            for target in (256, 64, 1024):
                assert lib().vd_conv3x3_wstream_set_variant(var, target) == 0
                out = ops.conv2d_nhwc(x, wp, b, w_stream=wsm, **kw)
                assert out.shape == ref.shape and rel_l2(out, ref) < 2e-3, (var, target)
    finally:
        lib().vd_conv3x3_wstream_set_variant(0, 256)
    out = ops.conv2d_nhwc(x, wp, b, w_stream=wsm, want_stats=True, **kw)
    st = ops.stats_of(out)
    assert st is not None and st.T == 1 and st.HW == 64
    _stats_close(st, _chan_stats_ref(out.view(B, 64, Co), B, 1), 64)


@pytest.mark.parametrize("case", [
    (8, 1280, 1280, 1280),   # output blocks 0 / 1 of the UNet: 2560 -> 1280 skip convolution, 40 one-tap chunks over 10 splits
    (2, 1280, 1280, 0),      # one image group, single skip source
    (4, 256, 64, 64),        # fewer skip chunks than splits
])
def test_conv3x3_wstream_with_folded_skip_conv(ops, dev, case):
    """The weight-streaming 8x8 convolution with ResBlock's skip 1x1 convolution folded in (fragment-ordered skip weights,
    one-tap chunks behind the 3x3 chunks): against torch fp32 and against the unfused pair, at several grid targets."""
    from vd_hip.loader import lib
    from vd_hip.pack import pack_conv_weight, pack_conv_weight_stream, pack_linear_weight_stream
    B, C, cs0, cs1 = case
    h = rnd((B, 8, 8, C), dev, 1.0, 720)
    s0 = rnd((B, 8, 8, cs0), dev, 1.0, 721)
    s1 = rnd((B, 8, 8, cs1), dev, 1.0, 722) if cs1 else None
    w3 = rnd((C, C, 3, 3), dev, 0.03, 723)
    w1 = rnd((C, cs0 + cs1), dev, 0.03, 724)
    b = rnd((C,), dev, 0.3, 725)
    xs = torch.cat([s0, s1], -1) if cs1 else s0
    ref = _conv_ref(h, w3, b, 1, 1, 0) + (xs.float().reshape(-1, cs0 + cs1) @ w1.float().t()).view(B, 8, 8, C)
    wp, wsm, w1s = pack_conv_weight(w3), pack_conv_weight_stream(w3), pack_linear_weight_stream(w1)
    try:
        for target in (256, 64, 1024):
            assert lib().vd_conv3x3_wstream_set_variant(0, target) == 0
            out = ops.conv2d_nhwc(h, wp, b, ksize=3, pad=1, w_stream=wsm, skip=(s0, s1, w1, w1s), want_stats=True)
            assert out is not None and out.shape == ref.shape and rel_l2(out, ref) < 2e-3, target
    finally:
        lib().vd_conv3x3_wstream_set_variant(0, 256)
    st = ops.stats_of(out)
    assert st is not None and st.T == 1
    _stats_close(st, _chan_stats_ref(out.view(B, 64, C), B, 1), 64)
    res = ops.gemm(s0.view(-1, cs0), w1, a1=s1.view(-1, cs1) if cs1 else None, K=cs0 + cs1, N=C)
    pair = ops.conv2d_nhwc(h, wp, b, ksize=3, pad=1, w_stream=wsm, res=res.view(B, 8, 8, C))
    assert rel_l2(out, pair) < 2e-3
    # without the fragment-ordered skip weights the launch is refused (the caller runs the 1x1 convolution itself)
    assert ops.conv2d_nhwc(h, wp, b, ksize=3, pad=1, w_stream=wsm, skip=(s0, s1, w1)) is None


@pytest.mark.parametrize("M,N,K,split", [(2048, 1280, 5120, 0), (512, 1280, 5120, 0), (128, 256, 64, 0), (256, 512, 2560, 0),
                                         (384, 256, 4096, 2), (128, 768, 3072, 1), (1024, 256, 2112, 0)])
def test_gemm_wstream(ops, dev, M, N, K, split):
    """vd_gemm_wstream_f16 (weights in MFMA-fragment order streamed into registers, activation tile through LDS, split over
    64-deep chunks + reduce) against torch fp32 and against gemm_f16_kernel: bias, residual, activation; one chunk per block up
    to the 32 the kernel unrolls (K = 2112 in one split = 33 chunks: the launcher must split)."""
    from vd_hip.pack import pack_linear_weight_stream
    a = rnd((M, K), dev, 1.0, 740)
    w = rnd((N, K), dev, 0.03, 741)
    b = rnd((N,), dev, 0.3, 742)
    res = rnd((M, N), dev, 1.0, 743)
    ws = pack_linear_weight_stream(w)
    ref = a.float() @ w.float().t() + b.float()
    out = ops.gemm(a, w, bias=b, w_stream=ws, split_k=split)
    assert out.shape == (M, N) and rel_l2(out, ref) < 2e-3
    names = []
    ops.profile_begin()
    out = ops.gemm(a, w, bias=b, res=res, w_stream=ws, split_k=split)
    names = [n for n, _, _, _ in ops.profile_end()]
    assert any("gemm_wstream" in n for n in names), names
    assert rel_l2(out, ref + res.float()) < 2e-3
    assert rel_l2(out, ops.gemm(a, w, bias=b, res=res)) < 2e-3
    out = ops.gemm(a, w, bias=b, res=res, act=ops.ACT_SILU, alpha=0.5, w_stream=ws, split_k=split)
    assert rel_l2(out, F.silu(ref) * 0.5 + res.float()) < 2e-3
    # shapes the kernel does not take fall through to gemm_f16_kernel
    out = ops.gemm(a[:100], w, bias=b, w_stream=ws)
    assert rel_l2(out, ref[:100]) < 2e-3


def test_blocks_are_placed_round_robin_over_the_xcds(ops, dev):
    """Block b of the linearised grid runs on XCD b % 8 (up to a rotation): what the XCD-aware tile orders assume for
    locality, and what the ticketed split of conv3x3_halo_kernel relies on for CORRECTNESS -- with a tile count that is a
    multiple of 8, all splits of a tile (same blockIdx.x, any blockIdx.y) share one XCD and hence one L2."""
    for gx, gy in ((64, 4), (128, 2), (256, 1), (8, 16)):
        ids = ops.probe_xcc_ids(dev, gx, gy).long()
        assert ids.min() >= 0 and ids.max() <= 7 and len(set(ids.flatten().tolist())) == 8
        assert torch.equal(ids, ids[0:1].expand(gy, gx)), "splits of a tile run on different XCDs"
        rot = (ids[0] - torch.arange(gx)) % 8
        assert bool((rot == rot[0]).all()), "placement is not round-robin in block order"


@pytest.mark.parametrize("case", [
    # B, H, C (3x3 input = output width), cs0, cs1
    (8, 64, 320, 640, 320),      # output block at the 64x64 level: one block per patch, fused epilogue
    (8, 32, 640, 1280, 640),     # 32x32 level: split 2, skip chunks shared between the splits
    (8, 16, 1280, 1280, 1280),   # 16x16 level: split 4
    (8, 32, 640, 320, 0),        # input block 4: single skip source (320 -> 640)
    (2, 64, 320, 64, 64),        # few patches: deep split, fewer skip chunks than splits
])
def test_conv3x3_with_folded_skip_conv(ops, dev, case):
    """ResBlock's `skip_connection(cat(x, skip)) + conv2(h)` with the 1x1 convolution folded into the halo-resident 3x3
    convolution as one-tap chunks (VdGemmDesc.skip_*): against torch fp32 (conv3x3 + conv1x1) and the unfused pair."""
    from vd_hip.pack import pack_conv_weight
    B, H, C, cs0, cs1 = case
    h = rnd((B, H, H, C), dev, 1.0, 700)
    s0 = rnd((B, H, H, cs0), dev, 1.0, 701)
    s1 = rnd((B, H, H, cs1), dev, 1.0, 702) if cs1 else None
    w3 = rnd((C, C, 3, 3), dev, 0.03, 703)
    w1 = rnd((C, cs0 + cs1), dev, 0.03, 704)
    b = rnd((C,), dev, 0.3, 705)
    xs = torch.cat([s0, s1], -1) if cs1 else s0
    ref = _conv_ref(h, w3, b, 1, 1, 0) + (xs.float().reshape(-1, cs0 + cs1) @ w1.float().t()).view(B, H, H, C)
    wp = pack_conv_weight(w3)
    out = ops.conv2d_nhwc(h, wp, b, ksize=3, pad=1, skip=(s0, s1, w1), want_stats=True)
    assert out is not None, "the halo kernel should take this launch"
    assert out.shape == ref.shape and rel_l2(out, ref) < 2e-3
    st = ops.stats_of(out)
    assert st is not None
    if st.T * 64 == H * H:
        _stats_close(st, _chan_stats_ref(out.view(B, H * H, C), B, st.T), 64)
    res = ops.gemm(s0.view(-1, cs0), w1, a1=s1.view(-1, cs1) if cs1 else None, K=cs0 + cs1, N=C)
    pair = ops.conv2d_nhwc(h, wp, b, ksize=3, pad=1, res=res.view(B, H, H, C))
    assert rel_l2(out, pair) < 2e-3
    assert torch.equal(out, ops.conv2d_nhwc(h, wp, b, ksize=3, pad=1, skip=(s0, s1, w1)))
    # launches the halo kernel does not take return None without running anything
    assert ops.conv2d_nhwc(h[:, :8, :8].contiguous(), wp, b, ksize=3, pad=1,
                           skip=(s0[:, :8, :8].contiguous(), None if s1 is None else s1[:, :8, :8].contiguous(), w1)) is None
