"""Tensor-level wrappers over the C ABI (include/vd_hip.h).

Inputs are torch CUDA(HIP) fp16 tensors; outputs are allocated with torch (caching allocator) and every
kernel is enqueued on torch's current stream, so `torch.cuda.graph` capture and stream semantics hold.
No arithmetic is done in torch here.
"""
import ctypes
import functools
import os
import threading

import torch

from .loader import VdFfChain, VdGemmDesc, VdHipError, lib

EPI_BIAS, EPI_ROWVEC, EPI_RESIDUAL, EPI_BIAS_ALONG_M, EPI_OUT_F32, EPI_LNFOLD, EPI_LN_INLOOP = 1, 2, 4, 8, 16, 32, 64
EPI_GROUPNORM, EPI_GN_SILU, EPI_LN_SUMS = 128, 256, 512
ACT_NONE, ACT_GEGLU, ACT_QUICK_GELU, ACT_SILU, ACT_GELU_TANH = 0, 1, 2, 3, 4

_ws_cache = {}
_tune_checked = [False]

# ---- optional per-launch instrumentation (bench.py roofline leg; off in the product path) -------------
_prof = None
PROFILE_SHAPES = False


def gemm_kernel_name(cfg):
    """Instantiation name of a planner tile configuration (vd_gemm_config_name)."""
    try:
        n = lib().vd_gemm_config_name(int(cfg))
    except AttributeError:   # older build loaded through VD_HIP_LIB (development A/B)
        return "gemm_f16_kernel<cfg %d>" % int(cfg)
    return n.decode() if n else "gemm_f16_kernel<?>"


def gemm_set_override(cfg):
    """Development hook: force the GEMM tile configuration (-1 = planner)."""
    _check(lib().vd_gemm_set_override(int(cfg)))



def profile_begin():
    """Start recording (kernel name, algorithmic flops, algorithmic bytes, start/stop events) per launch."""
    global _prof
    _prof = []


def profile_end():
    """Stop recording; returns [(name, flops, bytes, ms)] after synchronising."""
    global _prof
    rec, _prof = _prof, None
    torch.cuda.synchronize()
    return [(n, f, b, e0.elapsed_time(e1)) for n, f, b, e0, e1 in rec]


class _Timed(object):
    def __init__(self, name, flops, nbytes):
        self.name, self.flops, self.nbytes = name, flops, nbytes

    def __enter__(self):
        if _prof is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        if _prof is not None:
            self.e1.record()
            _prof.append((self.name, self.flops, self.nbytes, self.e0, self.e1))
        return False


LIB_CALLS = [0]   # calls into the C ABI that enqueue work (every launcher takes the stream exactly once): bench.py reports the count per step


def _stream():
    LIB_CALLS[0] += 1
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check(rc):
    if rc != 0:
        raise VdHipError("vd_hip call failed (%d): %s" % (rc, lib().vd_last_error().decode()))


def _req(t, name, dtype=torch.float16):
    if t is None:
        return
    if not t.is_cuda:
        raise VdHipError("%s must live on the GPU (no CPU fallback in the product path)" % name)
    if dtype is not None and t.dtype != dtype:
        raise VdHipError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise VdHipError("%s must be contiguous" % name)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def workspace(nbytes, device, tag="ws"):
    """Grow-only scratch per (device, stream, tag); consecutive kernels on one stream serialise on it."""
    key = (device.index, torch.cuda.current_stream().cuda_stream, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


SYNC_INTS = 16384   # VD_GEMM_SYNC_INTS
FIXUP_DEFAULT = os.environ.get("VD_GEMM_FIXUP", "0") == "1"


def sync_counters(device):
    """Split-K arrival counters of vd_gemm_f16 (VdGemmDesc.sync): zeroed once, left zero by every launch, one set per
    stream.  Under graph capture the zero fill is captured with the graph's own buffer."""
    key = (device.index, torch.cuda.current_stream().cuda_stream, "gemm_sync")
    buf = _ws_cache.get(key)
    if buf is None:
        buf = torch.zeros(SYNC_INTS, dtype=torch.int32, device=device)
        _ws_cache[key] = buf
    return buf


def drop_workspaces(stream_id):
    """Forget scratch buffers keyed to a (capture) stream; a captured graph keeps its own pool alive."""
    for k in [k for k in _ws_cache if k[1] == stream_id]:
        del _ws_cache[k]


def row_stats(x, C, rows, eps, ldx=None):
    """(mean, rstd) per row of x viewed as [rows][ldx >= C] -> fp32 [rows, 2] (the statistics half of a LayerNorm)."""
    _req(x, "x")
    # (one spare row: the LayerNorm-fold epilogue fetches 16 bytes per row, i.e. 8 bytes past the last (mean, rstd) pair)
    stats = torch.empty((rows + 1, 2), dtype=torch.float32, device=x.device)[:rows]
    with _Timed("row_stats_kernel", 0.0, rows * C * 2.0 + rows * 8.0):
        _check(lib().vd_row_stats_f16(_ptr(x), _ptr(stats), int(rows), int(C), int(ldx or C), float(eps), _stream()))
    return stats


class ChanStats(object):
    """Per-channel partial statistics of a tensor, emitted by its producer (VdGemmDesc.out_stats) or by chan_stats():
    buf fp32 [B * T, C, 2] = (mean, M2) over T blocks of HW / T rows per sample.  Travels as the attribute `_vd_stats` of the
    tensor it describes; views / copies do not carry it (the consumer then measures the tensor itself)."""
    __slots__ = ("buf", "T", "C", "HW", "sums")

    def __init__(self, buf, T, C, HW, sums=None):
        self.buf, self.T, self.C, self.HW = buf, int(T), int(C), int(HW)
        self.sums = sums   # int64 [B * C, 2]: per-(sample, channel) fixed-point sums of the same values (VdGemmDesc.stat_sums), or None


def stats_of(t):
    """ChanStats attached to tensor `t` by its producer, or None.  Statistics that do not describe `t` (stale attribute on a
    re-used or reshaped tensor: other channel count, rows per sample or batch) are dropped -- the consumer then measures `t`."""
    st = getattr(t, "_vd_stats", None) if t is not None else None
    if st is None:
        return None
    C = t.shape[-1]
    rows = t.numel() // max(C, 1)   # (a producer may hand the tensor over as [rows, C]: samples x rows per sample is what counts)
    if st.C != C or st.T <= 0 or (st.buf.shape[0] // st.T) * st.HW != rows or st.buf.device != t.device:
        return None
    return st


def repeat_batch(t, repeat):
    """t.repeat(repeat, 1, ...) on the batch axis, statistics included (the CFG replicas of run_unet)."""
    out = t.repeat(repeat, *([1] * (t.dim() - 1)))
    st = stats_of(t)
    if st is not None:
        out._vd_stats = ChanStats(st.buf.repeat(repeat, 1, 1), st.T, st.C, st.HW,
                                  st.sums.repeat(repeat, 1) if st.sums is not None else None)
    return out


def gemm(a0, w, *, a1=None, bias=None, rowvec=None, rows_per_batch=0, res=None, out=None, M=None, N=None, K=None,
         conv=None, act=ACT_NONE, alpha=1.0, out_f32=False, bias_along_m=False, batch=1, strides=(0, 0, 0, 0),
         lda0=0, lda1=0, ldw=0, ldc=0, ldr=0, c0=0, c1=0, split_k=0, out_shape=None, colsum=None, ln_eps=0.0, fixup=None,
         want_stats=False, stat_img_rows=0, w_stream=None, skip=None, row_sums=None, ln_sums=None):
    """out = epilogue(A @ W^T); see VdGemmDesc in include/vd_hip.h.

    conv = dict(Hin, Win, Hout, Wout, ksize, stride, pad, ups) selects the implicit-GEMM gather.
    colsum (fp32 [N]) + ln_eps: the rows of A are layer-normalised on the fly (VD_EPI_LNFOLD: row statistics from
    vd_row_stats_f16, applied in the GEMM epilogue); w / bias are then the folded gamma (*) W and beta W^T + bias
    (hip_layers.fold_layernorm).
    fixup=True: split-K slabs are summed by the last-arriving block of each tile (VdGemmDesc.sync) instead of the reduce
    kernel; default from VD_GEMM_FIXUP (off: the in-kernel tail measured 0.2 ms per UNet forward slower than the launch).
    want_stats=True: the producing epilogue (or the split-K reduce) also emits per-channel partial statistics of the stored
    output for a consuming GroupNorm (VdGemmDesc.out_stats); they come back as `out._vd_stats` (ChanStats) when the planned
    launch can emit them, else the attribute is absent.  stat_img_rows: rows of one sample for plain matrices (conv: Hout*Wout).
    w_stream: the same conv weights in MFMA-fragment order (pack.pack_conv_weight_stream); 3x3 convolutions on 8x8 images then
    run on the weight-streaming kernel (vd_conv3x3_wstream_f16) where its geometry fits.
    skip = (s0, s1 or None, w_skip [N, C_s0 + C_s1]): a 1x1 convolution of cat(s0, s1) on the output grid folded into this 3x3
    convolution as extra K (ResBlock's skip_connection(x) + h; its bias belongs into `bias`).  Returns None WITHOUT launching
    when the planned launch cannot take it (vd_gemm_skip_ok): the caller then runs the 1x1 convolution itself.
    row_sums: zeroed int64 [M, 2] (rowsum_take; fixed point, see VdGemmDesc.row_sums): the epilogue adds (sum, sum of squares) of every stored row -- the statistics of
    the LayerNorm folded into the NEXT projection; comes back as `out._vd_rowsums` when the planned launch can accumulate them
    (vd_gemm_row_sums_ok), else the attribute is absent and the consumer runs vd_row_stats_f16.
    ln_sums: with colsum, such a buffer describing the rows of a0 (VD_EPI_LN_SUMS) instead of the row_stats launch.
    """
    _req(a0, "a0"); _req(a1, "a1"); _req(w, "w"); _req(bias, "bias"); _req(rowvec, "rowvec"); _req(res, "res")
    _req(colsum, "colsum", torch.float32); _req(row_sums, "row_sums", torch.int64); _req(ln_sums, "ln_sums", torch.int64)
    if out is not None:   # a re-used output tensor must not keep the statistics of what it held before
        for attr in ("_vd_stats", "_vd_rowsums"):
            if hasattr(out, attr):
                delattr(out, attr)
    # (the plain N = 320 projections measure equal on both kernels -- 24.9 vs 24.2 us: with one 128-row block per CU in
    # lock-step a launch is its load / store phases either way; the LayerNorm-folded q | k | v projection is 59 vs 66 us + the
    # statistics launch)
    if ROW320 and colsum is not None and _row320_ok(a0, w, a1, rowvec, res, out, M, N, K, conv, act, alpha, out_f32, bias_along_m, batch, lda0, ldw, ldc, ldr, split_k,
                                                    c0, c1, strides, fixup, rows_per_batch):
        return gemm_row320(a0, w, bias, res, True, ln_eps, out_shape)
    if not _tune_checked[0]:     # shipped tuned launch table (vd_hip/tune.py), installed on first use
        _tune_checked[0] = True
        from . import tune
        tune.ensure_loaded()
    d = VdGemmDesc()
    if K is None:
        K = w.shape[-1]
    if N is None:
        N = w.shape[-2]
    if M is None:
        M = a0.numel() // a0.shape[-1] // max(batch, 1) if conv is None else None
    if conv is not None:
        d.Hin, d.Win, d.Hout, d.Wout = conv["Hin"], conv["Win"], conv["Hout"], conv["Wout"]
        d.ksize, d.stride, d.pad, d.ups = conv.get("ksize", 1), conv.get("stride", 1), conv.get("pad", 0), conv.get("ups", 0)
        if M is None:
            M = conv["B"] * d.Hout * d.Wout
    d.M, d.N, d.K = int(M), int(N), int(K)
    d.c0 = int(c0) if c0 else int(a0.shape[-1])
    d.c1 = int(c1) if c1 else (int(a1.shape[-1]) if a1 is not None else 0)
    d.lda0, d.lda1, d.ldw, d.ldc, d.ldr = int(lda0), int(lda1), int(ldw), int(ldc), int(ldr)
    n_out = N // 2 if act == ACT_GEGLU else N
    if out is None:
        shape = out_shape if out_shape is not None else ((batch, M, n_out) if batch > 1 else (M, n_out))
        out = torch.empty(shape, dtype=torch.float32 if out_f32 else torch.float16, device=a0.device)
    else:
        _req(out, "out", torch.float32 if out_f32 else torch.float16)
    flags = 0
    if bias is not None:
        flags |= EPI_BIAS
    if bias_along_m:
        flags |= EPI_BIAS_ALONG_M
    if rowvec is not None:
        flags |= EPI_ROWVEC
        d.rows_per_batch = int(rows_per_batch)
    if res is not None:
        flags |= EPI_RESIDUAL
    if out_f32:
        flags |= EPI_OUT_F32
    ln_stats = None
    if colsum is not None:
        if conv is not None or a1 is not None or max(batch, 1) != 1:
            raise VdHipError("gemm: the LayerNorm fold takes a plain single-source, unbatched A")
        flags |= EPI_LNFOLD
        d.colsum, d.ln_eps = colsum.data_ptr(), float(ln_eps)
        if ln_sums is not None and LN_SUMS:   # (sum, sum of squares) of the rows of a0, accumulated by the launch that stored a0
            if ln_sums.numel() != 2 * int(M):
                raise VdHipError("gemm: ln_sums does not describe the %d rows of a0" % int(M))
            flags |= EPI_LN_SUMS
            d.ln_stats = ln_sums.data_ptr()
        elif not LN_INLOOP:   # two-pass statistics from their own launch instead of the K loop's running sums
            ln_stats = row_stats(a0, int(K), int(M), float(ln_eps), ldx=int(lda0) if lda0 else int(K))
            d.ln_stats = ln_stats.data_ptr()
        else:
            flags |= EPI_LN_INLOOP
    d.flags, d.act, d.alpha = flags, int(act), float(alpha)
    d.batch, d.split_k = int(batch), int(split_k)
    d.stride_a, d.stride_w, d.stride_out, d.stride_res = [int(s) for s in strides]
    d.a0, d.a1, d.w = a0.data_ptr(), (a1.data_ptr() if a1 is not None else None), w.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.rowvec = rowvec.data_ptr() if rowvec is not None else None
    d.res = res.data_ptr() if res is not None else None
    d.out = out.data_ptr()
    skip_stream = False
    if skip is not None:
        s0, s1, wsk = skip[:3]
        wsk_stream = skip[3] if len(skip) > 3 else None   # the same weights in fragment order (weight-streaming conv)
        _req(s0, "skip s0"); _req(s1, "skip s1"); _req(wsk, "skip w"); _req(wsk_stream, "skip w (stream)")
        if not SKIP_FOLD or res is not None or conv is None or s0.shape[-1] % 64 or (s1 is not None and s1.shape[-1] % 64):
            return None
        d.skip_c0, d.skip_c1 = int(s0.shape[-1]), (int(s1.shape[-1]) if s1 is not None else 0)
        if int(wsk.shape[-1]) != d.skip_c0 + d.skip_c1:
            return None
        if (w_stream is not None and wsk_stream is not None and WSTREAM and colsum is None
                and lib().vd_conv3x3_wstream_supported(ctypes.byref(d))):
            skip_stream = True   # 8x8 level: one-tap chunks behind the weight stream (fields set below, on that path)
        else:
            d.skip_a0, d.skip_a1, d.skip_w = s0.data_ptr(), (s1.data_ptr() if s1 is not None else None), wsk.data_ptr()
            d.skip_lda0, d.skip_lda1, d.skip_ldw = d.skip_c0, d.skip_c1, int(wsk.shape[-1])
            if not lib().vd_gemm_skip_ok(ctypes.byref(d)):
                return None
    if w_stream is not None and WSTREAM and colsum is None and (skip is None or skip_stream) and lib().vd_conv3x3_wstream_supported(ctypes.byref(d)):
        _req(w_stream, "w_stream")
        if skip_stream:
            d.skip_a0, d.skip_a1, d.skip_w = s0.data_ptr(), (s1.data_ptr() if s1 is not None else None), wsk_stream.data_ptr()
            d.skip_lda0, d.skip_lda1 = d.skip_c0, d.skip_c1
        d.split_k = int(split_k)
        # the launcher's own split (0: the whole-K kernel, no slabs at all) sizes the workspace -- not the 32-slab upper bound
        pns = ctypes.c_int(0)
        _check(lib().vd_conv3x3_wstream_plan(ctypes.byref(d), ctypes.byref(pns)))
        if pns.value > 0:
            d.split_k = pns.value
            d.ws = workspace(lib().vd_gemm_workspace_bytes(ctypes.byref(d)), a0.device, "gemm").data_ptr()
        stats = None
        if want_stats:
            sbuf = torch.empty((int(M) // 64, n_out, 2), dtype=torch.float32, device=a0.device)
            d.out_stats = sbuf.data_ptr()
            stats = ChanStats(sbuf, 1, n_out, 64)
        extra = (float(M) * n_out if res is not None else 0.0) + (float(rowvec.numel()) if rowvec is not None else 0.0)
        nm = "conv3x3_wstream_kernel + reduce" if pns.value > 0 else "conv3x3_wsk_kernel"
        if PROFILE_SHAPES:
            nm += " M=%d N=%d K=%d" % (M, N, K)
        with _Timed(nm, 2.0 * M * N * K, 2.0 * (float(M) * (d.c0 + d.c1) + float(N) * K + float(M) * n_out + extra)):
            _check(lib().vd_conv3x3_wstream_f16(ctypes.byref(d), _ptr(w_stream), _stream()))
        if stats is not None:
            out._vd_stats = stats
        return out
    if (w_stream is not None and GEMM_WSTREAM and conv is None and a1 is None and colsum is None and rowvec is None and skip is None
            and not want_stats and max(batch, 1) == 1 and lib().vd_gemm_wstream_supported(ctypes.byref(d))):
        # long-K, small-M projection: weights in fragment order straight into registers (gemm_wstream_kernel.h) + reduce
        _req(w_stream, "w_stream")
        d.split_k = int(split_k)
        pns = ctypes.c_int(0)
        _check(lib().vd_gemm_wstream_plan(ctypes.byref(d), ctypes.byref(pns)))
        d.split_k = max(pns.value, 1)   # the launcher's own split sizes the workspace (not the 32-slab upper bound)
        d.ws = workspace(lib().vd_gemm_workspace_bytes(ctypes.byref(d)), a0.device, "gemm").data_ptr()
        stats = None
        nm = "gemm_wstream_kernel + reduce"
        if PROFILE_SHAPES:
            nm += " M=%d N=%d K=%d" % (M, N, K)
        extra = float(M) * n_out if res is not None else 0.0
        with _Timed(nm, 2.0 * M * N * K, 2.0 * (float(M) * K + float(N) * K + float(M) * n_out + extra)):
            _check(lib().vd_gemm_wstream_f16(ctypes.byref(d), _ptr(w_stream), _stream()))
        if stats is not None:
            out._vd_stats = stats
        return out
    # split-K (fp32 slabs + reduce) is the library's answer to small-M / deep-K problems: ask its planner first so
    # the workspace is sized for the split factor it will actually use
    name, d.ws = "gemm", None
    plan_cfg, plan_ns = ctypes.c_int(0), ctypes.c_int(1)
    if act != ACT_GEGLU and not out_f32 and colsum is None and (split_k > 1 or (M * N <= 384 * 128 * 128 and K >= 1024)):
        d.ws = 1  # any non-null value: planning only
        _check(lib().vd_gemm_plan(ctypes.byref(d), ctypes.byref(plan_cfg), ctypes.byref(plan_ns)))
        d.ws = None
        if plan_ns.value > 1:
            d.split_k = plan_ns.value
            # the last-arriver fix-up of gemm_f16_kernel is opt-in: the reduce launch measured faster (the halo conv always takes it)
            halo = plan_cfg.value >= lib().vd_gemm_num_configs()
            use_sync = (not halo) and (FIXUP_DEFAULT if fixup is None else fixup)
            d.sync = sync_counters(a0.device).data_ptr() if use_sync else None
            d.ws = workspace(lib().vd_gemm_workspace_bytes(ctypes.byref(d)), a0.device, "gemm").data_ptr()
    stats = None
    if want_stats and not out_f32 and act != ACT_GEGLU and colsum is None and max(batch, 1) == 1:
        d.stat_img_rows = int(stat_img_rows)
        rows = ctypes.c_int(0)
        _check(lib().vd_gemm_stat_rows(ctypes.byref(d), ctypes.byref(rows)))
        if rows.value > 0:
            hw = int(stat_img_rows) if stat_img_rows else (int(d.Hout) * int(d.Wout) if conv is not None else int(M))
            sbuf = torch.empty((int(M) // rows.value, n_out, 2), dtype=torch.float32, device=a0.device)
            d.out_stats = sbuf.data_ptr()
            stats = ChanStats(sbuf, hw // rows.value, n_out, hw, gn_sums_take(int(M) // hw, n_out, hw, a0.device))
            if stats.sums is not None:
                d.stat_img_rows = hw
                d.stat_sums = stats.sums.data_ptr()
    rs_on = False
    if row_sums is not None and LN_SUMS and max(batch, 1) == 1 and row_sums.numel() == 2 * int(M):
        if lib().vd_gemm_row_sums_ok(ctypes.byref(d)):
            d.row_sums = row_sums.data_ptr()
            rs_on = True
    if _prof is not None:
        _check(lib().vd_gemm_plan(ctypes.byref(d), ctypes.byref(plan_cfg), ctypes.byref(plan_ns)))
        name = gemm_kernel_name(plan_cfg.value)
        if PROFILE_SHAPES:   # cls: epilogue class of the tuned launch table (bit 0 GEGLU, 1 LayerNorm fold, 2 two-source A)
            cls = (1 if act == ACT_GEGLU else 0) | (2 if colsum is not None else 0) | (4 if a1 is not None else 0)
            name += " M=%d N=%d K=%d ks=%d cls=%d split=%d" % (M, N, K, max(int(d.ksize), 1), cls, plan_ns.value)
    nb = max(batch, 1)
    # algorithmic bytes = every operand ONCE: a convolution reads its input pixels (B * Hin * Win * (c0 + c1)), not the
    # M x K im2col matrix (9x that for a 3x3)
    if conv is not None:
        a_elems = float(conv["B"]) * conv["Hin"] * conv["Win"] * (d.c0 + d.c1)
    else:
        a_elems = float(M) * K
    extra = (float(M) * n_out if res is not None else 0.0) + (float(rowvec.numel()) if rowvec is not None else 0.0)
    with _Timed(name, 2.0 * nb * M * N * K, 2.0 * (nb * (a_elems + float(N) * K + float(M) * n_out) + extra)):
        _check(lib().vd_gemm_f16(ctypes.byref(d), _stream()))
    if stats is not None:
        out._vd_stats = stats
    if rs_on:
        out._vd_rowsums = row_sums
    return out


# LayerNorm row statistics from the producer (round 5): the GEMM that STORES the rows a folded LayerNorm will normalise adds
# (sum, sum of squares) per row into a zeroed int64 [rows, 2] fixed-point buffer (VdGemmDesc.row_sums), the consumer reads it with
# VD_EPI_LN_SUMS -- the 22 vd_row_stats_f16 launches of a UNet forward (and their extra read of x) disappear.  The buffers of
# one forward are slices of ONE arena zeroed by one fill (RowSumArena, begun by vd.run_unet); VD_LN_SUMS=0: row_stats launches.
LN_SUMS = os.environ.get("VD_LN_SUMS", "1") != "0"
_tls = threading.local()


class RowSumArena(object):
    """Bump allocator over one zeroed int64 [need, 2] tensor per forward; `need` is what the previous forward of this owner
    took (the first forward, and any request beyond it, falls back to a torch.zeros of its own)."""

    def __init__(self, need=0):
        self.need, self.used, self.buf = int(need), 0, None

    def begin(self, device):
        self.used = 0
        self.buf = torch.zeros((self.need, 2), dtype=torch.int64, device=device) if self.need > 0 else None
        _tls.arena = self

    def take(self, rows, device):
        rows = int(rows)
        start, self.used = self.used, self.used + rows
        if self.buf is not None and self.used <= self.buf.shape[0] and self.buf.device == device:
            return self.buf[start:start + rows]
        return torch.zeros((rows, 2), dtype=torch.int64, device=device)

    def end(self):
        self.need, self.buf = self.used, None
        _tls.arena = None


# GroupNorm statistics as ONE pair per (sample, channel) (round 5, VdGemmDesc.stat_sums): producers of tensors large enough for the
# table + apply pair add their partials into int64 fixed-point sums from the same zeroed arena, and the apply launch folds them itself
# (vd_gn_apply_sums_f16) -- no vd_gn_table_f32 launch.  VD_GN_SUMS=1 switches it on; default: the round-4 pair.
# Measured (five in-session pairs, tools/probes/gn_sums_ab.sh): 27 launches fewer per forward and +0.01 .. +0.04 ms -- the atomics cost the
# producers 0.045 ms and every apply block re-reads its image's 16 bytes per channel, which eats the table launches it saves.  Opt-in.
GN_SUMS = os.environ.get("VD_GN_SUMS", "0") != "0"
GN_SUMS_USE = os.environ.get("VD_GN_SUMS", "0") != "emit"   # development switch: producers accumulate, the consumers keep the table pair


def gn_sums_take(B, C, HW, device):
    """Zeroed int64 [B * C, 2] for VdGemmDesc.stat_sums when a GroupNorm over this tensor (alone, or as one half of a skip concat)
    would take the table + apply pair, else None."""
    if not GN_SUMS or 2 * B * HW * C <= GN_FUSED_MAX or GN_FORM == "fused":
        return None
    arena = getattr(_tls, "arena", None)
    if arena is not None:
        return arena.take(B * C, device)
    return torch.zeros((B * C, 2), dtype=torch.int64, device=device)


def rowsum_take(rows, device):
    """Zeroed int64 [rows, 2] for VdGemmDesc.row_sums, or None when producer row statistics are switched off."""
    if not LN_SUMS:
        return None
    arena = getattr(_tls, "arena", None)
    if arena is not None:
        return arena.take(rows, device)
    return torch.zeros((int(rows), 2), dtype=torch.int64, device=device)


SKIP_FOLD = os.environ.get("VD_SKIP_FOLD", "1") != "0"   # ResBlock skip 1x1 convolution as extra K of the second 3x3 conv
GEMM_WSTREAM = os.environ.get("VD_GEMM_WSTREAM", "1") != "0"   # long-K small-M Linear layers on gemm_wstream_kernel (FeedForward out at 16x16 / 8x8)
WSTREAM = os.environ.get("VD_WSTREAM", "1") != "0"   # development switch: 0 = the 8x8-level 3x3 convolutions stay on gemm_f16_kernel
ROW320 = os.environ.get("VD_GEMM_ROW320", "1") != "0"   # development switch: 0 = the K = 320 projections stay on gemm_f16_kernel


def _row320_ok(a0, w, a1, rowvec, res, out, M, N, K, conv, act, alpha, out_f32, bias_along_m, batch, lda0, ldw, ldc, ldr, split_k,
               c0=0, c1=0, strides=(0, 0, 0, 0), fixup=None, rows_per_batch=0):
    """vd_gemm_row320_f16 takes the plain K = 320 projections whose row blocks fill the chip (the UNet's 64x64 level).
    Every argument of gemm() the row-resident kernel has no notion of must be at its neutral value."""
    if a1 is not None or rowvec is not None or out is not None or out_f32 or bias_along_m or max(batch, 1) != 1:
        return False
    if int(c0) not in (0, 320) or int(c1) != 0 or any(int(v) != 0 for v in strides) or fixup or int(rows_per_batch) != 0:
        return False
    if act != ACT_NONE or alpha != 1.0 or split_k > 1 or w.dim() != 2 or not w.is_contiguous() or not a0.is_contiguous():
        return False
    if conv is not None:
        if conv.get("ksize", 1) != 1 or conv.get("stride", 1) != 1 or conv.get("pad", 0) != 0 or conv.get("ups", 0) != 0:
            return False
        if conv["Hin"] != conv["Hout"] or conv["Win"] != conv["Wout"]:
            return False
    k = w.shape[1] if K is None else K
    n = w.shape[0] if N is None else N
    if k != 320 or a0.shape[-1] != 320 or w.shape[1] != 320 or n != w.shape[0] or n % 320 != 0:
        return False
    m = a0.numel() // 320
    if M is not None and int(M) != m:
        return False
    if any(int(v) not in (0, e) for v, e in ((lda0, 320), (ldw, 320), (ldc, n), (ldr, n))):
        return False
    if res is not None and (not res.is_contiguous() or res.numel() != m * n):
        return False
    # one 128-row block per CU: worth it when the row blocks cover at least 3/4 of the device's CUs
    cus = torch.cuda.get_device_properties(a0.device).multi_processor_count if a0.is_cuda else 256
    return ((m + 127) // 128) * (n // 320) >= (3 * cus) // 4


def gemm_row320(a0, w, bias=None, res=None, layernorm=False, ln_eps=1e-5, out_shape=None):
    """[LayerNorm](a0 [M, 320]) @ w [N, 320]^T + bias (+ res) with the rows of a0 resident in registers (vd_gemm_row320_f16);
    with layernorm, w / bias are the folded operands (hip_layers.fold_layernorm)."""
    for t, nm in ((a0, "a0"), (w, "w"), (bias, "bias"), (res, "res")):
        _req(t, nm)
    n = w.shape[0]
    m = a0.numel() // 320
    out = torch.empty(out_shape if out_shape is not None else (m, n), dtype=torch.float16, device=a0.device)
    name = "rowgemm320_kernel" + ((" M=%d N=%d ln=%d" % (m, n, int(layernorm))) if PROFILE_SHAPES else "")
    with _Timed(name, 2.0 * m * n * 320, 2.0 * (m * 320 + n * 320 + m * n * (2 if res is not None else 1))):
        _check(lib().vd_gemm_row320_f16(_ptr(a0), _ptr(w), _ptr(bias), _ptr(res), _ptr(out), m, n, 1 if layernorm else 0,
                                        float(ln_eps), _stream()))
    return out


# opt-in (VD_ST_CHAIN=1): GroupNorm (as an affine map) -> proj_in -> LayerNorm -> q|k|v of a 64x64-level SpatialTransformer in one
# launch + a statistics launch.  Correct (test_row320_chain), measured neutral: 54 + 18 us against 23 + 24 + 39 us for the three
# launches it replaces, forward 11.18 vs 11.21 ms (tools/gpu_r03_ad.sh).
ST_CHAIN = os.environ.get("VD_ST_CHAIN", "1") != "0"   # round 4: on -- with the GroupNorm statistics coming from the producer the chained entry (GroupNorm affine -> proj_in -> LayerNorm -> q|k|v in one launch) saves the apply pass (measured -0.03 .. -0.09 ms per forward)


def st_chain_supported(B, HW, C, inner):
    """True when vd_gemm_row320_chain_f16 can take the entry of a SpatialTransformer (width 320, row blocks cover at least half
    of the device's CUs: at CFG batch 4 -- BASELINE configs[3]'s per-GPU share -- the one launch on 128 CUs still beats the
    gn_from_stats + proj_in + q|k|v launches it replaces, 14.5 + 19.6 + 36.4 us in profiles/r05_dual_per_shape.txt)."""
    if not (ST_CHAIN and ROW320 and C == 320 and inner == 320 and HW % 128 == 0):
        return False
    cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count if torch.cuda.is_available() else 256
    return B * HW // 128 >= cus // 2


ST_CENTER = os.environ.get("VD_ST_CENTER", "1") != "0"   # development switch: 0 = the plain fp16 map x * scale + shift (round 4)


def groupnorm_affine(x, gamma, beta, *, groups=32, eps=1e-5, centered=False):
    """GroupNorm of channels-last x [B, ..., C] as a per-(sample, channel) affine map -> (scale, shift) fp16 [B, C].
    centered=True -> (scale, shift, center): the map is (x - center) * scale + shift with center = fp16(group mean), which keeps
    the fp16 operands O(1) whatever |mean| / sigma is (center is None where the statistics have to be measured from x: plain form)."""
    _req(x, "x"); _req(gamma, "gamma"); _req(beta, "beta")
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    sc = torch.empty((B, C), dtype=torch.float16, device=x.device)
    sh = torch.empty((B, C), dtype=torch.float16, device=x.device)
    st = stats_of(x) if GN_STATS else None
    if st is not None and C % groups == 0:   # statistics from the producer: no pass over x
        ct = torch.empty((B, C), dtype=torch.float16, device=x.device) if (centered and ST_CENTER) else None
        with _Timed("gn_table_kernel", 0.0, 0.0):
            _check(lib().vd_gn_affine_from_stats_f16(_ptr(st.buf), st.T, st.C, None, 0, 0, B, HW, _ptr(gamma), _ptr(beta), groups,
                                                     float(eps), _ptr(sc), _ptr(sh), _ptr(ct), _stream()))
        return (sc, sh, ct) if centered else (sc, sh)
    ws = workspace(lib().vd_groupnorm_workspace_bytes(B, HW, C, groups), x.device, "gn")
    with _Timed("groupnorm statistics -> affine", 0.0, 2.0 * B * HW * C):
        _check(lib().vd_groupnorm_affine_f16(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(sc), _ptr(sh), _ptr(ws), B, HW, C, groups,
                                             float(eps), _stream()))
    return (sc, sh, None) if centered else (sc, sh)


def row320_chain(x, sc, sh, rows_per_image, w1, b1, w2, b2, ln_eps, center=None):
    """h = (x * sc[img] + sh[img]) @ w1^T + b1;  y2 = LayerNorm(h) @ w2^T + b2 (w2 / b2 LayerNorm-folded) in one launch
    (vd_gemm_row320_chain_f16).  x [..., 320] contiguous -> (h [..., 320], y2 [..., N2]).  center (fp16 [images, 320], from
    groupnorm_affine(centered=True)): the map is applied as (x - center) * sc + sh."""
    for t, n in ((x, "x"), (sc, "sc"), (sh, "sh"), (w1, "w1"), (b1, "b1"), (w2, "w2"), (b2, "b2"), (center, "center")):
        _req(t, n)
    M = x.numel() // 320
    N2 = w2.shape[0]
    h = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    y2 = torch.empty(x.shape[:-1] + (N2,), dtype=torch.float16, device=x.device)
    name = "rowchain320_kernel" + ((" M=%d N2=%d" % (M, N2)) if PROFILE_SHAPES else "")
    with _Timed(name, 2.0 * M * 320 * (320 + N2), 2.0 * (2 * M * 320 + M * N2 + 320 * (320 + N2))):
        _check(lib().vd_gemm_row320_chain_f16(_ptr(x), _ptr(sc), _ptr(sh), _ptr(center), int(rows_per_image), _ptr(w1), _ptr(b1), _ptr(h),
                                              _ptr(w2), _ptr(b2), _ptr(y2), M, N2, float(ln_eps), _stream()))
    return h, y2


LN_INLOOP = os.environ.get("VD_LN_INLOOP", "0") == "1"   # 1 = LN statistics inside the K loop of the folded GEMM (measured neutral: +3..7 us per GEMM = the vd_row_stats_f16 launches it removes; the two-pass statistics stay the default)
FF_FUSED = os.environ.get("VD_FF_FUSED", "1") != "0"   # development switch: 0 = always the three-launch chain


def ff_geglu_supported(C):
    """True when vd_ff_geglu_f16 is instantiated for inner width C (and not switched off)."""
    return FF_FUSED and bool(lib().vd_ff_geglu_supported(int(C)))


def ff_geglu(x, w1_packed, b1_packed, w2, b2, res, ln_eps):
    """res + (v * gelu(g)) @ w2^T + b2 with [v | g] = LayerNorm(x) @ W1^T + b1 in one launch (vd_ff_geglu_f16);
    w1_packed / b1_packed: gamma / beta folded, GEGLU-packed (hip_layers.fold_layernorm + pack.pack_geglu)."""
    for t, n in ((x, "x"), (w1_packed, "w1"), (b1_packed, "b1"), (w2, "w2"), (b2, "b2"), (res, "res")):
        _req(t, n)
    C = x.shape[-1]
    M = x.numel() // C
    if tuple(w1_packed.shape) != (8 * C, C) or tuple(w2.shape) != (C, 4 * C) or res.shape != x.shape:
        raise VdHipError("ff_geglu: operand shapes do not match C=%d" % C)
    y = torch.empty_like(x)
    flops = 2.0 * M * C * 8 * C + 2.0 * M * 4 * C * C
    with _Timed("ff_geglu_kernel" + ((" M=%d C=%d" % (M, C)) if PROFILE_SHAPES else ""), flops, 2.0 * (3 * M * C + 12 * C * C)):
        _check(lib().vd_ff_geglu_f16(_ptr(x), _ptr(w1_packed), _ptr(b1_packed), _ptr(w2), _ptr(b2), _ptr(res), _ptr(y), M, C,
                                     float(ln_eps), _stream()))
    return y


# round 5: the C x C projections on either side of the 64x64-level feed-forward in the feed-forward's launch (csrc/ff_chain.hip).
# VD_FF_CHAIN=0: separate launches; =pre / =post: only that projection is folded (A/B runs)
FF_CHAIN = os.environ.get("VD_FF_CHAIN", "1")


def ff_chain_supported(C, which="both"):
    """True when vd_ff_chain_f16 is instantiated for inner width C and `which` ('pre' / 'post') is not switched off."""
    if FF_CHAIN == "0" or not FF_FUSED or (FF_CHAIN in ("pre", "post") and which != FF_CHAIN):
        return False
    return bool(lib().vd_ff_chain_supported(int(C)))


def ff_chain(x, w1_packed, b1_packed, w2, b2, ln_eps, *, a=None, wo=None, bo=None, wp=None, bp=None, res=None, alpha=1.0,
             want_stats=False, stat_img_rows=0):
    """x1 = a wo^T + bo + x (with a);  y = x1 + FF(LayerNorm(x1));  out = alpha (y wp^T + bp) + res (with wp) in one launch
    (vd_ff_chain_f16, C = 320).  Returns out (or y); with want_stats and wp the per-channel statistics of the stored output
    ride on it as `_vd_stats` (partials of 128 rows)."""
    for t, n in ((x, "x"), (w1_packed, "w1"), (b1_packed, "b1"), (w2, "w2"), (b2, "b2"), (a, "a"), (wo, "wo"), (bo, "bo"), (wp, "wp"),
                 (bp, "bp"), (res, "res")):
        _req(t, n)
    C = x.shape[-1]
    M = x.numel() // C
    if tuple(w1_packed.shape) != (8 * C, C) or tuple(w2.shape) != (C, 4 * C):
        raise VdHipError("ff_chain: feed-forward operand shapes do not match C=%d" % C)
    if a is not None and (a.shape != x.shape or tuple(wo.shape) != (C, C) or bo is None):
        raise VdHipError("ff_chain: first projection operands do not match x")
    if wp is not None and (tuple(wp.shape) != (C, C) or bp is None or res is None or res.numel() != x.numel()):
        raise VdHipError("ff_chain: last projection operands do not match x")
    d = VdFfChain()
    d.x, d.w1_packed, d.b1_packed, d.w2, d.b2 = x.data_ptr(), w1_packed.data_ptr(), b1_packed.data_ptr(), w2.data_ptr(), b2.data_ptr()
    out = torch.empty_like(x)
    d.out, d.M, d.C, d.ln_eps, d.alpha = out.data_ptr(), int(M), int(C), float(ln_eps), float(alpha)
    scratch = None
    if a is not None:
        scratch = torch.empty_like(x)
        d.a, d.wo, d.bo, d.x1_scratch = a.data_ptr(), wo.data_ptr(), bo.data_ptr(), scratch.data_ptr()
    stats = None
    if wp is not None:
        d.wp, d.bp, d.res = wp.data_ptr(), bp.data_ptr(), res.data_ptr()
        hw = int(stat_img_rows) if stat_img_rows else M
        if want_stats and M % 128 == 0 and hw % 128 == 0:
            sbuf = torch.empty((M // 128, C, 2), dtype=torch.float32, device=x.device)
            d.out_stats = sbuf.data_ptr()
            stats = ChanStats(sbuf, hw // 128, C, hw, gn_sums_take(M // hw, C, hw, x.device))
            if stats.sums is not None:
                d.stat_sums, d.stat_img_rows = stats.sums.data_ptr(), hw
    nproj = (1 if a is not None else 0) + (1 if wp is not None else 0)
    flops = 2.0 * M * C * 8 * C + 2.0 * M * 4 * C * C + nproj * 2.0 * M * C * C
    nm = "ff_chain_kernel<%d,%d>" % (int(a is not None), int(wp is not None))
    with _Timed(nm + ((" M=%d C=%d" % (M, C)) if PROFILE_SHAPES else ""), flops, 2.0 * ((3 + nproj) * M * C + (12 + nproj) * C * C)):
        _check(lib().vd_ff_chain_f16(ctypes.byref(d), _stream()))
    if stats is not None:
        out._vd_stats = stats
    return out


def linear(x, w, bias=None, **kw):
    """x [..., K] @ w[N, K]^T (+bias) -> [..., N]"""
    lead = x.shape[:-1]
    act = kw.get("act", ACT_NONE)
    n_out = w.shape[0] // 2 if act == ACT_GEGLU else w.shape[0]
    out = gemm(x, w, bias=bias, M=x.numel() // x.shape[-1], **kw)
    v = out.view(*lead, n_out)
    for attr in ("_vd_stats", "_vd_rowsums"):   # a view does not carry the producer's statistics along: re-attach them
        a = getattr(out, attr, None)
        if a is not None:
            setattr(v, attr, a)
    return v


def conv2d_nhwc(x, w_packed, bias=None, *, ksize=3, stride=1, pad=1, ups=0, x1=None, cout=None, pad_hi=None, **kw):
    """x [B, H, W, C] (optionally ++ x1 on channels) -> [B, Hout, Wout, Cout]; w_packed [Cout][ks*ks*(C0+C1)]."""
    B, H, W, _ = x.shape
    Hv, Wv = H << ups, W << ups
    ph = pad if pad_hi is None else pad_hi
    Hout = (Hv + pad + ph - ksize) // stride + 1
    Wout = (Wv + pad + ph - ksize) // stride + 1
    cout = w_packed.shape[0] if cout is None else cout
    conv = dict(B=B, Hin=H, Win=W, Hout=Hout, Wout=Wout, ksize=ksize, stride=stride, pad=pad, ups=ups)
    out = gemm(x, w_packed, a1=x1, bias=bias, conv=conv, N=cout, K=w_packed.shape[1], out_shape=(B, Hout, Wout, cout), **kw)
    return out


def groupnorm_silu(x, gamma, beta, *, x1=None, groups=32, eps=1e-5, silu=True, out=None):
    """GroupNorm(+SiLU) over channels-last x [B, ..., C] (++ x1)."""
    _req(x, "x"); _req(x1, "x1"); _req(gamma, "gamma"); _req(beta, "beta")
    B = x.shape[0]
    c0 = x.shape[-1]
    c1 = x1.shape[-1] if x1 is not None else 0
    HW = x.numel() // (B * c0)
    C = c0 + c1
    if GN_STATS and _from_stats_ok(C, groups) and x.is_contiguous() and (x1 is None or x1.is_contiguous()):
        st0, st1 = stats_of(x), stats_of(x1)
        if st0 is not None or st1 is not None:   # a source without producer statistics is measured with one read of it alone
            st0 = st0 if st0 is not None else chan_stats(x)
            st1 = st1 if (st1 is not None or x1 is None) else chan_stats(x1)
            # one launch (block-local fold of the partials, slab-shaped panels): for everything (VD_GN_FORM=fused) or for tensors
            # of at most VD_GN_FUSED_MAX elements (small, L2-resident: the two launches are at their ~4.8 us floor each there)
            if GN_FORM == "fused" or B * HW * C <= GN_FUSED_MAX:
                return groupnorm_from_stats(x, gamma, beta, st0, x1=x1, st1=st1, groups=groups, eps=eps, silu=silu, out=out)
            # default: a tiny launch folds the partials into the per-(sample, channel) affine map, the apply launch streams
            # whole rows (measured: the slab-shaped single launch reads 80-byte pieces and ran no faster than the old pair)
            # round 5: where every source carries its producer's per-(sample, channel) sums the apply launch folds them itself
            if GN_SUMS and GN_SUMS_USE and groups <= 32 and st0.sums is not None and (x1 is None or st1.sums is not None):
                return gn_apply_sums(x, st0.sums, gamma, beta, x1=x1, sums1=st1.sums if x1 is not None else None, groups=groups,
                                     eps=eps, silu=silu, out=out)
            table = gn_table(st0, gamma, beta, st1=st1, B=B, groups=groups, eps=eps)
            return gn_apply_table(x, table, x1=x1, silu=silu, out=out)
    if out is None:
        out = torch.empty(x.shape[:-1] + (C,), dtype=torch.float16, device=x.device)
    nb = lib().vd_groupnorm_workspace_bytes(B, HW, C, groups)
    ws = workspace(nb, x.device, "gn")
    with _Timed("groupnorm (slab kernel or partial+apply)", 0.0, 2.0 * B * HW * C * 3):
        _check(lib().vd_groupnorm_silu_f16(_ptr(x), c0, _ptr(x1), c1, _ptr(gamma), _ptr(beta), _ptr(out), _ptr(ws), B, HW,
                                           groups, float(eps), 1 if silu else 0, _stream()))
    return out


# VD_GN_STATS=0: every GroupNorm measures its input itself (rounds 1-3: slab kernel or partial + apply); default: statistics
# come from the producers' epilogues where they emit them (csrc/gn_fused.hip)
GN_STATS = os.environ.get("VD_GN_STATS", "1") != "0"
GN_FUSED_MAX = int(os.environ.get("VD_GN_FUSED_MAX", "2700000"))   # the 16x16 and 8x8 levels (measured: -0.07 ms per forward; 5.3 M: neutral)
GN_FORM = os.environ.get("VD_GN_FORM", "table")   # table: vd_gn_table_f32 + vd_gn_apply_table_f16; fused: vd_groupnorm_from_stats_f16


def _from_stats_ok(C, groups):
    """Slab of vd_groupnorm_from_stats_f16: lcm(channels per group, 8) channels = at most 8 whole groups, dividing C."""
    if C % groups != 0 or C % 8 != 0:
        return False
    cg = C // groups
    if cg > 128 or C > 4096:   # limits of vd_gn_table_f32 (channels per group) and vd_gn_apply_table_f16 (row width)
        return False
    s = cg
    while s % 8:
        s += cg
    return C % s == 0 and s // cg <= 8 and s // 8 <= 256


def chan_stats(x, rows_per_partial=None):
    """Per-channel partial statistics of channels-last x [B, ..., C] with one read of x (the producers' out_stats format)."""
    _req(x, "x")
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    if rows_per_partial is None:
        rows_per_partial = 256 if HW % 256 == 0 else (64 if HW % 64 == 0 else HW)
    T = HW // rows_per_partial
    buf = torch.empty((B * T, C, 2), dtype=torch.float32, device=x.device)
    with _Timed("chan_stats_kernel", 0.0, 2.0 * B * HW * C):
        _check(lib().vd_chan_stats_f16(_ptr(x), B * HW, C, C, int(rows_per_partial), _ptr(buf), _stream()))
    return ChanStats(buf, T, C, HW)


def groupnorm_from_stats(x, gamma, beta, st0, *, x1=None, st1=None, groups=32, eps=1e-5, silu=True, out=None):
    """GroupNorm(+SiLU) over channels-last x (++ x1) from per-channel partial statistics: one launch, x read once."""
    _req(x, "x"); _req(x1, "x1"); _req(gamma, "gamma"); _req(beta, "beta")
    B = x.shape[0]
    c0 = x.shape[-1]
    c1 = x1.shape[-1] if x1 is not None else 0
    HW = x.numel() // (B * c0)
    C = c0 + c1
    if st0.C != c0 or st0.HW != HW or st0.buf.shape[0] != B * st0.T or (x1 is not None and (st1 is None or st1.C != c1 or st1.HW != HW or st1.buf.shape[0] != B * st1.T)):
        raise VdHipError("groupnorm_from_stats: statistics do not describe the input tensors")
    if out is None:
        out = torch.empty(x.shape[:-1] + (C,), dtype=torch.float16, device=x.device)
    with _Timed("gn_from_stats_kernel", 0.0, 2.0 * B * HW * C * 2):
        _check(lib().vd_groupnorm_from_stats_f16(_ptr(x), c0, _ptr(st0.buf), st0.T, _ptr(x1), c1, _ptr(st1.buf) if st1 is not None else None,
                                                 st1.T if st1 is not None else 0, _ptr(gamma), _ptr(beta), _ptr(out), B, HW, groups,
                                                 float(eps), 1 if silu else 0, _stream()))
    return out


def gn_table(st0, gamma, beta, *, st1=None, B, groups=32, eps=1e-5):
    """Partial statistics -> fp32 [B, 2, C] (scale, shift) of the GroupNorm as a per-(sample, channel) affine map."""
    C = st0.C + (st1.C if st1 is not None else 0)
    table = torch.empty((B, 2, C), dtype=torch.float32, device=st0.buf.device)
    with _Timed("gn_table_kernel", 0.0, 0.0):
        _check(lib().vd_gn_table_f32(_ptr(st0.buf), st0.T, st0.C, _ptr(st1.buf) if st1 is not None else None,
                                     st1.T if st1 is not None else 0, st1.C if st1 is not None else 0, B, st0.HW, _ptr(gamma), _ptr(beta),
                                     groups, float(eps), _ptr(table), _stream()))
    return table


def gn_apply_sums(x, sums0, gamma, beta, *, x1=None, sums1=None, groups=32, eps=1e-5, silu=True, out=None):
    """y = act(GroupNorm(cat(x, x1))) from the producers' per-(sample, channel) fixed-point sums (int64 [B * C, 2] each)."""
    _req(x, "x"); _req(x1, "x1"); _req(gamma, "gamma"); _req(beta, "beta"); _req(sums0, "sums0", torch.int64); _req(sums1, "sums1", torch.int64)
    B = x.shape[0]
    c0 = x.shape[-1]
    c1 = x1.shape[-1] if x1 is not None else 0
    HW = x.numel() // (B * c0)
    if sums0.numel() != 2 * B * c0 or (x1 is not None and (sums1 is None or sums1.numel() != 2 * B * c1)):
        raise VdHipError("gn_apply_sums: the sums do not describe the input tensors")
    if out is None:
        out = torch.empty(x.shape[:-1] + (c0 + c1,), dtype=torch.float16, device=x.device)
    with _Timed("gn_apply_sums_kernel", 0.0, 2.0 * B * HW * (c0 + c1) * 2):
        _check(lib().vd_gn_apply_sums_f16(_ptr(x), c0, _ptr(sums0), _ptr(x1), c1, _ptr(sums1), B, HW, _ptr(gamma), _ptr(beta), int(groups),
                                          float(eps), 1 if silu else 0, _ptr(out), _stream()))
    return out


def gn_apply_table(x, table, *, x1=None, silu=True, out=None):
    """y = act(cat(x, x1) * scale + shift) with the fp32 [B, 2, C] table of gn_table()."""
    _req(x, "x"); _req(x1, "x1")
    B, c0 = x.shape[0], x.shape[-1]
    c1 = x1.shape[-1] if x1 is not None else 0
    HW = x.numel() // (B * c0)
    if out is None:
        out = torch.empty(x.shape[:-1] + (c0 + c1,), dtype=torch.float16, device=x.device)
    with _Timed("gn_apply_table_kernel", 0.0, 2.0 * B * HW * (c0 + c1) * 2):
        _check(lib().vd_gn_apply_table_f16(_ptr(x), c0, _ptr(x1), c1, B, HW, _ptr(table), 1 if silu else 0, _ptr(out), _stream()))
    return out


def groupnorm0d_silu(x, gamma_sc, beta_sc, *, x1=None, groups=32, eps=1e-5, silu=True):
    """0-D data flow GroupNorm: x [B, S, C] (++ x1 on C), gamma_sc / beta_sc [S, C0+C1] -> [B, S, C0+C1]."""
    _req(x, "x"); _req(x1, "x1"); _req(gamma_sc, "gamma"); _req(beta_sc, "beta")
    B, S, c0 = x.shape
    c1 = x1.shape[-1] if x1 is not None else 0
    out = torch.empty((B, S, c0 + c1), dtype=torch.float16, device=x.device)
    _check(lib().vd_groupnorm0d_silu_f16(_ptr(x), c0, _ptr(x1), c1, _ptr(gamma_sc), _ptr(beta_sc), _ptr(out), B, S, groups,
                                         float(eps), 1 if silu else 0, _stream()))
    return out


def layernorm(x, gamma, beta, eps=1e-5, out=None):
    _req(x, "x"); _req(gamma, "gamma"); _req(beta, "beta")
    C = x.shape[-1]
    rows = x.numel() // C
    if out is None:
        out = torch.empty_like(x)
    with _Timed("layernorm_kernel", 0.0, 2.0 * rows * C * 2):
        _check(lib().vd_layernorm_f16(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(out), rows, C, float(eps), _stream()))
    return out


def attention(q, k, v, heads, *, scale=None, causal=False, out=None):
    """q [B, Nq, *], k/v [B, Nk, *]: last-dim views (possibly column slices of a fused projection) with stride 1.

    Head h uses columns h*D:(h+1)*D of each.  Returns [B, Nq, heads*D] contiguous."""
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        if not t.is_cuda or t.dtype != torch.float16 or t.stride(-1) != 1 or t.dim() != 3:
            raise VdHipError("%s must be a 3-d fp16 GPU tensor with unit inner stride" % n)
    B, Nq, C = q.shape
    Nk = k.shape[1]
    D = C // heads
    if out is None:
        out = torch.empty((B, Nq, C), dtype=torch.float16, device=q.device)
    if scale is None:
        scale = D ** -0.5
    with _Timed(("attn_fwd_kernel<%d>" % D) + ((" Nq=%d Nk=%d B=%d" % (Nq, Nk, B)) if PROFILE_SHAPES else ""), 4.0 * B * Nq * Nk * C, 2.0 * B * C * (2 * Nq + 2 * Nk)):
        _check(lib().vd_attention_f16(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, heads, Nq, Nk, D, q.stride(1), k.stride(1),
                                      v.stride(1), out.stride(1), q.stride(0), k.stride(0), v.stride(0), out.stride(0),
                                      float(scale), int(causal), _stream()))
    return out


XATTN_FUSED = os.environ.get("VD_XATTN_FUSED", "1") != "0"   # development switch: 0 = row_stats -> q GEMM -> attention


def xattn_supported(heads, D):
    """True when vd_xattn_f16 is instantiated for this head geometry (and not switched off)."""
    return XATTN_FUSED and bool(lib().vd_xattn_supported(int(heads), int(D)))


def xattn(x, wq, bq, colsum, ln_eps, k, v, heads, *, scale=None):
    """softmax((LayerNorm(x) wq_h^T) k_h^T * scale) v_h per head in one launch (vd_xattn_f16).  x [B, Nq, C] contiguous;
    wq / bq / colsum: the LayerNorm-folded query projection (hip_layers.fold_layernorm); k, v [B, Nk, *] last-dim views
    of the pre-projected context.  Returns [B, Nq, C]."""
    for t, n in ((x, "x"), (wq, "wq")):
        _req(t, n)
    _req(colsum, "colsum", torch.float32)
    if bq is not None:
        _req(bq, "bq")
    for t, n in ((k, "k"), (v, "v")):
        if not t.is_cuda or t.dtype != torch.float16 or t.stride(-1) != 1 or t.dim() != 3:
            raise VdHipError("%s must be a 3-d fp16 GPU tensor with unit inner stride" % n)
    B, Nq, C = x.shape
    Nk = k.shape[1]
    D = C // heads
    if tuple(wq.shape) != (C, C) or colsum.numel() != C or k.shape[0] != B or v.shape[1] != Nk or k.shape[2] != C or v.shape[2] != C:
        raise VdHipError("xattn: operand shapes do not match B=%d C=%d" % (B, C))
    if scale is None:
        scale = D ** -0.5
    out = torch.empty((B, Nq, C), dtype=torch.float16, device=x.device)
    flops = 2.0 * B * Nq * C * C + 4.0 * B * Nq * Nk * C
    with _Timed(("xattn_kernel<%d>" % D) + ((" Nq=%d Nk=%d B=%d" % (Nq, Nk, B)) if PROFILE_SHAPES else ""), flops,
                2.0 * (2 * B * Nq * C + C * C + 2 * B * Nk * C)):
        _check(lib().vd_xattn_f16(_ptr(x), _ptr(wq), _ptr(bq) if bq is not None else None, _ptr(colsum), float(ln_eps),
                                  _ptr(k), _ptr(v), _ptr(out), B, heads, Nq, Nk, D, k.stride(1), v.stride(1), k.stride(0),
                                  v.stride(0), float(scale), _stream()))
    return out


def softmax_rows(s, out=None):
    _req(s, "s", torch.float32)
    n = s.shape[-1]
    rows = s.numel() // n
    if out is None:
        out = torch.empty(s.shape, dtype=torch.float16, device=s.device)
    _check(lib().vd_softmax_rows_f32_f16(_ptr(s), _ptr(out), rows, n, _stream()))
    return out


def softmax_rows_f32(s, scale=1.0):
    """softmax(scale * s) over the last dim, fp32 in / fp32 out."""
    _req(s, "s", torch.float32)
    n = s.shape[-1]
    out = torch.empty_like(s)
    _check(lib().vd_softmax_rows_f32_f32(_ptr(s), _ptr(out), s.numel() // n, n, float(scale), _stream()))
    return out


def timestep_embedding(t, dim, max_period=10000.0):
    _req(t, "t", torch.int64)
    out = torch.empty((t.shape[0], dim), dtype=torch.float16, device=t.device)
    _check(lib().vd_timestep_embedding_f16(_ptr(t), _ptr(out), t.shape[0], dim, float(max_period), _stream()))
    return out


def cfg_ddim_step(x, eps, *, guided, guidance_scale, a_t, a_prev, sigma, sqrt_one_minus_at, noise=None,
                  want_pred_x0=True):
    _req(x, "x"); _req(eps, "eps"); _req(noise, "noise")
    n = x.numel()
    if eps.numel() != (2 * n if guided else n):
        raise VdHipError("eps has %d elements, expected %d" % (eps.numel(), 2 * n if guided else n))
    x_prev = torch.empty_like(x)
    pred_x0 = torch.empty_like(x) if want_pred_x0 else None
    _check(lib().vd_cfg_ddim_step_f16(_ptr(x), _ptr(eps), _ptr(noise), _ptr(x_prev), _ptr(pred_x0), n, 1 if guided else 0,
                                      float(guidance_scale), float(a_t), float(a_prev), float(sigma),
                                      float(sqrt_one_minus_at), _stream()))
    return x_prev, pred_x0


def cfg_ddim_step_dev(x, eps, coef, *, guided, x_prev, pred_x0=None, noise=None):
    """CFG + DDIM update with the step scalars in a device fp32[6] tensor; writes into caller-owned buffers."""
    _req(x, "x"); _req(eps, "eps"); _req(noise, "noise"); _req(coef, "coef", torch.float32); _req(x_prev, "x_prev"); _req(pred_x0, "pred_x0")
    n = x.numel()
    if eps.numel() != (2 * n if guided else n):
        raise VdHipError("eps has %d elements, expected %d" % (eps.numel(), 2 * n if guided else n))
    _check(lib().vd_cfg_ddim_step_dev_f16(_ptr(x), _ptr(eps), _ptr(noise), _ptr(x_prev), _ptr(pred_x0), n,
                                          1 if guided else 0, _ptr(coef), _stream()))
    return x_prev, pred_x0


def q_sample(x0, noise, sa, sb):
    _req(x0, "x0"); _req(noise, "noise"); _req(sa, "sa", torch.float32); _req(sb, "sb", torch.float32)
    out = torch.empty_like(x0)
    B = x0.shape[0]
    _check(lib().vd_q_sample_f16(_ptr(x0), _ptr(noise), _ptr(sa), _ptr(sb), _ptr(out), B, x0.numel() // B, _stream()))
    return out


def nchw_to_nhwc(x):
    _req(x, "x")
    B, C, H, W = x.shape
    y = torch.empty((B, H, W, C), dtype=torch.float16, device=x.device)
    _check(lib().vd_nchw_to_nhwc_f16(_ptr(x), _ptr(y), B, C, H, W, _stream()))
    return y


def nhwc_to_nchw(x, scale=1.0, shift=0.0, clamp01=False):
    _req(x, "x")
    B, H, W, C = x.shape
    y = torch.empty((B, C, H, W), dtype=torch.float16, device=x.device)
    _check(lib().vd_nhwc_to_nchw_f16(_ptr(x), _ptr(y), B, C, H, W, float(scale), float(shift), 1 if clamp01 else 0, _stream()))
    return y


def im2col_small(x, *, layout, ksize, stride=1, pad=1, pad_hi=None, in_scale=1.0, in_shift=0.0):
    """Small-Cin im2col. layout 'nchw' or 'nhwc'. Returns (A [M, kpad], (B, Hout, Wout))."""
    _req(x, "x")
    if layout == "nchw":
        B, C, H, W = x.shape
        sb, sc, sy, sx = C * H * W, H * W, W, 1
    else:
        B, H, W, C = x.shape
        sb, sc, sy, sx = H * W * C, 1, W * C, C
    ph = pad if pad_hi is None else pad_hi
    Hout = (H + pad + ph - ksize) // stride + 1
    Wout = (W + pad + ph - ksize) // stride + 1
    kk = ksize * ksize * C
    kpad = ((kk + 63) // 64) * 64
    a = torch.empty((B * Hout * Wout, kpad), dtype=torch.float16, device=x.device)
    _check(lib().vd_im2col_small_f16(_ptr(x), _ptr(a), B, C, H, W, Hout, Wout, ksize, stride, pad, sb, sc, sy, sx, kpad,
                                     float(in_scale), float(in_shift), _stream()))
    return a, (B, Hout, Wout)


def diag_gaussian_sample(moments, noise, B, zc, H, W, scale):
    _req(moments, "moments"); _req(noise, "noise")
    z = torch.empty((B, zc, H, W), dtype=torch.float16, device=moments.device)
    _check(lib().vd_diag_gaussian_sample_f16(_ptr(moments), _ptr(noise), _ptr(z), B, zc, H * W, float(scale), _stream()))
    return z


def axpby(x, y, a, b, out=None):
    _req(x, "x"); _req(y, "y")
    if out is None:
        out = torch.empty_like(x)
    _check(lib().vd_axpby_f16(_ptr(x), _ptr(y), _ptr(out), float(a), float(b), x.numel(), _stream()))
    return out


UNARY_GELU_ERF, UNARY_TANH = 0, 1


def unary(x, op, out=None):
    """out = erf-GELU(x) / tanh(x), element-wise (in place when out is x)."""
    _req(x, "x")
    if out is None:
        out = torch.empty_like(x)
    _check(lib().vd_unary_f16(_ptr(x), _ptr(out), int(op), x.numel(), _stream()))
    return out


def embed_tokens(ids, tok_emb, pos_emb):
    _req(ids, "ids", torch.int64); _req(tok_emb, "tok_emb"); _req(pos_emb, "pos_emb")
    B, L = ids.shape
    C = tok_emb.shape[1]
    out = torch.empty((B, L, C), dtype=torch.float16, device=ids.device)
    _check(lib().vd_embed_tokens_f16(_ptr(ids), _ptr(tok_emb), _ptr(pos_emb), _ptr(out), B, L, C, _stream()))
    return out


def clip_vision_embed(patches, class_emb, pos_emb, token_scale=None):
    _req(patches, "patches"); _req(class_emb, "class_emb"); _req(pos_emb, "pos_emb"); _req(token_scale, "token_scale", torch.float32)
    B, Lm1, C = patches.shape
    out = torch.empty((B, Lm1 + 1, C), dtype=torch.float16, device=patches.device)
    _check(lib().vd_clip_vision_embed_f16(_ptr(patches), _ptr(class_emb), _ptr(pos_emb), _ptr(token_scale), _ptr(out), B,
                                          Lm1 + 1, C, _stream()))
    return out


def patchify(pixels, P):
    _req(pixels, "pixels")
    B, C, H, W = pixels.shape
    kk = C * P * P
    kpad = ((kk + 63) // 64) * 64
    a = torch.empty((B * (H // P) * (W // P), kpad), dtype=torch.float16, device=pixels.device)
    _check(lib().vd_patchify_f16(_ptr(pixels), _ptr(a), B, C, H, W, P, kpad, _stream()))
    return a


def scale_by_row_norm_(z, *, ref=None, pool_idx=None, row_scale=None):
    _req(z, "z"); _req(ref, "ref"); _req(pool_idx, "pool_idx", torch.int32); _req(row_scale, "row_scale", torch.float32)
    B, L, C = z.shape
    _check(lib().vd_scale_by_row_norm_f16(_ptr(z), _ptr(ref), _ptr(pool_idx), _ptr(row_scale), B, L, C, _stream()))
    return z


def image_to_u8(images):
    """[B,3,H,W] float32 / float16 in [0,1] -> uint8 [B,H,W,3] exactly as torchvision ToPILImage quantises (app.py:319)."""
    assert images.dim() == 4 and images.shape[1] == 3 and images.is_cuda and images.dtype in (torch.float32, torch.float16)
    img = images.contiguous()
    B, _, H, W = img.shape
    out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=img.device)
    _check(lib().vd_image_to_u8(_ptr(img), 0 if img.dtype == torch.float32 else 1, B, H, W, _ptr(out), _stream()))
    return out


_pre_tables = {}


def clip_preprocess(images, size=224):
    """[B,3,H,W] float32 / float16 in [0,1] (quantised like ToPILImage) or uint8 -> CLIP pixel_values [B,3,size,size]
    fp16: Pillow-exact bicubic resize of the shortest edge, centre crop, rescale, normalise (vd_clip_preprocess_f16)."""
    from . import resample as R
    assert images.dim() == 4 and images.shape[1] == 3 and images.is_cuda
    kind = {torch.float32: 0, torch.float16: 1, torch.uint8: 2}[images.dtype]
    img = images.contiguous()
    B, _, H, W = img.shape
    rh, rw = R.resize_output_size(H, W, size)
    dev = img.device
    key = (dev.index, H, W, size)
    tabs = _pre_tables.get(key)
    if tabs is None:
        def taps(n_in, n_out):
            if n_in == n_out:
                return None, None, 0
            b, k, ks = R.pil_bicubic_taps(n_in, n_out)
            return torch.from_numpy(b).to(dev), torch.from_numpy(k).to(dev), ks
        tabs = (taps(W, rw), taps(H, rh), torch.from_numpy(R.clip_norm_table()).to(dev))
        _pre_tables[key] = tabs
    (hb, hk, hks), (vb, vk, vks), table = tabs
    tmp = workspace(B * 3 * H * size, dev, "clip_pre")
    out = torch.empty((B, 3, size, size), dtype=torch.float16, device=dev)
    _check(lib().vd_clip_preprocess_f16(_ptr(img), kind, B, H, W, rh, rw, _ptr(hb), _ptr(hk), hks, _ptr(vb), _ptr(vk), vks,
                                        (rh - size) // 2, (rw - size) // 2, size, _ptr(table), _ptr(tmp), _ptr(out), _stream()))
    return out


def mask_patch_weights(masks, size=224, patch=14):
    """[B,1,H,W] float32 / float16 mask -> fp32 [B, 1 + (size/patch)^2] = [global mean | patch means] of the mask clamped
    to [0,1] and resized bilinearly to size x size (reference clip.py:104-122)."""
    assert masks.dim() == 4 and masks.shape[1] == 1 and masks.is_cuda and masks.dtype in (torch.float32, torch.float16)
    m = masks.contiguous()
    B, _, H, W = m.shape
    out = torch.empty((B, 1 + (size // patch) ** 2), dtype=torch.float32, device=m.device)
    _check(lib().vd_mask_patch_weights(_ptr(m), 0 if m.dtype == torch.float32 else 1, B, H, W, size, patch, _ptr(out), _stream()))
    return out


def color_adjust(images, ref):
    """'Simple' colour adjustment (reference app.py:373-379): images [B,3,H,W] fp16, ref [3,H,W] / [1,3,H,W] (one input
    image for the whole batch) or [B,3,H,W] -> clamp((img - mean) / std * std(ref) + mean(ref), 0, 1) per channel."""
    _req(images, "images"); _req(ref, "ref")
    B, C, H, W = images.shape
    if C != 3 or tuple(ref.shape[-3:]) != (3, H, W) or ref.dim() not in (3, 4) or (ref.dim() == 4 and ref.shape[0] not in (1, B)):
        raise VdHipError("color_adjust: images [B,3,H,W] = %s need a reference of the same H x W ([3,H,W], [1,3,H,W] or [B,3,H,W]), "
                         "got %s -- resize the reference to the output size first (app.py:373-379 only uses its per-channel "
                         "mean / std)" % (tuple(images.shape), tuple(ref.shape)))
    per_image = ref.dim() == 4 and ref.shape[0] == B and B > 1
    out = torch.empty_like(images)
    _check(lib().vd_color_adjust_f16(_ptr(images), _ptr(ref), _ptr(out), B, H, W, 3 * H * W if per_image else 0, _stream()))
    return out


def adjust_rank(x, g, keep, iters=64):
    """x [B, L, C] fp16 -> keep * A + sum_i g[i] u_i u_i^T A + row means, rescaled to x's std (vd_adjust_rank_f16);
    g: fp32 [q] device tensor of singular-value scale offsets, keep in {0., 1.}."""
    _req(x, "x"); _req(g, "g", torch.float32)
    B, L, C = x.shape
    q = g.numel()
    ws = workspace(lib().vd_adjust_rank_workspace_bytes(B, L, C, q), x.device, "adjust_rank")
    y = torch.empty_like(x)
    _check(lib().vd_adjust_rank_f16(_ptr(x), _ptr(y), B, L, C, q, _ptr(g), float(keep), int(iters), _ptr(ws), _stream()))
    return y


def probe_lds_tr16(addr_bytes):
    """addr_bytes: (64,) int32 per-lane LDS byte addresses -> (64, 4) int16 values read by ds_read_b64_tr_b16."""
    _req(addr_bytes, "addr_bytes", torch.int32)
    out = torch.zeros((64, 4), dtype=torch.int16, device=addr_bytes.device)
    _check(lib().vd_probe_lds_tr16(_ptr(addr_bytes), _ptr(out), _stream()))
    return out.cpu()


def probe_xcc_ids(device, grid_x, grid_y=1):
    """XCD id of every block of a grid_x x grid_y launch -> int32 [grid_y, grid_x] (host)."""
    out = torch.full((grid_y, grid_x), -1, dtype=torch.int32, device=device)
    _check(lib().vd_probe_xcc_ids(_ptr(out), int(grid_x), int(grid_y), _stream()))
    return out.cpu()


def probe_mfma_layout(device):
    a_k = torch.full((16,), -2, dtype=torch.int32, device=device)
    c_row = torch.zeros((64, 16), dtype=torch.int32, device=device)
    c_col = torch.zeros((64, 16), dtype=torch.int32, device=device)
    _check(lib().vd_probe_mfma_layout(_ptr(a_k), _ptr(c_row), _ptr(c_col), _stream()))
    return a_k.cpu(), c_row.cpu(), c_col.cpu()


# ---- device guard ---------------------------------------------------------------------------------------------------
# Kernels go to torch's current stream OF THE DEVICE THE TENSORS LIVE ON: when that is not the calling thread's current
# device (model on cuda:1 while cuda:0 is current, as the reference allows) the call runs under torch.cuda.device(...);
# operands spread over several devices are rejected.  The common case costs one integer comparison.
def _guarded(fn):
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        dev = None
        for t in list(args) + list(kwargs.values()):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                if dev is None:
                    dev = t.device
                elif t.device != dev:
                    raise VdHipError("%s: operands live on different devices (%s and %s)" % (fn.__name__, dev, t.device))
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapper


for _name in ("gemm", "gemm_row320", "row320_chain", "groupnorm_affine", "ff_geglu", "xattn", "row_stats", "linear", "conv2d_nhwc", "groupnorm_silu", "groupnorm0d_silu", "layernorm", "attention", "softmax_rows", "softmax_rows_f32",
              "timestep_embedding", "cfg_ddim_step", "cfg_ddim_step_dev", "q_sample", "nchw_to_nhwc", "nhwc_to_nchw",
              "im2col_small", "diag_gaussian_sample", "axpby", "embed_tokens", "clip_vision_embed", "patchify",
              "unary", "scale_by_row_norm_", "image_to_u8", "clip_preprocess", "probe_lds_tr16", "mask_patch_weights", "color_adjust", "adjust_rank"):
    globals()[_name] = _guarded(globals()[_name])
