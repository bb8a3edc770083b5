"""One-time weight re-layouts (views/permutes only, done at load time, never in the step loop)."""
import torch


def pack_conv_weight(w):
    """torch conv weight [Cout, Cin, kh, kw] -> K-contiguous [Cout, kh*kw*Cin] with k = (ky, kx, cin)."""
    co, ci, kh, kw = w.shape
    return w.permute(0, 2, 3, 1).reshape(co, kh * kw * ci).contiguous()


def pack_conv_weight_stream(w):
    """torch conv weight [Cout, Cin, 3, 3] -> MFMA-fragment order of vd_conv3x3_wstream_f16:
    [Cout / 32][Cin / 64][9 taps][4 k-steps][64 lanes][8], lane l of (n tile t, chunk c, tap, k-step s) holding
    W[32 t + (l & 31)][64 c + 16 s + 8 (l >> 5) .. + 8][tap]: each A operand of an MFMA is one contiguous 1-KiB block."""
    co, ci, kh, kw = w.shape
    assert kh == 3 and kw == 3 and co % 32 == 0 and ci % 64 == 0
    v = w.reshape(co // 32, 32, ci // 64, 4, 2, 8, 3, 3)          # (t, l31, c, s, hi, e, ky, kx)
    v = v.permute(0, 2, 6, 7, 3, 4, 1, 5)                          # (t, c, ky, kx, s, hi, l31, e)
    return v.reshape(co // 32, ci // 64, 9, 4, 64, 8).contiguous()


def pack_linear_weight_stream(w):
    """1x1 conv / linear weight [Cout, Cin] -> MFMA-fragment order [Cout / 32][Cin / 64][4 k-steps][64 lanes][8] (the folded skip
    convolution of vd_conv3x3_wstream_f16): lane l of (n tile t, chunk c, k-step s) holds W[32 t + (l & 31)][64 c + 16 s + 8 (l >> 5) .. + 8]."""
    co, ci = w.shape
    assert co % 32 == 0 and ci % 64 == 0
    v = w.reshape(co // 32, 32, ci // 64, 4, 2, 8)      # (t, l31, c, s, hi, e)
    return v.permute(0, 2, 3, 4, 1, 5).reshape(co // 32, ci // 64, 4, 64, 8).contiguous()


def pack_conv_weight_small(w, kpad=None):
    """Same ordering, zero padded along K to a multiple of 64 (matches vd_im2col_small_f16)."""
    p = pack_conv_weight(w)
    k = p.shape[1]
    kpad = ((k + 63) // 64) * 64 if kpad is None else kpad
    out = torch.zeros((p.shape[0], kpad), dtype=p.dtype, device=p.device)
    out[:, :k] = p
    return out


def pack_patch_weight(w, kpad=None):
    """CLIP patch embedding conv [Cout, C, P, P] -> [Cout, kpad], k = (c, py, px) (matches vd_patchify_f16)."""
    co = w.shape[0]
    p = w.reshape(co, -1)
    k = p.shape[1]
    kpad = ((k + 63) // 64) * 64 if kpad is None else kpad
    out = torch.zeros((co, kpad), dtype=p.dtype, device=p.device)
    out[:, :k] = p
    return out


def pack_geglu(w, b=None):
    """GEGLU proj weight [2*inner, K] (rows: values then gates) -> per 64-row group [32 value | 32 gate], so the
    value and gate of an output column land in the same lane/register of one wave's MFMA tiles."""
    two_inner, k = w.shape
    inner = two_inner // 2
    assert inner % 64 == 0, "GEGLU inner dim must be a multiple of 64"
    val = w[:inner].reshape(inner // 32, 32, k)
    gate = w[inner:].reshape(inner // 32, 32, k)
    wp = torch.cat([val, gate], dim=1).reshape(two_inner, k).contiguous()
    bp = None
    if b is not None:
        bp = torch.cat([b[:inner].reshape(-1, 32), b[inner:].reshape(-1, 32)], dim=1).reshape(-1).contiguous()
    return wp, bp
