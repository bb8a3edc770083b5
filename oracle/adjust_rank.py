"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's focus control of the image context,
/root/reference/app.py:48-127 (`decompose` + class `adjust_rank`).  Never imported by the product package.

Pinned by tests/golden/adjust_rank.npz, which oracle/gen_golden_adjust_rank.py produces by executing the REFERENCE's own
source lines (app.py cannot be imported: it launches gradio and loads weights at import time).  torch.pca_lowrank is a
randomised range finder: with niter=100 it converges on the top-q singular triplets up to round-off wherever the
spectrum has a gap, so the pin is on the reconstructed tensor, not on the signs / rotations of u and v."""
import numpy as np
import torch


def decompose(x, q=20, niter=100):
    """app.py:48-55"""
    x_mean = x.mean(-1, keepdim=True)
    x_input = x - x_mean
    u, s, v = torch.pca_lowrank(x_input, q=q, center=False, niter=niter)
    ss = torch.stack([torch.diag(si) for si in s])
    x_lowrank = torch.bmm(torch.bmm(u, ss), torch.permute(v, [0, 2, 1]))
    return u, s, v, x_mean, x_input - x_lowrank


def level_scales(lvl, max_drop_rank=(1, 5), q=20):
    """The level -> singular-value multipliers of app.py:57-86 / :101-118: (f [q], keep_remainder)."""
    sem, sty = max_drop_rank
    f = np.ones(q)
    if lvl < 0.5:
        t0, y00, t1, y01 = np.exp((0 - 0.5) * 2), -sem, np.exp(0.0), 1
        y0 = (np.exp((lvl - 0.5) * 2) - t0) / (t1 - t0) * (y01 - y00) + y00
        for xi in range(0, sem + 1):
            f[xi] = max((xi - 0) / (sem + 1 - 0) * (1 - y0) + y0, 0)
        return f, True
    t0, y00, t1, y01 = np.exp((1 - 0.5) * 2), -(q - sty), np.exp(0.0), 1
    y0 = (np.exp((lvl - 0.5) * 2) - t0) / (t1 - t0) * (y01 - y00) + y00
    for xi in range(sty, q):
        f[xi] = max((xi - (q - 1)) / ((sty - 1) - (q - 1)) * (1 - y0) + y0, 0)
    return f, False


def adjust_rank(x, lvl, max_drop_rank=(1, 5), q=20, niter=100):
    """adjust_rank.__call__ (app.py:88-127) on x [B, L, C]."""
    if lvl == 0.5:
        return x
    fp16 = x.dtype == torch.float16
    x = x.float()
    std_save = x.std(dim=[-2, -1])
    u, s, v, x_mean, x_remain = decompose(x, q=q, niter=niter)
    f, keep = level_scales(lvl, max_drop_rank, q)
    s = s * torch.from_numpy(f).to(s.dtype)[None]
    if not keep:
        x_remain = 0
    ss = torch.stack([torch.diag(si) for si in s])
    x_new = torch.bmm(torch.bmm(u, ss), torch.permute(v, [0, 2, 1])) + x_mean + x_remain
    x_new = x_new / x_new.std(dim=[-2, -1])[:, None, None] * std_save[:, None, None]
    return x_new.half() if fp16 else x_new


def exact(x, lvl, max_drop_rank=(1, 5), q=20):
    """The limit pca_lowrank converges to: the same reconstruction from the exact SVD (float64)."""
    x = x.double()
    std_save = x.std(dim=[-2, -1])
    mean = x.mean(-1, keepdim=True)
    a = x - mean
    u, s, vh = torch.linalg.svd(a, full_matrices=False)
    f, keep = level_scales(lvl, max_drop_rank, q)
    g = torch.from_numpy(f - (1.0 if keep else 0.0))
    low = torch.einsum("blq,bq,bqc->blc", u[:, :, :q], s[:, :q] * g[None], vh[:, :q])
    x_new = (a if keep else 0) + low + mean
    return x_new / x_new.std(dim=[-2, -1])[:, None, None] * std_save[:, None, None]
