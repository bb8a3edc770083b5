"""Device versions of the tensor work the reference does in app.py around the model calls:
the 'Simple' colour adjustment of image variation (app.py:373-379) and the `adjust_rank` focus control of the image
context (app.py:48-127).  Same call signatures as there; the arithmetic runs in hand-written HIP kernels (vd_hip)."""
import numpy as np
import torch

from vd_hip import ops


def color_adjust_simple(imout, cx):
    """imout: [B,3,H,W] decoded images (or a list of [3,H,W]); cx: the input image [1,3,H,W] / [3,H,W].  Returns the
    adjusted images in the form they came in (reference app.py:373-379: per channel (x - mean) / std * std_cx + mean_cx,
    clamped to [0,1])."""
    as_list = isinstance(imout, (list, tuple))
    x = torch.stack(list(imout)) if as_list else imout
    dt = x.dtype
    out = ops.color_adjust(x.to(torch.float16).contiguous(), cx.to(device=x.device, dtype=torch.float16).contiguous()).to(dt)
    return list(out) if as_list else out


class adjust_rank(object):
    """Focus control of the image context: rescale the leading singular components of the (row-centred) CLIP token matrix
    -- lvl < 0.5 suppresses the top `max_drop_rank[0] + 1` ("semantic") components, lvl > 0.5 suppresses components
    `max_drop_rank[1]` .. q-1 and everything beyond rank q ("style") -- then restore the tensor's std.  Same constructor,
    call signature and level -> scale curves as the reference class (app.py:56-127); the decomposition and the
    reconstruction run in vd_adjust_rank_f16 (fp32 subspace iteration on the device instead of torch.pca_lowrank)."""

    def __init__(self, max_drop_rank=[1, 5], q=20):
        self.max_semantic_drop_rank = max_drop_rank[0]
        self.max_style_drop_rank = max_drop_rank[1]
        self.q = q
        t0, y00 = np.exp((0 - 0.5) * 2), -self.max_semantic_drop_rank
        t1, y01 = np.exp((0.5 - 0.5) * 2), 1
        self.t2y0_semf = lambda t: (np.exp((t - 0.5) * 2) - t0) / (t1 - t0) * (y01 - y00) + y00
        sx1 = self.max_semantic_drop_rank + 1
        self.x2y_semf = lambda x, y0: (x - 0) / (sx1 - 0) * (1 - y0) + y0
        u0, z00 = np.exp((1 - 0.5) * 2), -(q - self.max_style_drop_rank)
        self.t2y0_styf = lambda t: (np.exp((t - 0.5) * 2) - u0) / (t1 - u0) * (1 - z00) + z00
        tx0, tx1 = q - 1, self.max_style_drop_rank - 1
        self.x2y_styf = lambda x, y0: (x - tx0) / (tx1 - tx0) * (1 - y0) + y0

    def scales(self, lvl):
        """(f [q] multipliers of the singular values, keep_remainder) for a focus level in [0, 1], lvl != 0.5."""
        f = np.ones(self.q, dtype=np.float64)
        if lvl < 0.5:
            assert lvl >= 0
            for xi in range(0, self.max_semantic_drop_rank + 1):
                f[xi] = max(self.x2y_semf(xi, self.t2y0_semf(lvl)), 0)
            return f, True
        assert lvl <= 1
        for xi in range(self.max_style_drop_rank, self.q):
            f[xi] = max(self.x2y_styf(xi, self.t2y0_styf(lvl)), 0)
        return f, False

    def __call__(self, x, lvl):
        if lvl == 0.5:
            return x
        f, keep = self.scales(lvl)
        dt = x.dtype
        x16 = x.to(torch.float16).contiguous()
        g = torch.from_numpy((f - (1.0 if keep else 0.0)).astype(np.float32)).to(x16.device)
        return ops.adjust_rank(x16, g, 1.0 if keep else 0.0).to(dt)
