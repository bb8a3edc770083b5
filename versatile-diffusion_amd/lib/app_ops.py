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
