"""Generates tests/golden/adjust_rank.npz by running the REFERENCE's own `decompose` / `adjust_rank` code
(/root/reference/app.py:48-127, executed from the source text: the module itself starts gradio on import).

    python oracle/gen_golden_adjust_rank.py        (CPU container, needs /root/reference)

Input: one CLIP-like local-token matrix [1, 256, 768] with a geometrically decaying spectrum (so the randomised
pca_lowrank converges and the fixture does not depend on its random start), stored in fp16 as the product receives it.
Outputs: the reference's results (fed the fp16 tensor, as app.py does under `net.half()`) for focus levels 0.0, 0.3,
0.8 and 1.0."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_APP = "/root/reference/app.py"


def load_reference_adjust_rank():
    src = open(REF_APP).read().split("\n")
    start = next(i for i, l in enumerate(src) if l.startswith("def decompose("))
    end = next(i for i, l in enumerate(src) if l.startswith("def remove_duplicate_word("))
    ns = {"torch": torch, "np": np}
    exec("\n".join(src[start:end]), ns)
    return ns["adjust_rank"]


def make_input(seed=5):
    g = torch.Generator().manual_seed(seed)
    L, C, r = 256, 768, 48
    u, _ = torch.linalg.qr(torch.randn(L, r, generator=g, dtype=torch.float64))
    v, _ = torch.linalg.qr(torch.randn(C, r, generator=g, dtype=torch.float64))
    s = 6.0 * 0.82 ** torch.arange(r, dtype=torch.float64)
    x = (u * s) @ v.T + 0.004 * torch.randn(L, C, generator=g, dtype=torch.float64)
    x = x + 0.05 * torch.randn(L, 1, generator=g, dtype=torch.float64)       # row means the reference removes
    return x[None].half()


def main():
    ar = load_reference_adjust_rank()(max_drop_rank=[1, 5], q=20)
    x = make_input()
    out = {"x": x.numpy()}
    for lvl in (0.0, 0.3, 0.8, 1.0):
        torch.manual_seed(1234)   # pca_lowrank draws its start from the global generator
        out["y_%02d" % int(lvl * 10)] = ar(x.clone(), lvl).numpy()
    path = os.path.join(ROOT, "tests", "golden", "adjust_rank.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
