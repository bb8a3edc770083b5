"""Tile-configuration x split-K sweep over the UNet's GEMM / conv shapes (dev tool; calibrates vd_gemm_plan's cost model).

    python tools/gemm_sweep.py [out.json]      VD_SWEEP_FILTER=M=32768 restricts to one level

Every instantiation of the GEMM template is forced in-process through vd_gemm_set_override; per shape the table shows
the planner's own choice ('plan') next to every configuration, in us, and the best one in TF/s.  Shapes carry their
launch count in one UNet forward at the bench shape (CFG batch 8, 64x64 latent) so the bottom line is the GEMM time of a
forward under (a) the planner and (b) the per-shape best.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
import torch  # noqa: E402
from vd_hip import ops  # noqa: E402
from vd_hip.loader import lib  # noqa: E402
from vd_hip.pack import pack_conv_weight  # noqa: E402

dev = torch.device("cuda:0")
# (count per forward, M, N, K, ksize, kind)   kind: c = conv/plain with bias, g = GEGLU, l = LayerNorm fold, r = +residual
SHAPES = [
    (7, 32768, 320, 2880, 3, "c"), (2, 32768, 320, 5760, 3, "c"), (1, 32768, 320, 8640, 3, "c"), (1, 32768, 640, 5760, 3, "c"),
    (25, 32768, 320, 320, 1, "r"), (5, 32768, 960, 320, 1, "l"), (5, 32768, 320, 320, 1, "l"), (5, 32768, 320, 1280, 1, "r"),
    (5, 32768, 2560, 320, 1, "g"), (3, 32768, 320, 640, 1, "c"),
    (6, 8192, 640, 5760, 3, "c"), (1, 8192, 640, 2880, 3, "c"), (1, 8192, 640, 11520, 3, "c"), (1, 8192, 640, 17280, 3, "c"),
    (1, 8192, 640, 8640, 3, "c"), (1, 8192, 1280, 11520, 3, "c"), (1, 8192, 320, 2880, 3, "c"),
    (25, 8192, 640, 640, 1, "r"), (5, 8192, 1920, 640, 1, "l"), (5, 8192, 640, 640, 1, "l"), (5, 8192, 640, 2560, 1, "r"),
    (5, 8192, 5120, 640, 1, "g"),
    (7, 2048, 1280, 11520, 3, "c"), (2, 2048, 1280, 23040, 3, "c"), (1, 2048, 1280, 17280, 3, "c"), (1, 2048, 1280, 5760, 3, "c"),
    (1, 2048, 640, 5760, 3, "c"),
    (25, 2048, 1280, 1280, 1, "r"), (5, 2048, 3840, 1280, 1, "l"), (5, 2048, 1280, 1280, 1, "l"), (5, 2048, 1280, 5120, 1, "r"),
    (5, 2048, 10240, 1280, 1, "g"),
    (12, 512, 1280, 11520, 3, "c"), (3, 512, 1280, 23040, 3, "c"), (5, 512, 1280, 1280, 1, "r"), (1, 512, 3840, 1280, 1, "l"),
    (1, 512, 1280, 5120, 1, "r"), (1, 512, 10240, 1280, 1, "g"),
    # VAE decoder / large-M references
    (0, 4 * 512 * 512, 128, 1152, 3, "c"), (0, 4 * 256 * 256, 256, 2304, 3, "c"), (0, 4096, 4096, 4096, 1, "c"), (0, 8192, 8192, 8192, 1, "c"),
]
NCFG = lib().vd_gemm_num_configs()
SPLITS_SMALL = [0, 2, 3, 4, 6, 8, 12]   # 0 = the override's own fill-the-chip split


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def make_case(M, N, K, ks, kind):
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    if ks == 3:
        B = 8 if M >= 512 and M < (1 << 20) else 4
        side = int(round((M // B) ** 0.5))
        ci = K // 9
        x = torch.randn(B, side, side, ci, device=dev, dtype=torch.float16, generator=g)
        w = (torch.randn(N, K, device=dev, dtype=torch.float16, generator=g) * 0.02)
        b = torch.randn(N, device=dev, dtype=torch.float16, generator=g)
        return lambda split: ops.conv2d_nhwc(x, w, b, ksize=3, pad=1, split_k=split)
    a = torch.randn(M, K, device=dev, dtype=torch.float16, generator=g)
    w = torch.randn(N, K, device=dev, dtype=torch.float16, generator=g) * 0.02
    b = torch.randn(N, device=dev, dtype=torch.float16, generator=g)
    if kind == "g":
        return lambda split: ops.gemm(a, w, bias=b, act=ops.ACT_GEGLU)
    if kind == "l":
        cs = w.float().sum(1).contiguous()
        return lambda split: ops.gemm(a, w, bias=b, colsum=cs, ln_eps=1e-5)
    if kind == "r":
        r = torch.randn(M, N, device=dev, dtype=torch.float16, generator=g)
        return lambda split: ops.gemm(a, w, bias=b, res=r, split_k=split)
    return lambda split: ops.gemm(a, w, bias=b, split_k=split)


def main():
    filt = os.environ.get("VD_SWEEP_FILTER")
    names = [ops.gemm_kernel_name(c).replace("gemm_f16_kernel", "") for c in range(NCFG)]
    print("configs:", " | ".join("%d=%s" % (i, n) for i, n in enumerate(names)))
    results = []
    t_plan = t_best = 0.0
    for (cnt, M, N, K, ks, kind) in SHAPES:
        if filt and ("M=%d" % M) != filt:
            continue
        run = make_case(M, N, K, ks, kind)
        fl = 2.0 * M * N * K
        row = {"count": cnt, "M": M, "N": N, "K": K, "ks": ks, "kind": kind, "cfg_us": {}}
        ops.gemm_set_override(-1)
        try:
            row["plan_us"] = timeit(lambda: run(0))
        except Exception as e:  # noqa
            row["plan_us"] = None
            print("planner run failed:", e)
        best = (1e30, None)
        for c in range(NCFG):
            ops.gemm_set_override(c)
            splits = SPLITS_SMALL if (M <= 2048 and K >= 2048 and kind in ("c", "r")) else [0]
            for sp in splits:
                if sp > 1 and K // 64 // sp < 8:
                    continue
                try:
                    us = timeit(lambda: run(sp))
                except Exception:
                    continue
                row["cfg_us"]["%d/s%d" % (c, sp)] = us
                if us < best[0]:
                    best = (us, "%d/s%d" % (c, sp))
        ops.gemm_set_override(-1)
        row["best"] = best[1]
        row["best_us"] = best[0]
        results.append(row)
        per_cfg = {}
        for k, v in row["cfg_us"].items():
            c = int(k.split("/")[0])
            per_cfg[c] = min(per_cfg.get(c, 1e30), v)
        print("%-34s x%-2d plan %7.1f | %s | best %-7s %7.1f us %6.0f TF/s" % (
            "%s%d M=%d N=%d K=%d" % (kind, ks, M, N, K), cnt, row["plan_us"] or -1,
            " ".join("%d:%6.1f" % (c, per_cfg[c]) for c in sorted(per_cfg)), best[1], best[0], fl / best[0] / 1e6))
        sys.stdout.flush()
        if row["plan_us"]:
            t_plan += cnt * row["plan_us"]
        t_best += cnt * best[0]
    print("GEMM time per forward: planner %.3f ms, per-shape best %.3f ms" % (t_plan / 1e3, t_best / 1e3))
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump({"configs": names, "rows": results}, f, indent=1)


if __name__ == "__main__":
    main()
