"""conv3x3_halo_kernel: every variant against torch's fp32 convolution on a set of geometries (correctness), then a timing
table old kernel (setting 0) vs each variant on the UNet's 3x3 shapes (warm, back to back: an upper bound on what the
forward sees; tools/halo_forward.py measures inside the graph-replayed forward).

    python tools/halo_check.py [check|time|all]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
import torch
import torch.nn.functional as F
from vd_hip import ops
from vd_hip.loader import lib
from vd_hip.pack import pack_conv_weight

dev = torch.device("cuda:0")
NV = 12
mode = sys.argv[1] if len(sys.argv) > 1 else "all"


def rnd(shape, scale, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev, torch.float16)


def ref_conv(x, w, b, ups):
    xx = x.float().permute(0, 3, 1, 2)
    if ups:
        xx = F.interpolate(xx, scale_factor=2, mode="nearest")
    return F.conv2d(xx, w.float(), b.float() if b is not None else None, padding=1).permute(0, 2, 3, 1).contiguous()


def rel(a, b):
    return float((a.float() - b).norm() / b.norm().clamp_min(1e-12))


def planned(x, w, x1, ups):
    """(kernel name, nsplit) the library would use"""
    import ctypes
    from vd_hip.loader import VdGemmDesc
    B, H, W, c0 = x.shape
    d = VdGemmDesc()
    Hv, Wv = H << ups, W << ups
    d.M, d.N, d.K = B * Hv * Wv, w.shape[0], w.shape[1]
    d.a0 = d.w = d.out = 16
    d.c0, d.c1 = c0, (x1.shape[-1] if x1 is not None else 0)
    if x1 is not None:
        d.a1 = 16
    d.Hin, d.Win, d.Hout, d.Wout, d.ksize, d.stride, d.pad, d.ups = H, W, Hv, Wv, 3, 1, 1, ups
    d.ws = 16
    cfg, ns = ctypes.c_int(-1), ctypes.c_int(-1)
    assert lib().vd_gemm_plan(ctypes.byref(d), ctypes.byref(cfg), ctypes.byref(ns)) == 0
    return ops.gemm_kernel_name(cfg.value), ns.value


CHECKS = [
    # B, H, W, c0, c1, Cout, ups, rowvec, res
    (2, 32, 32, 64, 0, 160, 0, False, False),
    (2, 32, 32, 128, 64, 320, 0, True, False),
    (1, 64, 64, 320, 0, 320, 0, False, True),
    (2, 16, 16, 192, 0, 320, 0, True, True),
    (4, 8, 8, 128, 0, 160, 0, False, False),
    (8, 8, 8, 256, 128, 320, 0, True, True),
    (2, 16, 16, 128, 0, 128, 1, False, False),
    (1, 96, 96, 64, 0, 160, 0, False, True),
    (1, 48, 48, 64, 64, 128, 0, False, False),
    (1, 128, 128, 128, 0, 256, 0, False, True),
    (8, 16, 16, 640, 0, 1280, 0, True, False),      # split over chunks
    (2, 64, 32, 64, 0, 72, 0, False, False),         # N not a multiple of the tile
]

if mode in ("check", "all"):
    bad = 0
    for ci, (B, H, W, c0, c1, Co, ups, rv, rs) in enumerate(CHECKS):
        x = rnd((B, H, W, c0), 1.0, 100 + ci)
        x1 = rnd((B, H, W, c1), 1.0, 200 + ci) if c1 else None
        wt = rnd((Co, c0 + c1, 3, 3), 0.04, 300 + ci)
        b = rnd((Co,), 0.3, 400 + ci)
        Hv, Wv = H << ups, W << ups
        rowvec = rnd((B, Co), 0.5, 500 + ci) if rv else None
        res = rnd((B, Hv, Wv, Co), 1.0, 600 + ci) if rs else None
        ref = ref_conv(torch.cat([x, x1], -1) if c1 else x, wt, b, ups)
        if rv:
            ref = ref + rowvec.float().view(B, 1, 1, Co)
        if rs:
            ref = ref + res.float()
        wp = pack_conv_weight(wt)
        kw = dict(ksize=3, pad=1, ups=ups, x1=x1)
        if rv:
            kw.update(rowvec=rowvec, rows_per_batch=Hv * Wv)
        if rs:
            kw.update(res=res)
        line = "case %2d B%d %dx%d c%d+%d ->%d ups%d rv%d res%d:" % (ci, B, H, W, c0, c1, Co, ups, rv, rs)
        for v in range(0, NV + 1):
            assert lib().vd_conv_halo_set_variant(v) == 0
            name, ns = planned(x, wp, x1, ups)
            out = ops.conv2d_nhwc(x, wp, b, **kw)
            torch.cuda.synchronize()
            e = rel(out, ref)
            tag = "H" if name.startswith("conv3x3") else "g"
            ok = e < 2e-3 and bool(torch.isfinite(out).all())
            bad += 0 if ok else 1
            line += " v%d%s%d=%.1e%s" % (v, tag, ns, e, "" if ok else "(BAD)")
        print(line, flush=True)
    lib().vd_conv_halo_set_variant(-1)
    print("halo_check: %d bad results" % bad, flush=True)

SHAPES = [
    # B, H, W, c0, c1, Cout, ups   (the UNet's 3x3 convolutions at CFG batch 8; count per forward in the comment)
    (8, 64, 64, 320, 0, 320, 0),      # x7
    (8, 64, 64, 320, 320, 320, 0),    # x2
    (8, 64, 64, 640, 320, 320, 0),    # x1
    (8, 32, 32, 640, 0, 640, 1),      # x1 upsample to 64x64
    (8, 32, 32, 320, 0, 640, 0),      # x1
    (8, 32, 32, 640, 0, 640, 0),      # x6
    (8, 32, 32, 640, 640, 640, 0),    # x1
    (8, 32, 32, 1280, 640, 640, 0),   # x1
    (8, 32, 32, 640, 320, 640, 0),    # x1
    (8, 16, 16, 1280, 0, 1280, 1),    # x1 upsample to 32x32
    (8, 16, 16, 640, 0, 1280, 0),     # x1
    (8, 16, 16, 1280, 0, 1280, 0),    # x7
    (8, 16, 16, 1280, 1280, 1280, 0), # x2
    (8, 16, 16, 1280, 640, 1280, 0),  # x1
    (8, 8, 8, 1280, 0, 1280, 0),      # x12
    (8, 8, 8, 1280, 1280, 1280, 0),   # x3
    (8, 8, 8, 1280, 0, 1280, 1),      # x1 upsample to 16x16
    (4, 64, 64, 320, 0, 320, 0),      # shared CFG replica blocks
    (4, 64, 64, 512, 0, 512, 0),      # VAE
    (4, 128, 128, 512, 0, 512, 0),    # VAE
    (4, 256, 256, 256, 0, 256, 0),    # VAE
]

if mode in ("time", "all"):
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)   # larger than L2 + MALL
    print("%-44s %s" % ("shape", " ".join("%9s" % ("v%d" % v) for v in range(0, NV + 1))), flush=True)
    for (B, H, W, c0, c1, Co, ups) in SHAPES:
        x = rnd((B, H, W, c0), 1.0, 1)
        x1 = rnd((B, H, W, c1), 1.0, 2) if c1 else None
        wp = pack_conv_weight(rnd((Co, c0 + c1, 3, 3), 0.03, 3))
        b = rnd((Co,), 0.3, 4)
        Hv, Wv = H << ups, W << ups
        gflop = 2.0 * B * Hv * Wv * Co * 9 * (c0 + c1) / 1e9
        cells = []
        for cold in (0, 1):
            row = []
            for v in range(0, NV + 1):
                lib().vd_conv_halo_set_variant(v)
                for _ in range(2):
                    ops.conv2d_nhwc(x, wp, b, ksize=3, pad=1, ups=ups, x1=x1)
                torch.cuda.synchronize()
                n = 6 if cold else 20
                tot = 0.0
                if cold:
                    for _ in range(n):
                        flush.fill_(1)      # evict weights / activations: the forward streams weights cold
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        ops.conv2d_nhwc(x, wp, b, ksize=3, pad=1, ups=ups, x1=x1)
                        e1.record()
                        torch.cuda.synchronize()
                        tot += e0.elapsed_time(e1)
                else:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(n):
                        ops.conv2d_nhwc(x, wp, b, ksize=3, pad=1, ups=ups, x1=x1)
                    e1.record()
                    torch.cuda.synchronize()
                    tot = e0.elapsed_time(e1)
                row.append(tot / n * 1e3)
            cells.append(row)
        tag = "B%d %dx%d %d+%d->%d u%d %.0fGF" % (B, H, W, c0, c1, Co, ups, gflop)
        print("%-44s %s" % (tag + " warm us", " ".join("%9.1f" % t for t in cells[0])), flush=True)
        print("%-44s %s" % ("   cold us", " ".join("%9.1f" % t for t in cells[1])), flush=True)
        print("%-44s %s" % ("   warm TF/s", " ".join("%9.0f" % (gflop / t * 1e3) for t in cells[0])), flush=True)
    lib().vd_conv_halo_set_variant(-1)
