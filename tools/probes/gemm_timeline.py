"""Per-block phase timeline of the short-K projections (development probe, round 4).

    VD_BUILD_OUT=$PWD/versatile-diffusion_amd/libvd_hip_tl.so VD_EXTRA_DEFS=-DVD_TIMELINE python versatile-diffusion_amd/build.py
    VD_HIP_LIB=$PWD/versatile-diffusion_amd/libvd_hip_tl.so python tools/probes/gemm_timeline.py

The -DVD_TIMELINE build of gemm_f16_kernel / rowgemm320_kernel keeps s_memrealtime stamps (100 MHz) of a block's phase
boundaries in scalar registers and stores them at the very end: 0 start, 1 first K tile landed, 2 main loop done, 3 epilogue
tile in LDS, 4 output stores issued, 6 stores acknowledged, 7 XCC id.  This script runs the C x C projections of the three
transformer levels (to_out / proj_out: + bias + residual) and prints where a block's life goes and how the blocks of a launch
are spread over the launch's span -- the 6.7-GFLOP launches take 22-25 us in the forward against ~3 us of MFMA time.
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
import torch  # noqa: E402
from vd_hip import ops  # noqa: E402
from vd_hip.loader import lib  # noqa: E402

dev = torch.device("cuda:0")
TICK_US = 0.01   # s_memrealtime: 100 MHz


def rnd(shape, scale, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.float16).to(dev)


def analyse(tl, name, ev_us):
    import numpy as np
    t = tl.cpu().numpy().astype("int64")
    t = t[t[:, 0] != 0]
    if t.shape[0] == 0:
        print("%s: no stamps (a kernel without them took the launch)" % name)
        return
    if os.environ.get("TL_EPI") != "1":
        t[:, 1] = np.maximum(t[:, 1], t[:, 0]); t[:, 2] = np.maximum(t[:, 2], t[:, 1]); t[:, 3] = np.maximum(t[:, 3], t[:, 2])
    t0 = t[:, 0].min()
    span = (t[:, 6].max() - t0) * TICK_US
    ph = {"start offset": t[:, 0] - t0, "prologue (start -> tile 0 landed)": t[:, 1] - t[:, 0], "K loop": t[:, 2] - t[:, 1],
          "epilogue regs -> LDS": t[:, 3] - t[:, 2], "epilogue LDS -> stores issued": t[:, 4] - t[:, 3],
          "stores acknowledged": t[:, 6] - t[:, 4], "block life": t[:, 6] - t[:, 0]}
    print("%s: %d blocks, span %.2f us (event timing %.2f us per launch)" % (name, t.shape[0], span, ev_us))
    if os.environ.get("TL_EPI") == "1":   # -DVD_TIMELINE_EPI build of conv3x3_halo_kernel: slot 1 = epilogue operands arrived, slot 5 = part 2 done
        ph = {"main loop (start -> done)": t[:, 2] - t[:, 0], "epilogue operands arrive": t[:, 1] - t[:, 2], "regs -> LDS tile": t[:, 3] - t[:, 1],
              "part 2 (LDS -> stores issued)": t[:, 5] - t[:, 3], "statistics pass": t[:, 4] - t[:, 5], "stores acknowledged": t[:, 6] - t[:, 4]}
        for k, v in ph.items():
            v = v * TICK_US
            print("    %-36s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % (k, v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
        return
    for k, v in ph.items():
        v = v * TICK_US
        print("    %-36s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % (k, v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
    # how many blocks are alive over the span (10 samples)
    edges = np.linspace(0, span / TICK_US, 11)
    alive = [int(((t[:, 0] - t0 <= e) & (t[:, 6] - t0 > e)).sum()) for e in edges[:-1]]
    print("    blocks alive at 0, 10, .. 90 %% of the span: %s" % alive)


def run(name, fn, nblocks_hint, setter, flush=None, reps=6):
    tl = torch.zeros((nblocks_hint, 8), dtype=torch.int64, device=dev)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fn()
    e.record()
    torch.cuda.synchronize()
    ev = s.elapsed_time(e) * 1000.0 / 20
    setter(ctypes.c_void_p(tl.data_ptr()))
    for _ in range(reps):
        if flush is not None:
            flush.add_(1.0)
        fn()
    torch.cuda.synchronize()
    setter(ctypes.c_void_p(0))
    analyse(tl, name + (" [cold: 512 MB written in front]" if flush is not None else " [back to back]"), ev)


def main():
    h = lib()
    h.vd_debug_set_timeline.restype = None
    h.vd_debug_set_timeline.argtypes = [ctypes.c_void_p]
    h.vd_debug_set_timeline_row320.restype = None
    h.vd_debug_set_timeline_row320.argtypes = [ctypes.c_void_p]
    flush = torch.zeros(128 * 1024 * 1024, dtype=torch.float32, device=dev)
    for (M, C) in (() if os.environ.get("TL_ONLY", "") == "halo" else ((32768, 320), (8192, 640), (2048, 1280))):
        a = rnd((M, C), 1.0, 1)
        w = rnd((C, C), 0.05, 2)
        b = rnd((C,), 0.1, 3)
        r = rnd((M, C), 1.0, 4)
        fn = lambda: ops.gemm(a, w, bias=b, res=r)   # noqa: E731
        for fl in (None, flush):
            run("gemm_f16_kernel M=%d N=K=%d +bias +res" % (M, C), fn, 8192, h.vd_debug_set_timeline, fl)
        if C == 320:
            fn2 = lambda: ops.gemm_row320(a, w, b, r)   # noqa: E731
            for fl in (None, flush):
                run("rowgemm320_kernel M=%d N=K=320 +bias +res" % M, fn2, 8192, h.vd_debug_set_timeline_row320, fl)
    if os.environ.get("TL_ONLY", "") == "halo" or os.environ.get("TL_HALO", "1") != "0":
        from vd_hip.pack import pack_conv_weight
        for (B, H, C) in ((8, 64, 320), (8, 32, 640), (8, 16, 1280)):
            x = rnd((B, H, H, C), 1.0, 11)
            w = pack_conv_weight(rnd((C, C, 3, 3), 0.02, 12))
            b = rnd((C,), 0.1, 13)
            rv = rnd((B, C), 0.3, 14)
            r = rnd((B, H, H, C), 1.0, 15)
            fn = lambda: ops.conv2d_nhwc(x, w, b, ksize=3, pad=1, rowvec=rv, rows_per_batch=H * H, want_stats=True)   # noqa: E731
            run("conv3x3_halo_kernel %dx%d C=%d +bias +rowvec +stats" % (H, H, C), fn, 4096, h.vd_debug_set_timeline, None)
            fn = lambda: ops.conv2d_nhwc(x, w, b, ksize=3, pad=1, res=r, want_stats=True)   # noqa: E731
            run("conv3x3_halo_kernel %dx%d C=%d +bias +res +stats" % (H, H, C), fn, 4096, h.vd_debug_set_timeline, flush)
        if os.environ.get("TL_ONLY", "") == "halo":
            return
    # a long-K reference: the 16x16-level skip-free 1x1 (K = 2560) and the plain FF-sized projection
    for (M, N, K) in ((8192, 640, 2560), (8192, 5120, 640), (2048, 1280, 5120)):
        a = rnd((M, K), 1.0, 5)
        w = rnd((N, K), 0.03, 6)
        b = rnd((N,), 0.1, 7)
        fn = lambda: ops.gemm(a, w, bias=b)   # noqa: E731
        run("gemm_f16_kernel M=%d N=%d K=%d +bias" % (M, N, K), fn, 16384, h.vd_debug_set_timeline, None)


if __name__ == "__main__":
    main()
