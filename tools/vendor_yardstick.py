"""Vendor yardstick (tools only, never product): the vendor's fp16 GEMM (hipBLASLt / rocBLAS behind torch.matmul) next to
this library's kernels on the SAME box, warm, at 8192^3 and at the UNet's GEMM-equivalent shapes.

    python tools/vendor_yardstick.py            timing table (best of 3 x 30 launches, HIP events on the launch stream)
    python tools/vendor_yardstick.py once       every problem launched 4x by each side: target of `rocprofv3 --kernel-trace`
                                                (kernel names -> the vendor's tile / MFMA shape) and of `--pmc GRBM_GUI_ACTIVE`
                                                (effective clock = GRBM_GUI_ACTIVE / kernel duration)

Rows: (M, N, K, what the shape is in the forward).  `conv` rows also run the 3x3 convolution itself on conv3x3_halo_kernel
(the vendor side is the plain GEMM of the same M x N x K: no im2col, i.e. the yardstick is favourable to the vendor)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
os.environ.setdefault("VD_QUIET", "1")
import torch
from vd_hip import ops, pack

dev = torch.device("cuda:0")
CASES = [
    (8192, 8192, 8192, "square", None),
    (32768, 320, 2880, "conv 64^2 320->320", (8, 64, 64, 320, 320)),
    (32768, 320, 5760, "conv 64^2 640->320", (8, 64, 64, 640, 320)),
    (8192, 640, 5760, "conv 32^2 640->640", (8, 32, 32, 640, 640)),
    (2048, 1280, 11520, "conv 16^2 1280->1280", (8, 16, 16, 1280, 1280)),
    (8192, 5120, 640, "GEGLU proj 32^2", None),
    (2048, 10240, 1280, "GEGLU proj 16^2", None),
    (8192, 640, 640, "CxC 32^2", None),
    (2048, 1280, 1280, "CxC 16^2", None),
    (32768, 320, 320, "CxC 64^2", None),
    (2048, 1280, 5120, "FF out 16^2", None),
    (8192, 640, 2560, "FF out 32^2", None),
    (8192, 1920, 640, "q|k|v 32^2", None),
    (2048, 3840, 1280, "q|k|v 16^2", None),
]
once = len(sys.argv) > 1 and sys.argv[1] == "once"


def timeit(fn, n=30, reps=3):
    if once:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        return float("nan")
    best = 1e9
    for _ in range(reps):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


print("device: %s   torch %s   hip %s" % (torch.cuda.get_device_name(0), torch.__version__, torch.version.hip))
print("%-24s %6s %6s %6s | %9s %7s | %9s %7s | %9s %7s | %s" % ("shape", "M", "N", "K", "vendor us", "TF/s", "vd_gemm us", "TF/s",
                                                                 "vd conv us", "TF/s", "vendor/vd"))
for (M, N, K, what, conv) in CASES:
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn(M, K, device=dev, dtype=torch.float16, generator=g)
    w = torch.randn(N, K, device=dev, dtype=torch.float16, generator=g) * 0.02
    fl = 2.0 * M * N * K
    out_v = torch.empty(M, N, device=dev, dtype=torch.float16)
    wt = w.t()
    t_v = timeit(lambda: torch.matmul(a, wt, out=out_v))            # NT problem, as this library sees it: W stored [N, K]
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    t_g = timeit(lambda: ops.gemm(a, w, out=out))
    t_c = float("nan")
    if conv is not None:
        B, H, W_, Ci, Co = conv
        x = torch.randn(B, H, W_, Ci, device=dev, dtype=torch.float16, generator=g)
        wc = torch.randn(Co, 9 * Ci, device=dev, dtype=torch.float16, generator=g) * 0.02
        t_c = timeit(lambda: ops.conv2d_nhwc(x, wc, None, ksize=3, stride=1, pad=1))
    if not once:
        err = float((out.float() - out_v.float()).norm() / out_v.float().norm())
        best_vd = min(t_g, t_c) if t_c == t_c else t_g
        print("%-24s %6d %6d %6d | %9.1f %7.0f | %9.1f %7.0f | %9.1f %7.0f | %5.2f   (rel diff %.1e)" % (
            what, M, N, K, t_v, fl / t_v / 1e6, t_g, fl / t_g / 1e6, t_c, (fl / t_c / 1e6) if t_c == t_c else float("nan"),
            best_vd / t_v, err))
