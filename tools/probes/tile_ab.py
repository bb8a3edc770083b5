"""One GEMM shape under every instantiated tile (development probe): us per launch, warm, back to back.
    python tools/probes/tile_ab.py M N K [M N K ...]
MODE=geglu: the LayerNorm-folded GEGLU projection (N = 2 x hidden; value * GELU(gate) epilogue, row statistics from a buffer);
MODE=qkv: LayerNorm-folded plain projection (no residual)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
import torch
from vd_hip import ops
from vd_hip.loader import lib
dev = torch.device("cuda:0")


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


args = [int(v) for v in sys.argv[1:]]
for i in range(0, len(args), 3):
    M, N, K = args[i:i + 3]
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.02
    b = torch.randn(N, device=dev, dtype=torch.float16)
    r = torch.randn(M, N, device=dev, dtype=torch.float16)
    row = []
    mode = os.environ.get("MODE", "")
    cs = w.float().sum(1).contiguous()
    fn = lambda: ops.gemm(a, w, bias=b, res=r)
    if mode == "geglu":
        fn = lambda: ops.linear(a, w, b, act=ops.ACT_GEGLU, colsum=cs, ln_eps=1e-5)
    elif mode == "qkv":
        fn = lambda: ops.linear(a, w, b, colsum=cs, ln_eps=1e-5)
    for cfg in [-1] + list(range(lib().vd_gemm_num_configs())):
        if lib().vd_gemm_set_override(cfg) != 0:
            continue
        try:
            us = timeit(fn)
            row.append("%s=%.1f" % ("auto" if cfg < 0 else cfg, us))
        except Exception as e:
            row.append("%d=err" % cfg)
    lib().vd_gemm_set_override(-1)
    print("M=%d N=%d K=%d: %s" % (M, N, K, "  ".join(row)), flush=True)
