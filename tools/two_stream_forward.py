"""Experiment: the CFG batch of one UNet forward as TWO independent half-batch chains on two streams inside one HIP graph
(fork / join), against the single full-batch chain.  Idea: with one block per CU every launch runs its load / compute / store
phases in lock-step across the chip; two concurrent half-size launches desynchronise them.  (dev tool, MI355X box)

    python tools/two_stream_forward.py [nsplit]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
os.environ.setdefault("VD_QUIET", "1")
import torch
import bench
from vd_hip import ops
dev = torch.device("cuda:0")
net = bench.build_model(dev)
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 8
x = torch.randn(B, 4, 64, 64, device=dev, dtype=torch.float16)
t = torch.full((B,), 501, device=dev, dtype=torch.long)
c = torch.randn(B, 77, 768, device=dev, dtype=torch.float16) * 0.5


def timeit(g):
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    return best


with torch.no_grad():
    ci = {"type": "text", "c": c, "kv_cache": {}}
    for _ in range(2):
        full = net.apply_model({"type": "image", "x": x}, t, ci)
    torch.cuda.synchronize()
    g1 = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g1, stream=s):
            out1 = net.apply_model({"type": "image", "x": x}, t, ci)
    torch.cuda.current_stream().wait_stream(s)
    print("one chain of batch %d: %.3f ms" % (B, timeit(g1)), flush=True)

    hb = B // NS
    cis = [{"type": "text", "c": c[i * hb:(i + 1) * hb].contiguous(), "kv_cache": {}} for i in range(NS)]
    xs = [x[i * hb:(i + 1) * hb].contiguous() for i in range(NS)]
    ts = [t[i * hb:(i + 1) * hb].contiguous() for i in range(NS)]
    side = [torch.cuda.Stream() for _ in range(NS - 1)]
    for i in range(NS):   # warm the per-stream workspaces / caches outside the capture
        net.apply_model({"type": "image", "x": xs[i]}, ts[i], cis[i])
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    s2 = torch.cuda.Stream()
    s2.wait_stream(torch.cuda.current_stream())
    outs = [None] * NS
    with torch.cuda.stream(s2):
        with torch.cuda.graph(g2, stream=s2):
            for st in side:
                st.wait_stream(s2)
            outs[0] = net.apply_model({"type": "image", "x": xs[0]}, ts[0], cis[0])
            for i, st in enumerate(side):
                with torch.cuda.stream(st):
                    outs[i + 1] = net.apply_model({"type": "image", "x": xs[i + 1]}, ts[i + 1], cis[i + 1])
            for st in side:
                s2.wait_stream(st)
    torch.cuda.current_stream().wait_stream(s2)
    print("%d concurrent chains of batch %d: %.3f ms" % (NS, hb, timeit(g2)), flush=True)
    g1.replay(); g2.replay(); torch.cuda.synchronize()
    o2 = torch.cat(outs)
    print("rel-L2 between the two: %.3e" % float((o2.float() - out1.float()).norm() / out1.float().norm()))
