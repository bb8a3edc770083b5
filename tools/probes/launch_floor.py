"""Per-kernel cost of DEPENDENT trivial kernels replayed from a HIP graph: the launch floor every kernel of the captured
DDIM step pays (dispatch + the cache maintenance between dependent dispatches), measured with three kernel sizes."""
import torch
dev = torch.device("cuda:0")
for n in (1, 4096, 1 << 20):
    x = torch.zeros(n, device=dev, dtype=torch.float16)
    for _ in range(3):
        x.add_(1)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(400):
                x.add_(1)
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print("%8d elements: %.2f us per dependent kernel in a replayed graph" % (n, e0.elapsed_time(e1) * 1e3 / 4000))
