"""L2 -> LDS DMA bandwidth with every CU streaming (see dma_probe.hip).  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC
tools/probes/dma_probe.hip -o tools/probes/dma_probe.so"""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "dma_probe.so"))
L.dma_probe.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
buf = torch.randint(0, 255, (256 << 20,), dtype=torch.uint8, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(region, blocks, pieces, mode, depth, skew):
    for _ in range(2):
        L.dma_probe(buf.data_ptr(), region, blocks, pieces, mode, depth, skew, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        L.dma_probe(buf.data_ptr(), region, blocks, pieces, mode, depth, skew, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    return blocks * 8 * pieces * 1024 / ms / 1e9   # TB/s
print("mode 0 = every block the same region, mode 1 = own region per block; TB/s")
for mode, region in ((0, 16 << 10), (0, 128 << 10), (0, 2 << 20), (0, 32 << 20), (1, 128 << 10), (1, 1 << 20)):
    for depth in (2, 4, 8):
        for blocks in (256, 512):
            for skew in (0, 1):
                print("mode %d region %8d depth %d blocks %d skew %d: %6.2f TB/s" % (mode, region, depth, blocks, skew, run(region, blocks, 4096, mode, depth, skew)), flush=True)
