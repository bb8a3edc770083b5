"""PCIe-inclusive hand-over of one decoded batch: the drop-in boundary returns DEVICE tensors (as the
reference's vae_decode does); app.py:283-ff then moves the images to the host for PIL.  Times that copy for the t2i batch
(4 x 3 x 512 x 512): fp16 planar as decoded, and u8 HWC after vd_image_to_u8, into pageable and pinned host memory."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
import torch
dev = torch.device("cuda:0")
for name, t in (("fp16 [4,3,512,512]", torch.rand(4, 3, 512, 512, device=dev).half()),
                ("u8 [4,512,512,3]", (torch.rand(4, 512, 512, 3, device=dev) * 255).to(torch.uint8)),
                ("fp16 [8,3,512,512]", torch.rand(8, 3, 512, 512, device=dev).half()),
                ("fp16 [4,3,768,768]", torch.rand(4, 3, 768, 768, device=dev).half())):
    pinned = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    for label, fn in (("pageable .cpu()", lambda: t.cpu()), ("pinned copy_", lambda: (pinned.copy_(t, non_blocking=True), torch.cuda.synchronize()))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print("%-20s %-16s %7.3f ms  %6.1f GB/s" % (name, label, dt * 1e3, t.numel() * t.element_size() / dt / 1e9))
