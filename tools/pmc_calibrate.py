"""Known-byte streams for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this box (WRITE_SIZE 'uncalibrated for
four rounds'): a 16-byte-per-lane fp16 copy of N bytes (this library's GroupNorm-apply kernel with a unit table: read N, write
N) and torch's own copy kernel, each launched 4 times.  Run once per counter:
    rocprofv3 --pmc FETCH_SIZE -- python tools/pmc_calibrate.py      rocprofv3 --pmc WRITE_SIZE -- python tools/pmc_calibrate.py
tools/pmc_summary.py prints the per-kernel counter means; bytes moved per launch are printed here."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "versatile-diffusion_amd"))
os.environ.setdefault("VD_QUIET", "1")
import torch
from vd_hip import ops
dev = torch.device("cuda:0")
B, HW, C = 8, 4096 * 4, 320          # 335 MB of fp16: beyond L2 (32 MB) and the Infinity Cache (256 MB) taken together
x = torch.randn(B, HW, C, device=dev, dtype=torch.float16)
table = torch.zeros(B, 2, C, device=dev, dtype=torch.float32)
table[:, 0] = 1.0
y = torch.empty_like(x)
z = torch.empty_like(x)
for _ in range(4):
    ops.gn_apply_table(x, table, silu=False, out=y)
    z.copy_(x)
torch.cuda.synchronize()
print("bytes per launch: read %d, written %d (gn_apply_table_kernel and torch's copy kernel alike)" % (x.numel() * 2, x.numel() * 2))
