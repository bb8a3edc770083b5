"""Minimal logging shim with the reference's `print_log` name (lib/log_service.py:15-36 there).

The reference derives a local rank from torch.cuda.device_count() and divides by it (ZeroDivisionError on a
0-GPU host); here rank comes from the launcher environment (one process per GPU under torchrun)."""
import os


def _rank():
    try:
        return int(os.environ.get("RANK", "0"))
    except ValueError:
        return 0


def print_log(*console_info):
    if _rank() != 0 or os.environ.get("VD_QUIET", "0") == "1":
        return
    print(" ".join(str(i) for i in console_info))
