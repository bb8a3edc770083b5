import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "versatile-diffusion_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def dev():
    import torch
    return torch.device("cuda:0")


@pytest.fixture(scope="session", autouse=True)
def _pin_oracle_threads():
    """The fp32 CPU oracle is what the long GPU parity tests wait for.  On the GPU boxes' 256-thread hosts torch's default
    thread count oversubscribes it (one CFG-batch-2 UNet forward: 4.2 s on 16 threads, 6.3 s on 64, 11.2 s on 128 --
    measured through bench.py --cpu-baseline-only), so the oracle runs on 16 pinned threads there."""
    if _has_gpu():
        import torch
        n = int(os.environ.get("VD_CPU_THREADS", "16"))
        torch.set_num_threads(max(1, min(n, os.cpu_count() or 1)))
    yield
