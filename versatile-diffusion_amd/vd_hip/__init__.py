"""ctypes binding of libvd_hip.so (include/vd_hip.h) -- the only compute backend of this package.

There is deliberately no CPU / eager-PyTorch fallback: if the HIP library is missing or the tensors
are not on a gfx950 device the ops raise `VdHipError`.  torch is used for device memory, streams and
views only.
"""
from .loader import VdHipError, lib, lib_path, available  # noqa: F401
from . import ops  # noqa: F401
