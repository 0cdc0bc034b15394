"""MI355X-native visual-place-recognition + geopose inference hot path.

DINOv2 (PyTorch-ROCm plumbing) -> SALAD optimal-transport aggregation -> bf16 cosine kNN against a
(sharded) gallery -> (lat, lon, sin, cos) regression head, the last three as hand-written gfx950
HIP kernels behind the C ABI of include/vpr_amd.h (libvpr_amd.so).  Import as `vpr_amd`.

Nothing here falls back to a CPU path: every op raises if the HIP library is missing or the
tensors are not on a GPU.  The CPU restatement used by the tests lives in oracle/ and is never
imported from this package.
"""
__version__ = "0.1.0"

import os as _os

# ~300 kernel launches per pipeline step: kernel arguments in device memory save ~1 us of dispatch latency each
# (bench.py: 11.74 -> 11.41 ms per step).  Takes effect only if the HIP runtime has not initialised yet.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from . import _lib  # noqa: F401  (ctypes binding; loads lazily)
from .build import build_library, library_path  # noqa: F401
