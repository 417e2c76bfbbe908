"""raytracer_amd -- MI355X (gfx950) backend for DevDuoItaly/Raytracer's TraceRay hot path.

Python here is plumbing around the C-ABI of include/rt_amd.h: device memory, torch.distributed for the row-band
gather, scene construction for tests and bench.  All pixels come from the hand-written HIP kernels in
raytracer_amd/csrc; nothing in this package imports or calls oracle/.
"""
from . import scenes  # noqa: F401
from .renderer import Renderer, camera35f  # noqa: F401
from ._abi import RtError, LIB_PATH  # noqa: F401

__all__ = ["Renderer", "camera35f", "scenes", "RtError", "LIB_PATH"]
