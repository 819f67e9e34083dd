"""quadraturefields_amd -- the quadrature-field volumetric render path of ubc-vision/quadraturefields,
rebuilt for AMD Instinct MI355X (gfx950): hand-written HIP kernels behind a C ABI (include/qf_hip.h), under
the reference's own Python entry points.  See DESIGN.md and INTEGRATION.md.

There is no CPU fallback: every compute entry point needs libqf_hip.so and a HIP device.
"""
from . import _C  # noqa: F401
from .datasets.utils import Rays, namedtuple_map  # noqa: F401

__all__ = ["Rays", "namedtuple_map"]
__version__ = "0.1.0"
