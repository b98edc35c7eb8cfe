"""MI355X-native drop-in for the ray-tracing compute dispatch of chenxin030/Opengl_Raytracing.

Only the hot path lives here: ``csrc/`` (hand-written gfx950 HIP kernels + the C ABI declared in
``include/rt_mi355.h``) and the thin host-side mirror of the reference's dispatch site
(``host.RayTracer``), the reference's SSBO layouts (``layout``), the synthetic scenes of the
benchmark configs (``scenes``) and the multi-GPU strip tiling (``dist``).
"""
from . import layout  # noqa: F401

__all__ = ["layout", "host", "scenes", "dist", "build"]
