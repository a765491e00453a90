"""MI355X-native wavefront path tracer: the hot path of CobaltCrabby/ray_tracer
(shaders/raytrace.comp) as hand-written HIP kernels for gfx950 behind a C ABI
(include/rt_amd.h), plus the reference's host scene surface."""
from . import _capi  # noqa: F401
from .engine import (ASSET_DIR, Renderer, RtError, Scene, default_material, hits_to_numpy, placement,  # noqa: F401
                     push_constants)

__all__ = ["ASSET_DIR", "Renderer", "RtError", "Scene", "default_material", "hits_to_numpy", "placement",
           "push_constants"]
