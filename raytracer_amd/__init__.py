"""raytracer_amd -- MI355X (gfx950) backend for the ray-scene query path of a CPU ray tracer:
octree Build/Search plus the ray/triangle test, serving primary, secondary, shadow and area-light
rays.  The compute lives in raytracer_amd/lib/libraytracer_amd.so (HIP, C ABI in
include/raytracer_amd.h); this package is the thin host-side mirror of the reference's interfaces."""
from . import _capi  # noqa: F401
from .ads import Ads, GpuOctree, ads_kind_from_name, make_render_config  # noqa: F401

__all__ = ["Ads", "GpuOctree", "ads_kind_from_name", "make_render_config"]
