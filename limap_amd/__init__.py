"""limap_amd -- MI355X-native backend for the line-triangulation hot path of cvg/limap.

    from limap_amd import triangulation
    tri = triangulation.GlobalLineTriangulator(cfg["triangulation"])   # same API as limap.triangulation

The HIP extension (limap_amd/liblimap_amd.so) must be built (``limap_amd.build.build_extension()``)
and a HIP device must be visible; there is no CPU fallback.
"""
from . import base, synthetic  # noqa: F401

__all__ = ["base", "synthetic", "triangulation", "build"]


def __getattr__(name):  # lazy: importing the package must not require the GPU library
    if name in ("triangulation", "build", "_capi", "dist", "io"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
