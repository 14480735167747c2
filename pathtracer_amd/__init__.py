"""pathtracer_amd — MI355X-native rebuild of the DXR path-tracing hot path of domme/PathTracer.

Layout: csrc/ (HIP kernels, host BVH builder, C-ABI -> lib/libpathtracer_amd.so), api.py/context.py (ctypes binding of
include/pt_api.h), pathtracer.py (host mirror of the reference's PathTracer / Sky / Camera classes), scenes.py (OBJ
reader, Cornell fixture, procedural stand-in scenes), distributed.py (sample-partitioned multi-GPU render).
"""
from . import api  # noqa: F401
from .context import Context, SceneData  # noqa: F401
from .pathtracer import Camera, PathTracer, Sky, SetupEarthAtmosphere  # noqa: F401
