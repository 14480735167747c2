"""ctypes binding of include/pt_api.h (libpathtracer_amd.so).

This is plumbing only: every structure here is a field-for-field mirror of the C header, which in turn mirrors the
reference's constant-buffer / scene layouts (PathTracer/PathTracer.cpp:171-339,689-740; Sky.h:27-62).
The library is the HIP product; there is no CPU fallback — if the shared object is missing the import fails.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PT_AMD_LIB: tuning experiments only (tools/): load another build of the SAME library (e.g. compiled with -DPT_EXP_*)
LIB_PATH = os.environ.get("PT_AMD_LIB") or os.path.join(_HERE, "lib", "libpathtracer_amd.so")

PT_OK = 0
PT_ERR_INVALID_ARG = 1
PT_ERR_NO_DEVICE = 2
PT_ERR_HIP = 3
PT_ERR_NO_SCENE = 4
PT_ERR_NO_TARGET = 5
PT_ERR_SIZE_MISMATCH = 6
PT_ERR_NO_SKY = 7
PT_ERR_OUT_OF_MEMORY = 8
PT_ERR_UNSUPPORTED = 9

PT_MISS = 0xFFFFFFFF
PT_SCENE_DEFAULT = 0
PT_SCENE_TRANSFORM_NORMALS = 1
PT_SCENE_PAIR_TRIANGLES = 2
PT_SCENE_NO_PAIRS = 4
PT_SCENE_FAST_BUILD = 8

PT_OPT_COUNT_TRAVERSAL = 1
PT_OPT_TIME_KERNELS = 2
PT_OPT_SAMPLES_PER_BATCH = 3
PT_OPT_SKY_MODE = 4
PT_OPT_RUSSIAN_ROULETTE = 10
PT_OPT_PIPELINE = 11
PT_OPT_FUSED_PRIMARY = 14
PT_OPT_PRIMARY_PACKET = 15
PT_OPT_FRAME_LANES = 16
PT_OPT_SHARED_TRIANGLE_PHASE = 17

# include/pt_api_tuning.h: schedule knobs of the traversal kernel, set through pt_debug_set_tuning (tools and tests only)
PT_TUNE_EXTEND_FETCH_MIN = 5
PT_TUNE_EXTEND_POSTPONE_PCT = 6
PT_TUNE_EXTEND_BLOCKS_PER_CU = 7
PT_TUNE_EXTEND_LDS_STACK = 8
PT_TUNE_EXTEND_DEFER = 9
PT_TUNE_EXTEND_TOP_NODES = 12
PT_TUNE_EXTEND_CONTINUE_PCT = 13
PT_TUNE_SHARED_DEFER = 18
PT_TUNE_SHARED_TOP_NODES = 19
PT_TUNE_SHARED_POSTPONE_PCT = 20
PT_TUNE_SHARED_CONTINUE_PCT = 21

(PT_FN_RAND01, PT_FN_FRESNEL, PT_FN_SPEC_PROB, PT_FN_SAMPLE_PHONG, PT_FN_EVAL_PHONG, PT_FN_COS_HEMISPHERE,
 PT_FN_LAMBERT, PT_FN_SKY, PT_FN_PRIMARY_RAY, PT_FN_SKY_VIEW_LUT, PT_FN_GGX_SAMPLE, PT_FN_ALBEDO_PATTERN) = range(12)
FN_IN_WIDTH = [4, 6, 4, 6, 10, 5, 9, 6, 4, 6, 14, 4]
FN_OUT_WIDTH = [8, 1, 1, 4, 1, 3, 4, 3, 6, 3, 8, 1]

TLUT_W, TLUT_H = 256, 64


class PtAtmosphere(C.Structure):
    _fields_ = [
        ("RayleighScattering", C.c_float * 3), ("RayleighDensityExpScale", C.c_float),
        ("AbsorptionExtinction", C.c_float * 3), ("BottomRadius", C.c_float),
        ("GroundAlbedo", C.c_float * 3), ("TopRadius", C.c_float),
        ("MieScattering", C.c_float * 3), ("MieDensityExpScale", C.c_float),
        ("MieExtinction", C.c_float * 3), ("MiePhaseG", C.c_float),
        ("MieAbsorption", C.c_float * 3), ("AbsorptionDensity0LayerWidth", C.c_float),
        ("AbsorptionDensity0ConstantTerm", C.c_float), ("AbsorptionDensity0LinearTerm", C.c_float),
        ("AbsorptionDensity1ConstantTerm", C.c_float), ("AbsorptionDensity1LinearTerm", C.c_float),
    ]


class PtSkyConstants(C.Structure):
    _fields_ = [
        ("myAtmosphere", PtAtmosphere),
        ("mySunDirection", C.c_float * 3), ("myTransmissionLutTexIdx", C.c_uint32),
        ("mySunIlluminance", C.c_float * 3), ("_unused", C.c_float),
        ("myRayMarchMinMaxSPP", C.c_float * 2), ("_unused2", C.c_float * 2),
    ]


class PtRtConsts(C.Structure):
    _fields_ = [
        ("myNearPlaneCorner", C.c_float * 3), ("myAoDistance", C.c_float),
        ("myXAxis", C.c_float * 3), ("myOutTexIndex", C.c_uint32),
        ("myYAxis", C.c_float * 3), ("myAsIndex", C.c_uint32),
        ("myCameraPos", C.c_float * 3), ("myInstanceDataBufferIndex", C.c_uint32),
        ("myMaterialDataBufferIndex", C.c_uint32), ("mySampleBufferIndex", C.c_uint32),
        ("myFrameRandomSeed", C.c_uint32), ("myNumAccumulationFrames", C.c_uint32),
        ("myLinearClampSamplerIndex", C.c_uint32), ("myMaxRecursionDepth", C.c_uint32),
        ("myLightInstanceId", C.c_uint32), ("myNumHaltonSamples", C.c_uint32),
        ("myLightEmission", C.c_float * 3), ("mySampleSky", C.c_uint32),
        ("mySkyFallbackEmission", C.c_float * 3), ("myPhongSpecularPower", C.c_float),
        ("mySkyConsts", PtSkyConstants),
    ]


assert C.sizeof(PtAtmosphere) == 112
assert C.sizeof(PtSkyConstants) == 160
assert C.sizeof(PtRtConsts) == 288


class PtMeshDesc(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("num_vertices", C.c_uint32),
                ("indices", C.c_void_p), ("num_triangles", C.c_uint32)]


class PtInstanceDesc(C.Structure):
    _fields_ = [("myMeshIndex", C.c_uint32), ("myMaterialIndex", C.c_uint32), ("myTransform", C.c_float * 16)]


class PtMaterial(C.Structure):
    _fields_ = [("myEmission", C.c_float * 3), ("myColor", C.c_uint32)]


PT_MAT_GGX = 1


class PtMaterialExt(C.Structure):   # extension for BASELINE C4/C5, not in the reference (include/pt_api.h)
    _fields_ = [("roughness", C.c_float), ("metallic", C.c_float), ("albedo_pattern", C.c_uint32),
                ("pattern_scale", C.c_float), ("flags", C.c_uint32), ("reserved", C.c_uint32 * 3)]


class PtStats(C.Structure):   # versioned by size: struct_size is set by the caller (context.Context.stats does), members are append-only
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32), ("samples", C.c_uint64), ("rays", C.c_uint64), ("shaded_hits", C.c_uint64), ("sky_misses", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("extend_launches", C.c_uint64),
                ("extend_ms", C.c_double), ("render_ms", C.c_double),
                ("rays_bounce", C.c_uint64 * 16), ("extend_ms_bounce", C.c_double * 16),
                ("node_wave_steps", C.c_uint64), ("tri_wave_steps", C.c_uint64),
                ("extend_launches_bounce", C.c_uint64 * 16), ("packet_launches", C.c_uint64), ("packet_cal_ms", C.c_float * 2),
                ("shared_launches", C.c_uint64), ("shared_cal_ms", C.c_float * 2), ("shared_choice", C.c_int32), ("reserved1", C.c_uint32)]


class PtSceneInfo(C.Structure):
    _fields_ = [("num_triangles", C.c_uint32), ("num_vertices", C.c_uint32), ("num_instances", C.c_uint32),
                ("num_materials", C.c_uint32), ("num_bvh_nodes", C.c_uint32), ("bvh_max_depth", C.c_uint32),
                ("bvh_bytes", C.c_uint64), ("bounds_min", C.c_float * 3), ("bounds_max", C.c_float * 3),
                ("build_ms", C.c_double)]


# numpy views of the plain-data records
VERTEX_DTYPE = np.dtype([("pos", np.float32, 3), ("normal", np.float32, 3), ("uv", np.float32, 2)])
HIT_DTYPE = np.dtype([("t", np.float32), ("u", np.float32), ("v", np.float32), ("prim", np.uint32)])
RAY_DTYPE = np.dtype([("origin", np.float32, 3), ("tmin", np.float32), ("dir", np.float32, 3), ("tmax", np.float32)])
assert VERTEX_DTYPE.itemsize == 32 and HIT_DTYPE.itemsize == 16 and RAY_DTYPE.itemsize == 32

# every symbol include/pt_api.h and include/pt_api_tuning.h declare: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "pt_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "pt_destroy": (C.c_int, [_P]),
    "pt_last_error": (C.c_char_p, [_P]),
    "pt_set_stream": (C.c_int, [_P, _P]),
    "pt_synchronize": (C.c_int, [_P]),
    "pt_set_option": (C.c_int, [_P, C.c_int, C.c_int64]),
    "pt_debug_set_tuning": (C.c_int, [_P, C.c_int, C.c_int64]),   # include/pt_api_tuning.h
    "pt_scene_create": (C.c_int, [_P, C.POINTER(PtMeshDesc), C.c_uint32, C.POINTER(PtInstanceDesc), C.c_uint32,
                                  C.POINTER(PtMaterial), C.c_uint32, C.c_uint32]),
    "pt_scene_set_material_ext": (C.c_int, [_P, C.POINTER(PtMaterialExt), C.c_uint32]),
    "pt_scene_set_albedo_texture": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32]),
    "pt_scene_info": (C.c_int, [_P, C.POINTER(PtSceneInfo)]),
    "pt_sky_set": (C.c_int, [_P, C.POINTER(PtAtmosphere)]),
    "pt_sky_lut_read": (C.c_int, [_P, _P]),
    "pt_accum_reset": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "pt_accum_bind": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32]),
    "pt_accum_device_ptr": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "pt_accum_set_count": (C.c_int, [_P, C.c_uint64]),
    "pt_accum_read": (C.c_int, [_P, _P]),
    "pt_accum_read_sum": (C.c_int, [_P, _P, C.POINTER(C.c_uint64)]),
    "pt_render": (C.c_int, [_P, C.POINTER(PtRtConsts), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "pt_render_ao": (C.c_int, [_P, C.POINTER(PtRtConsts), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "pt_reduce": (C.c_int, [C.POINTER(_P), C.c_uint32]),
    "pt_primary_hits": (C.c_int, [_P, C.POINTER(PtRtConsts), C.c_uint32, C.c_uint32, C.c_uint32, _P]),
    "pt_trace_rays": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "pt_trace_rays_device": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "pt_trace_rays_device_noalloc": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "pt_occluded": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "pt_debug_eval": (C.c_int, [_P, C.c_int, C.POINTER(PtRtConsts), _P, C.c_uint64, _P]),
    "pt_debug_read_queue": (C.c_int, [_P, C.c_uint32, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "pt_debug_read_bvh": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_float),
                                   C.POINTER(C.c_uint32)]),
    "pt_stats_get": (C.c_int, [_P, C.POINTER(PtStats)]),
    "pt_stats_reset": (C.c_int, [_P]),
}

_lib = None


def load_library():
    """Load libpathtracer_amd.so and type its entry points. Raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C pathtracer_amd/csrc` (or __graft_entry__.build()). "
            "There is no CPU fallback for the path-tracing hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        if os.environ.get("PT_AMD_LIB") and not hasattr(lib, name):
            continue             # an older experiment build (tools/experiments/ab.sh) may lack newer entry points
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class PtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pt error {code}: {msg}")
        self.code = code


def ptr(a):
    """Pointer to a numpy array's storage (must be C-contiguous)."""
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def fill3(dst, src):
    for i in range(3):
        dst[i] = float(src[i])
