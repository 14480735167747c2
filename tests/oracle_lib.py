"""ctypes binding of the CPU oracle (oracle/libpt_oracle.so). TEST INFRASTRUCTURE: imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the pathtracer_amd package."""
import ctypes as C
import os
import subprocess

import numpy as np

from pathtracer_amd import api
from pathtracer_amd.context import pack_material_ext, pack_scene

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_DIR, "libpt_oracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(_DIR, f) for f in ("pt_oracle.cpp", "geometry.h", "shading.h", "sky.h", "hlsl_math.h")]
    srcs.append(os.path.join(_ROOT, "include", "pt_api.h"))
    stale = force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs if os.path.exists(s))
    if stale:
        subprocess.check_call(["make", "-C", _DIR, "-B", "libpt_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def load():
    global _lib
    if _lib is not None:
        return Oracle(_lib)
    try:
        build()
    except Exception:
        if not os.path.exists(_SO):
            raise
    lib = C.CDLL(_SO)
    P = C.c_void_p
    lib.orc_scene_create.restype = P
    lib.orc_scene_create.argtypes = [C.POINTER(api.PtMeshDesc), C.c_uint32, C.POINTER(api.PtInstanceDesc), C.c_uint32,
                                     C.POINTER(api.PtMaterial), C.c_uint32, C.c_uint32]
    lib.orc_scene_destroy.argtypes = [P]
    lib.orc_scene_set_material_ext.restype = C.c_int
    lib.orc_scene_set_material_ext.argtypes = [P, C.POINTER(api.PtMaterialExt), C.c_uint32]
    lib.orc_scene_set_albedo_texture.restype = C.c_int
    lib.orc_scene_set_albedo_texture.argtypes = [P, P, C.c_uint32, C.c_uint32]
    lib.orc_scene_num_triangles.restype = C.c_uint32
    lib.orc_scene_num_triangles.argtypes = [P]
    lib.orc_scene_triangle.argtypes = [P, C.c_uint32, P, P]
    lib.orc_trace_rays.argtypes = [P, P, C.c_uint64, P, C.c_int, P]
    lib.orc_primary_hits.argtypes = [P, C.POINTER(api.PtRtConsts), C.c_uint32, C.c_uint32, C.c_uint32, P, C.c_int, P]
    lib.orc_bounce_rays.argtypes = [P, C.POINTER(api.PtRtConsts), P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, P, P, P]
    lib.orc_render.argtypes = [P, C.POINTER(api.PtRtConsts), P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, P,
                               C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, P, P]
    lib.orc_render_ao.argtypes = [P, C.POINTER(api.PtRtConsts), P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, P, C.c_int, P]
    lib.orc_eval.restype = C.c_int
    lib.orc_eval.argtypes = [C.c_int, C.POINTER(api.PtRtConsts), P, P, C.c_uint64, P]
    lib.orc_transmittance_lut.argtypes = [C.POINTER(api.PtAtmosphere), P]
    lib.orc_earth_atmosphere.argtypes = [C.POINTER(api.PtAtmosphere)]
    lib.orc_pcg4d.argtypes = [P, P]
    lib.orc_rand01.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, P]
    lib.orc_encode_unorm_rgba.restype = C.c_uint32
    lib.orc_encode_unorm_rgba.argtypes = [P]
    lib.orc_float_to_half.restype = C.c_uint16
    lib.orc_float_to_half.argtypes = [C.c_float]
    lib.orc_half_to_float.restype = C.c_float
    lib.orc_half_to_float.argtypes = [C.c_uint16]
    lib.orc_set_russian_roulette.argtypes = [C.c_int]
    lib.orc_num_threads.restype = C.c_int
    lib.orc_set_num_threads.argtypes = [C.c_int]
    _lib = lib
    return Oracle(lib)


class OracleScene:
    def __init__(self, lib, scene, flags=0):
        m, nm, i, ni, mt, nmt, keep = pack_scene(scene)
        self.lib = lib
        self.h = lib.orc_scene_create(m, nm, i, ni, mt, nmt, flags)
        if not self.h:
            raise ValueError("orc_scene_create rejected the scene")
        self.num_triangles = lib.orc_scene_num_triangles(self.h)
        ext = pack_material_ext(scene)
        if ext is not None:
            assert lib.orc_scene_set_material_ext(self.h, ext, len(ext)) == 0
        if getattr(scene, "albedo_texture", None) is not None:
            img = np.ascontiguousarray(scene.albedo_texture, dtype=np.uint8)
            assert lib.orc_scene_set_albedo_texture(self.h, api.ptr(img), img.shape[1], img.shape[0]) == 0

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_scene_destroy(self.h)
            self.h = None

    def triangle(self, i):
        v = np.empty(9, dtype=np.float32)
        inst = C.c_uint32()
        self.lib.orc_scene_triangle(self.h, i, api.ptr(v), C.byref(inst))
        return v.reshape(3, 3), inst.value

    def trace_rays(self, rays, brute=False, counters=False):
        rays = np.ascontiguousarray(rays, dtype=api.RAY_DTYPE)
        out = np.empty(len(rays), dtype=api.HIT_DTYPE)
        cnt = np.zeros(2, dtype=np.uint64)
        self.lib.orc_trace_rays(self.h, api.ptr(rays), len(rays), api.ptr(out), int(brute), api.ptr(cnt) if counters else None)
        return (out, cnt) if counters else out

    def primary_hits(self, consts, w, h, sample=0, brute=False, want_rays=False):
        out = np.empty((h, w), dtype=api.HIT_DTYPE)
        rays = np.empty((h, w), dtype=api.RAY_DTYPE) if want_rays else None
        self.lib.orc_primary_hits(self.h, C.byref(consts), w, h, sample, api.ptr(out), int(brute),
                                  api.ptr(rays) if want_rays else None)
        return (out, rays) if want_rays else out

    def bounce_rays(self, consts, w, h, sample, bounce, lut=None, brute=False):
        """The ray every path of one sample index carries into RayGen-loop iteration `bounce` (= the product's bounce-b
        queue): (rays[h,w] RAY_DTYPE, alive[h,w] bool, state[h,w,8] = transmission rgb, 0, luminance rgb, 0)."""
        rays = np.zeros((h, w), dtype=api.RAY_DTYPE)
        alive = np.zeros((h, w), dtype=np.uint8)
        state = np.zeros((h, w, 8), dtype=np.float32)
        lut_p = api.ptr(np.ascontiguousarray(lut)) if lut is not None else None
        self.lib.orc_bounce_rays(self.h, C.byref(consts), lut_p, w, h, sample, bounce, int(brute), api.ptr(rays),
                                 api.ptr(alive), api.ptr(state))
        return rays, alive.astype(bool), state

    def render(self, consts, w, h, first_sample, num_samples, lut=None, accum=None, n_prev=0, accum_mode=0,
               jitter_mode=0, brute=False, count_traversal=False, per_sample=False):
        """Returns (accum[h,w,4], stats dict, per_sample or None). accum_mode 0 = sum, 1 = reference running mean."""
        if accum is None:
            accum = np.zeros((h, w, 4), dtype=np.float32)
        stats = np.zeros(8, dtype=np.uint64)
        ps = np.zeros((num_samples, h, w, 3), dtype=np.float32) if per_sample else None
        lut_p = api.ptr(np.ascontiguousarray(lut)) if lut is not None else None
        self.lib.orc_render(self.h, C.byref(consts), lut_p, w, h, first_sample, num_samples, api.ptr(accum), n_prev,
                            accum_mode, jitter_mode, int(brute), int(count_traversal), api.ptr(stats),
                            api.ptr(ps) if per_sample else None)
        keys = ("rays", "shaded_hits", "sky_misses", "samples", "nodes", "tris", "ns", "threads")
        return accum, dict(zip(keys, (int(x) for x in stats))), ps

    def render_ao(self, consts, w, h, first_sample, num_samples, lut=None, accum=None, brute=False):
        """The Ao.hlsl pipeline: returns (accum[h,w,4] sum of samples, stats dict)."""
        if accum is None:
            accum = np.zeros((h, w, 4), dtype=np.float32)
        stats = np.zeros(8, dtype=np.uint64)
        lut_p = api.ptr(np.ascontiguousarray(lut)) if lut is not None else None
        self.lib.orc_render_ao(self.h, C.byref(consts), lut_p, w, h, first_sample, num_samples, api.ptr(accum), int(brute),
                               api.ptr(stats))
        keys = ("rays", "shaded_hits", "sky_misses", "samples", "nodes", "tris", "ns", "threads")
        return accum, dict(zip(keys, (int(x) for x in stats)))


class Oracle:
    def __init__(self, lib):
        self.lib = lib

    def scene(self, scene, flags=0):
        return OracleScene(self.lib, scene, flags)

    def pcg4d(self, v):
        i = np.array(v, dtype=np.uint32)
        o = np.empty(4, dtype=np.uint32)
        self.lib.orc_pcg4d(api.ptr(i), api.ptr(o))
        return o

    def rand01(self, px, py, seed, n):
        o = np.empty(n, dtype=np.float32)
        self.lib.orc_rand01(px, py, seed, n, api.ptr(o))
        return o

    def eval(self, fn, rows, consts=None, lut=None):
        rows = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, api.FN_IN_WIDTH[fn])
        out = np.empty((len(rows), api.FN_OUT_WIDTH[fn]), dtype=np.float32)
        c = consts if consts is not None else api.PtRtConsts()
        lut_p = api.ptr(np.ascontiguousarray(lut)) if lut is not None else None
        rc = self.lib.orc_eval(fn, C.byref(c), lut_p, api.ptr(rows), len(rows), api.ptr(out))
        assert rc == 0
        return out

    def transmittance_lut(self, atm):
        out = np.empty((api.TLUT_H, api.TLUT_W, 4), dtype=np.uint16)
        self.lib.orc_transmittance_lut(C.byref(atm), api.ptr(out))
        return out

    def earth_atmosphere(self):
        a = api.PtAtmosphere()
        self.lib.orc_earth_atmosphere(C.byref(a))
        return a

    def encode_unorm_rgba(self, c):
        a = np.array(c, dtype=np.float32)
        return self.lib.orc_encode_unorm_rgba(api.ptr(a))

    def set_russian_roulette(self, on):
        self.lib.orc_set_russian_roulette(int(bool(on)))

    def num_threads(self):
        return self.lib.orc_num_threads()
