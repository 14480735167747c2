"""Thin object wrapper over the C-ABI (include/pt_api.h): one Context == one pt_ctx == one GPU.

Error behaviour mirrors the library: every call returns a status; non-zero raises PtError carrying pt_last_error.
"""
import ctypes as C

import numpy as np

from . import api


class SceneData:
    """What FANCY's MeshImporter hands to PathTracer::InitRtScene (PathTracer/PathTracer.cpp:115-132):
    meshes = list of (vertices[VERTEX_DTYPE], indices[uint32, 3*T]); instances = list of
    (mesh_index, material_index, 4x4 column-major transform or None); materials = list of
    (emission rgb, colour rgba in [0,1])."""

    def __init__(self, meshes, instances, materials, name=""):
        self.meshes = [(np.ascontiguousarray(v, dtype=api.VERTEX_DTYPE), np.ascontiguousarray(i, dtype=np.uint32).reshape(-1))
                       for v, i in meshes]
        self.instances = [(int(m), int(mat), None if t is None else np.asarray(t, dtype=np.float32).reshape(16))
                          for m, mat, t in instances]
        self.materials = [(tuple(float(x) for x in e), tuple(float(x) for x in c)) for e, c in materials]
        self.name = name
        # optional material extension (not in the reference): one (roughness, metallic, albedo_pattern, pattern_scale,
        # flags) tuple per material, see include/pt_api.h PtMaterialExt
        self.materials_ext = None
        # optional albedo image of that extension: (h, w, 4) uint8, sampled by materials with albedo_pattern 3
        self.albedo_texture = None

    @property
    def num_triangles(self):
        return sum(len(self.meshes[m][1]) // 3 for m, _, _ in self.instances)


def encode_unorm_rgba(c):
    """MathUtil::Encode_Unorm_RGBA (FANCY, absent): round(clamp(c,0,1)*255), R in the low byte."""
    out = 0
    for k in range(4):
        v = min(max(float(c[k]) if k < len(c) else 1.0, 0.0), 1.0)
        out |= int(np.floor(np.float32(v) * np.float32(255.0) + np.float32(0.5))) << (8 * k)
    return out


def pack_material_ext(scene):
    """SceneData.materials_ext -> ctypes array of PtMaterialExt, or None."""
    if not getattr(scene, "materials_ext", None):
        return None
    assert len(scene.materials_ext) == len(scene.materials)
    arr = (api.PtMaterialExt * len(scene.materials_ext))()
    for i, (rough, metal, pattern, scale, flags) in enumerate(scene.materials_ext):
        arr[i].roughness, arr[i].metallic = float(rough), float(metal)
        arr[i].albedo_pattern, arr[i].pattern_scale, arr[i].flags = int(pattern), float(scale), int(flags)
    return arr


def pack_scene(scene):
    """SceneData -> (ctypes arrays, keepalive list) in the layout pt_scene_create / orc_scene_create take."""
    keep = []
    meshes = (api.PtMeshDesc * max(1, len(scene.meshes)))()
    for i, (v, idx) in enumerate(scene.meshes):
        keep += [v, idx]
        meshes[i].vertices = v.ctypes.data
        meshes[i].num_vertices = len(v)
        meshes[i].indices = idx.ctypes.data
        meshes[i].num_triangles = len(idx) // 3
    inst = (api.PtInstanceDesc * max(1, len(scene.instances)))()
    ident = np.eye(4, dtype=np.float32).reshape(16)
    for i, (m, mat, t) in enumerate(scene.instances):
        inst[i].myMeshIndex = m
        inst[i].myMaterialIndex = mat
        tt = ident if t is None else t
        for k in range(16):
            inst[i].myTransform[k] = float(tt[k])
    mats = (api.PtMaterial * max(1, len(scene.materials)))()
    for i, (e, c) in enumerate(scene.materials):
        api.fill3(mats[i].myEmission, e)
        mats[i].myColor = encode_unorm_rgba(c)
    return meshes, len(scene.meshes), inst, len(scene.instances), mats, len(scene.materials), keep


class Context:
    def __init__(self, device=0):
        self.lib = api.load_library()
        h = C.c_void_p()
        rc = self.lib.pt_create(int(device), C.byref(h))
        if rc != api.PT_OK:
            raise api.PtError(rc, self.lib.pt_last_error(None).decode())
        self.h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "h", None):
            self.lib.pt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != api.PT_OK:
            raise api.PtError(rc, self.lib.pt_last_error(self.h).decode())

    # ---- scene / sky ----
    def scene_create(self, scene, flags=api.PT_SCENE_DEFAULT):
        m, nm, i, ni, mt, nmt, keep = pack_scene(scene)
        self._ck(self.lib.pt_scene_create(self.h, m, nm, i, ni, mt, nmt, flags))
        del keep
        ext = pack_material_ext(scene)
        if ext is not None:
            self._ck(self.lib.pt_scene_set_material_ext(self.h, ext, len(ext)))
        if getattr(scene, "albedo_texture", None) is not None:
            self.scene_set_albedo_texture(scene.albedo_texture)

    def scene_set_albedo_texture(self, rgba8):
        """pt_scene_set_albedo_texture: (h, w, 4) uint8 image for albedo pattern 3, or None to remove it."""
        if rgba8 is None:
            self._ck(self.lib.pt_scene_set_albedo_texture(self.h, None, 0, 0))
            return
        img = np.ascontiguousarray(rgba8, dtype=np.uint8)
        assert img.ndim == 3 and img.shape[2] == 4
        self._ck(self.lib.pt_scene_set_albedo_texture(self.h, api.ptr(img), img.shape[1], img.shape[0]))

    def scene_set_material_ext(self, scene_or_none):
        ext = pack_material_ext(scene_or_none) if scene_or_none is not None else None
        self._ck(self.lib.pt_scene_set_material_ext(self.h, ext, len(ext) if ext is not None else 0))

    def scene_info(self):
        info = api.PtSceneInfo()
        self._ck(self.lib.pt_scene_info(self.h, C.byref(info)))
        return info

    def sky_set(self, atmosphere):
        self._ck(self.lib.pt_sky_set(self.h, C.byref(atmosphere)))

    def sky_lut_read(self):
        out = np.empty((api.TLUT_H, api.TLUT_W, 4), dtype=np.uint16)
        self._ck(self.lib.pt_sky_lut_read(self.h, api.ptr(out)))
        return out

    # ---- accumulation ----
    def accum_reset(self, w, h):
        self._ck(self.lib.pt_accum_reset(self.h, w, h))
        self.w, self.hgt = w, h

    def accum_bind(self, device_ptr, w, h):
        self._ck(self.lib.pt_accum_bind(self.h, C.c_void_p(device_ptr), w, h))
        self.w, self.hgt = w, h

    def accum_device_ptr(self):
        p = C.c_void_p()
        n = C.c_uint64()
        self._ck(self.lib.pt_accum_device_ptr(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def accum_set_count(self, n):
        self._ck(self.lib.pt_accum_set_count(self.h, int(n)))

    def accum_read(self):
        out = np.empty((self.hgt, self.w, 4), dtype=np.float32)
        self._ck(self.lib.pt_accum_read(self.h, api.ptr(out)))
        return out

    def accum_read_sum(self):
        out = np.empty((self.hgt, self.w, 4), dtype=np.float32)
        n = C.c_uint64()
        self._ck(self.lib.pt_accum_read_sum(self.h, api.ptr(out), C.byref(n)))
        return out, n.value

    # ---- hot path ----
    def render(self, consts, w, h, first_sample, num_samples):
        self._ck(self.lib.pt_render(self.h, C.byref(consts), w, h, first_sample, num_samples))

    def render_ao(self, consts, w, h, first_sample, num_samples):
        self._ck(self.lib.pt_render_ao(self.h, C.byref(consts), w, h, first_sample, num_samples))

    def primary_hits(self, consts, w, h, sample=0):
        out = np.empty((h, w), dtype=api.HIT_DTYPE)
        self._ck(self.lib.pt_primary_hits(self.h, C.byref(consts), w, h, sample, api.ptr(out)))
        return out

    def trace_rays(self, rays):
        rays = np.ascontiguousarray(rays, dtype=api.RAY_DTYPE)
        out = np.empty(len(rays), dtype=api.HIT_DTYPE)
        self._ck(self.lib.pt_trace_rays(self.h, api.ptr(rays), len(rays), api.ptr(out)))
        return out

    def occluded(self, rays):
        rays = np.ascontiguousarray(rays, dtype=api.RAY_DTYPE)
        out = np.empty(len(rays), dtype=np.uint8)
        self._ck(self.lib.pt_occluded(self.h, api.ptr(rays), len(rays), api.ptr(out)))
        return out

    def trace_rays_device(self, rays_ptr, n, hits_ptr):
        self._ck(self.lib.pt_trace_rays_device(self.h, C.c_void_p(rays_ptr), int(n), C.c_void_p(hits_ptr)))

    def trace_rays_device_noalloc(self, rays_ptr, n, hits_ptr):
        self._ck(self.lib.pt_trace_rays_device_noalloc(self.h, C.c_void_p(rays_ptr), int(n), C.c_void_p(hits_ptr)))

    def debug_eval(self, fn, rows, consts=None):
        rows = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, api.FN_IN_WIDTH[fn])
        out = np.empty((len(rows), api.FN_OUT_WIDTH[fn]), dtype=np.float32)
        cp = C.byref(consts) if consts is not None else None
        self._ck(self.lib.pt_debug_eval(self.h, fn, cp, api.ptr(rows), len(rows), api.ptr(out)))
        return out

    def debug_read_queue(self, bounce):
        """The last pt_render batch's ray queue of `bounce` as RAY_DTYPE rows (queue order depends on block scheduling).
        The queues of even and odd bounces ping-pong between two buffers: after a render only the queue of its LAST bounce
        (and the one before it) is whole, an earlier queue's head has been overwritten by bounce + 2."""
        n = C.c_uint64()
        self._ck(self.lib.pt_debug_read_queue(self.h, bounce, None, 0, C.byref(n)))
        out = np.empty(n.value, dtype=api.RAY_DTYPE)
        self._ck(self.lib.pt_debug_read_queue(self.h, bounce, api.ptr(out), n.value, C.byref(n)))
        return out

    def debug_read_bvh(self):
        """(nodes uint8[n,80], records uint8[m,48 or 64], scene_mag): the BVH8 as it lives on the device (csrc/bvh8.h);
        48-byte records hold one triangle, 64-byte records one triangle or an edge-sharing pair."""
        nn, nt, mag, rb = C.c_uint64(), C.c_uint64(), C.c_float(), C.c_uint32()
        self._ck(self.lib.pt_debug_read_bvh(self.h, None, 0, None, 0, C.byref(nn), C.byref(nt), C.byref(mag), C.byref(rb)))
        nodes = np.empty((max(1, nn.value), 80), dtype=np.uint8)
        tris = np.empty((max(1, nt.value), rb.value), dtype=np.uint8)
        self._ck(self.lib.pt_debug_read_bvh(self.h, api.ptr(nodes), nn.value, api.ptr(tris), nt.value, None, None, None, None))
        return nodes[:nn.value], tris[:nt.value], mag.value

    # ---- misc ----
    def set_option(self, opt, value):
        self._ck(self.lib.pt_set_option(self.h, opt, int(value)))

    def set_tuning(self, knob, value):
        """include/pt_api_tuning.h: schedule knobs of the traversal kernel (api.PT_TUNE_*); hit records never depend on them."""
        self._ck(self.lib.pt_debug_set_tuning(self.h, knob, int(value)))

    def set_stream(self, stream_handle):
        self._ck(self.lib.pt_set_stream(self.h, C.c_void_p(stream_handle)))

    def synchronize(self):
        self._ck(self.lib.pt_synchronize(self.h))

    def stats(self):
        s = api.PtStats()
        s.struct_size = C.sizeof(api.PtStats)
        self._ck(self.lib.pt_stats_get(self.h, C.byref(s)))
        return s

    def stats_reset(self):
        self._ck(self.lib.pt_stats_reset(self.h))


def reduce_contexts(contexts):
    """pt_reduce: add the sum targets of contexts[1:] onto contexts[0] (single-process multi-GPU)."""
    arr = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
    lib = contexts[0].lib
    rc = lib.pt_reduce(arr, len(contexts))
    if rc != api.PT_OK:
        raise api.PtError(rc, lib.pt_last_error(contexts[0].h).decode())
