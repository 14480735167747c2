"""Builder experiments WITHOUT a GPU: build the product's BVH8 on the host (csrc/bvh_builder.cpp through tests/walker) and
walk it with the canonical schedule on camera rays and on the incoherent diffuse bounces that follow them.

    python tools/bvh_cpu_quality.py <atrium|street|street_hard|courtyard> [pair_mode] [w h] [bounces]

Prints nodes / triangles per ray (SURVEY 8(d)'s N_node, N_tri) and B_ray = 80 N_node + 48 N_tri + 48 per bounce. The
bounce rays are cosine-distributed about the geometric normal of the hit (seeded numpy RNG): not the renderer's exact
rays, but the same incoherent distribution, and identical for every builder variant compared."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import api, scenes   # noqa: E402
from tests.common import lookat_consts   # noqa: E402

WORKLOADS = {"atrium": (lambda: scenes.atrium(1), scenes.ATRIUM_CAM), "street": (lambda: scenes.street(2), scenes.STREET_CAM),
             "street_hard": (lambda: scenes.street_hard(4), scenes.STREET_HARD_CAM),
             "courtyard": (lambda: scenes.courtyard(3), scenes.COURTYARD_CAM)}


def flatten(scene):
    """World-space triangle soup (T,9) float32 in the library's flatten order (instance-major), as scene_flatten.cpp does."""
    out = []
    for m, _, t in scene.instances:
        v, idx = scene.meshes[m]
        p = v["pos"][idx.astype(np.int64)]
        if t is not None:
            M = np.asarray(t, dtype=np.float32).reshape(4, 4).T   # column-major
            p = (p @ M[:3, :3].T + M[:3, 3]).astype(np.float32)
        out.append(p.reshape(-1, 9))
    return np.ascontiguousarray(np.concatenate(out), dtype=np.float32)


def primary_rays(c, w, h, seed=0):
    rng = np.random.default_rng(seed)
    px, py = np.meshgrid(np.arange(w), np.arange(h), indexing="xy")
    u = (px + rng.uniform(0, 1, px.shape)) / w
    v = 1.0 - (py + rng.uniform(0, 1, py.shape)) / h
    corner, xa, ya, cp = (np.array(list(a), dtype=np.float64) for a in (c.myNearPlaneCorner, c.myXAxis, c.myYAxis, c.myCameraPos))
    o = corner + xa * u[..., None] + ya * v[..., None]
    d = o - cp
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    rays = np.zeros(w * h, dtype=api.RAY_DTYPE)
    rays["origin"] = o.reshape(-1, 3).astype(np.float32)
    rays["dir"] = d.reshape(-1, 3).astype(np.float32)
    rays["tmin"], rays["tmax"] = 0.0, 10000.0
    return rays


def bounce(rays, hits, tris, rng):
    ok = hits["prim"] != api.PT_MISS
    r, hrec = rays[ok], hits[ok]
    T = tris[hrec["prim"].astype(np.int64)].reshape(-1, 3, 3).astype(np.float64)
    n = np.cross(T[:, 1] - T[:, 0], T[:, 2] - T[:, 0])
    n /= np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-30)
    d = r["dir"].astype(np.float64)
    n = np.where((np.sum(n * d, axis=-1, keepdims=True) > 0), -n, n)
    p = r["origin"].astype(np.float64) + d * hrec["t"].astype(np.float64)[:, None]
    s = rng.normal(size=n.shape)
    s /= np.linalg.norm(s, axis=-1, keepdims=True)
    nd = n + s
    nd /= np.maximum(np.linalg.norm(nd, axis=-1, keepdims=True), 1e-30)
    out = np.zeros(len(r), dtype=api.RAY_DTYPE)
    out["origin"], out["dir"] = p.astype(np.float32), nd.astype(np.float32)
    out["tmin"], out["tmax"] = 0.001, 10000.0
    return out


def load_walker():
    so = os.path.join(ROOT, "tests", "walker", "libbvh8_walker.so")
    wl = C.CDLL(so)
    P = C.c_void_p
    wl.bvh8_build.restype = P
    wl.bvh8_build.argtypes = [P, C.c_uint32, C.c_int, C.c_int, P]
    wl.bvh8_free.argtypes = [P]
    wl.bvh8_walk_built.restype = C.c_int
    wl.bvh8_walk_built.argtypes = [P, P, C.c_uint64, P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]
    return wl


def measure(name, pair_mode=-1, w=320, h=180, bounces=3, threads=8, quiet=False):
    mk, cam = WORKLOADS[name]
    tris = flatten(mk())
    wl = load_walker()
    info = (C.c_uint64 * 8)()
    t0 = time.time()
    hnd = wl.bvh8_build(api.ptr(tris), len(tris), pair_mode, threads, info)
    build_s = time.time() - t0
    nn, nr, npairs, rb, depth = (int(info[k]) for k in range(5))
    sah = float(np.array([info[6]], dtype=np.uint64).view(np.float64)[0])
    res = {"workload": name, "triangles": len(tris), "nodes": nn, "records": nr, "pairs": npairs, "record_bytes": rb, "depth": depth,
           "bvh_mb": (nn * 80 + nr * rb) / 1e6, "build_s": build_s, "sah_cost": sah, "bounce": []}
    rng = np.random.default_rng(7)
    rays = primary_rays(lookat_consts(w, h, cam), w, h)
    for b in range(bounces + 1):
        hits = np.empty(len(rays), dtype=api.HIT_DTYPE)
        cn, ct = C.c_uint64(), C.c_uint64()
        rc = wl.bvh8_walk_built(hnd, api.ptr(rays), len(rays), api.ptr(hits), C.byref(cn), C.byref(ct), threads)
        assert rc == 0, rc
        n = max(1, len(rays))
        res["bounce"].append({"rays": len(rays), "nodes_per_ray": cn.value / n, "tris_per_ray": ct.value / n,
                              "bytes_per_ray": 80 * cn.value / n + 48 * ct.value / n + 48, "hit_rate": float((hits["prim"] != api.PT_MISS).mean()),
                              "hits_crc": int(np.bitwise_xor.reduce(hits["prim"].astype(np.uint64) * np.arange(1, len(hits) + 1, dtype=np.uint64)))})
        rays = bounce(rays, hits, tris, rng)
        if len(rays) == 0:
            break
    wl.bvh8_free(hnd)
    sec = res["bounce"][1:]
    tot = sum(b["rays"] for b in sec) or 1
    res["secondary"] = {k: sum(b[k] * b["rays"] for b in sec) / tot for k in ("nodes_per_ray", "tris_per_ray", "bytes_per_ray")}
    if not quiet:
        print(f"{name}: {len(tris)} tris, {nn} nodes, {nr} records x {rb} B ({npairs} pairs), {res['bvh_mb']:.1f} MB, depth {depth}, build {build_s:.1f} s, SAH cost of the 8-wide tree {sah:.6g}")
        for b, r in enumerate(res["bounce"]):
            print(f"  bounce {b}: {r['rays']:7d} rays  {r['nodes_per_ray']:6.2f} nodes + {r['tris_per_ray']:6.2f} tris per ray = {r['bytes_per_ray']:7.1f} B/ray  hit {r['hit_rate']:.3f} crc {r['hits_crc']:x}")
        s = res["secondary"]
        print(f"  secondary: {s['nodes_per_ray']:.2f} nodes + {s['tris_per_ray']:.2f} tris = {s['bytes_per_ray']:.1f} B/ray")
    return res


if __name__ == "__main__":
    a = sys.argv[1:]
    measure(a[0] if a else "atrium", int(a[1]) if len(a) > 1 else -1, int(a[2]) if len(a) > 3 else 320, int(a[3]) if len(a) > 3 else 180,
            int(a[4]) if len(a) > 4 else 3)
