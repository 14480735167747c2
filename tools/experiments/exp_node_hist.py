"""Which BVH8 nodes do the rays of a render visit? Reads the device BVH back, replays the bounce queues of one small
render through the host walker (tests/walker) and prints the share of node visits that fall on the first K nodes of
the (breadth-first) node array - the case for keeping the top of the tree somewhere closer than L1."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pathtracer_amd import PathTracer, api, scenes
from tests.common import lookat_consts
wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
scene, cam, depth = {"atrium": (scenes.atrium, scenes.ATRIUM_CAM, 4), "street": (scenes.street, scenes.STREET_CAM, 8),
                     "courtyard": (scenes.courtyard, scenes.COURTYARD_CAM, 16)}[wl]
wlk = C.CDLL(os.path.join(ROOT, "tests", "walker", "libbvh8_walker.so"))
P = C.c_void_p
wlk.bvh8_walk.restype = C.c_int
wlk.bvh8_walk.argtypes = [P, C.c_uint64, P, C.c_uint64, C.c_uint32, C.c_float, P, C.c_uint64, P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
wlk.bvh8_set_visit_hist.argtypes = [P]
from pathtracer_amd import Context
ctx = Context(0)
ctx.scene_create(scene())
from tests import oracle_lib
ctx.sky_set(oracle_lib.load().earth_atmosphere())
nodes, tris, mag = ctx.debug_read_bvh()
info = ctx.scene_info()
print(wl, "nodes", len(nodes), "tris", len(tris), "depth", info.bvh_max_depth)
w, h = 480, 270
c = lookat_consts(w, h, cam, depth=depth)
ctx.accum_reset(w, h)
ctx.render(c, w, h, 0, 1)
ctx.synchronize()
hist = np.zeros(len(nodes), dtype=np.uint32)
wlk.bvh8_set_visit_hist(api.ptr(hist))
tot_rays = 0
for b in range(min(depth + 1, 2)):          # queues of bounce 0 and 1 survive (b is overwritten by b+2)
    pass
for b in (depth - 1, depth):                 # the last two bounce queues are intact after the render
    rays = ctx.debug_read_queue(b)
    if len(rays) == 0:
        continue
    hits = np.empty(len(rays), dtype=api.HIT_DTYPE)
    cn, ct = C.c_uint64(), C.c_uint64()
    assert wlk.bvh8_walk(api.ptr(nodes), len(nodes), api.ptr(tris), len(tris), tris.shape[1], mag, api.ptr(rays), len(rays), api.ptr(hits), C.byref(cn), C.byref(ct)) == 0
    tot_rays += len(rays)
    print(f"bounce {b}: {len(rays)} rays, {cn.value / len(rays):.2f} nodes/ray, {ct.value / len(rays):.2f} tris/ray")
tot = hist.sum()
cum = np.cumsum(hist)
for k in (1, 9, 73, 200, 400, 585, 800, 1200, 2000, 4681, 10000, 20000):
    if k <= len(nodes):
        print(f"first {k:6d} nodes ({k * 80 / 1024:8.1f} KB): {cum[k - 1] / tot:.3f} of node visits")
order = np.sort(hist)[::-1]
cs = np.cumsum(order)
for k in (73, 400, 800, 1200, 2000):
    print(f"hottest {k:6d} nodes: {cs[k - 1] / tot:.3f}")
