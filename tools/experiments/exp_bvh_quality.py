"""BVH quality check: nodes / triangles per ray of the library's BVH8 (canonical schedule, COUNT kernel) next to the
oracle's own binary BVH (exact fp32 boxes, distance-ordered traversal) on the same incoherent rays."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pathtracer_amd import Context, api, scenes
from tests import oracle_lib
from tests.common import lookat_consts, random_rays
wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
scene, cam, depth = {"atrium": (scenes.atrium, scenes.ATRIUM_CAM, 4), "street": (scenes.street, scenes.STREET_CAM, 8)}[wl]
sc = scene()
ctx = Context(0)
ctx.scene_create(sc)
orc = oracle_lib.load()
ctx.sky_set(orc.earth_atmosphere())
osc = orc.scene(sc)
w, h = 480, 270
c = lookat_consts(w, h, cam, depth=depth)
ctx.accum_reset(w, h)
ctx.render(c, w, h, 0, 1)
ctx.synchronize()
rays = ctx.debug_read_queue(depth)          # the last bounce's (incoherent) rays
ctx.set_tuning(api.PT_TUNE_EXTEND_DEFER, 0); ctx.set_tuning(api.PT_TUNE_EXTEND_POSTPONE_PCT, 0); ctx.set_option(api.PT_OPT_COUNT_TRAVERSAL, 1)
ctx.stats_reset()
got = ctx.trace_rays(rays)
st = ctx.stats()
ref, cnt = osc.trace_rays(rays, counters=True)
assert np.array_equal(got["prim"], ref["prim"])
n = len(rays)
print(f"{wl}: {n} bounce-{depth} rays; BVH8 (quantised, octant order): {st.nodes_visited / n:.2f} nodes + {st.tris_tested / n:.2f} tris per ray;"
      f" oracle BVH2 (fp32 boxes, distance order): {cnt[0] / n:.2f} nodes + {cnt[1] / n:.2f} tris per ray; hit rate {(ref['prim'] != api.PT_MISS).mean():.3f}")
