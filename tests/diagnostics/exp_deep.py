import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pathtracer_amd import api
from pathtracer_amd.context import Context, SceneData
from tests import oracle_lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
ratio = float(sys.argv[2]) if len(sys.argv) > 2 else 1.028
k = np.arange(n, dtype=np.float64); s = ratio ** k
v = np.zeros(3 * n, dtype=api.VERTEX_DTYPE)
pos = np.zeros((n, 3, 3))
pos[:, 0] = np.stack([s, 0 * s, 0.001 * k], axis=1); pos[:, 1] = np.stack([-s, s, 0.001 * k], axis=1); pos[:, 2] = np.stack([-s, -s, 0.001 * k], axis=1)
v["pos"] = pos.reshape(-1, 3).astype(np.float32); v["normal"] = (0, 0, 1)
sc = SceneData([(v, np.arange(3 * n, dtype=np.uint32))], [(0, 0, None)], [((0, 0, 0), (0.5, 0.5, 0.5, 1))])
ctx = Context(0); ctx.scene_create(sc)
osc = oracle_lib.load().scene(sc)
print("depth", ctx.scene_info().bvh_max_depth, "nodes", ctx.scene_info().num_bvh_nodes)
rng = np.random.default_rng(2)
rays = np.zeros(6000, dtype=api.RAY_DTYPE)
r = ratio ** rng.uniform(0, n, 6000); ang = rng.uniform(0, 2 * np.pi, 6000)
rays["origin"] = np.stack([0.3 * r * np.cos(ang), 0.3 * r * np.sin(ang), np.full(6000, -5.0)], axis=1).astype(np.float32)
d = np.stack([rng.normal(0, 0.2, 6000), rng.normal(0, 0.2, 6000), np.ones(6000)], axis=1)
rays["dir"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
rays["tmin"], rays["tmax"] = 0.0, 1e30
got = ctx.trace_rays(rays); brute = osc.trace_rays(rays, brute=True); obvh = osc.trace_rays(rays)
bad = np.nonzero(got["prim"] != brute["prim"])[0]
print("gpu != brute:", len(bad), " oracle bvh != brute:", int((obvh["prim"] != brute["prim"]).sum()))
for i in bad[:12]:
    print(i, "origin", rays["origin"][i], "dir", rays["dir"][i], "| gpu", got["prim"][i], got["t"][i], "| brute", brute["prim"][i], brute["t"][i], "| obvh", obvh["prim"][i], obvh["t"][i])
