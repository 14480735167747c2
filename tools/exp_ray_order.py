"""Experiment: how much does ray ordering (coherence) matter for k_extend? Takes the real bounce-1 ray queue of a
workload, reorders it on the host in several ways and times pt_trace_rays_device on each order."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import PathTracer, api, scenes

wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
W, H = 1920, 1080
scene, cam, depth = {"atrium": (scenes.atrium, scenes.ATRIUM_CAM, 4), "street": (scenes.street, scenes.STREET_CAM, 8),
                     "courtyard": (scenes.courtyard, scenes.COURTYARD_CAM, 16)}[wl]
pt = PathTracer(0, W, H); pt.myHalfResRender = False; pt.UpdateOutputTexture()
pt.LoadScene(scene(), cam["eye"]); pt.myCamera.LookAt(cam["eye"], cam["target"])
pt.mySampleSky = True; pt.myLightInstanceIdx = -1; pt.myMaxRecursionDepth = 2
pt.mySky.ComputeTranmittanceLut(); pt._sky_dirty = False
c = pt.BuildRtConsts(); ctx = pt.ctx
ctx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, 4)
ctx.render(c, W, H, 0, 4)
rays = ctx.debug_read_queue(1)       # bounce-1 rays (incoherent secondary rays), compaction order
print(wl, "bounce-1 rays:", len(rays))
info = ctx.scene_info()
lo, hi = np.array(info.bounds_min), np.array(info.bounds_max)

def morton(o, bits=10):
    q = np.clip(((o - lo) / (hi - lo) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    code = np.zeros(len(o), dtype=np.int64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return code

d = rays["dir"]
octant = (d[:, 0] < 0).astype(np.int64) | ((d[:, 1] < 0).astype(np.int64) << 1) | ((d[:, 2] < 0).astype(np.int64) << 2)
mc = morton(rays["origin"])
# direction cell on an octahedral-ish 8x8 grid per octant
dq = (np.clip((np.abs(d[:, :2]) * 4).astype(np.int64), 0, 3))
dircell = octant * 16 + dq[:, 0] * 4 + dq[:, 1]
orders = {
    "as produced (compaction order)": np.arange(len(rays)),
    "random shuffle": np.random.default_rng(0).permutation(len(rays)),
    "octant, stable": np.argsort(octant, kind="stable"),
    "octant then morton(origin)": np.lexsort((mc, octant)),
    "morton(origin) then octant": np.lexsort((octant, mc >> 9)),
    "dir cell (128) then morton": np.lexsort((mc, dircell)),
    "morton(origin) only": np.argsort(mc, kind="stable"),
}
ctx.set_option(api.PT_OPT_TIME_KERNELS, 1)
hits = torch.empty(len(rays) * 4, dtype=torch.float32, device="cuda")
for name, order in orders.items():
    r = np.ascontiguousarray(rays[order])
    t = torch.from_numpy(r.view(np.float32).reshape(-1).copy()).cuda()
    ts = []
    for k in range(5):
        ctx.stats_reset()
        ctx.trace_rays_device(t.data_ptr(), len(r), hits.data_ptr())
        ts.append(ctx.stats().extend_ms)
    ms = float(np.median(ts[1:]))
    print(f"{name:36s} {ms:7.3f} ms  {len(r) / ms / 1e3:8.1f} Mrays/s", flush=True)
