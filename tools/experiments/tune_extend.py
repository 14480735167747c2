"""Sweeps the extend kernel's tuning options on one workload (interleaved rounds in one process)."""
import sys, os, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathtracer_amd import PathTracer, api, scenes

wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
W, H, SPP = (1920, 1080, 8)
scene, cam, depth = {"atrium": (scenes.atrium, scenes.ATRIUM_CAM, 4), "street": (scenes.street, scenes.STREET_CAM, 8),
                     "courtyard": (scenes.courtyard, scenes.COURTYARD_CAM, 16)}[wl]
pt = PathTracer(0, W, H); pt.myHalfResRender = False; pt.UpdateOutputTexture()
pt.LoadScene(scene(), cam["eye"]); pt.myCamera.LookAt(cam["eye"], cam["target"])
pt.mySampleSky = True; pt.myLightInstanceIdx = -1; pt.myMaxRecursionDepth = depth
pt.mySky.ComputeTranmittanceLut(); pt._sky_dirty = False
c = pt.BuildRtConsts(); ctx = pt.ctx
ctx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, SPP)
ctx.set_option(api.PT_OPT_TIME_KERNELS, 1)
if len(sys.argv) < 3:
    grid = [(f, p, 0, 0, 0) for f in (8, 16, 24, 32, 48) for p in (25, 50, 75)]
else:
    grid = [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]]
    grid = [g + (0,) * (5 - len(g)) if len(g) < 5 else g for g in grid]   # fetch_min, postpone, blocks/CU, lds_stack, defer[, top_nodes]
res = {g: [] for g in grid}
for rnd in range(3):
    for g in grid:
        ctx.set_tuning(api.PT_TUNE_EXTEND_FETCH_MIN, g[0]); ctx.set_tuning(api.PT_TUNE_EXTEND_POSTPONE_PCT, g[1])
        ctx.set_tuning(api.PT_TUNE_EXTEND_LDS_STACK, g[3] if g[3] else 6)
        ctx.set_tuning(api.PT_TUNE_EXTEND_BLOCKS_PER_CU, g[2])
        ctx.set_tuning(api.PT_TUNE_EXTEND_DEFER, g[4] if g[4] else 4)
        ctx.set_tuning(api.PT_TUNE_EXTEND_TOP_NODES, g[5] if len(g) > 5 else 73)
        ctx.set_tuning(api.PT_TUNE_EXTEND_CONTINUE_PCT, g[6] if len(g) > 6 else 25)
        ctx.render(c, W, H, 0, SPP); ctx.stats_reset()
        ctx.render(c, W, H, 0, 2 * SPP)
        s = ctx.stats()
        res[g].append((s.rays / s.extend_ms / 1e3, sum(s.rays_bounce[1:]) / max(sum(s.extend_ms_bounce[1:]), 1e-9) / 1e3, s.rays / s.render_ms / 1e3))
for g in grid:
    a = np.array(res[g])
    print(f"fetch_min={g[0]:2d} postpone={g[1]:2d} blocks/CU={g[2]} lds_stack={g[3]} defer={g[4]} top={g[5] if len(g) > 5 else 73} cont={g[6] if len(g) > 6 else 25}: extend {np.median(a[:,0]):8.1f} Mrays/s  secondary {np.median(a[:,1]):8.1f}  whole {np.median(a[:,2]):8.1f}", flush=True)
