"""Triangle-phase lane utilisation against extend rate: sweeps the deferred schedule's thresholds (postpone %, continue %,
queue depth) on one workload. Utilisation = lanes with work per wave-step, from the COUNT instantiation; rate from HIP
events around the ordinary kernel. usage: python tools/experiments/exp_tri_util.py [atrium|street|street_hard|courtyard] [postpone,continue,queue,lds_stack[,top_nodes[,shared]] ...]
shared = 1: the row runs the shared triangle phase (PT_OPT_SHARED_TRIANGLE_PHASE = 1) and its four numbers set that phase's own knobs."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathtracer_amd import PathTracer, api, scenes

wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
W, H, SPP = 1920, 1080, 8
scene, cam, depth = {"atrium": (scenes.atrium, scenes.ATRIUM_CAM, 4), "street": (scenes.street, scenes.STREET_CAM, 8),
                     "street_hard": (scenes.street_hard, scenes.STREET_HARD_CAM, 8), "courtyard": (scenes.courtyard, scenes.COURTYARD_CAM, 16)}[wl]
pt = PathTracer(0, W, H); pt.myHalfResRender = False; pt.UpdateOutputTexture()
pt.LoadScene(scene(), cam["eye"]); pt.myCamera.LookAt(cam["eye"], cam["target"])
pt.mySampleSky = True; pt.myLightInstanceIdx = -1; pt.myMaxRecursionDepth = depth
pt.mySky.ComputeTranmittanceLut(); pt._sky_dirty = False
c = pt.BuildRtConsts(); ctx = pt.ctx
ctx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, SPP)
grid = [(40, 25, 4, 6), (50, 25, 4, 6), (50, 50, 4, 6), (75, 50, 4, 6), (40, 25, 6, 6), (40, 25, 6, 4), (40, 25, 8, 4), (40, 25, 8, 6), (60, 40, 8, 4), (60, 40, 6, 4),
        (40, 25, 12, 6), (25, 10, 4, 6), (40, 25, 2, 6)]
if len(sys.argv) > 2:
    grid = [tuple(int(x) for x in g.split(",")) for g in sys.argv[2:]]
rows = []
for g in grid:
    (post, cont, defer, lds_stack), top, shared = g[:4], (g[4] if len(g) > 4 else 73), (g[5] if len(g) > 5 else 0)
    ctx.set_option(api.PT_OPT_SHARED_TRIANGLE_PHASE, shared)
    if shared:
        ctx.set_tuning(api.PT_TUNE_SHARED_POSTPONE_PCT, post); ctx.set_tuning(api.PT_TUNE_SHARED_CONTINUE_PCT, cont); ctx.set_tuning(api.PT_TUNE_SHARED_DEFER, defer)
        ctx.set_tuning(api.PT_TUNE_SHARED_TOP_NODES, top)
    else:
        ctx.set_tuning(api.PT_TUNE_EXTEND_POSTPONE_PCT, post); ctx.set_tuning(api.PT_TUNE_EXTEND_CONTINUE_PCT, cont); ctx.set_tuning(api.PT_TUNE_EXTEND_DEFER, defer)
        ctx.set_tuning(api.PT_TUNE_EXTEND_TOP_NODES, top)
    ctx.set_tuning(api.PT_TUNE_EXTEND_LDS_STACK, lds_stack)
    ctx.set_option(api.PT_OPT_COUNT_TRAVERSAL, 1); ctx.stats_reset()
    ctx.render(c, W, H, 0, SPP)
    s = ctx.stats()
    ctx.set_option(api.PT_OPT_COUNT_TRAVERSAL, 0)
    nu, tu = s.nodes_visited / (64.0 * s.node_wave_steps), s.tris_tested / (64.0 * s.tri_wave_steps)
    npr, tpr = s.nodes_visited / s.rays, s.tris_tested / s.rays
    rates = []
    ctx.set_option(api.PT_OPT_TIME_KERNELS, 1)
    for rep in range(3):
        ctx.stats_reset(); ctx.render(c, W, H, 0, 2 * SPP); t = ctx.stats()
        rates.append(t.rays / t.extend_ms / 1e3)
    ctx.set_option(api.PT_OPT_TIME_KERNELS, 0)
    rows.append((post, cont, defer, nu, tu, npr, tpr, float(np.median(rates))))
    print(f"postpone {post:3d}%  continue {cont:3d}%  queue {defer:2d}  lds stack {lds_stack}  top {top:2d}  {'shared  ' if shared else 'per-lane'}:  node-phase util {nu:.3f}  tri-phase util {tu:.3f}  nodes/ray {npr:6.2f}  tris/ray {tpr:5.2f}  extend {np.median(rates):8.1f} Mrays/s", flush=True)
