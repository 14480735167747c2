"""Lane utilisation of the extend kernel's node and triangle phases (counting build) for a few option settings."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathtracer_amd import PathTracer, api, scenes
wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
W, H, SPP = 1920, 1080, 4
scene, cam, depth = {"atrium": (scenes.atrium, scenes.ATRIUM_CAM, 4), "street": (scenes.street, scenes.STREET_CAM, 8)}[wl]
pt = PathTracer(0, W, H); pt.myHalfResRender = False; pt.UpdateOutputTexture()
pt.LoadScene(scene(), cam["eye"]); pt.myCamera.LookAt(cam["eye"], cam["target"])
pt.mySampleSky = True; pt.myLightInstanceIdx = -1; pt.myMaxRecursionDepth = depth
pt.mySky.ComputeTranmittanceLut(); pt._sky_dirty = False
c = pt.BuildRtConsts(); ctx = pt.ctx
ctx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, SPP)
ctx.set_option(api.PT_OPT_COUNT_TRAVERSAL, 1)
ctx.set_option(api.PT_OPT_TIME_KERNELS, 1)
for fm, pp, dq in [(16, 35, 0), (16, 20, 4), (16, 35, 4), (16, 50, 4), (16, 65, 4), (16, 50, 2), (16, 50, 8), (8, 50, 4), (24, 50, 4)]:
    ctx.set_tuning(api.PT_TUNE_EXTEND_FETCH_MIN, fm); ctx.set_tuning(api.PT_TUNE_EXTEND_POSTPONE_PCT, pp); ctx.set_tuning(api.PT_TUNE_EXTEND_DEFER, dq)
    ctx.stats_reset(); ctx.render(c, W, H, 0, SPP); s = ctx.stats()
    ctx.set_option(api.PT_OPT_COUNT_TRAVERSAL, 0); ctx.render(c, W, H, 0, SPP); ctx.stats_reset(); ctx.render(c, W, H, 0, 2 * SPP); t = ctx.stats()
    ctx.set_option(api.PT_OPT_COUNT_TRAVERSAL, 1)
    print(f"[{t.rays / t.extend_ms / 1e3:7.1f} Mrays/s] defer={dq} fetch_min={fm:2d} postpone={pp:2d}: nodes/ray {s.nodes_visited/s.rays:6.2f} tris/ray {s.tris_tested/s.rays:5.2f} | "
          f"node steps/ray(wave) {s.node_wave_steps*64/s.rays:6.2f} util {s.nodes_visited/(64*s.node_wave_steps):.3f} | "
          f"tri steps*64/ray {s.tri_wave_steps*64/s.rays:6.2f} util {s.tris_tested/(64*max(1,s.tri_wave_steps)):.3f}", flush=True)
