import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathtracer_amd import PathTracer, api, scenes
W, H, SPP = 1920, 1080, 4
pt = PathTracer(0, W, H); pt.myHalfResRender = False; pt.UpdateOutputTexture()
pt.LoadScene(scenes.atrium(), scenes.ATRIUM_CAM["eye"]); pt.myCamera.LookAt(scenes.ATRIUM_CAM["eye"], scenes.ATRIUM_CAM["target"])
pt.mySampleSky = True; pt.myLightInstanceIdx = -1; pt.myMaxRecursionDepth = 4
pt.mySky.ComputeTranmittanceLut(); pt._sky_dirty = False
c = pt.BuildRtConsts(); ctx = pt.ctx
ctx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, SPP); ctx.set_option(api.PT_OPT_TIME_KERNELS, 1)
for dq in (4, 0):
    ctx.set_tuning(api.PT_TUNE_EXTEND_DEFER, dq)
    for i in range(24):
        ctx.stats_reset(); ctx.render(c, W, H, i * SPP, SPP); s = ctx.stats()
        print(dq, i, " ".join(f"{s.extend_ms_bounce[b]:7.3f}" for b in range(5)), f"render {s.render_ms:7.3f}", flush=True)
