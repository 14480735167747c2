import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathtracer_amd import PathTracer, scenes, api
pt = PathTracer(0, 1280, 720)
pt.LoadScene(scenes.atrium(1), scenes.ATRIUM_CAM["eye"]); pt.myCamera.LookAt(scenes.ATRIUM_CAM["eye"], scenes.ATRIUM_CAM["target"]); pt.myLightInstanceIdx = -1
for lanes in (1, 2, 4, 6, 8):
    pt.ctx.set_option(api.PT_OPT_FRAME_LANES, lanes)
    pt.RestartAccumulation(); pt.RenderRT(1); pt.ctx.synchronize()
    for _ in range(20): pt.RenderRT(1)
    pt.ctx.synchronize()
    n = 300; t0 = time.perf_counter()
    for _ in range(n): pt.RenderRT(1)
    pt.ctx.synchronize(); dt = time.perf_counter() - t0
    # host-side cost of a call alone
    t1 = time.perf_counter()
    for _ in range(n): pt.BuildRtConsts()
    th = (time.perf_counter() - t1) / n
    print(f"lanes {lanes}: {dt / n * 1e3:.3f} ms/call  (BuildRtConsts {th * 1e3:.3f} ms)", flush=True)
