"""The reference's interactive use: 1280x720 window, half-res 640x360 target, ONE sample per RenderRT call
(PathTracer.cpp:573-595). Measures frames/s of that call pattern (tiny wavefront batches)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pathtracer_amd import PathTracer, scenes

for name, scene, cam in (("cornell", scenes.cornell_box(), None), ("atrium", scenes.atrium(1), scenes.ATRIUM_CAM)):
    pt = PathTracer(0, 1280, 720)           # reference defaults: half-res, depth 4
    if cam is None:
        pt.LoadScene(scene, scenes.CORNELL_CAM_POS); pt.mySampleSky = False
    else:
        pt.LoadScene(scene, cam["eye"]); pt.myCamera.LookAt(cam["eye"], cam["target"]); pt.myLightInstanceIdx = -1
    for per_call in (1, 4, 16):
        pt.RestartAccumulation()
        pt.RenderRT(per_call); pt.ctx.synchronize()
        n = 200 // per_call
        t0 = time.perf_counter()
        for _ in range(n):
            pt.RenderRT(per_call)
        pt.ctx.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name:8s} {pt.myOutputWidth}x{pt.myOutputHeight}  {per_call:2d} spp/call: {dt / n * 1e3:7.3f} ms/call  {n * per_call / dt:8.1f} spp/s  "
              f"{n * per_call * pt.myOutputWidth * pt.myOutputHeight / dt / 1e6:7.1f} Msamples/s", flush=True)
