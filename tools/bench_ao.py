"""AO pipeline throughput (pt_render_ao) on a bench workload at 1080p.  Run on the GPU box."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pathtracer_amd import PathTracer, api, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
scene, cam = {"atrium": (scenes.atrium, scenes.ATRIUM_CAM), "street": (scenes.street, scenes.STREET_CAM)}[name]
W, H, K = 1920, 1080, 10
pt = PathTracer(0, W, H)
pt.myHalfResRender = False
pt.UpdateOutputTexture()
pt.LoadScene(scene(), cam["eye"])
pt.myCamera.LookAt(cam["eye"], cam["target"])
pt.myLightInstanceIdx = -1
pt.myRenderAo = True
for dist in (1.0, 5.0):
    pt.myAoDistance = dist
    pt.RestartAccumulation()
    pt.RenderRT(2)
    pt.ctx.synchronize()
    pt.ctx.stats_reset()
    pt.ctx.set_option(api.PT_OPT_TIME_KERNELS, 1)
    t0 = time.perf_counter()
    pt.RenderRT(K)
    pt.ctx.synchronize()
    dt = time.perf_counter() - t0
    st = pt.ctx.stats()
    pt.ctx.set_option(api.PT_OPT_TIME_KERNELS, 0)
    ao_rays = st.rays_bounce[1]
    print(json.dumps({"workload": name, "ao_distance": dist, "ms_per_frame": round(dt / K * 1e3, 3), "mrays_per_s": round(st.rays / dt / 1e6, 1),
                      "ao_rays_per_frame": ao_rays // K, "ao_extend_mrays_per_s": round(ao_rays / max(st.extend_ms_bounce[1], 1e-9) / 1e3, 1),
                      "primary_extend_mrays_per_s": round(st.rays_bounce[0] / max(st.extend_ms_bounce[0], 1e-9) / 1e3, 1),
                      "mean_ao": round(float(pt.ReadOutput()[..., 0].mean()), 4)}))
