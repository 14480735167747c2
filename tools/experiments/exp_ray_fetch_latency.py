"""What does the HBM latency of the ray fetch cost k_extend? The bounce-1 queue of the atrium is traced (a) after the
caches were flushed (the rays come from HBM, as in a render) and (b) after a pass that just read the ray array (it sits
in the Infinity Cache / L2).  usage: python tools/exp_ray_fetch_latency.py [spp]"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathtracer_amd import PathTracer, api, scenes
SPP = int(sys.argv[1]) if len(sys.argv) > 1 else 2
W, H = 1920, 1080
pt = PathTracer(0, W, H); pt.myHalfResRender = False; pt.UpdateOutputTexture()
cam = scenes.ATRIUM_CAM
pt.LoadScene(scenes.atrium(), cam["eye"]); pt.myCamera.LookAt(cam["eye"], cam["target"])
pt.mySampleSky = True; pt.myLightInstanceIdx = -1; pt.myMaxRecursionDepth = 1
pt.mySky.ComputeTranmittanceLut(); pt._sky_dirty = False
c = pt.BuildRtConsts(); ctx = pt.ctx
ctx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, SPP)
ctx.render(c, W, H, 0, SPP)
rays = ctx.debug_read_queue(1)
n = len(rays)
raw = torch.from_numpy(np.ascontiguousarray(rays.view(np.float32).reshape(n, 8))).cuda()
hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
junk = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
print(f"{n} rays ({n * 32 / 1e6:.0f} MB of ray records)", flush=True)
def run(warm):
    ts = []
    for rep in range(7):
        junk.add_(1); torch.cuda.synchronize()               # 2 GB of traffic: nothing of the rays is left in any cache
        if warm:
            _ = raw.sum().item(); torch.cuda.synchronize()    # ... and now they are back in the last-level caches
        ctx.synchronize(); t0 = time.perf_counter()
        ctx.trace_rays_device(raw.data_ptr(), n, hits.data_ptr())
        ctx.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts[1:]))
for warm in (False, True, False, True):
    t = run(warm)
    print(f"rays {'warm in cache' if warm else 'cold (HBM)   '}: {t * 1e3:8.3f} ms  {n / t / 1e6:8.1f} Mrays/s", flush=True)
