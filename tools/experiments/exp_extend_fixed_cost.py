"""Experiment: fixed cost of one k_extend launch. Times the real bounce-1 ray queue of a workload at n, n/2, n/4, ...
(prefixes of a shuffled copy, so every subset has the same ray mix) and fits time = a + b*n."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathtracer_amd import PathTracer, api, scenes

wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
W, H = 1920, 1080
scene, cam, depth = {"atrium": (scenes.atrium, scenes.ATRIUM_CAM, 4), "street": (scenes.street, scenes.STREET_CAM, 8)}[wl]
pt = PathTracer(0, W, H); pt.myHalfResRender = False; pt.UpdateOutputTexture()
pt.LoadScene(scene(), cam["eye"]); pt.myCamera.LookAt(cam["eye"], cam["target"])
pt.mySampleSky = True; pt.myLightInstanceIdx = -1; pt.myMaxRecursionDepth = 2
pt.mySky.ComputeTranmittanceLut(); pt._sky_dirty = False
c = pt.BuildRtConsts(); ctx = pt.ctx
ctx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, 8)
ctx.render(c, W, H, 0, 8)
rays = ctx.debug_read_queue(1)
# keep compaction order locally but decorrelate subsets: shuffle blocks of 4096 rays
nb = len(rays) // 4096
perm = np.random.default_rng(0).permutation(nb)
rays = np.ascontiguousarray(rays[: nb * 4096].reshape(nb, 4096)[perm].reshape(-1))
print(wl, "rays:", len(rays))
t = torch.from_numpy(rays.view(np.float32).reshape(-1).copy()).cuda()
hits = torch.empty(len(rays) * 4, dtype=torch.float32, device="cuda")
ctx.set_option(api.PT_OPT_TIME_KERNELS, 1)
ns, ms = [], []
for div in (1, 2, 4, 8, 16, 32, 64, 128):
    n = len(rays) // div
    ts = []
    for k in range(6):
        ctx.stats_reset()
        ctx.trace_rays_device(t.data_ptr(), n, hits.data_ptr())
        ts.append(ctx.stats().extend_ms)
    m = float(np.median(ts[1:]))
    ns.append(n); ms.append(m)
    print(f"n={n:9d}  {m:8.4f} ms  {n / m / 1e3:8.1f} Mrays/s", flush=True)
A = np.stack([np.ones(len(ns)), np.array(ns, dtype=np.float64)], axis=1)
(a, b), *_ = np.linalg.lstsq(A[:4], np.array(ms[:4]), rcond=None)
print(f"fit on the 4 largest: a = {a * 1e3:.1f} us fixed, asymptotic rate = {1 / b / 1e3:.1f} Mrays/s")
