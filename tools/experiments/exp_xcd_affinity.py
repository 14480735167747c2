"""Does giving every XCD its own spatial region of the ray queue pay? (experiment library built with
-DPT_EXP_XCD_AFFINITY: the queue is cut into 8 contiguous segments and an XCD's blocks claim from their own segment
first.) The real bounce-1 queue of the atrium is re-ordered on the host so that segment k holds the rays whose origins
fall into spatial cell k, and pt_trace_rays_device is timed.
usage: PT_AMD_LIB=.../libpt_xcd.so python tools/exp_xcd_affinity.py [atrium|street] [bounce]"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathtracer_amd import PathTracer, api, scenes

wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
bounce = int(sys.argv[2]) if len(sys.argv) > 2 else 1
W, H, SPP = 1920, 1080, 8
scene, cam, depth = {"atrium": (scenes.atrium, scenes.ATRIUM_CAM, 4), "street": (scenes.street, scenes.STREET_CAM, 8)}[wl]
pt = PathTracer(0, W, H); pt.myHalfResRender = False; pt.UpdateOutputTexture()
pt.LoadScene(scene(), cam["eye"]); pt.myCamera.LookAt(cam["eye"], cam["target"])
pt.mySampleSky = True; pt.myLightInstanceIdx = -1; pt.myMaxRecursionDepth = depth
pt.mySky.ComputeTranmittanceLut(); pt._sky_dirty = False
c = pt.BuildRtConsts(); ctx = pt.ctx
ctx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, SPP)
ctx.set_option(api.PT_OPT_FUSED_PRIMARY, 0)
ctx.render(c, W, H, 0, SPP)
rays = ctx.debug_read_queue(bounce)
n = len(rays)
print(f"{wl}: bounce-{bounce} queue holds {n} rays  lib={os.environ.get('PT_AMD_LIB', 'main')}", flush=True)
raw = rays.view(np.float32).reshape(n, 8)
o = raw[:, 0:3]
idx = np.arange(n)
rng = np.random.default_rng(0)

def kd_cells(p, levels=3):
    """2^levels cells of equal population: median splits along the widest axis"""
    cell = np.zeros(len(p), dtype=np.int64)
    groups = [np.arange(len(p))]
    for _ in range(levels):
        nxt = []
        for g in groups:
            ext = p[g].max(0) - p[g].min(0)
            a = int(ext.argmax())
            order = g[np.argsort(p[g, a], kind="stable")]
            h = len(order) // 2
            nxt += [order[:h], order[h:]]
        groups = nxt
    for k, g in enumerate(groups):
        cell[g] = k
    return cell

cell8 = kd_cells(o, 3)
cell64 = kd_cells(o, 6)
orders = {
    "as queued": idx,
    "random shuffle": rng.permutation(n),
    "8 kd cells, queue order inside": np.argsort(cell8, kind="stable"),
    "8 kd cells, shuffled inside": np.lexsort((rng.random(n), cell8)),
    "64 kd cells, queue order inside": np.argsort(cell64, kind="stable"),
}
hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
for name, order in orders.items():
    r = torch.from_numpy(np.ascontiguousarray(raw[order])).cuda()
    torch.cuda.synchronize()
    ts = []
    for rep in range(6):
        ctx.synchronize(); t0 = time.perf_counter()
        ctx.trace_rays_device(r.data_ptr(), n, hits.data_ptr())
        ctx.synchronize(); ts.append(time.perf_counter() - t0)
    t = np.median(ts[1:])
    hv = hits.view(torch.int32)
    chk = int(hv[:, 3].to(torch.int64).sum().item()), float(hits[:, 0].double().sum().item()), float(hits[:, 1].double().sum().item())
    print(f"{name:36s} {t * 1e3:8.3f} ms  {n / t / 1e6:8.1f} Mrays/s   checksum {chk}", flush=True)
