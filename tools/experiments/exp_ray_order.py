"""What is ray ORDER worth to k_extend? Takes the real bounce-1 queue of the atrium (secondary rays in the order k_shade's
compaction leaves them), re-orders it on the host in several ways and times pt_trace_rays_device on each.
usage: python tools/exp_ray_order.py [atrium|street|street_hard|courtyard] [bounce]"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathtracer_amd import PathTracer, api, scenes

wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
bounce = int(sys.argv[2]) if len(sys.argv) > 2 else 1
W, H, SPP = 1920, 1080, 4
scene, cam, depth = {"atrium": (scenes.atrium, scenes.ATRIUM_CAM, 4), "street": (scenes.street, scenes.STREET_CAM, 8),
                     "street_hard": (scenes.street_hard, scenes.STREET_HARD_CAM, 8), "courtyard": (scenes.courtyard, scenes.COURTYARD_CAM, 16)}[wl]
pt = PathTracer(0, W, H); pt.myHalfResRender = False; pt.UpdateOutputTexture()
pt.LoadScene(scene(), cam["eye"]); pt.myCamera.LookAt(cam["eye"], cam["target"])
pt.mySampleSky = True; pt.myLightInstanceIdx = -1; pt.myMaxRecursionDepth = depth
pt.mySky.ComputeTranmittanceLut(); pt._sky_dirty = False
c = pt.BuildRtConsts(); ctx = pt.ctx
ctx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, SPP)
ctx.render(c, W, H, 0, SPP)
rays = ctx.debug_read_queue(bounce)
n = len(rays)
print(f"{wl}: bounce-{bounce} queue holds {n} rays", flush=True)
d = np.stack([rays["dx"], rays["dy"], rays["dz"]], 1) if "dx" in rays.dtype.names else None
names = rays.dtype.names
raw = rays.view(np.float32).reshape(n, 8)
o, dd = raw[:, 0:3], raw[:, 4:7]
octant = (dd[:, 0] < 0).astype(np.uint32) | ((dd[:, 1] < 0).astype(np.uint32) << 1) | ((dd[:, 2] < 0).astype(np.uint32) << 2)
idx = np.arange(n)

def morton(p, bits=10):
    lo, hi = p.min(0), p.max(0)
    q = np.clip(((p - lo) / np.maximum(hi - lo, 1e-9) * ((1 << bits) - 1)).astype(np.uint64), 0, (1 << bits) - 1)
    code = np.zeros(len(p), dtype=np.uint64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return code

def block_sort(key, block):
    out = idx.copy()
    for s in range(0, n, block):
        e = min(n, s + block)
        out[s:e] = s + np.argsort(key[s:e], kind="stable")
    return out

# 6-direction-bin key: dominant axis + sign (finer than the octant along the main axis)
dom = np.abs(dd).argmax(1)
face = dom * 2 + (dd[idx, dom] < 0)
orders = {
    "as queued": idx,
    "octant within 512": block_sort(octant, 512),
    "octant within 4096": block_sort(octant, 4096),
    "octant within 64k": block_sort(octant, 65536),
    "octant, whole queue": np.argsort(octant, kind="stable"),
    "octant + 32 direction cells within 64k": block_sort(octant * 1024 + np.clip(((dd[:, 0] * 0.5 + 0.5) * 8).astype(np.uint32), 0, 7) * 8 + np.clip(((dd[:, 2] * 0.5 + 0.5) * 8).astype(np.uint32), 0, 7), 65536),
    "morton(origin) whole queue": np.argsort(morton(o), kind="stable"),
    "octant then morton(origin)": np.lexsort((morton(o), octant)),
    "random shuffle": np.random.default_rng(0).permutation(n),
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
    print(f"{name:42s} {t * 1e3:8.3f} ms  {n / t / 1e6:8.1f} Mrays/s", flush=True)
