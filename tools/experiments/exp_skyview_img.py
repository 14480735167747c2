"""Diagnostic: sky mode 1 vs mode 0 images on the small atrium.  Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pathtracer_amd import api, scenes
from pathtracer_amd.context import Context
from pathtracer_amd.pathtracer import SetupEarthAtmosphere
from tests.common import lookat_consts

ctx = Context(0)
ctx.sky_set(SetupEarthAtmosphere())
ctx.scene_create(scenes.atrium(1))
w, h = 240, 135
for depth in (0, 1, 4):
    c = lookat_consts(w, h, scenes.ATRIUM_CAM, depth=depth)
    imgs = []
    for mode in (0, 1):
        ctx.set_option(api.PT_OPT_SKY_MODE, mode)
        ctx.accum_reset(w, h)
        ctx.render(c, w, h, 0, 4)
        imgs.append(ctx.accum_read()[..., :3].astype(np.float64))
    a, b = imgs
    print(f"depth {depth}: mean0 {a.mean():.4f} mean1 {b.mean():.4f} relMSE {((a-b)**2).mean()/(a**2).mean():.3e}")
    e = ((a - b) ** 2).sum(-1)
    ys, xs = np.unravel_index(np.argsort(-e.ravel())[:8], e.shape)
    for y, x in zip(ys, xs):
        print(f"   px ({x},{y}) mode0 {a[y,x]} mode1 {b[y,x]}")
    print("   rows with error:", np.nonzero(e.sum(1) > 1e-6 * e.sum())[0][:40])
info = ctx.scene_info()
print("scene bounds", list(info.bounds_min), list(info.bounds_max))
