"""Diagnostic: where does the sky-view LUT (sky mode 1) differ from the marched sky?  Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pathtracer_amd import api, scenes
from pathtracer_amd.context import Context
from pathtracer_amd.pathtracer import SetupEarthAtmosphere
from tests.common import lookat_consts

ctx = Context(0)
ctx.sky_set(SetupEarthAtmosphere())
rng = np.random.default_rng(11)
n = 200000
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
d[:, 1] = np.abs(d[:, 1]); d[: n // 10, 1] *= 0.02
d /= np.linalg.norm(d, axis=1, keepdims=True)
for sun in [(0.0, 1.0, 0.0), (0.3, 0.8, 0.52), (-0.7, 0.12, 0.2), (0.2, 0.02, -0.9)]:
    c = lookat_consts(64, 64, scenes.ATRIUM_CAM, sun_dir=sun)
    rows = np.concatenate([np.tile(np.asarray(c.myCameraPos, dtype=np.float32), (n, 1)), d], axis=1).astype(np.float32)
    m = ctx.debug_eval(api.PT_FN_SKY, rows, c).astype(np.float64)
    l = ctx.debug_eval(api.PT_FN_SKY_VIEW_LUT, rows, c).astype(np.float64)
    s = np.asarray(sun) / np.linalg.norm(sun)
    ang_sun = np.degrees(np.arccos(np.clip(d @ s, -1, 1)))
    elev = np.degrees(np.arcsin(d[:, 1]))
    err = ((l - m) ** 2).sum(1)
    print(f"sun {sun}: relMSE {err.mean() / (m ** 2).sum(1).mean():.3e}  worst/mean {np.abs(l - m).max() / m.mean():.3e}")
    for lo, hi in [(0, 2), (2, 5), (5, 10), (10, 20), (20, 45), (45, 180)]:
        k = (ang_sun >= lo) & (ang_sun < hi)
        if k.any():
            print(f"   angle to sun {lo:3d}-{hi:3d}: n={k.sum():6d} share of sq err {err[k].sum() / err.sum():.3f}  rel rms {np.sqrt(err[k].mean() / (m[k] ** 2).sum(1).mean()):.3e}")
    for lo, hi in [(0, 0.5), (0.5, 2), (2, 10), (10, 45), (45, 90)]:
        k = (elev >= lo) & (elev < hi)
        if k.any():
            print(f"   elevation {lo:4.1f}-{hi:4.1f}: n={k.sum():6d} share of sq err {err[k].sum() / err.sum():.3f}  rel rms {np.sqrt(err[k].mean() / (m[k] ** 2).sum(1).mean()):.3e}")

print("--- worst samples, sun zenith")
c = lookat_consts(64, 64, scenes.ATRIUM_CAM, sun_dir=(0, 1, 0))
print("camera", list(c.myCameraPos))
rows = np.concatenate([np.tile(np.asarray(c.myCameraPos, dtype=np.float32), (n, 1)), d], axis=1).astype(np.float32)
m = ctx.debug_eval(api.PT_FN_SKY, rows, c).astype(np.float64)
l = ctx.debug_eval(api.PT_FN_SKY_VIEW_LUT, rows, c).astype(np.float64)
err = np.abs(l - m).max(1)
for i in np.argsort(-err)[:12]:
    print(f"  elev {np.degrees(np.arcsin(d[i,1])):8.4f}  march {m[i]}  lut {l[i]}")
print("--- march and lut along an elevation sweep (azimuth 30 deg)")
el = np.radians(np.concatenate([np.linspace(0, 0.6, 25), np.linspace(0.7, 5, 10)]))
dd = np.stack([np.cos(el) * np.cos(0.5), np.sin(el), np.cos(el) * np.sin(0.5)], axis=1)
rows = np.concatenate([np.tile(np.asarray(c.myCameraPos, dtype=np.float32), (len(el), 1)), dd], axis=1).astype(np.float32)
m = ctx.debug_eval(api.PT_FN_SKY, rows, c); l = ctx.debug_eval(api.PT_FN_SKY_VIEW_LUT, rows, c)
for e, a, b in zip(np.degrees(el), m, l):
    print(f"  elev {e:7.4f}  march {a[0]:9.4f} {a[2]:9.4f}   lut {b[0]:9.4f} {b[2]:9.4f}")
