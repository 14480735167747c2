import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pathtracer_amd import api, scenes
from pathtracer_amd.context import Context
from pathtracer_amd.pathtracer import SetupEarthAtmosphere
from tests.common import lookat_consts
ctx = Context(0); ctx.sky_set(SetupEarthAtmosphere())
rng = np.random.default_rng(11); n = 100000
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
for alt in (1.7, 100.0, 1e3, 1e4, 5e4, 9.9e4, 1.5e5, 1e6):
    for sun in ((0.3, 0.8, 0.52), (0.9, 0.05, 0.1)):
        c = lookat_consts(64, 64, scenes.ATRIUM_CAM, sun_dir=sun)
        api.fill3(c.myCameraPos, (0.0, alt, 0.0))
        rows = np.concatenate([np.tile(np.array([0, alt, 0], dtype=np.float32), (n, 1)), d], axis=1).astype(np.float32)
        m = ctx.debug_eval(api.PT_FN_SKY, rows, c).astype(np.float64)
        l = ctx.debug_eval(api.PT_FN_SKY_VIEW_LUT, rows, c).astype(np.float64)
        up = d[:, 1] > 0
        r_all = ((l - m) ** 2).mean() / max((m ** 2).mean(), 1e-30)
        r_up = ((l[up] - m[up]) ** 2).mean() / max((m[up] ** 2).mean(), 1e-30)
        print(f"alt {alt:9.1f} sun {sun}: relMSE all dirs {r_all:.2e}  upward {r_up:.2e}  finite {np.isfinite(l).all()}", flush=True)
