import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pathtracer_amd import api, scenes
from pathtracer_amd.context import Context
from tests import oracle_lib
from tests.common import lookat_consts
orc = oracle_lib.load()
ctx = Context(0)
atm = orc.earth_atmosphere(); ctx.sky_set(atm); lut = orc.transmittance_lut(atm)
rng = np.random.default_rng(0)
n = 20000
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
for name, ylo, yhi in (("ground..2km", 0, 2e3), ("2..90 km", 2e3, 9e4), ("just above the top", 1.0e5, 1.2e5), ("orbit", 3e5, 5e6), ("below ground", -5e3, -1)):
    pos = np.stack([rng.uniform(-1e4, 1e4, n), rng.uniform(ylo, yhi, n), rng.uniform(-1e4, 1e4, n)], axis=1)
    rows = np.concatenate([pos, d], axis=1).astype(np.float32)
    for sun in ((0.3, 0.8, 0.52), (0.0, -0.2, 0.98), (1.0, 0.0, 0.0)):
        c = lookat_consts(64, 64, scenes.ATRIUM_CAM, sun_dir=sun)
        g = ctx.debug_eval(api.PT_FN_SKY, rows, c).astype(np.float64)
        o = orc.eval(api.PT_FN_SKY, rows, consts=c, lut=lut).astype(np.float64)
        fin = np.isfinite(o).all(1) & np.isfinite(g).all(1)
        err = np.abs(g[fin] - o[fin]).max() / max(np.abs(o[fin]).max(), 1e-30)
        print(f"{name:20s} sun {sun}: non-finite oracle {int((~np.isfinite(o).all(1)).sum())} gpu {int((~np.isfinite(g).all(1)).sum())}  nan mismatch {int((np.isfinite(o).all(1) != np.isfinite(g).all(1)).sum())}  max|diff|/max {err:.2e}  mean {o[fin].mean():.4f}", flush=True)
