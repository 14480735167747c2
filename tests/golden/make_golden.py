"""Generates the committed fixtures under tests/golden/ from the CPU oracle.

The reference ships no tests or golden vectors, and cannot be built or run here, so these pins come
from the oracle itself: they freeze its behaviour (regression pins) and give the GPU tests inputs/outputs that do not
need the oracle's BVH at run time. The exact integer KATs (rng_kat.json) were derived by hand from Random.hlsl:45-62,
84-99 in SURVEY §8c and are only re-checked here, never re-generated.
Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pathtracer_amd import api, scenes  # noqa: E402
from tests import oracle_lib  # noqa: E402
from tests.common import cornell_consts, function_inputs  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
os.makedirs(G, exist_ok=True)
orc = oracle_lib.load()

# ---- integer KATs : verify, then store ----
kat = {
    "pcg4d": [
        {"in": [0, 0, 0, 1], "out": [0x8d9effe2, 0xa833f3dc, 0x8246b1b2, 0x58ab6650]},
        {"in": [1, 2, 3, 1], "out": [0x34246d52, 0xd2f8a162, 0x1e43e003, 0x0353573d]},
        {"in": [639, 359, 7, 5], "out": [0x34f030c6, 0x6a1c549b, 0x84a91046, 0x9b04bbde]},
        {"in": [1919, 1079, 4095, 17], "out": [0x715e183a, 0x3cd018f8, 0x387a8603, 0x64535109]},
    ],
    "rand01_bits": [
        {"px": 0, "py": 0, "seed": 0, "bits": [0x3f0d9efe, 0x3f4c7bc6, 0x3f1f0aa2, 0x3f58983a, 0x3e97dc10]},
        {"px": 256, "py": 256, "seed": 3, "bits": [0x3f31a8b2, 0x3f2eaf56, 0x3ebb83fc, 0x3f4d1f7a, 0x3e9285c4]},
        {"px": 1919, "py": 1079, "seed": 255, "bits": [0x3ba5da00, 0x3f3af06c, 0x3f07bab6, 0x3f4ffd7c, 0x3e6cd9a8]},
    ],
}
for k in kat["pcg4d"]:
    assert list(orc.pcg4d(k["in"])) == k["out"], k
for k in kat["rand01_bits"]:
    assert list(orc.rand01(k["px"], k["py"], k["seed"], 5).view(np.uint32)) == k["bits"], k
json.dump(kat, open(os.path.join(G, "rng_kat.json"), "w"), indent=1)

# ---- Cornell 64x64 primary-hit map (ids exact; brute force == BVH checked) ----
sc = scenes.cornell_box()
osc = orc.scene(sc)
c = cornell_consts(64, 64)
h_bvh = osc.primary_hits(c, 64, 64, 0)
h_brute = osc.primary_hits(c, 64, 64, 0, brute=True)
assert np.array_equal(h_bvh["prim"], h_brute["prim"]) and np.array_equal(h_bvh["t"].view(np.uint32), h_brute["t"].view(np.uint32))
np.save(os.path.join(G, "cornell_hits_64.npy"), h_bvh)

# ---- Cornell 128^2, 16 spp radiance (sum target / 16), sky off, light on ----
c = cornell_consts(128, 128)
acc, st, _ = osc.render(c, 128, 128, 0, 16)
np.save(os.path.join(G, "cornell_128_16spp.npy"), (acc[..., :3] / 16.0).astype(np.float32))
json.dump({k: int(v) for k, v in st.items() if k in ("rays", "shaded_hits", "sky_misses", "samples")},
          open(os.path.join(G, "cornell_128_16spp_stats.json"), "w"))

# ---- transmittance LUT: checksum + 16 probe texels ----
atm = orc.earth_atmosphere()
lut = orc.transmittance_lut(atm)
probes = [(x, y) for y in (0, 21, 42, 63) for x in (0, 85, 170, 255)]
json.dump({"sum_u16": int(lut.astype(np.uint64).sum()), "xor_u16": int(np.bitwise_xor.reduce(lut.reshape(-1))),
           "probes": [{"x": x, "y": y, "rgba16f": [int(v) for v in lut[y, x]]} for x, y in probes]},
          open(os.path.join(G, "transmittance_lut_probes.json"), "w"), indent=1)

# ---- per-function tables (256 rows each), seeded inputs + oracle outputs ----
tables = {}
c_sky = cornell_consts(64, 64, sky=True)
for fn in range(1, api.PT_FN_COUNT if hasattr(api, "PT_FN_COUNT") else 9):
    x = function_inputs(fn, 256, seed=100 + fn)
    y = orc.eval(fn, x, consts=c_sky, lut=lut)
    tables[f"in_{fn}"] = x
    tables[f"out_{fn}"] = y
np.savez_compressed(os.path.join(G, "function_tables.npz"), **tables)
# ---- AO pipeline (Ao.hlsl): Cornell 64^2, 4 spp, sky off, AO distance 25: occluded-ray counts are integers ----
c = cornell_consts(64, 64, sky=False, fallback=0.0)
c.myAoDistance = 25.0
acc, st = osc.render_ao(c, 64, 64, 0, 4)
acc_b, _ = osc.render_ao(c, 64, 64, 0, 4, brute=True)
assert np.array_equal(acc, acc_b)
np.save(os.path.join(G, "cornell_ao_64_4spp.npy"), acc[..., 0].astype(np.float32))
json.dump({k: int(v) for k, v in st.items() if k in ("rays", "shaded_hits", "sky_misses", "samples")},
          open(os.path.join(G, "cornell_ao_64_4spp_stats.json"), "w"))

# ---- Russian roulette (extension): Cornell 64^2, 8 spp, depth 12: ray counts with and without ----
c = cornell_consts(64, 64, depth=12, sky=False, fallback=0.0)
rr = {}
for on in (0, 1):
    orc.set_russian_roulette(on)
    acc, st, _ = osc.render(c, 64, 64, 0, 8)
    rr[str(on)] = {"rays": st["rays"], "shaded_hits": st["shaded_hits"], "sky_misses": st["sky_misses"],
                   "sum_rgb": float(acc[..., :3].astype(np.float64).sum())}
orc.set_russian_roulette(0)
json.dump(rr, open(os.path.join(G, "cornell_rr_64_8spp_stats.json"), "w"), indent=1)

# ---- path probe (ClosestHit / payload rows a7-a9): the rays every path of sample 3 carries into bounces 1 and 2, Cornell 64^2 ----
c = cornell_consts(64, 64)
probe = {}
for b in (1, 2):
    r, alive, state = osc.bounce_rays(c, 64, 64, 3, b)
    probe[f"alive_{b}"] = np.packbits(alive)
    probe[f"rays_{b}"] = r[alive]
    probe[f"state_{b}"] = state[alive]
np.savez_compressed(os.path.join(G, "cornell_bounce_rays_64.npz"), **probe)

print("golden fixtures written to", G, sorted(os.listdir(G)))
