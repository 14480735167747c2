"""Experiment: does running two half-batches on two streams (two contexts) overlap shade with extend?"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pathtracer_amd import PathTracer, api, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
scene_fn, cam = {"atrium": (scenes.atrium, scenes.ATRIUM_CAM), "street": (scenes.street, scenes.STREET_CAM)}[name]
depth = 4 if name == "atrium" else 8
W, H, K = 1920, 1080, 20
scene = scene_fn()

def make(spp):
    pt = PathTracer(0, W, H)
    pt.myHalfResRender = False
    pt.UpdateOutputTexture()
    pt.LoadScene(scene, cam["eye"])
    pt.myCamera.LookAt(cam["eye"], cam["target"])
    pt.myLightInstanceIdx = -1
    pt.myMaxRecursionDepth = depth
    pt.mySky.mySunDir = np.array([0.3, 0.8, 0.52], dtype=np.float32) / np.float32(np.linalg.norm([0.3, 0.8, 0.52]))
    pt.mySky.ComputeTranmittanceLut(); pt._sky_dirty = False
    pt.ctx.accum_reset(W, H)
    pt.ctx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, spp)
    return pt

def run(pts, spp_each, blocks_per_cu=0):
    for p in pts:
        p.ctx.set_tuning(api.PT_TUNE_EXTEND_BLOCKS_PER_CU, blocks_per_cu)
    cs = [p.BuildRtConsts() for p in pts]
    def step(i):
        for k, p in enumerate(pts):
            p.ctx.render(cs[k], W, H, (i * len(pts) + k) * spp_each, spp_each)
    for i in range(3):
        step(i)
    for p in pts: p.ctx.synchronize()
    for p in pts: p.ctx.stats_reset()
    t0 = time.perf_counter()
    for i in range(K):
        step(3 + i)
    for p in pts: p.ctx.synchronize()
    dt = time.perf_counter() - t0
    rays = sum(p.ctx.stats().rays for p in pts)
    return dt / K * 1e3, rays / dt / 1e6

cfgs = [(1, 16), (2, 8), (2, 16), (3, 8), (1, 32)]
for nctx, spp in cfgs:
    pts = [make(spp) for _ in range(nctx)]
    ms, mr = run(pts, spp)
    print(f"{nctx} ctx x {spp} spp/step: {ms:8.3f} ms/step  {mr:8.1f} Mrays/s", flush=True)
    for p in pts: p.ctx.close()
    del pts
