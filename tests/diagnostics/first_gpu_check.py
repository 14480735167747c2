import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from pathtracer_amd import PathTracer, scenes, api
from tests import oracle_lib
orc = oracle_lib.load()
pt = PathTracer(0, 128, 128); pt.myHalfResRender = False; pt.UpdateOutputTexture()
sc = scenes.cornell_box()
pt.LoadScene(sc, scenes.CORNELL_CAM_POS)
pt.mySampleSky = False; pt.mySkyFallbackIntensity = 0.0
print("scene info", pt.ctx.scene_info().num_bvh_nodes, pt.ctx.scene_info().bvh_max_depth)
c = pt.BuildRtConsts()
W = H = 128
g = pt.ctx.primary_hits(c, W, H, 0)
os_ = orc.scene(sc)
o = os_.primary_hits(c, W, H, 0)
ob = os_.primary_hits(c, W, H, 0, brute=True)
print("prim mismatches gpu-vs-oracle:", int((g['prim'] != o['prim']).sum()), " oracle bvh-vs-brute:", int((o['prim'] != ob['prim']).sum()))
print("t bit mismatches:", int((g['t'].view(np.uint32) != o['t'].view(np.uint32)).sum()), "u:", int((g['u'].view(np.uint32) != o['u'].view(np.uint32)).sum()))
print("hit fraction", float((g['prim'] != api.PT_MISS).mean()))
pt.RenderRT(16)
img = pt.ReadOutput()
acc, st, ps = os_.render(c, W, H, 0, 16)
ref = acc / 16.0
num = ((img[..., :3] - ref[..., :3]) ** 2).mean(); den = (ref[..., :3] ** 2).mean()
print("relMSE", num / den, "mean gpu", img[..., :3].mean(), "mean ref", ref[..., :3].mean())
s = pt.ctx.stats()
print("gpu stats rays", s.rays, "hits", s.shaded_hits, "sky", s.sky_misses, "samples", s.samples, "| oracle", st)
# sky
pt.mySampleSky = True; pt.RestartAccumulation(); pt.RenderRT(4)
img2 = pt.ReadOutput()
lut = pt.ctx.sky_lut_read(); olut = orc.transmittance_lut(pt.mySky.myAtmosphereParams)
print("lut mismatching halfs", int((lut != olut).sum()), "of", lut.size)
c2 = pt.BuildRtConsts()
acc2, st2, _ = os_.render(c2, W, H, 0, 4, lut=olut)
ref2 = acc2 / 4.0
print("sky relMSE", ((img2[..., :3]-ref2[..., :3])**2).mean() / (ref2[..., :3]**2).mean(), img2[..., :3].mean(), ref2[..., :3].mean())
