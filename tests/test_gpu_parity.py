"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle and the committed fixtures.
Bars: bit-exact for integer / index work (RNG, triangle ids, t/u/v of the shared watertight test, LUT halfs);
rtol for fp32 shading functions that call transcendentals (libm vs ocml); image relMSE < 1e-4 (north_star)."""
import json
import os

import numpy as np
import pytest

from pathtracer_amd import api, scenes
from pathtracer_amd.context import SceneData
from tests.common import cornell_consts, function_inputs, lookat_consts, object_splits_only, random_rays, rel_mse

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def cornell(ctx, oracle):
    sc = scenes.cornell_box()
    ctx.scene_create(sc)
    return sc, oracle.scene(sc)


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


# ---------------------------------------------------------------------------------------------- integer / exact
def test_rand01_kats_on_device(ctx):
    kat = json.load(open(os.path.join(G, "rng_kat.json")))
    rows = [[k["px"], k["py"], k["seed"], 5] for k in kat["rand01_bits"]]
    out = ctx.debug_eval(api.PT_FN_RAND01, np.array(rows, dtype=np.float32))
    for k, o in zip(kat["rand01_bits"], out):
        assert list(_bits(o[:5])) == k["bits"]


def test_rand01_matches_oracle_bitwise(ctx, oracle):
    x = function_inputs(api.PT_FN_RAND01, 2048, 9)
    assert np.array_equal(_bits(ctx.debug_eval(api.PT_FN_RAND01, x)), _bits(oracle.eval(api.PT_FN_RAND01, x)))


def test_transmittance_lut_matches_oracle(ctx, oracle):
    atm = oracle.earth_atmosphere()
    ctx.sky_set(atm)
    lut, olut = ctx.sky_lut_read(), oracle.transmittance_lut(atm)
    diff = lut.astype(np.int32) - olut.astype(np.int32)
    # expf differs by <= 1 ulp between ocml and libm: allow a handful of halfs to round the other way by one step
    assert (np.abs(diff) <= 1).all() and (diff != 0).mean() < 1e-3
    g = json.load(open(os.path.join(G, "transmittance_lut_probes.json")))
    for p in g["probes"]:
        assert np.abs(lut[p["y"], p["x"]].astype(np.int32) - np.array(p["rgba16f"])).max() <= 1


# ---------------------------------------------------------------------------------------------- shading functions
@pytest.mark.parametrize("fn,rtol,exact", [
    (api.PT_FN_FRESNEL, 2e-6, False), (api.PT_FN_SPEC_PROB, 0, True), (api.PT_FN_SAMPLE_PHONG, 2e-5, False),
    (api.PT_FN_EVAL_PHONG, 5e-5, False), (api.PT_FN_COS_HEMISPHERE, 2e-5, False), (api.PT_FN_LAMBERT, 0, True),
    (api.PT_FN_PRIMARY_RAY, 0, True)])
def test_shading_functions_match_oracle_and_fixture(ctx, oracle, fn, rtol, exact):
    c = cornell_consts(64, 64, sky=True)
    x = function_inputs(fn, 4096, seed=100 + fn)
    got, want = ctx.debug_eval(fn, x, consts=c), oracle.eval(fn, x, consts=c)
    if exact:      # no transcendental on the path: +,-,*,/,sqrt are IEEE on both sides
        assert np.array_equal(_bits(got), _bits(want))
    else:
        np.testing.assert_allclose(got, want, rtol=rtol, atol=1e-5, equal_nan=True)   # atol: components of unit vectors near 0
    t = np.load(os.path.join(G, "function_tables.npz"))
    x256 = function_inputs(fn, 256, seed=100 + fn)
    assert np.array_equal(x256, t[f"in_{fn}"])
    np.testing.assert_allclose(ctx.debug_eval(fn, x256, consts=c), t[f"out_{fn}"], rtol=max(rtol, 2e-6), atol=1e-5, equal_nan=True)


def test_sky_function_matches_oracle(ctx, oracle):
    atm = oracle.earth_atmosphere()
    ctx.sky_set(atm)
    olut = oracle.transmittance_lut(atm)
    c = cornell_consts(64, 64, sky=True)
    x = function_inputs(api.PT_FN_SKY, 4096, seed=107)
    got, want = ctx.debug_eval(api.PT_FN_SKY, x, consts=c), oracle.eval(api.PT_FN_SKY, x, consts=c, lut=olut)
    assert np.isfinite(got).all()
    # positions ~6.36e6 m in fp32 amplify 1-ulp expf differences: relative to the brightest channel of each row
    scale = np.maximum(np.abs(want).max(axis=1, keepdims=True), 1e-3)
    assert (np.abs(got - want) / scale).max() < 2e-3
    assert np.median(np.abs(got - want) / scale) < 1e-5
    c.mySampleSky = 0
    api.fill3(c.mySkyFallbackEmission, (3.0, 2.0, 1.0))
    assert np.array_equal(ctx.debug_eval(api.PT_FN_SKY, x[:4], consts=c), np.tile([3.0, 2.0, 1.0], (4, 1)).astype(np.float32))


# ---------------------------------------------------------------------------------------------- intersection
@pytest.mark.parametrize("ylo,yhi", [(2e3, 9e4), (1.0e5, 1.2e5), (3e5, 5e6), (-5e3, -1.0)])
def test_sky_function_at_extreme_altitudes_and_sun_angles(ctx, oracle, ylo, yhi):
    """SampleSkyLuminance away from the ground-level inputs of the fixture: inside the upper atmosphere, just above
    its top (MoveToTopAtmosphere, sky/Common.hlsl:320-339), from orbit and from below the surface (where the reference
    arithmetic produces NaNs: the NaN pattern must be the same), with the sun high, below the horizon and on it."""
    atm = oracle.earth_atmosphere()
    ctx.sky_set(atm)
    lut = oracle.transmittance_lut(atm)
    rng = np.random.default_rng(0)
    n = 8000
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pos = np.stack([rng.uniform(-1e4, 1e4, n), rng.uniform(ylo, yhi, n), rng.uniform(-1e4, 1e4, n)], axis=1)
    rows = np.concatenate([pos, d], axis=1).astype(np.float32)
    for sun in ((0.3, 0.8, 0.52), (0.0, -0.2, 0.98), (1.0, 0.0, 0.0)):
        c = lookat_consts(64, 64, scenes.ATRIUM_CAM, sun_dir=sun)
        g = ctx.debug_eval(api.PT_FN_SKY, rows, c).astype(np.float64)
        o = oracle.eval(api.PT_FN_SKY, rows, consts=c, lut=lut).astype(np.float64)
        fo, fg = np.isfinite(o).all(1), np.isfinite(g).all(1)
        assert np.array_equal(fo, fg), sun
        assert np.abs(g[fo] - o[fo]).max() <= 1e-4 * max(np.abs(o[fo]).max(), 1e-12), sun


def test_cornell_primary_hits_bit_identical(ctx, cornell):
    sc, osc = cornell
    for (w, h, sample) in ((64, 64, 0), (512, 512, 0), (512, 512, 63), (33, 17, 5), (1, 1, 0)):
        c = cornell_consts(w, h)
        got, want = ctx.primary_hits(c, w, h, sample), osc.primary_hits(c, w, h, sample)
        assert np.array_equal(got["prim"], want["prim"]), (w, h, sample)
        assert np.array_equal(_bits(got["t"]), _bits(want["t"]))
        assert np.array_equal(_bits(got["u"]), _bits(want["u"])) and np.array_equal(_bits(got["v"]), _bits(want["v"]))
    g = np.load(os.path.join(G, "cornell_hits_64.npy"))
    got = ctx.primary_hits(cornell_consts(64, 64), 64, 64, 0)
    assert np.array_equal(got["prim"], g["prim"]) and np.array_equal(_bits(got["t"]), _bits(g["t"]))


@pytest.mark.parametrize("n_tris,n_rays", [(1, 2000), (3, 2000), (4, 2000), (500, 20000), (20000, 200000)])
def test_random_soup_incoherent_rays_match_oracle(ctx, oracle, n_tris, n_rays):
    sc = scenes.random_triangle_soup(21 + n_tris, n_tris)
    ctx.scene_create(sc)
    osc = oracle.scene(sc)
    rays = random_rays(n_tris, n_rays, (0, 0, 0), (10, 10, 10))
    got = ctx.trace_rays(rays)
    want = osc.trace_rays(rays, brute=(n_tris <= 500))
    assert np.array_equal(got["prim"], want["prim"])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    if n_tris >= 500:
        assert (got["prim"] != api.PT_MISS).mean() > 0.05


def test_atrium_primary_and_secondary_rays_bit_identical(ctx, oracle):
    sc = scenes.atrium(1)
    ctx.scene_create(sc)
    info = ctx.scene_info()
    assert info.num_triangles == 262144 and info.num_bvh_nodes > 1000 and info.bvh_max_depth >= 4
    osc = oracle.scene(sc)
    w, h = 320, 180
    c = lookat_consts(w, h, scenes.ATRIUM_CAM)
    got, (want, rays) = ctx.primary_hits(c, w, h, 3), osc.primary_hits(c, w, h, 3, want_rays=True)
    assert np.array_equal(got["prim"], want["prim"])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert (got["prim"] != api.PT_MISS).mean() > 0.5
    # incoherent secondary rays: from the primary hit points into random directions, tmin 0.001 (PathTracing.hlsl:127)
    hit = (want["prim"] != api.PT_MISS).reshape(-1)
    o = (rays["origin"] + rays["dir"] * want["t"][..., None]).reshape(-1, 3)[hit]
    sec = random_rays(5, len(o), (0, 0, 0), (1, 1, 1), tmin=0.001)
    sec["origin"] = o
    g2, w2 = ctx.trace_rays(sec), osc.trace_rays(sec)
    assert np.array_equal(g2["prim"], w2["prim"]) and np.array_equal(g2.view(np.uint32), w2.view(np.uint32))


def test_edge_cases_empty_scene_axis_rays_degenerate_triangles(ctx, oracle):
    v = np.zeros(0, dtype=api.VERTEX_DTYPE)
    empty = SceneData([(v, np.zeros(0, dtype=np.uint32))], [(0, 0, None)], [((0, 0, 0), (1, 1, 1, 1))])
    ctx.scene_create(empty)
    assert ctx.scene_info().num_triangles == 0
    rays = random_rays(1, 1000, (0, 0, 0), (1, 1, 1))
    h = ctx.trace_rays(rays)
    assert (h["prim"] == api.PT_MISS).all() and (h["t"] == -1).all()
    assert len(ctx.trace_rays(rays[:0])) == 0
    # axis-aligned rays (zero direction components), rays inside flat boxes, zero-area triangles
    v = np.zeros(9, dtype=api.VERTEX_DTYPE)
    v["pos"] = [[0, 0, 0], [4, 0, 0], [0, 4, 0], [0, 0, 2], [4, 0, 2], [0, 4, 2], [1, 1, 1], [1, 1, 1], [1, 1, 1]]
    sc = SceneData([(v, np.arange(9, dtype=np.uint32))], [(0, 0, None)], [((0, 0, 0), (1, 1, 1, 1))])
    ctx.scene_create(sc)
    osc = oracle.scene(sc)
    r = np.zeros(8, dtype=api.RAY_DTYPE)
    r["origin"] = [[1, 1, -1], [1, 1, 5], [1, 1, 1], [-1, 1, 0], [1, 1, -1], [0, 0, -1], [2, 2, -1], [1, 1, 0]]
    r["dir"] = [[0, 0, 1], [0, 0, -1], [0, 0, 1], [1, 0, 0], [0, 0, -1], [0, 0, 1], [0, 0, 1], [0, 0, 1]]
    r["tmax"] = 10000.0
    got, want = ctx.trace_rays(r), osc.trace_rays(r, brute=True)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert list(got["prim"][:3]) == [0, 1, 1]


@pytest.mark.parametrize("name", ["nan_vertex", "inf_vertex", "many_nan", "far_vertex", "all_identical", "on_a_line", "tiny", "huge"])
def test_pathological_vertices_do_not_break_builder_or_kernels(ctx, oracle, name):
    """NaN / Inf vertices make a triangle inactive (it can never be hit) without disturbing the others; one vertex at
    1e30 must not poison the SAH costs; degenerate and extreme-scale scenes still build. Checked against the oracle's
    brute-force loop over all triangles."""
    base = scenes.random_triangle_soup(3, 3000)
    v, i = base.meshes[0]
    v = v.copy()
    edit = {
        "nan_vertex": lambda: v["pos"].__setitem__((7, 1), np.nan),
        "inf_vertex": lambda: v["pos"].__setitem__((11, 0), np.inf),
        "many_nan": lambda: v["pos"].__setitem__((slice(0, 100), 2), np.nan),
        "far_vertex": lambda: v["pos"].__setitem__((5, slice(None)), 1e30),
        "all_identical": lambda: v["pos"].__setitem__(slice(None), 1.0),
        "on_a_line": lambda: v["pos"].__setitem__((slice(None), slice(1, 3)), 0.0),
        "tiny": lambda: v["pos"].__imul__(np.float32(1e-20)),
        "huge": lambda: v["pos"].__imul__(np.float32(1e18)),
    }[name]
    edit()
    sc = SceneData([(v, i)], base.instances, base.materials)
    ctx.scene_create(sc)
    osc = oracle.scene(sc)
    scale = {"tiny": 1e-20, "huge": 1e18}.get(name, 1.0)
    rays = random_rays(1, 4000, [-10 * scale] * 3, [10 * scale] * 3, tmax=1e30)
    got = ctx.trace_rays(rays)
    ref = osc.trace_rays(rays, brute=True)
    assert np.array_equal(got["prim"], ref["prim"])
    if name in ("nan_vertex", "inf_vertex", "many_nan", "far_vertex"):
        assert (ref["prim"] != api.PT_MISS).mean() > 0.05      # the healthy triangles are still found


def deep_hierarchy_scene():
    """Geometrically growing triangles: a hierarchy far deeper than the LDS part of the traversal stack, plus rays."""
    n = 3000
    k = np.arange(n, dtype=np.float64)
    s = 1.028 ** k
    v = np.zeros(3 * n, dtype=api.VERTEX_DTYPE)
    pos = np.zeros((n, 3, 3))
    pos[:, 0] = np.stack([s, 0 * s, 0.001 * k], axis=1)
    pos[:, 1] = np.stack([-s, s, 0.001 * k], axis=1)
    pos[:, 2] = np.stack([-s, -s, 0.001 * k], axis=1)
    v["pos"] = pos.reshape(-1, 3).astype(np.float32)
    v["normal"] = (0, 0, 1)
    idx = np.arange(3 * n, dtype=np.uint32)
    sc = SceneData([(v, idx)], [(0, 0, None)], [((0, 0, 0), (0.5, 0.5, 0.5, 1))])
    rng = np.random.default_rng(2)
    rays = np.zeros(6000, dtype=api.RAY_DTYPE)
    r = 1.028 ** rng.uniform(0, n, 6000)
    ang = rng.uniform(0, 2 * np.pi, 6000)
    rays["origin"] = np.stack([0.3 * r * np.cos(ang), 0.3 * r * np.sin(ang), np.full(6000, -5.0)], axis=1).astype(np.float32)
    d = np.stack([rng.normal(0, 0.2, 6000), rng.normal(0, 0.2, 6000), np.ones(6000)], axis=1)
    rays["dir"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays["tmin"], rays["tmax"] = 0.0, 1e30
    return sc, rays


def test_deep_hierarchy_with_every_stack_configuration(ctx, oracle):
    """A hierarchy far deeper than the LDS part of the traversal stack: the LDS clamp, the spill region and the
    deferred queue must all give the brute-force answer."""
    sc, rays = deep_hierarchy_scene()
    with object_splits_only():      # with spatial splits the builder turns this scene into a balanced tree of depth 8
        ctx.scene_create(sc)
    osc = oracle.scene(sc)
    depth = ctx.scene_info().bvh_max_depth
    assert depth >= 24, depth         # 2*depth+2 stack entries + 16 deferred: beyond what the kernel keeps in LDS
    ref = osc.trace_rays(rays, brute=True)
    assert (ref["prim"] != api.PT_MISS).mean() > 0.5
    try:
        for lds, defer in ((8, 4), (128, 16), (1, 0), (2, 1), (64, 0)):
            ctx.set_tuning(api.PT_TUNE_EXTEND_LDS_STACK, lds)
            ctx.set_tuning(api.PT_TUNE_EXTEND_DEFER, defer)
            got = ctx.trace_rays(rays)
            assert np.array_equal(got["prim"], ref["prim"]), (lds, defer)
            assert np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32)), (lds, defer)
    finally:
        ctx.set_tuning(api.PT_TUNE_EXTEND_LDS_STACK, 6)
        ctx.set_tuning(api.PT_TUNE_EXTEND_DEFER, 4)


def test_coincident_geometry_resolves_ties_to_the_lower_triangle_id(ctx, oracle):
    """Three instances of the same mesh at the same place: every hit is an exact t tie between three triangles, and
    closest-hit must report the lowest global id whatever order the traversal meets them in."""
    base = scenes.random_triangle_soup(8, 2000)
    sc = SceneData(base.meshes, [(0, 0, None), (0, 0, None), (0, 0, None)], base.materials)
    ctx.scene_create(sc)
    osc = oracle.scene(sc)
    rays = random_rays(5, 30000, [-10] * 3, [10] * 3)
    got, ref = ctx.trace_rays(rays), osc.trace_rays(rays, brute=True)
    hit = ref["prim"] != api.PT_MISS
    assert hit.mean() > 0.05
    assert np.array_equal(got["prim"], ref["prim"])
    assert (ref["prim"][hit] < 2000).all()                     # always the first instance's triangle
    assert np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32))


@pytest.mark.parametrize("w,h,spp", [(1, 1, 3), (67, 3, 5), (3, 129, 2), (65, 65, 1), (2048, 1, 2), (2049, 1, 2), (1, 1100000, 1),
                                     (1, 1, 3000)])
def test_odd_frame_sizes_match_oracle(ctx, cornell, w, h, spp):
    """Frame sizes that are not multiples of the wave or block size, down to a single pixel; the last four sit on both
    sides of the switch between the float-reciprocal and the integer pixel decode (device_math.h FastDiv)."""
    sc, osc = cornell
    ctx.scene_create(sc)
    c = cornell_consts(w, h, sky=False, fallback=100.0)
    ctx.accum_reset(w, h)
    ctx.stats_reset()
    ctx.render(c, w, h, 2, spp)
    img, n = ctx.accum_read_sum()
    st = ctx.stats()
    acc, ost, _ = osc.render(c, w, h, 2, spp)
    assert n == spp and st.samples == w * h * spp
    if w * h * spp < 100000:
        assert (st.rays, st.shaded_hits, st.sky_misses) == (ost["rays"], ost["shaded_hits"], ost["sky_misses"])
        assert rel_mse(img, acc) < 1e-8
    else:   # a million paths: a handful graze an edge where the last bit of a sampled direction decides (libm vs GPU sin/cos/pow)
        for a, b in ((st.rays, ost["rays"]), (st.shaded_hits, ost["shaded_hits"]), (st.sky_misses, ost["sky_misses"])):
            assert abs(a - b) <= 5e-5 * b, (a, b)
        assert rel_mse(img, acc) < 1e-4
    hits = ctx.primary_hits(c, w, h, 2)
    assert np.array_equal(hits["prim"], osc.primary_hits(c, w, h, 2)["prim"])


def test_non_finite_rays_miss_quickly(ctx, oracle):
    """NaN/Inf rays (the shading can produce them: Q7 0/0, normalize(0)) are misses and must not walk the whole BVH."""
    import time
    sc = scenes.atrium(1)
    ctx.scene_create(sc)
    osc = oracle.scene(sc)
    rays = random_rays(9, 4096, (-20, 0, -10), (20, 16, 10))
    rays["origin"][::7, 1] = np.nan
    rays["dir"][3::11, 0] = np.inf
    rays["dir"][5::13] = np.nan
    t0 = time.time()
    got = ctx.trace_rays(rays)
    assert time.time() - t0 < 2.0
    want = osc.trace_rays(rays)
    assert np.array_equal(got["prim"], want["prim"]) and np.array_equal(got["t"].view(np.uint32), want["t"].view(np.uint32))
    bad = ~np.isfinite(rays["origin"].sum(1) + rays["dir"].sum(1))
    assert bad.sum() > 500 and (got["prim"][bad] == api.PT_MISS).all()


def test_tmin_tmax_exclusive_on_device(ctx, oracle):
    v = np.zeros(3, dtype=api.VERTEX_DTYPE)
    v["pos"] = [[-1, -1, 5], [1, -1, 5], [0, 1, 5]]
    sc = SceneData([(v, np.arange(3, dtype=np.uint32))], [(0, 0, None)], [((0, 0, 0), (1, 1, 1, 1))])
    ctx.scene_create(sc)
    r = np.zeros(1, dtype=api.RAY_DTYPE)
    r["dir"] = [0, 0, 1]
    r["tmax"] = 10.0
    t0 = np.float32(ctx.trace_rays(r)["t"][0])
    r = np.zeros(4, dtype=api.RAY_DTYPE)
    r["dir"] = [0, 0, 1]
    r["tmin"] = [0.0, t0, 0.0, np.nextafter(t0, np.float32(0))]
    r["tmax"] = [10.0, 10.0, t0, np.nextafter(t0, np.float32(10))]
    assert list(ctx.trace_rays(r)["prim"]) == [0, api.PT_MISS, api.PT_MISS, 0]


def test_instance_transform_and_flatten_order(ctx, oracle):
    v = np.zeros(3, dtype=api.VERTEX_DTYPE)
    v["pos"] = [[-1, -1, 0], [1, -1, 0], [0, 1, 0]]
    v["normal"] = [[0, 0, -1]] * 3
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = [[0.5, 0, 0], [0, 2.0, 0], [0, 0, 1]]
    T[:3, 3] = [3, 0, 7]
    sc = SceneData([(v, np.arange(3, dtype=np.uint32))], [(0, 0, None), (0, 0, T.T.reshape(16))], [((0, 0, 0), (1, 1, 1, 1))])
    ctx.scene_create(sc)
    osc = oracle.scene(sc)
    rays = random_rays(2, 5000, (-1, -2, 0), (4, 2, 7))
    got, want = ctx.trace_rays(rays), osc.trace_rays(rays, brute=True)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert set(np.unique(got["prim"])) >= {0, 1}


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_instanced_scenes_match_brute_force(ctx, oracle, seed):
    """Fuzz: several meshes, many instances with random rotations, non-uniform scales, mirrors and translations
    (column-major 4x4 as in SceneMeshInstance, PathTracer.cpp:285-305); every hit record must equal the oracle's
    brute-force loop bit for bit, and the flags path that transforms normals must render like the oracle's."""
    rng = np.random.default_rng(seed)
    meshes = [scenes.random_triangle_soup(100 * seed + k, int(rng.integers(20, 400)), extent=2.0, size=0.6).meshes[0] for k in range(3)]
    inst = []
    for i in range(int(rng.integers(5, 40))):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        sc3 = np.diag(rng.uniform(0.3, 3.0, 3) * rng.choice([-1.0, 1.0], 3))
        T = np.eye(4)
        T[:3, :3] = q @ sc3
        T[:3, 3] = rng.uniform(-8, 8, 3)
        inst.append((int(rng.integers(0, 3)), int(rng.integers(0, 2)), None if i == 0 else T.T.reshape(16).astype(np.float32)))
    mats = [((0, 0, 0), (0.7, 0.6, 0.5, 1)), ((1, 2, 3), (0.2, 0.9, 0.4, 1))]
    sc = SceneData(meshes, inst, mats)
    ctx.scene_create(sc)
    osc = oracle.scene(sc)
    info = ctx.scene_info()
    rays = random_rays(seed, 20000, list(info.bounds_min), list(info.bounds_max))
    got, want = ctx.trace_rays(rays), osc.trace_rays(rays, brute=True)
    assert (want["prim"] != api.PT_MISS).mean() > 0.02
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    cam = {"eye": tuple(np.asarray(info.bounds_max) * 1.5), "target": (0.0, 0.0, 0.0)}
    c = lookat_consts(64, 48, cam, depth=3, sky=False)
    c.mySkyFallbackEmission[0] = c.mySkyFallbackEmission[1] = c.mySkyFallbackEmission[2] = 1.0
    ctx.accum_reset(64, 48)
    ctx.render(c, 64, 48, 0, 3)
    acc, _, _ = osc.render(c, 64, 48, 0, 3)
    assert rel_mse(ctx.accum_read_sum()[0], acc) < 1e-4


# ---------------------------------------------------------------------------------------------- images
def test_cornell_image_vs_oracle_and_fixture(ctx, cornell):
    sc, osc = cornell
    ctx.scene_create(sc)
    w = h = 128
    c = cornell_consts(w, h)
    ctx.accum_reset(w, h)
    ctx.stats_reset()
    ctx.render(c, w, h, 0, 16)
    img = ctx.accum_read()
    acc, ost, ps = osc.render(c, w, h, 0, 16, per_sample=True)
    assert rel_mse(img, acc / 16.0) < 1e-8          # same RNG streams: far below the 1e-4 bar
    assert rel_mse(img, np.load(os.path.join(G, "cornell_128_16spp.npy"))) < 1e-8
    assert (img[..., 3] == 0).all() and np.isfinite(img).all()
    st = ctx.stats()
    print("cornell rays", st.rays, ost["rays"], "shaded", st.shaded_hits, ost["shaded_hits"])
    assert st.samples == w * h * 16 and abs(int(st.rays) - ost["rays"]) <= 3 and abs(int(st.shaded_hits) - ost["shaded_hits"]) <= 3   # observed: 1 and 0
    # per-sample agreement: render sample 5 alone and compare with the oracle's per-sample radiance
    ctx.accum_reset(w, h)
    ctx.render(c, w, h, 5, 1)
    one, _ = ctx.accum_read_sum()
    d = np.abs(one[..., :3] - ps[5]) / np.maximum(np.abs(ps[5]), 1e-3)
    assert (d > 1e-3).mean() < 1e-3                  # a few samples may branch differently on a 1-ulp pow/sin difference


def test_cornell_c1_full_config(ctx, cornell):
    """BASELINE config C1: Cornell 512x512, 64 spp, depth 4, sky off."""
    sc, osc = cornell
    ctx.scene_create(sc)
    w = h = 512
    c = cornell_consts(w, h)
    ctx.accum_reset(w, h)
    ctx.render(c, w, h, 0, 64)
    img = ctx.accum_read()
    acc, _, _ = osc.render(c, w, h, 0, 64)
    r = rel_mse(img, acc / 64.0)
    assert r < 1e-4, r
    assert r < 1e-7


def test_sky_image_and_fallback_vs_oracle(ctx, cornell, oracle):
    sc, osc = cornell
    ctx.scene_create(sc)
    atm = oracle.earth_atmosphere()
    ctx.sky_set(atm)
    olut = oracle.transmittance_lut(atm)
    w, h = 96, 64
    for sky, fb in ((True, 0.0), (False, 100.0)):     # Sky.cpp sky, and the reference default fallback 100
        c = cornell_consts(w, h, sky=sky, fallback=fb)
        ctx.accum_reset(w, h)
        ctx.render(c, w, h, 0, 8)
        img = ctx.accum_read()
        acc, _, _ = osc.render(c, w, h, 0, 8, lut=olut)
        assert rel_mse(img, acc / 8.0) < 1e-6, (sky, rel_mse(img, acc / 8.0))


def test_atrium_image_vs_oracle(ctx, oracle):
    sc = scenes.atrium(1)
    ctx.scene_create(sc)
    osc = oracle.scene(sc)
    atm = oracle.earth_atmosphere()
    ctx.sky_set(atm)
    olut = oracle.transmittance_lut(atm)
    w, h = 240, 135
    c = lookat_consts(w, h, scenes.ATRIUM_CAM)
    ctx.accum_reset(w, h)
    ctx.render(c, w, h, 0, 4)
    img = ctx.accum_read()
    acc, _, _ = osc.render(c, w, h, 0, 4, lut=olut)
    assert rel_mse(img, acc / 4.0) < 1e-4
    assert img[..., :3].mean() > 0


def test_street_3m_triangles_hits_and_image_match_oracle(ctx, oracle):
    """BASELINE config C4's scene at full triangle count (3,145,728): random incoherent rays bit-identical to the
    oracle's own BVH, and a small depth-8 image within tolerance. Ray counts may differ by a few paths: bounce
    directions go through sin/cos/acos/pow (ocml vs libm, last-ulp differences), and over 8 bounces in a 3 M-triangle
    scene a handful of rays land on the other side of an edge."""
    sc = scenes.street(2)
    ctx.scene_create(sc)
    osc = oracle.scene(sc)
    info = ctx.scene_info()
    assert info.num_triangles == 3145728
    rays = random_rays(41, 200000, list(info.bounds_min), list(info.bounds_max), tmin=0.001, tmax=10000.0)
    got, ref = ctx.trace_rays(rays), osc.trace_rays(rays)
    assert np.array_equal(got["prim"], ref["prim"])
    hit = ref["prim"] != api.PT_MISS
    assert 0.3 < hit.mean() < 1.0
    for f in ("t", "u", "v"):
        assert np.array_equal(got[f][hit].view(np.uint32), ref[f][hit].view(np.uint32)), f
    atm = oracle.earth_atmosphere()
    ctx.sky_set(atm)
    olut = oracle.transmittance_lut(atm)
    w, h = 160, 90
    c = lookat_consts(w, h, scenes.STREET_CAM, depth=8)
    ctx.accum_reset(w, h)
    ctx.stats_reset()
    ctx.render(c, w, h, 0, 2)
    img = ctx.accum_read()
    st = ctx.stats()
    acc, ost, _ = osc.render(c, w, h, 0, 2, lut=olut)
    print("street rays", st.rays, ost["rays"], "sky", st.sky_misses, ost["sky_misses"])
    # observed on MI355X: 7 of 116,757 rays (6e-5): the bound is about twice that, not a blanket 1e-3
    assert abs(st.rays - ost["rays"]) <= 1.5e-4 * ost["rays"]
    assert abs(st.sky_misses - ost["sky_misses"]) <= 1.5e-4 * ost["sky_misses"] + 2
    assert rel_mse(img, acc / 2.0) < 1e-4, rel_mse(img, acc / 2.0)


def test_street_hard_sah_hostile_scene_hits_and_image_match_oracle(ctx, oracle):
    """The SAH-hostile C4 stand-in (scenes.street_hard: every block yawed, 7.4 % long thin triangles — cables, beams,
    banners — running through finely tessellated detail, instanced clutter with random rotations, litter quads) at its
    full 3,145,728 triangles. The builder answers it with spatial splits, so a triangle is referenced from several
    leaves; hit records must not notice: 200 k incoherent rays bit-identical to the oracle's own (object-split) BVH, and
    a depth-8 image within the north-star tolerance."""
    sc = scenes.street_hard(4)
    assert 0.05 <= sc.long_thin_fraction <= 0.10
    ctx.scene_create(sc)
    osc = oracle.scene(sc)
    info = ctx.scene_info()
    assert info.num_triangles == 3145728
    nodes, recs, _ = ctx.debug_read_bvh()
    assert len(recs) > 1.2 * info.num_triangles          # the spatial splits really happened (references > triangles)
    rays = random_rays(43, 200000, list(info.bounds_min), list(info.bounds_max), tmin=0.001, tmax=10000.0)
    got, ref = ctx.trace_rays(rays), osc.trace_rays(rays)
    assert np.array_equal(got["prim"], ref["prim"])
    hit = ref["prim"] != api.PT_MISS
    assert 0.2 < hit.mean() < 1.0
    for f in ("t", "u", "v"):
        assert np.array_equal(got[f][hit].view(np.uint32), ref[f][hit].view(np.uint32)), f
    # this is the scene the library runs its shared triangle phase on (PT_OPT_SHARED_TRIANGLE_PHASE = 2 picks it here):
    # forced on, the same rays give the same bits, the many duplicated references included
    try:
        ctx.set_option(api.PT_OPT_SHARED_TRIANGLE_PHASE, 1)
        ctx.stats_reset()
        got_shared = ctx.trace_rays(rays)
        assert ctx.stats().shared_launches > 0
    finally:
        ctx.set_option(api.PT_OPT_SHARED_TRIANGLE_PHASE, 2)
    assert np.array_equal(got_shared.view(np.uint32), got.view(np.uint32))
    atm = oracle.earth_atmosphere()
    ctx.sky_set(atm)
    olut = oracle.transmittance_lut(atm)
    # 16 spp: in this scene of slivers and litter a last-ulp difference of a bounce direction (ocml vs libm sin/cos/pow) lands
    # on another triangle more often than in the grid scenes (observed: 5 of 220,455 rays at 2 spp), and with few samples per
    # pixel those few paths carry the whole difference (2 spp: relMSE 2.9e-4; it falls as 1/spp)
    w, h, spp = 160, 90, 16
    c = lookat_consts(w, h, scenes.STREET_HARD_CAM, depth=8)
    ctx.accum_reset(w, h)
    ctx.stats_reset()
    ctx.render(c, w, h, 0, spp)
    img = ctx.accum_read()
    st = ctx.stats()
    acc, ost, _ = osc.render(c, w, h, 0, spp, lut=olut)
    print("street_hard rays", st.rays, ost["rays"], "sky", st.sky_misses, ost["sky_misses"], "relMSE", rel_mse(img, acc / spp))
    assert abs(st.rays - ost["rays"]) <= 3e-4 * ost["rays"]
    assert rel_mse(img, acc / spp) < 1e-4, rel_mse(img, acc / spp)


def test_stats_struct_is_versioned_by_size(ctx):
    """PtStats.struct_size: a caller built against an OLDER (shorter) header gets exactly the bytes it knows, a caller that
    forgot to set the size is refused."""
    import ctypes as C
    lib = ctx.lib
    full = api.PtStats()
    full.struct_size = C.sizeof(api.PtStats)
    assert lib.pt_stats_get(ctx.h, C.byref(full)) == api.PT_OK and full.struct_size == C.sizeof(api.PtStats)
    buf = (C.c_ubyte * C.sizeof(api.PtStats))(*([0xAB] * C.sizeof(api.PtStats)))
    short = api.PtStats.from_buffer(buf)
    short.struct_size = 72                                   # an older header that ended after extend_launches
    assert lib.pt_stats_get(ctx.h, C.byref(short)) == api.PT_OK and short.struct_size == 72
    assert bytes(buf[72:]) == b"\xab" * (C.sizeof(api.PtStats) - 72)      # nothing written past the caller's size
    assert short.rays == full.rays
    zero = api.PtStats()
    assert lib.pt_stats_get(ctx.h, C.byref(zero)) == api.PT_ERR_INVALID_ARG
    assert ctx.lib.pt_set_option(ctx.h, api.PT_TUNE_EXTEND_DEFER, 4) == api.PT_ERR_INVALID_ARG   # tuning ids are not options


# ---------------------------------------------------------------------------------------------- BASELINE C4 / C5 as stated
def test_street_c4_with_ggx_material_extension_matches_oracle(ctx, oracle):
    """BASELINE config C4 as it is worded ("~3M tris, 8 bounces, textured GGX materials"): street(2) with the material
    extension (GGX lobes + procedural albedo patterns) on every material, depth 8, against the oracle's restatement."""
    sc = scenes.with_material_ext(scenes.street(2))
    ctx.scene_create(sc)
    osc = oracle.scene(sc)
    assert ctx.scene_info().num_triangles == 3145728
    atm = oracle.earth_atmosphere()
    ctx.sky_set(atm)
    olut = oracle.transmittance_lut(atm)
    w, h, spp = 160, 90, 2
    c = lookat_consts(w, h, scenes.STREET_CAM, depth=8)
    ctx.accum_reset(w, h)
    ctx.stats_reset()
    ctx.render(c, w, h, 0, spp)
    img = ctx.accum_read()
    st = ctx.stats()
    acc, ost, _ = osc.render(c, w, h, 0, spp, lut=olut)
    print("street+ext rays", st.rays, ost["rays"], "relMSE", rel_mse(img, acc / spp))
    assert abs(st.rays - ost["rays"]) <= 6e-5 * ost["rays"]      # observed: 3 of 122,167
    assert rel_mse(img, acc / spp) < 1e-4, rel_mse(img, acc / spp)
    assert st.rays / st.samples > 2.0                   # the depth-8 paths really bounce


def test_courtyard_c5_10m_triangles_depth16_russian_roulette(ctx, oracle):
    """BASELINE config C5: courtyard(3) at its full 10,485,760 triangles (585 MB of BVH: beyond the 256 MB Infinity
    Cache), depth 16 (<= 17 rays per sample, PathTracing.hlsl:68), Russian roulette on. 200 k incoherent rays bit-identical
    to the oracle's own BVH; a small image within the north-star tolerance with ray counts within 5e-5."""
    sc = scenes.courtyard(3)
    ctx.scene_create(sc)
    info = ctx.scene_info()
    assert info.num_triangles == 10485760 and info.bvh_bytes > 256 * 2 ** 20
    osc = oracle.scene(sc)
    rays = random_rays(51, 200000, list(info.bounds_min), list(info.bounds_max), tmin=0.001, tmax=10000.0)
    got, ref = ctx.trace_rays(rays), osc.trace_rays(rays)
    assert np.array_equal(got["prim"], ref["prim"])
    hit = ref["prim"] != api.PT_MISS
    assert 0.3 < hit.mean() < 1.0
    for f in ("t", "u", "v"):
        assert np.array_equal(got[f][hit].view(np.uint32), ref[f][hit].view(np.uint32)), f
    atm = oracle.earth_atmosphere()
    ctx.sky_set(atm)
    olut = oracle.transmittance_lut(atm)
    w, h, spp = 128, 72, 4
    c = lookat_consts(w, h, scenes.COURTYARD_CAM, depth=16)
    try:
        ctx.set_option(api.PT_OPT_RUSSIAN_ROULETTE, 1)
        oracle.set_russian_roulette(1)
        ctx.accum_reset(w, h)
        ctx.stats_reset()
        ctx.render(c, w, h, 0, spp)
        img = ctx.accum_read()
        st = ctx.stats()
        acc, ost, _ = osc.render(c, w, h, 0, spp, lut=olut)
    finally:
        ctx.set_option(api.PT_OPT_RUSSIAN_ROULETTE, 0)
        oracle.set_russian_roulette(0)
    print("courtyard rays", st.rays, ost["rays"], "relMSE", rel_mse(img, acc / spp))
    assert abs(st.rays - ost["rays"]) <= 5e-5 * ost["rays"]      # observed: identical (114,518)
    assert abs(st.sky_misses - ost["sky_misses"]) <= 5e-5 * ost["sky_misses"] + 2
    assert rel_mse(img, acc / spp) < 1e-4, rel_mse(img, acc / spp)
    assert max(b for b in range(16) if st.rays_bounce[b] > 0) >= 8       # long paths exist


# ---------------------------------------------------------------------------------------------- roofline inputs
@pytest.mark.parametrize("pairs", [False, True], ids=["48-byte-triangles", "64-byte-pairs"])
def test_count_kernel_matches_host_walker_on_the_same_bvh8(ctx, oracle, pairs):
    """Both leaf-record formats of the builder (single triangles; edge-sharing pairs, its choice for scenes beyond the
    Infinity Cache). bench.py's roofline is built on nodes / triangles per ray reported by the COUNT instantiation of k_extend. A
    second implementation (tests/walker/bvh8_walker.cpp: scalar host walker over the very BVH8 the device holds, read
    back with pt_debug_read_bvh, canonical schedule) must report the same totals and the same hit records."""
    import ctypes as C
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "tests", "walker", "libbvh8_walker.so")
    assert os.path.exists(so), "build tests/walker first (__graft_entry__.build())"
    wl = C.CDLL(so)
    P = C.c_void_p
    wl.bvh8_walk.restype = C.c_int
    wl.bvh8_walk.argtypes = [P, C.c_uint64, P, C.c_uint64, C.c_uint32, C.c_float, P, C.c_uint64, P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    sc = scenes.atrium(1)
    ctx.scene_create(sc, api.PT_SCENE_PAIR_TRIANGLES if pairs else api.PT_SCENE_NO_PAIRS)
    info = ctx.scene_info()
    nodes, tris, mag = ctx.debug_read_bvh()
    assert len(nodes) == info.num_bvh_nodes and tris.shape[1] == (64 if pairs else 48)
    if pairs:
        assert info.num_triangles // 2 <= len(tris) < 0.6 * info.num_triangles   # made of quads: nearly every triangle finds its pair
    else:   # spatial splits may reference a triangle from several leaves
        assert info.num_triangles <= len(tris) <= 2 * info.num_triangles
    rays = random_rays(77, 100000, list(info.bounds_min), list(info.bounds_max), tmin=0.001, tmax=10000.0)
    whits = np.empty(len(rays), dtype=api.HIT_DTYPE)
    cn, ct = C.c_uint64(), C.c_uint64()
    assert wl.bvh8_walk(api.ptr(nodes), len(nodes), api.ptr(tris), len(tris), tris.shape[1], mag, api.ptr(rays), len(rays), api.ptr(whits),
                        C.byref(cn), C.byref(ct)) == 0
    try:
        ctx.set_tuning(api.PT_TUNE_EXTEND_DEFER, 0)
        ctx.set_tuning(api.PT_TUNE_EXTEND_POSTPONE_PCT, 0)        # canonical schedule: what bench.py's count_pass(0, 0) runs
        ctx.set_option(api.PT_OPT_COUNT_TRAVERSAL, 1)
        ctx.stats_reset()
        got = ctx.trace_rays(rays)
        st = ctx.stats()
    finally:
        ctx.set_option(api.PT_OPT_COUNT_TRAVERSAL, 0)
        ctx.set_tuning(api.PT_TUNE_EXTEND_DEFER, 4)
        ctx.set_tuning(api.PT_TUNE_EXTEND_POSTPONE_PCT, 40)
    assert np.array_equal(got.view(np.uint32), whits.view(np.uint32))
    assert (st.nodes_visited, st.tris_tested) == (cn.value, ct.value)
    assert 5 < cn.value / len(rays) < 60 and 1 < ct.value / len(rays) < 40
    assert np.array_equal(whits["prim"], oracle.scene(sc).trace_rays(rays)["prim"])      # and the oracle's own BVH agrees


# ---------------------------------------------------------------------------------------------- f3 / f2 against the oracle
def test_sky_view_lut_against_the_oracle_march(ctx, oracle):
    """sky mode 1 "within relMSE 1e-4 of the marched ORACLE" (SampleSky.hlsl:7-47 restated in
    oracle/sky.h), not merely of the library's own march: 50 k directions through PT_FN_SKY_VIEW_LUT vs
    orc_eval(PT_FN_SKY), and a mode-1 atrium image vs orc_render."""
    atm = oracle.earth_atmosphere()
    ctx.sky_set(atm)
    olut = oracle.transmittance_lut(atm)
    for sun_dir in ((0.3, 0.8, 0.52), (-0.7, 0.12, 0.2)):
        c = lookat_consts(64, 64, scenes.ATRIUM_CAM, sun_dir=sun_dir)
        rng = np.random.default_rng(12)
        n = 50000
        d = rng.normal(size=(n, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        d[:, 1] = np.abs(d[:, 1])
        rows = np.concatenate([np.tile(np.asarray(c.myCameraPos, dtype=np.float32), (n, 1)), d], axis=1).astype(np.float32)
        lut = ctx.debug_eval(api.PT_FN_SKY_VIEW_LUT, rows, c).astype(np.float64)
        want = oracle.eval(api.PT_FN_SKY, rows, consts=c, lut=olut).astype(np.float64)
        r = ((lut - want) ** 2).mean() / (want ** 2).mean()
        assert r < 1e-4, (sun_dir, r)
    sc = scenes.atrium(1)
    ctx.scene_create(sc)
    osc = oracle.scene(sc)
    w, h, spp = 240, 135, 4
    c = lookat_consts(w, h, scenes.ATRIUM_CAM)
    try:
        ctx.set_option(api.PT_OPT_SKY_MODE, 1)
        ctx.accum_reset(w, h)
        ctx.render(c, w, h, 0, spp)
        img = ctx.accum_read()
    finally:
        ctx.set_option(api.PT_OPT_SKY_MODE, 0)
    acc, _, _ = osc.render(c, w, h, 0, spp, lut=olut)
    assert rel_mse(img, acc / spp) < 1e-4, rel_mse(img, acc / spp)


def test_tonemapped_output_matches_oracle_image(ctx, cornell, oracle, tmp_path):
    """the output stage (tonemap_composit.hlsl:37-44, Reinhard c/(c+1)) applied by the host mirror to a
    GPU-accumulated target, against c/(c+1) of the oracle's image; then through the PFM and PNG writers."""
    from pathtracer_amd import PathTracer, image_io
    sc, osc = cornell
    pt = PathTracer(0, 192, 128)
    try:
        pt.LoadScene(sc, scenes.CORNELL_CAM_POS)
        pt.mySampleSky = False
        pt.mySkyFallbackIntensity = 100.0                # reference default (PathTracer.h:123)
        pt.RenderRT(8)
        hdr = pt.ReadOutput()
        ldr = pt.Tonemap(hdr)
        acc, _, _ = osc.render(pt.BuildRtConsts(), pt.myOutputWidth, pt.myOutputHeight, 0, 8)
        ref = acc[..., :3] / 8.0
        want = ref / (ref + 1.0)
        assert ldr.shape == want.shape and (ldr >= 0).all() and (ldr < 1).all()
        assert rel_mse(ldr, want) < 1e-8, rel_mse(ldr, want)
        assert np.array_equal(ldr, image_io.tonemap_reinhard(hdr))
        image_io.write_pfm(tmp_path / "o.pfm", ldr)
        assert np.array_equal(image_io.read_pfm(tmp_path / "o.pfm"), ldr)
        image_io.write_png(tmp_path / "o.png", ldr)
        assert open(tmp_path / "o.png", "rb").read(8) == b"\x89PNG\r\n\x1a\n"
    finally:
        pt.ctx.close()


def test_frame_lanes_overlap_small_frames_with_identical_bits(ctx, oracle):
    """PT_OPT_FRAME_LANES: the reference's interactive pattern (one sample per RenderRT call on a small target,
    PathTracer.cpp:573-595) queued round-robin on 6 (or 3) internal streams must leave the same bits in the target as every call
    serial on the ctx stream — across a camera change, a sky change (new LUT the lanes must wait for), a restart of the
    accumulation and a scene change (deeper stack: the lanes' spill regions are re-made)."""
    atm = oracle.earth_atmosphere()
    w, h = 320, 180

    def run(lanes):
        ctx.set_option(api.PT_OPT_FRAME_LANES, lanes)
        imgs = []
        for sc, cam in ((scenes.atrium(1), scenes.ATRIUM_CAM), (scenes.random_triangle_soup(5, 20000), None)):
            ctx.scene_create(sc)
            ctx.sky_set(atm)
            c = lookat_consts(w, h, cam) if cam else lookat_consts(w, h, {"eye": (5.0, 5.0, -8.0), "target": (5.0, 5.0, 5.0)})
            ctx.accum_reset(w, h)
            for i in range(9):
                ctx.render(c, w, h, i, 1)                   # 9 calls on 4 lanes: every lane is re-used
            c2 = lookat_consts(w, h, cam, depth=6) if cam else c
            for i in range(9, 14):
                ctx.render(c2, w, h, i, 1 + (i & 1))         # mixed 1- and 2-sample calls
            atm2 = oracle.earth_atmosphere()
            atm2.MiePhaseG = 0.6
            ctx.sky_set(atm2)                                # lanes must not read the LUT before this kernel has run
            for i in range(16, 20):
                ctx.render(c2, w, h, i, 1)
            imgs.append(ctx.accum_read_sum()[0].copy())
            ctx.accum_reset(w, h)                            # restart: the memset is ordered against the lanes' resolves
            for i in range(3):
                ctx.render(c, w, h, i, 1)
            imgs.append(ctx.accum_read_sum()[0].copy())
            q = ctx.debug_read_queue(1)                      # the last call's queue lives in a lane's buffers
            assert len(q) > 0
        return imgs
    try:
        a = run(6)
        c3 = run(3)
        b = run(1)
    finally:
        ctx.set_option(api.PT_OPT_FRAME_LANES, 6)
    for x, y, z in zip(a, b, c3):
        assert np.isfinite(x).all() and x[..., :3].sum() > 0
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)) and np.array_equal(z.view(np.uint32), y.view(np.uint32))
    with pytest.raises(api.PtError):
        ctx.set_option(api.PT_OPT_FRAME_LANES, 9)


def test_rccl_backend_runs_the_collectives_the_multi_gpu_path_uses():
    """The box has one GPU, so RCCL never sees a second rank here; what CAN be checked is that every collective the N > 1
    path issues (bench.py, pathtracer_amd/distributed.py) is accepted by the nccl (= RCCL) backend on this device with the
    dtypes and shapes used: eager init with device_id, reduce of a W*H*4 float target, gather of block sums, int64 SUM and
    float64 MAX / SUM all-reduces, barrier. One rank, in a process of its own (a process group must not leak into pytest)."""
    import subprocess
    import sys
    code = r"""
import os, socket, torch, torch.distributed as dist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
t = torch.full((1920 * 1080 * 4,), 0.5, dtype=torch.float32, device=dev)
dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
blocks = torch.arange(8 * 1024, dtype=torch.float32, device=dev).reshape(8, 1024)
parts = [torch.empty_like(blocks)]
dist.gather(blocks, gather_list=parts, dst=0)
cnt = torch.tensor([4096], dtype=torch.int64, device=dev)
dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
ms = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(ms, op=dist.ReduceOp.MAX)
dist.all_reduce(ms, op=dist.ReduceOp.SUM)
dist.barrier()
torch.cuda.synchronize()
assert float(t[0]) == 0.5 and torch.equal(parts[0], blocks) and int(cnt) == 4096 and float(ms) == 1.25
dist.destroy_process_group()
print("rccl ok")
"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stderr[-2000:]


# ---------------------------------------------------------------------------------------------- bench launcher
def test_bench_runs_an_obj_file_and_the_optional_sponza_leg(tmp_path):
    """'if $PT_SCENE_DIR/sponza.obj exists on the box, also run it'. bench.py --workload obj renders any OBJ/MTL
    file through the host's reader with a camera placed from the bounding box, and the default line gains a C2_sponza_obj leg
    when the file exists (rehearsed with a small stand-in file under that name; the real asset is not available offline)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sc = scenes.random_triangle_soup(5, 4000, extent=12.0, size=1.5)
    scenes.scene_to_obj(sc, str(tmp_path / "sponza.obj"))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PT_SCENE_DIR")}
    small = ["--steps", "1", "--warmup", "1", "--width", "320", "--height", "180", "--spp-per-step", "2", "--no-cpu-baseline"]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "obj", "--scene-obj", str(tmp_path / "sponza.obj"),
                        "--no-configs"] + small, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["triangles"] == 4000 and "sponza.obj" in d["config"]["workload"] and d["value"] > 0
    assert d["roofline"]["nodes_per_ray"] > 1.0 and d["config"]["rays_per_sample"] > 1.05     # the camera sees geometry
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "obj", "--no-configs"] + small,
                       capture_output=True, text=True, timeout=120, env=env, cwd=root)
    assert r.returncode == 2 and "sponza.obj" in r.stderr                                       # no file named: an error, not a default scene
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "cornell", "--configs", "C2_sponza_obj"] + small,
                       capture_output=True, text=True, timeout=300, env=dict(env, PT_SCENE_DIR=str(tmp_path)), cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    leg = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])["configs"]["C2_sponza_obj"]
    assert "error" not in leg and leg["triangles"] == 4000 and leg["mrays_per_s"] > 0 and (leg["width"], leg["height"]) == (1920, 1080)


def test_bench_gpus_2_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` (the driver's form) must run TWO ranks: rehearsed on this one-GPU box with gloo and
    both ranks on cuda:0. A --gpus value that disagrees with WORLD_SIZE is an error, not a silent 1-GPU run."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    args = ["--backend", "gloo", "--all-ranks-on-gpu0", "--steps", "1", "--warmup", "1", "--width", "320", "--height", "180",
            "--spp-per-step", "2", "--no-cpu-baseline", "--no-roofline"]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + args, capture_output=True, text=True, timeout=170, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    # N > 1 is strong scaling by default (BASELINE C3: a fixed sample count split over the GPUs): 1 step x 2 spp = 2 spp over 2 ranks
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "strong"
    assert out["config"]["job_spp"] == 2 and out["config"]["spp_this_rank"] == 1 and out["samples_reduced"] == 2
    assert len(out["per_rank_ms"]) == 2 and out["config"]["parallelism"].startswith("sample-partition x2")
    weak = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--weak"] + args, capture_output=True, text=True, timeout=170, env=env)
    assert weak.returncode == 0, weak.stderr[-2000:]
    wout = json.loads([ln for ln in weak.stdout.splitlines() if ln.startswith("{")][0])
    assert wout["scaling"] == "weak" and wout["config"]["job_spp"] == 4 and wout["samples_reduced"] == 4
    # --bit-identical: the 8 fixed virtual partitions give the same image bits for 1 and 2 ranks
    shas = []
    for n in (1, 2):
        r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--bit-identical", "--total-spp", "16"] + args,
                            capture_output=True, text=True, timeout=170, env=env)
        assert r2.returncode == 0, r2.stderr[-2000:]
        shas.append(json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][0])["image_sha256"])
    assert shas[0] and shas[0] == shas[1]
    env["WORLD_SIZE"] = "1"
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + args, capture_output=True, text=True, timeout=170, env=env)
    assert bad.returncode == 2 and "WORLD_SIZE" in bad.stderr


# ---------------------------------------------------------------------------------------------- schedule options
@pytest.mark.parametrize("fetch_min,postpone,blocks,lds_stack,defer,shared", [
    (16, 40, 0, 6, 4, 0),      # the shipped defaults (csrc/kernels.h LaunchCfg), set explicitly
    (16, 50, 0, 8, 4, 0),      # round-1 schedule: 50 % postpone threshold, 8-entry LDS stack
    (1, 0, 1, 1, 0, 0),        # stack postponing off, one LDS stack entry: everything deeper goes through the global spill
    (64, 100, 2, 2, 1, 0),     # refill only when the whole wave is idle, always postpone, one parked group
    (32, 25, 3, 3, 16, 0),     # large deferred queue
    (8, 75, 0, 128, 0, 0),     # whole stack in LDS (more than 64 KB of dynamic LDS), stack postponing
    (16, 25, 0, 6, 3, 1),      # shared triangle phase, its shipped schedule
    (64, 100, 2, 2, 1, 1),     # shared triangle phase: one parked group (every second group forces a pass), two LDS stack entries
    (4, 0, 0, 128, 16, 1),     # shared triangle phase at once, large queue, more than 64 KB of dynamic LDS
])
def test_extend_schedule_options_never_change_hits(ctx, oracle, fetch_min, postpone, blocks, lds_stack, defer, shared):
    """The PT_TUNE_EXTEND_* knobs (include/pt_api_tuning.h) only reorder work inside the traversal kernel: closest hits stay bit-identical to
    the oracle and occlusion answers stay equal, including through the global stack-spill path."""
    sc = scenes.atrium(1)
    ctx.scene_create(sc)
    osc = oracle.scene(sc)
    info = ctx.scene_info()
    rays = random_rays(31, 150000, list(info.bounds_min), list(info.bounds_max), tmin=0.001, tmax=10000.0)
    ref = osc.trace_rays(rays)
    if shared:
        opts = ((api.PT_TUNE_EXTEND_FETCH_MIN, fetch_min, 16), (api.PT_TUNE_SHARED_POSTPONE_PCT, postpone, 25), (api.PT_TUNE_SHARED_CONTINUE_PCT, postpone, 50),
                (api.PT_TUNE_EXTEND_BLOCKS_PER_CU, blocks, 0), (api.PT_TUNE_EXTEND_LDS_STACK, lds_stack, 6), (api.PT_TUNE_SHARED_DEFER, defer, 3))
    else:
        opts = ((api.PT_TUNE_EXTEND_FETCH_MIN, fetch_min, 16), (api.PT_TUNE_EXTEND_POSTPONE_PCT, postpone, 40),
                (api.PT_TUNE_EXTEND_BLOCKS_PER_CU, blocks, 0), (api.PT_TUNE_EXTEND_LDS_STACK, lds_stack, 6),
                (api.PT_TUNE_EXTEND_DEFER, defer, 4))
    try:
        ctx.set_option(api.PT_OPT_SHARED_TRIANGLE_PHASE, shared)
        ctx.stats_reset()
        for o, v, _ in opts:
            ctx.set_tuning(o, v)
        got = ctx.trace_rays(rays)
        occ = ctx.occluded(rays)
        assert (ctx.stats().shared_launches > 0) == bool(shared)
    finally:
        ctx.set_option(api.PT_OPT_SHARED_TRIANGLE_PHASE, 2)
        for o, _, d in opts:
            ctx.set_tuning(o, d)
    assert np.array_equal(got["prim"], ref["prim"])
    hit = ref["prim"] != api.PT_MISS
    for f in ("t", "u", "v"):
        assert np.array_equal(got[f][hit].view(np.uint32), ref[f][hit].view(np.uint32)), f
    assert np.array_equal(occ.astype(bool), hit)


def test_queue_length_never_changes_hits(ctx, oracle):
    """The traversal kernel deals a short queue (fewer than 64 rays per resident wave) out to its waves statically and
    claims a long one from the cursor in chunks (extend_kernel.inc): every prefix of one ray set, on both sides of that
    switch and at the sizes where a wave's share changes, gets the bits of the long launch, which match the oracle."""
    sc = scenes.atrium(1)
    ctx.scene_create(sc)
    info = ctx.scene_info()
    rays = random_rays(77, 700000, list(info.bounds_min), list(info.bounds_max), tmin=0.001, tmax=10000.0)
    full = ctx.trace_rays(rays)
    ref = oracle.scene(sc).trace_rays(rays[:30000])
    assert np.array_equal(full["prim"][:30000], ref["prim"])
    hit = ref["prim"] != api.PT_MISS
    for f in ("t", "u", "v"):
        assert np.array_equal(full[f][:30000][hit].view(np.uint32), ref[f][hit].view(np.uint32)), f
    sizes = (1, 3, 4, 5, 63, 64, 65, 6143, 6144, 6145, 24577, 200001, 393215, 393216, 393217, 450000)
    try:
        for blocks in (0, 1):            # 0: the resident grid (6 blocks per CU); 1: a sixth of it, so the switch sits at a sixth of the length
            ctx.set_tuning(api.PT_TUNE_EXTEND_BLOCKS_PER_CU, blocks)
            for n in sizes if blocks == 0 else (5, 1025, 65535, 65536, 65537, 200001):
                got = ctx.trace_rays(rays[:n])
                assert np.array_equal(got.view(np.uint32), full[:n].view(np.uint32)), (blocks, n)
            occ = ctx.occluded(rays[:50001])
            assert np.array_equal(occ.astype(bool), full["prim"][:50001] != api.PT_MISS)
    finally:
        ctx.set_tuning(api.PT_TUNE_EXTEND_BLOCKS_PER_CU, 0)


def test_fast_build_flag_gives_the_same_hits_with_a_plainer_tree(ctx, oracle):
    """PT_SCENE_FAST_BUILD skips the spatial splits and the optimisation pass over the binary tree: no reference is duplicated,
    the build is quicker, the hit records are the same bits (they never depend on the acceleration structure)."""
    sc = scenes.random_triangle_soup(11, 60000)
    rays = None
    got = {}
    for flags in (api.PT_SCENE_NO_PAIRS, api.PT_SCENE_NO_PAIRS | api.PT_SCENE_FAST_BUILD):
        ctx.scene_create(sc, flags)
        info = ctx.scene_info()
        if rays is None:
            rays = random_rays(5, 100000, list(info.bounds_min), list(info.bounds_max), tmin=0.001, tmax=10000.0)
        _, tris, _ = ctx.debug_read_bvh()
        got[flags] = (ctx.trace_rays(rays), len(tris), info.build_ms)
    full, fast = got[api.PT_SCENE_NO_PAIRS], got[api.PT_SCENE_NO_PAIRS | api.PT_SCENE_FAST_BUILD]
    assert np.array_equal(full[0].view(np.uint32), fast[0].view(np.uint32))
    ref = oracle.scene(sc).trace_rays(rays)
    assert np.array_equal(fast[0]["prim"], ref["prim"])
    assert fast[1] == 60000 and full[1] >= 60000     # one record per triangle without spatial splits


@pytest.mark.parametrize("pairs", [False, True], ids=["48-byte-triangles", "64-byte-pairs"])
def test_shared_triangle_phase_gives_the_same_hit_records(ctx, oracle, pairs):
    """PT_OPT_SHARED_TRIANGLE_PHASE = 1: the wave pools the pending leaf records of its 64 rays and tests them 64 at a time; a
    hit travels back to its ray through an LDS minimum on (t, triangle id). Closest hits (ties, t == tmax, duplicated
    references of the spatial splits included) must be bit-identical to the oracle's and to the per-lane phase, the counted
    counting instantiation must give the same records, and occlusion answers must agree."""
    sc = scenes.atrium(1)
    ctx.scene_create(sc, api.PT_SCENE_PAIR_TRIANGLES if pairs else api.PT_SCENE_NO_PAIRS)
    osc = oracle.scene(sc)
    info = ctx.scene_info()
    rays = random_rays(41, 200000, list(info.bounds_min), list(info.bounds_max), tmin=0.001, tmax=10000.0)
    # a block of rays whose tmax IS the distance of their closest hit: t == tmax is not a hit (exclusive interval), and the
    # shared phase has to reproduce that through its integer key
    ref0 = osc.trace_rays(rays[:20000])
    h = ref0["prim"] != api.PT_MISS
    rays["tmax"][:20000][h] = ref0["t"][h]
    ref = osc.trace_rays(rays)
    try:
        ctx.set_option(api.PT_OPT_SHARED_TRIANGLE_PHASE, 0)
        lane = ctx.trace_rays(rays)
        ctx.set_option(api.PT_OPT_SHARED_TRIANGLE_PHASE, 1)
        ctx.stats_reset()
        got = ctx.trace_rays(rays)
        occ = ctx.occluded(rays)
        st = ctx.stats()
        assert st.shared_launches >= 2
        ctx.set_option(api.PT_OPT_COUNT_TRAVERSAL, 1)
        ctx.stats_reset()
        got_counted = ctx.trace_rays(rays)
        stc = ctx.stats()
    finally:
        ctx.set_option(api.PT_OPT_COUNT_TRAVERSAL, 0)
        ctx.set_option(api.PT_OPT_SHARED_TRIANGLE_PHASE, 2)
    assert np.array_equal(got.view(np.uint32), lane.view(np.uint32))
    assert np.array_equal(got_counted.view(np.uint32), lane.view(np.uint32))
    assert np.array_equal(got["prim"], ref["prim"])
    hit = ref["prim"] != api.PT_MISS
    assert 0.3 < hit.mean() < 0.99
    for f in ("t", "u", "v"):
        assert np.array_equal(got[f][hit].view(np.uint32), ref[f][hit].view(np.uint32)), f
    assert np.array_equal(occ.astype(bool), hit)
    assert stc.tris_tested > 0 and stc.tri_wave_steps > 0


def test_shared_triangle_phase_auto_choice_times_both_and_keeps_the_image(ctx, oracle):
    """PT_OPT_SHARED_TRIANGLE_PHASE = 2 (default): of a scene's first large batches the second to fifth run the per-lane phase,
    the shared one, and both again, each between two events; a later call reads the times and fixes the choice. The
    accumulated image is the same bits as with the option off, whatever it chose."""
    atm = oracle.earth_atmosphere()
    ctx.sky_set(atm)
    sc = scenes.atrium(1)
    w, h, spp = 1024, 576, 4          # 2.4 M paths per batch: above the 2^21 the timing asks for
    c = lookat_consts(w, h, scenes.ATRIUM_CAM, depth=3)
    imgs = []
    for opt in (0, 2):
        ctx.scene_create(sc)
        ctx.set_option(api.PT_OPT_SHARED_TRIANGLE_PHASE, opt)
        ctx.accum_reset(w, h)
        ctx.stats_reset()
        for i in range(7):            # auto: batch 0 untimed, batches 1-4 per-lane / shared / per-lane / shared, then the choice
            ctx.render(c, w, h, i * spp, spp)
        st = ctx.stats()
        imgs.append(ctx.accum_read_sum()[0].copy())
        if opt == 0:
            assert st.shared_choice == -1 and st.shared_launches == 0
        else:
            assert st.shared_choice in (0, 1)
            assert st.shared_cal_ms[0] > 0.2 and st.shared_cal_ms[1] > 0.2
            # each of the two timed shared batches ran 3 secondary launches (4 when k_extend also traces the primaries); with
            # the shared form chosen, the two batches after them did as well
            assert st.shared_launches in ((12, 16) if st.shared_choice == 1 else (6, 8))
    # forced on, counted: the same bits again, and the phase fills its rounds (per-lane phase on this scene: 0.49 of the
    # lanes test a record per wave-step; the review's bar for the shared form was 0.70 with the node phase above 0.75)
    try:
        ctx.scene_create(sc)
        ctx.set_option(api.PT_OPT_SHARED_TRIANGLE_PHASE, 1)
        ctx.set_option(api.PT_OPT_COUNT_TRAVERSAL, 1)
        ctx.accum_reset(w, h)
        ctx.stats_reset()
        for i in range(7):
            ctx.render(c, w, h, i * spp, spp)
        st = ctx.stats()
        img1 = ctx.accum_read_sum()[0].copy()
    finally:
        ctx.set_option(api.PT_OPT_COUNT_TRAVERSAL, 0)
        ctx.set_option(api.PT_OPT_SHARED_TRIANGLE_PHASE, 2)
    assert np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32))
    assert np.array_equal(imgs[0].view(np.uint32), img1.view(np.uint32))
    tri_util, node_util = st.tris_tested / (64.0 * st.tri_wave_steps), st.nodes_visited / (64.0 * st.node_wave_steps)
    # (this small batch includes its primaries and the tails of a 2.4 M-path queue: the node phase reads 0.75 here, 0.82 on
    # the 133 M-path batches of the bench)
    assert tri_util > 0.70 and node_util > 0.70, (tri_util, node_util)


# ---------------------------------------------------------------------------------------------- AO pipeline (Ao.hlsl)
def test_occlusion_queries_match_oracle_closest_hit(ctx, oracle):
    """pt_occluded stops at the first accepted intersection; the answer must equal 'the closest-hit query hit
    something' for every ray (bit-exact: it is a boolean of the same watertight tests)."""
    for sc, n, tmax in ((scenes.random_triangle_soup(5, 20000), 100000, 3.0), (scenes.atrium(1), 200000, 2.5)):
        ctx.scene_create(sc)
        osc = oracle.scene(sc)
        info = ctx.scene_info()
        rays = random_rays(9, n, list(info.bounds_min), list(info.bounds_max), tmin=0.01, tmax=tmax)
        occ = ctx.occluded(rays)
        ref = osc.trace_rays(rays)["prim"] != api.PT_MISS
        assert 0.02 < ref.mean() < 0.98
        assert np.array_equal(occ.astype(bool), ref)
        full = ctx.trace_rays(rays)["prim"] != api.PT_MISS      # closest-hit kernel path on the same rays
        assert np.array_equal(full, ref)


@pytest.mark.parametrize("scene_name", ["cornell", "atrium"])
def test_ao_pipeline_matches_oracle(ctx, oracle, cornell, scene_name):
    """Ao.hlsl: 1 primary + 16 occlusion rays. AO pixels are k/16 and must be bit-identical; sky pixels follow the
    float tolerance of the sky march."""
    atm = oracle.earth_atmosphere()
    ctx.sky_set(atm)
    olut = oracle.transmittance_lut(atm)
    if scene_name == "cornell":
        sc, osc = cornell
        w, h, spp = 96, 96, 3
        c = cornell_consts(w, h, sky=True)
        c.myAoDistance = 40.0            # the box is ~55 units wide
    else:
        sc = scenes.atrium(1)
        osc = oracle.scene(sc)
        w, h, spp = 160, 90, 2
        c = lookat_consts(w, h, scenes.ATRIUM_CAM)
        c.myAoDistance = 1.0             # reference default (PathTracer.h:112)
    ctx.scene_create(sc)
    ctx.accum_reset(w, h)
    ctx.stats_reset()
    ctx.render_ao(c, w, h, 5, spp)
    img, n = ctx.accum_read_sum()
    st = ctx.stats()
    acc, ost = osc.render_ao(c, w, h, 5, spp, lut=olut)
    assert n == spp
    assert (st.rays, st.shaded_hits, st.sky_misses) == (ost["rays"], ost["shaded_hits"], ost["sky_misses"])
    hit_all = oracle_all_samples_hit(osc, c, w, h, 5, spp)
    assert hit_all.any()
    assert np.array_equal(img[hit_all], acc[hit_all])            # pure AO pixels: exact
    assert 0.0 < acc[hit_all][:, 0].mean() / spp < 1.0           # some occlusion, not all
    assert rel_mse(img, acc) < 1e-6


def oracle_all_samples_hit(osc, c, w, h, first, spp):
    m = np.ones((h, w), dtype=bool)
    for s in range(first, first + spp):
        m &= osc.primary_hits(c, w, h, s)["prim"] != api.PT_MISS
    return m


# ---------------------------------------------------------------------------------------------- material extension (GGX, patterns)
def test_ggx_and_pattern_functions_match_oracle(ctx, oracle):
    """Extension beyond the reference (ggx_device.h <-> oracle/ggx.h): the procedural albedo and the lobe decision
    are pure fp32 arithmetic + the integer hash (exact); sampled directions go through sin/cos/sqrt (rtol 5e-5)."""
    x = function_inputs(api.PT_FN_ALBEDO_PATTERN, 4096, 21)
    assert np.array_equal(ctx.debug_eval(api.PT_FN_ALBEDO_PATTERN, x), oracle.eval(api.PT_FN_ALBEDO_PATTERN, x))
    x = function_inputs(api.PT_FN_GGX_SAMPLE, 20000, 22)
    d = ctx.debug_eval(api.PT_FN_GGX_SAMPLE, x)
    o = oracle.eval(api.PT_FN_GGX_SAMPLE, x)
    assert np.array_equal(d[:, 7], o[:, 7])
    same = d[:, 0] == o[:, 0]                       # alive flag: may flip only for directions grazing the surface
    assert same.mean() > 0.9995
    np.testing.assert_allclose(d[same, 1:4], o[same, 1:4], rtol=5e-5, atol=2e-6)      # throughput weight
    # direction: components of a unit vector near 0 — the atol of test_shading_functions_match_oracle_and_fixture (the
    # diffuse lobe is GetCosineWeightedHemisphereDirection, whose libm form in the oracle loses ~3e-8/sqrt(u.y) to acos)
    np.testing.assert_allclose(d[same, 4:7], o[same, 4:7], rtol=5e-5, atol=1e-5)


def test_material_extension_image_matches_oracle(ctx, oracle):
    sc = scenes.with_material_ext(scenes.atrium(1))
    ctx.scene_create(sc)
    osc = oracle.scene(sc)
    atm = oracle.earth_atmosphere()
    ctx.sky_set(atm)
    olut = oracle.transmittance_lut(atm)
    w, h, spp = 200, 112, 4
    c = lookat_consts(w, h, scenes.ATRIUM_CAM)
    ctx.accum_reset(w, h)
    ctx.stats_reset()
    ctx.render(c, w, h, 0, spp)
    img = ctx.accum_read()
    st = ctx.stats()
    acc, ost, _ = osc.render(c, w, h, 0, spp, lut=olut)
    assert abs(st.rays - ost["rays"]) <= 1e-4 * ost["rays"]
    assert rel_mse(img, acc / spp) < 1e-4, rel_mse(img, acc / spp)
    plain = scenes.atrium(1)
    ctx.scene_create(plain)                             # a new scene drops the extension
    ctx.accum_reset(w, h)
    ctx.render(c, w, h, 0, spp)
    base = ctx.accum_read()
    assert rel_mse(base, acc / spp) > 1e-3              # the extension does change the picture
    plain.materials_ext = [(2.0, 0.0, 0, 1.0, 0)] * len(plain.materials)      # roughness outside [0, 1]
    with pytest.raises(api.PtError):
        ctx.scene_set_material_ext(plain)


# ---------------------------------------------------------------------------------------------- two pipeline lanes
def test_albedo_image_texture_matches_oracle(ctx, oracle):
    """Albedo pattern 3 of the material extension (pt_scene_set_albedo_texture; BASELINE C4 "textured GGX materials", not in
    the reference): a quad filling the view, its material sampling the brick image at uv x 3, lit by a constant white
    environment (sky off, fallback emission 1), depth 1 — the image IS the filtered texture times the BRDF weight. GPU vs
    the oracle's restatement (oracle/ggx.h), for the Lambert + Phong and the GGX model; removing the image changes it."""
    v = np.zeros(4, dtype=api.VERTEX_DTYPE)
    v["pos"] = [[-4, -3, 6], [4, -3, 6], [4, 3, 6], [-4, 3, 6]]
    v["normal"] = (0, 0, -1)
    v["uv"] = [[0, 0], [1, 0], [1, 1], [0, 1]]
    w, h, spp = 256, 192, 8
    for flags in (0, api.PT_MAT_GGX):
        sc = SceneData([(v, np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32))], [(0, 0, None)], [((0, 0, 0), (0.9, 0.8, 0.7, 1))])
        sc.materials_ext = [(0.6, 0.0, 3, 3.0, flags)]
        sc.albedo_texture = scenes.albedo_image(5, 64)
        ctx.scene_create(sc)
        osc = oracle.scene(sc)
        c = lookat_consts(w, h, {"eye": (0.0, 0.0, 0.0), "target": (0.0, 0.0, 6.0)}, depth=1, sky=False)
        for k in range(3):
            c.mySkyFallbackEmission[k] = 1.0
        ctx.accum_reset(w, h)
        ctx.render(c, w, h, 0, spp)
        img = ctx.accum_read()
        acc, _, _ = osc.render(c, w, h, 0, spp)
        assert rel_mse(img, acc / spp) < 1e-8, (flags, rel_mse(img, acc / spp))
        rows = img[h // 2, :, 0]
        assert rows.std() > 0.02 * rows.mean() > 0                    # the bricks are visible along a scan line
        ctx.scene_set_albedo_texture(None)                           # no image: factor 1
        ctx.accum_reset(w, h)
        ctx.render(c, w, h, 0, spp)
        flat = ctx.accum_read()
        assert rel_mse(flat, acc / spp) > 1e-3 and flat[..., 0].mean() > img[..., 0].mean()
    with pytest.raises(api.PtError):
        ctx.scene_set_albedo_texture(np.zeros((0, 4, 4), dtype=np.uint8))


def test_pipeline_lanes_are_bit_identical_to_single_lane(ctx, oracle):
    """PT_OPT_PIPELINE splits a batch over two streams; the halves are resolved in sample order, so the target and
    the counters must not change by a bit (odd sample counts and several batches included)."""
    sc = scenes.atrium(1)
    ctx.scene_create(sc)
    ctx.sky_set(oracle.earth_atmosphere())
    w, h = 200, 120
    c = lookat_consts(w, h, scenes.ATRIUM_CAM)
    out = {}
    try:
        for pipe in (0, 1):
            ctx.set_option(api.PT_OPT_PIPELINE, pipe)
            ctx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, 5)
            ctx.accum_reset(w, h)
            ctx.stats_reset()
            ctx.render(c, w, h, 3, 13)          # batches of 5, 5, 3 -> lanes of 3+2, 3+2, 2+1
            ctx.render(c, w, h, 16, 1)          # a single sample cannot be split
            img, n = ctx.accum_read_sum()
            st = ctx.stats()
            out[pipe] = (img, n, st.rays, st.shaded_hits, st.sky_misses, st.samples, list(st.rays_bounce))
    finally:
        ctx.set_option(api.PT_OPT_PIPELINE, 0)
        ctx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, 0)
    assert out[0][1:] == out[1][1:]
    assert np.array_equal(out[0][0], out[1][0])
    assert out[0][1] == 14


def test_batch_size_never_changes_the_image(ctx, oracle):
    """PT_OPT_SAMPLES_PER_BATCH only groups samples into wavefront batches: the RNG is keyed on the sample index and
    the resolve adds in sample order, so the target is bit-identical for any batch size."""
    ctx.scene_create(scenes.atrium(1))
    ctx.sky_set(oracle.earth_atmosphere())
    w, h = 120, 70
    c = lookat_consts(w, h, scenes.ATRIUM_CAM)
    imgs = []
    try:
        for spb in (0, 1, 2, 3):
            ctx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, spb)
            ctx.accum_reset(w, h)
            ctx.render(c, w, h, 1, 5)
            imgs.append(ctx.accum_read_sum()[0])
        assert all(np.array_equal(imgs[0], im) for im in imgs[1:])
        # k_resolve's two paths: batches of a multiple of 8 samples go through the LDS transpose (path ids are pixel-major),
        # others through the per-thread loop; frames that do not fill the last 256-pixel block, one and several rounds
        for (w2, h2), n in (((37, 23), 24), ((64, 8), 16), ((9, 5), 8)):
            c2 = lookat_consts(w2, h2, scenes.ATRIUM_CAM, depth=2)
            ref = None
            for spb in (1, 8, 0, 3):
                ctx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, spb)
                ctx.accum_reset(w2, h2)
                ctx.render(c2, w2, h2, 2, n)
                img = ctx.accum_read_sum()[0]
                ref = img if ref is None else ref
                assert np.array_equal(ref.view(np.uint32), img.view(np.uint32)), (w2, h2, spb)
    finally:
        ctx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, 0)


def test_pt_reduce_adds_targets_of_several_contexts(ctx, cornell):
    """pt_reduce (single-process multi-GPU): two more contexts on this device stand in for two more GPUs."""
    from pathtracer_amd.context import Context, reduce_contexts
    sc, _ = cornell
    w, h = 64, 48
    c = cornell_consts(w, h)
    others = [Context(0), Context(0)]
    try:
        parts = []
        for k, cx in enumerate([ctx] + others):
            cx.scene_create(sc)
            cx.accum_reset(w, h)
            cx.render(c, w, h, 4 * k, 4)
            parts.append(cx.accum_read_sum()[0])
        reduce_contexts([ctx] + others)
        total, n = ctx.accum_read_sum()
        assert n == 12
        assert np.array_equal(total, (parts[0] + parts[1]) + parts[2])
        assert np.array_equal(others[0].accum_read_sum()[0], parts[1])        # sources untouched
        others[1].accum_reset(w + 1, h)
        with pytest.raises(api.PtError):
            reduce_contexts([ctx] + others)
    finally:
        for cx in others:
            cx.close()


def test_virtual_partitions_give_identical_bits_for_1_2_4_8_ranks(ctx, oracle):
    """SURVEY §8e on the device: the ranks of a 1/2/4/8-GPU job are played one after the other on this GPU."""
    import torch
    from pathtracer_amd import distributed as ptd
    sc = scenes.atrium(1)
    ctx.scene_create(sc)
    ctx.sky_set(oracle.earth_atmosphere())
    w, h, total = 160, 90, 21
    c = lookat_consts(w, h, scenes.ATRIUM_CAM)
    dev = torch.device("cuda", 0)
    ref = None
    for world in (1, 2, 4, 8):
        parts = [ptd.render_virtual_partitions(ctx, c, w, h, total, world, r, dev) for r in range(world)]
        tot = ptd.sum_blocks_in_order(parts).cpu().numpy()
        ref = tot if ref is None else ref
        assert np.array_equal(tot, ref), world
    ctx.accum_reset(w, h)
    ctx.render(c, w, h, 0, total)
    single, n = ctx.accum_read_sum()
    assert n == total
    assert rel_mse(ref.reshape(h, w, 4), single) < 1e-10


# ---------------------------------------------------------------------------------------------- Russian roulette (extension)
def test_russian_roulette_matches_oracle_and_is_unbiased(ctx, cornell, oracle):
    """PT_OPT_RUSSIAN_ROULETTE: same rays, hits and image as the oracle's restatement (ray counts exact, image to the
    float tolerance); fewer rays than without; and the converged image agrees with the roulette-free one."""
    sc, osc = cornell
    ctx.scene_create(sc)
    w, h, spp = 64, 64, 8
    c = cornell_consts(w, h, depth=12, sky=False, fallback=0.0)
    res = {}
    try:
        for rr in (0, 1):
            ctx.set_option(api.PT_OPT_RUSSIAN_ROULETTE, rr)
            oracle.set_russian_roulette(rr)
            ctx.accum_reset(w, h)
            ctx.stats_reset()
            ctx.render(c, w, h, 0, spp)
            img = ctx.accum_read()
            st = ctx.stats()
            acc, ost, _ = osc.render(c, w, h, 0, spp)
            assert (st.rays, st.shaded_hits, st.sky_misses) == (ost["rays"], ost["shaded_hits"], ost["sky_misses"]), rr
            assert rel_mse(img, acc / spp) < 1e-6, rr
            res[rr] = (st.rays, img)
        assert res[1][0] < 0.9 * res[0][0]                 # the roulette removes long paths
        # unbiasedness, at a sample count where Monte-Carlo noise is below the bound: compare image means
        means = {}
        for rr in (0, 1):
            ctx.set_option(api.PT_OPT_RUSSIAN_ROULETTE, rr)
            ctx.accum_reset(w, h)
            ctx.render(c, w, h, 100, 512)
            means[rr] = ctx.accum_read()[..., :3].mean()
        assert abs(means[1] - means[0]) / means[0] < 0.02, means
    finally:
        ctx.set_option(api.PT_OPT_RUSSIAN_ROULETTE, 0)
        oracle.set_russian_roulette(0)


# ---------------------------------------------------------------------------------------------- sky mode 1 (extension)
@pytest.mark.parametrize("sun_dir", [(0.0, 1.0, 0.0), (0.3, 0.8, 0.52), (-0.7, 0.12, 0.2), (0.2, 0.02, -0.9)])
def test_sky_view_lut_tracks_marched_sky(ctx, oracle, sun_dir):
    """PT_OPT_SKY_MODE=1 replaces SampleSkyLuminance by a 256x256 direction LUT (sky_device.h). The reference has no
    such path in its ray-tracing pipeline, so the check is against the library's own marched sky (which is pinned to
    the oracle in test_sky_function_matches_oracle). Tolerances: relMSE < 1e-4 over uniformly distributed sky-facing
    directions; above 1 degree of elevation every direction within 1 % of the march. Below that the march itself
    jumps by up to 25 % where a sample point crosses 10 m altitude (sky_device.h, "Known limit") and the table
    smooths the step over one row, so only the aggregate bound applies there."""
    ctx.sky_set(oracle.earth_atmosphere())
    c = lookat_consts(64, 64, scenes.ATRIUM_CAM, sun_dir=sun_dir)
    rng = np.random.default_rng(11)
    n = 50000
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:, 1] = np.abs(d[:, 1])                         # sky-facing directions: what a miss ray in an outdoor scene has
    rows = np.concatenate([np.tile(np.asarray(c.myCameraPos, dtype=np.float32), (n, 1)), d], axis=1).astype(np.float32)
    march = ctx.debug_eval(api.PT_FN_SKY, rows, c).astype(np.float64)
    lut = ctx.debug_eval(api.PT_FN_SKY_VIEW_LUT, rows, c).astype(np.float64)
    assert np.isfinite(lut).all()
    r = ((lut - march) ** 2).mean() / (march ** 2).mean()
    assert r < 1e-4, (sun_dir, r)
    high = d[:, 1] > np.sin(np.radians(1.0))
    rel = np.abs(lut[high] - march[high]).max(axis=1) / march[high].max(axis=1)
    assert rel.max() < 0.01, (sun_dir, rel.max())


def test_sky_mode_1_image_matches_mode_0(ctx, oracle):
    sc = scenes.atrium(1)
    ctx.scene_create(sc)
    ctx.sky_set(oracle.earth_atmosphere())
    w, h = 240, 135
    c = lookat_consts(w, h, scenes.ATRIUM_CAM)
    imgs = []
    try:
        for mode in (0, 1):
            ctx.set_option(api.PT_OPT_SKY_MODE, mode)
            ctx.accum_reset(w, h)
            ctx.render(c, w, h, 0, 4)
            imgs.append(ctx.accum_read())
    finally:
        ctx.set_option(api.PT_OPT_SKY_MODE, 0)
    assert imgs[0][..., :3].mean() > 0
    assert rel_mse(imgs[1], imgs[0]) < 1e-4, rel_mse(imgs[1], imgs[0])
    with pytest.raises(api.PtError):
        ctx.set_option(api.PT_OPT_SKY_MODE, 2)


# ---------------------------------------------------------------------------------------------- full-size properties
def test_full_size_properties_1080p(ctx, oracle):
    """At BASELINE's full size (1920x1080 atrium) the oracle is too slow: check size-independent properties."""
    sc = scenes.atrium(1)
    ctx.scene_create(sc)
    ctx.sky_set(oracle.earth_atmosphere())
    w, h = 1920, 1080
    c = lookat_consts(w, h, scenes.ATRIUM_CAM)
    ctx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, 4)
    ctx.accum_reset(w, h)
    ctx.stats_reset()
    ctx.render(c, w, h, 0, 8)
    a, n = ctx.accum_read_sum()
    st = ctx.stats()
    assert n == 8 and st.samples == w * h * 8
    assert st.rays == st.shaded_hits + st.sky_misses                 # every ray either shades a hit or samples the sky
    assert w * h * 8 <= st.rays <= w * h * 8 * 5                     # depth 4 => at most 5 rays per sample
    assert np.isfinite(a).all() and (a[..., :3] >= 0).all() and (a[..., 3] == 0).all()
    # determinism + independence from the batch split: 8 at once == 3 + 5, bit for bit (resolve adds in sample order)
    ctx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, 3)
    ctx.accum_reset(w, h)
    ctx.render(c, w, h, 0, 3)
    ctx.render(c, w, h, 3, 5)
    b, _ = ctx.accum_read_sum()
    assert np.array_equal(a, b)
    # virtual 2-GPU partition on one device: sum of the two per-"rank" targets == single target up to fp32 reassociation
    ctx.accum_reset(w, h)
    ctx.render(c, w, h, 0, 4)
    p0, _ = ctx.accum_read_sum()
    ctx.accum_reset(w, h)
    ctx.render(c, w, h, 4, 4)
    p1, _ = ctx.accum_read_sum()
    assert rel_mse(p0 + p1, a) < 1e-12
    # counting traversal returns the same image and sane per-ray figures
    ctx.set_option(api.PT_OPT_COUNT_TRAVERSAL, 1)
    ctx.stats_reset()
    ctx.accum_reset(w, h)
    ctx.render(c, w, h, 0, 3)
    ctx.render(c, w, h, 3, 5)
    cimg, _ = ctx.accum_read_sum()
    st = ctx.stats()
    ctx.set_option(api.PT_OPT_COUNT_TRAVERSAL, 0)
    ctx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, 0)
    assert np.array_equal(cimg, a)
    assert 2 < st.nodes_visited / st.rays < 200 and 0.5 < st.tris_tested / st.rays < 100


# ---------------------------------------------------------------------------------------------- fused primaries
@pytest.mark.parametrize("sky,ext,size", [(True, False, (200, 120)), (False, False, (200, 120)), (True, True, (200, 120)),
                                          (True, False, (204, 118)), (True, False, (64, 8)), (True, False, (8, 72))])
def test_fused_primary_rays_are_bit_identical_to_the_generate_pass(ctx, oracle, sky, ext, size):
    """PT_OPT_FUSED_PRIMARY (default): primaries are computed inside k_extend / k_shade and the path state starts as
    constants instead of a k_generate pass through memory. Same RayGen arithmetic (PathTracing.hlsl:40-61) in all
    three places, so target sums, ray counts and per-bounce queues must not change by a bit; both pipeline settings.
    Frames whose width and height are multiples of 8 fetch their fused primaries in 8x8-pixel tile order (PrimaryRayOfSlot,
    shading_device.h), other frames in row order: both kinds, and the one-tile-row / one-tile-column edge cases."""
    sc = scenes.with_material_ext(scenes.atrium(1)) if ext else scenes.atrium(1)
    ctx.scene_create(sc)
    ctx.sky_set(oracle.earth_atmosphere())
    (w, h), spp = size, 5
    c = lookat_consts(w, h, scenes.ATRIUM_CAM, depth=4)
    if not sky:
        c.mySampleSky = 0
        api.fill3(c.mySkyFallbackEmission, (1.5, 1.0, 0.5))
    out = {}
    try:
        for fused in (0, 1):
            for pipe in (0, 1):
                ctx.set_option(api.PT_OPT_FUSED_PRIMARY, fused)
                ctx.set_option(api.PT_OPT_PIPELINE, pipe)
                ctx.accum_reset(w, h)
                ctx.stats_reset()
                ctx.render(c, w, h, 3, spp)
                st = ctx.stats()
                out[(fused, pipe)] = (ctx.accum_read_sum()[0].copy(), (st.rays, st.shaded_hits, st.sky_misses, tuple(st.rays_bounce)))
        ctx.set_option(api.PT_OPT_PIPELINE, 0)
        ctx.accum_reset(w, h)
        ctx.render(c, w, h, 3, spp)                        # one lane again: the whole batch sits in the first lane's queues
        q1 = ctx.debug_read_queue(1)                       # later queues stay readable
        assert len(q1) == out[(1, 0)][1][3][1]
        with pytest.raises(Exception):
            ctx.debug_read_queue(0)                        # ... the bounce-0 queue does not exist in fused mode
    finally:
        ctx.set_option(api.PT_OPT_FUSED_PRIMARY, 1)
        ctx.set_option(api.PT_OPT_PIPELINE, 0)
    ref = out[(0, 0)]
    for k, v in out.items():
        assert v[1] == ref[1], k
        assert np.array_equal(v[0].view(np.uint32), ref[0].view(np.uint32)), k


@pytest.mark.parametrize("pairs", [False, True], ids=["48-byte-triangles", "64-byte-pairs"])
@pytest.mark.parametrize("size", [(200, 120), (204, 118), (64, 8)])
def test_packet_kernel_for_primaries_is_bit_identical_to_the_per_lane_traversal(ctx, oracle, pairs, size):
    """PT_OPT_PRIMARY_PACKET (default 1): the fused primaries are traced by k_extend_packet, one BVH walk per wave with
    wave-uniform node / triangle fetches, instead of k_extend's walk per lane. The triangle test and the closest-hit rule
    are the same, so target sums, ray counts and the bounce-1 queue's content must not change by a bit: tiled and
    row-ordered frames (coherent and mixed-octant waves: the camera looks along an axis, so the image centre splits
    octants), both leaf-record formats, and a camera far outside the scene whose rays mostly miss."""
    ctx.scene_create(scenes.atrium(1), api.PT_SCENE_PAIR_TRIANGLES if pairs else api.PT_SCENE_NO_PAIRS)
    ctx.sky_set(oracle.earth_atmosphere())
    (w, h), spp = size, 3
    cams = [scenes.ATRIUM_CAM, {"eye": (0.0, 60.0, -90.0), "target": (0.0, 4.0, 0.0)}, {"eye": (1.0, 1.7, 0.5), "target": (1.0, 1.7, 9.0)}]
    try:
        for cam in cams:
            c = lookat_consts(w, h, cam, depth=2)
            out = {}
            for packet in (0, 1, 2):                                               # 2 = the library times both and picks
                ctx.set_option(api.PT_OPT_PRIMARY_PACKET, packet)
                ctx.accum_reset(w, h)
                ctx.stats_reset()
                ctx.render(c, w, h, 5, spp)
                st = ctx.stats()
                q1 = ctx.debug_read_queue(1)
                key = np.sort(q1.view(np.uint32).reshape(len(q1), 8), axis=0)      # queue order depends on block scheduling
                out[packet] = (ctx.accum_read_sum()[0].copy(), (st.rays, st.shaded_hits, st.sky_misses, tuple(st.rays_bounce)), key)
                assert (st.packet_launches > 0) == (packet == 1) or packet == 2
            for packet in (1, 2):
                assert out[0][1] == out[packet][1], cam
                assert np.array_equal(out[0][0].view(np.uint32), out[packet][0].view(np.uint32)), cam
                assert np.array_equal(out[0][2], out[packet][2]), cam
            assert out[1][1][0] > w * h * spp                                       # some secondary rays exist
        assert min(ctx.stats().packet_cal_ms) > 0.0                                 # the automatic mode did time both kernels
    finally:
        ctx.set_option(api.PT_OPT_PRIMARY_PACKET, 2)


@pytest.mark.parametrize("scene_name", ["cornell", "atrium", "atrium_ext"])
def test_bounce_queues_match_the_oracle_path_probe(ctx, oracle, scene_name):
    """SURVEY 8 rows a7-a9 in isolation (payload, ClosestHit PathTracing.hlsl:16-38, the loaders Common.hlsl:75-127):
    the ray queue k_shade writes for bounce b is, row for row, what the oracle's RayGen loop carries into iteration b
    (orc_bounce_rays): origin = myHitPos = origin + dir * t, direction sampled about the interpolated, viewer-facing
    normal, TMin 0.001 / TMax 10000 (:126-127). Bounce 1: origins BIT-identical (no transcendental between the primary ray
    and myHitPos), directions within the shading tolerance (2e-4 absolute on unit vectors; the kernels use the hardware
    transcendental unit). Later bounces inherit the direction tolerance through the hit point, so their rows are
    matched one to one by nearest neighbour and >= 99.8 % must agree (the rest are paths that took another triangle after
    a last-ulp difference, as in the image tests). Queue lengths are exact at every bounce."""
    from scipy.spatial import cKDTree
    if scene_name == "cornell":
        ctx.scene_create(scenes.cornell_box())
        w, h = 96, 96
        c = cornell_consts(w, h)
    else:
        sc = scenes.atrium(1)
        ctx.scene_create(scenes.with_material_ext(sc) if scene_name == "atrium_ext" else sc)
        w, h = 160, 96
        c = lookat_consts(w, h, scenes.ATRIUM_CAM, depth=3)
    osc = oracle.scene(scenes.cornell_box() if scene_name == "cornell" else
                       (scenes.with_material_ext(scenes.atrium(1)) if scene_name == "atrium_ext" else scenes.atrium(1)))
    atm = oracle.earth_atmosphere()
    ctx.sky_set(atm)
    lut = oracle.transmittance_lut(atm)
    sample = 7
    for b in (1, 2, 3):
        # the queues ping-pong between two buffers, so only the LAST bounce's queue is whole after a render: one render per
        # bounce with the recursion depth cut to it (a path's first b bounces do not depend on the depth)
        c.myMaxRecursionDepth = b
        ctx.accum_reset(w, h)
        ctx.stats_reset()
        ctx.render(c, w, h, sample, 1)
        st = ctx.stats()
        q = ctx.debug_read_queue(b)
        r, alive, _ = osc.bounce_rays(c, w, h, sample, b, lut=lut)
        ref = r[alive]
        assert len(ref) > 0.05 * w * h, (b, len(ref))
        assert abs(len(q) - len(ref)) <= max(1, len(ref) // 2000) * (b > 1), (b, len(q), len(ref))
        assert len(q) == st.rays_bounce[b]
        assert (q["tmin"] == np.float32(0.001)).all() and (q["tmax"] == np.float32(10000.0)).all()
        if b == 1:
            assert len(q) == len(ref)
            ko = np.lexsort(ref["origin"].view(np.uint32).T[::-1])
            kq = np.lexsort(q["origin"].view(np.uint32).T[::-1])
            assert np.array_equal(q["origin"][kq].view(np.uint32), ref["origin"][ko].view(np.uint32))
            uniq = np.ones(len(ko), dtype=bool)          # two paths on one hit point (none expected) would make the pairing ambiguous
            same = (ref["origin"][ko][1:] == ref["origin"][ko][:-1]).all(axis=1)
            uniq[1:] &= ~same
            uniq[:-1] &= ~same
            d = np.abs(q["dir"][kq] - ref["dir"][ko])[uniq]
            assert uniq.mean() > 0.99 and d.max() < 2e-4, (uniq.mean(), d.max())
            continue
        extent = float(np.ptp(ref["origin"], axis=0).max())
        feat = lambda a: np.concatenate([a["origin"] / extent, a["dir"]], axis=1).astype(np.float64)
        dist, idx = cKDTree(feat(q)).query(feat(ref), k=1)
        ok = dist < 2e-3
        assert ok.mean() >= 0.998, (b, ok.mean(), np.sort(dist)[-5:])
        assert len(np.unique(idx[ok])) == int(ok.sum())   # one to one


def test_bounce_queue_matches_the_committed_fixture(ctx):
    """The same check against data instead of the live oracle: tests/golden/cornell_bounce_rays_64.npz (written by
    make_golden.py from the oracle) holds the rays the paths of sample 3 carry into bounce 1 of the Cornell box at 64x64;
    k_shade's bounce-1 queue has exactly those origins (bit for bit) and directions within the shading tolerance."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cornell_bounce_rays_64.npz"))
    want = g["rays_1"]
    ctx.scene_create(scenes.cornell_box())
    c = cornell_consts(64, 64)
    c.myMaxRecursionDepth = 1
    ctx.accum_reset(64, 64)
    ctx.render(c, 64, 64, 3, 1)
    q = ctx.debug_read_queue(1)
    assert len(q) == len(want) == int(np.unpackbits(g["alive_1"]).sum())
    ko = np.lexsort(want["origin"].view(np.uint32).T[::-1])
    kq = np.lexsort(q["origin"].view(np.uint32).T[::-1])
    assert np.array_equal(q["origin"][kq].view(np.uint32), want["origin"][ko].view(np.uint32))
    assert np.abs(q["dir"][kq] - want["dir"][ko]).max() < 2e-4
    assert (q["tmin"] == np.float32(0.001)).all() and (q["tmax"] == np.float32(10000.0)).all()


# ---------------------------------------------------------------------------------------------- API behaviour
def test_default_batch_shrinks_when_device_memory_is_short():
    """The default wavefront batch is 2^27 paths (20 GB). When that much device memory is not free, pt_render halves the
    batch until it fits instead of failing, and the image does not depend on the batch size."""
    import torch
    from pathtracer_amd.context import Context
    cx = Context(0)
    hold = None
    try:
        cx.scene_create(scenes.cornell_box())
        w = h = 1024
        spp = 128                                  # 2^27 paths at this size
        c = cornell_consts(w, h, sky=False, fallback=1.0)
        cx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, 8)
        cx.accum_reset(w, h)
        cx.render(c, w, h, 0, spp)
        ref = cx.accum_read_sum()[0]
        cx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, 0)
        free, _ = torch.cuda.mem_get_info(0)
        hold = torch.empty(max(0, free - (6 << 30)), dtype=torch.uint8, device="cuda:0")    # leave 6 GB + the old buffers
        cx.accum_reset(w, h)
        cx.render(c, w, h, 0, spp)                 # 20 GB and 10 GB do not fit, 5 GB does
        got = cx.accum_read_sum()[0]
        assert rel_mse(got, ref) < 1e-12
    finally:
        del hold
        torch.cuda.empty_cache()      # hand the block back to the driver: later tests (and their subprocesses) allocate through HIP
        cx.close()


def test_api_call_sequences_with_reallocation(oracle):
    """One context driven through scene replacement, target resizes (growing and shrinking wavefront state), both
    pipelines, both lanes and option changes; every stage is checked against the oracle."""
    from pathtracer_amd.context import Context
    cx = Context(0)
    try:
        atm = oracle.earth_atmosphere()
        cx.sky_set(atm)
        olut = oracle.transmittance_lut(atm)
        cornell, atrium = scenes.cornell_box(), scenes.atrium(1)
        ocornell, oatrium = oracle.scene(cornell), oracle.scene(atrium)

        def check(scene_o, c, w, h, first, spp, ao=False):
            cx.accum_reset(w, h)
            (cx.render_ao if ao else cx.render)(c, w, h, first, spp)
            img = cx.accum_read_sum()[0]
            ref = (scene_o.render_ao(c, w, h, first, spp, lut=olut)[0] if ao else scene_o.render(c, w, h, first, spp, lut=olut)[0])
            assert rel_mse(img, ref) < 1e-4, (w, h, spp, ao, rel_mse(img, ref))      # the north-star bar; typically 1e-8..1e-6

        cx.scene_create(cornell)
        check(ocornell, cornell_consts(64, 64), 64, 64, 0, 2)
        cx.scene_create(atrium)                                   # replaces the scene (new BVH, new spill regions)
        cx.set_option(api.PT_OPT_PIPELINE, 1)
        check(oatrium, lookat_consts(100, 60, scenes.ATRIUM_CAM), 100, 60, 7, 5)      # grows both lanes
        check(oatrium, lookat_consts(30, 20, scenes.ATRIUM_CAM), 30, 20, 0, 3)        # smaller frame, same buffers
        cx.set_option(api.PT_OPT_PIPELINE, 0)
        cx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, 2)
        check(oatrium, lookat_consts(120, 70, scenes.ATRIUM_CAM), 120, 70, 1, 5)      # batches of 2, 2, 1
        cx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, 0)
        cx.scene_create(cornell)
        c = cornell_consts(48, 48, sky=True)
        c.myAoDistance = 30.0
        check(ocornell, c, 48, 48, 3, 2, ao=True)
        cx.sky_set(atm)                                           # recomputing the LUT is idempotent
        assert np.array_equal(cx.sky_lut_read(), olut)
        hits = cx.primary_hits(cornell_consts(200, 100), 200, 100, 1)     # larger than any wavefront state so far
        assert np.array_equal(hits["prim"], ocornell.primary_hits(cornell_consts(200, 100), 200, 100, 1)["prim"])
        check(ocornell, cornell_consts(64, 64), 64, 64, 0, 2)
    finally:
        cx.close()


def test_ao_buffers_survive_a_growing_trace_staging_buffer(oracle):
    """Regression (round-1 review): pt_trace_rays growing its staging buffers used to free the AO ray buffers without
    resetting their capacity. Order: render_ao -> trace_rays with more rays than ever before -> render_ao again."""
    from pathtracer_amd.context import Context
    cx = Context(0)
    try:
        atm = oracle.earth_atmosphere()
        cx.sky_set(atm)
        olut = oracle.transmittance_lut(atm)
        sc = scenes.cornell_box()
        osc = oracle.scene(sc)
        cx.scene_create(sc)
        w, h, spp = 80, 60, 2
        c = cornell_consts(w, h, sky=True)
        c.myAoDistance = 40.0
        ref, _ = osc.render_ao(c, w, h, 0, spp, lut=olut)
        for rnd in range(2):
            cx.accum_reset(w, h)
            cx.render_ao(c, w, h, 0, spp)
            hit_all = oracle_all_samples_hit(osc, c, w, h, 0, spp)
            img = cx.accum_read_sum()[0]
            assert np.array_equal(img[hit_all], ref[hit_all]), rnd
            assert rel_mse(img, ref) < 1e-6, rnd
            rays = random_rays(3 + rnd, 400000 * (rnd + 1), (-10, 95, -10), (10, 110, 10))   # grows tmp_rays / tmp_hits
            got = cx.trace_rays(rays)
            assert np.array_equal(got["prim"], osc.trace_rays(rays)["prim"])
            occ = cx.occluded(rays[:1000])
            assert np.array_equal(occ.astype(bool), got["prim"][:1000] != api.PT_MISS)
    finally:
        cx.close()


def test_error_codes(oracle):
    from pathtracer_amd import Context
    c2 = Context(0)
    consts = cornell_consts(8, 8)
    with pytest.raises(api.PtError) as e:
        c2.render(consts, 8, 8, 0, 1)
    assert e.value.code == api.PT_ERR_NO_SCENE
    c2.scene_create(scenes.cornell_box())
    with pytest.raises(api.PtError) as e:
        c2.render(consts, 8, 8, 0, 1)
    assert e.value.code == api.PT_ERR_NO_TARGET
    c2.accum_reset(8, 8)
    with pytest.raises(api.PtError) as e:
        c2.render(consts, 16, 8, 0, 1)
    assert e.value.code == api.PT_ERR_SIZE_MISMATCH
    sky = cornell_consts(8, 8, sky=True)
    with pytest.raises(api.PtError) as e:
        c2.render(sky, 8, 8, 0, 1)
    assert e.value.code == api.PT_ERR_NO_SKY
    consts.myMaxRecursionDepth = 1000
    with pytest.raises(api.PtError) as e:
        c2.render(consts, 8, 8, 0, 1)
    assert e.value.code == api.PT_ERR_INVALID_ARG
    consts.myMaxRecursionDepth = 0                   # depth 0: exactly one ray per sample (inclusive loop bound, Q2)
    c2.stats_reset()
    c2.render(consts, 8, 8, 0, 2)
    c2.render(consts, 8, 8, 2, 0)                    # zero samples is a no-op
    assert c2.stats().rays == 8 * 8 * 2
    bad = SceneData([(np.zeros(3, dtype=api.VERTEX_DTYPE), np.array([0, 1, 7], dtype=np.uint32))], [(0, 0, None)], [((0, 0, 0), (1, 1, 1, 1))])
    with pytest.raises(api.PtError) as e:
        c2.scene_create(bad)
    assert e.value.code == api.PT_ERR_INVALID_ARG
    c2.close()


def test_external_target_via_torch_and_host_mirror(ctx, oracle):
    import torch
    from pathtracer_amd import PathTracer
    from pathtracer_amd import distributed as ptd
    pt = PathTracer(0, 128, 96)
    assert (pt.myOutputWidth, pt.myOutputHeight) == (64, 48)          # myHalfResRender default (PathTracer.cpp:849-852)
    sc = scenes.cornell_box()
    pt.LoadScene(sc, scenes.CORNELL_CAM_POS)
    pt.mySampleSky = False
    pt.mySkyFallbackIntensity = 0.0
    t = ptd.bind_torch_target(pt.ctx, 64, 48, torch.device("cuda", 0))
    pt.myAccumulationNeedsClear = False
    pt.RenderRT(4)
    pt.RenderRT(4)
    assert pt.myNumAccumulationFrames == 8
    pt.ctx.synchronize()
    total = t.reshape(48, 64, 4).cpu().numpy()
    osc = oracle.scene(sc)
    acc, _, _ = osc.render(pt.BuildRtConsts(), 64, 48, 0, 8)
    assert rel_mse(total, acc) < 1e-8
    red, cnt = ptd.reduce_sum_to_root(t, 8)                            # world size 1: identity
    assert cnt == 8 and red.data_ptr() == t.data_ptr()
    pt.ctx.accum_bind(0, 0, 0)
    pt.ctx.close()


def test_cpp_host_cli_matches_python_host(ctx, tmp_path):
    """host/pt_render_cli (C++ mirror of PathTracer/Sky/Camera over the C-ABI) renders the same image as the Python
    mirror: same library, same sample indices -> identical sums."""
    import subprocess
    from pathtracer_amd import PathTracer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "host", "pt_render_cli")
    assert os.path.exists(cli), "build host/pt_render_cli first (__graft_entry__.build())"
    sc = scenes.cornell_box()
    obj = str(tmp_path / "cornell.obj")
    scenes.scene_to_obj(sc, obj)
    out = str(tmp_path / "out.pfm")
    r = subprocess.run([cli, obj, out, "--spp", "8", "--width", "128", "--height", "96"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    with open(out, "rb") as f:
        assert f.readline() == b"PF\n"
        w, h = (int(x) for x in f.readline().split())
        assert f.readline().strip() == b"-1.0"
        img = np.frombuffer(f.read(), dtype="<f4").reshape(h, w, 3)[::-1]
    assert (w, h) == (64, 48)                                         # half-res default
    pt = PathTracer(0, 128, 96)
    pt.LoadScene(scenes.load_obj(obj), scenes.CORNELL_CAM_POS)
    pt.RenderRT(8)                                                    # reference defaults: sky on, light instance 4
    ref = pt.ReadOutput()
    pt.ctx.close()
    assert np.array_equal(img, ref[..., :3])
    bad = subprocess.run([cli, str(tmp_path / "missing.obj"), out], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 1 and "Failed importing scene" in bad.stderr


def test_textured_obj_file_renders_like_the_scene_it_was_written_from(ctx, oracle, tmp_path):
    """Scene ingestion of the material extension from files (MTL Pr / Pm / map_Kd -s / pt_albedo_pattern + an 8-bit PNG):
    a scene with every albedo kind and both material models, written by scene_to_obj and read back by load_obj (the path
    PathTracer.LoadScene takes for a file name), renders to the SAME bits as the in-memory scene, and within the image
    tolerance of the oracle given the file-loaded scene."""
    soup = scenes.random_triangle_soup(3, 3000, extent=6.0, size=1.5)
    v, i = soup.meshes[0]
    v["uv"] = np.random.default_rng(4).random((len(v), 2)).astype(np.float32) * 3.0
    parts = np.array_split(i.reshape(-1, 3), 6)
    sc = SceneData([(v, p.reshape(-1)) for p in parts], [(k, k, None) for k in range(6)],
                   [((0.3, 0.3, 0.3) if k == 5 else (0.0, 0.0, 0.0), (0.2 + 0.1 * k, 0.6, 0.9 - 0.1 * k, 1.0)) for k in range(6)])
    sc = scenes.with_material_ext(sc, seed=9)
    obj = str(tmp_path / "tex.obj")
    scenes.scene_to_obj(sc, obj)
    assert os.path.exists(str(tmp_path / "tex_albedo.png"))
    back = scenes.load_obj(obj)
    assert back.albedo_texture is not None and [e[2] for e in back.materials_ext] == [0, 1, 2, 3, 0, 1]
    w, h, spp = 160, 96, 8
    c = lookat_consts(w, h, {"eye": (3.0, 3.0, -7.0), "target": (3.0, 3.0, 3.0)}, depth=3)
    c.myLightInstanceId = 0xFFFFFFFF
    ctx.sky_set(oracle.earth_atmosphere())
    imgs = []
    for s in (sc, back):
        ctx.scene_create(s)
        ctx.accum_reset(w, h)
        ctx.render(c, w, h, 0, spp)
        imgs.append(ctx.accum_read_sum()[0].copy())
    assert np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32))
    lut = oracle.transmittance_lut(oracle.earth_atmosphere())
    ref, _, _ = oracle.scene(back).render(c, w, h, 0, spp, lut=lut)
    assert rel_mse(imgs[1][..., :3] / spp, ref[..., :3] / spp) < 1e-4
    # the C++ host (own MTL / PNG readers) renders the file to the same bits as the Python host
    import subprocess
    from pathtracer_amd import PathTracer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "host", "pt_render_cli")
    out = str(tmp_path / "tex.pfm")
    r = subprocess.run([cli, obj, out, "--spp", "4", "--width", "160", "--height", "96", "--cam", "3", "3", "-7", "--depth", "3",
                        "--light-instance", "5"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    from pathtracer_amd.image_io import read_pfm
    img = read_pfm(out)
    pt = PathTracer(0, 160, 96)
    pt.LoadScene(obj, (3.0, 3.0, -7.0))
    pt.myMaxRecursionDepth = 3
    pt.myLightInstanceIdx = 5
    pt.RenderRT(4)
    mine = pt.ReadOutput()
    pt.ctx.close()
    assert img.shape == mine[..., :3].shape and np.array_equal(img, mine[..., :3]) and img.max() > 0
