"""CPU tests of the host side: C-ABI library loads and exports every declared symbol, struct layouts match the
reference's constant buffer, camera / OBJ reader / scene generators / sample partitioning, and the N>1 reduce path
under gloo with world_size 2 (the per-rank 'renderer' is the CPU oracle, standing in for a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from pathtracer_amd import api, scenes
from pathtracer_amd.context import SceneData, encode_unorm_rgba, pack_material_ext, pack_scene
from pathtracer_amd.distributed import partition_samples
from pathtracer_amd.pathtracer import Camera, SetupEarthAtmosphere
from tests.common import cornell_camera, cornell_consts

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = api.load_library()
    header = open(os.path.join(ROOT, "include", "pt_api.h")).read() + open(os.path.join(ROOT, "include", "pt_api_tuning.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(pt_[a-z_]+)\s*\(", header, flags=re.M))
    assert declared and "pt_debug_set_tuning" in declared, "no declarations parsed from include/*.h"
    assert declared == set(api.SYMBOLS), declared ^ set(api.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_rtconsts_layout_matches_reference_offsets():
    """Byte offsets of RtConsts (PathTracer.cpp:709-740 == raytracing/Common.hlsl:24-55), SURVEY §8 a2."""
    off = {f[0]: getattr(api.PtRtConsts, f[0]).offset for f in api.PtRtConsts._fields_}
    want = {"myNearPlaneCorner": 0, "myAoDistance": 12, "myXAxis": 16, "myOutTexIndex": 28, "myYAxis": 32,
            "myAsIndex": 44, "myCameraPos": 48, "myInstanceDataBufferIndex": 60, "myMaterialDataBufferIndex": 64,
            "mySampleBufferIndex": 68, "myFrameRandomSeed": 72, "myNumAccumulationFrames": 76,
            "myLinearClampSamplerIndex": 80, "myMaxRecursionDepth": 84, "myLightInstanceId": 88,
            "myNumHaltonSamples": 92, "myLightEmission": 96, "mySampleSky": 108, "mySkyFallbackEmission": 112,
            "myPhongSpecularPower": 124, "mySkyConsts": 128}
    assert off == want
    sk = {f[0]: getattr(api.PtSkyConstants, f[0]).offset for f in api.PtSkyConstants._fields_}
    assert sk["mySunDirection"] == 112 and sk["myTransmissionLutTexIdx"] == 124 and sk["mySunIlluminance"] == 128
    assert sk["myRayMarchMinMaxSPP"] == 144 and C.sizeof(api.PtRtConsts) == 288 and C.sizeof(api.PtAtmosphere) == 112
    assert C.sizeof(api.PtMaterial) == 16 and C.sizeof(api.PtInstanceDesc) == 72


def test_ctypes_mirrors_have_the_c_header_struct_sizes(tmp_path):
    """The Python mirror (pathtracer_amd/api.py) and include/pt_api.h must agree on every struct that crosses the C ABI by
    pointer: a C program compiled against the header prints sizeof / offsetof, ctypes must report the same."""
    import subprocess
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pt_api_tuning.h"\nint main(void) {\n'
                   '  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", offsetof(PtStats, struct_size), offsetof(PtStats, samples), sizeof(PtStats), offsetof(PtStats, extend_launches_bounce), '
                   'offsetof(PtStats, packet_launches), offsetof(PtStats, packet_cal_ms), sizeof(PtSceneInfo), sizeof(PtAtmosphere), sizeof(PtInstanceDesc), '
                   'sizeof(PtRtConsts), sizeof(PtMaterialExt), sizeof(PtMeshDesc));\n  return 0;\n}\n')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [0, 8, C.sizeof(api.PtStats), api.PtStats.extend_launches_bounce.offset, api.PtStats.packet_launches.offset, api.PtStats.packet_cal_ms.offset,
            C.sizeof(api.PtSceneInfo), C.sizeof(api.PtAtmosphere), C.sizeof(api.PtInstanceDesc), C.sizeof(api.PtRtConsts), C.sizeof(api.PtMaterialExt),
            C.sizeof(api.PtMeshDesc)]
    assert got == want


def test_no_device_error_is_reported_not_hidden():
    """Without a GPU the product must fail loudly (there is no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = api.load_library()
    h = C.c_void_p()
    rc = lib.pt_create(0, C.byref(h))
    assert rc == api.PT_ERR_NO_DEVICE and not h.value
    assert b"no HIP device" in lib.pt_last_error(None)
    from pathtracer_amd import Context
    with pytest.raises(api.PtError):
        Context(0)
    assert lib.pt_render(None, None, 1, 1, 0, 1) == api.PT_ERR_INVALID_ARG
    assert lib.pt_destroy(None) == api.PT_ERR_INVALID_ARG


def test_camera_near_plane_and_primary_ray_geometry():
    cam = cornell_camera(512, 512)
    v = cam.GetVerticesOnNearPlane()
    half = np.float32(np.tan(np.radians(30.0)))
    np.testing.assert_allclose(v[0], [1 - half, 102 - half, -29], rtol=1e-6)
    np.testing.assert_allclose(v[1] - v[0], [2 * half, 0, 0], atol=2e-5)   # fp32 ulp at |y|~102 is 7.6e-6
    np.testing.assert_allclose(v[3] - v[0], [0, 2 * half, 0], atol=2e-5)
    cam2 = Camera()
    cam2.myWidth, cam2.myHeight = 1920.0, 1080.0
    cam2.LookAt((0, 0, 0), (0, 0, 5))
    w = cam2.GetVerticesOnNearPlane()
    np.testing.assert_allclose((w[1] - w[0])[0] / (w[3] - w[0])[1], 16 / 9, rtol=1e-6)
    c = cornell_consts(64, 64)
    assert tuple(c.myCameraPos) == (1.0, 102.0, -30.0) and c.myMaxRecursionDepth == 4 and c.myLightInstanceId == 4
    assert tuple(c.myLightEmission) == (100.0, 100.0, 100.0) and c.myPhongSpecularPower == 10.0
    assert tuple(c.mySkyConsts.myRayMarchMinMaxSPP) == (4.0, 14.0)


def test_cornell_fixture_matches_reference_scene_shape():
    sc = scenes.cornell_box()
    assert sc.num_triangles == 36 and len(sc.meshes) == 6 and len(sc.materials) == 5
    assert [len(i) // 3 for _, i in sc.meshes] == [6, 2, 2, 12, 2, 12]     # light = instance 4 (PathTracer.h:125)
    light = sc.meshes[4][0]
    np.testing.assert_allclose(light["pos"][:, 1], 107.944885, rtol=1e-7)
    np.testing.assert_allclose(light["normal"], [[0, -1, 0]] * len(light), atol=1e-6)
    allp = np.concatenate([v["pos"] for v, _ in sc.meshes])
    np.testing.assert_allclose(allp.min(0), [-4.732368, 97.092285, -6.766661], rtol=1e-6)
    cols = sorted(encode_unorm_rgba(c) & 0xFFFFFF for _, c in sc.materials)
    assert 0x000080 in cols and 0x008000 in cols            # red and green walls, Kd 0.5 -> 128


def test_obj_reader_roundtrip(tmp_path):
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 0.2 0.4 0.6\nKe 1 2 3\nnewmtl b\nKd 1 1 1\n")
    (tmp_path / "m.obj").write_text(
        "mtllib m.mtl\no one\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
        "usemtl a\nf 1/1/1 2/2/1 3/3/1 4/4/1\no two\nusemtl b\nf -4 -3 -2\n")
    sc = scenes.load_obj(str(tmp_path / "m.obj"))
    assert sc.num_triangles == 3 and len(sc.meshes) == 2
    assert sc.instances[0][:2] == (0, 0) and sc.instances[1][:2] == (1, 1)
    assert sc.materials[0] == ((1.0, 2.0, 3.0), (0.2, 0.4, 0.6, 1.0))
    v, i = sc.meshes[0]
    assert len(v) == 4 and list(i) == [0, 1, 2, 0, 2, 3] and tuple(v["uv"][2]) == (1.0, 1.0)
    v2, _ = sc.meshes[1]
    np.testing.assert_allclose(v2["normal"], [[0, 0, 1]] * 3)    # no vn: face normal
    m, nm, ins, ni, mats, nmat, keep = pack_scene(sc)
    assert (nm, ni, nmat) == (2, 2, 2) and m[0].num_triangles == 2 and list(ins[1].myTransform)[::5] == [1.0] * 4


def _png_bytes(w, h, ctype, px, filters, plte=None, trns=None):
    """A PNG encoder for the reader test only: 8-bit samples `px` (h, w, channels) with the given row filter per row."""
    import struct
    import zlib
    ch = px.shape[2]
    rows, prev = [], np.zeros(w * ch, dtype=np.int32)
    for y in range(h):
        cur = px[y].reshape(-1).astype(np.int32)
        left = np.concatenate([np.zeros(ch, dtype=np.int32), cur[:-ch]])
        upleft = np.concatenate([np.zeros(ch, dtype=np.int32), prev[:-ch]])
        ft = filters[y % len(filters)]
        if ft == 0:
            pred = 0
        elif ft == 1:
            pred = left
        elif ft == 2:
            pred = prev
        elif ft == 3:
            pred = (left + prev) >> 1
        else:
            pa, pb, pc = np.abs(prev - upleft), np.abs(left - upleft), np.abs(left + prev - 2 * upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
        rows.append(bytes([ft]) + ((cur - pred) & 255).astype(np.uint8).tobytes())
        prev = cur
    chunk = lambda tag, d: struct.pack(">I", len(d)) + tag + d + struct.pack(">I", zlib.crc32(tag + d) & 0xFFFFFFFF)
    raw = zlib.compress(b"".join(rows))
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
    if plte is not None:
        out += chunk(b"PLTE", plte.tobytes())
    if trns is not None:
        out += chunk(b"tRNS", trns.tobytes())
    return out + chunk(b"IDAT", raw[:len(raw) // 2]) + chunk(b"IDAT", raw[len(raw) // 2:]) + chunk(b"IEND", b"")


def test_png_and_ppm_readers(tmp_path):
    """Texture files for the albedo image (BASELINE C4 'textured'): 8-bit PNG in every colour type, with all
    five row filters and split IDAT chunks, and binary PPM, decode to the RGBA8 texels pt_scene_set_albedo_texture takes;
    damaged or unsupported files are a ValueError."""
    from pathtracer_amd import image_io as io
    rng = np.random.default_rng(5)
    w, h = 11, 9
    for ctype, ch in ((0, 1), (2, 3), (4, 2), (6, 4)):
        px = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
        f = tmp_path / f"t{ctype}.png"
        f.write_bytes(_png_bytes(w, h, ctype, px, [4, 1, 3, 2, 0]))
        img = io.read_image_rgba8(str(f))
        want = np.full((h, w, 4), 255, dtype=np.uint8)
        if ctype == 0:
            want[..., :3] = px
        elif ctype == 4:
            want[..., :3], want[..., 3] = px[..., :1], px[..., 1]
        else:
            want[..., :ch] = px
        assert np.array_equal(img, want), ctype
    plte = rng.integers(0, 256, (5, 3), dtype=np.uint8)
    trns = np.array([0, 128, 200], dtype=np.uint8)
    idx = rng.integers(0, 5, (h, w, 1), dtype=np.uint8)
    (tmp_path / "p.png").write_bytes(_png_bytes(w, h, 3, idx, [0, 2], plte, trns))
    img = io.read_png(str(tmp_path / "p.png"))
    assert np.array_equal(img[..., :3], plte[idx[..., 0]])
    assert np.array_equal(img[..., 3], np.array([0, 128, 200, 255, 255], dtype=np.uint8)[idx[..., 0]])
    a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    io.write_png_rgba8(str(tmp_path / "w.png"), a)
    assert np.array_equal(io.read_png(str(tmp_path / "w.png")), a)
    (tmp_path / "c.ppm").write_bytes(b"P6\n# a comment\n%d %d\n255\n" % (w, h) + a[..., :3].tobytes())
    assert np.array_equal(io.read_image_rgba8(str(tmp_path / "c.ppm"))[..., :3], a[..., :3])
    good = (tmp_path / "w.png").read_bytes()
    for name, data in (("crc", good[:40] + bytes([good[40] ^ 1]) + good[41:]), ("sig", b"GIF89a" + good[6:]), ("cut", good[:30]),
                       ("deep", _png_bytes(w, h, 2, a[..., :3], [0]).replace(b"\x08\x02\x00\x00\x00", b"\x10\x02\x00\x00\x00", 1))):
        (tmp_path / f"{name}.png").write_bytes(data)
        with pytest.raises(ValueError):
            io.read_image_rgba8(str(tmp_path / f"{name}.png"))


def test_textured_obj_mtl_roundtrip(tmp_path):
    """MTL PBR statements (Pr / Pm, map_Kd with -s, pt_albedo_pattern) become the material extension and its albedo image:
    scene_to_obj -> load_obj gives back the uv, the extension tuples and the texels; a hand-written MTL with options
    before the file name, a second texture file (ignored with a warning: one image per scene) and a PPM texture."""
    from pathtracer_amd import image_io as io
    soup = scenes.random_triangle_soup(3, 40)
    v, i = soup.meshes[0]
    v["uv"] = np.random.default_rng(4).random((len(v), 2)).astype(np.float32)
    sc = SceneData([(v, i)], [(0, k, None) for k in range(6)],                 # six materials: every albedo kind, both models
                   [((0.0, 0.0, 0.0), (0.1 + 0.1 * k, 0.5, 0.9 - 0.1 * k, 1.0)) for k in range(6)])
    sc = scenes.with_material_ext(sc, seed=9)
    obj = str(tmp_path / "s.obj")
    scenes.scene_to_obj(sc, obj)
    back = scenes.load_obj(obj)
    assert back.num_triangles == sc.num_triangles and len(back.materials) == len(sc.materials)
    pa, pb = (np.concatenate([x.meshes[m][0]["pos"][x.meshes[m][1]] for m, _, _ in x.instances]) for x in (sc, back))
    ua, ub = (np.concatenate([x.meshes[m][0]["uv"][x.meshes[m][1]] for m, _, _ in x.instances]) for x in (sc, back))
    assert np.array_equal(pa, pb) and np.array_equal(ua, ub)            # corner by corner: positions and uv round-trip exactly
    assert np.array_equal(back.albedo_texture, sc.albedo_texture)
    for (r0, m0, p0, s0, f0), (r1, m1, p1, s1, f1) in zip(sc.materials_ext, back.materials_ext):
        assert (p0, f0) == (p1, f1) and (p0 == 0 or np.float32(s0) == np.float32(s1))
        if f0 & api.PT_MAT_GGX:
            assert np.float32(r0) == np.float32(r1) and np.float32(m0) == np.float32(m1)
    arr = pack_material_ext(back)
    assert len(arr) == len(back.materials) and any(a.albedo_pattern == 3 for a in arr)

    tex = np.random.default_rng(2).integers(0, 256, (4, 6, 3), dtype=np.uint8)
    (tmp_path / "wall.ppm").write_bytes(b"P6 6 4 255\n" + tex.tobytes())
    io.write_png_rgba8(str(tmp_path / "other.png"), np.zeros((2, 2, 4), dtype=np.uint8))
    (tmp_path / "h.mtl").write_text("newmtl a\nKd 0.5 0.5 0.5\nmap_Kd -clamp off -s 4 4 1 -o 0 0 0 wall.ppm\nPr 0.25\n"
                                    "newmtl b\nKd 1 0 0\nmap_Kd other.png\nnewmtl c\nKd 0 1 0\nPm 1\n")
    (tmp_path / "h.obj").write_text("mtllib h.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nusemtl a\nf 1/1 2/2 3/3\n"
                                    "usemtl b\nf 1/1 2/2 3/3\nusemtl c\nf 1/1 2/2 3/3\n")
    with pytest.warns(UserWarning, match="one albedo image per scene"):
        hs = scenes.load_obj(str(tmp_path / "h.obj"))
    assert hs.materials_ext == [(0.25, 0.0, 3, 4.0, api.PT_MAT_GGX), (0.5, 0.0, 0, 1.0, 0), (0.5, 1.0, 0, 1.0, api.PT_MAT_GGX)]
    assert hs.albedo_texture.shape == (4, 6, 4) and np.array_equal(hs.albedo_texture[..., :3], tex)
    assert len(hs.skipped_textures) == 1 and hs.skipped_textures[0].endswith("other.png")
    # the C++ host's importer (host/PathTracerHost.cpp: own MTL parser, PNG / PPM decoders over zlib) reads the same files alike
    import subprocess
    import zlib
    cli = os.path.join(ROOT, "host", "pt_render_cli")
    assert os.path.exists(cli), "build host/pt_render_cli first (__graft_entry__.build())"
    for path, want in ((obj, back), (str(tmp_path / "h.obj"), hs)):
        r = subprocess.run([cli, "--check-obj", path], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        rows = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("ext ")]
        assert len(rows) == len(want.materials_ext)
        for row, (rough, metal, pattern, scale, flags) in zip(rows, want.materials_ext):
            assert (int(row[7]), int(row[11])) == (pattern, flags) and np.float32(row[9]) == np.float32(scale)
            assert np.float32(row[3]) == np.float32(rough) and np.float32(row[5]) == np.float32(metal)
        tex = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("texture ")][0]
        th, tw = want.albedo_texture.shape[:2]
        assert tex[1] == f"{tw}x{th}" and int(tex[3]) == zlib.adler32(want.albedo_texture.tobytes())
        assert int(tex[5]) == len(getattr(want, "skipped_textures", []))
    (tmp_path / "bad.mtl").write_text("newmtl a\nmap_Kd missing.png\n")
    (tmp_path / "bad.obj").write_text("mtllib bad.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl a\nf 1 2 3\n")
    r = subprocess.run([cli, "--check-obj", str(tmp_path / "bad.obj")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "missing.png" in r.stderr
    with pytest.raises(OSError):
        scenes.load_obj(str(tmp_path / "bad.obj"))
    (tmp_path / "n.mtl").write_text("newmtl a\nKd 1 1 1\n")
    (tmp_path / "n.obj").write_text("mtllib n.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl a\nf 1 2 3\n")
    assert scenes.load_obj(str(tmp_path / "n.obj")).materials_ext is None     # no PBR statement: the reference's material model


@pytest.mark.parametrize("face,what", [("f 1 2 9", "out of range"), ("f 0 1 2", "index"), ("f 1 2 -7", "out of range"),
                                       ("f 1/5 2 3", "out of range"), ("f 1//4 2 3", "out of range"), ("f 1 2 x3", "malformed")])
def test_obj_readers_reject_bad_face_indices(tmp_path, face, what):
    """Round-1 review: face indices straight from the file must be range-checked (index 0, beyond the arrays, too
    negative, malformed tokens) in both readers — an error, never an out-of-bounds read or an uncaught exception."""
    import subprocess
    path = tmp_path / "bad.obj"
    path.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\n" + face + "\n")
    with pytest.raises(ValueError):
        scenes.load_obj(str(path))
    cli = os.path.join(ROOT, "host", "pt_render_cli")
    assert os.path.exists(cli), "build host/pt_render_cli first (__graft_entry__.build())"
    r = subprocess.run([cli, "--check-obj", str(path)], capture_output=True, text=True, timeout=30)
    assert r.returncode == 1 and "Failed importing scene" in r.stderr and "bad.obj:6" in r.stderr, r.stderr
    good = tmp_path / "good.obj"
    good.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\nf 1/1/1 2/1/1 -1/-1/-1\n")
    r = subprocess.run([cli, "--check-obj", str(good)], capture_output=True, text=True, timeout=30)
    assert r.returncode == 0 and "triangles 1" in r.stdout
    assert scenes.load_obj(str(good)).num_triangles == 1


def test_atrium_is_exact_and_deterministic():
    a, b = scenes.atrium(1), scenes.atrium(1)
    assert a.num_triangles == 262144 and len(a.materials) == 24
    for (va, ia), (vb, ib) in zip(a.meshes, b.meshes):
        assert np.array_equal(va, vb) and np.array_equal(ia, ib)
    allp = np.concatenate([v["pos"] for v, _ in a.meshes])
    assert allp.min() > -21 and allp.max() < 21 and np.isfinite(allp).all()
    n = np.concatenate([v["normal"] for v, _ in a.meshes])
    np.testing.assert_allclose(np.linalg.norm(n, axis=1), 1.0, rtol=1e-4)
    for e, c in a.materials:
        assert e == (0.0, 0.0, 0.0) and all(0.2 <= x <= 0.8 for x in c[:3])


def test_partition_samples_covers_range_exactly():
    for total in (0, 1, 7, 64, 4096, 4099):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                f, n = partition_samples(total, world, r)
                seen += list(range(f, f + n))
            assert seen == list(range(total))
    assert partition_samples(4096, 8, 3) == (1536, 512)


def _gloo_worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    from pathtracer_amd import distributed as ptd
    from tests import oracle_lib
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    orc = oracle_lib.load()
    osc = orc.scene(scenes.cornell_box())
    W = H = 32
    total = 10
    c = cornell_consts(W, H)
    first, n = ptd.partition_samples(total, world, rank)
    acc, _, _ = osc.render(c, W, H, first, n)                  # this rank's sample block, full frame, SUM target
    t = torch.from_numpy(acc.reshape(-1).copy())
    t2 = t.clone()
    red, cnt = ptd.reduce_sum_to_root(t, n)
    ordered, cnt2 = ptd.gather_ordered_sum_to_root(t2, n)
    if rank == 0:
        np.save(os.path.join(tmp, "reduced.npy"), red.numpy().reshape(H, W, 4))
        np.save(os.path.join(tmp, "ordered.npy"), ordered.numpy().reshape(H, W, 4))
        np.save(os.path.join(tmp, "count.npy"), np.array([cnt, cnt2]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_reduce_equals_single_process(tmp_path, oracle):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_gloo_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    red = np.load(tmp_path / "reduced.npy")
    ordered = np.load(tmp_path / "ordered.npy")
    cnt = np.load(tmp_path / "count.npy")
    assert list(cnt) == [10, 10]
    osc = oracle.scene(scenes.cornell_box())
    c = cornell_consts(32, 32)
    single, _, _ = osc.render(c, 32, 32, 0, 10)
    # fp32 addition is not associative: (s0..s4)+(s5..s9) vs s0..s9 differ in the last bits only
    np.testing.assert_allclose(red, single, rtol=2e-6, atol=1e-6)
    # the ordered gather is exactly the rank-order sum of the two partial targets
    a, _, _ = osc.render(c, 32, 32, 0, 5)
    b, _, _ = osc.render(c, 32, 32, 5, 5)
    assert np.array_equal(ordered, a + b)
    assert np.array_equal(red, a + b)      # with two ranks any reduction order gives a+b


def _oracle_block_sums(osc, c, W, H, total, world, rank):
    import torch
    from pathtracer_amd import distributed as ptd
    rows = []
    for v in ptd.rank_partitions(world, rank):
        first, n = ptd.partition_samples(total, ptd.VIRTUAL_PARTITIONS, v)
        acc, _, _ = osc.render(c, W, H, first, n)
        rows.append(torch.from_numpy(acc.reshape(-1).copy()))
    return torch.stack(rows)


def _gloo_vp_worker(rank, world, port, tmp):
    import torch.distributed as dist
    from pathtracer_amd import distributed as ptd
    from tests import oracle_lib
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    osc = oracle_lib.load().scene(scenes.cornell_box())
    blocks = _oracle_block_sums(osc, cornell_consts(24, 24), 24, 24, 19, world, rank)
    total = ptd.combine_virtual_partitions(blocks)
    if rank == 0:
        np.save(os.path.join(tmp, "vp.npy"), total.numpy())
    else:
        assert total is None
    dist.barrier()
    dist.destroy_process_group()


def test_virtual_partitions_make_the_sum_independent_of_rank_count(tmp_path, oracle):
    """8 fixed virtual partitions combined in a fixed order -> the same bits for 1, 2, 4 or 8 ranks.
    World sizes 1, 4 and 8 are simulated in-process; world size 2 runs under gloo (the per-rank renderer is the oracle)."""
    import torch.multiprocessing as mp
    from pathtracer_amd import distributed as ptd
    osc = oracle.scene(scenes.cornell_box())
    c = cornell_consts(24, 24)
    ref = None
    for world in (1, 4, 8):
        parts = [_oracle_block_sums(osc, c, 24, 24, 19, world, r) for r in range(world)]
        tot = ptd.sum_blocks_in_order(parts).numpy()
        ref = tot if ref is None else ref
        assert np.array_equal(tot, ref), world
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_gloo_vp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert np.array_equal(np.load(tmp_path / "vp.npy"), ref)
    single, _, _ = osc.render(c, 24, 24, 0, 19)                      # one chain of 19 additions: close, not identical
    np.testing.assert_allclose(ref.reshape(24, 24, 4), single, rtol=3e-6, atol=1e-6)
    with pytest.raises(ValueError):
        ptd.rank_partitions(3, 0)


def test_host_mirror_defaults_match_reference_header():
    """PathTracer.h:112-131 and Sky.h:80-83 defaults, checked on the class source without creating a GPU ctx."""
    import inspect
    from pathtracer_amd import pathtracer
    src = inspect.getsource(pathtracer.PathTracer.__init__)
    for line in ("self.myAoDistance = 1.0", "self.myAccumulate = True", "self.myHalfResRender = True",
                 "self.mySampleSky = True", "self.mySkyFallbackIntensity = 100.0", "self.myMaxRecursionDepth = 4",
                 "self.myLightInstanceIdx = 4", "self.myLightStrength = 100.0", "self.myPhongSpecularPower = 10.0"):
        assert line in src, line
    a = SetupEarthAtmosphere()
    assert a.MiePhaseG == np.float32(0.8) and a.AbsorptionDensity0LayerWidth == 25000.0
    assert abs(a.RayleighDensityExpScale + 1 / 8000.0) < 1e-9 and abs(a.MieAbsorption[0] - 4.44e-7) < 1e-9


def test_image_io_roundtrip_and_png_signature(tmp_path):
    from pathtracer_amd import image_io
    rng = np.random.default_rng(3)
    img = rng.uniform(0, 5, (7, 9, 4)).astype(np.float32)
    image_io.write_pfm(tmp_path / "a.pfm", img)
    assert np.array_equal(image_io.read_pfm(tmp_path / "a.pfm"), img[..., :3])
    ldr = image_io.tonemap_reinhard(img)
    assert np.allclose(ldr, img[..., :3] / (img[..., :3] + 1.0)) and ldr.max() < 1.0
    image_io.write_png(tmp_path / "a.png", ldr)
    data = open(tmp_path / "a.png", "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n" and data[12:16] == b"IHDR" and data[-8:-4] == b"IEND"
    import struct, zlib
    w, h = struct.unpack(">II", data[16:24])
    assert (w, h) == (9, 7)
    i = data.index(b"IDAT")
    n = struct.unpack(">I", data[i - 4:i])[0]
    raw = zlib.decompress(data[i + 4:i + 4 + n])
    assert len(raw) == h * (1 + 3 * w)


# ---------------------------------------------------------------------------------------------- accumulation control
class _RecorderCtx:
    """Stands in for context.Context: records the C-ABI calls the host mirror would make (no GPU needed)."""
    def __init__(self):
        self.calls = []
    def accum_reset(self, w, h): self.calls.append(("accum_reset", w, h))
    def sky_set(self, atm): self.calls.append(("sky_set",))
    def scene_create(self, scene, flags=0): self.calls.append(("scene_create",))
    def render(self, consts, w, h, first, n): self.calls.append(("render", w, h, first, n))
    def render_ao(self, consts, w, h, first, n): self.calls.append(("render_ao", w, h, first, n))


def test_update_restarts_accumulation_like_the_reference():
    """PathTracer::Update (PathTracer.cpp:557-571) + CameraHasChanged (:446-453, epsilon 1e-4 on the view-projection
    matrix) + the sky-settings check (:566): accumulation restarts exactly when the reference's would, and the frame
    counter / clear flag drive pt_accum_reset and the first_sample argument of pt_render as RenderRT (:668-678,763-764) does."""
    from pathtracer_amd.pathtracer import PathTracer
    rec = _RecorderCtx()
    pt = PathTracer(0, 1280, 720, ctx=rec)
    pt.LoadScene(scenes.cornell_box(), scenes.CORNELL_CAM_POS)
    def frame(n=1):
        pt.Update()
        pt.RenderRT(n)
    rec.calls.clear()
    frame(); frame(); frame(2)
    renders = [c for c in rec.calls if c[0] == "render"]
    assert [c[3:] for c in renders] == [(0, 1), (1, 1), (2, 2)]            # frames accumulate: first_sample = frames so far
    assert sum(c[0] == "accum_reset" for c in rec.calls) == 1              # LoadScene asked for one clear (camera moved too)
    assert pt.myNumAccumulationFrames == 4 and not pt.CameraHasChanged()
    # a move below the epsilon keeps accumulating; above it restarts
    rec.calls.clear()
    pt.myCamera.myPosition = pt.myCamera.myPosition + np.float32(1e-6)
    assert not pt.CameraHasChanged()
    frame()
    assert [c for c in rec.calls if c[0] == "accum_reset"] == [] and rec.calls[-1][3:] == (4, 1)
    pt.myCamera.myPosition = pt.myCamera.myPosition + np.array([0.5, 0.0, 0.0], dtype=np.float32)
    assert pt.CameraHasChanged()
    frame()
    assert rec.calls[-2][0] == "accum_reset" and rec.calls[-1][3:] == (0, 1) and pt.myNumAccumulationFrames == 1
    assert not pt.CameraHasChanged()                                        # myLastViewMat was refreshed by Update (:570)
    # rotating the camera counts as well
    pt.myCamera.LookAt(pt.myCamera.myPosition, pt.myCamera.myPosition + np.array([0.3, 0.0, 1.0], dtype=np.float32))
    frame()
    assert rec.calls[-2][0] == "accum_reset" and rec.calls[-1][3:] == (0, 1)
    # sky settings: a new sun direction recomputes the LUT and restarts (Sky_Imgui::HaveSettingsChanged)
    rec.calls.clear()
    pt.mySky.mySunDir = np.array([0.0, 0.6, 0.8], dtype=np.float32)
    frame()
    assert [c[0] for c in rec.calls] == ["accum_reset", "sky_set", "render"] and rec.calls[-1][3:] == (0, 1)
    # myAccumulate = false: always frame 0 (:562-564)
    pt.myAccumulate = False
    rec.calls.clear()
    frame(); frame()
    assert [c[3:] for c in rec.calls if c[0] == "render"] == [(0, 1), (0, 1)] and sum(c[0] == "accum_reset" for c in rec.calls) == 2
    # the AO toggle selects the other pipeline for the same dispatch (:680-683)
    pt.myRenderAo = True
    frame()
    assert rec.calls[-1][0] == "render_ao"


# ---------------------------------------------------------------------------------------------- BVH8 builder on the CPU
def _flat_triangles(sc):
    """Triangle list in primitive order for scenes whose instances carry no transform (InitRtScene order: instance by
    instance, PathTracer.cpp:171-339)."""
    out = []
    for mesh, _, xf in sc.instances:
        assert xf is None
        v, idx = sc.meshes[mesh]
        out.append(v["pos"][idx].reshape(-1, 9))
    return np.ascontiguousarray(np.concatenate(out), dtype=np.float32)


@pytest.mark.parametrize("scene", ["atrium", "soup+duplicates"])
def test_bvh8_builder_both_leaf_formats_walk_to_the_oracle_hits(oracle, scene):
    """The product's host-side builder (csrc/bvh_builder.cpp) run on the CPU, walked by tests/walker: 48-byte triangle
    records and 64-byte edge-sharing pairs must give the oracle's hit records bit for bit (the pair test shears the
    shared vertices once, which must not change a bit), and the paired hierarchy is the smaller one."""
    so = os.path.join(ROOT, "tests", "walker", "libbvh8_walker.so")
    assert os.path.exists(so), "build tests/walker first (__graft_entry__.build())"
    wl = C.CDLL(so)
    P = C.c_void_p
    wl.bvh8_build_and_walk.restype = C.c_int
    wl.bvh8_build_and_walk.argtypes = [P, C.c_uint32, C.c_int, P, C.c_uint64, P, P, P, P]
    if scene == "atrium":
        sc = scenes.atrium(1)
        lo, hi, n_rays = (-20, 0, -10), (20, 16, 10), 30000
    else:
        sc = scenes.random_triangle_soup(5, 4000)
        v, idx = sc.meshes[0]
        # the same triangle three times (every edge shared by three records) and a sliver with two equal vertices
        extra = np.zeros(12, dtype=api.VERTEX_DTYPE)
        extra["pos"] = [[3, 1, 3], [4, 1, 3], [3, 1, 4]] * 3 + [[5, 5, 5], [5, 5, 5], [6, 5, 5]]
        extra["normal"] = [0, 1, 0]
        sc = SceneData([(np.concatenate([v, extra]), np.arange(len(v) + 12, dtype=np.uint32))], [(0, 0, None)], sc.materials)
        lo, hi, n_rays = (0, 0, 0), (10, 10, 10), 30000
    tris = _flat_triangles(sc)
    from tests.common import random_rays
    rays = random_rays(9, n_rays, lo, hi, tmin=0.001, tmax=1e4)
    want = oracle.scene(sc).trace_rays(rays)
    stats = {}
    for mode in (0, 1):
        hits = np.empty(len(rays), dtype=api.HIT_DTYPE)
        cn, ct, info = np.zeros(1, np.uint64), np.zeros(1, np.uint64), np.zeros(5, np.uint64)
        assert wl.bvh8_build_and_walk(api.ptr(tris), len(tris), mode, api.ptr(rays), len(rays), api.ptr(hits), api.ptr(cn), api.ptr(ct),
                                      api.ptr(info)) == 0
        assert np.array_equal(hits.view(np.uint32), want.view(np.uint32)), mode
        stats[mode] = (int(info[0]), int(info[1]), int(info[2]), int(info[3]), int(cn[0]), int(ct[0]))
    # spatial splits (SBVH) may reference a triangle from more than one leaf: records >= triangles, within the split budget
    assert len(tris) <= stats[0][1] <= 1.3 * len(tris) + 1 and stats[0][2:4] == (0, 48) and stats[1][3] == 64
    assert len(tris) <= stats[1][1] + stats[1][2] <= 1.3 * len(tris) + 1     # every triangle is in at least one record
    if scene == "atrium":
        assert stats[1][2] > 0.45 * len(tris) and stats[1][0] < 0.8 * stats[0][0]     # quads pair up; fewer nodes
    else:
        assert 1 <= stats[1][2] <= 8                              # a random soup shares no edges; the duplicates do


def test_insertion_based_tree_optimisation_keeps_hits_and_pays_on_clutter(oracle):
    """csrc/bvh_builder.cpp optimises the finished binary tree by taking its largest nodes out and putting their subtrees back
    where they add the least area (two forms), and keeps the result only if the 8-wide tree it collapses to is better by the
    collapse's own SAH cost. Hit records cannot depend on it; on overlapping, randomly rotated clutter (what a top-down
    builder is worst at) it must lower both that cost and the node visits of incoherent rays."""
    so = os.path.join(ROOT, "tests", "walker", "libbvh8_walker.so")
    wl = C.CDLL(so)
    P = C.c_void_p
    wl.bvh8_build.restype = P
    wl.bvh8_build.argtypes = [P, C.c_uint32, C.c_int, C.c_int, P]
    wl.bvh8_free.argtypes = [P]
    wl.bvh8_walk_built.restype = C.c_int
    wl.bvh8_walk_built.argtypes = [P, P, C.c_uint64, P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]
    rng = np.random.default_rng(12)
    # 600 long thin boxes (beams) with random orientations through one another, 12 triangles each
    tris = []
    corners = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], dtype=np.float64)
    faces = [(0, 1, 3), (0, 3, 2), (4, 6, 7), (4, 7, 5), (0, 4, 5), (0, 5, 1), (2, 3, 7), (2, 7, 6), (0, 2, 6), (0, 6, 4), (1, 5, 7), (1, 7, 3)]
    for _ in range(600):
        q, _r = np.linalg.qr(rng.normal(size=(3, 3)))
        c = (corners * np.array([rng.uniform(2, 9), 0.08, 0.12])) @ q.T + rng.uniform(-10, 10, size=3)
        tris += [np.concatenate([c[a], c[b], c[d]]) for a, b, d in faces]
    tris = np.ascontiguousarray(np.array(tris), dtype=np.float32)
    from tests.common import random_rays
    rays = random_rays(3, 40000, (-12, -12, -12), (12, 12, 12), tmin=0.001, tmax=1e4)

    def build_and_walk(passes):
        old = os.environ.get("PT_BVH_REINSERT_PASSES")
        if passes is None:
            os.environ.pop("PT_BVH_REINSERT_PASSES", None)
        else:
            os.environ["PT_BVH_REINSERT_PASSES"] = str(passes)
        try:
            info = (C.c_uint64 * 8)()
            h = wl.bvh8_build(api.ptr(tris), len(tris), 0, 4, info)
        finally:
            if old is None:
                os.environ.pop("PT_BVH_REINSERT_PASSES", None)
            else:
                os.environ["PT_BVH_REINSERT_PASSES"] = old
        hits = np.empty(len(rays), dtype=api.HIT_DTYPE)
        cn, ct = C.c_uint64(), C.c_uint64()
        assert wl.bvh8_walk_built(h, api.ptr(rays), len(rays), api.ptr(hits), C.byref(cn), C.byref(ct), 4) == 0
        wl.bvh8_free(h)
        sah = float(np.array([info[6]], dtype=np.uint64).view(np.float64)[0])
        return hits, cn.value, ct.value, sah
    h0, n0, t0, s0 = build_and_walk(0)
    h1, n1, t1, s1 = build_and_walk(None)        # the shipped default
    h2, n2, t2, s2 = build_and_walk(None)        # and again: the build is deterministic
    assert np.array_equal(h0.view(np.uint32), h1.view(np.uint32))
    assert (h0["prim"] != api.PT_MISS).mean() > 0.2
    assert (n1, t1, s1) == (n2, t2, s2) and np.array_equal(h1.view(np.uint32), h2.view(np.uint32))
    assert s1 < 0.99 * s0, (s0, s1)              # kept only when better by 1 % ...
    assert n1 < 0.97 * n0, (n0, n1)              # ... and the rays agree
    assert 80 * n1 + 48 * t1 < 80 * n0 + 48 * t0


def test_roofline_pmc_entry_is_tied_to_the_build(tmp_path, monkeypatch):
    """bench.py prices k_extend's launch time against committed PMC counters only when they were measured on THIS build of
    the kernels: every entry of profiles/<round>/pmc_traffic.json carries the fingerprint of the sources it ran on."""
    import json
    import bench
    fp = bench.build_fingerprint()
    assert len(fp) == 16 and fp == bench.build_fingerprint()
    # the committed round-3 entry of the driver's default command belongs to the build in the tree
    entry, stale = bench.find_pmc("atrium", 1920, 1080, 64, 4, 0, 0, fp)
    assert entry is not None and entry["round"] == "r03"
    assert not stale, "profiles/r03/pmc_traffic.json was measured on another build: run tools/run_final_refresh.sh r03"
    entry, stale = bench.find_pmc("atrium", 1920, 1080, 64, 4, 0, 0, "0" * 16)
    assert entry is not None and stale
    assert bench.find_pmc("atrium", 1921, 1080, 64, 4, 0, 0, fp) == (None, False)     # another size: no entry at all
    # a changed source changes the fingerprint
    root = tmp_path / "r"
    (root / "pathtracer_amd" / "csrc").mkdir(parents=True)
    (root / "include").mkdir()
    (root / "pathtracer_amd" / "csrc" / "kernels.hip").write_text("a")
    (root / "include" / "pt_api.h").write_text("b")
    monkeypatch.setattr(bench, "ROOT", str(root))
    f1 = bench.build_fingerprint()
    (root / "pathtracer_amd" / "csrc" / "kernels.hip").write_text("a ")
    f2 = bench.build_fingerprint()
    assert f1 != f2
    # entries without a fingerprint (round 2's) are always stale; newest round wins
    (root / "profiles" / "r02").mkdir(parents=True)
    e = {"workload": "atrium", "width": 8, "height": 8, "spp_per_step": 1, "depth": 4}
    (root / "profiles" / "r02" / "pmc_traffic.json").write_text(json.dumps([e]))
    entry, stale = bench.find_pmc("atrium", 8, 8, 1, 4, 0, 0, f2)
    assert entry["round"] == "r02" and stale
    (root / "profiles" / "r03").mkdir()
    (root / "profiles" / "r03" / "pmc_traffic.json").write_text(json.dumps([dict(e, build_fingerprint=f2)]))
    entry, stale = bench.find_pmc("atrium", 8, 8, 1, 4, 0, 0, f2)
    assert entry["round"] == "r03" and not stale
