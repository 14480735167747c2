"""Scene ingestion for the hot path: the stand-in for FANCY's MeshImporter (PathTracer/PathTracer.cpp:115-132).

* load_obj      minimal Wavefront OBJ/MTL reader producing the importer's output contract: per (group, material)
                one mesh with interleaved {pos f3, normal f3, uv f2} vertices (PathTracer.cpp:116-120), one instance
                per mesh with identity transform, materials {emission, colour}.
* cornell_box   the reference's default scene (resources/models/CornellBox.obj) from a committed data fixture.
* atrium / street / courtyard
                deterministic procedural stand-ins for Sponza / Bistro / San Miguel (BASELINE.json configs 2-5): the
                real assets are neither in the reference (.gitignore:45-46) nor available offline. Triangle counts are
                exact; all randomness is an integer hash of the seed.
"""
import json
import os

import numpy as np

from . import api
from .context import SceneData

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


# ----------------------------------------------------------------------------------------------- OBJ / MTL
def _parse_mtl(path):
    """Kd / Ke (the reference's two material inputs, PathTracer.cpp:142-160) and, for the material extension, the PBR
    statements of MTL: `Pr` roughness and `Pm` metallic (either one selects the GGX model), `map_Kd [-s su sv sw] file`
    (base-colour image; -s = uv scale) and the non-standard `pt_albedo_pattern <id> <scale>` for the procedural patterns."""
    mats, cur = {}, None
    if not os.path.exists(path):
        return mats
    with open(path) as f:
        for lineno, line in enumerate(f, 1):
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "newmtl":
                cur = t[1]
                mats[cur] = {"Kd": (1.0, 1.0, 1.0), "Ke": (0.0, 0.0, 0.0)}
            elif cur and t[0] in ("Kd", "Ke") and len(t) >= 4:
                mats[cur][t[0]] = tuple(float(x) for x in t[1:4])
            elif cur and t[0] in ("Pr", "Pm") and len(t) >= 2:
                mats[cur][t[0]] = float(t[1])
            elif cur and t[0] == "pt_albedo_pattern" and len(t) >= 3:
                mats[cur]["pattern"] = (int(t[1]), float(t[2]))
            elif cur and t[0] == "map_Kd" and len(t) >= 2:
                scale, k, name = 1.0, 1, None
                while k < len(t):                       # options: "-s u v w", "-o …", "-clamp on", …; the file name comes last
                    if t[k].startswith("-") and len(t[k]) > 1 and not t[k][1].isdigit() and k < len(t) - 1:
                        opt, k = t[k], k + 1
                        vals = []
                        while k < len(t) - 1 and (t[k] in ("on", "off") or _is_number(t[k])):
                            vals.append(t[k])
                            k += 1
                        if opt == "-s" and vals and _is_number(vals[0]):
                            scale = float(vals[0])
                    else:
                        name = " ".join(t[k:])
                        break
                if name is None:
                    raise ValueError(f"{path}:{lineno}: map_Kd without a file name")
                mats[cur]["map_Kd"] = (os.path.join(os.path.dirname(path), name.replace("\\", "/")), scale)
    return mats


def _is_number(tok):
    try:
        float(tok)
        return True
    except ValueError:
        return False


def load_obj(path):
    """One mesh per (o/g group, usemtl) run in file order — the order Assimp's OBJ importer yields and the reference
    relies on (the Cornell light must be instance 4, PathTracer.h:125)."""
    pos, nrm, tex = [], [], []
    groups = []  # (name, material, [faces]) ; face = list of (vi, ti, ni)
    mtl = {}
    cur = None
    cur_name, cur_mat = "default", None
    def resolve(tok, count, what, lineno, required=False):
        """1-based / negative-relative OBJ index -> 0-based, range-checked (-1 = absent)."""
        if not tok:
            if required:
                raise ValueError(f"{path}:{lineno}: face corner without a position index")
            return -1
        try:
            i = int(tok)
        except ValueError:
            raise ValueError(f"{path}:{lineno}: malformed {what} index {tok!r}") from None
        j = i - 1 if i > 0 else count + i
        if i == 0 or not 0 <= j < count:
            raise ValueError(f"{path}:{lineno}: {what} index {i} out of range (have {count})")
        return j

    with open(path) as f:
        for lineno, line in enumerate(f, 1):
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "v":
                pos.append([float(x) for x in t[1:4]])
            elif t[0] == "vn":
                nrm.append([float(x) for x in t[1:4]])
            elif t[0] == "vt":
                tex.append([float(x) for x in t[1:3]])
            elif t[0] == "mtllib":
                mtl.update(_parse_mtl(os.path.join(os.path.dirname(path), t[1])))
            elif t[0] in ("o", "g"):
                cur_name = t[1] if len(t) > 1 else "default"
                cur = None
            elif t[0] == "usemtl":
                cur_mat = t[1]
                cur = None
            elif t[0] == "f":
                if cur is None:
                    cur = (cur_name, cur_mat, [])
                    groups.append(cur)
                corners = []
                for c in t[1:]:
                    p = (c.split("/") + ["", ""])[:3]
                    corners.append((resolve(p[0], len(pos), "position", lineno, required=True),
                                    resolve(p[1], len(tex), "texture-coordinate", lineno),
                                    resolve(p[2], len(nrm), "normal", lineno)))
                for k in range(1, len(corners) - 1):  # fan triangulation
                    cur[2].append((corners[0], corners[k], corners[k + 1]))
    mat_names = list(mtl.keys())
    for _, m, _ in groups:
        if m is not None and m not in mat_names:
            mat_names.append(m)
            mtl[m] = {"Kd": (1.0, 1.0, 1.0), "Ke": (0.0, 0.0, 0.0)}
    if not mat_names:
        mat_names = ["default"]
        mtl["default"] = {"Kd": (1.0, 1.0, 1.0), "Ke": (0.0, 0.0, 0.0)}
    meshes, instances = [], []
    for name, m, faces in groups:
        vmap, verts, idx = {}, [], []
        for tri in faces:
            for key in tri:
                if key not in vmap:
                    vmap[key] = len(verts)
                    p = pos[key[0]]
                    uv = tex[key[1]] if key[1] >= 0 else [0.0, 0.0]
                    n = nrm[key[2]] if key[2] >= 0 else None
                    verts.append((p, n, uv, tri))
                idx.append(vmap[key])
        v = np.zeros(len(verts), dtype=api.VERTEX_DTYPE)
        for i, (p, n, uv, tri) in enumerate(verts):
            if n is None:  # no normal in the file: face normal
                a, b, c = (np.array(pos[k[0]]) for k in tri)
                n = np.cross(b - a, c - a)
                n = n / max(np.linalg.norm(n), 1e-30)
            v[i] = (p, n, uv)
        meshes.append((v, np.array(idx, dtype=np.uint32)))
        instances.append((len(meshes) - 1, mat_names.index(m) if m in mat_names else 0, None))
    materials = [(mtl[n]["Ke"], tuple(mtl[n]["Kd"]) + (1.0,)) for n in mat_names]
    scene = SceneData(meshes, instances, materials, name=os.path.basename(path))
    _attach_material_ext(scene, [mtl[n] for n in mat_names])
    return scene


def _attach_material_ext(scene, mats):
    """MTL PBR statements -> SceneData.materials_ext / albedo_texture (the material extension; include/pt_api.h
    PtMaterialExt). The device holds ONE albedo image per scene (pt_scene_set_albedo_texture): the first map_Kd file in
    material order is decoded (image_io.read_image_rgba8: 8-bit PNG, binary PPM) and every material naming it samples it;
    materials naming another file keep their plain Kd and are listed in scene.skipped_textures (with a warning)."""
    if not any(k in m for m in mats for k in ("Pr", "Pm", "map_Kd", "pattern")):
        return
    from . import image_io
    tex_path, skipped, ext = None, [], []
    for m in mats:
        pattern, scale = m.get("pattern", (0, 1.0))
        if "map_Kd" in m:
            f, sc = m["map_Kd"]
            if tex_path is None:
                tex_path = f
            if os.path.normpath(f) == os.path.normpath(tex_path):
                pattern, scale = 3, sc
            else:
                skipped.append(f)
        ggx = "Pr" in m or "Pm" in m
        ext.append((float(m.get("Pr", 0.5)), float(m.get("Pm", 0.0)), int(pattern), float(scale), api.PT_MAT_GGX if ggx else 0))
    scene.materials_ext = ext
    scene.skipped_textures = skipped
    if tex_path is not None:
        scene.albedo_texture = image_io.read_image_rgba8(tex_path)
    if skipped:
        import warnings
        warnings.warn(f"{scene.name}: one albedo image per scene; {len(set(skipped))} further map_Kd file(s) ignored "
                      f"(those materials keep their Kd): {sorted(set(skipped))[:3]}")


def scene_to_obj(scene, path):
    """Write a SceneData as OBJ + MTL (one group per instance, %.9g so fp32 values round-trip exactly). A material
    extension goes out as Pr / Pm (GGX materials), pt_albedo_pattern (procedural patterns) and map_Kd -s (the albedo
    image, written beside the MTL as <name>_albedo.png), which load_obj reads back."""
    mtl_path = os.path.splitext(path)[0] + ".mtl"
    ext = getattr(scene, "materials_ext", None)
    tex_name = None
    if ext and getattr(scene, "albedo_texture", None) is not None and any(e[2] == 3 for e in ext):
        from . import image_io
        tex_name = os.path.basename(os.path.splitext(path)[0]) + "_albedo.png"
        image_io.write_png_rgba8(os.path.join(os.path.dirname(path), tex_name), scene.albedo_texture)
    with open(mtl_path, "w") as f:
        for k, (e, c) in enumerate(scene.materials):
            f.write(f"newmtl m{k}\nKd {c[0]:.9g} {c[1]:.9g} {c[2]:.9g}\nKe {e[0]:.9g} {e[1]:.9g} {e[2]:.9g}\n")
            if ext:
                rough, metal, pattern, scale, flags = ext[k]
                if flags & api.PT_MAT_GGX:
                    f.write(f"Pr {rough:.9g}\nPm {metal:.9g}\n")
                if pattern == 3 and tex_name:
                    f.write(f"map_Kd -s {scale:.9g} {scale:.9g} 1 {tex_name}\n")
                elif pattern:
                    f.write(f"pt_albedo_pattern {int(pattern)} {scale:.9g}\n")
    with open(path, "w") as f:
        f.write(f"mtllib {os.path.basename(mtl_path)}\n")
        off = 0
        for gi, (m, mat, t) in enumerate(scene.instances):
            assert t is None, "OBJ export supports identity instances only"
            v, idx = scene.meshes[m]
            f.write(f"o g{gi}\n")
            for p in v["pos"]:
                f.write(f"v {p[0]:.9g} {p[1]:.9g} {p[2]:.9g}\n")
            for n in v["normal"]:
                f.write(f"vn {n[0]:.9g} {n[1]:.9g} {n[2]:.9g}\n")
            for u in v["uv"]:
                f.write(f"vt {u[0]:.9g} {u[1]:.9g}\n")
            f.write(f"usemtl m{mat}\n")
            for a, b, c in idx.reshape(-1, 3) + 1 + off:
                f.write(f"f {a}/{a}/{a} {b}/{b}/{b} {c}/{c}/{c}\n")
            off += len(v)


def scene_to_json(scene, path):
    d = {"name": scene.name,
         "meshes": [{"pos": v["pos"].tolist(), "normal": v["normal"].tolist(), "uv": v["uv"].tolist(),
                     "indices": i.tolist()} for v, i in scene.meshes],
         "instances": [[m, mat, None if t is None else t.tolist()] for m, mat, t in scene.instances],
         "materials": [[list(e), list(c)] for e, c in scene.materials]}
    with open(path, "w") as f:
        json.dump(d, f, separators=(",", ":"))


def scene_from_json(path):
    with open(path) as f:
        d = json.load(f)
    meshes = []
    for m in d["meshes"]:
        v = np.zeros(len(m["pos"]), dtype=api.VERTEX_DTYPE)
        v["pos"] = np.array(m["pos"], dtype=np.float32).reshape(-1, 3)
        v["normal"] = np.array(m["normal"], dtype=np.float32).reshape(-1, 3)
        v["uv"] = np.array(m["uv"], dtype=np.float32).reshape(-1, 2)
        meshes.append((v, np.array(m["indices"], dtype=np.uint32)))
    return SceneData(meshes, [tuple(i) for i in d["instances"]], [tuple(m) for m in d["materials"]], name=d.get("name", ""))


def cornell_box():
    """resources/models/CornellBox.obj as data (36 triangles, 6 mesh groups, light = instance 4).
    Camera of the reference: position (1,102,-30), identity orientation (PathTracer.cpp:40,134-136)."""
    return scene_from_json(os.path.join(_ASSETS, "cornell_box.json"))


CORNELL_CAM_POS = (1.0, 102.0, -30.0)


# ----------------------------------------------------------------------------------------------- procedural helpers
def _hash_u32(x):
    x = np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
    x = (x ^ (x >> 16)) * 0x7FEB352D & 0xFFFFFFFF
    x = (x ^ (x >> 15)) * 0x846CA68B & 0xFFFFFFFF
    return (x ^ (x >> 16)).astype(np.uint64)


def _rand01(seed, *idx):
    h = np.uint64(seed) * np.uint64(0x9E3779B1) & np.uint64(0xFFFFFFFF)
    for i in idx:
        h = _hash_u32(h ^ (np.asarray(i, dtype=np.uint64) * np.uint64(0x85EBCA6B) & np.uint64(0xFFFFFFFF)))
    return (h >> np.uint64(8)).astype(np.float64) / float(1 << 24)


def _value_noise(seed, u, v, freq):
    x, y = u * freq, v * freq
    x0, y0 = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)
    fx, fy = x - x0, y - y0
    sx, sy = fx * fx * (3 - 2 * fx), fy * fy * (3 - 2 * fy)
    def c(ix, iy):
        return _rand01(seed, ix & 0xFFFF, iy & 0xFFFF)
    top = c(x0, y0) * (1 - sx) + c(x0 + 1, y0) * sx
    bot = c(x0, y0 + 1) * (1 - sx) + c(x0 + 1, y0 + 1) * sx
    return top * (1 - sy) + bot * sy - 0.5


def _grid_mesh(fn, nu, nv):
    """Mesh of nu x nv quads over (u,v) in [0,1]^2; fn(u,v) -> (..., 3). Normals from the grid's own tangents."""
    u, v = np.meshgrid(np.linspace(0.0, 1.0, nu + 1), np.linspace(0.0, 1.0, nv + 1), indexing="xy")
    P = fn(u, v)                                     # (nv+1, nu+1, 3)
    du = np.gradient(P, axis=1)
    dv = np.gradient(P, axis=0)
    N = np.cross(du, dv)
    N /= np.maximum(np.linalg.norm(N, axis=-1, keepdims=True), 1e-30)
    verts = np.zeros((nv + 1) * (nu + 1), dtype=api.VERTEX_DTYPE)
    verts["pos"] = P.reshape(-1, 3).astype(np.float32)
    verts["normal"] = N.reshape(-1, 3).astype(np.float32)
    verts["uv"] = np.stack([u, v], axis=-1).reshape(-1, 2).astype(np.float32)
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="xy")
    a = (j * (nu + 1) + i).reshape(-1)
    b, c, d = a + 1, a + (nu + 1), a + (nu + 1) + 1
    idx = np.stack([a, b, d, a, d, c], axis=-1).reshape(-1).astype(np.uint32)
    return verts, idx


def _merge(parts):
    vs, is_, off = [], [], 0
    for v, i in parts:
        vs.append(v)
        is_.append(i + np.uint32(off))
        off += len(v)
    return np.concatenate(vs), np.concatenate(is_)


def _palette(seed, n):
    """n untextured materials: albedo per channel uniform in [0.2,0.8] (quantised to unorm8 at upload), no emitters."""
    mats = []
    for k in range(n):
        rgb = [0.2 + 0.6 * float(_rand01(seed, 1000 + k, c)) for c in range(3)]
        mats.append(((0.0, 0.0, 0.0), (rgb[0], rgb[1], rgb[2], 1.0)))
    return mats


def _cylinder(cx, cz, y0, y1, r, seg, rings, seed):
    def fn(u, v):
        ang = 2 * np.pi * u
        rr = r * (1.0 + 0.06 * np.cos(12 * ang) + 0.04 * _value_noise(seed, u * 4, v * 4, 8.0))   # fluted shaft
        return np.stack([cx + rr * np.cos(ang), y0 + (y1 - y0) * v, cz + rr * np.sin(ang)], axis=-1)
    return _grid_mesh(fn, seg, rings)


def _arch(x0, x1, y, z, thick, depth, seg):
    """Half-ring with rectangular cross-section spanning x0..x1, springing at height y."""
    cx, R = 0.5 * (x0 + x1), 0.5 * (x1 - x0)
    parts = []
    for (ra, rb, za, zb) in ((R - thick, R - thick, z - depth, z + depth), (R, R, z - depth, z + depth),
                             (R - thick, R, z - depth, z - depth), (R - thick, R, z + depth, z + depth)):
        def fn(u, v, ra=ra, rb=rb, za=za, zb=zb):
            ang = np.pi * u
            rr = ra + (rb - ra) * v
            return np.stack([cx - rr * np.cos(ang), y + rr * np.sin(ang), za + (zb - za) * v], axis=-1)
        parts.append(_grid_mesh(fn, seg, 1))
    return _merge(parts)


def atrium(seed=1):
    """"Sponza-class" stand-in (BASELINE.json config 2/3): open-top courtyard 40 x 16 x 20 m, two storeys of 12 x 4
    fluted columns with arches, gallery slabs, displaced floor and walls, hanging cloth. Exactly 262,144 triangles,
    24 materials, no emitters (lit by the sky through the open top)."""
    meshes, instances = [], []
    def add(mesh, mat):
        meshes.append(mesh)
        instances.append((len(meshes) - 1, mat % 24, None))

    amp = 0.02
    # floor: 232 x 96 quads
    add(_grid_mesh(lambda u, v: np.stack([-20 + 40 * u, amp * _value_noise(seed, u, v, 64.0) * 2, -10 + 20 * v], axis=-1), 232, 96), 0)
    # long walls (z = -10, +10): 160 x 64 quads each; short walls (x = -20, +20): 80 x 64
    for k, zz in enumerate((-10.0, 10.0)):
        add(_grid_mesh(lambda u, v, zz=zz, k=k: np.stack([-20 + 40 * u, 16 * v, zz + amp * 2 * _value_noise(seed + 10 + k, u, v, 48.0) + 0 * u], axis=-1), 160, 64), 1 + k)
    for k, xx in enumerate((-20.0, 20.0)):
        add(_grid_mesh(lambda u, v, xx=xx, k=k: np.stack([xx + amp * 2 * _value_noise(seed + 20 + k, u, v, 48.0) + 0 * u, 16 * v, -10 + 20 * u], axis=-1), 80, 64), 3 + k)
    # gallery slabs at y = 8 along both long walls: top and bottom faces, 128 x 16 quads each
    for k, (za, zb) in enumerate(((-10.0, -5.0), (5.0, 10.0))):
        top = _grid_mesh(lambda u, v, za=za, zb=zb: np.stack([-20 + 40 * u, 8.0 + amp * _value_noise(seed + 30 + k, u, v, 64.0), za + (zb - za) * v], axis=-1), 128, 16)
        bot = _grid_mesh(lambda u, v, za=za, zb=zb: np.stack([-20 + 40 * u, 7.6 + 0 * u, za + (zb - za) * v], axis=-1), 128, 16)
        add(_merge([top, bot]), 5 + k)
    # columns: 2 storeys x 12 x 4, 32 segments x 16 rings
    xs = np.linspace(-17.5, 17.5, 12)
    zs = (-8.0, -5.0, 5.0, 8.0)
    for s, (y0, y1) in enumerate(((0.0, 6.2), (8.0, 14.2))):
        for iz, z in enumerate(zs):
            for ix, x in enumerate(xs):
                add(_cylinder(x, z, y0, y1, 0.35, 32, 16, seed + 100 + s * 48 + iz * 12 + ix), 7 + (ix + iz + s) % 5)
    # arches between neighbouring columns of every row: 24 segments
    for s, y in enumerate((6.2, 14.2)):
        for iz, z in enumerate(zs):
            for ix in range(11):
                add(_arch(xs[ix], xs[ix + 1], y, z, 0.35, 0.3, 24), 12 + (ix + s) % 4)
    # hanging cloth: 8 curved sheets, 48 x 32 quads
    for k in range(8):
        x0 = -16.0 + 4.5 * k
        def fn(u, v, x0=x0, k=k):
            sag = 1.2 * np.sin(np.pi * u) * (0.6 + 0.4 * np.cos(2 * np.pi * v))
            return np.stack([x0 + 2.5 * v + 0.15 * _value_noise(seed + 200 + k, u, v, 6.0), 13.5 - sag, -4.5 + 9.0 * u], axis=-1)
        add(_grid_mesh(fn, 48, 32), 16 + k)
    scene = SceneData(meshes, instances, _palette(seed, 24), name=f"atrium(seed={seed})")
    assert scene.num_triangles == 262144, scene.num_triangles
    return scene


ATRIUM_CAM = {"eye": (-17.0, 1.7, 0.6), "target": (10.0, 5.5, -0.8)}


def _building_block(seed, x0, x1, z0, z1, height, nu, nv, amp=0.08):
    """Four displaced facade grids + a roof."""
    parts = []
    faces = (((x0, z0), (x1, z0)), ((x1, z0), (x1, z1)), ((x1, z1), (x0, z1)), ((x0, z1), (x0, z0)))
    for k, ((ax, az), (bx, bz)) in enumerate(faces):
        nx, nz = (bz - az), -(bx - ax)
        ln = max(np.hypot(nx, nz), 1e-9)
        nx, nz = nx / ln, nz / ln
        def fn(u, v, ax=ax, az=az, bx=bx, bz=bz, nx=nx, nz=nz, k=k):
            # window-like relief: periodic recess + noise
            relief = amp * (np.sign(np.sin(2 * np.pi * 6 * u) * np.sin(2 * np.pi * 5 * v)) * 0.5 + _value_noise(seed + k, u, v, 24.0))
            return np.stack([ax + (bx - ax) * u + nx * relief, height * v, az + (bz - az) * u + nz * relief], axis=-1)
        parts.append(_grid_mesh(fn, nu, nv))
    parts.append(_grid_mesh(lambda u, v: np.stack([x0 + (x1 - x0) * u, height + 0.3 * _value_noise(seed + 9, u, v, 8.0), z0 + (z1 - z0) * v], axis=-1), nu // 2, nu // 2))
    return _merge(parts)


def _filler_grid(seed, n_tris, x0, x1, z0, z1, y):
    """A displaced ground grid with exactly n_tris triangles (n_tris even)."""
    assert n_tris % 2 == 0 and n_tris > 0
    q = n_tris // 2
    nv = int(np.sqrt(q))
    while q % nv:
        nv -= 1
    nu = q // nv
    return _grid_mesh(lambda u, v: np.stack([x0 + (x1 - x0) * u, y + 0.03 * _value_noise(seed, u, v, 96.0), z0 + (z1 - z0) * v], axis=-1), nu, nv)


def street(seed=2, target_tris=3145728):
    """"Bistro-class" stand-in (config 4): a street canyon of relief facades, lamp posts, awnings and a displaced
    ground, exactly 3,145,728 triangles, 64 materials."""
    meshes, instances = [], []
    def add(mesh, mat):
        meshes.append(mesh)
        instances.append((len(meshes) - 1, mat % 64, None))
    total = 0
    k = 0
    for side, z0, z1 in ((0, 8.0, 28.0), (1, -28.0, -8.0)):
        x = -120.0
        while x < 120.0:
            wdt = 14.0 + 10.0 * float(_rand01(seed, 10, k))
            hgt = 10.0 + 18.0 * float(_rand01(seed, 11, k))
            m = _building_block(seed + 50 + k, x, x + wdt, z0, z1, hgt, 112, 96)
            add(m, k)
            total += len(m[1]) // 3
            x += wdt + 1.0
            k += 1
    for j in range(40):  # lamp posts
        m = _cylinder(-115.0 + 6.0 * j, 6.5 if j % 2 else -6.5, 0.0, 5.0, 0.08, 24, 32, seed + 500 + j)
        add(m, 40 + j % 8)
        total += len(m[1]) // 3
    for j in range(24):  # awnings (cloth-like)
        x0 = -110.0 + 9.0 * j
        zs = 8.0 if j % 2 else -8.0
        m = _grid_mesh(lambda u, v, x0=x0, zs=zs, j=j: np.stack([x0 + 4.0 * u, 3.2 - 0.5 * v + 0.1 * np.sin(8 * np.pi * u), zs - np.sign(zs) * 2.0 * v + 0.05 * _value_noise(seed + 600 + j, u, v, 8.0)], axis=-1), 64, 32)
        add(m, 48 + j % 8)
        total += len(m[1]) // 3
    rest = target_tris - total
    assert rest > 0 and rest % 2 == 0, (total, rest)
    add(_filler_grid(seed + 700, rest, -125.0, 125.0, -30.0, 30.0, 0.0), 60)
    scene = SceneData(meshes, instances, _palette(seed, 64), name=f"street(seed={seed})")
    assert scene.num_triangles == target_tris, scene.num_triangles
    return scene


STREET_CAM = {"eye": (-100.0, 1.7, 0.5), "target": (60.0, 9.0, -1.0)}


def _yaw_mesh(mesh, cx, cz, ang):
    """Rotate a mesh about the vertical axis through (cx, cz): positions and normals (a rigid motion baked in)."""
    v, idx = mesh
    v = v.copy()
    c, s = np.float64(np.cos(ang)), np.float64(np.sin(ang))
    p = v["pos"].astype(np.float64)
    x, z = p[:, 0] - cx, p[:, 2] - cz
    v["pos"][:, 0] = (cx + c * x + s * z).astype(np.float32)
    v["pos"][:, 2] = (cz - s * x + c * z).astype(np.float32)
    n = v["normal"].astype(np.float64)
    v["normal"][:, 0] = (c * n[:, 0] + s * n[:, 2]).astype(np.float32)
    v["normal"][:, 2] = (-s * n[:, 0] + c * n[:, 2]).astype(np.float32)
    return v, idx


def _random_rotations(seed, tag, n):
    """n uniformly distributed rotation matrices (n,3,3) from three hashed numbers each (Shoemake's quaternion method)."""
    i = np.arange(n)
    u1, u2, u3 = _rand01(seed, tag, i), _rand01(seed, tag + 1, i), _rand01(seed, tag + 2, i)
    qx, qy = np.sqrt(1 - u1) * np.sin(2 * np.pi * u2), np.sqrt(1 - u1) * np.cos(2 * np.pi * u2)
    qz, qw = np.sqrt(u1) * np.sin(2 * np.pi * u3), np.sqrt(u1) * np.cos(2 * np.pi * u3)
    R = np.empty((n, 3, 3))
    R[:, 0, 0] = 1 - 2 * (qy * qy + qz * qz); R[:, 0, 1] = 2 * (qx * qy - qz * qw); R[:, 0, 2] = 2 * (qx * qz + qy * qw)
    R[:, 1, 0] = 2 * (qx * qy + qz * qw); R[:, 1, 1] = 1 - 2 * (qx * qx + qz * qz); R[:, 1, 2] = 2 * (qy * qz - qx * qw)
    R[:, 2, 0] = 2 * (qx * qz - qy * qw); R[:, 2, 1] = 2 * (qy * qz + qx * qw); R[:, 2, 2] = 1 - 2 * (qx * qx + qy * qy)
    return R


def _boxes(centres, R, half):
    """n oriented boxes (12 triangles, 24 vertices each) as one mesh: centres (n,3), rotations (n,3,3), half extents (n,3)."""
    n = len(centres)
    faces = []   # (normal axis, sign): 4 corners counter-clockwise seen from outside
    for a in range(3):
        b, c = (a + 1) % 3, (a + 2) % 3
        for sgn in (-1.0, 1.0):
            cs = []
            for sb, sc in ((-1, -1), (1, -1), (1, 1), (-1, 1)) if sgn > 0 else ((-1, -1), (-1, 1), (1, 1), (1, -1)):
                e = np.zeros(3); e[a] = sgn; e[b] = sb; e[c] = sc
                cs.append(e)
            nn = np.zeros(3); nn[a] = sgn
            faces.append((np.array(cs), nn))
    local = np.stack([f[0] for f in faces])               # (6,4,3) corner signs
    fn = np.stack([f[1] for f in faces])                  # (6,3)
    P = local[None] * half[:, None, None, :]              # (n,6,4,3)
    P = np.einsum("nij,nfcj->nfci", R, P) + centres[:, None, None, :]
    N = np.einsum("nij,fj->nfi", R, fn)                   # (n,6,3)
    v = np.zeros(n * 24, dtype=api.VERTEX_DTYPE)
    v["pos"] = P.reshape(-1, 3).astype(np.float32)
    v["normal"] = np.repeat(N.reshape(-1, 3), 4, axis=0).astype(np.float32)
    v["uv"] = np.tile(np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float32), (n * 6, 1))
    base = (4 * np.arange(n * 6)).astype(np.uint32)
    idx = np.stack([base, base + 1, base + 2, base, base + 2, base + 3], axis=-1).reshape(-1).astype(np.uint32)
    return v, idx


def _rigid(R, t):
    """Column-major 4x4 (glm::float4x4, PtInstanceDesc.myTransform) of x -> R x + t."""
    m = np.eye(4)
    m[:3, :3] = R
    m[:3, 3] = t
    return m.T.reshape(16).astype(np.float32)


def street_hard(seed=4, target_tris=3145728):
    """A SAH-hostile "Bistro-class" stand-in at exactly 3,145,728 triangles (review of round 2: the grid scenes flatter the
    traversal). Nothing is axis-aligned: every building block is yawed by a random angle along a curved street and the
    ground is yawed and tilted; 7.5 % of the triangles are long and thin (sagging cables of 6-sided tubes spanning
    20-60 m, untessellated beams, banner quads tens of metres long, all running through finely tessellated detail);
    clutter (bumpy shrubs, tessellated crates) is INSTANCED with uniformly random rotations in overlapping clusters
    (real PtInstanceDesc transforms, flattened by pt_scene_create); the ground grid is 1 % of the budget and the
    remainder is litter: small randomly oriented quads. 64 materials."""
    meshes, instances = [], []
    def add(mesh, mat, xf=None):
        meshes.append(mesh)
        instances.append((len(meshes) - 1, mat % 64, xf))
    def centre_z(x):
        return 6.0 * np.sin(np.asarray(x) / 40.0)
    total = 0
    # ---- 24 building blocks, each yawed: never within 8 degrees of an axis ----
    k = 0
    for side in (1.0, -1.0):
        x = -118.0
        for _ in range(12):
            wdt = 14.0 + 8.0 * float(_rand01(seed, 10, k))
            dpt = 16.0 + 4.0 * float(_rand01(seed, 12, k))
            hgt = 10.0 + 18.0 * float(_rand01(seed, 11, k))
            cx, cz = x + 0.5 * wdt, float(centre_z(x + 0.5 * wdt)) + side * 20.0
            yaw = np.radians(8.0 + 74.0 * float(_rand01(seed, 13, k))) + 0.5 * np.pi * np.floor(4.0 * float(_rand01(seed, 14, k)))
            m = _building_block(seed + 50 + k, cx - 0.5 * wdt, cx + 0.5 * wdt, cz - 0.5 * dpt, cz + 0.5 * dpt, hgt, 112, 96)
            m = _yaw_mesh(m, cx, cz, yaw)
            add(m, k)
            total += len(m[1]) // 3
            x += wdt + 1.5
            k += 1
    # ---- long thin geometry ----
    # cables: 6-sided tubes of radius 1.5 cm, 10 segments along 20-60 m of sagging span across and along the street
    n_cables = 1800
    parts = []
    for j in range(n_cables):
        r = [float(_rand01(seed, 100 + q, j)) for q in range(8)]
        xa = -115.0 + 230.0 * r[0]
        xb = min(118.0, max(-118.0, xa + (20.0 + 40.0 * r[1]) * (1.0 if r[2] < 0.5 else -1.0)))
        za, zb = float(centre_z(xa)) + (-14.0 + 28.0 * r[3]), float(centre_z(xb)) + (-14.0 + 28.0 * r[4])
        ya, yb = 3.0 + 20.0 * r[5], 3.0 + 20.0 * r[6]
        sag = 0.5 + 2.5 * r[7]
        A, B = np.array([xa, ya, za]), np.array([xb, yb, zb])
        axis = (B - A) / max(np.linalg.norm(B - A), 1e-9)
        side_v = np.cross(axis, [0.0, 1.0, 0.0]); side_v /= max(np.linalg.norm(side_v), 1e-9)
        up_v = np.cross(side_v, axis)
        def fn(u, v, A=A, B=B, sag=sag, side_v=side_v, up_v=up_v):
            c = A + (B - A) * u[..., None]
            c[..., 1] -= sag * 4.0 * u * (1.0 - u)
            ang = 2 * np.pi * v
            return c + 0.015 * (np.cos(ang)[..., None] * side_v + np.sin(ang)[..., None] * up_v)
        parts.append(_grid_mesh(fn, 10, 6))
    cables = _merge(parts)
    add(cables, 40)
    total += len(cables[1]) // 3
    # beams: 15 cm x 15 cm boxes 15-40 m long, random yaw, tilted up to 12 degrees, through the facades
    n_beams = 1500
    i = np.arange(n_beams)
    bx = -118.0 + 236.0 * _rand01(seed, 120, i)
    bc = np.stack([bx, 4.0 + 21.0 * _rand01(seed, 121, i), centre_z(bx) + (-24.0 + 48.0 * _rand01(seed, 122, i))], axis=-1)
    yaw, tilt = 2 * np.pi * _rand01(seed, 123, i), np.radians(-12.0 + 24.0 * _rand01(seed, 124, i))
    Rb = np.zeros((n_beams, 3, 3))
    Rb[:, 0, 0] = np.cos(yaw) * np.cos(tilt); Rb[:, 0, 1] = -np.cos(yaw) * np.sin(tilt); Rb[:, 0, 2] = np.sin(yaw)
    Rb[:, 1, 0] = np.sin(tilt); Rb[:, 1, 1] = np.cos(tilt)
    Rb[:, 2, 0] = -np.sin(yaw) * np.cos(tilt); Rb[:, 2, 1] = np.sin(yaw) * np.sin(tilt); Rb[:, 2, 2] = np.cos(yaw)
    beams = _boxes(bc, Rb, np.stack([7.5 + 12.5 * _rand01(seed, 125, i), np.full(n_beams, 0.075), np.full(n_beams, 0.075)], axis=-1))
    add(beams, 41)
    total += len(beams[1]) // 3
    # banners / canopies: single quads 16-40 m x 2-5 m at random orientations above the street (two triangles spanning tens
    # of metres each)
    n_ban = 120
    i = np.arange(n_ban)
    Rq = _random_rotations(seed, 130, n_ban)
    qx = -110.0 + 220.0 * _rand01(seed, 134, i)
    qc = np.stack([qx, 9.0 + 16.0 * _rand01(seed, 135, i), centre_z(qx) + (-20.0 + 40.0 * _rand01(seed, 136, i))], axis=-1)
    e1 = Rq[:, :, 0] * (8.0 + 12.0 * _rand01(seed, 137, i))[:, None]
    e1[:, 1] *= 0.25                                                             # mostly level: they hang over the street, not into it
    e2 = Rq[:, :, 1] * (1.0 + 1.5 * _rand01(seed, 138, i))[:, None]
    v = np.zeros(4 * n_ban, dtype=api.VERTEX_DTYPE)
    v["pos"] = np.stack([qc - e1 - e2, qc + e1 - e2, qc + e1 + e2, qc - e1 + e2], axis=1).reshape(-1, 3).astype(np.float32)
    nq = np.cross(e1, e2)
    v["normal"] = np.repeat(nq / np.maximum(np.linalg.norm(nq, axis=-1, keepdims=True), 1e-30), 4, axis=0).astype(np.float32)
    v["uv"] = np.tile(np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float32), (n_ban, 1))
    base = (4 * i).astype(np.uint32)
    add((v, np.stack([base, base + 1, base + 2, base, base + 2, base + 3], axis=-1).reshape(-1).astype(np.uint32)), 42)
    total += 2 * n_ban
    long_thin = total - 24 * (4 * 112 * 96 * 2 + 56 * 56 * 2)
    # ---- instanced clutter with random rotations, in overlapping clusters ----
    def shrub(u, v):
        th, ph = 2 * np.pi * u, np.pi * (0.02 + 0.96 * v)
        rr = 0.9 * (1.0 + 0.9 * _value_noise(seed + 300, u, v, 10.0) + 0.25 * np.cos(9 * th) * np.sin(7 * ph))
        return np.stack([rr * np.sin(ph) * np.cos(th), rr * np.cos(ph), rr * np.sin(ph) * np.sin(th)], axis=-1)
    shrub_mesh = _grid_mesh(shrub, 32, 32)                                                  # 2,048 triangles
    parts = []
    for a in range(3):                                                                      # crate: 6 faces of 8 x 8 quads = 768 triangles
        for sgn in (-0.5, 0.5):
            def fn(u, v, a=a, sgn=sgn):
                p = [None, None, None]
                p[a] = sgn + 0.02 * _value_noise(seed + 310 + a, u, v, 6.0)
                p[(a + 1) % 3] = (u - 0.5) if sgn > 0 else (0.5 - u)
                p[(a + 2) % 3] = v - 0.5
                return np.stack(p, axis=-1)
            parts.append(_grid_mesh(fn, 8, 8))
    crate_mesh = _merge(parts)
    meshes.append(shrub_mesh); shrub_id = len(meshes) - 1
    meshes.append(crate_mesh); crate_id = len(meshes) - 1
    n_shrub, n_crate, n_clusters = 220, 200, 60
    cl = np.arange(n_clusters)
    clx = -112.0 + 224.0 * _rand01(seed, 200, cl)
    clz = centre_z(clx) + np.where(_rand01(seed, 201, cl) < 0.5, -1.0, 1.0) * (2.0 + 9.0 * _rand01(seed, 202, cl))
    for kind, (mesh_id, count, ntri, scale0) in enumerate(((shrub_id, n_shrub, 2048, 1.0), (crate_id, n_crate, 768, 1.2))):
        j = np.arange(count)
        Rc = _random_rotations(seed, 210 + 10 * kind, count)
        which = (_rand01(seed, 213 + 10 * kind, j) * n_clusters).astype(np.int64) % n_clusters
        off = np.stack([-1.6 + 3.2 * _rand01(seed, 214 + 10 * kind, j), 0.3 + 1.8 * _rand01(seed, 215 + 10 * kind, j),
                        -1.6 + 3.2 * _rand01(seed, 216 + 10 * kind, j)], axis=-1)
        for q in range(count):
            t = np.array([clx[which[q]], 0.0, clz[which[q]]]) + off[q]
            instances.append((mesh_id, (44 + q) % 64, _rigid(Rc[q] * scale0, t)))
        total += count * ntri
    # ---- ground: 128 x 128 quads (1 % of the budget), yawed by 17 degrees and tilted, so its cells are not axis-aligned either ----
    def ground(u, v):
        x, z = -150.0 + 300.0 * u, -60.0 + 120.0 * v
        c17, s17 = np.cos(np.radians(17.0)), np.sin(np.radians(17.0))
        return np.stack([c17 * x + s17 * z, 0.004 * x - 0.006 * z + 0.03 * _value_noise(seed + 700, u, v, 96.0), -s17 * x + c17 * z], axis=-1)
    g = _grid_mesh(ground, 128, 128)
    add(g, 60)
    total += len(g[1]) // 3
    # ---- the remainder: litter, small randomly oriented quads near the ground along the street ----
    rest = target_tris - total
    assert rest > 0 and rest % 2 == 0, (total, rest)
    n_lit = rest // 2
    i = np.arange(n_lit)
    Rl = _random_rotations(seed, 400, n_lit)
    lx = -118.0 + 236.0 * _rand01(seed, 404, i)
    lc = np.stack([lx, 0.05 + 2.5 * _rand01(seed, 405, i) ** 3, centre_z(lx) + (-12.0 + 24.0 * _rand01(seed, 406, i))], axis=-1)
    e1, e2 = Rl[:, :, 0] * 0.12, Rl[:, :, 1] * 0.07
    v = np.zeros(4 * n_lit, dtype=api.VERTEX_DTYPE)
    v["pos"] = np.stack([lc - e1 - e2, lc + e1 - e2, lc + e1 + e2, lc - e1 + e2], axis=1).reshape(-1, 3).astype(np.float32)
    v["normal"] = np.repeat(Rl[:, :, 2], 4, axis=0).astype(np.float32)
    base = (4 * i).astype(np.uint32)
    add((v, np.stack([base, base + 1, base + 2, base, base + 2, base + 3], axis=-1).reshape(-1).astype(np.uint32)), 61)
    scene = SceneData(meshes, instances, _palette(seed, 64), name=f"street_hard(seed={seed})")
    scene.long_thin_fraction = long_thin / float(target_tris)
    assert scene.num_triangles == target_tris, scene.num_triangles
    assert 0.05 <= scene.long_thin_fraction <= 0.10
    return scene


STREET_HARD_CAM = {"eye": (-95.0, 1.7, float(6.0 * np.sin(-95.0 / 40.0))), "target": (40.0, 8.0, float(6.0 * np.sin(1.0)))}


def courtyard(seed=3, target_tris=10485760):
    """"San-Miguel-class" stand-in (config 5): an arcaded courtyard with many small leaf-like quads (high depth
    complexity). Exactly 10,485,760 triangles."""
    meshes, instances = [], []
    def add(mesh, mat):
        meshes.append(mesh)
        instances.append((len(meshes) - 1, mat % 64, None))
    total = 0
    # surrounding arcade buildings
    blocks = ((-40.0, 40.0, 30.0, 45.0), (-40.0, 40.0, -45.0, -30.0), (-55.0, -40.0, -45.0, 45.0), (40.0, 55.0, -45.0, 45.0))
    for k, (x0, x1, z0, z1) in enumerate(blocks):
        m = _building_block(seed + 10 + k, x0, x1, z0, z1, 14.0, 512, 256, amp=0.12)
        add(m, k)
        total += len(m[1]) // 3
    # trees: trunk + a cloud of small leaf quads (2 triangles each)
    n_trees, leaves_per_tree = 24, 100000
    for t in range(n_trees):
        cx = -30.0 + 60.0 * float(_rand01(seed, 20, t))
        cz = -22.0 + 44.0 * float(_rand01(seed, 21, t))
        trunk = _cylinder(cx, cz, 0.0, 5.0, 0.25, 24, 16, seed + 100 + t)
        add(trunk, 8 + t % 4)
        total += len(trunk[1]) // 3
        i = np.arange(leaves_per_tree)
        r = 3.2 * np.cbrt(_rand01(seed + t, 30, i))
        th = 2 * np.pi * _rand01(seed + t, 31, i)
        ph = np.arccos(1 - 2 * _rand01(seed + t, 32, i))
        c = np.stack([cx + r * np.sin(ph) * np.cos(th), 6.5 + 0.8 * r * np.cos(ph), cz + r * np.sin(ph) * np.sin(th)], axis=-1)
        a = 2 * np.pi * _rand01(seed + t, 33, i)
        b = np.pi * _rand01(seed + t, 34, i)
        e1 = 0.09 * np.stack([np.cos(a) * np.sin(b), np.cos(b), np.sin(a) * np.sin(b)], axis=-1)
        e2 = 0.05 * np.cross(e1 / 0.09, np.stack([np.sin(a), 0 * a, -np.cos(a)], axis=-1))
        nrm = np.cross(e1, e2)
        nrm /= np.maximum(np.linalg.norm(nrm, axis=-1, keepdims=True), 1e-30)
        v = np.zeros(4 * leaves_per_tree, dtype=api.VERTEX_DTYPE)
        v["pos"] = np.stack([c - e1 - e2, c + e1 - e2, c + e1 + e2, c - e1 + e2], axis=1).reshape(-1, 3).astype(np.float32)
        v["normal"] = np.repeat(nrm, 4, axis=0).astype(np.float32)
        base = (4 * i).astype(np.uint32)
        idx = np.stack([base, base + 1, base + 2, base, base + 2, base + 3], axis=-1).reshape(-1).astype(np.uint32)
        add((v, idx), 12 + t % 8)
        total += 2 * leaves_per_tree
    rest = target_tris - total
    assert rest > 0 and rest % 2 == 0, (total, rest)
    add(_filler_grid(seed + 900, rest, -40.0, 40.0, -30.0, 30.0, 0.0), 30)
    scene = SceneData(meshes, instances, _palette(seed, 64), name=f"courtyard(seed={seed})")
    assert scene.num_triangles == target_tris, scene.num_triangles
    return scene


COURTYARD_CAM = {"eye": (-34.0, 1.7, -24.0), "target": (10.0, 6.0, 8.0)}


def albedo_image(seed=11, size=256):
    """A procedural RGBA8 albedo image (size x size): staggered bricks with mortar lines, per-brick tint and fine grain —
    the stand-in for a base-colour texture file (none ships with the reference; its one DDS is never sampled)."""
    y, x = np.mgrid[0:size, 0:size]
    bw, bh = size // 8, size // 16
    row = y // bh
    xs = (x + (row % 2) * (bw // 2)) % size
    col = xs // bw
    mortar = ((xs % bw) < 2) | ((y % bh) < 2)
    tint = 0.55 + 0.45 * _rand01(seed, 1, (row * 131 + col).ravel()).reshape(size, size)
    grain = 0.92 + 0.08 * _rand01(seed, 2, (y * size + x).ravel()).reshape(size, size)
    rgb = np.stack([tint * grain, (0.8 * tint + 0.1) * grain, (0.65 * tint + 0.2) * grain], axis=-1)
    rgb = np.where(mortar[..., None], 0.35, rgb)
    img = np.empty((size, size, 4), dtype=np.uint8)
    img[..., :3] = np.clip(np.floor(rgb * 255.0 + 0.5), 0, 255).astype(np.uint8)
    img[..., 3] = 255
    return img


def with_material_ext(scene, seed=7):
    """Attach the C4/C5 material extension (textured GGX; not in the reference) to a scene: per material a roughness in
    [0.08, 0.9], one in five metallic, the albedo cycling none / checker / value noise / IMAGE (albedo_image, sampled
    bilinearly at uv x scale), two in three shaded with GGX (the rest keep the reference's Lambert + Phong so both models
    are exercised)."""
    n = len(scene.materials)
    i = np.arange(n)
    rough = 0.08 + 0.82 * _rand01(seed, 1, i)
    metal = (_rand01(seed, 2, i) < 0.2).astype(np.float64)
    scale = 2.0 + np.floor(6.0 * _rand01(seed, 3, i))
    scene.materials_ext = [(float(rough[k]), float(metal[k]), int(k % 4), float(scale[k]), api.PT_MAT_GGX if k % 3 != 2 else 0)
                           for k in range(n)]
    scene.albedo_texture = albedo_image(seed + 4)
    return scene


def random_triangle_soup(seed, n, extent=10.0, size=1.0):
    """n random triangles in a cube — intersection-parity tests."""
    i = np.arange(n)
    c = np.stack([_rand01(seed, 1, i), _rand01(seed, 2, i), _rand01(seed, 3, i)], axis=-1) * extent
    e = (np.stack([_rand01(seed, 4 + k, i) for k in range(9)], axis=-1).reshape(n, 3, 3) - 0.5) * size
    P = (c[:, None, :] + e).astype(np.float32)
    v = np.zeros(3 * n, dtype=api.VERTEX_DTYPE)
    v["pos"] = P.reshape(-1, 3)
    nrm = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
    nrm /= np.maximum(np.linalg.norm(nrm, axis=-1, keepdims=True), 1e-30)
    v["normal"] = np.repeat(nrm, 3, axis=0)
    idx = np.arange(3 * n, dtype=np.uint32)
    return SceneData([(v, idx)], [(0, 0, None)], [((0.0, 0.0, 0.0), (0.7, 0.7, 0.7, 1.0))], name=f"soup({seed},{n})")
