"""Turns the reference's default scene (resources/models/CornellBox.obj + .mtl, an Apache-2.0 data asset) into the
JSON data fixture pathtracer_amd/assets/cornell_box.json. Run in the authoring container only (/root/reference is not
on the GPU box). The fixture holds numbers only: vertices, indices, instance and material tables."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd.scenes import load_obj, scene_to_json  # noqa: E402

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/resources/models/CornellBox.obj"
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pathtracer_amd", "assets", "cornell_box.json")
scene = load_obj(src)
scene_to_json(scene, dst)
print(dst, "meshes", len(scene.meshes), "tris", scene.num_triangles, "materials", len(scene.materials))
for i, (m, mat, _) in enumerate(scene.instances):
    print(" instance", i, "mesh", m, "tris", len(scene.meshes[m][1]) // 3, "material", mat, scene.materials[mat])
