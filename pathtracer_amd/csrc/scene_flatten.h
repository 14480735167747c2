// scene_flatten.h — host-side flattening of meshes x instances into the SoA buffers the kernels read.
// Restates PathTracer::InitRtScene (PathTracer/PathTracer.cpp:171-339) for a world-space triangle soup:
//   * per mesh the reference strips {normal, uv} into a 20-byte VertexData array and copies uvec3 indices
//     (:214-248); per instance it records {indexBuffer, vertexBuffer, materialIndex} (:276-295) and hands the
//     4x4 transform to the TLAS (:303); materials become {float3 emission, uint rgba8} (:318-329).
//   * here every instance is baked: positions are transformed on the host, indices are rebased into one global
//     vertex array, and each triangle remembers its instance id (== InstanceID() in ClosestHit).
// The multi-part pointer bug of the reference (:229-248) is not reproduced: a mesh is one part.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/pt_api.h"

namespace pt {

struct FlatScene {
  std::vector<float> tri_verts;          // 9 floats per triangle, world space, instance-major (global triangle id order)
  std::vector<uint32_t> prim_data;       // 4 per triangle: i0, i1, i2 (into the global vertex arrays), instance id
  std::vector<float> vertex_normal;      // 4 per vertex: normal.xyz, 0      (VertexData.myNormal)
  std::vector<float> vertex_uv;          // 2 per vertex                      (VertexData.myUv)
  std::vector<uint32_t> instance_material;
  std::vector<float> materials;          // 4 per material: emission.xyz, rgba8 bits
  uint32_t num_triangles() const { return (uint32_t)(tri_verts.size() / 9); }
  uint32_t num_vertices() const { return (uint32_t)(vertex_normal.size() / 4); }
};

// Returns PT_OK or PT_ERR_INVALID_ARG (err filled).
int flatten_scene(const PtMeshDesc* meshes, uint32_t num_meshes, const PtInstanceDesc* instances, uint32_t num_instances,
                  const PtMaterial* materials, uint32_t num_materials, uint32_t flags, FlatScene& out, std::string& err);

}  // namespace pt
