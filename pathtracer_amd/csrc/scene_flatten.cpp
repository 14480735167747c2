// scene_flatten.cpp — see scene_flatten.h. Reference: PathTracer/PathTracer.cpp:171-339.
#include "scene_flatten.h"

#include <cmath>
#include <cstring>

namespace pt {

static bool transform_is_identity(const float* m) {
  for (int i = 0; i < 16; ++i)
    if (m[i] != ((i % 5 == 0) ? 1.0f : 0.0f)) return false;
  return true;
}

int flatten_scene(const PtMeshDesc* meshes, uint32_t num_meshes, const PtInstanceDesc* instances, uint32_t num_instances,
                  const PtMaterial* materials, uint32_t num_materials, uint32_t flags, FlatScene& out, std::string& err) {
  out = FlatScene();
  if ((!meshes && num_meshes) || (!instances && num_instances) || (!materials && num_materials)) {
    err = "pt_scene_create: null array with non-zero count";
    return PT_ERR_INVALID_ARG;
  }
  size_t total_tris = 0, total_verts = 0;
  for (uint32_t i = 0; i < num_instances; ++i) {
    const PtInstanceDesc& in = instances[i];
    if (in.myMeshIndex >= num_meshes) { err = "pt_scene_create: instance " + std::to_string(i) + " mesh index out of range"; return PT_ERR_INVALID_ARG; }
    if (in.myMaterialIndex >= num_materials) { err = "pt_scene_create: instance " + std::to_string(i) + " material index out of range"; return PT_ERR_INVALID_ARG; }
    const PtMeshDesc& m = meshes[in.myMeshIndex];
    if ((m.num_vertices && !m.vertices) || (m.num_triangles && !m.indices)) { err = "pt_scene_create: mesh with null data"; return PT_ERR_INVALID_ARG; }
    total_tris += m.num_triangles;
    total_verts += m.num_vertices;
  }
  if (total_tris >= 0xFFFFFFFFull || total_verts >= 0xFFFFFFFFull) { err = "pt_scene_create: more than 2^32-2 triangles or vertices"; return PT_ERR_INVALID_ARG; }
  // every index must address a vertex of its mesh (checked once per mesh)
  for (uint32_t mi = 0; mi < num_meshes; ++mi) {
    const PtMeshDesc& m = meshes[mi];
    if ((m.num_vertices && !m.vertices) || (m.num_triangles && !m.indices)) { err = "pt_scene_create: mesh " + std::to_string(mi) + " has null data"; return PT_ERR_INVALID_ARG; }
    for (size_t k = 0; k < (size_t)m.num_triangles * 3; ++k)
      if (m.indices[k] >= m.num_vertices) { err = "pt_scene_create: mesh " + std::to_string(mi) + " has an index beyond its vertex count"; return PT_ERR_INVALID_ARG; }
  }

  out.tri_verts.reserve(total_tris * 9);
  out.prim_data.reserve(total_tris * 4);
  out.vertex_normal.reserve(total_verts * 4);
  out.vertex_uv.reserve(total_verts * 2);
  for (uint32_t i = 0; i < num_instances; ++i) {
    const PtInstanceDesc& in = instances[i];
    const PtMeshDesc& m = meshes[in.myMeshIndex];
    const float* M = in.myTransform;  // column-major glm::float4x4
    const bool ident = transform_is_identity(M);
    const uint32_t vbase = out.num_vertices();
    for (uint32_t v = 0; v < m.num_vertices; ++v) {
      const float* n = m.vertices[v].normal;
      float nx = n[0], ny = n[1], nz = n[2];
      if ((flags & PT_SCENE_TRANSFORM_NORMALS) && !ident) {
        float tx = M[0] * nx + M[4] * ny + M[8] * nz;
        float ty = M[1] * nx + M[5] * ny + M[9] * nz;
        float tz = M[2] * nx + M[6] * ny + M[10] * nz;
        nx = tx; ny = ty; nz = tz;
      }
      out.vertex_normal.push_back(nx);
      out.vertex_normal.push_back(ny);
      out.vertex_normal.push_back(nz);
      out.vertex_normal.push_back(0.0f);
      out.vertex_uv.push_back(m.vertices[v].uv[0]);
      out.vertex_uv.push_back(m.vertices[v].uv[1]);
    }
    for (uint32_t t = 0; t < m.num_triangles; ++t) {
      for (int k = 0; k < 3; ++k) {
        const uint32_t vi = m.indices[3 * (size_t)t + k];
        const float* p = m.vertices[vi].pos;
        float x = p[0], y = p[1], z = p[2];
        if (!ident) {
          float tx = M[0] * x + M[4] * y + M[8] * z + M[12];
          float ty = M[1] * x + M[5] * y + M[9] * z + M[13];
          float tz = M[2] * x + M[6] * y + M[10] * z + M[14];
          x = tx; y = ty; z = tz;
        }
        out.tri_verts.push_back(x);
        out.tri_verts.push_back(y);
        out.tri_verts.push_back(z);
        out.prim_data.push_back(vbase + vi);
      }
      out.prim_data.push_back(i);
    }
    out.instance_material.push_back(in.myMaterialIndex);
  }
  out.materials.resize((size_t)num_materials * 4);
  for (uint32_t k = 0; k < num_materials; ++k) {
    memcpy(&out.materials[(size_t)k * 4], materials[k].myEmission, 12);
    memcpy(&out.materials[(size_t)k * 4 + 3], &materials[k].myColor, 4);
  }
  return PT_OK;
}

}  // namespace pt
