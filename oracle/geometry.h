// ORACLE — TEST INFRASTRUCTURE ONLY (see hlsl_math.h).
// Scene flattening (restates PathTracer::InitRtScene, PathTracer/PathTracer.cpp:171-339) and the
// TraceRay closest-hit query (PathTracing.hlsl:71). In the reference the BVH build and the
// ray/triangle test are a driver/hardware black box (PathTracer.cpp:271-272,338-339), so the oracle
// DEFINES them ("parity unpinned" against real DXR hardware):
//   * closest hit over opaque, two-sided triangles, tmin < t < tmax (DXR triangle interval is exclusive),
//   * watertight test of Woop/Benthin/Wald 2013 in fp32 with the fp64 edge-function fallback,
//   * ties in t resolved towards the LOWER global triangle id, so the answer does not depend on the
//     acceleration structure (brute force and any BVH must return the same id bit for bit).
// The oracle's own accelerator is an independent binned-SAH BVH2; it shares no code with the product's
// 8-wide compressed BVH.
#pragma once
#include <algorithm>
#include <cfloat>
#include <vector>
#include "hlsl_math.h"
#include "../include/pt_api.h"

namespace orc {

struct Tri { float3 v0, v1, v2; };

struct Scene {
  // world-space triangles in instance-major order: global triangle id = position in this array
  std::vector<Tri> tris;
  std::vector<uint32_t> tri_instance;    // InstanceID() of each triangle
  std::vector<uint32_t> tri_indices;     // 3 per triangle, rebased into `vertex_normal/uv` (LoadTriangleIndices, Common.hlsl:102-105)
  std::vector<float3> vertex_normal;     // VertexData.myNormal (PathTracer.cpp:214-217)
  std::vector<float2> vertex_uv;         // VertexData.myUv
  std::vector<uint32_t> instance_material;  // InstanceData.myMaterialIndex (PathTracer.cpp:276-295)
  std::vector<PtMaterial> materials;     // MaterialData (PathTracer.cpp:318-329)
  std::vector<uint32_t> albedo_tex;          // albedo pattern 3: RGBA8 image of the material extension (orc_scene_set_albedo_texture)
  uint32_t tex_w = 0, tex_h = 0;
  std::vector<PtMaterialExt> materials_ext;  // material extension (include/pt_api.h), empty = none; NOT in the reference
  // BVH2
  struct Node { float lo[3], hi[3]; uint32_t left, count; };  // count>0: leaf, left = first index into order[]
  std::vector<Node> nodes;
  std::vector<uint32_t> order;
};

static inline bool is_identity(const float* m) {
  for (int i = 0; i < 16; ++i) if (m[i] != ((i % 5 == 0) ? 1.0f : 0.0f)) return false;
  return true;
}
// p' = M * (p,1), column-major glm::float4x4, evaluated left to right; exact identity is skipped.
static inline float3 xform_point(const float* m, float3 p) {
  if (is_identity(m)) return p;
  return f3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
            m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
static inline float3 xform_dir(const float* m, float3 n) {
  if (is_identity(m)) return n;
  return f3(m[0] * n.x + m[4] * n.y + m[8] * n.z, m[1] * n.x + m[5] * n.y + m[9] * n.z,
            m[2] * n.x + m[6] * n.y + m[10] * n.z);
}

static inline void flatten(Scene& S, const PtMeshDesc* meshes, uint32_t nm, const PtInstanceDesc* inst, uint32_t ni,
                           const PtMaterial* mats, uint32_t nmat, uint32_t flags) {
  S.materials.assign(mats, mats + nmat);
  for (uint32_t i = 0; i < ni; ++i) {
    const PtMeshDesc& m = meshes[inst[i].myMeshIndex];
    const float* M = inst[i].myTransform;
    uint32_t vbase = (uint32_t)S.vertex_normal.size();
    for (uint32_t v = 0; v < m.num_vertices; ++v) {
      float3 n = f3(m.vertices[v].normal[0], m.vertices[v].normal[1], m.vertices[v].normal[2]);
      if (flags & PT_SCENE_TRANSFORM_NORMALS) n = xform_dir(M, n);
      S.vertex_normal.push_back(n);  // reference keeps object-space normals
      S.vertex_uv.push_back(float2{m.vertices[v].uv[0], m.vertices[v].uv[1]});
    }
    for (uint32_t t = 0; t < m.num_triangles; ++t) {
      Tri tr;
      float3* dst[3] = {&tr.v0, &tr.v1, &tr.v2};
      for (int k = 0; k < 3; ++k) {
        uint32_t vi = m.indices[3 * t + k];
        *dst[k] = xform_point(M, f3(m.vertices[vi].pos[0], m.vertices[vi].pos[1], m.vertices[vi].pos[2]));
        S.tri_indices.push_back(vbase + vi);
      }
      S.tris.push_back(tr);
      S.tri_instance.push_back(i);
    }
    S.instance_material.push_back(inst[i].myMaterialIndex);
  }
}

// ------------------------------------------------------------------ watertight ray/triangle ----
struct RayPre {  // per-ray constants of the watertight test
  float3 org;
  int kx, ky, kz;
  float Sx, Sy, Sz;
};
static inline float comp(float3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }
static inline RayPre ray_precompute(float3 org, float3 dir) {
  RayPre r;
  r.org = org;
  float ax = fabsf(dir.x), ay = fabsf(dir.y), az = fabsf(dir.z);
  int kz = 0; float m = ax;
  if (ay > m) { kz = 1; m = ay; }
  if (az > m) { kz = 2; }
  int kx = kz + 1; if (kx == 3) kx = 0;
  int ky = kx + 1; if (ky == 3) ky = 0;
  if (comp(dir, kz) < 0.0f) std::swap(kx, ky);
  r.kx = kx; r.ky = ky; r.kz = kz;
  r.Sx = comp(dir, kx) / comp(dir, kz);
  r.Sy = comp(dir, ky) / comp(dir, kz);
  r.Sz = 1.0f / comp(dir, kz);
  return r;
}
// Returns true and overwrites (t,u,v) when the triangle is hit with tmin < t and (t < tbest, or t == tbest
// and prim < best_prim). u = weight of v1, v = weight of v2 (DXR barycentrics).
static inline bool intersect_tri(const RayPre& r, const Tri& tri, uint32_t prim, float tmin, PtHit& best) {
  const float3 A = tri.v0 - r.org, B = tri.v1 - r.org, C = tri.v2 - r.org;
  const float Akz = comp(A, r.kz), Bkz = comp(B, r.kz), Ckz = comp(C, r.kz);
  const float Ax = comp(A, r.kx) - r.Sx * Akz, Ay = comp(A, r.ky) - r.Sy * Akz;
  const float Bx = comp(B, r.kx) - r.Sx * Bkz, By = comp(B, r.ky) - r.Sy * Bkz;
  const float Cx = comp(C, r.kx) - r.Sx * Ckz, Cy = comp(C, r.ky) - r.Sy * Ckz;
  float U = Cx * By - Cy * Bx;
  float V = Ax * Cy - Ay * Cx;
  float W = Bx * Ay - By * Ax;
  if (U == 0.0f || V == 0.0f || W == 0.0f) {
    double CxBy = (double)Cx * (double)By, CyBx = (double)Cy * (double)Bx;
    U = (float)(CxBy - CyBx);
    double AxCy = (double)Ax * (double)Cy, AyCx = (double)Ay * (double)Cx;
    V = (float)(AxCy - AyCx);
    double BxAy = (double)Bx * (double)Ay, ByAx = (double)By * (double)Ax;
    W = (float)(BxAy - ByAx);
  }
  if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
  const float det = U + V + W;
  if (det == 0.0f) return false;
  const float Az = r.Sz * Akz, Bz = r.Sz * Bkz, Cz = r.Sz * Ckz;
  const float T = U * Az + V * Bz + W * Cz;
  const float rcpDet = 1.0f / det;
  const float t = T * rcpDet;
  if (!(t > tmin)) return false;
  if (t < best.t || (t == best.t && best.prim != PT_MISS && prim < best.prim)) {  // t == tmax itself is NOT a hit
    best.t = t;
    best.u = V * rcpDet;
    best.v = W * rcpDet;
    best.prim = prim;
    return true;
  }
  return false;
}

// A ray with a NaN/Inf origin or direction hits nothing (all comparisons of the triangle test fail): defined as a miss.
static inline bool ray_is_finite(float3 org, float3 dir) { return std::isfinite((org.x + org.y + org.z) + (dir.x + dir.y + dir.z)); }

static inline PtHit trace_brute(const Scene& S, float3 org, float tmin, float3 dir, float tmax) {
  RayPre r = ray_precompute(org, dir);
  PtHit best; best.t = tmax; best.u = 0; best.v = 0; best.prim = PT_MISS;
  if (!ray_is_finite(org, dir)) { best.t = -1.0f; return best; }
  for (uint32_t i = 0; i < (uint32_t)S.tris.size(); ++i) intersect_tri(r, S.tris[i], i, tmin, best);
  if (best.prim == PT_MISS) best.t = -1.0f;
  return best;
}

// ------------------------------------------------------------------ BVH2 (oracle's own) ----
static inline void build_bvh2(Scene& S) {
  const uint32_t n = (uint32_t)S.tris.size();
  S.order.resize(n);
  std::vector<float> clo(3 * (size_t)n), chi(3 * (size_t)n), cen(3 * (size_t)n);
  for (uint32_t i = 0; i < n; ++i) {
    S.order[i] = i;
    const Tri& t = S.tris[i];
    const float vx[3][3] = {{t.v0.x, t.v0.y, t.v0.z}, {t.v1.x, t.v1.y, t.v1.z}, {t.v2.x, t.v2.y, t.v2.z}};
    for (int a = 0; a < 3; ++a) {
      float lo = std::min(vx[0][a], std::min(vx[1][a], vx[2][a]));
      float hi = std::max(vx[0][a], std::max(vx[1][a], vx[2][a]));
      float pad = 1e-5f * std::max(1.0f, std::max(fabsf(lo), fabsf(hi)));  // conservative vs fp32 error of the triangle test
      clo[3 * (size_t)i + a] = lo - pad;
      chi[3 * (size_t)i + a] = hi + pad;
      cen[3 * (size_t)i + a] = 0.5f * (lo + hi);
    }
  }
  S.nodes.clear();
  S.nodes.reserve(2 * (size_t)n + 1);
  struct Task { uint32_t node, begin, end; };
  std::vector<Task> stack;
  S.nodes.push_back(Scene::Node{});
  if (n == 0) { Scene::Node& r = S.nodes[0]; for (int a = 0; a < 3; ++a) { r.lo[a] = 1; r.hi[a] = -1; } r.left = 0; r.count = 0; return; }
  stack.push_back({0, 0, n});
  const int NB = 16;
  while (!stack.empty()) {
    Task tk = stack.back(); stack.pop_back();
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    float clo3[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi3[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (uint32_t i = tk.begin; i < tk.end; ++i) {
      size_t p = S.order[i];
      for (int a = 0; a < 3; ++a) {
        lo[a] = std::min(lo[a], clo[3 * p + a]); hi[a] = std::max(hi[a], chi[3 * p + a]);
        clo3[a] = std::min(clo3[a], cen[3 * p + a]); chi3[a] = std::max(chi3[a], cen[3 * p + a]);
      }
    }
    {
      Scene::Node& nd = S.nodes[tk.node];
      for (int a = 0; a < 3; ++a) { nd.lo[a] = lo[a]; nd.hi[a] = hi[a]; }
    }
    uint32_t cnt = tk.end - tk.begin;
    if (cnt <= 4) { S.nodes[tk.node].left = tk.begin; S.nodes[tk.node].count = cnt; continue; }
    // binned SAH over the 3 axes
    int bestAxis = -1, bestSplit = -1; float bestCost = FLT_MAX;
    for (int a = 0; a < 3; ++a) {
      float ext = chi3[a] - clo3[a];
      if (!(ext > 0.0f)) continue;
      float blo[NB][3], bhi[NB][3]; uint32_t bc[NB];
      for (int b = 0; b < NB; ++b) { bc[b] = 0; for (int k = 0; k < 3; ++k) { blo[b][k] = FLT_MAX; bhi[b][k] = -FLT_MAX; } }
      float scale = (float)NB / ext;
      for (uint32_t i = tk.begin; i < tk.end; ++i) {
        size_t p = S.order[i];
        int b = (int)((cen[3 * p + a] - clo3[a]) * scale); if (b >= NB) b = NB - 1; if (b < 0) b = 0;
        bc[b]++;
        for (int k = 0; k < 3; ++k) { blo[b][k] = std::min(blo[b][k], clo[3 * p + k]); bhi[b][k] = std::max(bhi[b][k], chi[3 * p + k]); }
      }
      float rarea[NB]; uint32_t rcnt[NB];
      float l3[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, h3[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX}; uint32_t c = 0;
      for (int b = NB - 1; b > 0; --b) {
        for (int k = 0; k < 3; ++k) { l3[k] = std::min(l3[k], blo[b][k]); h3[k] = std::max(h3[k], bhi[b][k]); }
        c += bc[b];
        float dx = h3[0] - l3[0], dy = h3[1] - l3[1], dz = h3[2] - l3[2];
        rarea[b] = c ? (dx * dy + dy * dz + dz * dx) : 0.0f; rcnt[b] = c;
      }
      for (int k = 0; k < 3; ++k) { l3[k] = FLT_MAX; h3[k] = -FLT_MAX; } c = 0;
      for (int b = 0; b < NB - 1; ++b) {
        for (int k = 0; k < 3; ++k) { l3[k] = std::min(l3[k], blo[b][k]); h3[k] = std::max(h3[k], bhi[b][k]); }
        c += bc[b];
        if (c == 0 || rcnt[b + 1] == 0) continue;
        float dx = h3[0] - l3[0], dy = h3[1] - l3[1], dz = h3[2] - l3[2];
        float cost = (dx * dy + dy * dz + dz * dx) * (float)c + rarea[b + 1] * (float)rcnt[b + 1];
        if (cost < bestCost) { bestCost = cost; bestAxis = a; bestSplit = b; }
      }
    }
    uint32_t mid;
    if (bestAxis < 0) {
      mid = tk.begin + cnt / 2;  // all centroids coincide: split the list in half
    } else {
      float ext = chi3[bestAxis] - clo3[bestAxis];
      float scale = (float)NB / ext;
      auto it = std::partition(S.order.begin() + tk.begin, S.order.begin() + tk.end, [&](uint32_t p) {
        int b = (int)((cen[3 * (size_t)p + bestAxis] - clo3[bestAxis]) * scale); if (b >= NB) b = NB - 1; if (b < 0) b = 0;
        return b <= bestSplit;
      });
      mid = (uint32_t)(it - S.order.begin());
      if (mid == tk.begin || mid == tk.end) mid = tk.begin + cnt / 2;
    }
    uint32_t l = (uint32_t)S.nodes.size();
    S.nodes.push_back(Scene::Node{}); S.nodes.push_back(Scene::Node{});
    S.nodes[tk.node].left = l; S.nodes[tk.node].count = 0;
    stack.push_back({l, tk.begin, mid});
    stack.push_back({l + 1, mid, tk.end});
  }
}

// slab test, conservative: returns entry distance or +inf on a miss
static inline float slab(const Scene::Node& nd, const float o[3], const float d[3], const float id[3], float tmin, float tbest) {
  float tn = tmin, tf = tbest;
  for (int a = 0; a < 3; ++a) {
    if (d[a] == 0.0f) {
      if (o[a] < nd.lo[a] || o[a] > nd.hi[a]) return INFINITY;
    } else {
      float t1 = (nd.lo[a] - o[a]) * id[a], t2 = (nd.hi[a] - o[a]) * id[a];
      float a0 = std::min(t1, t2), a1 = std::max(t1, t2);
      a0 -= fabsf(a0) * 4e-7f; a1 += fabsf(a1) * 4e-7f;
      tn = std::max(tn, a0); tf = std::min(tf, a1);
    }
  }
  return tn <= tf ? tn : INFINITY;
}

struct TraceCounters { uint64_t nodes = 0, tris = 0; };

static inline PtHit trace_bvh(const Scene& S, float3 org, float tmin, float3 dir, float tmax, TraceCounters* cnt = nullptr) {
  RayPre r = ray_precompute(org, dir);
  PtHit best; best.t = tmax; best.u = 0; best.v = 0; best.prim = PT_MISS;
  if (S.tris.empty() || !ray_is_finite(org, dir)) { best.t = -1.0f; return best; }
  const float o[3] = {org.x, org.y, org.z}, d[3] = {dir.x, dir.y, dir.z};
  const float id[3] = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
  uint32_t stack[128]; int sp = 0;
  stack[sp++] = 0;
  while (sp) {
    const Scene::Node& nd = S.nodes[stack[--sp]];
    if (cnt) cnt->nodes++;
    if (slab(nd, o, d, id, tmin, best.t) == INFINITY) continue;
    if (nd.count) {
      for (uint32_t i = 0; i < nd.count; ++i) {
        uint32_t p = S.order[nd.left + i];
        if (cnt) cnt->tris++;
        intersect_tri(r, S.tris[p], p, tmin, best);
      }
    } else {
      float tl = slab(S.nodes[nd.left], o, d, id, tmin, best.t);
      float tr = slab(S.nodes[nd.left + 1], o, d, id, tmin, best.t);
      if (tl == INFINITY && tr == INFINITY) continue;
      // push the farther child first; slab() is re-run at pop time against the (possibly shrunk) interval
      if (tl <= tr) { if (tr != INFINITY) stack[sp++] = nd.left + 1; stack[sp++] = nd.left; }
      else { if (tl != INFINITY) stack[sp++] = nd.left; stack[sp++] = nd.left + 1; }
    }
  }
  if (best.prim == PT_MISS) best.t = -1.0f;
  return best;
}

}  // namespace orc
