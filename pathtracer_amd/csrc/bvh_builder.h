// bvh_builder.h — host-side BVH construction for the traversal kernel (replaces the driver BLAS/TLAS build the
// reference requests at PathTracer/PathTracer.cpp:271-272,338-339).
#pragma once
#include <cstdint>
#include <vector>
#include "bvh8.h"

namespace pt {

struct Bvh8 {
  std::vector<PtBvh8Node> nodes;   // node 0 is the root
  std::vector<PtTriRecord> tris;   // leaf order
  uint32_t max_depth = 0;          // number of node levels on the longest root-to-leaf path
  float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  double build_ms = 0.0;
  double sah_cost = 0.0;
};

// verts: 9 floats per triangle (v0,v1,v2 world space); triangle i gets global id i.
// Binned-SAH binary BVH (single-triangle leaves) -> SAH-optimal collapse to 8-wide nodes with <=3 triangles per
// leaf -> octant-ordered child slots -> 8-bit quantisation (conservative, padded by a scene-relative epsilon).
void build_bvh8(const float* verts, uint32_t num_tris, Bvh8& out, int num_threads = 0);

}  // namespace pt
