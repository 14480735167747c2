// bvh_builder.h — host-side BVH construction for the traversal kernel (replaces the driver BLAS/TLAS build the
// reference requests at PathTracer/PathTracer.cpp:271-272,338-339).
#pragma once
#include <cstdint>
#include <vector>
#include <cstddef>

#include "bvh8.h"

namespace pt {

struct Bvh8 {
  std::vector<PtBvh8Node> nodes;   // node 0 is the root
  bool paired = false;             // which record format the leaves reference
  std::vector<PtTriRecord> tris;   // !paired: 48-byte triangle records, leaf order
  std::vector<PtPrimRecord> prims; // paired: 64-byte records (one triangle or an edge-sharing pair each), leaf order
  uint32_t num_pairs = 0;          // records that hold two triangles
  size_t num_records() const { return paired ? prims.size() : tris.size(); }
  size_t record_bytes() const { return paired ? sizeof(PtPrimRecord) : sizeof(PtTriRecord); }
  const void* records() const { return paired ? (const void*)prims.data() : (const void*)tris.data(); }
  uint32_t max_depth = 0;          // number of node levels on the longest root-to-leaf path
  float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  double build_ms = 0.0;
  double sah_cost = 0.0;
  uint64_t num_references = 0;     // leaf references of the binary tree: primitives + the duplicates spatial splits made
  uint64_t spatial_splits = 0;     // nodes of the binary tree that were split by a plane instead of by partitioning objects
  int reinsertion_kept = 0;        // 0 = the top-down binary tree was collapsed as built; 1 / 2 = its insertion-optimised copy (form 1 / 2) gave the better 8-wide tree
};

// verts: 9 floats per triangle (v0,v1,v2 world space); triangle i gets global id i.
// Edge-sharing triangles are paired into primitives first (greedy, by exact vertex positions). Binned-SAH binary BVH
// with spatial splits (SBVH: a long or diagonal primitive may be referenced from several leaves, each with the tight box
// of its piece; single-reference leaves) -> insertion-based optimisation of that tree, kept when the 8-wide tree gets
// better -> SAH-optimal collapse to 8-wide nodes with <=3 records per
// leaf -> octant-ordered child slots -> 8-bit quantisation (conservative, padded by a scene-relative epsilon).
// pair_mode: 0 = triangle records, 1 = pair edge-sharing triangles, -1 = decide by size (pairs once the unpaired hierarchy
// would exceed the Infinity Cache: measured +9.5 % on 10.5 M triangles, -4 % on 262 k).
// fast: no spatial splits and no insertion-based optimisation of the binary tree (PT_SCENE_FAST_BUILD).
void build_bvh8(const float* verts, uint32_t num_tris, Bvh8& out, int num_threads = 0, int pair_mode = -1, bool fast = false);

}  // namespace pt
