// bvh8.h — memory layout of the compressed 8-wide BVH shared by the host builder and the HIP traversal kernel.
//
// Replaces the driver-built BLAS/TLAS of the reference (PathTracer/PathTracer.cpp:271-272,338-339), which is a black
// box there. Layout follows the published compressed-wide-BVH scheme (Ylitie, Karras, Laine, HPG 2017): 80-byte nodes
// with 8-bit quantised child boxes on a per-node power-of-two grid, fetched as five 128-bit loads per lane.
//
//   dword  0..2  : p            float3, origin of the quantisation grid (node box minimum)
//   dword  3     : e[0] | e[1]<<8 | e[2]<<16 | imask<<24     e = biased fp32 exponent of the grid step per axis,
//                                                            imask bit s = child slot s is an internal node
//   dword  4     : child_base   index of the first internal child (internal children are contiguous, in slot order)
//   dword  5     : tri_base     index of the first primitive record referenced by this node's leaf children
//   dword  6..7  : meta[8]      per slot: 0 = empty; internal: 0b001_11sss (sss = slot); leaf: high 3 bits = record
//                               count in unary (1:001 2:011 3:111), low 5 bits = first record relative to tri_base
//   dword  8..9  : qlo_x[8]     child box minima / maxima in grid units, one byte per slot
//   dword 10..11 : qlo_y[8]
//   dword 12..13 : qlo_z[8]
//   dword 14..15 : qhi_x[8]
//   dword 16..17 : qhi_y[8]
//   dword 18..19 : qhi_z[8]
// child box = p + 2^(e-127) * q. Slots are assigned so that for a ray whose direction-sign octant is r
// (bit a set <=> dir[a] < 0), visiting slots in DESCENDING (slot ^ r) order is approximately front to back.
#pragma once
#include <stdint.h>

#define PT_BVH8_NODE_BYTES 80
#define PT_TRI_RECORD_BYTES 48
#define PT_PRIM_RECORD_BYTES 64
#define PT_BVH8_MAX_LEAF_TRIS 3   // primitive records per leaf child (a record holds one triangle or an edge-sharing pair)

typedef struct PtBvh8Node {
  float p[3];
  uint8_t e[3];
  uint8_t imask;
  uint32_t child_base;
  uint32_t tri_base;
  uint8_t meta[8];
  uint8_t qlo[3][8];
  uint8_t qhi[3][8];
} PtBvh8Node;

// A hierarchy references ONE of two record formats (Bvh8::paired, chosen per scene by the builder):
//
// 48-byte triangle record, in BVH leaf order: three world-space vertices + the global triangle id.
//   dword 0..2 v0, dword 3 prim id | dword 4..6 v1, dword 7 pad | dword 8..10 v2, dword 11 pad
typedef struct PtTriRecord {
  float v0[3];
  uint32_t prim;
  float v1[3];
  uint32_t pad0;
  float v2[3];
  uint32_t pad1;
} PtTriRecord;

// 64-byte primitive record, in BVH leaf order: ONE triangle, or TWO triangles that share an edge (most meshes are
// quads split in two: the pair costs four 128-bit loads instead of six, the hierarchy is built over half as many
// primitives and the scene shrinks by a third — what counts once it no longer fits the 256 MB Infinity Cache). Triangle 0 is (v0, v1, v2) in its original vertex order. Triangle 1 (prim1 != 0xFFFFFFFF) keeps ITS
// original vertex order too, as three 2-bit indices into {v0, v1, v2, v3}: code = i0 | i1 << 2 | i2 << 4 — the
// intersection arithmetic of either triangle is exactly that of a stand-alone test, so results do not depend on pairing.
//   dword 0..2 v0, 3 prim0 | 4..6 v1, 7 prim1 | 8..10 v2, 11 code | 12..14 v3, 15 pad
typedef struct PtPrimRecord {
  float v0[3];
  uint32_t prim0;
  float v1[3];
  uint32_t prim1;
  float v2[3];
  uint32_t code;
  float v3[3];
  uint32_t pad;
} PtPrimRecord;

#ifdef __cplusplus
static_assert(sizeof(PtBvh8Node) == PT_BVH8_NODE_BYTES, "BVH8 node must be 80 bytes");
static_assert(sizeof(PtTriRecord) == PT_TRI_RECORD_BYTES, "triangle record must be 48 bytes");
static_assert(sizeof(PtPrimRecord) == PT_PRIM_RECORD_BYTES, "primitive record must be 64 bytes");
#endif
