// bvh8_walker.cpp — TEST INFRASTRUCTURE. A scalar host walker over the compressed 8-wide BVH exactly as the library
// built it (read back with pt_debug_read_bvh), written from the layout documented in pathtracer_amd/csrc/bvh8.h.
// It runs the CANONICAL schedule (nearest child first by octant order, every triangle group tested as soon as its node
// is opened, no postponing / deferral) and counts nodes opened and triangles tested per ray, so that the figures
// bench.py's roofline is built on (N_node, N_tri of SURVEY §8d) are cross-checked by a second implementation on the
// SAME hierarchy: tests assert that the kernel's COUNT instantiation reports the same totals and the same hits.
// Build: g++ -O2 -ffp-contract=off -shared -fPIC (tests/walker/Makefile). The arithmetic that decides a box hit is the
// kernel's (one fused multiply-add per plane, 1.000001 widening), the triangle test is the watertight test shared
// with the oracle.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../pathtracer_amd/csrc/bvh8.h"

namespace {

struct Ray { float o[3], tmin, d[3], tmax; };
struct Hit { float t, u, v; uint32_t prim; };

inline float sel3(const float* x, int k) { return x[k]; }

struct Tri {
  float o[3]; int kx, ky, kz; float Sx, Sy, Sz;
  void init(const Ray& r) {
    for (int a = 0; a < 3; ++a) o[a] = r.o[a];
    const float ax = std::fabs(r.d[0]), ay = std::fabs(r.d[1]), az = std::fabs(r.d[2]);
    kz = 0; float m = ax;
    if (ay > m) { kz = 1; m = ay; }
    if (az > m) { kz = 2; }
    kx = kz + 1; if (kx == 3) kx = 0;
    ky = kx + 1; if (ky == 3) ky = 0;
    const float dz = r.d[kz];
    if (dz < 0.0f) { int t = kx; kx = ky; ky = t; }
    Sx = r.d[kx] / dz; Sy = r.d[ky] / dz; Sz = 1.0f / dz;
  }
  struct Sh { float x, y, kz; };
  Sh shear(const float* v) const {
    const float a[3] = {v[0] - o[0], v[1] - o[1], v[2] - o[2]};
    Sh r;
    r.kz = a[kz];
    r.x = a[kx] - Sx * r.kz;
    r.y = a[ky] - Sy * r.kz;
    return r;
  }
  void test3(const Sh& A, const Sh& B, const Sh& C, uint32_t prim, float tmin, Hit& best) const {
    float U = C.x * B.y - C.y * B.x, V = A.x * C.y - A.y * C.x, W = B.x * A.y - B.y * A.x;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
      U = (float)((double)C.x * (double)B.y - (double)C.y * (double)B.x);
      V = (float)((double)A.x * (double)C.y - (double)A.y * (double)C.x);
      W = (float)((double)B.x * (double)A.y - (double)B.y * (double)A.x);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return;
    const float det = U + V + W;
    if (det == 0.0f) return;
    const float Az = Sz * A.kz, Bz = Sz * B.kz, Cz = Sz * C.kz;
    const float Tt = U * Az + V * Bz + W * Cz;
    const float rcp = 1.0f / det;
    const float t = Tt * rcp;
    if (!(t > tmin)) return;
    if (t < best.t || (t == best.t && best.prim != 0xFFFFFFFFu && prim < best.prim)) {
      best.t = t; best.u = V * rcp; best.v = W * rcp; best.prim = prim;
    }
  }
  // one primitive record: triangle 0 = (v0,v1,v2); triangle 1 picks its vertices out of {v0,v1,v2,v3} by the 2-bit codes
  uint32_t test(const PtPrimRecord& T, float tmin, Hit& best) const {
    const Sh s[4] = {shear(T.v0), shear(T.v1), shear(T.v2), shear(T.v3)};
    test3(s[0], s[1], s[2], T.prim0, tmin, best);
    if (T.prim1 == 0xFFFFFFFFu) return 1;
    test3(s[T.code & 3u], s[(T.code >> 2) & 3u], s[(T.code >> 4) & 3u], T.prim1, tmin, best);
    return 2;
  }
  uint32_t test(const PtTriRecord& T, float tmin, Hit& best) const {
    test3(shear(T.v0), shear(T.v1), shear(T.v2), T.prim, tmin, best);
    return 1;
  }
};

inline float safe_dir(float d) { return std::fabs(d) < 1e-25f ? std::copysign(1e-25f, d) : d; }
inline float pow2_from_biased(uint8_t e) { uint32_t b = (uint32_t)e << 23; float f; std::memcpy(&f, &b, 4); return f; }

}  // namespace

static uint32_t* g_visit_hist = nullptr;   // optional: visits per node index (tools/exp_node_hist.py)
extern "C" void bvh8_set_visit_hist(uint32_t* hist) { g_visit_hist = hist; }

// records: num_tris leaf records of record_bytes each (48 = PtTriRecord, 64 = PtPrimRecord)
extern "C" int bvh8_walk(const PtBvh8Node* nodes, uint64_t num_nodes, const void* records, uint64_t num_tris, uint32_t record_bytes,
                         float scene_mag, const Ray* rays, uint64_t n, Hit* hits, uint64_t* nodes_opened, uint64_t* tris_tested) {
  if (record_bytes != sizeof(PtTriRecord) && record_bytes != sizeof(PtPrimRecord)) return -3;
  const bool paired = record_bytes == sizeof(PtPrimRecord);
  const PtPrimRecord* prims = static_cast<const PtPrimRecord*>(records);
  const PtTriRecord* tris = static_cast<const PtTriRecord*>(records);
  uint64_t cn = 0, ct = 0;
  std::vector<std::pair<uint32_t, uint32_t>> stack;
  for (uint64_t i = 0; i < n; ++i) {
    const Ray& r = rays[i];
    Hit best{r.tmax, 0.0f, 0.0f, 0xFFFFFFFFu};
    const float chk = (r.o[0] + r.o[1] + r.o[2]) + (r.d[0] + r.d[1] + r.d[2]);
    if (!std::isfinite(chk)) { hits[i] = Hit{-1.0f, 0.0f, 0.0f, 0xFFFFFFFFu}; continue; }   // non-finite rays miss at once
    Tri tt; tt.init(r);
    const float sd[3] = {safe_dir(r.d[0]), safe_dir(r.d[1]), safe_dir(r.d[2])};
    const float id[3] = {1.0f / sd[0], 1.0f / sd[1], 1.0f / sd[2]};
    const uint32_t oct = (sd[0] < 0.0f ? 1u : 0u) | (sd[1] < 0.0f ? 2u : 0u) | (sd[2] < 0.0f ? 4u : 0u);
    const float reach = std::fmax(scene_mag, std::fmax(std::fabs(r.o[0]), std::fmax(std::fabs(r.o[1]), std::fabs(r.o[2]))));
    const bool risky = !(reach * std::fmax(std::fabs(id[0]), std::fmax(std::fabs(id[1]), std::fabs(id[2]))) < 1e36f);
    stack.clear();
    uint32_t gbase = 0u, gbits = 0x80000000u;   // root: one pending internal child at priority 7 of a group with base 0
    for (;;) {
      if (gbits > 0x00FFFFFFu) {
        int b = 31; while (!((gbits >> b) & 1u)) --b;
        const uint32_t before = gbits;
        gbits &= ~(1u << b);
        if (gbits > 0x00FFFFFFu) stack.push_back({gbase, gbits});
        const uint32_t slot = ((uint32_t)b - 24u) ^ oct;
        uint32_t rel = 0;
        for (uint32_t s = 0; s < slot; ++s) rel += (before >> s) & 1u;    // internal children below this slot (imask = low byte)
        const uint64_t ni = (uint64_t)gbase + rel;
        if (ni >= num_nodes) return -1;
        const PtBvh8Node& nd = nodes[ni];
        ++cn;
        if (g_visit_hist) ++g_visit_hist[ni];
        uint32_t hitmask = 0u;
        for (int s = 0; s < 8; ++s) {
          const uint8_t meta = nd.meta[s];
          float tn = r.tmin, tf = best.t;
          float tnear[3], tfar[3];
          for (int a = 0; a < 3; ++a) {
            const bool neg = (oct >> a) & 1u;
            const float qn = (float)(neg ? nd.qhi[a][s] : nd.qlo[a][s]), qf = (float)(neg ? nd.qlo[a][s] : nd.qhi[a][s]);
            const float sc = pow2_from_biased(nd.e[a]), p = nd.p[a] - r.o[a];
            if (risky) { tnear[a] = std::fmaf(qn, sc, p) * id[a]; tfar[a] = std::fmaf(qf, sc, p) * id[a]; }
            else { tnear[a] = std::fmaf(qn, sc * id[a], p * id[a]); tfar[a] = std::fmaf(qf, sc * id[a], p * id[a]); }
          }
          tn = std::fmax(std::fmax(tnear[0], tnear[1]), std::fmax(tnear[2], r.tmin));
          tf = std::fmin(std::fmin(tfar[0], tfar[1]), std::fmin(tfar[2], best.t));
          // the kernel's decision (traverse_device.h, PT_BOX_SDWA_MASK): the sign of ONE fused tf * 1.000001 - tn
          if (std::signbit(std::fmaf(tf, 1.000001f, -tn))) continue;
          const bool inner = (meta & 0x18u) == 0x18u;          // low five bits 0b11sss: only internal children reach 24+
          const uint32_t bits = meta >> 5;                       // internal: 1; leaf: triangle count in unary
          const uint32_t pos = inner ? 24u + (((uint32_t)meta & 7u) ^ oct) : ((uint32_t)meta & 0x1fu);
          hitmask |= bits << pos;
        }
        gbase = nd.child_base;
        gbits = (hitmask & 0xFF000000u) | nd.imask;
        uint32_t tb = hitmask & 0x00FFFFFFu;
        while (tb) {                                            // the node's triangle group, lowest index first
          int k = 0; while (!((tb >> k) & 1u)) ++k;
          tb &= tb - 1u;
          const uint64_t ti = (uint64_t)nd.tri_base + (uint32_t)k;
          if (ti >= num_tris) return -2;
          ct += paired ? tt.test(prims[ti], r.tmin, best) : tt.test(tris[ti], r.tmin, best);
        }
      } else if (!stack.empty()) {
        gbase = stack.back().first; gbits = stack.back().second;
        stack.pop_back();
      } else {
        break;
      }
    }
    if (best.prim == 0xFFFFFFFFu) best.t = -1.0f;
    hits[i] = best;
  }
  *nodes_opened = cn; *tris_tested = ct;
  return 0;
}

// CPU-only check of the product's host-side builder (pathtracer_amd/csrc/bvh_builder.cpp is plain C++): build from a
// triangle list (9 floats each) in the given leaf format (0 = 48-byte triangles, 1 = 64-byte pairs, -1 = builder's choice)
// and walk it. info = {nodes, records, pairs, record_bytes, max_depth}.
#include "../../pathtracer_amd/csrc/bvh_builder.h"
extern "C" int bvh8_build_and_walk(const float* verts, uint32_t num_tris, int pair_mode, const Ray* rays, uint64_t n, Hit* hits,
                                   uint64_t* nodes_opened, uint64_t* tris_tested, uint64_t* info) {
  pt::Bvh8 b;
  pt::build_bvh8(verts, num_tris, b, 4, pair_mode);
  float mag = 0.0f;
  for (int a = 0; a < 3; ++a) mag = std::fmax(mag, std::fmax(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
  info[0] = b.nodes.size(); info[1] = b.num_records(); info[2] = b.num_pairs; info[3] = b.record_bytes(); info[4] = b.max_depth;
  return bvh8_walk(b.nodes.data(), b.nodes.size(), b.records(), b.num_records(), (uint32_t)b.record_bytes(), mag, rays, n, hits,
                   nodes_opened, tris_tested);
}

// Build once, walk many ray sets (tools/bvh_cpu_quality.py: builder experiments without a GPU). The walk is split over
// `threads` host threads by ray ranges; the counters are summed.
#include <thread>
extern "C" void* bvh8_build(const float* verts, uint32_t num_tris, int pair_mode, int threads, uint64_t* info) {
  pt::Bvh8* b = new pt::Bvh8();
  pt::build_bvh8(verts, num_tris, *b, threads, pair_mode);
  info[0] = b->nodes.size(); info[1] = b->num_records(); info[2] = b->num_pairs; info[3] = b->record_bytes(); info[4] = b->max_depth;
  info[5] = (uint64_t)b->build_ms;
  { const double c = (double)b->sah_cost; memcpy(&info[6], &c, 8); }   // SAH cost of the 8-wide tree (the collapse's objective)
  return b;
}
extern "C" void bvh8_free(void* h) { delete static_cast<pt::Bvh8*>(h); }
extern "C" int bvh8_walk_built(void* h, const Ray* rays, uint64_t n, Hit* hits, uint64_t* nodes_opened, uint64_t* tris_tested, int threads) {
  const pt::Bvh8& b = *static_cast<pt::Bvh8*>(h);
  float mag = 0.0f;
  for (int a = 0; a < 3; ++a) mag = std::fmax(mag, std::fmax(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
  if (threads < 1) threads = 1;
  std::vector<uint64_t> cn(threads, 0), ct(threads, 0);
  std::vector<int> rc(threads, 0);
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) {
    const uint64_t a = n * (uint64_t)t / threads, e = n * (uint64_t)(t + 1) / threads;
    pool.emplace_back([&, t, a, e] {
      rc[t] = bvh8_walk(b.nodes.data(), b.nodes.size(), b.records(), b.num_records(), (uint32_t)b.record_bytes(), mag, rays + a, e - a, hits + a, &cn[t], &ct[t]);
    });
  }
  for (auto& th : pool) th.join();
  *nodes_opened = 0; *tris_tested = 0;
  for (int t = 0; t < threads; ++t) { if (rc[t]) return rc[t]; *nodes_opened += cn[t]; *tris_tested += ct[t]; }
  return 0;
}
