// bvh_builder.cpp — binned-SAH BVH2 -> SAH-optimal 8-wide collapse -> compressed 80-byte nodes.
// The reference has no BVH code at all (DXR driver builds it, PathTracer/PathTracer.cpp:271-272,338-339); this is the
// MI355X-side replacement. Host C++17, multi-threaded with std::thread.
#include "bvh_builder.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

namespace pt {
namespace {

struct Node2 {
  float lo[3], hi[3];
  uint32_t left;          // children are left, left+1; 0 => leaf
  uint32_t first, count;  // range in the permuted primitive index array
};

struct Prim {
  float lo[3], hi[3], c[3];
};

// Areas and SAH costs are doubles: a single far-away (but finite) vertex makes fp32 areas overflow to +inf, and an
// infinite cost poisons every comparison of the builder and of the collapse.
static inline double half_area(const float* lo, const float* hi) {
  const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
  return dx * dy + dy * dz + dz * dx;
}

#ifndef PT_BVH_BINS
#define PT_BVH_BINS 16
#endif
constexpr int kBins = PT_BVH_BINS;
constexpr uint32_t kParallelThreshold = 1u << 15;

class Builder2 {
 public:
  Builder2(const std::vector<Prim>& prims, std::vector<uint32_t>& order, std::vector<Node2>& nodes, int threads)
      : prims_(prims), order_(order), nodes_(nodes), threads_(threads) {}

  void run() {
    const uint32_t n = (uint32_t)order_.size();
    nodes_.resize(n ? 2 * (size_t)n - 1 : 1);
    next_node_.store(1);
    outstanding_.store(1);
    tasks_.push_back({0, 0, n});
    std::vector<std::thread> pool;
    for (int t = 1; t < threads_; ++t) pool.emplace_back([this] { worker(); });
    worker();
    for (auto& th : pool) th.join();
    nodes_.resize(next_node_.load());
  }

 private:
  struct Task { uint32_t node, begin, end; };

  void worker() {
    for (;;) {
      Task tk;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return !tasks_.empty() || outstanding_.load() == 0; });
        if (tasks_.empty()) return;
        tk = tasks_.back();
        tasks_.pop_back();
      }
      process(tk);
      if (outstanding_.fetch_sub(1) == 1) {
        std::lock_guard<std::mutex> lk(mu_);
        cv_.notify_all();
      }
    }
  }

  void spawn(const Task& t) {
    outstanding_.fetch_add(1);
    {
      std::lock_guard<std::mutex> lk(mu_);
      tasks_.push_back(t);
    }
    cv_.notify_one();
  }

  // Splits [begin,end) of node; returns mid and allocates the two children. Returns false if it became a leaf.
  bool split(const Task& tk, uint32_t& mid, uint32_t& left) {
    Node2& nd = nodes_[tk.node];
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    float clo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (uint32_t i = tk.begin; i < tk.end; ++i) {
      const Prim& p = prims_[order_[i]];
      for (int a = 0; a < 3; ++a) {
        lo[a] = std::min(lo[a], p.lo[a]); hi[a] = std::max(hi[a], p.hi[a]);
        clo[a] = std::min(clo[a], p.c[a]); chi[a] = std::max(chi[a], p.c[a]);
      }
    }
    for (int a = 0; a < 3; ++a) { nd.lo[a] = lo[a]; nd.hi[a] = hi[a]; }
    nd.first = tk.begin;
    nd.count = tk.end - tk.begin;
    nd.left = 0;
    if (nd.count <= 1) return false;

    struct Bin { float lo[3], hi[3]; uint32_t n; };
    Bin bins[3][kBins];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < kBins; ++b) {
        bins[a][b].n = 0;
        for (int k = 0; k < 3; ++k) { bins[a][b].lo[k] = FLT_MAX; bins[a][b].hi[k] = -FLT_MAX; }
      }
    float scale[3];
    for (int a = 0; a < 3; ++a) { float ext = chi[a] - clo[a]; scale[a] = ext > 0.0f ? (float)kBins / ext : 0.0f; }
    for (uint32_t i = tk.begin; i < tk.end; ++i) {
      const Prim& p = prims_[order_[i]];
      for (int a = 0; a < 3; ++a) {
        if (scale[a] == 0.0f) continue;
        int b = (int)((p.c[a] - clo[a]) * scale[a]);
        b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
        Bin& bn = bins[a][b];
        bn.n++;
        for (int k = 0; k < 3; ++k) { bn.lo[k] = std::min(bn.lo[k], p.lo[k]); bn.hi[k] = std::max(bn.hi[k], p.hi[k]); }
      }
    }
    int bestAxis = -1, bestSplit = -1;
    double bestCost = DBL_MAX;
    for (int a = 0; a < 3; ++a) {
      if (scale[a] == 0.0f) continue;
      double rarea[kBins]; uint32_t rcnt[kBins];
      float l3[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, h3[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
      uint32_t c = 0;
      for (int b = kBins - 1; b > 0; --b) {
        for (int k = 0; k < 3; ++k) { l3[k] = std::min(l3[k], bins[a][b].lo[k]); h3[k] = std::max(h3[k], bins[a][b].hi[k]); }
        c += bins[a][b].n;
        rarea[b] = c ? half_area(l3, h3) : 0.0;
        rcnt[b] = c;
      }
      for (int k = 0; k < 3; ++k) { l3[k] = FLT_MAX; h3[k] = -FLT_MAX; }
      c = 0;
      for (int b = 0; b < kBins - 1; ++b) {
        for (int k = 0; k < 3; ++k) { l3[k] = std::min(l3[k], bins[a][b].lo[k]); h3[k] = std::max(h3[k], bins[a][b].hi[k]); }
        c += bins[a][b].n;
        if (c == 0 || rcnt[b + 1] == 0) continue;
        const double cost = half_area(l3, h3) * (double)c + rarea[b + 1] * (double)rcnt[b + 1];
        if (cost < bestCost) { bestCost = cost; bestAxis = a; bestSplit = b; }
      }
    }
    if (bestAxis < 0) {
      mid = tk.begin + nd.count / 2;  // coincident centroids: halve the list
    } else {
      const int a = bestAxis;
      auto it = std::partition(order_.begin() + tk.begin, order_.begin() + tk.end, [&](uint32_t pi) {
        int b = (int)((prims_[pi].c[a] - clo[a]) * scale[a]);
        b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
        return b <= bestSplit;
      });
      mid = (uint32_t)(it - order_.begin());
      if (mid == tk.begin || mid == tk.end) mid = tk.begin + nd.count / 2;
    }
    left = next_node_.fetch_add(2);
    nd.left = left;
    return true;
  }

  void process(const Task& root) {
    std::vector<Task> stack;
    stack.push_back(root);
    while (!stack.empty()) {
      Task tk = stack.back();
      stack.pop_back();
      uint32_t mid, left;
      if (!split(tk, mid, left)) continue;
      Task l{left, tk.begin, mid}, r{left + 1, mid, tk.end};
      // hand large subtrees to other workers; keep going on the other one
      if (threads_ > 1 && (tk.end - tk.begin) > kParallelThreshold) {
        spawn(r);
        stack.push_back(l);
      } else {
        stack.push_back(r);
        stack.push_back(l);
      }
    }
  }

  const std::vector<Prim>& prims_;
  std::vector<uint32_t>& order_;
  std::vector<Node2>& nodes_;
  int threads_;
  std::atomic<uint32_t> next_node_{1};
  std::atomic<int> outstanding_{0};
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<Task> tasks_;
};

// decision record of the collapse DP
enum : uint8_t { D_LEAF = 0, D_INTERNAL = 1, D_DISTRIBUTE = 2 };
static inline uint8_t mk_dec(uint8_t type, int kl, int kr) { return (uint8_t)(type | (kl << 2) | (kr << 5)); }
static inline int dec_type(uint8_t d) { return d & 3; }
static inline int dec_kl(uint8_t d) { return (d >> 2) & 7; }
static inline int dec_kr(uint8_t d) { return (d >> 5) & 7; }

struct ChildRef { uint32_t n2; bool leaf; };

// ---- pairing of edge-sharing triangles ----
// Greedy matching in triangle-id order over an open-addressing table keyed by the (order-independent) hash of an
// edge's two vertex positions. A table hit is only a candidate: the pair is accepted after comparing positions bit
// for bit. partner[i] = the triangle paired with i (the lower id of a pair owns the record) or 0xFFFFFFFF.
static inline uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}
static inline uint64_t vertex_key(const float* v) {
  uint32_t b[3];
  memcpy(b, v, 12);
  return mix64(((uint64_t)b[0] << 32 | b[1]) ^ mix64((uint64_t)b[2] + 0x9e3779b97f4a7c15ull));
}
static inline bool same_vertex(const float* a, const float* b) { return memcmp(a, b, 12) == 0; }

static int shared_vertices(const float* ta, const float* tb) {
  int n = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      if (same_vertex(ta + 3 * i, tb + 3 * j)) { ++n; break; }
  return n;
}

static void pair_triangles(const float* verts, const std::vector<uint32_t>& active, uint32_t n, std::vector<uint32_t>& partner) {
  partner.assign(n, 0xFFFFFFFFu);
  if (active.size() < 2) return;
  size_t cap = 64;
  while (cap < active.size() * 6) cap <<= 1;
  struct Slot { uint32_t tag, tri; };
  std::vector<Slot> table(cap, Slot{0u, 0xFFFFFFFFu});
  const size_t mask = cap - 1;
  for (uint32_t i : active) {
    const float* t = verts + (size_t)i * 9;
    uint64_t vk[3] = {vertex_key(t), vertex_key(t + 3), vertex_key(t + 6)};
    uint64_t ek[3];
    for (int e = 0; e < 3; ++e) {   // order-independent edge key
      const uint64_t a = vk[e], b = vk[(e + 1) % 3];
      ek[e] = mix64((a ^ b) + (a + b) * 0x9e3779b97f4a7c15ull);
    }
    bool paired = false;
    for (int e = 0; e < 3 && !paired; ++e) {
      const uint32_t tag = (uint32_t)(ek[e] >> 32) | 1u;
      for (size_t s = ek[e] & mask;; s = (s + 1) & mask) {
        const Slot sl = table[s];
        if (sl.tri == 0xFFFFFFFFu && sl.tag == 0u) break;            // empty: this edge was never entered
        if (sl.tag == tag && sl.tri != 0xFFFFFFFFu && sl.tri != i && partner[sl.tri] == 0xFFFFFFFFu &&
            shared_vertices(verts + (size_t)sl.tri * 9, t) >= 2) {
          partner[sl.tri] = i;
          partner[i] = sl.tri;
          paired = true;
          break;
        }
      }
    }
    if (paired) continue;
    for (int e = 0; e < 3; ++e) {   // offer this triangle's edges to later ones
      const uint32_t tag = (uint32_t)(ek[e] >> 32) | 1u;
      for (size_t s = ek[e] & mask;; s = (s + 1) & mask) {
        if (table[s].tag == 0u) { table[s] = Slot{tag, i}; break; }
      }
    }
  }
}

}  // namespace

void build_bvh8(const float* verts, uint32_t n, Bvh8& out, int num_threads, int pair_mode) {
  auto t0 = std::chrono::steady_clock::now();
  if (num_threads <= 0) {
    num_threads = (int)std::thread::hardware_concurrency();
    if (const char* e = getenv("OMP_NUM_THREADS")) { int v = atoi(e); if (v > 0) num_threads = std::min(num_threads, v); }
    num_threads = std::max(1, std::min(num_threads, 16));
  }
  out.nodes.clear();
  out.tris.clear();
  out.prims.clear();
  out.num_pairs = 0;
  out.max_depth = 0;
  if (const char* e = getenv("PT_BVH_PAIRS")) pair_mode = atoi(e);   // tuning runs
  // auto: an unpaired hierarchy costs ~56 bytes per triangle (48-byte record + its share of the nodes)
  out.paired = pair_mode > 0 || (pair_mode < 0 && (double)n * 56.0 > 224.0 * 1024.0 * 1024.0);

  // ---- active triangles, scene bounds and the conservative pad (see DESIGN.md "box conservativeness") ----
  // A triangle with a NaN or infinite vertex can never be hit (every comparison of the intersection test fails), so it
  // is left out of the hierarchy — DXR calls such primitives inactive. Its id stays reserved.
  std::vector<uint32_t> order;
  order.reserve(n);
  float slo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, shi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (uint32_t i = 0; i < n; ++i) {
    const float* v = verts + (size_t)i * 9;
    bool finite = true;
    for (int k = 0; k < 9; ++k) finite = finite && std::isfinite(v[k]);
    if (!finite) continue;
    order.push_back(i);
    for (int k = 0; k < 9; ++k) { slo[k % 3] = std::min(slo[k % 3], v[k]); shi[k % 3] = std::max(shi[k % 3], v[k]); }
  }
  if (order.empty()) { for (int a = 0; a < 3; ++a) { slo[a] = 0; shi[a] = 0; } }
  float mag = 0.0f;
  for (int a = 0; a < 3; ++a) mag = std::max(mag, std::max(std::fabs(slo[a]), std::fabs(shi[a])));
  const float pad = 4e-6f * std::max(mag, 1e-3f);
  for (int a = 0; a < 3; ++a) { out.lo[a] = slo[a]; out.hi[a] = shi[a]; }

  if (order.empty()) {
    PtBvh8Node root;
    memset(&root, 0, sizeof(root));
    root.e[0] = root.e[1] = root.e[2] = 127;
    for (int a = 0; a < 3; ++a) for (int s = 0; s < 8; ++s) { root.qlo[a][s] = 255; root.qhi[a][s] = 0; }
    out.nodes.push_back(root);
    out.max_depth = 1;
    return;
  }

  // ---- primitives: edge-sharing triangles in pairs, the rest alone ----
  std::vector<uint32_t> partner;
  if (out.paired) pair_triangles(verts, order, n, partner);
  else partner.assign(n, 0xFFFFFFFFu);
  struct PrimRef { uint32_t a, b; };
  std::vector<PrimRef> refs;
  refs.reserve(order.size());
  for (uint32_t i : order) {
    if (partner[i] == 0xFFFFFFFFu) refs.push_back({i, 0xFFFFFFFFu});
    else if (partner[i] > i) refs.push_back({i, partner[i]});
  }
  const uint32_t np = (uint32_t)refs.size();
  std::vector<Prim> prims(np);
  std::vector<float> weight(np);      // SAH weight of a record: one triangle test, or two that share their set-up
  for (uint32_t k = 0; k < np; ++k) {
    Prim& p = prims[k];
    for (int a = 0; a < 3; ++a) { p.lo[a] = FLT_MAX; p.hi[a] = -FLT_MAX; }
    for (int h = 0; h < 2; ++h) {
      const uint32_t t = h ? refs[k].b : refs[k].a;
      if (t == 0xFFFFFFFFu) continue;
      const float* v = verts + (size_t)t * 9;
      for (int a = 0; a < 3; ++a) {
        p.lo[a] = std::min(p.lo[a], std::min(v[a], std::min(v[3 + a], v[6 + a])));
        p.hi[a] = std::max(p.hi[a], std::max(v[a], std::max(v[3 + a], v[6 + a])));
      }
    }
    for (int a = 0; a < 3; ++a) {
      p.c[a] = 0.5f * p.lo[a] + 0.5f * p.hi[a];   // not 0.5*(lo+hi): the sum can overflow for |coordinates| near FLT_MAX
      p.lo[a] -= pad;
      p.hi[a] += pad;
    }
    weight[k] = refs[k].b == 0xFFFFFFFFu ? 1.0f : 1.6f;
    if (refs[k].b != 0xFFFFFFFFu) out.num_pairs++;
  }
  order.resize(np);
  for (uint32_t k = 0; k < np; ++k) order[k] = k;
  std::vector<Node2> n2;
  Builder2(prims, order, n2, num_threads).run();
  const uint32_t nn = (uint32_t)n2.size();

  // ---- SAH-optimal collapse (dynamic programme over "up to i roots", i = 1..7) ----
  // SAH constants of the collapse: cost of one 8-wide node step vs one triangle test. PT_BVH_CPRIM overrides the
  // triangle cost for tuning runs (tools/tune_bvh.sh).
  double Cnode = 1.0, Cprim = 0.3;
  if (const char* e = getenv("PT_BVH_CPRIM")) { const double v = atof(e); if (v > 0.0) Cprim = v; }
  std::vector<double> cost((size_t)nn * 7);
  std::vector<uint8_t> dec((size_t)nn * 7);
  std::vector<uint8_t> isplit(nn, 0);
  std::vector<float> wsum(nn, 0.0f);   // SAH weight of the records below a node (children are allocated after their parent)
  for (uint32_t idx = nn; idx-- > 0;) {
    const Node2& nd = n2[idx];
    const double A = half_area(nd.lo, nd.hi);
    double* c = &cost[(size_t)idx * 7];
    uint8_t* d = &dec[(size_t)idx * 7];
    if (nd.left == 0) {
      for (uint32_t i = 0; i < nd.count; ++i) wsum[idx] += weight[order[nd.first + i]];
      for (int i = 0; i < 7; ++i) { c[i] = A * Cprim * (double)wsum[idx]; d[i] = mk_dec(D_LEAF, 0, 0); }
      continue;
    }
    wsum[idx] = wsum[nd.left] + wsum[nd.left + 1];
    const double* cl = &cost[(size_t)nd.left * 7];
    const double* cr = &cost[(size_t)(nd.left + 1) * 7];
    double cdist[9]; int ksplit[9];
    for (int j = 2; j <= 8; ++j) {
      double best = DBL_MAX; int bk = 1;
      for (int k = 1; k < j; ++k) {
        if (k > 7 || j - k > 7) continue;
        const double v = cl[k - 1] + cr[j - k - 1];
        if (v < best) { best = v; bk = k; }
      }
      cdist[j] = best; ksplit[j] = bk;
    }
    const double cleaf = nd.count <= PT_BVH8_MAX_LEAF_TRIS ? A * Cprim * (double)wsum[idx] : DBL_MAX;
    const double cint = cdist[8] + A * Cnode;
    isplit[idx] = (uint8_t)ksplit[8];
    if (cleaf <= cint) { c[0] = cleaf; d[0] = mk_dec(D_LEAF, 0, 0); }
    else { c[0] = cint; d[0] = mk_dec(D_INTERNAL, 0, 0); }
    for (int i = 1; i < 7; ++i) {
      if (cdist[i + 1] < c[i - 1]) { c[i] = cdist[i + 1]; d[i] = mk_dec(D_DISTRIBUTE, ksplit[i + 1], i + 1 - ksplit[i + 1]); }
      else { c[i] = c[i - 1]; d[i] = d[i - 1]; }
    }
  }
  out.sah_cost = (float)cost[0];

  // ---- emit 8-wide nodes breadth-first ----
  struct Work { uint32_t n2; uint32_t out; uint32_t depth; };
  std::deque<Work> queue;
  out.nodes.reserve(nn / 4 + 16);
  if (out.paired) out.prims.reserve(np); else out.tris.reserve(np);
  out.nodes.emplace_back();
  queue.push_back({0, 0, 1});
  std::vector<ChildRef> kids;
  std::vector<std::pair<uint32_t, int>> cstack;
  while (!queue.empty()) {
    Work w = queue.front();
    queue.pop_front();
    out.max_depth = std::max(out.max_depth, w.depth);
    const Node2& nd = n2[w.n2];
    kids.clear();
    const bool root_is_leaf = (nd.left == 0) || (w.out == 0 && dec_type(dec[(size_t)w.n2 * 7]) == D_LEAF);
    if (root_is_leaf) {
      kids.push_back({w.n2, true});  // tiny scene: the root holds one leaf child with all (<=3) triangles
    } else {
      cstack.clear();
      const int k = isplit[w.n2];
      cstack.push_back({nd.left + 1, 8 - k});
      cstack.push_back({nd.left, k});
      while (!cstack.empty()) {
        auto [m, budget] = cstack.back();
        cstack.pop_back();
        uint8_t d = dec[(size_t)m * 7 + (budget - 1)];
        if (dec_type(d) == D_LEAF) kids.push_back({m, true});
        else if (dec_type(d) == D_INTERNAL) kids.push_back({m, false});
        else { cstack.push_back({n2[m].left + 1, dec_kr(d)}); cstack.push_back({n2[m].left, dec_kl(d)}); }
      }
    }
    const int nk = (int)kids.size();  // 1..8

    // octant-aware slot assignment: child in slot s is the one lying farthest along diagonal d_s
    // (component a of d_s is -1 when bit a of s is set), chosen greedily by descending score.
    float pc[3];
    for (int a = 0; a < 3; ++a) pc[a] = 0.5f * (nd.lo[a] + nd.hi[a]);
    float score[8][8];
    for (int c = 0; c < nk; ++c) {
      const Node2& ch = n2[kids[c].n2];
      float off[3];
      for (int a = 0; a < 3; ++a) off[a] = 0.5f * (ch.lo[a] + ch.hi[a]) - pc[a];
      for (int s = 0; s < 8; ++s) {
        float v = 0.0f;
        for (int a = 0; a < 3; ++a) v += ((s >> a) & 1) ? -off[a] : off[a];
        score[c][s] = v;
      }
    }
    int slot_child[8];
    for (int s = 0; s < 8; ++s) slot_child[s] = -1;
    bool child_done[8] = {false, false, false, false, false, false, false, false};
    for (int it = 0; it < nk; ++it) {
      float best = -FLT_MAX; int bc = -1, bs = -1;
      for (int c = 0; c < nk; ++c) {
        if (child_done[c]) continue;
        for (int s = 0; s < 8; ++s) {
          if (slot_child[s] >= 0) continue;
          if (score[c][s] > best) { best = score[c][s]; bc = c; bs = s; }
        }
      }
      slot_child[bs] = bc;
      child_done[bc] = true;
    }

    // quantisation grid
    PtBvh8Node node;
    memset(&node, 0, sizeof(node));
    int e[3];
    for (int a = 0; a < 3; ++a) {
      node.p[a] = nd.lo[a];
      float ext = nd.hi[a] - nd.lo[a];
      int ex = -126;
      if (ext > 0.0f) {
        int fe;
        std::frexp(ext / 255.0f, &fe);  // ext/255 = m * 2^fe, m in [0.5,1)  => 2^fe >= ext/255
        ex = fe;
      }
      ex = std::max(ex, -126);
      e[a] = ex;
    }
    for (bool ok = false; !ok;) {
      ok = true;
      for (int s = 0; s < 8 && ok; ++s) {
        if (slot_child[s] < 0) continue;
        const Node2& ch = n2[kids[slot_child[s]].n2];
        for (int a = 0; a < 3; ++a) {
          const double sc = std::ldexp(1.0, e[a]);
          double ql = std::floor(((double)ch.lo[a] - (double)node.p[a]) / sc);
          double qh = std::ceil(((double)ch.hi[a] - (double)node.p[a]) / sc);
          if (ql < 0.0) ql = 0.0;
          if (qh > 255.0) { e[a]++; ok = false; break; }
          if (qh < ql) qh = ql;
          node.qlo[a][s] = (uint8_t)ql;
          node.qhi[a][s] = (uint8_t)qh;
        }
      }
    }
    for (int a = 0; a < 3; ++a) node.e[a] = (uint8_t)(e[a] + 127);
    for (int s = 0; s < 8; ++s)
      if (slot_child[s] < 0)
        for (int a = 0; a < 3; ++a) { node.qlo[a][s] = 255; node.qhi[a][s] = 0; }

    // children: internal ones contiguous in slot order, leaf triangles appended in slot order
    node.child_base = (uint32_t)out.nodes.size();
    node.tri_base = (uint32_t)out.num_records();
    uint32_t ninternal = 0, tri_off = 0;
    for (int s = 0; s < 8; ++s) {
      if (slot_child[s] < 0) { node.meta[s] = 0; continue; }
      const ChildRef& cr = kids[slot_child[s]];
      const Node2& ch = n2[cr.n2];
      if (!cr.leaf) {
        node.imask |= (uint8_t)(1u << s);
        node.meta[s] = (uint8_t)((1u << 5) | (24u + (uint32_t)s));
        queue.push_back({cr.n2, node.child_base + ninternal, w.depth + 1});
        ninternal++;
      } else {
        const uint32_t cnt = ch.count;  // 1..3
        const uint32_t unary = cnt == 1 ? 1u : (cnt == 2 ? 3u : 7u);
        node.meta[s] = (uint8_t)((unary << 5) | tri_off);
        for (uint32_t i = 0; i < cnt; ++i) {
          const PrimRef& pr = refs[order[ch.first + i]];
          const float* v = verts + (size_t)pr.a * 9;
          if (!out.paired) {
            PtTriRecord t3;
            memcpy(t3.v0, v, 12); memcpy(t3.v1, v + 3, 12); memcpy(t3.v2, v + 6, 12);
            t3.prim = pr.a; t3.pad0 = 0; t3.pad1 = 0;
            out.tris.push_back(t3);
            continue;
          }
          PtPrimRecord tr;
          memcpy(tr.v0, v, 12); memcpy(tr.v1, v + 3, 12); memcpy(tr.v2, v + 6, 12);
          memcpy(tr.v3, v + 6, 12);
          tr.prim0 = pr.a; tr.prim1 = 0xFFFFFFFFu; tr.code = 0u; tr.pad = 0u;
          if (pr.b != 0xFFFFFFFFu) {
            // triangle b in ITS vertex order, each vertex as an index into {v0, v1, v2, v3}; v3 = its one vertex
            // that triangle a does not have (pairs share at least two vertices, so there is at most one such)
            const float* w = verts + (size_t)pr.b * 9;
            uint32_t code = 0u;
            for (int k = 0; k < 3; ++k) {
              uint32_t idx = 3u;
              for (int m = 0; m < 3; ++m)
                if (same_vertex(w + 3 * k, v + 3 * m)) { idx = (uint32_t)m; break; }
              if (idx == 3u) memcpy(tr.v3, w + 3 * k, 12);
              code |= idx << (2 * k);
            }
            tr.prim1 = pr.b;
            tr.code = code;
          }
          out.prims.push_back(tr);
        }
        tri_off += cnt;
      }
    }
    out.nodes.resize(out.nodes.size() + ninternal);
    out.nodes[w.out] = node;
  }
  auto t1 = std::chrono::steady_clock::now();
  out.build_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
}

}  // namespace pt
