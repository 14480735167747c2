// bvh_builder.cpp — binned-SAH BVH2 -> SAH-optimal 8-wide collapse -> compressed 80-byte nodes.
// The reference has no BVH code at all (DXR driver builds it, PathTracer/PathTracer.cpp:271-272,338-339); this is the
// MI355X-side replacement. Host C++17, multi-threaded with std::thread.
#include "bvh_builder.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
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

// A reference: (part of) a primitive with the box of that part. Without spatial splits every primitive has exactly one;
// a spatial split cuts a reference in two, each half keeping the box of its side of the plane.
struct Ref {
  float lo[3], hi[3];
  uint32_t prim;
};

// the one or two triangles of a primitive record (b = 0xFFFFFFFF: a single triangle)
struct PrimRef { uint32_t a, b; };

// Areas and SAH costs are doubles: a single far-away (but finite) vertex makes fp32 areas overflow to +inf, and an
// infinite cost poisons every comparison of the builder and of the collapse.
static inline double half_area(const float* lo, const float* hi) {
  const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
  return dx * dy + dy * dz + dz * dx;
}

struct Box3 {
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  void grow(const float* l, const float* h) {
    for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], l[k]); hi[k] = std::max(hi[k], h[k]); }
  }
  void grow(const Box3& b) { grow(b.lo, b.hi); }
  bool valid() const { return lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]; }
  double area() const { return valid() ? half_area(lo, hi) : 0.0; }
};

#ifndef PT_BVH_REINSERT_PASSES_DEFAULT
#define PT_BVH_REINSERT_PASSES_DEFAULT 0
#endif
#ifndef PT_BVH_BINS
#define PT_BVH_BINS 16
#endif
constexpr int kBins = PT_BVH_BINS;
constexpr uint32_t kParallelThreshold = 1u << 15;   // subtrees above this size become tasks of their own
constexpr size_t kParallelPass = 1u << 19;          // passes over more references than this are split over the threads
constexpr int kMaxDepth2 = 160;                     // binary levels after which spatial splits stop (the 8-wide tree is <= 62 deep)

// Bounds of (triangle ∩ {lo <= x[axis] <= hi}) grown into (blo, bhi): the vertices inside the slab and the points where
// the edges cross its two planes are exactly the vertices of the clipped polygon. Returns false when nothing is inside.
static bool clip_triangle(const float* v, int axis, double lo, double hi, double* blo, double* bhi) {
  bool any = false;
  auto grow = [&](const double* p) {
    for (int k = 0; k < 3; ++k) { blo[k] = std::min(blo[k], p[k]); bhi[k] = std::max(bhi[k], p[k]); }
    any = true;
  };
  for (int i = 0; i < 3; ++i) {
    const float* p = v + 3 * i;
    const float* q = v + 3 * ((i + 1) % 3);
    const double pa = p[axis], qa = q[axis];
    if (pa >= lo && pa <= hi) { const double pp[3] = {p[0], p[1], p[2]}; grow(pp); }
    for (int side = 0; side < 2; ++side) {
      const double pl = side ? hi : lo;
      if ((pa < pl && qa > pl) || (pa > pl && qa < pl)) {
        const double t = (pl - pa) / (qa - pa);
        double x[3];
        for (int k = 0; k < 3; ++k) x[k] = (double)p[k] + t * ((double)q[k] - (double)p[k]);
        x[axis] = pl;
        grow(x);
      }
    }
  }
  return any;
}

static inline float round_down(double x) { float f = (float)x; return (double)f > x ? std::nextafterf(f, -FLT_MAX) : f; }
static inline float round_up(double x) { float f = (float)x; return (double)f < x ? std::nextafterf(f, FLT_MAX) : f; }

// Top-down builder over references: binned SAH object splits, and (Stich, Friedrich, Dietrich, "Spatial Splits in Bounding
// Volume Hierarchies", HPG 2009) binned SPATIAL splits where the two sides of the best object split overlap: a long or
// diagonal triangle is then referenced from several leaves, each with the tight box of its piece, instead of dragging one
// loose box through the hierarchy. Leaves hold one reference. A triangle reached through several leaves is simply tested
// more than once: the closest hit with its lower-id tie-break does not depend on that, so hit records cannot change.
// Deterministic for any thread count: the extra-reference budget is handed down the tree in proportion to the box areas
// of each side's references, never taken from a shared counter.
class SplitBuilder {
 public:
  SplitBuilder(const float* verts, const std::vector<PrimRef>& prims, float pad, std::vector<Node2>& nodes, int threads, double alpha, double gain)
      : verts_(verts), prims_(prims), pad_(pad), nodes_(nodes), threads_(threads), alpha_(alpha), gain_(gain) {
    if (const char* e = getenv("PT_BVH_SPLIT_SINGLE_AXIS")) single_axis_below_ = atoi(e);
  }

  // refs: one per primitive; extra: how many more references spatial splits may create in total
  void run(std::vector<Ref>&& refs, uint64_t extra) {
    const size_t n = refs.size();
    nodes_.resize(n ? 2 * (n + extra) : 1);
    next_node_.store(1);
    outstanding_.store(1);
    Box3 b;
    for (const Ref& r : refs) b.grow(r.lo, r.hi);
    root_area_ = std::max(b.area(), 1e-300);
    tasks_.push_back(Task{0, 0, extra, std::move(refs)});
    std::vector<std::thread> pool;
    for (int t = 1; t < threads_; ++t) pool.emplace_back([this] { worker(); });
    worker();
    for (auto& th : pool) th.join();
    nodes_.resize(next_node_.load());
  }
  uint64_t spatial_splits() const { return n_spatial_.load(); }

 private:
  struct Task { uint32_t node; int depth; uint64_t budget; std::vector<Ref> refs; };
  struct Bin { Box3 box; uint32_t n = 0, enter = 0, leave = 0; };

  void worker() {
    for (;;) {
      Task tk;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return !tasks_.empty() || outstanding_.load() == 0; });
        if (tasks_.empty()) return;
        tk = std::move(tasks_.back());
        tasks_.pop_back();
      }
      process(std::move(tk));
      if (outstanding_.fetch_sub(1) == 1) {
        std::lock_guard<std::mutex> lk(mu_);
        cv_.notify_all();
      }
    }
  }

  void spawn(Task&& t) {
    outstanding_.fetch_add(1);
    {
      std::lock_guard<std::mutex> lk(mu_);
      tasks_.push_back(std::move(t));
    }
    cv_.notify_one();
  }

  // f(begin, end, slot) over [0, n) in `slots` contiguous chunks; the chunks of a large pass run on their own threads
  // (only the top few nodes of a big scene are that large, and the pool is idle while they are processed)
  template <class F>
  int chunked(size_t n, F&& f) const {
    const int slots = (threads_ > 1 && n >= kParallelPass) ? threads_ : 1;
    if (slots == 1) { f((size_t)0, n, 0); return 1; }
    std::vector<std::thread> th;
    for (int s = 1; s < slots; ++s) th.emplace_back([&, s] { f(n * s / slots, n * (s + 1) / slots, s); });
    f((size_t)0, n / slots, 0);
    for (auto& t : th) t.join();
    return slots;
  }

  // the part of a reference's primitive inside lo <= x[axis] <= hi, as a box (within the reference's own box); false = none
  bool clip_ref(const Ref& r, int axis, double lo, double hi, Box3& out) const {
    double blo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, bhi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    const PrimRef& pr = prims_[r.prim];
    bool any = clip_triangle(verts_ + (size_t)pr.a * 9, axis, lo, hi, blo, bhi);
    if (pr.b != 0xFFFFFFFFu) any = clip_triangle(verts_ + (size_t)pr.b * 9, axis, lo, hi, blo, bhi) || any;
    if (!any) return false;
    for (int k = 0; k < 3; ++k) {
      out.lo[k] = std::max(r.lo[k], round_down(blo[k]));
      out.hi[k] = std::min(r.hi[k], round_up(bhi[k]));
    }
    return out.valid();
  }

  static inline int bin_of(float x, float lo, float scale) {
    const int b = (int)((x - lo) * scale);
    return b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
  }

  void process(Task&& root) {
    std::vector<Task> stack;
    stack.push_back(std::move(root));
    while (!stack.empty()) {
      Task tk = std::move(stack.back());
      stack.pop_back();
      Task l, r;
      if (!split(tk, l, r)) continue;
      // hand large subtrees to other workers; keep going on the other one
      if (threads_ > 1 && l.refs.size() + r.refs.size() > kParallelThreshold) {
        spawn(std::move(r));
        stack.push_back(std::move(l));
      } else {
        stack.push_back(std::move(r));
        stack.push_back(std::move(l));
      }
    }
  }

  // Splits the references of tk between two new children; returns false if the node became a leaf.
  bool split(Task& tk, Task& L, Task& R) {
    std::vector<Ref>& refs = tk.refs;
    const size_t n = refs.size();
    Node2& nd = nodes_[tk.node];
    // ---- bounds of the boxes and of their centres ----
    struct Bounds { Box3 box, cen; };
    std::vector<Bounds> pb((size_t)std::max(1, threads_));
    const int sl0 = chunked(n, [&](size_t b, size_t e, int s) {
      Bounds w;
      for (size_t i = b; i < e; ++i) {
        const Ref& f = refs[i];
        w.box.grow(f.lo, f.hi);
        float c[3];
        for (int a = 0; a < 3; ++a) c[a] = 0.5f * f.lo[a] + 0.5f * f.hi[a];   // not 0.5*(lo+hi): the sum can overflow near FLT_MAX
        w.cen.grow(c, c);
      }
      pb[s] = w;
    });
    Bounds bb = pb[0];
    for (int s = 1; s < sl0; ++s) { bb.box.grow(pb[s].box); bb.cen.grow(pb[s].cen); }
    // References carry the exact (outward-rounded) boxes of their pieces; the conservative pad goes onto the NODE boxes only.
    // Padding the references themselves would wipe out their centres where the pad exceeds a coordinate (the pad is
    // relative to the scene's magnitude: a scene spanning 1e36 pads by 4e30, and every centre of a small triangle reads 0).
    for (int a = 0; a < 3; ++a) { nd.lo[a] = bb.box.lo[a] - pad_; nd.hi[a] = bb.box.hi[a] + pad_; }
    nd.left = 0;
    nd.count = 1;
    if (n <= 1) {
      nd.first = n ? refs[0].prim : 0u;
      return false;
    }

    // ---- object split: binned SAH over the centres ----
    float oscale[3];
    for (int a = 0; a < 3; ++a) { const float ext = bb.cen.hi[a] - bb.cen.lo[a]; oscale[a] = ext > 0.0f ? (float)kBins / ext : 0.0f; }
    std::vector<std::array<Bin, 3 * kBins>> pbin((size_t)std::max(1, threads_));
    const int sl1 = chunked(n, [&](size_t b, size_t e, int s) {
      std::array<Bin, 3 * kBins>& bins = pbin[s];
      for (Bin& x : bins) x = Bin();
      for (size_t i = b; i < e; ++i) {
        const Ref& f = refs[i];
        for (int a = 0; a < 3; ++a) {
          if (oscale[a] == 0.0f) continue;
          Bin& bn = bins[a * kBins + bin_of(0.5f * f.lo[a] + 0.5f * f.hi[a], bb.cen.lo[a], oscale[a])];
          bn.n++;
          bn.box.grow(f.lo, f.hi);
        }
      }
    });
    for (int s = 1; s < sl1; ++s)
      for (int k = 0; k < 3 * kBins; ++k) { pbin[0][k].n += pbin[s][k].n; pbin[0][k].box.grow(pbin[s][k].box); }
    const std::array<Bin, 3 * kBins>& obins = pbin[0];
    int oAxis = -1, oSplit = -1;
    double oCost = DBL_MAX;
    Box3 oL, oR;
    for (int a = 0; a < 3; ++a) {
      if (oscale[a] == 0.0f) continue;
      Box3 rbox[kBins]; uint32_t rcnt[kBins];
      Box3 acc; uint32_t c = 0;
      for (int b = kBins - 1; b > 0; --b) {
        if (obins[a * kBins + b].n) acc.grow(obins[a * kBins + b].box);
        c += obins[a * kBins + b].n;
        rbox[b] = acc; rcnt[b] = c;
      }
      acc = Box3(); c = 0;
      for (int b = 0; b < kBins - 1; ++b) {
        if (obins[a * kBins + b].n) acc.grow(obins[a * kBins + b].box);
        c += obins[a * kBins + b].n;
        if (c == 0 || rcnt[b + 1] == 0) continue;
        const double cost = acc.area() * (double)c + rbox[b + 1].area() * (double)rcnt[b + 1];
        if (cost < oCost) { oCost = cost; oAxis = a; oSplit = b; oL = acc; oR = rbox[b + 1]; }
      }
    }

    // ---- spatial split: only where the object split's sides overlap noticeably (relative to the root, Stich et al. eq. 3) ----
    int sAxis = -1, sSplit = -1;
    double sCost = DBL_MAX;
    uint32_t sNL = 0, sNR = 0;
    double splane_lo[3], swidth[3];
    bool try_spatial = tk.budget > 0 && tk.depth < kMaxDepth2;
    if (oAxis < 0) {   // coincident centres (the two triangles of a flat quad): the list is halved, both halves keep the box
      oCost = bb.box.area() * (double)n;
      try_spatial = try_spatial && bb.box.area() > alpha_ * root_area_;
    }
    if (try_spatial && oAxis >= 0) {
      Box3 ov;
      for (int k = 0; k < 3; ++k) { ov.lo[k] = std::max(oL.lo[k], oR.lo[k]); ov.hi[k] = std::min(oL.hi[k], oR.hi[k]); }
      try_spatial = ov.valid() && ov.area() > alpha_ * root_area_;
    }
    if (try_spatial) {
      float sscale[3];
      int longest = 0;
      for (int a = 0; a < 3; ++a) {
        const double ext = (double)bb.box.hi[a] - bb.box.lo[a];
        splane_lo[a] = bb.box.lo[a];
        swidth[a] = ext / kBins;
        sscale[a] = ext > 0.0 && std::isfinite(ext) ? (float)(kBins / ext) : 0.0f;
        if (ext > (double)bb.box.hi[longest] - bb.box.lo[longest]) longest = a;
      }
      // Small nodes (where most of the build's reference visits happen, and where a reference spans most of the bins)
      // are only cut across their longest axis: a third of the clipping work
      if (n <= (size_t)single_axis_below_)
        for (int a = 0; a < 3; ++a) if (a != longest) sscale[a] = 0.0f;
      const int sl2 = chunked(n, [&](size_t b, size_t e, int s) {
        std::array<Bin, 3 * kBins>& bins = pbin[s];
        for (Bin& x : bins) x = Bin();
        for (size_t i = b; i < e; ++i) {
          const Ref& f = refs[i];
          for (int a = 0; a < 3; ++a) {
            if (sscale[a] == 0.0f) continue;
            const int b0 = bin_of(f.lo[a], bb.box.lo[a], sscale[a]), b1 = bin_of(f.hi[a], bb.box.lo[a], sscale[a]);
            if (b0 == b1) {
              Bin& bn = bins[a * kBins + b0];
              bn.box.grow(f.lo, f.hi); bn.enter++; bn.leave++;
              continue;
            }
            int first = -1, last = -1;
            for (int k = b0; k <= b1; ++k) {
              Box3 piece;
              const double plo = k == b0 ? -DBL_MAX : splane_lo[a] + swidth[a] * k, phi = k == b1 ? DBL_MAX : splane_lo[a] + swidth[a] * (k + 1);
              if (!clip_ref(f, a, plo, phi, piece)) continue;
              bins[a * kBins + k].box.grow(piece);
              if (first < 0) first = k;
              last = k;
            }
            if (first < 0) { first = last = b0; bins[a * kBins + b0].box.grow(f.lo, f.hi); }   // (cannot happen: the primitive lies inside its box)
            bins[a * kBins + first].enter++;
            bins[a * kBins + last].leave++;
          }
        }
      });
      for (int s = 1; s < sl2; ++s)
        for (int k = 0; k < 3 * kBins; ++k) {
          pbin[0][k].enter += pbin[s][k].enter; pbin[0][k].leave += pbin[s][k].leave; pbin[0][k].box.grow(pbin[s][k].box);
        }
      const std::array<Bin, 3 * kBins>& sb = pbin[0];
      for (int a = 0; a < 3; ++a) {
        if (sscale[a] == 0.0f) continue;
        Box3 rbox[kBins]; uint32_t rcnt[kBins];
        Box3 acc; uint32_t c = 0;
        for (int b = kBins - 1; b > 0; --b) {
          acc.grow(sb[a * kBins + b].box);
          c += sb[a * kBins + b].leave;
          rbox[b] = acc; rcnt[b] = c;
        }
        acc = Box3(); c = 0;
        for (int b = 0; b < kBins - 1; ++b) {
          acc.grow(sb[a * kBins + b].box);
          c += sb[a * kBins + b].enter;
          if (c == 0 || rcnt[b + 1] == 0) continue;
          const double cost = acc.area() * (double)c + rbox[b + 1].area() * (double)rcnt[b + 1];
          if (cost < sCost) { sCost = cost; sAxis = a; sSplit = b; sNL = c; sNR = rcnt[b + 1]; }
        }
      }
    }

    // ---- partition ----
    std::vector<Ref> left, right;
    uint64_t used = 0;
    bool done = false;
    if (sAxis >= 0 && sCost < gain_ * oCost && (uint64_t)sNL + sNR - n <= tk.budget && sNL < n + (n >> 1) && sNR < n + (n >> 1)) {
      const int a = sAxis;
      const double pos = splane_lo[a] + swidth[a] * (sSplit + 1);
      left.reserve(sNL); right.reserve(sNR);
      for (const Ref& f : refs) {
        if ((double)f.hi[a] <= pos) { left.push_back(f); continue; }
        if ((double)f.lo[a] >= pos) { right.push_back(f); continue; }
        Ref fl = f, fr = f;
        Box3 bl, br;
        const bool hl = clip_ref(f, a, -DBL_MAX, pos, bl), hr = clip_ref(f, a, pos, DBL_MAX, br);
        if (hl) { for (int k = 0; k < 3; ++k) { fl.lo[k] = bl.lo[k]; fl.hi[k] = bl.hi[k]; } }
        if (hr) { for (int k = 0; k < 3; ++k) { fr.lo[k] = br.lo[k]; fr.hi[k] = br.hi[k]; } }
        if (hl && hr) { left.push_back(fl); right.push_back(fr); }
        else if (hl) left.push_back(f);
        else right.push_back(f);
      }
      if (!left.empty() && !right.empty() && left.size() + right.size() - n <= tk.budget) {
        used = left.size() + right.size() - n;
        n_spatial_.fetch_add(1);
        done = true;
      } else {
        left.clear(); right.clear();
      }
    }
    if (!done) {
      left.reserve(n / 2 + 1); right.reserve(n / 2 + 1);
      if (oAxis >= 0) {
        const int a = oAxis;
        for (const Ref& f : refs)
          (bin_of(0.5f * f.lo[a] + 0.5f * f.hi[a], bb.cen.lo[a], oscale[a]) <= oSplit ? left : right).push_back(f);
      }
      if (left.empty() || right.empty()) {   // coincident centres: halve the list
        left.assign(refs.begin(), refs.begin() + n / 2);
        right.assign(refs.begin() + n / 2, refs.end());
      }
    }
    std::vector<Ref>().swap(refs);
    // the unused part of the budget goes down in proportion to the box areas on each side (where the loose boxes are)
    uint64_t bl = 0, br = 0;
    const uint64_t rest = tk.budget - used;
    if (rest > 0) {
      double al = 0.0, ar = 0.0;
      for (const Ref& f : left) al += half_area(f.lo, f.hi);
      for (const Ref& f : right) ar += half_area(f.lo, f.hi);
      const double frac = (al + ar) > 0.0 && std::isfinite(al + ar) ? al / (al + ar) : 0.5;
      bl = (uint64_t)((double)rest * frac);
      if (bl > rest) bl = rest;
      br = rest - bl;
    }
    const uint32_t li = next_node_.fetch_add(2);
    nd.left = li;
    L = Task{li, tk.depth + 1, bl, std::move(left)};
    R = Task{li + 1, tk.depth + 1, br, std::move(right)};
    return true;
  }

  const float* verts_;
  const std::vector<PrimRef>& prims_;
  float pad_;
  std::vector<Node2>& nodes_;
  int threads_;
  double alpha_;
  int single_axis_below_ = 32;
  double gain_;   // a spatial split must cost less than gain_ x the best object split
  double root_area_ = 1.0;
  std::atomic<uint32_t> next_node_{1};
  std::atomic<int> outstanding_{0};
  std::atomic<uint64_t> n_spatial_{0};
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<Task> tasks_;
};

// Insertion-based optimisation of the finished binary tree (after Bittner, Hapala, Havran: "Fast insertion-based optimization
// of bounding volume hierarchies", CGF 2013). A top-down builder fixes every split before it has seen what happens below it;
// here the nodes with the largest boxes are taken out again — node N and its parent leave the tree, N's sibling moves up —
// and N's two subtrees are put back, each at the place where it adds the least surface area to the tree (branch and bound
// over "area added to the ancestors so far", cheapest first). The tree's SAH cost can only fall: putting a subtree back
// where it came from is always among the candidates. Sequential and deterministic; hit records cannot depend on it.
// Two forms, which find different trees (build_bvh8 keeps whichever gives the best 8-wide tree):
//   careful = false: every pass works on the same slice (the largest nodes), and an operation may leave the area sum higher
//                    than it found it (the second subtree goes back into a tree the first one changed);
//   careful = true:  such an operation is undone, so the area sum never grows, and pass p works on the p-th slice by area.
// Returns false when it gave up after the first pass (the tree is left as it came).
static bool optimize_by_reinsertion(std::vector<Node2>& nodes, int passes, double fraction, size_t max_per_pass, bool careful) {
  const uint32_t n = (uint32_t)nodes.size();
  if (n < 64u || passes <= 0) return false;
  constexpr uint32_t NONE = 0xFFFFFFFFu;
  {   // the parallel builder hands out node ids in the order its threads get there; ties below are broken by id, so the
      // nodes are first renumbered breadth-first (a function of the topology alone)
    std::vector<Node2> bfs(n);
    std::vector<uint32_t> old_of(n);
    old_of[0] = 0;
    uint32_t next = 1;
    for (uint32_t i = 0; i < n; ++i) {
      bfs[i] = nodes[old_of[i]];
      if (bfs[i].left) { old_of[next] = bfs[i].left; old_of[next + 1] = bfs[i].left + 1; bfs[i].left = next; next += 2; }
    }
    nodes.swap(bfs);
  }
  std::vector<uint32_t> parent(n, NONE), c0(n, NONE), c1(n, NONE);
  std::vector<double> area(n);
  for (uint32_t i = 0; i < n; ++i) {
    area[i] = half_area(nodes[i].lo, nodes[i].hi);
    if (nodes[i].left) { c0[i] = nodes[i].left; c1[i] = nodes[i].left + 1; parent[c0[i]] = i; parent[c1[i]] = i; }
  }
  uint32_t root = 0;
  double acc = 0.0;   // change of the area sum over the internal nodes since the current operation began
  auto refit_up = [&](uint32_t i) {   // boxes of i and its ancestors from their children, until one does not change
    while (i != NONE) {
      Node2& nd = nodes[i];
      const Node2 &a = nodes[c0[i]], &b = nodes[c1[i]];
      float lo[3], hi[3];
      for (int k = 0; k < 3; ++k) { lo[k] = std::min(a.lo[k], b.lo[k]); hi[k] = std::max(a.hi[k], b.hi[k]); }
      if (memcmp(lo, nd.lo, 12) == 0 && memcmp(hi, nd.hi, 12) == 0) break;
      memcpy(nd.lo, lo, 12); memcpy(nd.hi, hi, 12);
      const double na = half_area(lo, hi);
      acc += na - area[i];
      area[i] = na;
      i = parent[i];
    }
  };
  // takes the internal node `q` out: its child `keep` moves up into q's place
  auto unlink = [&](uint32_t q, uint32_t keep) {
    const uint32_t g = parent[q];
    acc -= area[q];
    parent[keep] = g;
    if (g == NONE) root = keep;
    else { if (c0[g] == q) c0[g] = keep; else c1[g] = keep; refit_up(g); }
  };
  // puts the free node id `q` between x and x's parent, with children (x, sub)
  auto link = [&](uint32_t q, uint32_t x, uint32_t sub) {
    const uint32_t px = parent[x];
    c0[q] = x; c1[q] = sub;
    parent[q] = px; parent[x] = q; parent[sub] = q;
    if (px == NONE) root = q;
    else if (c0[px] == x) c0[px] = q; else c1[px] = q;
    Node2& qn = nodes[q];
    for (int k = 0; k < 3; ++k) { qn.lo[k] = std::min(nodes[x].lo[k], nodes[sub].lo[k]); qn.hi[k] = std::max(nodes[x].hi[k], nodes[sub].hi[k]); }
    area[q] = half_area(qn.lo, qn.hi);
    acc += area[q];
    refit_up(px);
  };
  auto union_area = [&](uint32_t x, const Node2& s) {
    const Node2& a = nodes[x];
    float lo[3], hi[3];
    for (int k = 0; k < 3; ++k) { lo[k] = std::min(a.lo[k], s.lo[k]); hi[k] = std::max(a.hi[k], s.hi[k]); }
    return half_area(lo, hi);
  };
  constexpr uint32_t kMaxSearchSteps = 2048;   // per insertion: 45 on average, 172 at most on the scenes measured
  uint32_t max_pops = 0; uint64_t sum_pops = 0, n_searches = 0;
  struct Cand { double induced; uint32_t node; };
  auto worse = [](const Cand& a, const Cand& b) { return a.induced > b.induced || (a.induced == b.induced && a.node > b.node); };
  std::vector<Cand> heap;
  auto reinsert = [&](uint32_t sub, uint32_t q) {   // subtree `sub` (detached) goes back under the free node id q
    // (Seeding `best` with the place the subtree came from prunes the search a little earlier, but then that place wins every
    // tie, and on regular geometry — many boxes of equal area — the 8-wide tree that comes out is measurably worse: atrium
    // 13.8 instead of 13.4 node visits per secondary ray. Ties go to the candidate with the least induced cost, found first.)
    const Node2 sb = nodes[sub];
    const double sub_area = area[sub];
    double best = DBL_MAX;
    uint32_t best_x = root;
    heap.clear();
    heap.push_back({0.0, root});
    uint32_t pops = 0;
    while (!heap.empty()) {
      if (++pops > kMaxSearchSteps) break;   // bounds the build time on a pathological tree; the best place found so far is still a valid one
      std::pop_heap(heap.begin(), heap.end(), worse);
      const Cand c = heap.back();
      heap.pop_back();
      if (c.induced + sub_area >= best) break;   // everything left in the heap has at least this much induced cost
      const double direct = union_area(c.node, sb);
      if (c.induced + direct < best) { best = c.induced + direct; best_x = c.node; }
      if (c0[c.node] != NONE) {
        const double ind = c.induced + direct - area[c.node];
        if (ind + sub_area < best) {
          heap.push_back({ind, c0[c.node]}); std::push_heap(heap.begin(), heap.end(), worse);
          heap.push_back({ind, c1[c.node]}); std::push_heap(heap.begin(), heap.end(), worse);
        }
      }
    }
    max_pops = std::max(max_pops, pops); sum_pops += pops; ++n_searches;
    link(q, best_x, sub);
  };
  std::vector<uint32_t> sel;
  const bool verbose = getenv("PT_BVH_REINSERT_VERBOSE") != nullptr;
  auto total_area = [&]() { double t = 0.0; for (uint32_t i = 0; i < n; ++i) if (c0[i] != NONE) t += area[i]; return t; };
  const double area_before = total_area();
  if (verbose) fprintf(stderr, "reinsertion: binary-tree area sum %.6g\n", area_before);
  for (int pass = 0; pass < passes; ++pass) {
    if (verbose && pass) fprintf(stderr, "reinsertion: after pass %d: %.6g\n", pass, total_area());
    // a first pass that gains less than 1 % says the top-down tree is as good as this optimiser gets it (regular scenes:
    // street -0.5 %, courtyard -0.1 %; atrium -3.3 %, street_hard -10.8 %): stop, build_bvh8 would discard the result anyway
    if (pass == 1 && total_area() > 0.99 * area_before) return false;
    sel.clear();
    for (uint32_t i = 0; i < n; ++i)
      if (c0[i] != NONE && i != root) sel.push_back(i);
    size_t take = std::min<size_t>(max_per_pass, (size_t)((double)sel.size() * fraction));
    if (take == 0) break;
    take = std::min(take, sel.size());
    // pass p works on the nodes ranked [p * take, (p + 1) * take) by area: the largest first, every node at most once
    // (repeating the top slice converges after two or three passes and leaves the levels below untouched)
    const size_t lo_rank = careful ? std::min((size_t)pass * take, sel.size() - take) : 0, hi_rank = lo_rank + take;
    auto larger = [&](uint32_t a, uint32_t b) { return area[a] > area[b] || (area[a] == area[b] && a < b); };
    if (hi_rank < sel.size()) std::nth_element(sel.begin(), sel.begin() + (ptrdiff_t)hi_rank, sel.end(), larger);
    if (lo_rank > 0) std::nth_element(sel.begin(), sel.begin() + (ptrdiff_t)lo_rank, sel.begin() + (ptrdiff_t)hi_rank, larger);
    std::sort(sel.begin() + (ptrdiff_t)lo_rank, sel.begin() + (ptrdiff_t)hi_rank, larger);
    sel.erase(sel.begin(), sel.begin() + (ptrdiff_t)lo_rank);
    for (size_t t = 0; t < take; ++t) {
      const uint32_t N = sel[t];
      if (N == root || c0[N] == NONE) continue;   // (ids are reused inside the pass)
      // N and its parent P leave the tree (N's sibling S moves up), N's two subtrees go back one after the other, the larger
      // first. Unlike putting N back as a whole this can END higher than it began (the second subtree finds the tree changed
      // by the first): such an operation is undone — the tree is again what it was — so the area sum never grows.
      const uint32_t P = parent[N], L = c0[N], R = c1[N];
      const uint32_t S = c0[P] == N ? c1[P] : c0[P];
      acc = -area[N];
      unlink(P, S);
      const bool l_first = area[L] >= area[R];
      reinsert(l_first ? L : R, N);
      reinsert(l_first ? R : L, P);
      if (careful && acc > 0.0) {
        const uint32_t a = l_first ? L : R, b = l_first ? R : L;   // a hangs under N, b under P
        unlink(P, c0[P] == b ? c1[P] : c0[P]);
        unlink(N, c0[N] == a ? c1[N] : c0[N]);
        c0[N] = L; c1[N] = R; parent[L] = N; parent[R] = N;
        Node2& nn_ = nodes[N];
        for (int k = 0; k < 3; ++k) { nn_.lo[k] = std::min(nodes[L].lo[k], nodes[R].lo[k]); nn_.hi[k] = std::max(nodes[L].hi[k], nodes[R].hi[k]); }
        area[N] = half_area(nn_.lo, nn_.hi);
        parent[N] = NONE;
        link(P, S, N);
      }
    }
  }
  if (verbose) fprintf(stderr, "reinsertion: after pass %d: %.6g; %llu searches, %.1f steps on average, %u at most\n", passes, total_area(),
                       (unsigned long long)n_searches, (double)sum_pops / (double)std::max<uint64_t>(1, n_searches), max_pops);
  // back into the array form the collapse expects: root at 0, the two children of a node adjacent and after their parent
  std::vector<Node2> out(n);
  std::vector<uint32_t> old_of(n);
  old_of[0] = root;
  uint32_t next = 1;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t o = old_of[i];
    out[i] = nodes[o];
    if (c0[o] != NONE) {
      out[i].left = next;
      old_of[next] = c0[o]; old_of[next + 1] = c1[o];
      next += 2;
    } else {
      out[i].left = 0;
    }
  }
  nodes.swap(out);
  return true;
}

// Leaves of the binary tree hold one reference each (Node2::first = its primitive while building). This pass lays the
// references out in depth-first order and gives every node the range of its subtree: first / count into `order`.
static void assign_ranges(std::vector<Node2>& nodes, std::vector<uint32_t>& order) {
  order.clear();
  std::vector<std::pair<uint32_t, bool>> stack;   // (node, children done)
  stack.push_back({0u, false});
  while (!stack.empty()) {
    auto [i, after] = stack.back();
    stack.pop_back();
    Node2& nd = nodes[i];
    if (nd.left == 0) {
      const uint32_t prim = nd.first;
      nd.first = (uint32_t)order.size();
      nd.count = 1;
      order.push_back(prim);
    } else if (!after) {
      stack.push_back({i, true});
      stack.push_back({nd.left + 1, false});
      stack.push_back({nd.left, false});
    } else {
      nd.first = nodes[nd.left].first;
      nd.count = nodes[nd.left].count + nodes[nd.left + 1].count;
    }
  }
}

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
        // both ways round: a degenerate triangle (two equal positions) "shares two vertices" with any triangle through
        // that position, but the record has room for ONE vertex of the second triangle that the first does not have
        if (sl.tag == tag && sl.tri != 0xFFFFFFFFu && sl.tri != i && partner[sl.tri] == 0xFFFFFFFFu &&
            shared_vertices(verts + (size_t)sl.tri * 9, t) >= 2 && shared_vertices(t, verts + (size_t)sl.tri * 9) >= 2) {
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

void build_bvh8(const float* verts, uint32_t n, Bvh8& out, int num_threads, int pair_mode, bool fast) {
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
  if (pair_mode < 0)
    if (const char* e = getenv("PT_BVH_PAIRS")) pair_mode = atoi(e);   // tuning runs only: an explicit flag of the caller is never overridden
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
  std::vector<PrimRef> refs;
  refs.reserve(order.size());
  for (uint32_t i : order) {
    if (partner[i] == 0xFFFFFFFFu) refs.push_back({i, 0xFFFFFFFFu});
    else if (partner[i] > i) refs.push_back({i, partner[i]});
  }
  const uint32_t np = (uint32_t)refs.size();
  std::vector<Ref> boxes(np);
  std::vector<float> weight(np);      // SAH weight of a record: one triangle test, or two that share their set-up
  for (uint32_t k = 0; k < np; ++k) {
    Ref& p = boxes[k];
    p.prim = k;
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
    weight[k] = refs[k].b == 0xFFFFFFFFu ? 1.0f : 1.6f;
    if (refs[k].b != 0xFFFFFFFFu) out.num_pairs++;
  }
  // Spatial splits may add up to split_budget x the primitive count in references; a node is considered for one when the
  // overlap of its best object split exceeds split_alpha x the root's area, and the split is taken when it costs less than
  // split_gain x the object split. Chosen on the CPU walker (tools/bvh_cpu_quality.py, secondary rays): on street_hard
  // 69.8 nodes + 178.8 triangles per ray without splits, 37.7 + 19.1 with these values (gain 1.0: 35.2 + 19.9, but it costs
  // the regular scenes 1-2 % more node visits; gain 0.9 leaves atrium / street within 0.1 %; gain 0.75: 64 + 26.5).
  // PT_BVH_SPLIT_BUDGET / _ALPHA / _GAIN override them in tuning runs (budget 0 = object splits only).
  double split_budget = 1.0, split_alpha = 1e-6, split_gain = 0.9;
  if (const char* e = getenv("PT_BVH_SPLIT_BUDGET")) { const double v = atof(e); if (v >= 0.0 && v <= 4.0) split_budget = v; }
  if (const char* e = getenv("PT_BVH_SPLIT_ALPHA")) { const double v = atof(e); if (v >= 0.0) split_alpha = v; }
  if (const char* e = getenv("PT_BVH_SPLIT_GAIN")) { const double v = atof(e); if (v > 0.0) split_gain = v; }
  if (fast) split_budget = 0.0;
  std::vector<Node2> n2;
  {
    SplitBuilder sb(verts, refs, pad, n2, num_threads, split_alpha, split_gain);
    sb.run(std::move(boxes), (uint64_t)((double)np * split_budget));
    out.spatial_splits = sb.spatial_splits();
  }
  // ---- SAH-optimal collapse (dynamic programme over "up to i roots", i = 1..7) ----
  // SAH constants of the collapse: cost of one 8-wide node step vs one triangle test. PT_BVH_CPRIM overrides the
  // triangle cost for tuning runs (tools/tune_bvh.sh).
  double Cnode = 1.0, Cprim = 0.3;
  if (const char* e = getenv("PT_BVH_CPRIM")) { const double v = atof(e); if (v > 0.0) Cprim = v; }
  struct Collapse {
    std::vector<uint32_t> order;
    std::vector<double> cost;
    std::vector<uint8_t> dec, isplit;
  };
  // lays the tree's references out (assign_ranges) and fills the decision tables; returns the SAH cost of the 8-wide tree
  auto collapse = [&](std::vector<Node2>& tree, Collapse& cl) -> double {
    assign_ranges(tree, cl.order);
    const uint32_t nn = (uint32_t)tree.size();
    cl.cost.assign((size_t)nn * 7, 0.0);
    cl.dec.assign((size_t)nn * 7, 0);
    cl.isplit.assign(nn, 0);
    std::vector<float> wsum(nn, 0.0f);   // SAH weight of the records below a node (children are allocated after their parent)
    for (uint32_t idx = nn; idx-- > 0;) {
      const Node2& nd = tree[idx];
      const double A = half_area(nd.lo, nd.hi);
      double* c = &cl.cost[(size_t)idx * 7];
      uint8_t* d = &cl.dec[(size_t)idx * 7];
      if (nd.left == 0) {
        for (uint32_t i = 0; i < nd.count; ++i) wsum[idx] += weight[cl.order[nd.first + i]];
        for (int i = 0; i < 7; ++i) { c[i] = A * Cprim * (double)wsum[idx]; d[i] = mk_dec(D_LEAF, 0, 0); }
        continue;
      }
      wsum[idx] = wsum[nd.left] + wsum[nd.left + 1];
      const double* cl_ = &cl.cost[(size_t)nd.left * 7];
      const double* cr = &cl.cost[(size_t)(nd.left + 1) * 7];
      double cdist[9]; int ksplit[9];
      for (int j = 2; j <= 8; ++j) {
        double best = DBL_MAX; int bk = 1;
        for (int k = 1; k < j; ++k) {
          if (k > 7 || j - k > 7) continue;
          const double v = cl_[k - 1] + cr[j - k - 1];
          if (v < best) { best = v; bk = k; }
        }
        cdist[j] = best; ksplit[j] = bk;
      }
      const double cleaf = nd.count <= PT_BVH8_MAX_LEAF_TRIS ? A * Cprim * (double)wsum[idx] : DBL_MAX;
      const double cint = cdist[8] + A * Cnode;
      cl.isplit[idx] = (uint8_t)ksplit[8];
      if (cleaf <= cint) { c[0] = cleaf; d[0] = mk_dec(D_LEAF, 0, 0); }
      else { c[0] = cint; d[0] = mk_dec(D_INTERNAL, 0, 0); }
      for (int i = 1; i < 7; ++i) {
        if (cdist[i + 1] < c[i - 1]) { c[i] = cdist[i + 1]; d[i] = mk_dec(D_DISTRIBUTE, ksplit[i + 1], i + 1 - ksplit[i + 1]); }
        else { c[i] = c[i - 1]; d[i] = d[i - 1]; }
      }
    }
    return cl.cost[0];
  };
  // Insertion-based optimisation of the binary tree (optimize_by_reinsertion), kept only if the 8-wide tree that comes out of
  // it is better by the collapse's own objective: the optimiser minimises the area of ALL binary nodes, the traversal pays
  // for the ones the collapse keeps, and the two do not always move together. CPU walker, node visits per secondary ray
  // (tools/bvh_cpu_quality.py), 5 passes over the 2 % largest nodes: atrium 13.88 -> 13.42 (8-wide SAH cost -3.9 %),
  // street_hard 37.7 -> 35.4 (-9.8 %), courtyard 18.76 -> 18.67 (-0.4 %), street 14.11 -> 14.51 (+0.01 %): the last two keep
  // the top-down tree. PT_BVH_REINSERT_PASSES / _FRAC / _CAP: tuning runs (0 passes = off).
  int reinsert_passes = 5;
  double reinsert_frac = 0.02;
  size_t reinsert_cap = (size_t)1 << 20;
  if (const char* e = getenv("PT_BVH_REINSERT_PASSES")) reinsert_passes = atoi(e);
  if (const char* e = getenv("PT_BVH_REINSERT_FRAC")) { const double v = atof(e); if (v > 0.0 && v <= 1.0) reinsert_frac = v; }
  if (const char* e = getenv("PT_BVH_REINSERT_CAP")) { const long v = atol(e); if (v > 0) reinsert_cap = (size_t)v; }
  if (fast) reinsert_passes = 0;
  Collapse cl;
  {
    // the top-down tree and the two optimised copies: three collapses, the cheapest 8-wide tree wins (an optimised one has
    // to be better by 1 %); the copies are worked on by two more threads when the caller gave any
    std::vector<Node2> cand[2];
    Collapse ccl[2];
    double ccost[2] = {DBL_MAX, DBL_MAX};
    const bool optimise = reinsert_passes > 0 && n2.size() >= 64;
    auto work = [&](int v) {
      cand[v] = n2;
      if (optimize_by_reinsertion(cand[v], reinsert_passes, reinsert_frac, reinsert_cap, v == 1)) ccost[v] = collapse(cand[v], ccl[v]);
      else std::vector<Node2>().swap(cand[v]);
    };
    std::vector<std::thread> pool;
    if (optimise) {
      if (num_threads > 1) { pool.emplace_back(work, 0); pool.emplace_back(work, 1); }
      else { work(0); work(1); }
    }
    double c = 0.0;
    {
      std::vector<Node2> keep;
      if (optimise && num_threads > 1) keep = n2;   // the threads copy from n2 while it is collapsed: collapse a copy of our own
      c = collapse(optimise && num_threads > 1 ? keep : n2, cl);
      for (auto& th : pool) th.join();
      if (optimise && num_threads > 1) n2.swap(keep);
    }
    const int v = ccost[1] < ccost[0] ? 1 : 0;
    if (optimise && ccost[v] < 0.99 * c) {
      n2.swap(cand[v]);
      std::swap(cl, ccl[v]);
      c = ccost[v];
      out.reinsertion_kept = 1 + v;
    }
    out.sah_cost = c;
  }
  const std::vector<uint32_t>& ord = cl.order;   // references in depth-first order of the chosen tree
  const std::vector<uint8_t>&dec = cl.dec, &isplit = cl.isplit;
  out.num_references = ord.size();
  const uint32_t nn = (uint32_t)n2.size();

  // ---- emit 8-wide nodes breadth-first ----
  struct Work { uint32_t n2; uint32_t out; uint32_t depth; };
  std::deque<Work> queue;
  out.nodes.reserve(nn / 4 + 16);
  if (out.paired) out.prims.reserve(ord.size()); else out.tris.reserve(ord.size());
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
        // 1..3 references; two pieces of one primitive that a spatial split sent into the same leaf need one record only
        uint32_t uniq[PT_BVH8_MAX_LEAF_TRIS], cnt = 0;
        for (uint32_t i = 0; i < ch.count; ++i) {
          const uint32_t pk = ord[ch.first + i];
          bool seen = false;
          for (uint32_t j = 0; j < cnt; ++j) seen = seen || uniq[j] == pk;
          if (!seen) uniq[cnt++] = pk;
        }
        const uint32_t unary = cnt == 1 ? 1u : (cnt == 2 ? 3u : 7u);
        node.meta[s] = (uint8_t)((unary << 5) | tri_off);
        for (uint32_t i = 0; i < cnt; ++i) {
          const PrimRef& pr = refs[uniq[i]];
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
