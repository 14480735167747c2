// pt_api.cpp — the C-ABI of libpathtracer_amd.so (declared in include/pt_api.h).
// Host side of the hot path: what PathTracer::InitRtScene / RenderRT / RestartAccumulation / UpdateOutputTexture and
// Sky::ComputeTranmittanceLut do through the FANCY RenderCore in the reference (PathTracer/PathTracer.cpp:171-339,
// 659-787,834-878,912-915; Sky.cpp:127-148) is done here with plain HIP runtime calls on one stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pt_api.h"
#include "../../include/pt_api_tuning.h"
#ifndef PT_NODE_STRIDE_U4
#define PT_NODE_STRIDE_U4 5
#endif
#include "bvh_builder.h"
#include "kernels.h"
#include "scene_flatten.h"

using namespace pt;

static thread_local std::string g_create_error;

#ifndef PT_DEFAULT_FUSED_PRIMARY
#define PT_DEFAULT_FUSED_PRIMARY 1   // A/B builds: -DPT_DEFAULT_FUSED_PRIMARY=0
#endif

struct pt_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
  LaunchCfg cfg;

  // scene
  bool has_scene = false;
  DeviceScene ds;
  std::vector<void*> scene_allocs;
  std::vector<uint32_t> instance_material;   // host copy, for pt_scene_set_material_ext
  uint32_t num_materials = 0;
  uint64_t num_records = 0;                  // leaf records of the BVH (48-byte triangles or 64-byte pairs, ds.paired)
  float4* instance_ext = nullptr;
  uint32_t* albedo_tex = nullptr;            // pt_scene_set_albedo_texture
  PtSceneInfo info{};

  // sky
  uint2* lut = nullptr;
  float4* skyview = nullptr;
  bool has_sky = false;

  // accumulation target
  float4* accum = nullptr;
  bool accum_owned = false;
  uint32_t width = 0, height = 0;
  uint64_t num_samples = 0;

  // wavefront state
  WavefrontBuffers wb;
  std::vector<void*> wb_allocs;
  // second pipeline lane (PT_OPT_PIPELINE): its own wavefront state, traversal-stack spill region and stream, so the
  // two halves of a batch run concurrently (one half's shade and kernel drains under the other half's extend)
  WavefrontBuffers wb2;
  std::vector<void*> wb2_allocs;
  uint2* spill2 = nullptr;
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool pipeline = false;
  // Frame lanes (PT_OPT_FRAME_LANES): the reference's interactive loop calls RenderRT once per frame with ONE sample per
  // pixel (PathTracer.cpp:573-595) — at 640x360 a batch of 230 k paths whose dozen kernels each run at latency, not at
  // throughput, with the GPU nearly empty. Such small pt_render calls are issued round-robin onto a few internal streams
  // with their own wavefront state, so consecutive calls overlap; only the resolves (the accumulation target's
  // read-modify-write) stay on the ctx stream, in call order, so the target is bit-identical to the serial schedule.
  static constexpr int kMaxFrameLanes = 8;
  struct FrameLane {
    WavefrontBuffers wb;
    std::vector<void*> allocs;
    uint2* spill = nullptr;            // traversal-stack spill region of this lane's extend launches
    hipStream_t stream = nullptr;
    hipEvent_t ev_done = nullptr;      // lane stream: everything up to k_sky of the lane's current call
    hipEvent_t ev_resolved = nullptr;  // ctx stream: the resolve that consumed the lane's path state
    bool resolve_recorded = false;
    uint64_t seen_epoch = 0;
  };
  FrameLane fl[kMaxFrameLanes];
  int frame_lanes = 6;                 // PT_OPT_FRAME_LANES (1 = off). 640x360 atrium, ms per 1-spp call: 1.10 / 0.61 / 0.61 / 0.45 / 0.49 at 1 / 2 / 4 / 6 / 8 lanes
                                       // (tools/exp_frame_lanes.py; the streams share 4 hardware queues, hence the steps)
  int fl_next = 0;
  uint64_t fork_epoch = 1;             // bumped by whatever queues work on the ctx stream that lane kernels depend on (scene, sky LUT, stream change)
  hipEvent_t ev_epoch = nullptr;
  const WavefrontBuffers* last_wb = nullptr;   // wavefront state of the last render call (pt_debug_read_queue)
  bool fused_primary = PT_DEFAULT_FUSED_PRIMARY != 0;   // PT_OPT_FUSED_PRIMARY
  int primary_packet = 2;             // PT_OPT_PRIMARY_PACKET: 0 = k_extend, 1 = k_extend_packet (one BVH walk per wave), 2 = whichever is
                                      // faster on this scene at this frame size (timed once, calibrate_primary_kernel)
  int packet_choice = -1;             // the outcome of that timing (-1 = not timed yet), for packet_choice_w x packet_choice_h
  uint32_t packet_choice_w = 0, packet_choice_h = 0;
  float packet_cal_ms[2] = {0.0f, 0.0f};   // what it measured: k_extend, k_extend_packet
  uint64_t packet_launches = 0;
  // PT_OPT_SHARED_TRIANGLE_PHASE: 0 = per-lane triangle tests, 1 = shared triangle phase (kernels.hip), 2 = whichever traces
  // this scene's secondary rays faster: the first large batch of a scene runs its bounce-2 queue (bounce 1 at depth 1) twice more under HIP
  // events (same hit records every time), and the next call reads them. Until then: per-lane.
  int shared_tri = 2;
  int shared_choice = -1;             // -1 = not timed yet
  int shared_force = -1;              // the batch being queued runs this form whatever the choice (the timing's two batches)
  int shared_cal_stage = 0;           // large batches of this scene seen by the timing: 0 none, 1 one (untimed: first launches, cold caches), 2..5 = 1..4 timed ones
  uint64_t shared_cal_paths = 0;      // paths of the timed batches (both must be the same size)
  bool shared_cal_pending = false;
  std::pair<hipEvent_t, hipEvent_t> shared_cal_ev[4] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};   // per-lane, shared, per-lane, shared
  float shared_cal_ms[2] = {0.0f, 0.0f};   // per-lane, shared: two batches each
  bool shared_possible = true;        // record indices fit the phase's 26-bit list entries
  uint64_t shared_launches = 0;
  int noalloc_blocks = 0;             // grid of pt_trace_rays_device_noalloc's extend instantiation (extend_grid)
  int packet_blocks = 0;              // its persistent grid (CUs x occupancy), looked up on first use per scene
  bool last_render_fused = false;     // the bounce-0 queue of the last pt_render was never written (pt_debug_read_queue)
  float4* ao_o = nullptr;       // AO pipeline: 16 rays per pixel-sample (origin,tmin / dir,tmax / hit records)
  float4* ao_d = nullptr;
  float4* ao_hits = nullptr;
  uint64_t ao_capacity = 0;
  float4* tmp_rays = nullptr;   // staging for pt_trace_rays (AoS PtRay)
  float4* tmp_hits = nullptr;
  uint64_t tmp_capacity = 0;
  unsigned char* scratch = nullptr;   // this device's copy of the library's static scratch array (pt_trace_rays_device_noalloc)

  // options
  bool count_traversal = false;
  bool time_kernels = false;
  int64_t samples_per_batch = 0;
  int sky_mode = 0;
  bool russian_roulette = false;

  // timing
  std::vector<std::pair<hipEvent_t, hipEvent_t>> extend_events, render_events, event_pool;
  std::vector<int> extend_event_bounce;
  double extend_ms = 0.0, render_ms = 0.0;
  double extend_ms_bounce[16] = {0};
  uint64_t extend_launches = 0;
  uint64_t extend_launches_bounce[16] = {0};

  int fail(int code, const std::string& msg) {
    err = msg;
    return code;
  }
};

#define PT_HIP(call)                                                                                   \
  do {                                                                                                 \
    hipError_t e__ = (call);                                                                           \
    if (e__ != hipSuccess)                                                                             \
      return ctx->fail(e__ == hipErrorOutOfMemory ? PT_ERR_OUT_OF_MEMORY : PT_ERR_HIP,                 \
                       std::string(#call) + ": " + hipGetErrorString(e__));                            \
  } while (0)

#define PT_REQUIRE_CTX()                                         \
  do {                                                           \
    if (!ctx) return PT_ERR_INVALID_ARG;                         \
    hipError_t e__ = hipSetDevice(ctx->device);                  \
    if (e__ != hipSuccess) return ctx->fail(PT_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e__)); \
  } while (0)

namespace {

template <class T>
int dev_alloc(pt_ctx* ctx, T** p, size_t count, std::vector<void*>* track) {
  *p = nullptr;
  if (count == 0) count = 1;
  void* v = nullptr;
  PT_HIP(hipMalloc(&v, count * sizeof(T)));
  *p = (T*)v;
  if (track) track->push_back(v);
  return PT_OK;
}

void free_list(std::vector<void*>& l) {
  for (void* p : l) (void)hipFree(p);
  l.clear();
}

std::pair<hipEvent_t, hipEvent_t> get_event_pair(pt_ctx* ctx) {
  if (!ctx->event_pool.empty()) {
    auto p = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return p;
  }
  hipEvent_t a = nullptr, b = nullptr;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  return {a, b};
}

// Fold finished event pairs into the running sums (call after the stream is synchronised).
void collect_events(pt_ctx* ctx) {
  for (size_t i = 0; i < ctx->extend_events.size(); ++i) {
    auto& p = ctx->extend_events[i];
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
      ctx->extend_ms += ms;
      ctx->extend_ms_bounce[std::min(15, ctx->extend_event_bounce[i])] += ms;
    }
    ctx->event_pool.push_back(p);
  }
  ctx->extend_events.clear();
  ctx->extend_event_bounce.clear();
  for (auto& p : ctx->render_events) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) ctx->render_ms += ms;
    ctx->event_pool.push_back(p);
  }
  ctx->render_events.clear();
}

int ensure_wavefront_for(pt_ctx* ctx, WavefrontBuffers& wb, std::vector<void*>& allocs, uint64_t paths) {
  if (wb.capacity >= paths && wb.counters) return PT_OK;
  PT_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->stream2) PT_HIP(hipStreamSynchronize(ctx->stream2));
  unsigned long long* stats = wb.stats;  // statistics survive a re-allocation
  std::vector<void*> keep;
  for (void* p : allocs)
    if (p == (void*)stats) keep.push_back(p); else (void)hipFree(p);
  allocs = keep;
  wb = WavefrontBuffers();
  wb.stats = stats;
  int rc;
  for (int k = 0; k < 2; ++k) {
    if ((rc = dev_alloc(ctx, &wb.ray_o[k], paths, &allocs))) return rc;
    if ((rc = dev_alloc(ctx, &wb.ray_d[k], paths, &allocs))) return rc;
    if ((rc = dev_alloc(ctx, &wb.qid[k], paths, &allocs))) return rc;
  }
  if ((rc = dev_alloc(ctx, &wb.hits, paths, &allocs))) return rc;
  if ((rc = dev_alloc(ctx, &wb.st_T, paths, &allocs))) return rc;
  if ((rc = dev_alloc(ctx, &wb.st_L, paths, &allocs))) return rc;
  if ((rc = dev_alloc(ctx, &wb.sky_o, paths, &allocs))) return rc;
  if ((rc = dev_alloc(ctx, &wb.sky_d, paths, &allocs))) return rc;
  if ((rc = dev_alloc(ctx, &wb.gen, (size_t)8, &allocs))) return rc;
  if ((rc = dev_alloc(ctx, &wb.counters, (size_t)CTR_WORDS, &allocs))) return rc;
  if (!wb.stats) {
    if ((rc = dev_alloc(ctx, &wb.stats, (size_t)ST_WORDS, &allocs))) return rc;
    PT_HIP(hipMemsetAsync(wb.stats, 0, ST_WORDS * sizeof(unsigned long long), ctx->stream));
  }
  wb.capacity = paths;
  return PT_OK;
}

int ensure_wavefront(pt_ctx* ctx, uint64_t paths) { return ensure_wavefront_for(ctx, ctx->wb, ctx->wb_allocs, paths); }

int ensure_lane2(pt_ctx* ctx, uint64_t paths) {
  if (!ctx->stream2) {
    PT_HIP(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    PT_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    PT_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  }
  return ensure_wavefront_for(ctx, ctx->wb2, ctx->wb2_allocs, paths);
}

size_t spill_region_entries(const pt_ctx* ctx) {   // uint2 entries: the largest persistent grid ever launched (16 blocks per CU x 4 waves)
  return (size_t)ctx->cfg.num_cus * 16u * 4u * ctx->ds.stack_entries * 64u;
}

int ensure_frame_lane(pt_ctx* ctx, pt_ctx::FrameLane& L, uint64_t paths) {
  if (!L.stream) {
    PT_HIP(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
    PT_HIP(hipEventCreateWithFlags(&L.ev_done, hipEventDisableTiming));
    PT_HIP(hipEventCreateWithFlags(&L.ev_resolved, hipEventDisableTiming));
  }
  if (!ctx->ev_epoch) PT_HIP(hipEventCreateWithFlags(&ctx->ev_epoch, hipEventDisableTiming));
  int rc = ensure_wavefront_for(ctx, L.wb, L.allocs, paths);   // (re-allocation synchronises the ctx stream, which waits on every lane)
  if (rc != PT_OK) return rc;
  if (!L.spill) {
    if ((rc = dev_alloc(ctx, &L.spill, spill_region_entries(ctx), nullptr))) return rc;
  }
  return PT_OK;
}

// Scene or stream changes: the lanes' spill regions belong to the old scene's stack depth
void release_frame_lane_spills(pt_ctx* ctx) {
  for (auto& L : ctx->fl) {
    if (L.stream) (void)hipStreamSynchronize(L.stream);
    if (L.spill) { (void)hipFree(L.spill); L.spill = nullptr; }
  }
}

int ensure_tmp(pt_ctx* ctx, uint64_t rays) {
  if (ctx->tmp_capacity >= rays) return PT_OK;
  PT_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->tmp_rays) (void)hipFree(ctx->tmp_rays);
  if (ctx->tmp_hits) (void)hipFree(ctx->tmp_hits);
  ctx->tmp_rays = nullptr; ctx->tmp_hits = nullptr; ctx->tmp_capacity = 0;
  int rc;
  if ((rc = dev_alloc(ctx, &ctx->tmp_rays, rays * 2, nullptr))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->tmp_hits, rays, nullptr))) return rc;
  ctx->tmp_capacity = rays;
  return PT_OK;
}

// Persistent grid of the extend kernel: CUs x resident blocks of the instantiation that will be launched.
int extend_grid(pt_ctx* ctx) {
  {   // the grid of pt_trace_rays_device_noalloc's instantiation (no private segment, its own register count), so that
      // entry point has nothing to look up
    const uint32_t le = std::min<uint32_t>(std::min(ctx->ds.stack_entries, ctx->cfg.lds_stack_entries) + ctx->cfg.defer_entries, PT_EXTEND_MAX_LDS_ENTRIES);
    ctx->noalloc_blocks = ctx->cfg.num_cus * std::max(1, std::min(8, extend_occupancy_blocks(le, std::min<uint32_t>(ctx->cfg.top_nodes, ctx->ds.num_nodes),
                                                                                             ctx->cfg.defer_entries > 0, ctx->ds.paired, true)));
  }
  int per_cu = ctx->cfg.extend_blocks_per_cu;
  if (per_cu <= 0) {
    const uint32_t lds_entries = std::min<uint32_t>(std::min(ctx->ds.stack_entries, ctx->cfg.lds_stack_entries) + ctx->cfg.defer_entries, PT_EXTEND_MAX_LDS_ENTRIES);
    per_cu = std::max(1, std::min(8, extend_occupancy_blocks(lds_entries, std::min<uint32_t>(ctx->cfg.top_nodes, ctx->ds.num_nodes), ctx->cfg.defer_entries > 0, ctx->ds.paired)));
  }
  return ctx->cfg.num_cus * per_cu;
}

// The direction LUT of sky mode 1 is within relMSE 1e-4 of the marched sky for cameras up to ~50 km of altitude
// (tools/exp_skyview_alt.py); higher up its horizon-split parametrisation loses accuracy, so those renders march.
bool skyview_valid_for(const PtRtConsts& c) {
  const double R = c.mySkyConsts.myAtmosphere.BottomRadius;
  const double x = c.myCameraPos[0], y = (double)c.myCameraPos[1] + R, z = c.myCameraPos[2];
  const double altitude = std::sqrt(x * x + y * y + z * z) - R;
  return std::isfinite(altitude) && altitude <= 5.0e4;
}

int ensure_ao(pt_ctx* ctx, uint64_t rays) {
  if (ctx->ao_capacity >= rays) return PT_OK;
  PT_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->ao_o) (void)hipFree(ctx->ao_o);
  if (ctx->ao_d) (void)hipFree(ctx->ao_d);
  if (ctx->ao_hits) (void)hipFree(ctx->ao_hits);
  ctx->ao_o = ctx->ao_d = ctx->ao_hits = nullptr; ctx->ao_capacity = 0;
  int rc;
  if ((rc = dev_alloc(ctx, &ctx->ao_o, rays, nullptr))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->ao_d, rays, nullptr))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->ao_hits, rays, nullptr))) return rc;
  ctx->ao_capacity = rays;
  return PT_OK;
}

int check_consts(pt_ctx* ctx, const PtRtConsts* c, bool need_sky) {
  if (!c) return ctx->fail(PT_ERR_INVALID_ARG, "consts is null");
  if (c->myMaxRecursionDepth > 62u) return ctx->fail(PT_ERR_INVALID_ARG, "myMaxRecursionDepth > 62 is not supported");
  if (need_sky && c->mySampleSky && !ctx->has_sky) return ctx->fail(PT_ERR_NO_SKY, "mySampleSky is set but pt_sky_set has not been called");
  return PT_OK;
}

bool use_shared_tri(const pt_ctx* ctx) {
  if (!ctx->shared_possible || ctx->cfg.defer_entries == 0) return false;
  if (ctx->shared_force >= 0) return ctx->shared_force == 1;
  return ctx->shared_tri == 1 || (ctx->shared_tri == 2 && ctx->shared_choice == 1);
}

// PT_OPT_SHARED_TRIANGLE_PHASE = 2. The two forms are timed where they run: of a scene's first large batches the first is
// left alone (first launches, cold caches) and the next four run the per-lane phase, the shared one, and both again, each
// between two events that span its secondary bounces (extend and shade of bounces >= 1: the same work either way except
// for the phase); the shared form is kept for the scene if its two batches together were faster by 1.5 % (scenes it does
// not suit are within +-1 % either way). No launch is added and nothing waits; this reads the times once the four batches
// are through (it blocks only if the last one is still running). Two earlier versions: re-tracing one bounce's queue
// twice on the spot saw 1-2 % on the courtyard where the render gains 4-20 % per bounce (caches warmed by the trace just
// before); one batch of each form drifted by up to 2 % between a scene's second and third batch, enough to flip the
// courtyard (3 % apart) in one run of five.
void drop_shared_cal(pt_ctx* ctx) {   // forget a timing in progress (scene change, destroy): its events go back to the pool
  for (auto& ev : ctx->shared_cal_ev) {
    if (ev.first) ctx->event_pool.push_back(ev);
    ev = {nullptr, nullptr};
  }
  ctx->shared_cal_pending = false;
  ctx->shared_cal_stage = 0;
  ctx->shared_force = -1;
}

int resolve_shared_cal(pt_ctx* ctx) {
  if (!ctx->shared_cal_pending) return PT_OK;
  ctx->shared_cal_pending = false;
  PT_HIP(hipEventSynchronize(ctx->shared_cal_ev[3].second));
  ctx->shared_cal_ms[0] = ctx->shared_cal_ms[1] = 0.0f;
  for (int i = 0; i < 4; ++i) {
    float ms = 0.0f;
    PT_HIP(hipEventElapsedTime(&ms, ctx->shared_cal_ev[i].first, ctx->shared_cal_ev[i].second));
    ctx->shared_cal_ms[i & 1] += ms;
    ctx->event_pool.push_back(ctx->shared_cal_ev[i]);
    ctx->shared_cal_ev[i] = {nullptr, nullptr};
  }
  ctx->shared_choice = ctx->shared_cal_ms[1] <= 0.985f * ctx->shared_cal_ms[0] ? 1 : 0;
  return PT_OK;
}

void extend_one(pt_ctx* ctx, const float4* ro, const float4* rd, uint32_t stride, float4* hits, const uint32_t* count, uint32_t* cursor, int bounce = 0,
                bool any_hit = false, int lane = 0, const float4* primary_gen = nullptr, bool packet = false, uint64_t max_rays = 0) {
  const hipStream_t st = lane >= 2 ? ctx->fl[lane - 2].stream : (lane ? ctx->stream2 : ctx->stream);
  std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
  if (ctx->time_kernels) {
    ev = get_event_pair(ctx);
    (void)hipEventRecord(ev.first, st);
  }
  if (packet) {   // coherent queue (fused primaries): one BVH walk per wave, no per-lane stacks (no spill region, no statistics)
    if (ctx->packet_blocks <= 0)
      ctx->packet_blocks = ctx->cfg.num_cus * std::max(1, std::min(8, extend_packet_occupancy_blocks(ctx->ds.stack_entries, ctx->ds.paired)));
    launch_extend_packet(ctx->ds, ro, rd, hits, count, cursor, ctx->packet_blocks, st, primary_gen);
    ctx->packet_launches++;
  } else if (lane >= 2) {
    // frame lane: a small queue (at most max_rays rays) on a grid cut to 64 rays per wave, so that the extend launches of
    // several frames in flight are resident side by side instead of one persistent grid filling every CU; at that size the
    // kernel deals the queue out to its waves statically (extend_kernel.inc: no cursor traffic)
    DeviceScene ds2 = ctx->ds;
    ds2.spill = ctx->fl[lane - 2].spill;
    LaunchCfg cfg2 = ctx->cfg;
    cfg2.shared_tri = use_shared_tri(ctx);
#ifndef PT_FRAME_LANE_GRID_DIV
#define PT_FRAME_LANE_GRID_DIV 256u
#endif
    // rays per block. With every claim an atomic on the cursor, 512 was best of 256 / 512 / 1024 / 2048 by 5-8 %; with the
    // static deal (at most 64 rays per wave): 128 / 256 / 384 -> 0.44 / 0.36 / 0.48 ms per 640x360 one-sample call on the atrium
    const uint64_t grid_div = PT_FRAME_LANE_GRID_DIV;
    if (max_rays) cfg2.extend_blocks = (int)std::min<uint64_t>((uint64_t)cfg2.extend_blocks, std::max<uint64_t>((uint64_t)cfg2.num_cus, (max_rays + grid_div - 1) / grid_div));
    launch_extend(ds2, ro, rd, stride, hits, count, cursor, ctx->fl[lane - 2].wb.stats, ctx->count_traversal, any_hit, cfg2, st, primary_gen);
  } else if (lane) {
    DeviceScene ds2 = ctx->ds;
    ds2.spill = ctx->spill2;
    LaunchCfg cfg2 = ctx->cfg;
    cfg2.shared_tri = use_shared_tri(ctx);
    launch_extend(ds2, ro, rd, stride, hits, count, cursor, ctx->wb2.stats, ctx->count_traversal, any_hit, cfg2, st, primary_gen);
  } else {
    LaunchCfg cfg2 = ctx->cfg;
    cfg2.shared_tri = use_shared_tri(ctx);
    launch_extend(ctx->ds, ro, rd, stride, hits, count, cursor, ctx->wb.stats, ctx->count_traversal, any_hit, cfg2, st, primary_gen);
  }
  if (!packet && use_shared_tri(ctx)) ctx->shared_launches++;
  ctx->extend_launches++;
  ctx->extend_launches_bounce[std::min(15, std::max(0, bounce))]++;
  if (ctx->time_kernels) {
    (void)hipEventRecord(ev.second, st);
    ctx->extend_events.push_back(ev);
    ctx->extend_event_bounce.push_back(bounce);
  }
}

// PT_OPT_PRIMARY_PACKET = 2: which kernel traces the fused primaries of this scene at this frame size faster? The wave-level
// walk of k_extend_packet wins while the 64 rays of an 8x8-pixel tile share most of their path (atrium at 1080p: 11.2 vs
// 8.1 Grays/s; courtyard at 4K: +12 %) and loses once the triangles are small against a tile's footprint (street at 1080p:
// 6.7 vs 8.3). Both produce the same hit records, so the choice is made by timing them once on up to 2^24 primaries of the
// frame about to be rendered (HIP events, one synchronisation per scene and frame size); the hit records this leaves in the
// first lane's buffers are overwritten by the render itself.
int calibrate_primary_kernel(pt_ctx* ctx, const PtRtConsts& consts, uint32_t width, uint32_t height, uint32_t first_sample, uint32_t k) {
  WavefrontBuffers& wb = ctx->wb;
  const uint64_t n_paths = (uint64_t)width * height * k;
  const uint32_t n_cal = (uint32_t)std::min<uint64_t>(n_paths, 1ull << 24);   // large enough for the work-claim chunks of a real batch
  uint32_t* count = wb.counters + CTR_AT(CTR_QCOUNT0 + 63);   // bounce slot 63 is never a real bounce (myMaxRecursionDepth <= 62)
  uint32_t* cursor = wb.counters + CTR_AT(CTR_WORK0 + 63);
  PT_HIP(hipMemsetAsync(wb.counters, 0, CTR_WORDS * sizeof(uint32_t), ctx->stream));
  launch_primary_setup(consts, wb, width, height, first_sample, k, ctx->stream);
  launch_set_count(count, n_cal, ctx->stream);
  if (ctx->packet_blocks <= 0)
    ctx->packet_blocks = ctx->cfg.num_cus * std::max(1, std::min(8, extend_packet_occupancy_blocks(ctx->ds.stack_entries, ctx->ds.paired)));
  std::pair<hipEvent_t, hipEvent_t> ev[2] = {get_event_pair(ctx), get_event_pair(ctx)};
  for (int pass = 0; pass < 3; ++pass) {   // packet (untimed: warms the caches and loads both code objects' neighbours), k_extend, packet
    const int which = pass == 1 ? 0 : 1;
    PT_HIP(hipMemsetAsync(cursor, 0, sizeof(uint32_t), ctx->stream));
    if (pass > 0) PT_HIP(hipEventRecord(ev[which].first, ctx->stream));
    if (which) launch_extend_packet(ctx->ds, nullptr, nullptr, wb.hits, count, cursor, ctx->packet_blocks, ctx->stream, wb.gen);
    else launch_extend(ctx->ds, nullptr, nullptr, 1, wb.hits, count, cursor, wb.stats, false, false, ctx->cfg, ctx->stream, wb.gen);
    if (pass > 0) PT_HIP(hipEventRecord(ev[which].second, ctx->stream));
  }
  PT_HIP(hipEventSynchronize(ev[1].second));
  for (int i = 0; i < 2; ++i) {
    ctx->packet_cal_ms[i] = 0.0f;
    PT_HIP(hipEventElapsedTime(&ctx->packet_cal_ms[i], ev[i].first, ev[i].second));
    ctx->event_pool.push_back(ev[i]);
  }
  // the packet kernel must win by 5 %: within the timing noise the general kernel is kept, so the choice (and with it the
  // performance of later calls) is reproducible from run to run
  ctx->packet_choice = ctx->packet_cal_ms[1] <= 0.95f * ctx->packet_cal_ms[0] ? 1 : 0;
  ctx->packet_choice_w = width;
  ctx->packet_choice_h = height;
  return PT_OK;
}

}  // namespace

extern "C" {

int pt_create(int device, pt_ctx** out_ctx) {
  if (!out_ctx) { g_create_error = "pt_create: out_ctx is null"; return PT_ERR_INVALID_ARG; }
  *out_ctx = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_create_error = std::string("pt_create: no HIP device (") + hipGetErrorString(e) + ")";
    return PT_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) { g_create_error = "pt_create: device index out of range"; return PT_ERR_NO_DEVICE; }
  if ((e = hipSetDevice(device)) != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return PT_ERR_HIP; }
  pt_ctx* ctx = new pt_ctx();
  ctx->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->cfg.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) {
    g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
    delete ctx;
    return PT_ERR_HIP;
  }
  ctx->stream = ctx->own_stream;
  {
    void* sp = nullptr;
    if ((e = scratch_address(&sp)) != hipSuccess) {   // once per device: every ctx of a device sees the same address
      g_create_error = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e);
      (void)hipStreamDestroy(ctx->own_stream);
      delete ctx;
      return PT_ERR_HIP;
    }
    ctx->scratch = (unsigned char*)sp;
  }
  ctx->cfg.stream_blocks = ctx->cfg.num_cus * 8;
  ctx->cfg.extend_blocks = ctx->cfg.num_cus * 4;
  *out_ctx = ctx;
  return PT_OK;
}

int pt_destroy(pt_ctx* ctx) {
  if (!ctx) return PT_ERR_INVALID_ARG;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  collect_events(ctx);
  drop_shared_cal(ctx);
  for (auto& p : ctx->event_pool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  free_list(ctx->scene_allocs);
  if (ctx->instance_ext) (void)hipFree(ctx->instance_ext);
  if (ctx->albedo_tex) (void)hipFree(ctx->albedo_tex);
  if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
  free_list(ctx->wb_allocs);
  free_list(ctx->wb2_allocs);
  release_frame_lane_spills(ctx);
  for (auto& L : ctx->fl) {
    free_list(L.allocs);
    if (L.ev_done) (void)hipEventDestroy(L.ev_done);
    if (L.ev_resolved) (void)hipEventDestroy(L.ev_resolved);
    if (L.stream) (void)hipStreamDestroy(L.stream);
  }
  if (ctx->ev_epoch) (void)hipEventDestroy(ctx->ev_epoch);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
  if (ctx->tmp_rays) (void)hipFree(ctx->tmp_rays);
  if (ctx->tmp_hits) (void)hipFree(ctx->tmp_hits);
  if (ctx->ao_o) (void)hipFree(ctx->ao_o);     // the AO ray buffers belong to ensure_ao alone
  if (ctx->ao_d) (void)hipFree(ctx->ao_d);
  if (ctx->ao_hits) (void)hipFree(ctx->ao_hits);
  if (ctx->lut) (void)hipFree(ctx->lut);
  if (ctx->skyview) (void)hipFree(ctx->skyview);
  if (ctx->accum && ctx->accum_owned) (void)hipFree(ctx->accum);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return PT_OK;
}

const char* pt_last_error(const pt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int pt_set_stream(pt_ctx* ctx, void* hip_stream) {
  PT_REQUIRE_CTX();
  PT_HIP(hipStreamSynchronize(ctx->stream));
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  ctx->fork_epoch++;
  return PT_OK;
}

int pt_synchronize(pt_ctx* ctx) {
  PT_REQUIRE_CTX();
  PT_HIP(hipStreamSynchronize(ctx->stream));
  return resolve_shared_cal(ctx);   // free by now
}

int pt_set_option(pt_ctx* ctx, int option, int64_t value) {
  PT_REQUIRE_CTX();
  switch (option) {
    case PT_OPT_COUNT_TRAVERSAL: ctx->count_traversal = value != 0; return PT_OK;
    case PT_OPT_TIME_KERNELS: ctx->time_kernels = value != 0; return PT_OK;
    case PT_OPT_SAMPLES_PER_BATCH:
      if (value < 0 || value > 4096) return ctx->fail(PT_ERR_INVALID_ARG, "samples per batch must be 0..4096");
      ctx->samples_per_batch = value;
      return PT_OK;
    case PT_OPT_RUSSIAN_ROULETTE: ctx->russian_roulette = value != 0; return PT_OK;
    case PT_OPT_PIPELINE: ctx->pipeline = value != 0; return PT_OK;
    case PT_OPT_SKY_MODE:
      if (value != 0 && value != 1) return ctx->fail(PT_ERR_UNSUPPORTED, "sky mode must be 0 (ray-march on every miss) or 1 (direction LUT)");
      ctx->sky_mode = (int)value;
      return PT_OK;
    case PT_OPT_FRAME_LANES:
      if (value < 1 || value > pt_ctx::kMaxFrameLanes) return ctx->fail(PT_ERR_INVALID_ARG, "PT_OPT_FRAME_LANES must be 1..8");
      ctx->frame_lanes = (int)value;
      return PT_OK;
    case PT_OPT_PRIMARY_PACKET: ctx->primary_packet = value < 0 ? 0 : (value > 2 ? 2 : (int)value); return PT_OK;
    case PT_OPT_SHARED_TRIANGLE_PHASE: ctx->shared_tri = value < 0 ? 0 : (value > 2 ? 2 : (int)value); return PT_OK;
    case PT_OPT_FUSED_PRIMARY:
      if (value != 0 && value != 1) return ctx->fail(PT_ERR_INVALID_ARG, "PT_OPT_FUSED_PRIMARY must be 0 or 1");
      ctx->fused_primary = value != 0;
      return PT_OK;
    default: return ctx->fail(PT_ERR_INVALID_ARG, "unknown option (the traversal kernel's tuning knobs go through pt_debug_set_tuning, include/pt_api_tuning.h)");
  }
}

// include/pt_api_tuning.h: the schedule knobs of k_extend (no counterpart in the reference; none changes a hit record)
int pt_debug_set_tuning(pt_ctx* ctx, int knob, int64_t value) {
  PT_REQUIRE_CTX();
  switch (knob) {
    case PT_TUNE_EXTEND_FETCH_MIN:
      if (value < 1 || value > 64) return ctx->fail(PT_ERR_INVALID_ARG, "fetch_min must be 1..64");
      ctx->cfg.fetch_min = (uint32_t)value;
      return PT_OK;
    case PT_TUNE_EXTEND_POSTPONE_PCT:
      if (value < 0 || value > 100) return ctx->fail(PT_ERR_INVALID_ARG, "postpone_pct must be 0..100");
      ctx->cfg.postpone_pct = (uint32_t)value;
      return PT_OK;
    case PT_TUNE_EXTEND_CONTINUE_PCT:
      if (value < 0 || value > 100) return ctx->fail(PT_ERR_INVALID_ARG, "continue_pct must be 0..100");
      ctx->cfg.continue_pct = (uint32_t)value;
      return PT_OK;
    case PT_TUNE_SHARED_DEFER:
      if (value < 1 || value > 16) return ctx->fail(PT_ERR_INVALID_ARG, "shared phase: parked groups per lane must be 1..16");
      ctx->cfg.sh_defer_entries = (uint32_t)value;
      return PT_OK;
    case PT_TUNE_SHARED_TOP_NODES:
      if (value < 0 || value > 585) return ctx->fail(PT_ERR_INVALID_ARG, "top_nodes must be 0..585");
      ctx->cfg.sh_top_nodes = (uint32_t)value;
      return PT_OK;
    case PT_TUNE_SHARED_POSTPONE_PCT:
      if (value < 0 || value > 100) return ctx->fail(PT_ERR_INVALID_ARG, "postpone_pct must be 0..100");
      ctx->cfg.sh_postpone_pct = (uint32_t)value;
      return PT_OK;
    case PT_TUNE_SHARED_CONTINUE_PCT:
      if (value < 0 || value > 100) return ctx->fail(PT_ERR_INVALID_ARG, "continue_pct must be 0..100");
      ctx->cfg.sh_continue_pct = (uint32_t)value;
      return PT_OK;
    case PT_TUNE_EXTEND_LDS_STACK:
      if (value < 1 || value > 128) return ctx->fail(PT_ERR_INVALID_ARG, "LDS stack entries must be 1..128");
      ctx->cfg.lds_stack_entries = (uint32_t)value;
      break;
    case PT_TUNE_EXTEND_DEFER:
      if (value < 0 || value > 16) return ctx->fail(PT_ERR_INVALID_ARG, "deferred-triangle queue must be 0..16 entries");
      ctx->cfg.defer_entries = (uint32_t)value;
      break;
    case PT_TUNE_EXTEND_TOP_NODES:
      if (value < 0 || value > 1024) return ctx->fail(PT_ERR_INVALID_ARG, "top nodes kept in LDS must be 0..1024");
      ctx->cfg.top_nodes = (uint32_t)value;
      break;
    case PT_TUNE_EXTEND_BLOCKS_PER_CU:
      if (value < 0 || value > 16) return ctx->fail(PT_ERR_INVALID_ARG, "blocks per CU must be 0..16");
      ctx->cfg.extend_blocks_per_cu = (int)value;
      break;
    default: return ctx->fail(PT_ERR_INVALID_ARG, "unknown tuning knob");
  }
  if (ctx->has_scene) ctx->cfg.extend_blocks = extend_grid(ctx);   // these four change the kernel's LDS footprint or grid
  return PT_OK;
}

int pt_scene_create(pt_ctx* ctx, const PtMeshDesc* meshes, uint32_t num_meshes, const PtInstanceDesc* instances,
                    uint32_t num_instances, const PtMaterial* materials, uint32_t num_materials, uint32_t flags) {
  PT_REQUIRE_CTX();
  FlatScene fs;
  std::string err;
  int rc = flatten_scene(meshes, num_meshes, instances, num_instances, materials, num_materials, flags, fs, err);
  if (rc != PT_OK) return ctx->fail(rc, err);
  Bvh8 bvh;
  build_bvh8(fs.tri_verts.data(), fs.num_triangles(), bvh, 0, (flags & PT_SCENE_PAIR_TRIANGLES) ? 1 : ((flags & PT_SCENE_NO_PAIRS) ? 0 : -1),
             (flags & PT_SCENE_FAST_BUILD) != 0);

  // the traversal kernel addresses nodes and triangle records with 32-bit byte offsets
  if (bvh.nodes.size() * (size_t)(PT_NODE_STRIDE_U4 * 16) >= (1ull << 32) || bvh.num_records() * bvh.record_bytes() >= (1ull << 32))
    return ctx->fail(PT_ERR_UNSUPPORTED, "scene too large: more than 4 GB of BVH nodes or of leaf records");
  PT_HIP(hipStreamSynchronize(ctx->stream));
  free_list(ctx->scene_allocs);
  if (ctx->instance_ext) { (void)hipFree(ctx->instance_ext); ctx->instance_ext = nullptr; }
  if (ctx->albedo_tex) { (void)hipFree(ctx->albedo_tex); ctx->albedo_tex = nullptr; }
  ctx->has_scene = false;
  ctx->ds = DeviceScene();

  uint4 *d_nodes, *d_tris, *d_prim;
  float4* d_ps;
  float2* d_uv;
  if ((rc = dev_alloc(ctx, &d_nodes, bvh.nodes.size() * PT_NODE_STRIDE_U4, &ctx->scene_allocs))) return rc;
  if ((rc = dev_alloc(ctx, &d_tris, bvh.num_records() * bvh.record_bytes() / 16, &ctx->scene_allocs))) return rc;
  if ((rc = dev_alloc(ctx, &d_prim, (size_t)fs.num_triangles(), &ctx->scene_allocs))) return rc;
  if ((rc = dev_alloc(ctx, &d_uv, (size_t)fs.num_vertices(), &ctx->scene_allocs))) return rc;
  if ((rc = dev_alloc(ctx, &d_ps, (size_t)fs.num_triangles() * 4, &ctx->scene_allocs))) return rc;
  // LoadInstanceData + LoadMaterialData (Common.hlsl:75-100) resolved once per instance on the host; the vertex normals,
  // instance -> material table and materials themselves only live in the ClosestHit rows below (no kernel reads them)
  std::vector<float> inst_mat(fs.instance_material.size() * 4);
  for (size_t i = 0; i < fs.instance_material.size(); ++i)
    memcpy(&inst_mat[4 * i], &fs.materials[4 * (size_t)fs.instance_material[i]], 16);
  auto up = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
    return bytes ? hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
  };
#if PT_NODE_STRIDE_U4 == 5
  PT_HIP(up(d_nodes, bvh.nodes.data(), bvh.nodes.size() * sizeof(PtBvh8Node)));
#else
  {
    std::vector<uint8_t> padded(bvh.nodes.size() * PT_NODE_STRIDE_U4 * 16, 0);
    for (size_t i = 0; i < bvh.nodes.size(); ++i) memcpy(&padded[i * PT_NODE_STRIDE_U4 * 16], &bvh.nodes[i], sizeof(PtBvh8Node));
    PT_HIP(up(d_nodes, padded.data(), padded.size()));
  }
#endif
  PT_HIP(up(d_tris, bvh.records(), bvh.num_records() * bvh.record_bytes()));
  PT_HIP(up(d_prim, fs.prim_data.data(), fs.prim_data.size() * 4));
  PT_HIP(up(d_uv, fs.vertex_uv.data(), fs.vertex_uv.size() * 4));
  {
    // ClosestHit record per triangle (PathTracing.hlsl:16-38 reads instance -> material, 3 indices, 3 x VertexData: four
    // dependent fetches; here they are resolved at upload into one 64-byte row)
    const size_t nt = fs.num_triangles();
    std::vector<float> rows(nt * 16);
    for (size_t t = 0; t < nt; ++t) {
      const uint32_t inst = fs.prim_data[4 * t + 3];
      float* r = &rows[16 * t];
      for (int k = 0; k < 3; ++k) {
        memcpy(r + 4 * k, &fs.vertex_normal[4 * (size_t)fs.prim_data[4 * t + k]], 12);
        r[4 * k + 3] = 0.0f;
      }
      memcpy(r + 3, &inst, 4);
      memcpy(r + 12, &inst_mat[4 * (size_t)inst], 16);
      // Decode_Unorm_RGBA (byte / 255, R in the low byte) done here once, with the same IEEE division the kernel and the
      // oracle would do per hit: r, g in the free lanes of the normal rows, b over the packed colour
      uint32_t rgba;
      memcpy(&rgba, r + 15, 4);
      r[7] = (float)(rgba & 0xffu) / 255.0f;
      r[11] = (float)((rgba >> 8) & 0xffu) / 255.0f;
      r[15] = (float)((rgba >> 16) & 0xffu) / 255.0f;
    }
    PT_HIP(up(d_ps, rows.data(), rows.size() * 4));
  }

  ctx->ds.nodes = d_nodes;
  ctx->ds.tris = d_tris;
  ctx->ds.paired = bvh.paired;
  ctx->num_records = bvh.num_records();
  ctx->shared_possible = bvh.num_records() < (1ull << 26);
  ctx->ds.prim_data = d_prim;
  ctx->ds.vertex_uv = d_uv;
  ctx->ds.prim_shade = d_ps;
  ctx->ds.num_nodes = (uint32_t)bvh.nodes.size();
  for (int a = 0; a < 3; ++a) ctx->ds.scene_mag = std::max(ctx->ds.scene_mag, std::max(std::fabs(bvh.lo[a]), std::fabs(bvh.hi[a])));
  ctx->ds.stack_entries = 2 * bvh.max_depth + 2;   // <=1 sibling group + <=1 postponed triangle group per level
  if (bvh.max_depth > 62)
    return ctx->fail(PT_ERR_UNSUPPORTED, "BVH deeper than the traversal stack supports (" + std::to_string(bvh.max_depth) + " levels)");
  {
    // spill region sized for the largest persistent grid we ever launch (16 blocks per CU x 4 waves)
    uint2* d_spill = nullptr;
    if ((rc = dev_alloc(ctx, &d_spill, spill_region_entries(ctx), &ctx->scene_allocs))) return rc;
    ctx->ds.spill = d_spill;
    if ((rc = dev_alloc(ctx, &ctx->spill2, spill_region_entries(ctx), &ctx->scene_allocs))) return rc;   // pipeline lane 2
    release_frame_lane_spills(ctx);   // re-allocated for this scene's stack depth on their next use
    ctx->fork_epoch++;
  }
  ctx->cfg.extend_blocks = extend_grid(ctx);
  ctx->instance_material = fs.instance_material;
  ctx->num_materials = num_materials;

  PtSceneInfo& in = ctx->info;
  in.num_triangles = fs.num_triangles();
  in.num_vertices = fs.num_vertices();
  in.num_instances = num_instances;
  in.num_materials = num_materials;
  in.num_bvh_nodes = (uint32_t)bvh.nodes.size();
  in.bvh_max_depth = bvh.max_depth;
  in.bvh_bytes = bvh.nodes.size() * sizeof(PtBvh8Node) + bvh.num_records() * bvh.record_bytes();
  for (int a = 0; a < 3; ++a) { in.bounds_min[a] = bvh.lo[a]; in.bounds_max[a] = bvh.hi[a]; }
  in.build_ms = bvh.build_ms;
  ctx->has_scene = true;
  ctx->packet_blocks = 0;   // depends on the scene (stack depth, record format)
  ctx->packet_choice = -1;
  drop_shared_cal(ctx);
  ctx->shared_choice = -1;
  ctx->shared_cal_ms[0] = ctx->shared_cal_ms[1] = 0.0f;
  return PT_OK;
}

int pt_scene_set_material_ext(pt_ctx* ctx, const PtMaterialExt* ext, uint32_t num_materials) {
  PT_REQUIRE_CTX();
  if (!ctx->has_scene) return ctx->fail(PT_ERR_NO_SCENE, "pt_scene_set_material_ext before pt_scene_create");
  PT_HIP(hipStreamSynchronize(ctx->stream));
  if (!ext || num_materials == 0) {
    if (ctx->instance_ext) { (void)hipFree(ctx->instance_ext); ctx->instance_ext = nullptr; }
    ctx->ds.instance_ext = nullptr;
    return PT_OK;
  }
  if (num_materials != ctx->num_materials) return ctx->fail(PT_ERR_SIZE_MISMATCH, "one PtMaterialExt per material of the scene is required");
  for (uint32_t m = 0; m < num_materials; ++m) {
    const PtMaterialExt& e = ext[m];
    if (!(e.roughness >= 0.0f && e.roughness <= 1.0f) || !(e.metallic >= 0.0f && e.metallic <= 1.0f) || e.albedo_pattern > 3u ||
        !std::isfinite(e.pattern_scale) || (e.flags & ~PT_MAT_GGX) != 0u)
      return ctx->fail(PT_ERR_INVALID_ARG, "PtMaterialExt " + std::to_string(m) + ": roughness/metallic outside [0,1], unknown pattern or flag");
  }
  std::vector<float> rows(ctx->instance_material.size() * 4);   // resolved per instance, like instance_mat4
  for (size_t i = 0; i < ctx->instance_material.size(); ++i) {
    const PtMaterialExt& e = ext[ctx->instance_material[i]];
    const uint32_t bits = (e.albedo_pattern & 0xffu) | ((e.flags & PT_MAT_GGX) << 8);
    rows[4 * i] = e.roughness; rows[4 * i + 1] = e.metallic; rows[4 * i + 2] = e.pattern_scale;
    memcpy(&rows[4 * i + 3], &bits, 4);
  }
  if (!ctx->instance_ext) {
    int rc;
    if ((rc = dev_alloc(ctx, &ctx->instance_ext, ctx->instance_material.size(), nullptr))) return rc;
  }
  if (!rows.empty()) PT_HIP(hipMemcpy(ctx->instance_ext, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
  ctx->ds.instance_ext = ctx->instance_ext;
  ctx->fork_epoch++;
  return PT_OK;
}

int pt_scene_set_albedo_texture(pt_ctx* ctx, const uint8_t* rgba8, uint32_t width, uint32_t height) {
  PT_REQUIRE_CTX();
  if (!ctx->has_scene) return ctx->fail(PT_ERR_NO_SCENE, "pt_scene_set_albedo_texture before pt_scene_create");
  PT_HIP(hipStreamSynchronize(ctx->stream));   // (covers the frame lanes: their resolves are queued on the ctx stream)
  if (ctx->albedo_tex) { (void)hipFree(ctx->albedo_tex); ctx->albedo_tex = nullptr; }
  ctx->ds.albedo_tex = nullptr;
  ctx->ds.tex_w = ctx->ds.tex_h = 0;
  ctx->fork_epoch++;
  if (!rgba8) return PT_OK;
  if (width == 0 || height == 0 || width > 16384u || height > 16384u) return ctx->fail(PT_ERR_INVALID_ARG, "albedo image must be 1..16384 texels on each side");
  int rc;
  if ((rc = dev_alloc(ctx, &ctx->albedo_tex, (size_t)width * height, nullptr))) return rc;
  PT_HIP(hipMemcpy(ctx->albedo_tex, rgba8, (size_t)width * height * 4, hipMemcpyHostToDevice));
  ctx->ds.albedo_tex = ctx->albedo_tex;
  ctx->ds.tex_w = width;
  ctx->ds.tex_h = height;
  return PT_OK;
}

int pt_scene_info(const pt_ctx* ctx, PtSceneInfo* out) {
  if (!ctx || !out) return PT_ERR_INVALID_ARG;
  if (!ctx->has_scene) return PT_ERR_NO_SCENE;
  *out = ctx->info;
  return PT_OK;
}

int pt_sky_set(pt_ctx* ctx, const PtAtmosphere* atmosphere) {
  PT_REQUIRE_CTX();
  if (!atmosphere) return ctx->fail(PT_ERR_INVALID_ARG, "atmosphere is null");
  if (!ctx->lut) {
    void* v = nullptr;
    PT_HIP(hipMalloc(&v, 256 * 64 * sizeof(uint2)));
    ctx->lut = (uint2*)v;
  }
  if (!ctx->skyview) {   // direction LUT of sky mode 1 (1 MB), filled per pt_render
    void* v = nullptr;
    PT_HIP(hipMalloc(&v, (size_t)PT_SKYVIEW_TEXELS * sizeof(float4)));
    ctx->skyview = (float4*)v;
  }
  launch_transmittance_lut(*atmosphere, ctx->lut, ctx->stream);
  PT_HIP(hipGetLastError());
  ctx->fork_epoch++;   // frame lanes must not read the LUT before this kernel has run
  ctx->has_sky = true;
  return PT_OK;
}

int pt_sky_lut_read(pt_ctx* ctx, uint16_t* rgba16f) {
  PT_REQUIRE_CTX();
  if (!rgba16f) return ctx->fail(PT_ERR_INVALID_ARG, "output pointer is null");
  if (!ctx->has_sky) return ctx->fail(PT_ERR_NO_SKY, "pt_sky_set has not been called");
  PT_HIP(hipMemcpyAsync(rgba16f, ctx->lut, 256 * 64 * sizeof(uint2), hipMemcpyDeviceToHost, ctx->stream));
  PT_HIP(hipStreamSynchronize(ctx->stream));
  return PT_OK;
}

int pt_accum_reset(pt_ctx* ctx, uint32_t width, uint32_t height) {
  PT_REQUIRE_CTX();
  if (width == 0 || height == 0 || (uint64_t)width * height > (1ull << 28)) return ctx->fail(PT_ERR_INVALID_ARG, "bad target size");
  if (!ctx->accum || !ctx->accum_owned || ctx->width != width || ctx->height != height) {
    PT_HIP(hipStreamSynchronize(ctx->stream));  // WaitForIdle before re-creating the texture (PathTracer.cpp:835)
    if (ctx->accum && ctx->accum_owned) (void)hipFree(ctx->accum);
    ctx->accum = nullptr;
    void* v = nullptr;
    PT_HIP(hipMalloc(&v, (size_t)width * height * sizeof(float4)));
    ctx->accum = (float4*)v;
    ctx->accum_owned = true;
    ctx->width = width;
    ctx->height = height;
  }
  PT_HIP(hipMemsetAsync(ctx->accum, 0, (size_t)width * height * sizeof(float4), ctx->stream));  // clear_texture.hlsl:8-12
  ctx->num_samples = 0;
  return PT_OK;
}

int pt_accum_bind(pt_ctx* ctx, float* device_rgba, uint32_t width, uint32_t height) {
  PT_REQUIRE_CTX();
  if (!device_rgba) {
    if (ctx->accum && !ctx->accum_owned) { ctx->accum = nullptr; ctx->width = ctx->height = 0; ctx->num_samples = 0; }
    return PT_OK;
  }
  if (width == 0 || height == 0 || (uint64_t)width * height > (1ull << 28)) return ctx->fail(PT_ERR_INVALID_ARG, "bad target size");
  if (((uintptr_t)device_rgba & 15u) != 0) return ctx->fail(PT_ERR_INVALID_ARG, "target must be 16-byte aligned");
  PT_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->accum && ctx->accum_owned) (void)hipFree(ctx->accum);
  ctx->accum = (float4*)device_rgba;
  ctx->accum_owned = false;
  ctx->width = width;
  ctx->height = height;
  PT_HIP(hipMemsetAsync(ctx->accum, 0, (size_t)width * height * sizeof(float4), ctx->stream));
  ctx->num_samples = 0;
  return PT_OK;
}

int pt_accum_device_ptr(pt_ctx* ctx, float** device_rgba, uint64_t* num_samples) {
  PT_REQUIRE_CTX();
  if (!ctx->accum) return ctx->fail(PT_ERR_NO_TARGET, "no accumulation target");
  if (device_rgba) *device_rgba = (float*)ctx->accum;
  if (num_samples) *num_samples = ctx->num_samples;
  return PT_OK;
}

int pt_accum_set_count(pt_ctx* ctx, uint64_t num_samples) {
  PT_REQUIRE_CTX();
  if (!ctx->accum) return ctx->fail(PT_ERR_NO_TARGET, "no accumulation target");
  ctx->num_samples = num_samples;
  return PT_OK;
}

int pt_accum_read_sum(pt_ctx* ctx, float* host_rgba_sum, uint64_t* num_samples) {
  PT_REQUIRE_CTX();
  if (!ctx->accum) return ctx->fail(PT_ERR_NO_TARGET, "no accumulation target");
  if (!host_rgba_sum) return ctx->fail(PT_ERR_INVALID_ARG, "output pointer is null");
  PT_HIP(hipMemcpyAsync(host_rgba_sum, ctx->accum, (size_t)ctx->width * ctx->height * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
  PT_HIP(hipStreamSynchronize(ctx->stream));
  if (num_samples) *num_samples = ctx->num_samples;
  return PT_OK;
}

int pt_accum_read(pt_ctx* ctx, float* host_rgba_mean) {
  uint64_t n = 0;
  int rc = pt_accum_read_sum(ctx, host_rgba_mean, &n);
  if (rc != PT_OK) return rc;
  if (n > 0) {
    const float fn = (float)n;
    const size_t cnt = (size_t)ctx->width * ctx->height * 4;
    for (size_t i = 0; i < cnt; ++i) host_rgba_mean[i] = host_rgba_mean[i] / fn;
  }
  return PT_OK;
}

int pt_render(pt_ctx* ctx, const PtRtConsts* consts, uint32_t width, uint32_t height, uint32_t first_sample, uint32_t num_samples) {
  PT_REQUIRE_CTX();
  int rc = check_consts(ctx, consts, true);
  if (rc != PT_OK) return rc;
  if (!ctx->has_scene) return ctx->fail(PT_ERR_NO_SCENE, "pt_render before pt_scene_create");
  if (!ctx->accum) return ctx->fail(PT_ERR_NO_TARGET, "pt_render before pt_accum_reset / pt_accum_bind");
  if (width != ctx->width || height != ctx->height) return ctx->fail(PT_ERR_SIZE_MISMATCH, "w,h differ from the accumulation target");
  if (num_samples == 0) return PT_OK;
  if ((uint64_t)first_sample + num_samples > 0xFFFFFFFFull) return ctx->fail(PT_ERR_INVALID_ARG, "sample index overflows 32 bits");

  const uint64_t wh = (uint64_t)width * height;
  // Samples per wavefront batch. Every kernel of a batch ends with a drain of ~0.1 ms (the last waves of the persistent
  // extend kernel finish their rays at memory latency, with nothing left to hide it), so batches are sized for 288 GB
  // of HBM: 2^27 paths (20 GB of queues and path state; 16 -> 64 spp per batch at 1080p: +3.2 % whole job) unless the
  // caller asks otherwise. If that much memory is not free the default is halved until the batch fits.
  const bool default_batch = ctx->samples_per_batch <= 0;
  uint64_t spb = default_batch ? std::max<uint64_t>(1, (1ull << 27) / wh) : (uint64_t)ctx->samples_per_batch;
  spb = std::min<uint64_t>(spb, num_samples);
  spb = std::min<uint64_t>(spb, std::max<uint64_t>(1, ((1ull << 31) - 1) / wh));
  // Small frames (the reference's interactive loop: one 640x360 sample per RenderRT call) go round-robin onto the frame
  // lanes, see pt_ctx::FrameLane: this call's kernels are queued on a lane stream and overlap the previous calls', only
  // its resolve is queued on the ctx stream. Not with per-kernel timing / counting (their figures assume one kernel at a
  // time), the two-lane pipeline (large batches) or the per-call sky-view LUT (one shared table).
  if (ctx->frame_lanes > 1 && num_samples <= spb && wh * num_samples < (1ull << 20) && !ctx->time_kernels && !ctx->count_traversal &&
      !(ctx->sky_mode == 1 && consts->mySampleSky)) {
    const int li = ctx->fl_next;
    pt_ctx::FrameLane& L = ctx->fl[li];
    if ((rc = ensure_frame_lane(ctx, L, wh * num_samples)) != PT_OK) return rc;
    const bool fused = ctx->fused_primary;
    bool packet = fused && ctx->primary_packet == 1;
    if (fused && ctx->primary_packet == 2) {
      if (ctx->packet_choice < 0 || ctx->packet_choice_w != width || ctx->packet_choice_h != height) {
        if ((rc = ensure_wavefront(ctx, wh * num_samples)) != PT_OK) return rc;
        if ((rc = calibrate_primary_kernel(ctx, *consts, width, height, first_sample, num_samples)) != PT_OK) return rc;
      }
      packet = ctx->packet_choice == 1;
    }
    ctx->fl_next = (li + 1) % ctx->frame_lanes;
    ctx->last_render_fused = fused;
    ctx->last_wb = &L.wb;
    if (L.seen_epoch != ctx->fork_epoch) {   // scene / sky LUT / stream changed since this lane last ran: order it behind the ctx stream once
      PT_HIP(hipEventRecord(ctx->ev_epoch, ctx->stream));
      PT_HIP(hipStreamWaitEvent(L.stream, ctx->ev_epoch, 0));
      L.seen_epoch = ctx->fork_epoch;
    } else if (L.resolve_recorded) {         // the lane's path state is free once the resolve that read it has run
      PT_HIP(hipStreamWaitEvent(L.stream, L.ev_resolved, 0));
    }
    const uint32_t bounces = consts->myMaxRecursionDepth + 1;
    const uint32_t k = num_samples;
    PT_HIP(hipMemsetAsync(L.wb.counters, 0, CTR_WORDS * sizeof(uint32_t), L.stream));
    if (fused) launch_primary_setup(*consts, L.wb, width, height, first_sample, k, L.stream);
    else launch_generate(*consts, L.wb, width, height, first_sample, k, L.stream);
    for (uint32_t b = 0; b < bounces; ++b) {
      const int in = (int)(b & 1u);
      extend_one(ctx, L.wb.ray_o[in], L.wb.ray_d[in], 1, L.wb.hits, L.wb.counters + CTR_AT(CTR_QCOUNT0 + b), L.wb.counters + CTR_AT(CTR_WORK0 + b), (int)b, false,
                 2 + li, (fused && b == 0) ? L.wb.gen : nullptr, /*packet=*/packet && b == 0, /*max_rays=*/wh * k);
      launch_shade(*consts, ctx->ds, L.wb, (int)b, width, height, first_sample, wh * k, ctx->russian_roulette, ctx->cfg, L.stream, fused);
    }
    if (consts->mySampleSky) launch_sky(*consts, ctx->lut, nullptr, L.wb, wh * k, ctx->cfg, L.stream);
    PT_HIP(hipEventRecord(L.ev_done, L.stream));
    PT_HIP(hipStreamWaitEvent(ctx->stream, L.ev_done, 0));
    launch_resolve(L.wb, ctx->accum, width, height, k, bounces, ctx->stream);   // call order == sample order on the ctx stream
    PT_HIP(hipEventRecord(L.ev_resolved, ctx->stream));
    L.resolve_recorded = true;
    PT_HIP(hipGetLastError());
    ctx->num_samples += num_samples;
    return PT_OK;
  }
  ctx->last_wb = &ctx->wb;

  // Two pipeline lanes: a batch of k >= 2 samples is split into a lower and an upper half that run on two streams with
  // their own queues, so one half's shade kernels and kernel drains run under the other half's extend. The halves are
  // resolved in sample order on the ctx stream, so the target is bit-identical to the single-lane result.
  bool two_lanes;
  for (;;) {
    two_lanes = ctx->pipeline && spb >= 2;
    const uint64_t spb_a = two_lanes ? (spb + 1) / 2 : spb;
    rc = ensure_wavefront(ctx, wh * spb_a);
    if (rc == PT_OK && two_lanes) rc = ensure_lane2(ctx, wh * (spb - spb_a));
    if (rc == PT_OK) break;
    if (rc != PT_ERR_OUT_OF_MEMORY || !default_batch || spb <= 1) return rc;
    (void)hipGetLastError();   // the failed hipMalloc's error is handled here, not by the launch check below
    spb = (spb + 1) / 2;
  }

  const uint32_t bounces = consts->myMaxRecursionDepth + 1;  // inclusive loop bound (PathTracing.hlsl:68)
  const bool use_skyview = ctx->sky_mode == 1 && consts->mySampleSky && skyview_valid_for(*consts);
  if (use_skyview) launch_skyview_lut(*consts, ctx->lut, ctx->skyview, ctx->stream);   // 64 Ki texels, a few microseconds
  const bool fused = ctx->fused_primary;   // primaries computed inside k_extend / k_shade instead of a k_generate pass
  ctx->last_render_fused = fused;
  ctx->shared_force = -1;
  if ((rc = resolve_shared_cal(ctx)) != PT_OK) return rc;
  // may batches of this call take part in the timing of the two triangle phases (resolve_shared_cal)? Large single-lane
  // batches only, and never while the caller is collecting statistics or timings of its own
  const bool may_time_shared = ctx->shared_tri == 2 && ctx->shared_possible && ctx->cfg.defer_entries > 0 && bounces >= 2 && !two_lanes &&
                               !ctx->count_traversal && !ctx->time_kernels;
  bool packet = fused && ctx->primary_packet == 1 && !ctx->count_traversal;
  if (fused && ctx->primary_packet == 2 && !ctx->count_traversal) {
    if (ctx->packet_choice < 0 || ctx->packet_choice_w != width || ctx->packet_choice_h != height) {
      const uint32_t k0 = (uint32_t)std::min<uint64_t>(two_lanes && spb >= 2 ? (spb + 1) / 2 : spb, num_samples);
      if ((rc = calibrate_primary_kernel(ctx, *consts, width, height, first_sample, std::max<uint32_t>(1u, k0))) != PT_OK) return rc;
    }
    packet = ctx->packet_choice == 1;
  }
  std::pair<hipEvent_t, hipEvent_t> rev{nullptr, nullptr};
  if (ctx->time_kernels) {   // after the one-off timing above: render_ms is the render, not the calibration
    rev = get_event_pair(ctx);
    PT_HIP(hipEventRecord(rev.first, ctx->stream));
  }
  for (uint32_t done = 0; done < num_samples;) {
    const uint32_t k = (uint32_t)std::min<uint64_t>(spb, num_samples - done);
    const uint32_t k_lane[2] = {two_lanes && k >= 2 ? (k + 1) / 2 : k, two_lanes && k >= 2 ? k / 2 : 0};
    const uint32_t first_lane[2] = {first_sample + done, first_sample + done + k_lane[0]};
    const int n_lanes = k_lane[1] ? 2 : 1;
    WavefrontBuffers* wbs[2] = {&ctx->wb, &ctx->wb2};
    const hipStream_t streams[2] = {ctx->stream, ctx->stream2};
    if (n_lanes == 2) {   // lane 1 starts after everything queued on the ctx stream so far (previous resolve included)
      PT_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
      PT_HIP(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    }
    for (int l = 0; l < n_lanes; ++l) {
      PT_HIP(hipMemsetAsync(wbs[l]->counters, 0, CTR_WORDS * sizeof(uint32_t), streams[l]));
      if (fused) launch_primary_setup(*consts, *wbs[l], width, height, first_lane[l], k_lane[l], streams[l]);
      else launch_generate(*consts, *wbs[l], width, height, first_lane[l], k_lane[l], streams[l]);
    }
    // this batch's part in the timing of the triangle phases: -1 none, 0..3 = the timed batches (even: per-lane, odd: shared)
    int timed = -1;
    if (may_time_shared && ctx->shared_choice < 0 && !ctx->shared_cal_pending && wh * k >= (1ull << 21)) {
      if (ctx->shared_cal_stage == 0) ctx->shared_cal_stage = 1;
      else if (ctx->shared_cal_stage == 1) { timed = 0; ctx->shared_cal_paths = wh * k; }
      else if (ctx->shared_cal_stage <= 4 && ctx->shared_cal_paths == wh * k) timed = ctx->shared_cal_stage - 1;
      else ctx->shared_cal_stage = 1;   // another batch size: start over with the next batch
    }
    if (timed >= 0) {
      ctx->shared_force = timed & 1;
      if (ctx->shared_cal_ev[timed].first) ctx->event_pool.push_back(ctx->shared_cal_ev[timed]);   // (a timing that started over)
      ctx->shared_cal_ev[timed] = get_event_pair(ctx);
    }
    for (uint32_t b = 0; b < bounces; ++b) {
      const int in = (int)(b & 1u);
      if (timed >= 0 && b == 1) PT_HIP(hipEventRecord(ctx->shared_cal_ev[timed].first, ctx->stream));
      for (int l = 0; l < n_lanes; ++l) {
        WavefrontBuffers& wb = *wbs[l];
        extend_one(ctx, wb.ray_o[in], wb.ray_d[in], 1, wb.hits, wb.counters + CTR_AT(CTR_QCOUNT0 + b), wb.counters + CTR_AT(CTR_WORK0 + b), (int)b, false, l,
                   (fused && b == 0) ? wb.gen : nullptr, /*packet=*/packet && b == 0);
        launch_shade(*consts, ctx->ds, wb, (int)b, width, height, first_lane[l], wh * k_lane[l], ctx->russian_roulette, ctx->cfg, streams[l], fused);
      }
    }
    if (timed >= 0) {
      PT_HIP(hipEventRecord(ctx->shared_cal_ev[timed].second, ctx->stream));
      ctx->shared_force = -1;
      ctx->shared_cal_stage = 2 + timed;
      if (timed == 3) ctx->shared_cal_pending = true;
    }
    for (int l = 0; l < n_lanes; ++l)
      if (consts->mySampleSky) launch_sky(*consts, ctx->lut, use_skyview ? ctx->skyview : nullptr, *wbs[l], wh * k_lane[l], ctx->cfg, streams[l]);
    if (n_lanes == 2) {
      PT_HIP(hipEventRecord(ctx->ev_join, ctx->stream2));
      PT_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    }
    for (int l = 0; l < n_lanes; ++l) launch_resolve(*wbs[l], ctx->accum, width, height, k_lane[l], bounces, ctx->stream);   // sample order
    done += k;
  }
  PT_HIP(hipGetLastError());
  if (ctx->time_kernels) {
    PT_HIP(hipEventRecord(rev.second, ctx->stream));
    ctx->render_events.push_back(rev);
  }
  ctx->num_samples += num_samples;
  return PT_OK;
}

int pt_render_ao(pt_ctx* ctx, const PtRtConsts* consts, uint32_t width, uint32_t height, uint32_t first_sample, uint32_t num_samples) {
  PT_REQUIRE_CTX();
  int rc = check_consts(ctx, consts, true);
  if (rc != PT_OK) return rc;
  if (!ctx->has_scene) return ctx->fail(PT_ERR_NO_SCENE, "pt_render_ao before pt_scene_create");
  if (!ctx->accum) return ctx->fail(PT_ERR_NO_TARGET, "pt_render_ao before pt_accum_reset / pt_accum_bind");
  if (width != ctx->width || height != ctx->height) return ctx->fail(PT_ERR_SIZE_MISMATCH, "w,h differ from the accumulation target");
  if (num_samples == 0) return PT_OK;
  if ((uint64_t)first_sample + num_samples > 0xFFFFFFFFull) return ctx->fail(PT_ERR_INVALID_ARG, "sample index overflows 32 bits");
  const uint64_t wh = (uint64_t)width * height;
  if (wh * 16u >= (1ull << 31)) return ctx->fail(PT_ERR_INVALID_ARG, "w*h*16 AO rays must stay below 2^31");
  if ((rc = ensure_wavefront(ctx, wh)) != PT_OK) return rc;
  if ((rc = ensure_ao(ctx, wh * 16u)) != PT_OK) return rc;

  std::pair<hipEvent_t, hipEvent_t> rev{nullptr, nullptr};
  if (ctx->time_kernels) {
    rev = get_event_pair(ctx);
    PT_HIP(hipEventRecord(rev.first, ctx->stream));
  }
  const bool use_skyview = ctx->sky_mode == 1 && consts->mySampleSky && skyview_valid_for(*consts);
  if (use_skyview) launch_skyview_lut(*consts, ctx->lut, ctx->skyview, ctx->stream);
  for (uint32_t s = 0; s < num_samples; ++s) {   // one reference frame (1 primary + <=16 AO rays per pixel) per batch
    PT_HIP(hipMemsetAsync(ctx->wb.counters, 0, CTR_WORDS * sizeof(uint32_t), ctx->stream));
    launch_generate(*consts, ctx->wb, width, height, first_sample + s, 1, ctx->stream);
    ctx->last_render_fused = false;
    ctx->last_wb = &ctx->wb;
    extend_one(ctx, ctx->wb.ray_o[0], ctx->wb.ray_d[0], 1, ctx->wb.hits, ctx->wb.counters + CTR_AT(CTR_QCOUNT0), ctx->wb.counters + CTR_AT(CTR_WORK0), 0);
    launch_ao_primary(*consts, ctx->wb, wh, ctx->cfg, ctx->stream);
    launch_ao_rays(*consts, ctx->ds, ctx->wb, ctx->ao_o, ctx->ao_d, width, height, first_sample + s, wh, ctx->cfg, ctx->stream);
    extend_one(ctx, ctx->ao_o, ctx->ao_d, 1, ctx->ao_hits, ctx->wb.counters + CTR_AT(CTR_QCOUNT0 + 1), ctx->wb.counters + CTR_AT(CTR_WORK0 + 1), 1,
               /*any_hit=*/true);
    launch_ao_count(ctx->wb, ctx->ao_hits, wh, ctx->cfg, ctx->stream);
    if (consts->mySampleSky) launch_sky(*consts, ctx->lut, use_skyview ? ctx->skyview : nullptr, ctx->wb, wh, ctx->cfg, ctx->stream);
    launch_resolve(ctx->wb, ctx->accum, width, height, 1, 2, ctx->stream);
  }
  PT_HIP(hipGetLastError());
  if (ctx->time_kernels) {
    PT_HIP(hipEventRecord(rev.second, ctx->stream));
    ctx->render_events.push_back(rev);
  }
  ctx->num_samples += num_samples;
  return PT_OK;
}

int pt_reduce(pt_ctx** ctxs, uint32_t n) {
  if (!ctxs || n == 0 || !ctxs[0]) return PT_ERR_INVALID_ARG;
  pt_ctx* ctx = ctxs[0];   // errors are reported on the destination context
  if (!ctx->accum) return ctx->fail(PT_ERR_NO_TARGET, "pt_reduce: ctxs[0] has no accumulation target");
  for (uint32_t i = 1; i < n; ++i) {
    if (!ctxs[i]) return ctx->fail(PT_ERR_INVALID_ARG, "pt_reduce: null context");
    if (!ctxs[i]->accum) return ctx->fail(PT_ERR_NO_TARGET, "pt_reduce: a source context has no accumulation target");
    if (ctxs[i]->width != ctx->width || ctxs[i]->height != ctx->height) return ctx->fail(PT_ERR_SIZE_MISMATCH, "pt_reduce: target sizes differ");
    if (ctxs[i] == ctx) return ctx->fail(PT_ERR_INVALID_ARG, "pt_reduce: ctxs[0] appears twice");
  }
  const size_t texels = (size_t)ctx->width * ctx->height;
  for (uint32_t i = 1; i < n; ++i) {   // everything queued on the sources must have landed before their targets are read
    if (hipSetDevice(ctxs[i]->device) != hipSuccess || hipStreamSynchronize(ctxs[i]->stream) != hipSuccess)
      return ctx->fail(PT_ERR_HIP, "pt_reduce: cannot synchronise a source context");
  }
  PT_REQUIRE_CTX();
  float4* staging = nullptr;
  int rc;
  if (n > 1 && (rc = dev_alloc(ctx, &staging, texels, nullptr))) return rc;
  hipError_t e = hipSuccess;
  for (uint32_t i = 1; i < n && e == hipSuccess; ++i) {
    e = hipMemcpyPeerAsync(staging, ctx->device, ctxs[i]->accum, ctxs[i]->device, texels * sizeof(float4), ctx->stream);
    if (e == hipSuccess) {
      launch_add(ctx->accum, staging, texels, ctx->stream);
      e = hipGetLastError();
    }
    if (e == hipSuccess) ctx->num_samples += ctxs[i]->num_samples;
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (staging) (void)hipFree(staging);
  if (e != hipSuccess) return ctx->fail(PT_ERR_HIP, std::string("pt_reduce: ") + hipGetErrorString(e));
  return PT_OK;
}

int pt_primary_hits(pt_ctx* ctx, const PtRtConsts* consts, uint32_t width, uint32_t height, uint32_t sample, PtHit* hits) {
  PT_REQUIRE_CTX();
  int rc = check_consts(ctx, consts, false);
  if (rc != PT_OK) return rc;
  if (!ctx->has_scene) return ctx->fail(PT_ERR_NO_SCENE, "pt_primary_hits before pt_scene_create");
  if (!hits || width == 0 || height == 0 || (uint64_t)width * height > (1ull << 28)) return ctx->fail(PT_ERR_INVALID_ARG, "bad arguments");
  const uint64_t wh = (uint64_t)width * height;
  if ((rc = ensure_wavefront(ctx, wh)) != PT_OK) return rc;
  PT_HIP(hipMemsetAsync(ctx->wb.counters, 0, CTR_WORDS * sizeof(uint32_t), ctx->stream));
  launch_generate(*consts, ctx->wb, width, height, sample, 1, ctx->stream);
  ctx->last_render_fused = false;
  ctx->last_wb = &ctx->wb;
  extend_one(ctx, ctx->wb.ray_o[0], ctx->wb.ray_d[0], 1, ctx->wb.hits, ctx->wb.counters + CTR_AT(CTR_QCOUNT0), ctx->wb.counters + CTR_AT(CTR_WORK0));
  PT_HIP(hipGetLastError());
  PT_HIP(hipMemcpyAsync(hits, ctx->wb.hits, wh * sizeof(PtHit), hipMemcpyDeviceToHost, ctx->stream));
  PT_HIP(hipStreamSynchronize(ctx->stream));
  return PT_OK;
}

int pt_trace_rays_device(pt_ctx* ctx, const PtRay* device_rays, uint64_t num_rays, PtHit* device_hits) {
  PT_REQUIRE_CTX();
  if (!ctx->has_scene) return ctx->fail(PT_ERR_NO_SCENE, "pt_trace_rays before pt_scene_create");
  if (num_rays == 0) return PT_OK;
  if (!device_rays || !device_hits || num_rays >= (1ull << 31)) return ctx->fail(PT_ERR_INVALID_ARG, "bad arguments");
  int rc;
  if ((rc = ensure_wavefront(ctx, 1)) != PT_OK) return rc;
  PT_HIP(hipMemsetAsync(ctx->wb.counters, 0, CTR_WORDS * sizeof(uint32_t), ctx->stream));
  launch_set_count(ctx->wb.counters + CTR_AT(CTR_QCOUNT0), (uint32_t)num_rays, ctx->stream);
  const float4* base = (const float4*)device_rays;  // AoS: {origin,tmin},{dir,tmax}
  extend_one(ctx, base, base + 1, 2, (float4*)device_hits, ctx->wb.counters + CTR_AT(CTR_QCOUNT0), ctx->wb.counters + CTR_AT(CTR_WORK0));
  PT_HIP(hipGetLastError());
  return PT_OK;
}

int pt_trace_rays_device_noalloc(pt_ctx* ctx, const PtRay* device_rays, uint64_t num_rays, PtHit* device_hits) {
  PT_REQUIRE_CTX();
  if (!ctx->has_scene) return ctx->fail(PT_ERR_NO_SCENE, "pt_trace_rays_device_noalloc before pt_scene_create");
  if (num_rays == 0) return PT_OK;
  if (!device_rays || !device_hits) return ctx->fail(PT_ERR_INVALID_ARG, "null ray or hit array");
  if (num_rays >= (1ull << 31)) return ctx->fail(PT_ERR_INVALID_ARG, "pt_trace_rays_device_noalloc: at most 2^31 - 1 rays per call");
  // The spill region of the per-lane stacks comes out of the static scratch, so the persistent grid is cut to the
  // waves it has room for; a BVH so deep that not even one block per CU fits is rejected.
  {   // the contract rests on k_extend_ns having no private segment: ask the runtime about the instantiation to be launched
    size_t priv = 0;
    const int qe = extend_ns_private_bytes(ctx->cfg.defer_entries > 0, ctx->ds.paired, &priv);
    if (qe != 0) return ctx->fail(PT_ERR_HIP, "pt_trace_rays_device_noalloc: hipFuncGetAttributes failed");
    if (priv != 0)
      return ctx->fail(PT_ERR_UNSUPPORTED, "pt_trace_rays_device_noalloc: this build's k_extend_ns uses " + std::to_string(priv) +
                                               " bytes of private segment per lane, which the runtime would allocate at launch");
  }
  LaunchCfg cfg = ctx->cfg;
  // k_extend pins its register budget and spills a few registers to the private segment, which the runtime allocates
  // on demand; this entry point launches the instantiation that has none (one wave per SIMD less), on its own grid
  cfg.no_private_segment = true;
  if (ctx->noalloc_blocks > 0) cfg.extend_blocks = std::min(cfg.extend_blocks, ctx->noalloc_blocks);   // never beyond the spill region sized for cfg.extend_blocks
  const uint32_t lds_entries = std::min<uint32_t>(std::min(ctx->ds.stack_entries, cfg.lds_stack_entries), PT_EXTEND_MAX_LDS_ENTRIES - cfg.defer_entries);
  const uint64_t spill_bytes_per_wave = (uint64_t)(ctx->ds.stack_entries - lds_entries) * 64u * sizeof(uint2);
  if (spill_bytes_per_wave) {
    const uint64_t blocks_fit = (PT_SCRATCH_BYTES - PT_SCRATCH_HEADER) / spill_bytes_per_wave / 4u;
    if (blocks_fit < (uint64_t)cfg.num_cus)
      return ctx->fail(PT_ERR_UNSUPPORTED, "pt_trace_rays_device_noalloc: the scene's BVH (" + std::to_string(ctx->info.bvh_max_depth) +
                                               " levels) needs " + std::to_string(spill_bytes_per_wave) + " bytes of stack spill per wave; the " +
                                               std::to_string(PT_SCRATCH_BYTES >> 20) + " MiB static scratch cannot hold one block per CU");
    cfg.extend_blocks = (int)std::min<uint64_t>((uint64_t)cfg.extend_blocks, blocks_fit);
  }
  uint32_t* count = (uint32_t*)ctx->scratch;
  uint32_t* cursor = (uint32_t*)(ctx->scratch + 256);
  PT_HIP(hipMemsetAsync(ctx->scratch, 0, PT_SCRATCH_HEADER, ctx->stream));   // the array keeps the last call's contents
  launch_set_count(count, (uint32_t)num_rays, ctx->stream);
  DeviceScene ds = ctx->ds;
  ds.spill = (uint2*)(ctx->scratch + PT_SCRATCH_HEADER);
  const float4* base = (const float4*)device_rays;
  launch_extend(ds, base, base + 1, 2, (float4*)device_hits, count, cursor, nullptr, false, false, cfg, ctx->stream);
  ctx->extend_launches++;
  PT_HIP(hipGetLastError());
  return PT_OK;
}

int pt_trace_rays(pt_ctx* ctx, const PtRay* rays, uint64_t num_rays, PtHit* hits) {
  PT_REQUIRE_CTX();
  if (!ctx->has_scene) return ctx->fail(PT_ERR_NO_SCENE, "pt_trace_rays before pt_scene_create");
  if (num_rays == 0) return PT_OK;
  if (!rays || !hits || num_rays >= (1ull << 31)) return ctx->fail(PT_ERR_INVALID_ARG, "bad arguments");
  int rc;
  if ((rc = ensure_tmp(ctx, num_rays)) != PT_OK) return rc;
  PT_HIP(hipMemcpyAsync(ctx->tmp_rays, rays, num_rays * sizeof(PtRay), hipMemcpyHostToDevice, ctx->stream));
  if ((rc = pt_trace_rays_device(ctx, (const PtRay*)ctx->tmp_rays, num_rays, (PtHit*)ctx->tmp_hits)) != PT_OK) return rc;
  PT_HIP(hipMemcpyAsync(hits, ctx->tmp_hits, num_rays * sizeof(PtHit), hipMemcpyDeviceToHost, ctx->stream));
  PT_HIP(hipStreamSynchronize(ctx->stream));
  return PT_OK;
}

int pt_occluded(pt_ctx* ctx, const PtRay* rays, uint64_t num_rays, uint8_t* occluded) {
  PT_REQUIRE_CTX();
  if (!ctx->has_scene) return ctx->fail(PT_ERR_NO_SCENE, "pt_occluded before pt_scene_create");
  if (num_rays == 0) return PT_OK;
  if (!rays || !occluded || num_rays >= (1ull << 31)) return ctx->fail(PT_ERR_INVALID_ARG, "bad arguments");
  int rc;
  if ((rc = ensure_tmp(ctx, num_rays)) != PT_OK) return rc;
  if ((rc = ensure_wavefront(ctx, 1)) != PT_OK) return rc;
  PT_HIP(hipMemcpyAsync(ctx->tmp_rays, rays, num_rays * sizeof(PtRay), hipMemcpyHostToDevice, ctx->stream));
  PT_HIP(hipMemsetAsync(ctx->wb.counters, 0, CTR_WORDS * sizeof(uint32_t), ctx->stream));
  launch_set_count(ctx->wb.counters + CTR_AT(CTR_QCOUNT0), (uint32_t)num_rays, ctx->stream);
  const float4* base = (const float4*)ctx->tmp_rays;
  extend_one(ctx, base, base + 1, 2, ctx->tmp_hits, ctx->wb.counters + CTR_AT(CTR_QCOUNT0), ctx->wb.counters + CTR_AT(CTR_WORK0), 0, /*any_hit=*/true);
  // flags are written over the front of the ray staging buffer (already consumed), then copied out
  uint8_t* d_flags = (uint8_t*)ctx->tmp_rays;
  launch_hit_flags(ctx->tmp_hits, num_rays, d_flags, ctx->stream);
  PT_HIP(hipGetLastError());
  PT_HIP(hipMemcpyAsync(occluded, d_flags, num_rays, hipMemcpyDeviceToHost, ctx->stream));
  PT_HIP(hipStreamSynchronize(ctx->stream));
  return PT_OK;
}

int pt_debug_eval(pt_ctx* ctx, int fn, const PtRtConsts* consts, const float* in, uint64_t n, float* out) {
  PT_REQUIRE_CTX();
  static const int in_w[PT_FN_COUNT] = {4, 6, 4, 6, 10, 5, 9, 6, 4, 6, 14, 4};
  static const int out_w[PT_FN_COUNT] = {8, 1, 1, 4, 1, 3, 4, 3, 6, 3, 8, 1};
  if (fn < 0 || fn >= PT_FN_COUNT) return ctx->fail(PT_ERR_INVALID_ARG, "unknown function id");
  if (n == 0) return PT_OK;
  if (!in || !out) return ctx->fail(PT_ERR_INVALID_ARG, "null array");
  PtRtConsts c;
  if (consts) c = *consts; else memset(&c, 0, sizeof(c));
  if ((fn == PT_FN_SKY || fn == PT_FN_SKY_VIEW_LUT) && c.mySampleSky && !ctx->has_sky) return ctx->fail(PT_ERR_NO_SKY, "pt_sky_set has not been called");
  float *d_in = nullptr, *d_out = nullptr;
  int rc;
  if ((rc = dev_alloc(ctx, &d_in, n * in_w[fn], nullptr))) return rc;
  if ((rc = dev_alloc(ctx, &d_out, n * out_w[fn], nullptr))) { (void)hipFree(d_in); return rc; }
  hipError_t e = hipMemcpyAsync(d_in, in, n * in_w[fn] * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    if (fn == PT_FN_SKY_VIEW_LUT && c.mySampleSky) launch_skyview_lut(c, ctx->lut, ctx->skyview, ctx->stream);
    launch_debug_eval(fn, c, ctx->lut, ctx->skyview, d_in, n, d_out, ctx->stream);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n * out_w[fn] * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  if (e != hipSuccess) return ctx->fail(PT_ERR_HIP, std::string("pt_debug_eval: ") + hipGetErrorString(e));
  return PT_OK;
}

int pt_debug_read_queue(pt_ctx* ctx, uint32_t bounce, PtRay* host_rays, uint64_t capacity, uint64_t* count_out) {
  PT_REQUIRE_CTX();
  const WavefrontBuffers& lw = ctx->last_wb ? *ctx->last_wb : ctx->wb;   // a small frame may have run on a frame lane
  if (!lw.counters || !count_out || bounce > 62) return ctx->fail(PT_ERR_INVALID_ARG, "no wavefront state or bad arguments");
  if (bounce == 0 && ctx->last_render_fused)
    return ctx->fail(PT_ERR_UNSUPPORTED, "the bounce-0 queue is not materialised under PT_OPT_FUSED_PRIMARY (set it to 0 before pt_render)");
  PT_HIP(hipStreamSynchronize(ctx->stream));
  uint32_t n = 0;
  PT_HIP(hipMemcpy(&n, lw.counters + CTR_AT(CTR_QCOUNT0 + bounce), sizeof(n), hipMemcpyDeviceToHost));
  *count_out = n;
  if (!host_rays) return PT_OK;
  if (capacity < n) return ctx->fail(PT_ERR_INVALID_ARG, "host array too small");
  std::vector<float4> o(n), d(n);
  const int buf = (int)(bounce & 1u);
  if (n) {
    PT_HIP(hipMemcpy(o.data(), lw.ray_o[buf], (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
    PT_HIP(hipMemcpy(d.data(), lw.ray_d[buf], (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
  }
  for (uint32_t i = 0; i < n; ++i) {
    PtRay& r = host_rays[i];
    r.origin[0] = o[i].x; r.origin[1] = o[i].y; r.origin[2] = o[i].z; r.tmin = o[i].w;
    r.dir[0] = d[i].x; r.dir[1] = d[i].y; r.dir[2] = d[i].z; r.tmax = d[i].w;
  }
  return PT_OK;
}

int pt_debug_read_bvh(pt_ctx* ctx, void* nodes, uint64_t node_capacity, void* tris, uint64_t tri_capacity, uint64_t* num_nodes,
                      uint64_t* num_tris, float* scene_mag, uint32_t* record_bytes) {
  PT_REQUIRE_CTX();
  if (!ctx->has_scene) return ctx->fail(PT_ERR_NO_SCENE, "pt_debug_read_bvh before pt_scene_create");
#if PT_NODE_STRIDE_U4 != 5
  return ctx->fail(PT_ERR_UNSUPPORTED, "pt_debug_read_bvh needs packed 80-byte nodes");
#else
  const uint64_t nn = ctx->info.num_bvh_nodes;
  const uint64_t nt = ctx->num_records;
  const size_t rb = ctx->ds.paired ? sizeof(PtPrimRecord) : sizeof(PtTriRecord);
  if (record_bytes) *record_bytes = (uint32_t)rb;
  if (num_nodes) *num_nodes = nn;
  if (num_tris) *num_tris = nt;
  if (scene_mag) *scene_mag = ctx->ds.scene_mag;
  if (!nodes && !tris) return PT_OK;
  if (!nodes || !tris || node_capacity < nn || tri_capacity < nt) return ctx->fail(PT_ERR_INVALID_ARG, "pt_debug_read_bvh: arrays too small");
  PT_HIP(hipStreamSynchronize(ctx->stream));
  if (nn) PT_HIP(hipMemcpy(nodes, ctx->ds.nodes, nn * sizeof(PtBvh8Node), hipMemcpyDeviceToHost));
  if (nt) PT_HIP(hipMemcpy(tris, ctx->ds.tris, nt * rb, hipMemcpyDeviceToHost));
  return PT_OK;
#endif
}

int pt_stats_get(pt_ctx* ctx, PtStats* out) {
  PT_REQUIRE_CTX();
  if (!out) return ctx->fail(PT_ERR_INVALID_ARG, "output pointer is null");
  // PtStats is versioned by size (append-only): fill a local copy, hand back as many bytes as the caller's header has
  const uint32_t caller_size = out->struct_size;
  if (caller_size < 16u || caller_size > (1u << 16))
    return ctx->fail(PT_ERR_INVALID_ARG, "PtStats.struct_size must be set to sizeof(PtStats) before pt_stats_get");
  PtStats local;
  PtStats* const caller_out = out;
  out = &local;
  memset(out, 0, sizeof(*out));
  PT_HIP(hipStreamSynchronize(ctx->stream));
  collect_events(ctx);
  {
    const int rc = resolve_shared_cal(ctx);
    if (rc != PT_OK) return rc;
  }
  std::vector<const WavefrontBuffers*> all_wb{&ctx->wb, &ctx->wb2};   // pipeline and frame lanes
  for (const auto& L : ctx->fl) all_wb.push_back(&L.wb);
  for (const WavefrontBuffers* wb : all_wb) {
    if (!wb->stats) continue;
    unsigned long long h[ST_WORDS];
    PT_HIP(hipMemcpy(h, wb->stats, sizeof(h), hipMemcpyDeviceToHost));
    out->samples += h[ST_SAMPLES];
    out->rays += h[ST_RAYS];
    out->shaded_hits += h[ST_SHADED];
    out->sky_misses += h[ST_SKY];
    out->nodes_visited += h[ST_NODES];
    out->tris_tested += h[ST_TRIS];
    out->node_wave_steps += h[ST_NODE_WAVE_STEPS];
    out->tri_wave_steps += h[ST_TRI_WAVE_STEPS];
    for (int b = 0; b < 64; ++b) out->rays_bounce[std::min(15, b)] += h[ST_RAYS_BOUNCE0 + b];
  }
  for (int b = 0; b < 16; ++b) out->extend_ms_bounce[b] = ctx->extend_ms_bounce[b];
  out->extend_launches = ctx->extend_launches;
  out->packet_launches = ctx->packet_launches;
  out->packet_cal_ms[0] = ctx->packet_cal_ms[0];
  out->packet_cal_ms[1] = ctx->packet_cal_ms[1];
  out->shared_launches = ctx->shared_launches;
  out->shared_cal_ms[0] = ctx->shared_cal_ms[0];
  out->shared_cal_ms[1] = ctx->shared_cal_ms[1];
  out->shared_choice = ctx->shared_choice;
  for (int b = 0; b < 16; ++b) out->extend_launches_bounce[b] = ctx->extend_launches_bounce[b];
  out->extend_ms = ctx->extend_ms;
  out->render_ms = ctx->render_ms;
  const uint32_t n = std::min<uint32_t>(caller_size, (uint32_t)sizeof(PtStats));
  local.struct_size = n;
  memcpy(caller_out, &local, n);
  return PT_OK;
}

int pt_stats_reset(pt_ctx* ctx) {
  PT_REQUIRE_CTX();
  PT_HIP(hipStreamSynchronize(ctx->stream));
  collect_events(ctx);
  ctx->extend_ms = 0.0;
  ctx->render_ms = 0.0;
  ctx->extend_launches = 0;
  ctx->packet_launches = 0;
  ctx->shared_launches = 0;
  for (int b = 0; b < 16; ++b) ctx->extend_launches_bounce[b] = 0;
  for (int b = 0; b < 16; ++b) ctx->extend_ms_bounce[b] = 0.0;
  if (ctx->wb.stats) PT_HIP(hipMemset(ctx->wb.stats, 0, ST_WORDS * sizeof(unsigned long long)));
  if (ctx->wb2.stats) PT_HIP(hipMemset(ctx->wb2.stats, 0, ST_WORDS * sizeof(unsigned long long)));
  for (auto& L : ctx->fl)
    if (L.wb.stats) PT_HIP(hipMemset(L.wb.stats, 0, ST_WORDS * sizeof(unsigned long long)));
  return PT_OK;
}

}  // extern "C"
