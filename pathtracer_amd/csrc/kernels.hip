// kernels.hip — the wavefront path tracer's HIP kernels for gfx950 (MI355X).
//
// One reference "frame" (DispatchRays(w,h,1) of RayGen, PathTracer/PathTracer.cpp:777-784, PathTracing.hlsl:40-138)
// is re-expressed as a pipeline over a batch of w*h*k path samples kept resident in HBM:
//   generate : InitRNG + jitter + GetPrimaryRay                      (PathTracing.hlsl:43-66)
//   extend   : TraceRay -> closest hit                               (PathTracing.hlsl:71)        [dominant kernel]
//   shade    : ClosestHit + BRDF sampling + throughput/emission      (PathTracing.hlsl:16-38,73-127), survivors are
//              compacted into the next bounce's ray queue with block-aggregated atomics; misses go to the sky queue
//   sky      : SampleSkyLuminance for every queued miss              (PathTracing.hlsl:75, SampleSky.hlsl:7-47)
//   resolve  : NaN/Inf scrub + accumulation in sample order          (PathTracing.hlsl:130-137)
// No kernel needs a host round trip: queue lengths live in device counters and the extend/shade/sky kernels are
// persistent / grid-stride loops over them.
#include <type_traits>
#include "kernels.h"

#include "shading_device.h"
#include "sky_device.h"
#include "ggx_device.h"
#include "traverse_device.h"

namespace pt {
using namespace ptd;

static_assert(PT_SKYVIEW_W * PT_SKYVIEW_H == PT_SKYVIEW_TEXELS, "sky-view LUT size");
static_assert(sizeof(PtRtConsts) == 288, "RtConsts is 288 bytes (PathTracer.cpp:709-740)");

// ------------------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Streaming accesses (queue records, hit records, path state: touched once per bounce) carry the non-temporal hint, so
// they do not push BVH nodes and ClosestHit rows out of L2 / Infinity Cache. PT_STREAM_NT=0 builds the plain forms.
#ifndef PT_STREAM_NT
#define PT_STREAM_NT 1
#endif
typedef float pt_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float4* p) {
#if PT_STREAM_NT
  const pt_f4v v = __builtin_nontemporal_load((const pt_f4v*)p);
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}
__device__ __forceinline__ uint32_t ld_stream(const uint32_t* p) {
#if PT_STREAM_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
__device__ __forceinline__ void st_stream(float4* p, float4 v) {
#if PT_STREAM_NT
  pt_f4v w;
  w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
  __builtin_nontemporal_store(w, (pt_f4v*)p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ void st_stream(uint32_t* p, uint32_t v) {
#if PT_STREAM_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// PT_DEBUG_BOUNDS=1 (make debug): every queue append checks its slot against the queue's capacity and traps with a message
// instead of writing past the buffer. The slots come from atomicAdd on ONE counter word per queue and at most one append
// per path per bounce reaches it, so slot < paths-in-batch <= capacity holds by construction; the check exists because
// an experiment that broke that invariant (round 2: the appends spread over several counter words)
// turned into out-of-bounds stores here and faulted the GPU.
#ifndef PT_DEBUG_BOUNDS
#define PT_DEBUG_BOUNDS 0
#endif
#if PT_DEBUG_BOUNDS
#define PT_CHECK_SLOT(slot, cap, what)                                                                                   \
  do {                                                                                                                   \
    if ((slot) >= (cap)) {                                                                                               \
      printf("pt: %s slot %u >= capacity %u (block %u thread %u)\n", what, (unsigned)(slot), (unsigned)(cap), blockIdx.x, threadIdx.x); \
      __builtin_trap();                                                                                                  \
    }                                                                                                                    \
  } while (0)
#else
#define PT_CHECK_SLOT(slot, cap, what) do { } while (0)
#endif

// Wave-aggregated queue append: one atomic per wave. Must be called by all 64 lanes convergently.
__device__ __forceinline__ uint32_t wave_append(uint32_t* counter, bool pred) {
  const unsigned long long mask = __ballot(pred);
  const uint32_t lane = lane_id();
  const uint32_t prefix = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
  uint32_t base = 0u;
  if (mask != 0ull) {
    const int leader = __ffsll((long long)mask) - 1;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = (uint32_t)__shfl((int)base, leader);
  }
  return base + prefix;
}


// Block-aggregated append into two queues at once: one returning atomic per queue per BLOCK iteration (a single
// counter word sustains only ~90 returning atomics/us chip-wide, so per-wave appends would bound the kernel).
// Must be reached by every thread of the block; `parity` alternates per call so one barrier pair suffices.
// 512 threads: two blocks are resident per CU (the kernel needs ~100 VGPRs), so one block's wait on its append
// atomic is covered by the other; 1024 was 5 % slower, 256 doubles the atomics and loses the gain again.
#define PT_SHADE_THREADS 512
struct BlockAppendLds {
  uint32_t cnt[2][2][PT_SHADE_THREADS / 64];   // [parity][queue][wave] -> count, then exclusive prefix
  uint32_t base[2][2];                          // [parity][queue]
};
__device__ __forceinline__ void block_append2(BlockAppendLds& L, int parity, uint32_t* counter0, bool pred0,
                                              uint32_t* counter1, bool pred1, uint32_t& slot0, uint32_t& slot1) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const unsigned long long m0 = __ballot(pred0), m1 = __ballot(pred1);
  if (lane == 0u) {
    L.cnt[parity][0][wave] = (uint32_t)__popcll(m0);
    L.cnt[parity][1][wave] = (uint32_t)__popcll(m1);
  }
  __syncthreads();
  if (threadIdx.x < 2u) {
    const uint32_t q = threadIdx.x;
    uint32_t run = 0u;
    for (uint32_t w = 0; w < nwaves; ++w) {
      const uint32_t c = L.cnt[parity][q][w];
      L.cnt[parity][q][w] = run;
      run += c;
    }
    L.base[parity][q] = run ? atomicAdd(q == 0u ? counter0 : counter1, run) : 0u;
  }
  __syncthreads();
  const unsigned long long below = (1ull << lane) - 1ull;
  slot0 = L.base[parity][0] + L.cnt[parity][0][wave] + (uint32_t)__popcll(m0 & below);
  slot1 = L.base[parity][1] + L.cnt[parity][1][wave] + (uint32_t)__popcll(m1 & below);
}

// ------------------------------------------------------------------------------------------------ transmittance LUT
// compute_transmittance_lut.hlsl main, [numthreads(8,8,1)] over 256x64 (Sky.cpp:145)
__global__ void __launch_bounds__(256) k_transmittance_lut(PtAtmosphere atm, uint2* __restrict__ lut) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= PT_TLUT_W * PT_TLUT_H) return;
  Atmo A;
  __builtin_memcpy(&A, &atm, sizeof(A));
  lut[i] = TransmittanceLutTexel(A, i % PT_TLUT_W, i / PT_TLUT_W);
}

// ------------------------------------------------------------------------------------------------ generate
__global__ void __launch_bounds__(256) k_generate(PtRtConsts c, float4* __restrict__ ray_o, float4* __restrict__ ray_d,
                                                  float4* __restrict__ st_T, float4* __restrict__ st_L,
                                                  uint32_t* __restrict__ counters, uint32_t w, uint32_t h,
                                                  uint32_t first_sample, uint32_t n_paths) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p == 0) counters[CTR_AT(CTR_QCOUNT0)] = n_paths;
  if (p >= n_paths) return;
  PrimaryFrame f;
  f.corner = ld3(c.myNearPlaneCorner); f.xAxis = ld3(c.myXAxis); f.yAxis = ld3(c.myYAxis); f.camPos = ld3(c.myCameraPos);
  f.w = w; f.h = h; f.wh = w * h; f.first_sample = first_sample; f.k = n_paths / (w * h);
  v3 o, d;
  const uint32_t rng_w = PrimaryRayOfPath(p, f, o, d);
  ray_o[p] = make_float4(o.x, o.y, o.z, 0.0f);       // TMin 0.0   (:59)
  ray_d[p] = make_float4(d.x, d.y, d.z, 10000.0f);   // TMax 10000 (:61)
  st_T[p] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(rng_w));
  st_L[p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// Fused primaries (PT_OPT_FUSED_PRIMARY, default): no k_generate pass. k_extend computes a primary ray when it fetches
// path p, bounce 0 of k_shade computes it again, and the path state starts as constants (T = 1, L = 0, two draws taken)
// instead of being written and read back: 64 B written + 96 B read per path less. This kernel leaves the frame for
// k_extend (4 x float4: corner|w, xAxis|h, yAxis|w*h, camPos|first_sample) and the queue count.
__global__ void k_primary_setup(PtRtConsts c, float4* __restrict__ gen, uint32_t* __restrict__ counters, uint32_t w, uint32_t h,
                                uint32_t first_sample, uint32_t n_paths) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  gen[0] = make_float4(c.myNearPlaneCorner[0], c.myNearPlaneCorner[1], c.myNearPlaneCorner[2], __uint_as_float(w));
  gen[1] = make_float4(c.myXAxis[0], c.myXAxis[1], c.myXAxis[2], __uint_as_float(h));
  gen[2] = make_float4(c.myYAxis[0], c.myYAxis[1], c.myYAxis[2], __uint_as_float(w * h));
  gen[3] = make_float4(c.myCameraPos[0], c.myCameraPos[1], c.myCameraPos[2], __uint_as_float(first_sample));
  gen[4] = make_float4(__uint_as_float(n_paths / (w * h)), 0.0f, 0.0f, 0.0f);   // samples per pixel of the batch (path ids are pixel-major)
  counters[CTR_AT(CTR_QCOUNT0)] = n_paths;
}

// ------------------------------------------------------------------------------------------------ extend
// Persistent waves. Each wave owns a local pool of queue slots (claimed from the global cursor in chunks, one atomic
// per chunk: a single word sustains only ~90 returning atomics/us chip-wide, so per-64-ray claims would bound the
// kernel) and keeps its 64 lanes busy: lanes whose ray has finished are refilled from the pool once enough of them are
// idle (fetch_min). Two schedules for the triangle tests: DEFER (default) parks triangle groups in a small per-lane
// queue and runs a triangle phase only when enough lanes have parked work; !DEFER postpones them on the traversal
// stack while fewer than postpone_pct % of the busy lanes have any. Either way the wave spends its instructions on box
// tests or on triangle tests with most lanes active.

// SHARED (deferred schedule only): the triangle phase pools the pending records of all 64 lanes and tests them 64 at a
// time, whoever owns them (extend_kernel.inc). It pays on scenes whose leaves are big and uneven (street_hard: triangle
// phase at 0.30 lane utilisation -> 0.72, extend +8 %) and costs about 1 % elsewhere (its LDS comes out of the top-node
// cache and the parked-group queue), so pt_api.cpp picks per scene by timing both once (PT_OPT_SHARED_TRIANGLE_PHASE).
#ifndef PT_SHARED_LIST
#define PT_SHARED_LIST 256u  // records per pass of a shared triangle phase (4 bytes of LDS each, per wave)
#endif
#ifndef PT_SHARED_TAKE
#define PT_SHARED_TAKE 4u    // records a lane offers per pass (64 x PT_SHARED_TAKE <= PT_SHARED_LIST)
#endif
static_assert(64u * PT_SHARED_TAKE <= PT_SHARED_LIST, "a pass of the shared triangle phase lists up to PT_SHARED_TAKE records per lane");
#ifndef PT_EXTEND_COMPACT_ADDR
#define PT_EXTEND_COMPACT_ADDR(PAIRS, SHARED) ((PAIRS) && (SHARED))   // which instantiations rebuild their stack / queue addresses at every access (extend_kernel.inc)
#endif
#ifndef PT_EXTEND_MIN_CLAIM
#define PT_EXTEND_MIN_CLAIM 4u   // rays of a wave's first claim when the queue has fewer than that per wave
#endif
#ifndef PT_EXTEND_STATIC_SHARE
#define PT_EXTEND_STATIC_SHARE 64u  // a queue with at most this many rays per wave of the grid is dealt out statically (extend_kernel.inc)
#endif
#ifndef PT_EXTEND_STATIC_FIRST
#define PT_EXTEND_STATIC_FIRST 0 // 0: a wave's first claim is static only when the queue has fewer than 64 rays per wave; 1: always (extend_kernel.inc)
#endif
#ifndef PT_EXTEND_NODE_STEPS
#define PT_EXTEND_NODE_STEPS 3   // box tests per scheduler iteration of the deferred schedule (1..4; C2 extend: 6.79, 6.95, 6.98, 6.83 Grays/s)
#endif
// 6 waves per SIMD (80 VGPRs) pinned: left to itself the register allocator lands between 78 and 90 VGPRs depending on
// small changes to the box test; at 6 waves with 3-5 registers spilled to the private segment (cold paths) the kernel is
// 1-3 % faster than at 5 waves without spills (tools/ab.sh).
#ifndef PT_EXTEND_WAVES_PER_EU
#define PT_EXTEND_WAVES_PER_EU 6
#endif
#define PT_EXTEND_NAME k_extend
#define PT_EXTEND_ATTR __attribute__((amdgpu_waves_per_eu(PT_EXTEND_WAVES_PER_EU, PT_EXTEND_WAVES_PER_EU)))
#include "extend_kernel.inc"
#undef PT_EXTEND_NAME
#undef PT_EXTEND_ATTR
// The same traversal with the register allocation left alone: no private segment, so a launch can never make the
// runtime allocate one. For pt_trace_rays_device_noalloc (callers that must not allocate).
#define PT_EXTEND_NAME k_extend_ns
#define PT_EXTEND_ATTR
#include "extend_kernel.inc"
#undef PT_EXTEND_NAME
#undef PT_EXTEND_ATTR

// ------------------------------------------------------------------------------------------------ extend, coherent packets
// Primary rays of a fused batch arrive as 8x8-pixel tiles (PrimaryRayOfSlot): the 64 rays of a wave share almost their
// whole path through the tree. This kernel walks the BVH ONCE PER WAVE: the node index is wave-uniform, so the 80-byte node
// and the triangle records come through the scalar cache into SGPRs (no per-lane gathers: the vector-memory pipe, which
// k_extend shares its time with, is idle here), every lane tests its own ray against the 8 child boxes, the hit masks are
// OR-ed across the wave and the wave descends into every child that ANY lane hits, nearest-first in the octant order of the
// wave's first ray. One stack per wave (LDS, 8 bytes per entry), no refill, no deferral: all lanes are active in every box
// and triangle test. A lane tests triangles its own ray's box test would have culled; the triangle test is exact and the
// closest hit with the lower-id tie-break does not depend on the order or on extra candidates, so the hit records are
// bit-identical to k_extend's (test_fused_primary_rays_are_bit_identical_to_the_generate_pass). Incoherent queues would
// visit the union of 64 unrelated paths here: pt_render uses it for bounce 0 only.
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
  // OR across the 64 lanes with DPP row shifts and the two row broadcasts; lane 63 ends up with the total
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);   // row_bcast:15 into rows 1 and 3
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);   // row_bcast:31 into rows 2 and 3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Box tests of one node for a COHERENT wave (all rays in one octant): what depends on the node alone — near / far plane
// choice, child bits and mask positions — is wave-uniform and compiles to scalar instructions; a lane spends its vector
// instructions on the 48 plane FMAs, the min / max reductions and the 8 compares, and a child enters the wave's hit mask
// when any lane's ray touches its box (the compare's lane mask is non-zero). A CU has ONE scalar unit for its four SIMDs
// (one instruction per cycle), so the split is balanced rather than "everything uniform on the scalar side": the byte
// -> half expansion of the 12 plane dwords is pinned to the vector pipe (a full-rate v_and_b32 and a v_perm_b32 each),
// and the hit mask is assembled from an 8-bit child mask, visiting only the children that were hit.
typedef _Float16 pt_h2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t packet_child_tests(uint32_t meta_lo, uint32_t meta_hi, const uint2& pnx, const uint2& pny, const uint2& pnz,
                                                       const uint2& pfx, const uint2& pfy, const uint2& pfz, float ax, float ay, float az, float bx,
                                                       float by, float bz, uint32_t oct4, float tmin, float bestt) {
  // pn* / pf*: the near / far plane dwords of the wave's octant (children 0-3 in .x, 4-7 in .y), fetched in that order
  const float btk = bestt * 1.000001f;   // the ~8 ulp widening of the far side (as in Traverser::child_tests), once per node
  uint32_t hit8 = 0u;                    // bit (7 - c): child c = 4 h + j was hit by some lane
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t nx = h ? pnx.y : pnx.x, ny = h ? pny.y : pny.x, nz = h ? pnz.y : pnz.x;
    const uint32_t fx = h ? pfx.y : pfx.x, fy = h ? pfy.y : pfy.x, fz = h ? pfz.y : pfz.x;
    // bytes (0, 2) and (1, 3) of a plane dword as the halves of one register: binary16 denormals, exactly byte * 2^-24
    uint32_t e_[6], o_[6];
    const uint32_t q_[6] = {nx, ny, nz, fx, fy, fz};
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      asm("v_and_b32 %0, %1, %2" : "=v"(e_[a]) : "s"(q_[a]), "v"(0x00ff00ffu));
      asm("v_perm_b32 %0, 0, %1, %2" : "=v"(o_[a]) : "s"(q_[a]), "v"(0x0c030c01u));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = j >> 1;
#define PT_PK_PLANE(a) (float)(__builtin_bit_cast(pt_h2u, (j & 1) ? o_[a] : e_[a])[k])
      const float tnx = __builtin_fmaf(PT_PK_PLANE(0), ax, bx);
      const float tny = __builtin_fmaf(PT_PK_PLANE(1), ay, by);
      const float tnz = __builtin_fmaf(PT_PK_PLANE(2), az, bz);
      const float tfx = __builtin_fmaf(PT_PK_PLANE(3), ax, bx);
      const float tfy = __builtin_fmaf(PT_PK_PLANE(4), ay, by);
      const float tfz = __builtin_fmaf(PT_PK_PLANE(5), az, bz);
#undef PT_PK_PLANE
      const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
      const float tf = fminf(fminf(tfx, tfy), tfz);
      const bool hit = tn <= fminf(tf * 1.000001f, btk);
      // hit8 = 2 hit8 + (some lane hit): the compare's lane mask against zero sets SCC, the add-with-carry shifts it in
      asm volatile("s_cmp_lg_u64 %1, 0\n\ts_addc_u32 %0, %0, %0" : "+s"(hit8) : "s"(__ballot(hit)) : "scc");
    }
  }
  uint32_t hm = 0u;
  while (hit8 != 0u) {   // wave-uniform, scalar: the children some lane hit (two or three of eight, typically)
    const int c = 7 - (31 - __clz((int)hit8));   // highest set bit first: child 0 sits at bit 7
    hit8 &= ~(0x80u >> c);
    const uint32_t meta4 = c >= 4 ? meta_hi : meta_lo;
    const uint32_t meta = (meta4 >> (8 * (c & 3))) & 0xffu;
    const uint32_t inner = ((meta & (meta << 1)) & 0x10u) ? 7u : 0u;
    hm |= (meta >> 5) << ((meta ^ (oct4 & inner)) & 0x1fu);
  }
  return hm;
}

#ifndef PT_PACKET_WAVES_PER_EU
#define PT_PACKET_WAVES_PER_EU 8   // 64 VGPRs: the coherent path fits; the rarely taken per-lane fallback spills 10 registers (6 / 7 / 8 waves: 10.97 / 11.10 / 11.18 Grays/s)
#endif
template <bool PAIRS>
__attribute__((amdgpu_waves_per_eu(PT_PACKET_WAVES_PER_EU, PT_PACKET_WAVES_PER_EU)))
__global__ void __launch_bounds__(256) k_extend_packet(const uint4* __restrict__ nodes, const uint4* __restrict__ tris,
                                                       const float4* __restrict__ ray_o, const float4* __restrict__ ray_d,
                                                       float4* __restrict__ hits, const uint32_t* __restrict__ count_ptr,
                                                       uint32_t* __restrict__ work_cursor, uint32_t stack_entries, float scene_mag,
                                                       const float4* __restrict__ gen) {
  extern __shared__ uint2 s_stack[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  lds_u2* stk = (lds_u2*)s_stack + (size_t)wave * stack_entries;
  const uint32_t n = *count_ptr;
  const uint32_t total_waves = gridDim.x * (blockDim.x >> 6);
  uint32_t chunk = (n / (total_waves * 8u) + 63u) & ~63u;
  chunk = chunk < 64u ? 64u : (chunk > 2048u ? 2048u : chunk);
  PrimaryFrame f;
  if (gen != nullptr) {
    const float4 g0 = gen[0], g1 = gen[1], g2 = gen[2], g3 = gen[3];
    f.corner = mk3(g0.x, g0.y, g0.z); f.xAxis = mk3(g1.x, g1.y, g1.z); f.yAxis = mk3(g2.x, g2.y, g2.z); f.camPos = mk3(g3.x, g3.y, g3.z);
    f.w = __float_as_uint(g0.w); f.h = __float_as_uint(g1.w); f.wh = __float_as_uint(g2.w); f.first_sample = __float_as_uint(g3.w);
    f.k = __float_as_uint(gen[4].x);
  }
  for (;;) {
    uint32_t base = 0u;
    if (lane == 0u) base = atomicAdd(work_cursor, chunk);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (base >= n) break;
    const uint32_t end = base + chunk < n ? base + chunk : n;
    for (uint32_t tile = base; tile < end; tile += 64u) {   // wave-uniform
      const uint32_t ray_i = tile + lane;
      const bool valid = ray_i < end;
      TraceRay r;
      r.ox = 0.0f; r.oy = 0.0f; r.oz = 0.0f; r.tmin = 0.0f; r.dx = 0.0f; r.dy = 0.0f; r.dz = 1.0f; r.tmax = -1.0f;   // a ray that hits nothing
      if (valid) {
        if (gen != nullptr) {
          v3 o, d;
          (void)PrimaryRayOfSlot(ray_i, f, o, d);
          r.ox = o.x; r.oy = o.y; r.oz = o.z; r.tmin = 0.0f;          // TMin 0.0, TMax 10000 (PathTracing.hlsl:59-61)
          r.dx = d.x; r.dy = d.y; r.dz = d.z; r.tmax = 10000.0f;
        } else {
          const float4 ro = ld_stream(ray_o + ray_i), rd = ld_stream(ray_d + ray_i);
          r.ox = ro.x; r.oy = ro.y; r.oz = ro.z; r.tmin = ro.w;
          r.dx = rd.x; r.dy = rd.y; r.dz = rd.z; r.tmax = rd.w;
        }
        const float chk = (r.ox + r.oy + r.oz) + (r.dx + r.dy + r.dz);
        if (!isfinite(chk)) { r.ox = 0.0f; r.oy = 0.0f; r.oz = 0.0f; r.tmin = 0.0f; r.dx = 0.0f; r.dy = 0.0f; r.dz = 1.0f; r.tmax = -1.0f; }   // as k_extend: a miss
      }
      Traverser tr;
      tr.init(r, scene_mag);
      const bool wave_risky = __ballot(tr.risky) != 0ull;
      // traversal order and the bit positions of the hit mask: the octant of the wave's first ray, for every lane (the
      // near / far planes of the slab test stay each lane's own)
      const uint32_t oct = (uint32_t)__builtin_amdgcn_readfirstlane((int)tr.oct());
      const uint32_t oct4 = oct * 0x01010101u;
      const bool negx = tr.idx < 0.0f, negy = tr.idy < 0.0f, negz = tr.idz < 0.0f;
      // coherent wave: every lane in the first ray's octant, with the same dominant axis and winding of the triangle test,
      // and no lane on the overflow-safe box test; then the per-node and per-triangle selections below are scalar
      const bool wnegx = (oct & 1u) != 0u, wnegy = (oct & 2u) != 0u, wnegz = (oct & 4u) != 0u;
      const int wkx = __builtin_amdgcn_readfirstlane(tr.tt.kx), wky = __builtin_amdgcn_readfirstlane(tr.tt.ky), wkz = __builtin_amdgcn_readfirstlane(tr.tt.kz);
      const bool coherent = !wave_risky && __ballot(tr.oct() != oct || tr.tt.kx != wkx || tr.tt.ky != wky || tr.tt.kz != wkz) == 0ull;
      // the ray origin in the triangle test's axis order (coherent waves: the same permutation for every lane)
      const float okx = sel3(tr.tt.ox, tr.tt.oy, tr.tt.oz, wkx), oky = sel3(tr.tt.ox, tr.tt.oy, tr.tt.oz, wky), okz = sel3(tr.tt.ox, tr.tt.oy, tr.tt.oz, wkz);
      // where the octant's near / far plane dwords sit among the node's six plane pairs (lo x, y, z, hi x, y, z): the
      // selection is done by the scalar loads' addresses, once per tile, not by selects per node
      const uint32_t inx = wnegx ? 3u : 0u, iny = wnegy ? 4u : 1u, inz = wnegz ? 5u : 2u;
      const uint32_t ifx = wnegx ? 0u : 3u, ify = wnegy ? 1u : 4u, ifz = wnegz ? 2u : 5u;
      uint32_t ngx = 0u, ngy = 0x80000000u;   // wave-uniform pending node group (root)
      uint32_t sp = 0u;
      for (;;) {
        if (ngy <= 0x00FFFFFFu) {
          if (sp == 0u) break;
          --sp;
          const pt_u2v e = stk[sp];
          ngx = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x);
          ngy = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y);
        }
        const uint32_t hitsw = ngy;
        const int b = 31 - __clz((int)hitsw);
        ngy &= ~(1u << b);
        if (ngy > 0x00FFFFFFu) {
          pt_u2v e;
          e.x = ngx; e.y = ngy;
          if (lane == 0u) stk[sp] = e;
          ++sp;
        }
        const uint32_t slot = ((uint32_t)b - 24u) ^ oct;
        const uint32_t rel = __popc(hitsw & ~(0xFFFFFFFFu << slot));
        const uint32_t ni = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ngx + rel));
        const uint4* np = nodes + (size_t)ni * PT_NODE_STRIDE_U4;
        const uint4 n0 = np[0], n1 = np[1];
        const uint32_t e = n0.w;
        const float px = __uint_as_float(n0.x) - tr.ray.ox, py = __uint_as_float(n0.y) - tr.ray.oy, pz = __uint_as_float(n0.z) - tr.ray.oz;
        uint32_t hm;
        if (coherent) {
#if PT_MIX_FMA
          const float sx = __uint_as_float(((e & 0xffu) << 23) + 0x0C000000u), sy = __uint_as_float((((e >> 8) & 0xffu) << 23) + 0x0C000000u),
                      sz = __uint_as_float((((e >> 16) & 0xffu) << 23) + 0x0C000000u);
          const uint2* pl = (const uint2*)(np + 2);
          hm = packet_child_tests(n1.z, n1.w, pl[inx], pl[iny], pl[inz], pl[ifx], pl[ify], pl[ifz], sx * tr.idx, sy * tr.idy, sz * tr.idz, px * tr.idx,
                                  py * tr.idy, pz * tr.idz, oct4, tr.ray.tmin, tr.best.t);
#else
#error "k_extend_packet's coherent path is written for PT_MIX_FMA"
#endif
        } else {
          const uint4 n2 = np[2], n3 = np[3], n4 = np[4];
          uint32_t hitmask = 0u;
          if (wave_risky && tr.risky) {
            const float sx = __uint_as_float((e & 0xffu) << 23), sy = __uint_as_float(((e >> 8) & 0xffu) << 23), sz = __uint_as_float(((e >> 16) & 0xffu) << 23);
            tr.template child_tests<true>(n1, n2, n3, n4, sx, sy, sz, px, py, pz, negx, negy, negz, oct4, hitmask);
          } else {
#if PT_MIX_FMA
            const float sx = __uint_as_float(((e & 0xffu) << 23) + 0x0C000000u), sy = __uint_as_float((((e >> 8) & 0xffu) << 23) + 0x0C000000u),
                        sz = __uint_as_float((((e >> 16) & 0xffu) << 23) + 0x0C000000u);
#else
            const float sx = __uint_as_float((e & 0xffu) << 23), sy = __uint_as_float(((e >> 8) & 0xffu) << 23), sz = __uint_as_float(((e >> 16) & 0xffu) << 23);
#endif
            tr.template child_tests<false>(n1, n2, n3, n4, sx * tr.idx, sy * tr.idy, sz * tr.idz, px * tr.idx, py * tr.idy, pz * tr.idz, negx, negy, negz,
                                           oct4, hitmask);
          }
          hm = wave_or(hitmask);
        }
        ngx = n1.x;
        ngy = (hm & 0xFF000000u) | (e >> 24);
        const uint32_t tbase = n1.y;
        uint32_t tbits = hm & 0x00FFFFFFu;
        while (tbits != 0u) {   // wave-uniform: every record some lane's box test let through
          const uint32_t ti = tbase + (uint32_t)(__ffs((int)tbits) - 1);
          tbits &= tbits - 1u;
          const uint4* tp = tris + (size_t)ti * (PAIRS ? 4 : 3);
          if (coherent) {
            // the vertex components arrive in the wave's axis order: three scalar dword loads per vertex at offsets fixed per
            // tile, no selects; the arithmetic is TriTester::shear's
            const uint32_t* vp = (const uint32_t*)tp;
#define PT_PK_SHEAR(v, S)                                                                                                  \
            {                                                                                                                 \
              (S).kz = __uint_as_float(vp[4 * (v) + wkz]) - okz;                                                              \
              (S).x = (__uint_as_float(vp[4 * (v) + wkx]) - okx) - tr.tt.Sx * (S).kz;                                          \
              (S).y = (__uint_as_float(vp[4 * (v) + wky]) - oky) - tr.tt.Sy * (S).kz;                                          \
            }
            TriTester::Sheared A, B, C;
            PT_PK_SHEAR(0, A) PT_PK_SHEAR(1, B) PT_PK_SHEAR(2, C)
            tr.tt.test_sheared(A, B, C, vp[3], tr.ray.tmin, tr.best);
            if (PAIRS && vp[7] != 0xFFFFFFFFu) {
              TriTester::Sheared D;
              PT_PK_SHEAR(3, D)
              const uint32_t code = vp[11];
              tr.tt.test_sheared(TriTester::pick(A, B, C, D, code & 3u), TriTester::pick(A, B, C, D, (code >> 2) & 3u),
                                 TriTester::pick(A, B, C, D, (code >> 4) & 3u), vp[7], tr.ray.tmin, tr.best);
            }
#undef PT_PK_SHEAR
            continue;
          }
          const uint4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
          uint4 q3 = make_uint4(0u, 0u, 0u, 0u);
          if (PAIRS) q3 = tp[3];
          if (PAIRS) {
            (void)tr.tt.test(q0, q1, q2, q3, tr.ray.tmin, tr.best);
          } else {
            tr.tt.test_sheared(tr.tt.shear(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z)),
                               tr.tt.shear(__uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z)),
                               tr.tt.shear(__uint_as_float(q2.x), __uint_as_float(q2.y), __uint_as_float(q2.z)), q0.w, tr.ray.tmin, tr.best);
          }
        }
      }
      if (valid) {
        HitRec hr = tr.best;
        if (hr.prim == 0xFFFFFFFFu) hr.t = -1.0f;
        st_stream(hits + ray_i, make_float4(hr.t, hr.u, hr.v, __uint_as_float(hr.prim)));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ shade
#ifndef PT_SHADE_MIN_WAVES
#define PT_SHADE_MIN_WAVES 1
#endif
template <bool EXT, bool FUSED>
__global__ void __launch_bounds__(PT_SHADE_THREADS, PT_SHADE_MIN_WAVES) k_shade(PtRtConsts c, DeviceScene sc, const float4* __restrict__ ray_o_in,
                                               const float4* __restrict__ ray_d_in, const uint32_t* __restrict__ qid_in,
                                               const float4* __restrict__ hits, float4* __restrict__ st_T,
                                               float4* __restrict__ st_L, float4* __restrict__ ray_o_out,
                                               float4* __restrict__ ray_d_out, uint32_t* __restrict__ qid_out,
                                               float4* __restrict__ sky_o, float4* __restrict__ sky_d,
                                               uint32_t* __restrict__ counters, int bounce, uint32_t w, uint32_t h,
                                               uint32_t first_sample, uint32_t russian_roulette, uint32_t spp, uint32_t capacity) {
  // spp: samples per pixel of the batch (path id = pixel * spp + sample); capacity: entries of every queue (PT_DEBUG_BOUNDS)
  // FUSED (only at bounce 0): no k_generate ran; the ray is recomputed from the path index, T = 1, L = 0 and the
  // RNG counter after the jitter are constants, and this kernel (k_sky for sky misses) gives st_L its first value
  const uint32_t n = counters[CTR_AT(CTR_QCOUNT0 + bounce)];
  __shared__ BlockAppendLds s_app;
  const uint32_t lane = threadIdx.x & 63u;
  const bool last_bounce = (uint32_t)bounce >= c.myMaxRecursionDepth;
  const uint32_t wh = w * h;
  uint32_t shaded_local = 0u, miss_local = 0u;
  const FastDiv div_w(w, wh);   // path ids stay below 2^31 (pt_api.cpp batch sizing)
  PrimaryFrame pf;
  pf.corner = ld3(c.myNearPlaneCorner); pf.xAxis = ld3(c.myXAxis); pf.yAxis = ld3(c.myYAxis); pf.camPos = ld3(c.myCameraPos);
  pf.w = w; pf.h = h; pf.wh = wh; pf.first_sample = first_sample; pf.k = spp;
  int parity = 0;
  for (uint32_t j0 = blockIdx.x * blockDim.x; j0 < n; j0 += gridDim.x * blockDim.x, parity ^= 1) {  // block-uniform trip count
    const uint32_t j = j0 + threadIdx.x;
    const bool active = j < n;
    bool want_ray = false, want_sky = false;
    uint32_t pid = 0u;
    v3 origin = mk3(0.0f), dir = mk3(0.0f), newOrigin = mk3(0.0f), newDir = mk3(0.0f);
    if (active) {
      pid = qid_in ? ld_stream(qid_in + j) : j;
      const float4 hr = ld_stream(hits + j);
      if (FUSED) {
        pid = PrimaryRayOfSlot(j, pf, origin, dir);   // the slot order k_extend fetched in
      } else {
        const float4 ro = ld_stream(ray_o_in + j), rd = ld_stream(ray_d_in + j);
        origin = mk3(ro.x, ro.y, ro.z);
        dir = mk3(rd.x, rd.y, rd.z);
      }
      const uint32_t prim = __float_as_uint(hr.w);
      if (prim == 0xFFFFFFFFu) {
        // miss: PathTracing.hlsl:73-77
        ++miss_local;
        if (c.mySampleSky) {
          want_sky = true;
        } else if (FUSED) {   // T = 1, L = 0
          const v3 add = mk3(1.0f) * ld3(c.mySkyFallbackEmission);
          st_L[pid] = make_float4(0.0f + add.x, 0.0f + add.y, 0.0f + add.z, 0.0f);
        } else {
          const float4 T = st_T[pid];
          float4 L = st_L[pid];
          const v3 add = mk3(T.x, T.y, T.z) * ld3(c.mySkyFallbackEmission);
          L.x += add.x; L.y += add.y; L.z += add.z;
          st_L[pid] = L;
        }
      } else {
        ++shaded_local;
        // ---- ClosestHit: PathTracing.hlsl:16-38, loaders Common.hlsl:75-127 ----
        // LoadTriangleIndices + InstanceID() + LoadInstanceData + LoadMaterialData + 3 x LoadVertexData, resolved at
        // upload into one 64-byte row per triangle: one dependent fetch instead of a chain of three
        const float4* ps = sc.prim_shade + (size_t)prim * 4;
        const float4 n0 = ps[0], n1 = ps[1], n2 = ps[2], mat = ps[3];
        const uint32_t instanceId = __float_as_uint(n0.w);
        const float bu = hr.y, bv = hr.z;                    // attrib.bary (DXR meaning, Q5)
        const float baryZ = 1.0f - (bu + bv);
        v3 hitNormal = bu * mk3(n0.x, n0.y, n0.z) + bv * mk3(n1.x, n1.y, n1.z) + baryZ * mk3(n2.x, n2.y, n2.z);
        const v3 hitPos = origin + dir * hr.x;
        v3 color = mk3(n1.w, n2.w, mat.w);   // Decode_Unorm_RGBA of the material colour, done at upload (pt_api.cpp)
        bool ggx = false;
        float roughness = 0.0f, metallic = 0.0f;
        if (EXT) {   // material extension (ggx_device.h): procedural albedo from the interpolated uv, GGX flag
          const float4 ext = sc.instance_ext[instanceId];
          const uint32_t bits = __float_as_uint(ext.w);
          if (bits & 0xffu) {
            const uint4 pd = sc.prim_data[prim];
            const float2 t0 = sc.vertex_uv[pd.x], t1 = sc.vertex_uv[pd.y], t2 = sc.vertex_uv[pd.z];
            v2 uv;   // LoadInterpolatedVertexData weights (Common.hlsl:121-126), as for the normal
            uv.x = bu * t0.x + bv * t1.x + baryZ * t2.x;
            uv.y = bu * t0.y + bv * t1.y + baryZ * t2.y;
            if ((bits & 0xffu) == 3u) color = color * SampleAlbedoTexture(sc.albedo_tex, sc.tex_w, sc.tex_h, ext.z, uv);
            else color = color * AlbedoPattern(bits & 0xffu, ext.z, uv);
          }
          ggx = ((bits >> 8) & PT_MAT_GGX_BIT) != 0u;
          roughness = ext.x;
          metallic = ext.y;
        }
        v3 emission = mk3(mat.x, mat.y, mat.z);
        if (dot(hitNormal, -dir) < 0.0f) hitNormal = -hitNormal;
        if (instanceId == c.myLightInstanceId) emission = ld3(c.myLightEmission);

        // ---- RayGen loop body: PathTracing.hlsl:83-127 ----
        float4 Ts;
        if (FUSED) Ts = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(2u));   // two jitter draws taken (Random.hlsl:96-99)
        else Ts = ld_stream(st_T + pid);
        v3 transmission = mk3(Ts.x, Ts.y, Ts.z);
        Rng rng;
        uint32_t pix, si;
        SplitPathId(pid, spp, si, pix);
        div_w.divmod(pix, rng.y, rng.x);
        rng.z = first_sample + si; rng.w = __float_as_uint(Ts.w);
        const float specularStrength = 0.9f;
        const float specularPower = c.myPhongSpecularPower;
        const v3 V = -dir;
        bool alive = true;
        if (EXT && ggx) {
          alive = SampleGgxMaterial(color, roughness, metallic, hitNormal, V, rng, transmission, newDir);
        } else {
        const float fresnel = GetFresnelSchlick(hitNormal, V);
        const float specRayProbability = EstimateSpecularRayProbability(specularStrength, color, fresnel);
        // the three draws are the same sequence on both sides of the branch (:86, then two in either sampler): taken
        // before it, so that a wave whose lanes split runs the generator once
        const float pick = GetRand01(rng);
        v2 u;
        u.x = GetRand01(rng);
        u.y = GetRand01(rng);
        if (pick < specRayProbability) {
          float pdf;
          const v3 nextSampleDir = SampleModifiedPhong(u, hitNormal, specularPower, pdf);
          const v3 brdf = mk3(fresnel * EvaluateModifiedPhong(hitNormal, nextSampleDir, V, specularStrength, specularPower));
          transmission = transmission * (brdf / fmaxf(0.01f, pdf));
          transmission = transmission / specRayProbability;
          newDir = nextSampleDir;
        } else {
          const v3 brdf = (1.0f - fresnel) * GetLambertianBRDF(color, hitNormal, V);
          const float pdf = GetLambertianPDF(hitNormal, V);
          transmission = transmission * (brdf / pdf);
          transmission = transmission / (1.0f - specRayProbability);
          newDir = GetCosineWeightedHemisphereDirection(u, hitNormal);
        }
        }
        const v3 add = alive ? transmission * emission : mk3(0.0f);   // :108 — after the throughput update (Q3)
        if (FUSED) {
          st_L[pid] = make_float4(0.0f + add.x, 0.0f + add.y, 0.0f + add.z, 0.0f);   // first value of L
        } else if (!(add.x == 0.0f && add.y == 0.0f && add.z == 0.0f)) {   // L + 0 == L: skip the RMW (a NaN term still lands, Q7)
          float4 Ls = st_L[pid];
          Ls.x += add.x; Ls.y += add.y; Ls.z += add.z;
          st_L[pid] = Ls;
        }
        // Russian roulette (extension, PT_OPT_RUSSIAN_ROULETTE): the unbiased form of the sketch at PathTracing.hlsl:110-124,
        // same place in the loop, same minBounces = 3: survive with probability q = clamp(lum(T), 0.05, 1), T /= q.
        if (alive && russian_roulette && bounce > 3) {
          const float q = clampf(GetLuminance(transmission), 0.05f, 1.0f);
          if (GetRand01(rng) >= q) alive = false;
          else transmission = transmission / q;
        }
        if (!last_bounce && alive) {
          st_stream(st_T + pid, make_float4(transmission.x, transmission.y, transmission.z, __uint_as_float(rng.w)));
          newOrigin = hitPos;   // :126
          want_ray = true;
        }
      }
    }
    // compaction (every thread of the block takes part)
    uint32_t slot, sslot;
    block_append2(s_app, parity, &counters[CTR_AT(CTR_QCOUNT0 + bounce + 1)], want_ray, &counters[CTR_AT(CTR_SKY)], want_sky, slot, sslot);
    if (want_ray) {
      PT_CHECK_SLOT(slot, capacity, "k_shade ray queue");
      st_stream(ray_o_out + slot, make_float4(newOrigin.x, newOrigin.y, newOrigin.z, 0.001f));  // TMin 0.001 (:127)
      st_stream(ray_d_out + slot, make_float4(newDir.x, newDir.y, newDir.z, 10000.0f));
      st_stream(qid_out + slot, pid);
    }
    if (want_sky) {
      PT_CHECK_SLOT(sslot, capacity, "k_shade sky queue");
      st_stream(sky_o + sslot, make_float4(origin.x, origin.y, origin.z, __uint_as_float(pid)));
      st_stream(sky_d + sslot, make_float4(dir.x, dir.y, dir.z, FUSED ? 1.0f : 0.0f));   // w != 0: T = 1 and L = 0, not in memory
    }
  }
  // one atomic per wave for the statistics
  for (int off = 32; off > 0; off >>= 1) {
    shaded_local += (uint32_t)__shfl_down((int)shaded_local, off);
    miss_local += (uint32_t)__shfl_down((int)miss_local, off);
  }
  if (lane == 0u && shaded_local) atomicAdd(&counters[CTR_AT(CTR_SHADED)], shaded_local);
  if (lane == 0u && miss_local) atomicAdd(&counters[CTR_AT(CTR_MISS)], miss_local);
}

// ------------------------------------------------------------------------------------------------ sky
// Sky-view LUT build (PT_OPT_SKY_MODE = 1; sky_device.h): one thread per texel, rebuilt at the head of every pt_render.
__global__ void __launch_bounds__(256) k_skyview_lut(PtRtConsts c, const uint2* __restrict__ tlut, float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= PT_SKYVIEW_W * PT_SKYVIEW_H) return;
  const v3 L = SkyViewLutTexel(c, tlut, i % PT_SKYVIEW_W, i / PT_SKYVIEW_W);
  out[i] = make_float4(L.x, L.y, L.z, 1.0f);
}

template <bool VIEW_LUT>
__global__ void __launch_bounds__(256) k_sky(PtRtConsts c, const uint2* __restrict__ lut, const float4* __restrict__ skyview,
                                             const float4* __restrict__ sky_o,
                                             const float4* __restrict__ sky_d, const float4* __restrict__ st_T,
                                             float4* __restrict__ st_L, const uint32_t* __restrict__ counters) {
  const uint32_t n = counters[CTR_AT(CTR_SKY)];
  SkyViewFrame frame;
  if (VIEW_LUT) frame = MakeSkyViewFrame(c);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 o = sky_o[i], d = sky_d[i];
    const uint32_t pid = __float_as_uint(o.w);
    v3 sky;
    if (VIEW_LUT) sky = SampleSkyLuminanceViewLut(mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z), c, skyview, frame);
    else sky = SampleSkyLuminance(mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z), c, lut);
    float4 T, L;
    if (d.w != 0.0f) {   // a primary that missed under PT_OPT_FUSED_PRIMARY: its state is still the initial constants
      T = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
      L = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    } else {
      T = st_T[pid];
      L = st_L[pid];
    }
    const v3 add = mk3(T.x, T.y, T.z) * sky;
    L.x += add.x; L.y += add.y; L.z += add.z;
    st_L[pid] = L;
  }
}

// ------------------------------------------------------------------------------------------------ AO pipeline
// Ao.hlsl RayGen :39-103 as a wavefront: generate -> extend (closest) -> k_ao_primary -> k_ao_rays -> extend (any hit)
// -> k_ao_count -> k_sky -> k_resolve.
// k_ao_primary: primaries that hit are appended to the hit list (qid), misses go to the sky queue (:96).
__global__ void __launch_bounds__(PT_SHADE_THREADS) k_ao_primary(PtRtConsts c, const float4* __restrict__ ray_o,
                                                                 const float4* __restrict__ ray_d, const float4* __restrict__ hits,
                                                                 float4* __restrict__ st_L, uint32_t* __restrict__ qid_out,
                                                                 float4* __restrict__ sky_o, float4* __restrict__ sky_d,
                                                                 uint32_t* __restrict__ counters, uint32_t capacity) {
  const uint32_t n = counters[CTR_AT(CTR_QCOUNT0)];
  __shared__ BlockAppendLds s_app;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t shaded_local = 0u, miss_local = 0u;
  int parity = 0;
  for (uint32_t j0 = blockIdx.x * blockDim.x; j0 < n; j0 += gridDim.x * blockDim.x, parity ^= 1) {
    const uint32_t j = j0 + threadIdx.x;
    bool want_hit = false, want_sky = false;
    if (j < n) {
      if (__float_as_uint(hits[j].w) == 0xFFFFFFFFu) {
        ++miss_local;
        if (c.mySampleSky) want_sky = true;
        else st_L[j] = make_float4(c.mySkyFallbackEmission[0], c.mySkyFallbackEmission[1], c.mySkyFallbackEmission[2], 0.0f);
      } else {
        ++shaded_local;
        want_hit = true;
      }
    }
    uint32_t slot, sslot;
    block_append2(s_app, parity, &counters[CTR_AT(CTR_AO_HITS)], want_hit, &counters[CTR_AT(CTR_SKY)], want_sky, slot, sslot);
    if (want_hit) { PT_CHECK_SLOT(slot, capacity, "k_ao_primary hit list"); qid_out[slot] = j; }
    if (want_sky) {
      PT_CHECK_SLOT(sslot, capacity, "k_ao_primary sky queue");
      const float4 o = ray_o[j], d = ray_d[j];
      sky_o[sslot] = make_float4(o.x, o.y, o.z, __uint_as_float(j));
      sky_d[sslot] = make_float4(d.x, d.y, d.z, 0.0f);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    shaded_local += (uint32_t)__shfl_down((int)shaded_local, off);
    miss_local += (uint32_t)__shfl_down((int)miss_local, off);
  }
  if (lane == 0u && shaded_local) atomicAdd(&counters[CTR_AT(CTR_SHADED)], shaded_local);
  if (lane == 0u && miss_local) atomicAdd(&counters[CTR_AT(CTR_MISS)], miss_local);
}

// k_ao_rays: one thread per AO ray (16 per primary hit). ClosestHitPrimary (:13-26): interpolated normal as loaded
// (no flip, not renormalised), hitPos = O + D*t. The RNG is counter based (Random.hlsl:96-99), so ray i takes draws
// number 3+2i and 4+2i of the pixel's stream directly (draws 1,2 were the jitter).
#define PT_AO_RAYS 16u
__global__ void __launch_bounds__(256) k_ao_rays(PtRtConsts c, DeviceScene sc, const float4* __restrict__ ray_o,
                                                 const float4* __restrict__ ray_d, const float4* __restrict__ hits,
                                                 const uint32_t* __restrict__ qid, float4* __restrict__ ao_o,
                                                 float4* __restrict__ ao_d, uint32_t* __restrict__ counters, uint32_t w,
                                                 uint32_t h, uint32_t first_sample) {
  const uint32_t n = counters[CTR_AT(CTR_AO_HITS)] * PT_AO_RAYS;
  if (blockIdx.x == 0 && threadIdx.x == 0) counters[CTR_AT(CTR_QCOUNT0 + 1)] = n;   // the AO-ray queue length
  const uint32_t wh = w * h;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const uint32_t pid = qid[t / PT_AO_RAYS], i = t % PT_AO_RAYS;
    const float4 ro = ray_o[pid], rd = ray_d[pid], hr = hits[pid];
    const float4* ps = sc.prim_shade + (size_t)__float_as_uint(hr.w) * 4;
    const float4 n0 = ps[0], n1 = ps[1], n2 = ps[2];
    const float bu = hr.y, bv = hr.z;
    const float baryZ = 1.0f - (bu + bv);
    const v3 hitNormal = bu * mk3(n0.x, n0.y, n0.z) + bv * mk3(n1.x, n1.y, n1.z) + baryZ * mk3(n2.x, n2.y, n2.z);
    const v3 hitPos = mk3(ro.x, ro.y, ro.z) + mk3(rd.x, rd.y, rd.z) * hr.x;
    Rng rng;
    uint32_t pix, si;
    si = 0u; pix = pid;   // the AO pipeline renders one sample per batch: path id = pixel
    FastDiv(w, wh).divmod(pix, rng.y, rng.x);
    rng.z = first_sample + si; rng.w = 2u + 2u * i;
    v2 rand11;
    rand11.x = GetRand01(rng) * 2.0f - 1.0f;
    rand11.y = GetRand01(rng) * 2.0f - 1.0f;
    const v3 dir = GetHemisphereDirection(rand11, hitNormal);
    ao_o[t] = make_float4(hitPos.x, hitPos.y, hitPos.z, 0.01f);        // TMin 0.01 (:79)
    ao_d[t] = make_float4(dir.x, dir.y, dir.z, c.myAoDistance);        // TMax myAoDistance (:81)
  }
}

// k_ao_count: one thread per AO ray; the 16 rays of a primary hit sit in a quarter wave. ao = 1 - hits/16 (:91-92).
__global__ void __launch_bounds__(256) k_ao_count(const float4* __restrict__ ao_hits, const uint32_t* __restrict__ qid,
                                                  float4* __restrict__ st_L, const uint32_t* __restrict__ counters) {
  const uint32_t n = counters[CTR_AT(CTR_QCOUNT0 + 1)];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, total_waves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t base = wave_global * 64u; base < n; base += total_waves * 64u) {   // wave-uniform trip count
    const uint32_t t = base + lane;
    const bool occluded = t < n && __float_as_uint(ao_hits[t].w) != 0xFFFFFFFFu;
    const unsigned long long m = __ballot(occluded);
    if ((lane % PT_AO_RAYS) == 0u && t < n) {
      const uint32_t numAoHits = (uint32_t)__popcll((m >> lane) & 0xFFFFull);
      const float ao = 1.0f - (float)numAoHits / (float)PT_AO_RAYS;
      st_L[qid[t / PT_AO_RAYS]] = make_float4(ao, ao, ao, 0.0f);
    }
  }
}

// hit records -> one byte per ray (1 = something was hit): the output format of occlusion queries
__global__ void __launch_bounds__(256) k_hit_flags(const float4* __restrict__ hits, uint64_t n, uint8_t* __restrict__ flags) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = __float_as_uint(hits[i].w) != 0xFFFFFFFFu ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ resolve
// PathTracing.hlsl:130-137 per sample, in sample-index order; the target keeps the SUM (pt_api.h header).
__global__ void __launch_bounds__(256) k_resolve(const float4* __restrict__ st_L, float4* __restrict__ accum, uint32_t wh,
                                                 uint32_t samples, uint32_t* __restrict__ counters,
                                                 unsigned long long* __restrict__ stats, uint32_t num_bounces) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    unsigned long long rays = 0ull;
    for (uint32_t b = 0; b < num_bounces; ++b) {
      const unsigned long long q = counters[CTR_AT(CTR_QCOUNT0 + b)];
      rays += q;
      stats[ST_RAYS_BOUNCE0 + b] += q;
    }
    stats[ST_RAYS] += rays;
    stats[ST_SHADED] += counters[CTR_AT(CTR_SHADED)];
    stats[ST_SKY] += counters[CTR_AT(CTR_MISS)];
    stats[ST_SAMPLES] += (unsigned long long)wh * samples;
  }
  // Path ids are pixel-major: a pixel's samples are contiguous (samples x 16 bytes), a thread's own pixel is not what its
  // neighbours read. With samples a multiple of 8 the block streams 8 samples (one 128-byte line) of each of its 256
  // pixels through LDS per round: 8 consecutive threads read one line, the transposed copy gives every thread its own
  // pixel's 8 values, added in sample order as before. Other sample counts (small batches) take the plain loop.
  __shared__ float4 s_t[256 * 9];   // [pixel][8 samples + 1 pad]: a thread's 8 reads fall into different banks from its neighbours'
  const bool in_range = i < wh;
  float4 a = in_range ? accum[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if ((samples & 7u) == 0u) {   // block-uniform
    const uint32_t pix0 = blockIdx.x * blockDim.x;
    for (uint32_t s0 = 0; s0 < samples; s0 += 8u) {
#pragma unroll
      for (uint32_t it = 0; it < 8u; ++it) {
        const uint32_t lp = it * 32u + (threadIdx.x >> 3), j = threadIdx.x & 7u;   // local pixel, sample within the round
        if (pix0 + lp < wh) s_t[lp * 9u + j] = ld_stream(st_L + (size_t)(pix0 + lp) * samples + s0 + j);
      }
      __syncthreads();
      if (in_range) {
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) {
          float4 L = s_t[threadIdx.x * 9u + j];
          if (isnan(L.x) || isnan(L.y) || isnan(L.z) || isinf(L.x) || isinf(L.y) || isinf(L.z)) { L.x = 0.0f; L.y = 0.0f; L.z = 0.0f; }
          a.x += L.x; a.y += L.y; a.z += L.z;
        }
      }
      __syncthreads();
    }
  } else if (in_range) {
    for (uint32_t s = 0; s < samples; ++s) {
      float4 L = st_L[(size_t)i * samples + s];   // path ids are pixel-major
      if (isnan(L.x) || isnan(L.y) || isnan(L.z) || isinf(L.x) || isinf(L.y) || isinf(L.z)) { L.x = 0.0f; L.y = 0.0f; L.z = 0.0f; }
      a.x += L.x; a.y += L.y; a.z += L.z;
    }
  }
  if (in_range) accum[i] = a;
}

__global__ void k_set_count(uint32_t* counter, uint32_t value) { *counter = value; }

// ------------------------------------------------------------------------------------------------ debug eval
__global__ void __launch_bounds__(256) k_debug_eval(int fn, PtRtConsts c, const uint2* __restrict__ lut, const float4* __restrict__ skyview,
                                                    const float* __restrict__ in, uint64_t n, float* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  switch (fn) {
    case PT_FN_RAND01: {
      const float* r = in + 4 * i;
      Rng s = InitRNG((uint32_t)r[0], (uint32_t)r[1], (uint32_t)r[2]);
      const uint32_t nd = (uint32_t)r[3];
      for (uint32_t k = 0; k < 8; ++k) out[8 * i + k] = k < nd ? GetRand01(s) : 0.0f;
    } break;
    case PT_FN_FRESNEL: {
      const float* r = in + 6 * i;
      out[i] = GetFresnelSchlick(ld3(r), ld3(r + 3));
    } break;
    case PT_FN_SPEC_PROB: {
      const float* r = in + 4 * i;
      out[i] = EstimateSpecularRayProbability(0.9f, ld3(r), r[3]);
    } break;
    case PT_FN_SAMPLE_PHONG: {
      const float* r = in + 6 * i;
      float pdf;
      v2 u; u.x = r[0]; u.y = r[1];
      const v3 d = SampleModifiedPhong(u, ld3(r + 2), r[5], pdf);
      out[4 * i] = d.x; out[4 * i + 1] = d.y; out[4 * i + 2] = d.z; out[4 * i + 3] = pdf;
    } break;
    case PT_FN_EVAL_PHONG: {
      const float* r = in + 10 * i;
      out[i] = EvaluateModifiedPhong(ld3(r), ld3(r + 3), ld3(r + 6), 0.9f, r[9]);
    } break;
    case PT_FN_COS_HEMISPHERE: {
      const float* r = in + 5 * i;
      v2 u; u.x = r[0]; u.y = r[1];
      const v3 d = GetCosineWeightedHemisphereDirection(u, ld3(r + 2));
      out[3 * i] = d.x; out[3 * i + 1] = d.y; out[3 * i + 2] = d.z;
    } break;
    case PT_FN_LAMBERT: {
      const float* r = in + 9 * i;
      const v3 b = GetLambertianBRDF(ld3(r), ld3(r + 3), ld3(r + 6));
      out[4 * i] = b.x; out[4 * i + 1] = b.y; out[4 * i + 2] = b.z; out[4 * i + 3] = GetLambertianPDF(ld3(r + 3), ld3(r + 6));
    } break;
    case PT_FN_SKY: {
      const float* r = in + 6 * i;
      const v3 L = SampleSkyLuminance(ld3(r), ld3(r + 3), c, lut);
      out[3 * i] = L.x; out[3 * i + 1] = L.y; out[3 * i + 2] = L.z;
    } break;
    case PT_FN_SKY_VIEW_LUT: {
      const float* r = in + 6 * i;
      const v3 L = SampleSkyLuminanceViewLut(ld3(r), ld3(r + 3), c, skyview, MakeSkyViewFrame(c));
      out[3 * i] = L.x; out[3 * i + 1] = L.y; out[3 * i + 2] = L.z;
    } break;
    case PT_FN_GGX_SAMPLE: {
      const float* r = in + 14 * i;
      Rng rng = InitRNG((uint32_t)r[11], (uint32_t)r[12], (uint32_t)r[13]);
      v3 T = mk3(1.0f), d = mk3(0.0f);
      const bool alive = SampleGgxMaterial(ld3(r), r[3], r[4], ld3(r + 5), ld3(r + 8), rng, T, d);
      float* o = out + 8 * i;
      o[0] = alive ? 1.0f : 0.0f; o[1] = T.x; o[2] = T.y; o[3] = T.z; o[4] = d.x; o[5] = d.y; o[6] = d.z; o[7] = (float)rng.w;
    } break;
    case PT_FN_ALBEDO_PATTERN: {
      const float* r = in + 4 * i;
      v2 uv; uv.x = r[2]; uv.y = r[3];
      out[i] = AlbedoPattern((uint32_t)r[0], r[1], uv);
    } break;
    case PT_FN_PRIMARY_RAY: {
      const float* r = in + 4 * i;
      v2 px; px.x = r[0]; px.y = r[1];
      v3 o, d;
      GetPrimaryRay(px, (uint32_t)r[2], (uint32_t)r[3], ld3(c.myNearPlaneCorner), ld3(c.myXAxis), ld3(c.myYAxis), ld3(c.myCameraPos), o, d);
      out[6 * i] = o.x; out[6 * i + 1] = o.y; out[6 * i + 2] = o.z; out[6 * i + 3] = d.x; out[6 * i + 4] = d.y; out[6 * i + 5] = d.z;
    } break;
    default: break;
  }
}

// ------------------------------------------------------------------------------------------------ launchers
static inline int grid_for(uint64_t n, int cap) {
  uint64_t g = (n + 255) / 256;
  if (g < 1) g = 1;
  if (cap > 0 && g > (uint64_t)cap) g = (uint64_t)cap;
  return (int)g;
}

void launch_transmittance_lut(const PtAtmosphere& atm, uint2* lut, hipStream_t s) {
  hipLaunchKernelGGL(k_transmittance_lut, dim3(PT_TLUT_W * PT_TLUT_H / 256), dim3(256), 0, s, atm, lut);
}

void launch_generate(const PtRtConsts& c, const WavefrontBuffers& wb, uint32_t w, uint32_t h, uint32_t first_sample,
                     uint32_t samples_in_batch, hipStream_t s) {
  const uint32_t n = w * h * samples_in_batch;
  hipLaunchKernelGGL(k_generate, dim3(grid_for(n, 0)), dim3(256), 0, s, c, wb.ray_o[0], wb.ray_d[0], wb.st_T, wb.st_L,
                     wb.counters, w, h, first_sample, n);
}

void launch_primary_setup(const PtRtConsts& c, const WavefrontBuffers& wb, uint32_t w, uint32_t h, uint32_t first_sample,
                          uint32_t samples_in_batch, hipStream_t s) {
  hipLaunchKernelGGL(k_primary_setup, dim3(1), dim3(64), 0, s, c, wb.gen, wb.counters, w, h, first_sample, w * h * samples_in_batch);
}

size_t extend_lds_bytes(uint32_t lds_entries_total, uint32_t top_nodes, bool shared) {
  return (size_t)lds_entries_total * 64u * sizeof(uint2) * 4u + (size_t)top_nodes * PT_NODE_STRIDE_U4 * 16u +
         (shared ? 4u * (1024u + PT_SHARED_LIST * 4u) : 0u);   // per wave: 64 hit keys, 64 (u, v) pairs, the record list of a shared triangle phase
}

int extend_occupancy_blocks(uint32_t lds_entries_total, uint32_t top_nodes, bool defer, bool paired, bool no_private_segment) {
  int nb = 0;
  const size_t lds = extend_lds_bytes(lds_entries_total, top_nodes, false);
  if (no_private_segment) {
    const hipError_t e = paired ? (defer ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_extend_ns<false, true, true, false>, 256, lds)
                                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_extend_ns<false, false, true, false>, 256, lds))
                                : (defer ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_extend_ns<false, true, false, false>, 256, lds)
                                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_extend_ns<false, false, false, false>, 256, lds));
    return (e != hipSuccess || nb < 1) ? 1 : nb;
  }
  // per instantiation: the pair test holds one more vertex in registers (5 waves per SIMD instead of 6)
  const hipError_t e = paired ? (defer ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_extend<false, true, true, false>, 256, lds)
                                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_extend<false, false, true, false>, 256, lds))
                              : (defer ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_extend<false, true, false, false>, 256, lds)
                                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_extend<false, false, false, false>, 256, lds));
  if (e != hipSuccess || nb < 1) nb = 1;
  return nb;
}

void launch_extend(const DeviceScene& sc, const float4* ray_o, const float4* ray_d, uint32_t stride, float4* hits, const uint32_t* count,
                   uint32_t* work_cursor, unsigned long long* stats, bool count_traversal, bool any_hit, const LaunchCfg& cfg, hipStream_t s,
                   const float4* primary_gen) {
  // the shared triangle phase exists in the deferred schedule's ordinary instantiations only; its LDS (2 KB per wave) comes
  // out of the parked-group queue and the top-node cache, so that as many blocks stay resident as without it
  const bool shared = cfg.shared_tri && cfg.defer_entries > 0 && !cfg.no_private_segment;
  const uint32_t dq = shared ? cfg.sh_defer_entries : cfg.defer_entries;
  const uint32_t postpone_pct = shared ? cfg.sh_postpone_pct : cfg.postpone_pct, continue_pct = shared ? cfg.sh_continue_pct : cfg.continue_pct;
  uint32_t lds_entries = sc.stack_entries < cfg.lds_stack_entries ? sc.stack_entries : cfg.lds_stack_entries;
  if (lds_entries + dq > PT_EXTEND_MAX_LDS_ENTRIES)   // 2 KB of LDS per entry and block: stay within one CU's LDS
    lds_entries = PT_EXTEND_MAX_LDS_ENTRIES - dq;     // the rest of the stack goes to the spill region
  const uint32_t spill_entries = sc.stack_entries - lds_entries;
  const uint32_t want_top = shared ? cfg.sh_top_nodes : cfg.top_nodes;
  uint32_t top_nodes = sc.num_nodes < want_top ? sc.num_nodes : want_top;
  while (top_nodes > 0 && extend_lds_bytes(lds_entries + dq, top_nodes, shared) > 160u * 1024u) top_nodes /= 2;   // one CU's LDS
  const size_t lds = extend_lds_bytes(lds_entries + dq, top_nodes, shared);
  if (lds > 64 * 1024) {   // deep BVHs: opt in to more than the default dynamic-LDS limit. The attribute belongs to the CURRENT
                           // device (one process may drive several, pt_reduce), so it is set on every such launch: no cached flag.
#define PT_EXTEND_ATTR(C, D, P, S) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_extend<C, D, P, S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
    PT_EXTEND_ATTR(false, false, false, false); PT_EXTEND_ATTR(true, false, false, false); PT_EXTEND_ATTR(false, true, false, false); PT_EXTEND_ATTR(true, true, false, false);
    PT_EXTEND_ATTR(false, false, true, false); PT_EXTEND_ATTR(true, false, true, false); PT_EXTEND_ATTR(false, true, true, false); PT_EXTEND_ATTR(true, true, true, false);
    PT_EXTEND_ATTR(false, true, false, true); PT_EXTEND_ATTR(true, true, false, true); PT_EXTEND_ATTR(false, true, true, true); PT_EXTEND_ATTR(true, true, true, true);
#undef PT_EXTEND_ATTR
  }
#define PT_LAUNCH_EXTEND(C, D, P, S)                                                                                           \
  hipLaunchKernelGGL((k_extend<C, D, P, S>), dim3(cfg.extend_blocks), dim3(256), lds, s, sc.nodes, sc.tris, ray_o, ray_d, hits, count, \
                     work_cursor, stats, lds_entries, spill_entries, sc.spill, stride, cfg.fetch_min, postpone_pct, dq, any_hit ? 1u : 0u, sc.scene_mag, top_nodes, continue_pct, primary_gen)
#define PT_LAUNCH_EXTEND2(C, D, S) do { if (sc.paired) PT_LAUNCH_EXTEND(C, D, true, S); else PT_LAUNCH_EXTEND(C, D, false, S); } while (0)
  if (cfg.no_private_segment) {   // pt_trace_rays_device_noalloc: the instantiations that use no scratch memory (never the counting ones)
#define PT_LAUNCH_EXTEND_NS(D, P)                                                                                            \
  hipLaunchKernelGGL((k_extend_ns<false, D, P, false>), dim3(cfg.extend_blocks), dim3(256), lds, s, sc.nodes, sc.tris, ray_o, ray_d, hits, count, \
                     work_cursor, stats, lds_entries, spill_entries, sc.spill, stride, cfg.fetch_min, postpone_pct, dq, any_hit ? 1u : 0u, sc.scene_mag, top_nodes, continue_pct, primary_gen)
    if (lds > 64 * 1024) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_extend_ns<false, true, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_extend_ns<false, true, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_extend_ns<false, false, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_extend_ns<false, false, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (dq > 0) { if (sc.paired) PT_LAUNCH_EXTEND_NS(true, true); else PT_LAUNCH_EXTEND_NS(true, false); }
    else { if (sc.paired) PT_LAUNCH_EXTEND_NS(false, true); else PT_LAUNCH_EXTEND_NS(false, false); }
#undef PT_LAUNCH_EXTEND_NS
  } else if (shared) {
    if (count_traversal) PT_LAUNCH_EXTEND2(true, true, true); else PT_LAUNCH_EXTEND2(false, true, true);
  } else if (dq > 0) {
    if (count_traversal) PT_LAUNCH_EXTEND2(true, true, false); else PT_LAUNCH_EXTEND2(false, true, false);
  } else {
    if (count_traversal) PT_LAUNCH_EXTEND2(true, false, false); else PT_LAUNCH_EXTEND2(false, false, false);
  }
#undef PT_LAUNCH_EXTEND2
#undef PT_LAUNCH_EXTEND
}

// pt_trace_rays_device_noalloc may only launch k_extend_ns if that kernel has NO private segment (register spills, stack
// arrays): with one, the HIP runtime allocates scratch memory at launch, behind the caller's back. Nothing but the compiler
// decides that, so it is asked here: bytes of private segment of the instantiation that would be launched.
int extend_ns_private_bytes(bool defer, bool paired, size_t* bytes_out) {
  hipFuncAttributes a;
  const void* f = paired ? (defer ? reinterpret_cast<const void*>(&k_extend_ns<false, true, true, false>) : reinterpret_cast<const void*>(&k_extend_ns<false, false, true, false>))
                         : (defer ? reinterpret_cast<const void*>(&k_extend_ns<false, true, false, false>) : reinterpret_cast<const void*>(&k_extend_ns<false, false, false, false>));
  const hipError_t e = hipFuncGetAttributes(&a, f);
  if (e != hipSuccess) return (int)e;
  *bytes_out = a.localSizeBytes;
  return 0;
}

int extend_packet_occupancy_blocks(uint32_t stack_entries, bool paired) {
  int nb = 0;
  const size_t lds = (size_t)stack_entries * 4u * sizeof(uint2);
  const hipError_t e = paired ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_extend_packet<true>, 256, lds)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_extend_packet<false>, 256, lds);
  if (e != hipSuccess || nb < 1) nb = 1;
  return nb;
}

// Coherent queues only (the fused primaries of pt_render): one BVH walk per wave (k_extend_packet).
void launch_extend_packet(const DeviceScene& sc, const float4* ray_o, const float4* ray_d, float4* hits, const uint32_t* count,
                          uint32_t* work_cursor, int blocks, hipStream_t s, const float4* primary_gen) {
  const size_t lds = (size_t)sc.stack_entries * 4u * sizeof(uint2);   // one 8-byte entry per BVH level and wave
  if (sc.paired)
    hipLaunchKernelGGL((k_extend_packet<true>), dim3(blocks), dim3(256), lds, s, sc.nodes, sc.tris, ray_o, ray_d, hits, count, work_cursor,
                       sc.stack_entries, sc.scene_mag, primary_gen);
  else
    hipLaunchKernelGGL((k_extend_packet<false>), dim3(blocks), dim3(256), lds, s, sc.nodes, sc.tris, ray_o, ray_d, hits, count, work_cursor,
                       sc.stack_entries, sc.scene_mag, primary_gen);
}

void launch_shade(const PtRtConsts& c, const DeviceScene& sc, const WavefrontBuffers& wb, int bounce, uint32_t w, uint32_t h,
                  uint32_t first_sample, uint64_t max_paths, bool russian_roulette, const LaunchCfg& cfg, hipStream_t s, bool fused_primary) {
  const int in = bounce & 1, out = in ^ 1;
  uint64_t g = (max_paths + PT_SHADE_THREADS - 1) / PT_SHADE_THREADS;
  const uint64_t cap = (uint64_t)cfg.num_cus * 8u;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
#define PT_LAUNCH_SHADE(E, F)                                                                                               \
  hipLaunchKernelGGL((k_shade<E, F>), dim3((unsigned)g), dim3(PT_SHADE_THREADS), 0, s, c, sc, wb.ray_o[in], wb.ray_d[in],   \
                     bounce == 0 ? (const uint32_t*)nullptr : wb.qid[in], wb.hits, wb.st_T, wb.st_L, wb.ray_o[out], wb.ray_d[out], \
                     wb.qid[out], wb.sky_o, wb.sky_d, wb.counters, bounce, w, h, first_sample, russian_roulette ? 1u : 0u, (uint32_t)(max_paths / ((uint64_t)w * h)), \
                     (uint32_t)(wb.capacity > 0xFFFFFFFFull ? 0xFFFFFFFFull : wb.capacity))
  if (fused_primary && bounce == 0) {
    if (sc.instance_ext) PT_LAUNCH_SHADE(true, true); else PT_LAUNCH_SHADE(false, true);
  } else {
    if (sc.instance_ext) PT_LAUNCH_SHADE(true, false); else PT_LAUNCH_SHADE(false, false);
  }
#undef PT_LAUNCH_SHADE
}

void launch_ao_primary(const PtRtConsts& c, const WavefrontBuffers& wb, uint64_t n_paths, const LaunchCfg& cfg, hipStream_t s) {
  uint64_t g = (n_paths + PT_SHADE_THREADS - 1) / PT_SHADE_THREADS;
  const uint64_t cap = (uint64_t)cfg.num_cus * 8u;
  g = g > cap ? cap : (g < 1 ? 1 : g);
  hipLaunchKernelGGL(k_ao_primary, dim3((unsigned)g), dim3(PT_SHADE_THREADS), 0, s, c, wb.ray_o[0], wb.ray_d[0], wb.hits, wb.st_L, wb.qid[0],
                     wb.sky_o, wb.sky_d, wb.counters, (uint32_t)(wb.capacity > 0xFFFFFFFFull ? 0xFFFFFFFFull : wb.capacity));
}

void launch_ao_rays(const PtRtConsts& c, const DeviceScene& sc, const WavefrontBuffers& wb, float4* ao_o, float4* ao_d, uint32_t w, uint32_t h,
                    uint32_t first_sample, uint64_t n_paths, const LaunchCfg& cfg, hipStream_t s) {
  hipLaunchKernelGGL(k_ao_rays, dim3(grid_for(n_paths * PT_AO_RAYS, cfg.stream_blocks)), dim3(256), 0, s, c, sc, wb.ray_o[0], wb.ray_d[0], wb.hits,
                     wb.qid[0], ao_o, ao_d, wb.counters, w, h, first_sample);
}

void launch_ao_count(const WavefrontBuffers& wb, const float4* ao_hits, uint64_t n_paths, const LaunchCfg& cfg, hipStream_t s) {
  hipLaunchKernelGGL(k_ao_count, dim3(grid_for(n_paths * PT_AO_RAYS, cfg.stream_blocks)), dim3(256), 0, s, ao_hits, wb.qid[0], wb.st_L, wb.counters);
}

void launch_hit_flags(const float4* hits, uint64_t n, uint8_t* flags, hipStream_t s) {
  hipLaunchKernelGGL(k_hit_flags, dim3(grid_for(n, 0)), dim3(256), 0, s, hits, n, flags);
}

void launch_skyview_lut(const PtRtConsts& c, const uint2* tlut, float4* out, hipStream_t s) {
  hipLaunchKernelGGL(k_skyview_lut, dim3((PT_SKYVIEW_W * PT_SKYVIEW_H + 255) / 256), dim3(256), 0, s, c, tlut, out);
}

void launch_sky(const PtRtConsts& c, const uint2* lut, const float4* skyview, const WavefrontBuffers& wb, uint64_t max_paths, const LaunchCfg& cfg, hipStream_t s) {
  if (skyview) {
    hipLaunchKernelGGL(k_sky<true>, dim3(grid_for(max_paths, cfg.stream_blocks)), dim3(256), 0, s, c, lut, skyview, wb.sky_o, wb.sky_d, wb.st_T, wb.st_L,
                       wb.counters);
    return;
  }
  hipLaunchKernelGGL(k_sky<false>, dim3(grid_for(max_paths, cfg.stream_blocks)), dim3(256), 0, s, c, lut, skyview, wb.sky_o, wb.sky_d, wb.st_T, wb.st_L,
                     wb.counters);
}

void launch_resolve(const WavefrontBuffers& wb, float4* accum, uint32_t w, uint32_t h, uint32_t samples_in_batch, uint32_t num_bounces,
                    hipStream_t s) {
  const uint32_t wh = w * h;
  hipLaunchKernelGGL(k_resolve, dim3(grid_for(wh, 0)), dim3(256), 0, s, wb.st_L, accum, wh, samples_in_batch, wb.counters, wb.stats,
                     num_bounces);
}

__global__ void __launch_bounds__(256) k_add(float4* __restrict__ dst, const float4* __restrict__ src, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 a = dst[i];
  const float4 b = src[i];
  a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  dst[i] = a;
}

void launch_add(float4* dst, const float4* src, uint64_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_add, dim3(grid_for(n, 0)), dim3(256), 0, s, dst, src, n);
}

// ------------------------------------------------------------------------------------------------ static scratch
// Device memory of the entry point that allocates nothing (pt_trace_rays_device_noalloc): the two counter lines of the
// extend kernel and the spill region of its per-lane stacks live in this file-scope array instead of hipMalloc'ed
// buffers. Every GPU the library runs on holds its own copy at its own address until the process ends. In host code
// the array's NAME is only a handle for hipGetSymbolAddress: the address that kernels, memsets and copies use is the
// one this function returns (looked up once per device by pt_create, with that device current).
static __device__ __align__(256) unsigned char scratch[PT_SCRATCH_BYTES];

hipError_t scratch_address(void** out) {
  *out = nullptr;
  void* p = nullptr;
  const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
  if (e == hipSuccess) *out = p;
  return e;
}

void launch_set_count(uint32_t* counter, uint32_t value, hipStream_t s) {
  hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, s, counter, value);
}

void launch_debug_eval(int fn, const PtRtConsts& c, const uint2* lut, const float4* skyview, const float* in, uint64_t n, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_debug_eval, dim3(grid_for(n, 0)), dim3(256), 0, s, fn, c, lut, skyview, in, n, out);
}

}  // namespace pt
