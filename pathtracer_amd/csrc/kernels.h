// kernels.h — host-callable launchers of the HIP kernels (kernels.hip). Internal to libpathtracer_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pt_api.h"

namespace pt {

// device-resident scene (all pointers are device memory)
struct DeviceScene {
  const uint4* nodes = nullptr;        // 5 x uint4 per 80-byte BVH8 node
  const uint4* tris = nullptr;         // leaf records: 48-byte triangles, or (paired) 64-byte records of one triangle or an edge-sharing pair
  bool paired = false;                 // record format of `tris` (csrc/bvh8.h)
  const uint4* prim_data = nullptr;    // per global triangle id: i0,i1,i2 (global vertex ids), instance id (read with the material extension only)
  const float4* prim_shade = nullptr;  // 4 x float4 per global triangle id: {n0, instance id bits}, {n1, colour r}, {n2, colour g}, {emission, colour b} (colour = decoded unorm, alpha unused):
                                       // everything ClosestHit loads, in ONE 64-byte record (one dependent fetch instead of three)
  const float2* vertex_uv = nullptr;
  const float4* instance_ext = nullptr;  // material extension per instance (roughness, metallic, pattern scale, pattern | ggx << 8); null = none
  const uint32_t* albedo_tex = nullptr;  // material extension, albedo pattern 3: RGBA8 image (pt_scene_set_albedo_texture), tex_w x tex_h texels
  uint32_t tex_w = 0, tex_h = 0;
  float scene_mag = 0.0f;              // largest |coordinate| of the active triangles (overflow guard of the box test)
  uint32_t num_nodes = 0;              // BVH8 nodes in `nodes`
  uint32_t stack_entries = 0;          // stack entries per lane the traversal may need (2 per BVH level + 2)
  uint2* spill = nullptr;              // global spill region of the per-lane stacks: [wave][entry][lane]
};

// device counter block (u32), zeroed at the start of every wavefront batch. Every counter sits on its own
// 256-byte line (CTR_AT): returning atomics on one line serialise, so hot counters must not share one.
#define CTR_AT(i) ((i) * 64)
enum {
  CTR_QCOUNT0 = 0,      // [0..63]   rays queued for bounce b
  CTR_WORK0 = 64,       // [64..127] work-distribution cursor of the extend kernel for bounce b
  CTR_SKY = 128,        // sky-queue length
  CTR_SHADED = 129,     // ClosestHit-equivalents this batch
  CTR_MISS = 130,       // rays that hit nothing (SampleSkyLuminance calls, sky on or off)
  CTR_AO_HITS = 131,    // AO pipeline: primary rays that hit (each spawns 16 AO rays)
  CTR_SLOTS = 132,
  CTR_WORDS = 132 * 64
};
// persistent device statistics (u64)
enum { ST_SAMPLES = 0, ST_RAYS, ST_SHADED, ST_SKY, ST_NODES, ST_TRIS, ST_NODE_WAVE_STEPS, ST_TRI_WAVE_STEPS, ST_RAYS_BOUNCE0 = 8, ST_WORDS = 8 + 64 };

struct WavefrontBuffers {
  float4* ray_o[2] = {nullptr, nullptr};  // origin.xyz, tmin
  float4* ray_d[2] = {nullptr, nullptr};  // dir.xyz, tmax
  uint32_t* qid[2] = {nullptr, nullptr};  // path id of queue slot
  float4* hits = nullptr;                 // t,u,v,prim bits
  float4* st_T = nullptr;                 // throughput.xyz, rng draw counter bits
  float4* st_L = nullptr;                 // luminance.xyz, -
  float4* sky_o = nullptr;                // ray origin.xyz, path id bits
  float4* sky_d = nullptr;                // ray dir.xyz, 1 = primary of a fused batch (state still the initial constants)
  float4* gen = nullptr;                  // 4 x float4: the primary-ray frame of the batch (k_primary_setup)
  uint32_t* counters = nullptr;           // CTR_WORDS
  unsigned long long* stats = nullptr;    // ST_WORDS
  uint64_t capacity = 0;                  // paths
};

#define PT_SKYVIEW_TEXELS (256 * 256)   // PT_SKYVIEW_W x PT_SKYVIEW_H of sky_device.h

#define PT_EXTEND_MAX_LDS_ENTRIES 64u   // stack + deferred-queue entries per lane kept in LDS (128 KB per block at most)

// Static device scratch of pt_trace_rays_device_noalloc (kernels.hip): [0,256) ray-count line, [256,512) work-cursor
// line, [512, PT_SCRATCH_BYTES) spill region of the per-lane traversal stacks.
#define PT_SCRATCH_BYTES (64u << 20)
#define PT_SCRATCH_HEADER 512u
hipError_t scratch_address(void** out);   // address of the CURRENT device's copy of the array

#ifndef PT_DEF_POSTPONE_PCT
#define PT_DEF_POSTPONE_PCT 40
#endif
#ifndef PT_DEF_CONTINUE_PCT
#define PT_DEF_CONTINUE_PCT 25
#endif
struct LaunchCfg {
  int num_cus = 256;
  int extend_blocks = 1024;   // persistent grid of the extend kernel
  int stream_blocks = 2048;   // grid cap of the grid-stride kernels
  uint32_t fetch_min = 16;    // extend: refill idle lanes once this many of the 64 are idle
  uint32_t postpone_pct = PT_DEF_POSTPONE_PCT; // extend: postpone triangle tests while fewer than this % of busy lanes have any
  uint32_t continue_pct = PT_DEF_CONTINUE_PCT; // extend (deferred schedule): a triangle phase goes on while at least this % of busy lanes have work
  int extend_blocks_per_cu = 0; // 0 = from the occupancy query
  uint32_t lds_stack_entries = 6; // per-lane stack entries kept in LDS; deeper entries spill to global memory
  uint32_t defer_entries = 4;     // >0: deferred-triangle schedule with this many parked groups per lane
  bool no_private_segment = false; // launch the extend instantiation that uses no scratch memory (pt_trace_rays_device_noalloc)
  uint32_t top_nodes = 73;        // nodes of the tree's top (breadth-first order: 1 + 8 + 64) kept in LDS by every extend block
  // shared triangle phase (k_extend<.., SHARED = true>): its own schedule, tuned with it (tools/experiments/run_shared_tri.sh)
  bool shared_tri = false;
  uint32_t sh_defer_entries = 3, sh_top_nodes = 8, sh_postpone_pct = 25, sh_continue_pct = 50;
};

void launch_transmittance_lut(const PtAtmosphere& atm, uint2* lut, hipStream_t s);
void launch_generate(const PtRtConsts& c, const WavefrontBuffers& wb, uint32_t w, uint32_t h, uint32_t first_sample,
                     uint32_t samples_in_batch, hipStream_t s);
void launch_primary_setup(const PtRtConsts& c, const WavefrontBuffers& wb, uint32_t w, uint32_t h, uint32_t first_sample,
                          uint32_t samples_in_batch, hipStream_t s);
void launch_extend(const DeviceScene& sc, const float4* ray_o, const float4* ray_d, uint32_t stride, float4* hits, const uint32_t* count,
                   uint32_t* work_cursor, unsigned long long* stats, bool count_traversal, bool any_hit, const LaunchCfg& cfg, hipStream_t s,
                   const float4* primary_gen = nullptr);
// AO pipeline (Ao.hlsl): classify primaries, spawn 16 rays per primary hit into ao_o/ao_d, count occluded rays into st_L
void launch_ao_primary(const PtRtConsts& c, const WavefrontBuffers& wb, uint64_t n_paths, const LaunchCfg& cfg, hipStream_t s);
void launch_ao_rays(const PtRtConsts& c, const DeviceScene& sc, const WavefrontBuffers& wb, float4* ao_o, float4* ao_d, uint32_t w, uint32_t h,
                    uint32_t first_sample, uint64_t n_paths, const LaunchCfg& cfg, hipStream_t s);
void launch_ao_count(const WavefrontBuffers& wb, const float4* ao_hits, uint64_t n_paths, const LaunchCfg& cfg, hipStream_t s);
void launch_hit_flags(const float4* hits, uint64_t n, uint8_t* flags, hipStream_t s);
void launch_shade(const PtRtConsts& c, const DeviceScene& sc, const WavefrontBuffers& wb, int bounce, uint32_t w, uint32_t h,
                  uint32_t first_sample, uint64_t max_paths, bool russian_roulette, const LaunchCfg& cfg, hipStream_t s, bool fused_primary = false);
void launch_skyview_lut(const PtRtConsts& c, const uint2* tlut, float4* out, hipStream_t s);   // out: PT_SKYVIEW_TEXELS float4
void launch_sky(const PtRtConsts& c, const uint2* lut, const float4* skyview, const WavefrontBuffers& wb, uint64_t max_paths, const LaunchCfg& cfg, hipStream_t s);
void launch_resolve(const WavefrontBuffers& wb, float4* accum, uint32_t w, uint32_t h, uint32_t samples_in_batch, uint32_t num_bounces,
                    hipStream_t s);
void launch_add(float4* dst, const float4* src, uint64_t n, hipStream_t s);   // dst[i] += src[i]
void launch_set_count(uint32_t* counter, uint32_t value, hipStream_t s);
void launch_debug_eval(int fn, const PtRtConsts& c, const uint2* lut, const float4* skyview, const float* in, uint64_t n, float* out, hipStream_t s);
// k_extend_packet: one BVH walk per wave for coherent queues (fused primaries); blocks = CUs x extend_packet_occupancy_blocks
void launch_extend_packet(const DeviceScene& sc, const float4* ray_o, const float4* ray_d, float4* hits, const uint32_t* count,
                          uint32_t* work_cursor, int blocks, hipStream_t s, const float4* primary_gen);
int extend_packet_occupancy_blocks(uint32_t stack_entries, bool paired);
size_t extend_lds_bytes(uint32_t lds_entries_total, uint32_t top_nodes, bool shared = false);
int extend_ns_private_bytes(bool defer, bool paired, size_t* bytes_out);   // 0 = ok; private-segment bytes of the k_extend_ns instantiation
int extend_occupancy_blocks(uint32_t lds_entries_total, uint32_t top_nodes, bool defer, bool paired, bool no_private_segment = false);

}  // namespace pt
