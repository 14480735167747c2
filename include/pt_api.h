/* pt_api.h — C-ABI boundary of the MI355X path-tracing hot path (libpathtracer_amd.so).
 *
 * The reference (domme/PathTracer) has no plugin/FFI API; the narrowest seam of its DXR path is
 *   PathTracer::InitRtScene            PathTracer/PathTracer.cpp:171-339   (scene -> GPU buffers + BLAS/TLAS)
 *   Sky::ComputeTranmittanceLut        PathTracer/Sky.cpp:127-148          (256x64 RGBA16F LUT)
 *   PathTracer::RenderRT               PathTracer/PathTracer.cpp:659-787   (clear, RtConsts, DispatchRays)
 *   PathTracer::RestartAccumulation    PathTracer/PathTracer.cpp:912-915
 * Each entry point below names the piece of that seam it replaces. All functions return an int
 * status (PT_OK == 0); nothing but plain pointers and sizes crosses the boundary; the caller owns
 * every host array and the library copies what it keeps. One pt_ctx per device, driven by one host
 * thread at a time (the reference is single render thread, PathTracer.cpp:576,591).
 *
 * Documented deviations from the reference:
 *   Q1  sample position = uPixel + jitter, jitter in [0,1)^2, written to texel uPixel (race-free).
 *   Q9  the RNG stream is keyed on (px, py, sampleIndex) with sampleIndex the 0-based accumulation
 *       index (myFrameRandomSeed of the i-th sample of a pt_render call = first_sample + i).
 *   The accumulation target holds the float SUM of samples (alpha 0) and the sample count; the mean
 *   (the reference's running mean, PathTracing.hlsl:134-137) is formed on read.
 */
#ifndef PT_API_H
#define PT_API_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (reference: ASSERT / IsValid(), PathTracer.cpp:102,105,108,273) ---- */
enum {
  PT_OK = 0,
  PT_ERR_INVALID_ARG = 1,   /* null pointer, zero size, index out of range */
  PT_ERR_NO_DEVICE = 2,     /* no HIP device / device index out of range */
  PT_ERR_HIP = 3,           /* a HIP runtime call failed; see pt_last_error */
  PT_ERR_NO_SCENE = 4,      /* render/trace before pt_scene_create */
  PT_ERR_NO_TARGET = 5,     /* render/read before pt_accum_reset / pt_accum_bind */
  PT_ERR_SIZE_MISMATCH = 6, /* w,h differ from the accumulation target */
  PT_ERR_NO_SKY = 7,        /* mySampleSky != 0 before pt_sky_set */
  PT_ERR_OUT_OF_MEMORY = 8,
  PT_ERR_UNSUPPORTED = 9
};

typedef struct pt_ctx pt_ctx;

/* ---- AtmosphereParameters: PathTracer/Sky.h:27-62 == resources/shaders/sky/Common.hlsl:7-43 (112 B) ---- */
typedef struct PtAtmosphere {
  float RayleighScattering[3];
  float RayleighDensityExpScale;
  float AbsorptionExtinction[3];
  float BottomRadius;
  float GroundAlbedo[3];
  float TopRadius;
  float MieScattering[3];
  float MieDensityExpScale;
  float MieExtinction[3];
  float MiePhaseG;
  float MieAbsorption[3];
  float AbsorptionDensity0LayerWidth;
  float AbsorptionDensity0ConstantTerm;
  float AbsorptionDensity0LinearTerm;
  float AbsorptionDensity1ConstantTerm;
  float AbsorptionDensity1LinearTerm;
} PtAtmosphere;

/* ---- SkyConstants: PathTracer.cpp:689-707 == raytracing/Common.hlsl:10-22 (160 B) ---- */
typedef struct PtSkyConstants {
  PtAtmosphere myAtmosphere;
  float mySunDirection[3];
  uint32_t myTransmissionLutTexIdx; /* ignored: the LUT lives in the ctx */
  float mySunIlluminance[3];
  float _unused;
  float myRayMarchMinMaxSPP[2];     /* reference always passes (4,14), PathTracer.cpp:707 */
  float _unused2[2];
} PtSkyConstants;

/* ---- RtConsts: PathTracer.cpp:709-740 == raytracing/Common.hlsl:24-55 (288 B, bit-for-bit) ---- */
typedef struct PtRtConsts {
  float myNearPlaneCorner[3];
  float myAoDistance;
  float myXAxis[3];
  uint32_t myOutTexIndex;               /* ignored (bindless index) */
  float myYAxis[3];
  uint32_t myAsIndex;                   /* ignored */
  float myCameraPos[3];
  uint32_t myInstanceDataBufferIndex;   /* ignored */
  uint32_t myMaterialDataBufferIndex;   /* ignored */
  uint32_t mySampleBufferIndex;         /* ignored (Halton buffer is never read by the shaders) */
  uint32_t myFrameRandomSeed;           /* ignored by pt_render (Q9: seed = sample index); used by nothing else */
  uint32_t myNumAccumulationFrames;     /* ignored: the ctx tracks the sample count */
  uint32_t myLinearClampSamplerIndex;   /* ignored */
  uint32_t myMaxRecursionDepth;         /* inclusive loop bound: depth d => up to d+1 rays (PathTracing.hlsl:68) */
  uint32_t myLightInstanceId;
  uint32_t myNumHaltonSamples;          /* ignored */
  float myLightEmission[3];
  uint32_t mySampleSky;
  float mySkyFallbackEmission[3];
  float myPhongSpecularPower;
  PtSkyConstants mySkyConsts;
} PtRtConsts;

/* ---- scene description: the inputs of InitRtScene (PathTracer.cpp:171-339) ---- */
/* One interleaved vertex = {pos f3, normal f3, uv f2}, 32-byte stride (LoadScene, PathTracer.cpp:116-120). */
typedef struct PtVertex {
  float pos[3];
  float normal[3];
  float uv[2];
} PtVertex;

typedef struct PtMeshDesc {
  const PtVertex* vertices;   /* num_vertices entries */
  uint32_t num_vertices;
  const uint32_t* indices;    /* 3*num_triangles entries, R_32UI (PathTracer.cpp:204) */
  uint32_t num_triangles;
} PtMeshDesc;

/* SceneMeshInstance (PathTracer.cpp:285-305): instance id == array index (:298). */
typedef struct PtInstanceDesc {
  uint32_t myMeshIndex;
  uint32_t myMaterialIndex;
  float myTransform[16];      /* column-major 4x4 (glm::float4x4), applied to positions */
} PtInstanceDesc;

/* MaterialData (PathTracer.cpp:318-329): {float3 emission; uint rgba8 colour, R in the low byte}. */
typedef struct PtMaterial {
  float myEmission[3];
  uint32_t myColor;
} PtMaterial;

/* pt_scene_create flags */
enum {
  PT_SCENE_DEFAULT = 0,
  /* The reference never transforms normals by the instance matrix (PathTracing.hlsl:24-27).
     Set this to rotate them by the upper 3x3 of myTransform instead. */
  PT_SCENE_TRANSFORM_NORMALS = 1,
  /* Leaf records of the acceleration structure: by default the builder pairs edge-sharing triangles into 64-byte records
     once the scene is too large for the 256 MB Infinity Cache (a third less memory, +9.5 % on 10.5 M triangles) and keeps
     48-byte single-triangle records below that (pairs cost 4 % there). These flags force one or the other; hit records
     are identical either way. */
  PT_SCENE_PAIR_TRIANGLES = 2,
  PT_SCENE_NO_PAIRS = 4,
  /* Build quickly rather than well: no spatial splits and no optimisation pass over the tree (a third of the build time;
     traversal is a few per cent slower on regular scenes and several times slower on long thin or overlapping geometry).
     For interactive loading; hit records are identical either way. */
  PT_SCENE_FAST_BUILD = 8
};

/* closest-hit record returned by the trace entry points */
typedef struct PtHit {
  float t;            /* RayTCurrent; < 0 on miss */
  float u;            /* DXR barycentric .x  = weight of vertex 1 */
  float v;            /* DXR barycentric .y  = weight of vertex 2 */
  uint32_t prim;      /* global triangle id (instance-major order of pt_scene_create); 0xFFFFFFFF on miss */
} PtHit;

#define PT_MISS 0xFFFFFFFFu

/* a ray for pt_trace_rays: RayDesc {Origin, TMin, Direction, TMax} (PathTracing.hlsl:57-61), 32 B */
typedef struct PtRay {
  float origin[3];
  float tmin;
  float dir[3];
  float tmax;
} PtRay;

/* Versioned by size: the caller sets struct_size = sizeof(PtStats) of the header IT was built against before calling
 * pt_stats_get; the library fills at most that many bytes (and writes back how many it filled). Members are only ever
 * APPENDED from this layout on, so a host built against an older header keeps reading the fields it knows. */
typedef struct PtStats {
  uint32_t struct_size;    /* in: sizeof(PtStats) of the caller's header; out: bytes filled (<= in) */
  uint32_t reserved0;
  uint64_t samples;        /* pixel-samples completed since the last pt_stats_reset */
  uint64_t rays;           /* TraceRay-equivalents issued (primary included) */
  uint64_t shaded_hits;    /* ClosestHit-equivalents */
  uint64_t sky_misses;     /* SampleSkyLuminance calls */
  uint64_t nodes_visited;  /* wide-BVH nodes fetched  (only when counting is on) */
  uint64_t tris_tested;    /* triangles tested         (only when counting is on) */
  uint64_t extend_launches;/* launches of the traversal (extend) kernel */
  double extend_ms;        /* sum of their durations, HIP events on the ctx stream (only when timing is on) */
  double render_ms;        /* sum of whole pt_render durations, HIP events (only when timing is on) */
  uint64_t rays_bounce[16];      /* rays traced at bounce b (b >= 15 folded into [15]) */
  double extend_ms_bounce[16];   /* extend-kernel time at bounce b (only when timing is on) */
  uint64_t node_wave_steps;      /* wave-level node steps (counting on): nodes_visited / (64*this) = lane utilisation */
  uint64_t tri_wave_steps;       /* wave-level triangle steps (counting on) */
  uint64_t extend_launches_bounce[16]; /* traversal-kernel launches at bounce b */
  uint64_t packet_launches;      /* of those, launches of k_extend_packet (the coherent-primaries kernel, PT_OPT_PRIMARY_PACKET) */
  float packet_cal_ms[2];        /* PT_OPT_PRIMARY_PACKET = 2: what the one-off timing measured on <= 2^24 primaries: k_extend, k_extend_packet */
  uint64_t shared_launches;      /* traversal-kernel launches that ran the shared triangle phase (PT_OPT_SHARED_TRIANGLE_PHASE) */
  float shared_cal_ms[2];        /* PT_OPT_SHARED_TRIANGLE_PHASE = 2: ms of the secondary bounces of the timed batches, two of each form summed: per-lane, shared */
  int32_t shared_choice;         /* its outcome: -1 = not timed yet (per-lane until then), 0 = per-lane, 1 = shared */
  uint32_t reserved1;
} PtStats;

typedef struct PtSceneInfo {
  uint32_t num_triangles;
  uint32_t num_vertices;
  uint32_t num_instances;
  uint32_t num_materials;
  uint32_t num_bvh_nodes;     /* 80-byte compressed 8-wide nodes */
  uint32_t bvh_max_depth;
  uint64_t bvh_bytes;         /* nodes + 48-byte triangle records */
  float bounds_min[3];
  float bounds_max[3];
  double build_ms;
} PtSceneInfo;

/* options for pt_set_option (the traversal kernel's tuning knobs are not here: include/pt_api_tuning.h) */
enum {
  PT_OPT_COUNT_TRAVERSAL = 1, /* 0/1: run the counting instantiation of the extend kernel (fills nodes_visited/tris_tested) */
  PT_OPT_TIME_KERNELS = 2,    /* 0/1: bracket kernels with HIP events (fills extend_ms/render_ms) */
  PT_OPT_SAMPLES_PER_BATCH = 3, /* samples per pixel kept in flight per wavefront batch (0 = auto) */
  PT_OPT_SKY_MODE = 4,        /* 0 = ray-march on every miss (reference, SampleSky.hlsl:7-47); 1 = 256x256 direction LUT built per pt_render
                                 for the camera position (extension after sky/compute_skyView_lut.hlsl; image relMSE vs mode 0 < 1e-4;
                                 cameras above 50 km of altitude are marched as in mode 0) */
  PT_OPT_RUSSIAN_ROULETTE = 10,    /* 0/1 (default 0 = reference behaviour): unbiased Russian roulette after bounce 3 — survive with
                                      q = clamp(luminance(throughput), 0.05, 1), throughput /= q (extension; the reference only has
                                      a commented-out, biased sketch at PathTracing.hlsl:110-124) */
  PT_OPT_PIPELINE = 11,            /* 0/1 (default 0): split every batch of >= 2 samples into two halves that run on two streams (the
                                      second one internal): one half's shade kernels and kernel drains run under the other half's
                                      extend, +1..2 % throughput, results bit-identical; off by default because concurrent kernels
                                      blur the per-kernel durations that profiles and the bench's roofline are built on */
  PT_OPT_FUSED_PRIMARY = 14,       /* 0/1 (default 1): pt_render computes primary rays inside the traversal and shading kernels
                                      (RayGen's ray set-up, PathTracing.hlsl:40-61) instead of writing them to the bounce-0 queue
                                      and reading them back: 160 B of memory traffic per path less, results bit-identical */
  PT_OPT_FRAME_LANES = 16,         /* 1..8 (default 6): small pt_render calls (one batch of < 2^20 paths: the reference's interactive loop, one
                                      640x360 sample per RenderRT call, PathTracer.cpp:573-595) are queued round-robin on this many internal
                                      streams with their own wavefront state, so consecutive calls overlap on the GPU; their resolves stay
                                      on the ctx stream in call order, so the target is bit-identical to 1 (= every call serial on the ctx
                                      stream). Each lane in use holds its own queues and ~150 MB of traversal-stack spill region. */
  PT_OPT_PRIMARY_PACKET = 15,      /* which kernel traces the fused primaries of pt_render (coherent 8x8-pixel tiles): 0 = the general per-lane
                                      traversal, 1 = one BVH walk per wave with wave-uniform node fetches (k_extend_packet), 2 (default) = the
                                      faster of the two, timed once per scene and frame size on <= 2^24 primaries. That FIRST pt_render call
                                      per scene and frame size blocks the host (one stream synchronisation) and cannot be stream-captured;
                                      callers that need every call asynchronous set 0 or 1. Same hit records either way. */
  PT_OPT_SHARED_TRIANGLE_PHASE = 17 /* how the traversal kernel runs its triangle tests: 0 = every lane tests the records of its own ray, 1 = the
                                      wave pools the pending records of its 64 rays and tests them 64 at a time, whoever owns them (a hit goes
                                      back to its ray through an LDS minimum on (t, triangle id), so the closest hit and its tie-break are the
                                      same), 2 (default) = the faster of the two for this scene: of a scene's first pt_render batches of >= 2^21 paths the
                                      first is left alone and the next four run the per-lane form, the shared one, and both again, each between
                                      two HIP events; a later call reads the times and keeps the shared form if its two batches together were
                                      faster by 1.5 % (no launch is added, nothing blocks; until then the per-lane form runs). Pays where leaves are big and uneven (long thin triangles, spatial
                                      splits: +8 % on the SAH-hostile street), costs ~1 % on regular scenes. Same hit records either way. */
};

/* ---- lifetime (replaces the RenderCore device + handle ownership, PathTracer.cpp:45-77,417-437) ---- */
int pt_create(int device, pt_ctx** out_ctx);
int pt_destroy(pt_ctx* ctx);
/* Human-readable text for the last non-OK status on this ctx (never NULL). ctx may be NULL for create errors. */
const char* pt_last_error(const pt_ctx* ctx);
/* Run subsequent work on this hipStream_t (passed as void*); NULL = the ctx's own stream. */
int pt_set_stream(pt_ctx* ctx, void* hip_stream);
int pt_synchronize(pt_ctx* ctx);
int pt_set_option(pt_ctx* ctx, int option, int64_t value);

/* Material extension for BASELINE configs C4/C5 — NOT in the reference, whose materials are untextured Lambert +
 * modified Phong (PathTracing.hlsl:83-106, MaterialData at PathTracer.cpp:318-329). One record per material of the last
 * pt_scene_create, same order. Materials with PT_MAT_GGX set are shaded with a GGX specular lobe (visible-normal
 * sampling, separable Smith) + Lambert diffuse in the metallic/roughness parametrisation; albedo_pattern modulates the
 * base colour by a procedural function of the interpolated uv (either material model). The definition is
 * pathtracer_amd/csrc/ggx_device.h; oracle/ggx.h restates it. ext == NULL or num_materials == 0 removes the extension. */
#define PT_MAT_GGX 1u
typedef struct PtMaterialExt {
  float roughness;          /* GGX alpha = max(roughness^2, 1e-3) */
  float metallic;           /* F0 = lerp(0.04, base, metallic); diffuse = base * (1 - metallic) */
  uint32_t albedo_pattern;  /* 0 = none, 1 = checker (1 / 0.35), 2 = smooth value noise in [0.35, 1), 3 = the scene's albedo image
                               (pt_scene_set_albedo_texture; factor 1 while none is set) */
  float pattern_scale;      /* uv multiplier */
  uint32_t flags;           /* PT_MAT_GGX */
  uint32_t reserved[3];
} PtMaterialExt;
int pt_scene_set_material_ext(pt_ctx* ctx, const PtMaterialExt* ext, uint32_t num_materials);
/* Albedo image of the material extension (BASELINE C4 "textured GGX materials"; NOT in the reference, whose one DDS file
 * is never sampled, PathTracer.cpp:435-436): width x height RGBA8 texels, row-major, R in the low byte. Materials with
 * albedo_pattern == 3 multiply their base colour by the image sampled at uv * pattern_scale (repeat addressing, bilinear,
 * texel centres at half-integer coordinates; definition: pathtracer_amd/csrc/ggx_device.h SampleAlbedoTexture). The library
 * copies the texels; rgba8 == NULL removes the image. Belongs to the current scene (pt_scene_create drops it). */
int pt_scene_set_albedo_texture(pt_ctx* ctx, const uint8_t* rgba8, uint32_t width, uint32_t height);

/* ---- scene: replaces PathTracer::InitRtScene incl. the driver BLAS/TLAS build (PathTracer.cpp:171-339) ----
 * Flattens meshes x instances into world-space SoA triangle / vertex{normal,uv} / material buffers,
 * builds a binned-SAH BVH collapsed to an 8-wide compressed BVH on the host and uploads everything.
 * Instance flags are the reference's: cull disabled, forced opaque, mask 0xFF (PathTracer.cpp:302-304). */
int pt_scene_create(pt_ctx* ctx,
                    const PtMeshDesc* meshes, uint32_t num_meshes,
                    const PtInstanceDesc* instances, uint32_t num_instances,
                    const PtMaterial* materials, uint32_t num_materials,
                    uint32_t flags);
int pt_scene_info(const pt_ctx* ctx, PtSceneInfo* out);

/* ---- sky: replaces Sky::ComputeTranmittanceLut + compute_transmittance_lut.hlsl (Sky.cpp:127-148) ----
 * Recomputes the 256x64 RGBA16F transmittance LUT on the device for these atmosphere parameters. */
int pt_sky_set(pt_ctx* ctx, const PtAtmosphere* atmosphere);
/* Copy the LUT back: 256*64*4 binary16 values, row-major, RGBA. */
int pt_sky_lut_read(pt_ctx* ctx, uint16_t* rgba16f);

/* ---- accumulation target: replaces UpdateOutputTexture + RestartAccumulation + clear_texture.hlsl
 *      (PathTracer.cpp:834-878, 912-915, 668-678) ---- */
/* (Re)allocate a w x h RGBA32F target if the size changed, zero it, sample count = 0. */
int pt_accum_reset(pt_ctx* ctx, uint32_t width, uint32_t height);
/* Use caller-owned DEVICE memory (w*h*4 floats) as the target instead; zeroes it; count = 0.
   Lets the host reduce it across GPUs (RCCL) without a copy. Pass NULL to return to an internal target. */
int pt_accum_bind(pt_ctx* ctx, float* device_rgba, uint32_t width, uint32_t height);
/* Device address + sample count of the current target (sum of samples, alpha 0). */
int pt_accum_device_ptr(pt_ctx* ctx, float** device_rgba, uint64_t* num_samples);
/* Overwrite the sample count (after an external reduce has summed other ranks' targets into this one). */
int pt_accum_set_count(pt_ctx* ctx, uint64_t num_samples);
/* Mean image = sum / count -> host, w*h*4 floats (what the reference's texture holds, PathTracing.hlsl:134-137). */
int pt_accum_read(pt_ctx* ctx, float* host_rgba_mean);
/* Raw sum -> host. */
int pt_accum_read_sum(pt_ctx* ctx, float* host_rgba_sum, uint64_t* num_samples);

/* Multi-GPU in ONE process (SURVEY §8b "pt_reduce"; the reference is single-GPU, no counterpart): adds the sum targets
 * of ctxs[1..n-1] onto ctxs[0] in array order (peer copies into a staging buffer on ctxs[0]'s device + one add kernel
 * each), adds their sample counts, and waits for completion. All targets must have the same size; the source
 * contexts keep their targets unchanged. One-process-per-GPU jobs use torch.distributed instead
 * (pathtracer_amd/distributed.py). */
int pt_reduce(pt_ctx** ctxs, uint32_t n);

/* ---- the hot path: replaces PathTracer::RenderRT + RayGen/ClosestHit/(miss) (PathTracer.cpp:659-787,
 *      PathTracing.hlsl:16-138). Renders samples first_sample .. first_sample+num_samples-1 of every pixel
 *      (one reference "frame" each) and adds them to the accumulation target. Asynchronous on the ctx stream. ---- */
int pt_render(pt_ctx* ctx, const PtRtConsts* consts, uint32_t width, uint32_t height,
              uint32_t first_sample, uint32_t num_samples);

/* The reference's alternative "AO" pipeline (resources/shaders/raytracing/Ao.hlsl:39-103; selected by
 * PathTracer::myRenderAo, PathTracer.cpp:341-370,680-683): one primary ray per pixel-sample; on a hit, 16 occlusion rays
 * (TMin 0.01, TMax consts->myAoDistance) around the interpolated normal and L = 1 - occluded/16; on a miss the sky
 * (SampleSkyLuminance) or the fallback emission. Same target, sample indexing and asynchrony as pt_render. */
int pt_render_ao(pt_ctx* ctx, const PtRtConsts* consts, uint32_t width, uint32_t height, uint32_t first_sample, uint32_t num_samples);

/* Occlusion query (host arrays): occluded[i] = 1 when anything is hit in (tmin, tmax), else 0. What Ao.hlsl's
 * ClosestHitAo (:33-37) computes per AO ray; traversal stops at the first accepted intersection. */
int pt_occluded(pt_ctx* ctx, const PtRay* rays, uint64_t num_rays, uint8_t* occluded);

/* ---- parity / measurement helpers (no counterpart in the reference: TraceRay is a driver black box) ---- */
/* Primary-ray closest hit per pixel for one sample index (RayGen lines 43-71 only). hits: w*h entries, host. */
int pt_primary_hits(pt_ctx* ctx, const PtRtConsts* consts, uint32_t width, uint32_t height,
                    uint32_t sample, PtHit* hits);
/* Closest hit of arbitrary rays (TraceRay semantics: tmin < t < tmax, two-sided, opaque). Host arrays. */
int pt_trace_rays(pt_ctx* ctx, const PtRay* rays, uint64_t num_rays, PtHit* hits);
/* Same on DEVICE arrays already resident (used by the bench): no copies, asynchronous. */
int pt_trace_rays_device(pt_ctx* ctx, const PtRay* device_rays, uint64_t num_rays, PtHit* device_hits);
/* pt_trace_rays_device for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc). NO counterpart in the reference. It calls no hipMalloc / hipFree / allocating
 * routine: the kernel's counters and the spill region of its traversal stacks live in a static 64 MiB device array
 * that every GPU the library runs on holds until the process ends. Limits: num_rays < 2^31; the scene's BVH must be
 * shallow enough that the spill of one block per CU fits the array (PT_ERR_UNSUPPORTED otherwise; every BASELINE scene
 * fits) — deeper scenes only shrink the persistent grid. Same results as pt_trace_rays_device, bit for bit. All calls on
 * one GPU share that GPU's array (whatever the ctx or stream): calls that overlap on one GPU race. */
int pt_trace_rays_device_noalloc(pt_ctx* ctx, const PtRay* device_rays, uint64_t num_rays, PtHit* device_hits);
/* Evaluate one shading/sky device function on n rows of host inputs (see PT_FN_* for row layouts). */
int pt_debug_eval(pt_ctx* ctx, int fn, const PtRtConsts* consts, const float* in, uint64_t n, float* out);
/* Copy out the ray queue that the LAST wavefront batch built for `bounce` (valid until the next render call; the
   queue of bounce b is overwritten by bounce b+2 of the same batch). host_rays may be NULL to query the count. */
int pt_debug_read_queue(pt_ctx* ctx, uint32_t bounce, PtRay* host_rays, uint64_t capacity, uint64_t* count_out);
/* Copy the acceleration structure as it lives on the device back to the host: 80-byte nodes and leaf records of
   *record_bytes each (48 = one triangle, 64 = one triangle or an edge-sharing pair; layout: pathtracer_amd/csrc/bvh8.h),
   plus the scene magnitude the box test's overflow guard uses. With both arrays NULL only the counts are returned
   (capacities are in records). Lets tests walk the very hierarchy the kernel walks (tests/walker/). */
int pt_debug_read_bvh(pt_ctx* ctx, void* nodes, uint64_t node_capacity, void* tris, uint64_t tri_capacity,
                      uint64_t* num_nodes, uint64_t* num_tris, float* scene_mag, uint32_t* record_bytes);
int pt_stats_get(pt_ctx* ctx, PtStats* out);   /* out->struct_size must be set (see PtStats); synchronises the stream first */
int pt_stats_reset(pt_ctx* ctx);

/* function ids for pt_debug_eval: floats per input row -> floats per output row */
enum {
  PT_FN_RAND01 = 0,            /* in: px,py,seed,ndraws (as floats holding integers) -> out: 8 draws (first min(ndraws,8)) */
  PT_FN_FRESNEL = 1,           /* in: N3,V3 (6)            -> out: 1   GetFresnelSchlick      Common.hlsl:196-201 */
  PT_FN_SPEC_PROB = 2,         /* in: color3,fresnel (4)   -> out: 1   EstimateSpecularRayProbability brdfSampling.hlsl:41-49 */
  PT_FN_SAMPLE_PHONG = 3,      /* in: u2,N3,power (6)      -> out: 4   dir3,pdf SampleModifiedPhong brdfSampling.hlsl:24-39 */
  PT_FN_EVAL_PHONG = 4,        /* in: N3,L3,V3,power (10)  -> out: 1   EvaluateModifiedPhong(.,.,.,0.9,power) brdfSampling.hlsl:16-22 */
  PT_FN_COS_HEMISPHERE = 5,    /* in: u2,N3 (5)            -> out: 3   GetCosineWeightedHemisphereDirection Common.hlsl:137-141 */
  PT_FN_LAMBERT = 6,           /* in: color3,N3,L3 (9)     -> out: 4   brdf3,pdf brdfSampling.hlsl:6-14 */
  PT_FN_SKY = 7,               /* in: viewPos3,viewDir3 (6)-> out: 3   SampleSkyLuminance SampleSky.hlsl:7-47 (uses consts + ctx LUT) */
  PT_FN_PRIMARY_RAY = 8,       /* in: pixel2,res2 (4)      -> out: 6   origin3,dir3 GetPrimaryRay Common.hlsl:183-189 */
  PT_FN_SKY_VIEW_LUT = 9,      /* in: viewPos3,viewDir3 (6)-> out: 3   sky-mode-1 stand-in for PT_FN_SKY: bilinear fetch from the
                                  direction LUT built for consts->myCameraPos (extension; viewPos only feeds the march's domain tests) */
  PT_FN_GGX_SAMPLE = 10,       /* in: base3,roughness,metallic,N3,V3,px,py,seed (14) -> out: alive,weight3,dir3,draws (8)
                                  SampleGgxMaterial of the material extension (ggx_device.h), RNG = InitRNG(px,py,seed) */
  PT_FN_ALBEDO_PATTERN = 11,   /* in: pattern,scale,u,v (4) -> out: 1   AlbedoPattern of the material extension */
  PT_FN_COUNT = 12
};

#ifdef __cplusplus
}
#endif
#endif /* PT_API_H */
