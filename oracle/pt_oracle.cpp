// ORACLE — TEST INFRASTRUCTURE ONLY.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
// (pathtracer_amd/, libpathtracer_amd.so) never includes, links or calls anything under oracle/.
//
// What it is: a scalar C++ restatement of the reference's DXR path-tracing pipeline
//   resources/shaders/raytracing/PathTracing.hlsl  (RayGen :40-138, ClosestHit :16-38)
//   + Common.hlsl / Random.hlsl / brdfSampling.hlsl / SampleSky.hlsl / sky/Common.hlsl /
//     sky/compute_transmittance_lut.hlsl, host side PathTracer/PathTracer.cpp:171-339,659-787, Sky.cpp:32-56,127-148
// built with  g++ -O2 -ffp-contract=off -fopenmp  (oracle/Makefile).
//
// PINNING STATUS: the reference ships NO tests, golden vectors or fixtures for this path and cannot
// be built or run here (Win32 + D3D12/DXR; FANCY engine submodule absent; no dxc). The only exact pins are the
// integer KATs of pcg4d/GetRand01, derived by hand from Random.hlsl:45-62,84-99 and committed under tests/golden/.
// Everything floating-point is therefore "PARITY UNPINNED" against a real DXR run: it is a careful transcription with
// the adopted definitions listed in hlsl_math.h / sky.h / geometry.h, not a verified reproduction.
//
// Deviations from the literal HLSL, shared with the product:
//   Q1  sample position = uPixel + jitter (jitter in [0,1)^2), written to texel uPixel. jitter_mode=1 switches to the
//       literal reference behaviour (pixel += jitter-0.5, write to trunc(pixel)), race-free here because it runs serially.
//   Q9  frame seed = 0-based sample (accumulation) index.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "geometry.h"
#include "shading.h"
#include "sky.h"
#include "ggx.h"

using namespace orc;

static_assert(sizeof(PtRtConsts) == 288, "RtConsts must be 288 bytes (PathTracer.cpp:709-740)");
static_assert(sizeof(PtSkyConstants) == 160, "SkyConstants must be 160 bytes");
static_assert(offsetof(PtRtConsts, mySkyConsts) == 128, "SkyConstants offset");
static_assert(offsetof(PtRtConsts, myLightEmission) == 96, "lightEmission offset");

namespace {

static inline float3 ld3(const float* p) { return f3(p[0], p[1], p[2]); }

struct HitInfo {  // PathTracing.hlsl:7-14
  float3 myHitPos, myHitNormal, myColor, myEmission;
  bool myHasHit;
  // material extension (not in the reference): GGX flag and parameters of the hit's material
  bool myGgx = false;
  float myRoughness = 0.0f, myMetallic = 0.0f;
};

// ClosestHit — PathTracing.hlsl:16-38 (+ loaders Common.hlsl:75-127)
static inline void ClosestHit(const Scene& S, const PtRtConsts& c, const PtHit& h, float3 rayOrigin, float3 rayDir,
                              HitInfo& payload) {
  uint32_t instanceId = S.tri_instance[h.prim];                       // InstanceID()
  uint32_t materialIndex = S.instance_material[instanceId];           // LoadInstanceData
  const PtMaterial& enc = S.materials[materialIndex];                 // LoadMaterialData
  float3 matColor = Decode_Unorm_RGB(enc.myColor);
  float3 matEmission = ld3(enc.myEmission);
  // LoadInterpolatedVertexData — Common.hlsl:112-127. attrib.bary = DXR barycentrics (x: weight of vertex 1,
  // y: weight of vertex 2) fed into the reference's formula unchanged. myUv is interpolated by the
  // reference too but never read by the path-tracing pipeline.
  const uint32_t* idx = &S.tri_indices[3 * (size_t)h.prim];
  float3 n0 = S.vertex_normal[idx[0]], n1 = S.vertex_normal[idx[1]], n2 = S.vertex_normal[idx[2]];
  float baryZ = 1.0f - (h.u + h.v);
  float3 normal = h.u * n0 + h.v * n1 + baryZ * n2;
  payload.myHasHit = true;
  payload.myHitNormal = normal;
  payload.myHitPos = rayOrigin + rayDir * h.t;
  payload.myColor = matColor;
  payload.myEmission = matEmission;
  payload.myGgx = false;
  if (!S.materials_ext.empty()) {   // extension: procedural albedo from the interpolated uv, GGX parameters
    const PtMaterialExt& e = S.materials_ext[materialIndex];
    if (e.albedo_pattern != 0u) {
      float2 t0 = S.vertex_uv[idx[0]], t1 = S.vertex_uv[idx[1]], t2 = S.vertex_uv[idx[2]];
      float2 uv;   // LoadInterpolatedVertexData weights (Common.hlsl:121-126), as for the normal
      uv.x = h.u * t0.x + h.v * t1.x + baryZ * t2.x;
      uv.y = h.u * t0.y + h.v * t1.y + baryZ * t2.y;
      if (e.albedo_pattern == 3u)
        payload.myColor = payload.myColor * SampleAlbedoTexture(S.albedo_tex.empty() ? nullptr : S.albedo_tex.data(), S.tex_w, S.tex_h, e.pattern_scale, uv);
      else
        payload.myColor = payload.myColor * AlbedoPattern(e.albedo_pattern, e.pattern_scale, uv);
    }
    payload.myGgx = (e.flags & PT_MAT_GGX_BIT) != 0u;
    payload.myRoughness = e.roughness;
    payload.myMetallic = e.metallic;
  }
  if (dot(payload.myHitNormal, -rayDir) < 0.0f) payload.myHitNormal = -payload.myHitNormal;
  if (instanceId == c.myLightInstanceId) payload.myEmission = ld3(c.myLightEmission);
}

// Russian roulette (extension; BASELINE config C5). The reference only carries a commented-out sketch
// (PathTracing.hlsl:110-124) that divides by the termination probability and "results in fireflies"; what is
// implemented, here and in the HIP shade kernel, is the unbiased form at the same place in the loop and with the same
// minBounces = 3: survive with probability q = clamp(luminance(transmission), 0.05, 1), then transmission /= q.
static int g_russian_roulette = 0;

struct SampleStats { uint64_t rays = 0, hits = 0, sky = 0, nodes = 0, tris = 0; };

// Test probe (orc_bounce_rays): what one path carries INTO iteration `bounce` of the RayGen loop (PathTracing.hlsl:63-71):
// the TraceRay arguments and the payload-derived state so far. `reached` stays false for a path that ended earlier.
struct PathProbe {
  uint32_t bounce = 0;
  bool reached = false;
  float3 origin, dir, transmission, luminance;
  float tmin = 0.0f, tmax = 0.0f;
};

// RayGen — PathTracing.hlsl:40-138, one sample of one pixel. Returns the (scrubbed) luminance and the texel to write.
static inline float3 RayGenSample(const Scene& S, const PtRtConsts& c, const uint16_t* lut, uint32_t px, uint32_t py,
                                  uint32_t resX, uint32_t resY, uint32_t frameSeed, int jitter_mode, int brute,
                                  SampleStats& st, int& outX, int& outY, bool count_traversal, PathProbe* probe = nullptr) {
  uint4 rngState = InitRNG(px, py, frameSeed);
  float2 pixel = {(float)px, (float)py};
  float jx = GetRand01(rngState);
  float jy = GetRand01(rngState);
  if (jitter_mode == 1) {  // literal: pixel += lerp(-0.5.xx, 0.5.xx, jitter)  (:50)
    pixel.x += lerpf(-0.5f, 0.5f, jx);
    pixel.y += lerpf(-0.5f, 0.5f, jy);
  } else {                 // Q1 deviation
    pixel.x += jx;
    pixel.y += jy;
  }
  pixel.x = clampf(pixel.x, 0.0f, (float)resX);
  pixel.y = clampf(pixel.y, 0.0f, (float)resY);

  float3 origin, dir;
  GetPrimaryRay(pixel, resX, resY, ld3(c.myNearPlaneCorner), ld3(c.myXAxis), ld3(c.myYAxis), ld3(c.myCameraPos), origin, dir);
  float tmin = 0.0f;
  const float tmax = 10000.0f;

  float3 luminance = f3(0.0f);
  float3 transmission = f3(1.0f);
  HitInfo hitInfo;
  TraceCounters tc;
  for (uint32_t bounceIdx = 0u; bounceIdx <= c.myMaxRecursionDepth; ++bounceIdx) {
    hitInfo.myHasHit = false;
    if (probe && bounceIdx == probe->bounce) {
      probe->reached = true;
      probe->origin = origin; probe->dir = dir; probe->tmin = tmin; probe->tmax = tmax;
      probe->transmission = transmission; probe->luminance = luminance;
    }
    st.rays++;
    PtHit h = brute ? trace_brute(S, origin, tmin, dir, tmax) : trace_bvh(S, origin, tmin, dir, tmax, count_traversal ? &tc : nullptr);
    if (h.prim != PT_MISS) ClosestHit(S, c, h, origin, dir, hitInfo);
    if (!hitInfo.myHasHit) {
      st.sky++;
      luminance += transmission * SampleSkyLuminance(origin, dir, c, lut);
      break;
    }
    st.hits++;
    float specularStrength = 0.9f;
    float specularPower = c.myPhongSpecularPower;
    if (hitInfo.myGgx) {   // material extension (ggx.h); a direction below the surface ends the path with no contribution
      float3 nextDir;
      if (!SampleGgxMaterial(hitInfo.myColor, hitInfo.myRoughness, hitInfo.myMetallic, hitInfo.myHitNormal, -dir, rngState, transmission, nextDir))
        break;
      dir = nextDir;
    } else {
    float fresnel = GetFresnelSchlick(hitInfo.myHitNormal, -dir);
    float specRayProbability = EstimateSpecularRayProbability(specularStrength, hitInfo.myColor, fresnel);
    if (GetRand01(rngState) < specRayProbability) {
      float pdf;
      float u0 = GetRand01(rngState);  // float2(GetRand01, GetRand01): evaluated left to right (adopted)
      float u1 = GetRand01(rngState);
      float3 nextSampleDir = SampleModifiedPhong(float2{u0, u1}, hitInfo.myHitNormal, specularPower, pdf);
      float3 brdf = f3(fresnel * EvaluateModifiedPhong(hitInfo.myHitNormal, nextSampleDir, -dir, specularStrength, specularPower));
      transmission *= brdf / hmax(0.01f, pdf);
      transmission /= specRayProbability;
      dir = nextSampleDir;
    } else {
      float3 brdf = (1.0f - fresnel) * GetLambertianBRDF(hitInfo.myColor, hitInfo.myHitNormal, -dir);
      float pdf = GetLambertianPDF(hitInfo.myHitNormal, -dir);
      transmission *= brdf / pdf;
      transmission /= (1.0f - specRayProbability);
      float u0 = GetRand01(rngState);
      float u1 = GetRand01(rngState);
      dir = GetCosineWeightedHemisphereDirection(float2{u0, u1}, hitInfo.myHitNormal);
    }
    }
    luminance += transmission * hitInfo.myEmission;
    if (g_russian_roulette && bounceIdx > 3u) {
      const float q = clampf(GetLuminance(transmission), 0.05f, 1.0f);
      if (GetRand01(rngState) >= q) break;
      transmission /= q;
    }
    origin = hitInfo.myHitPos;
    tmin = 0.001f;
  }
  st.nodes += tc.nodes; st.tris += tc.tris;
  if (std::isnan(luminance.x) || std::isnan(luminance.y) || std::isnan(luminance.z) || std::isinf(luminance.x) ||
      std::isinf(luminance.y) || std::isinf(luminance.z))
    luminance = f3(0.0f);
  if (jitter_mode == 1) { outX = (int)pixel.x; outY = (int)pixel.y; }  // theRwTextures2D[..][pixel]: float index truncated
  else { outX = (int)px; outY = (int)py; }
  return luminance;
}

// Ao.hlsl RayGen — :39-103, one sample of one pixel (the alternative "AO" pipeline: 1 primary + 16 short rays).
// ClosestHitPrimary (:13-26) keeps the interpolated normal as it is (no flip towards the viewer, not renormalised);
// ClosestHitAo (:33-37) only reports that something was hit, so the AO rays need existence, not the closest hit.
// Same Q1 deviation for the jitter as RayGenSample. AO rays: TMin 0.01, TMax myAoDistance (:79-81).
static inline float3 RayGenAoSample(const Scene& S, const PtRtConsts& c, const uint16_t* lut, uint32_t px, uint32_t py,
                                    uint32_t resX, uint32_t resY, uint32_t frameSeed, int brute, SampleStats& st) {
  uint4 rngState = InitRNG(px, py, frameSeed);
  float2 pixel = {(float)px, (float)py};
  float jx = GetRand01(rngState);
  float jy = GetRand01(rngState);
  pixel.x += jx;
  pixel.y += jy;
  pixel.x = clampf(pixel.x, 0.0f, (float)resX);
  pixel.y = clampf(pixel.y, 0.0f, (float)resY);
  float3 origin, dir;
  GetPrimaryRay(pixel, resX, resY, ld3(c.myNearPlaneCorner), ld3(c.myXAxis), ld3(c.myYAxis), ld3(c.myCameraPos), origin, dir);
  st.rays++;
  PtHit h = brute ? trace_brute(S, origin, 0.0f, dir, 10000.0f) : trace_bvh(S, origin, 0.0f, dir, 10000.0f, nullptr);
  if (h.prim == PT_MISS) {
    st.sky++;
    return SampleSkyLuminance(origin, dir, c, lut);   // :96
  }
  st.hits++;
  const uint32_t* idx = &S.tri_indices[3 * (size_t)h.prim];
  float3 n0 = S.vertex_normal[idx[0]], n1 = S.vertex_normal[idx[1]], n2 = S.vertex_normal[idx[2]];
  float baryZ = 1.0f - (h.u + h.v);
  float3 hitNormal = h.u * n0 + h.v * n1 + baryZ * n2;
  float3 hitPos = origin + dir * h.t;
  const uint32_t numAoRays = 16;
  uint32_t numAoHits = 0;
  for (uint32_t i = 0u; i < numAoRays; ++i) {
    float r0 = GetRand01(rngState);
    float r1 = GetRand01(rngState);
    float2 rand11 = {r0 * 2.0f - 1.0f, r1 * 2.0f - 1.0f};
    float3 aoDir = GetHemisphereDirection(rand11, hitNormal);
    st.rays++;
    PtHit ah = brute ? trace_brute(S, hitPos, 0.01f, aoDir, c.myAoDistance) : trace_bvh(S, hitPos, 0.01f, aoDir, c.myAoDistance, nullptr);
    if (ah.prim != PT_MISS) ++numAoHits;
  }
  float ao = 1.0f - (float)numAoHits / (float)numAoRays;
  return f3(ao, ao, ao);
}

}  // namespace

extern "C" {

// ---- exact integer pins ----
void orc_pcg4d(const uint32_t in[4], uint32_t out[4]) {
  uint4 r = pcg4d(uint4{in[0], in[1], in[2], in[3]});
  out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void orc_rand01(uint32_t px, uint32_t py, uint32_t seed, uint32_t n, float* out) {
  uint4 s = InitRNG(px, py, seed);
  for (uint32_t i = 0; i < n; ++i) out[i] = GetRand01(s);
}
uint32_t orc_encode_unorm_rgba(const float c[4]) {
  uint32_t r = 0;
  for (int k = 0; k < 4; ++k) {
    float v = clampf(c[k], 0.0f, 1.0f);
    r |= (uint32_t)floorf(v * 255.0f + 0.5f) << (8 * k);
  }
  return r;
}
uint16_t orc_float_to_half(float f) { return float_to_half(f); }
float orc_half_to_float(uint16_t h) { return half_to_float(h); }

void orc_earth_atmosphere(PtAtmosphere* out) {
  Atmosphere a;
  SetupEarthAtmosphere(a);
  memcpy(out, &a, sizeof(a));
}

// compute_transmittance_lut.hlsl over the whole 256x64 texture -> RGBA16F
void orc_transmittance_lut(const PtAtmosphere* atm, uint16_t* out_rgba16f) {
  Atmosphere A;
  memcpy(&A, atm, sizeof(A));
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = 0; y < TRANSMITTANCE_TEXTURE_HEIGHT; ++y)
    for (int x = 0; x < TRANSMITTANCE_TEXTURE_WIDTH; ++x)
      TransmittanceLutTexel(A, (uint32_t)x, (uint32_t)y, &out_rgba16f[((size_t)y * TRANSMITTANCE_TEXTURE_WIDTH + x) * 4]);
}

// Row layouts identical to pt_debug_eval (include/pt_api.h, PT_FN_*).
int orc_eval(int fn, const PtRtConsts* c, const uint16_t* lut, const float* in, uint64_t n, float* out) {
  for (uint64_t i = 0; i < n; ++i) {
    switch (fn) {
      case PT_FN_RAND01: {
        const float* r = in + 4 * i;
        uint4 s = InitRNG((uint32_t)r[0], (uint32_t)r[1], (uint32_t)r[2]);
        uint32_t nd = (uint32_t)r[3];
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
        float3 d = SampleModifiedPhong(float2{r[0], r[1]}, ld3(r + 2), r[5], pdf);
        out[4 * i] = d.x; out[4 * i + 1] = d.y; out[4 * i + 2] = d.z; out[4 * i + 3] = pdf;
      } break;
      case PT_FN_EVAL_PHONG: {
        const float* r = in + 10 * i;
        out[i] = EvaluateModifiedPhong(ld3(r), ld3(r + 3), ld3(r + 6), 0.9f, r[9]);
      } break;
      case PT_FN_COS_HEMISPHERE: {
        const float* r = in + 5 * i;
        float3 d = GetCosineWeightedHemisphereDirection(float2{r[0], r[1]}, ld3(r + 2));
        out[3 * i] = d.x; out[3 * i + 1] = d.y; out[3 * i + 2] = d.z;
      } break;
      case PT_FN_LAMBERT: {
        const float* r = in + 9 * i;
        float3 b = GetLambertianBRDF(ld3(r), ld3(r + 3), ld3(r + 6));
        out[4 * i] = b.x; out[4 * i + 1] = b.y; out[4 * i + 2] = b.z; out[4 * i + 3] = GetLambertianPDF(ld3(r + 3), ld3(r + 6));
      } break;
      case PT_FN_SKY: {
        const float* r = in + 6 * i;
        float3 L = SampleSkyLuminance(ld3(r), ld3(r + 3), *c, lut);
        out[3 * i] = L.x; out[3 * i + 1] = L.y; out[3 * i + 2] = L.z;
      } break;
      case PT_FN_PRIMARY_RAY: {
        const float* r = in + 4 * i;
        float3 o, d;
        GetPrimaryRay(float2{r[0], r[1]}, (uint32_t)r[2], (uint32_t)r[3], ld3(c->myNearPlaneCorner), ld3(c->myXAxis),
                      ld3(c->myYAxis), ld3(c->myCameraPos), o, d);
        out[6 * i] = o.x; out[6 * i + 1] = o.y; out[6 * i + 2] = o.z; out[6 * i + 3] = d.x; out[6 * i + 4] = d.y; out[6 * i + 5] = d.z;
      } break;
      case PT_FN_GGX_SAMPLE: {
        const float* r = in + 14 * i;
        uint4 rng = InitRNG((uint32_t)r[11], (uint32_t)r[12], (uint32_t)r[13]);
        float3 T = f3(1.0f), d = f3(0.0f);
        const bool alive = SampleGgxMaterial(ld3(r), r[3], r[4], ld3(r + 5), ld3(r + 8), rng, T, d);
        float* o = out + 8 * i;
        o[0] = alive ? 1.0f : 0.0f; o[1] = T.x; o[2] = T.y; o[3] = T.z; o[4] = d.x; o[5] = d.y; o[6] = d.z; o[7] = (float)rng.w;
      } break;
      case PT_FN_ALBEDO_PATTERN: {
        const float* r = in + 4 * i;
        out[i] = AlbedoPattern((uint32_t)r[0], r[1], float2{r[2], r[3]});
      } break;
      default: return 1;
    }
  }
  return 0;
}

// ---- scene ----
void* orc_scene_create(const PtMeshDesc* meshes, uint32_t nm, const PtInstanceDesc* inst, uint32_t ni,
                       const PtMaterial* mats, uint32_t nmat, uint32_t flags) {
  for (uint32_t i = 0; i < ni; ++i)
    if (inst[i].myMeshIndex >= nm || inst[i].myMaterialIndex >= nmat) return nullptr;
  Scene* S = new Scene();
  flatten(*S, meshes, nm, inst, ni, mats, nmat, flags);
  build_bvh2(*S);
  return S;
}
void orc_scene_destroy(void* s) { delete (Scene*)s; }
// material extension (include/pt_api.h PtMaterialExt): one record per material, or n == 0 to remove
int orc_scene_set_material_ext(void* s, const PtMaterialExt* ext, uint32_t n) {
  Scene& S = *(Scene*)s;
  if (!ext || n == 0) { S.materials_ext.clear(); return 0; }
  if (n != S.materials.size()) return 1;
  S.materials_ext.assign(ext, ext + n);
  return 0;
}
int orc_scene_set_albedo_texture(void* s, const uint8_t* rgba8, uint32_t w, uint32_t h) {
  Scene& S = *(Scene*)s;
  S.albedo_tex.clear();
  S.tex_w = S.tex_h = 0;
  if (!rgba8) return 0;
  if (w == 0 || h == 0) return 1;
  S.albedo_tex.resize((size_t)w * h);
  memcpy(S.albedo_tex.data(), rgba8, (size_t)w * h * 4);
  S.tex_w = w;
  S.tex_h = h;
  return 0;
}
uint32_t orc_scene_num_triangles(void* s) { return (uint32_t)((Scene*)s)->tris.size(); }
// world-space triangle i as 9 floats + its instance id (so tests can check the product's flattening)
void orc_scene_triangle(void* s, uint32_t i, float out9[9], uint32_t* instance) {
  const Scene* S = (const Scene*)s;
  const Tri& t = S->tris[i];
  const float v[9] = {t.v0.x, t.v0.y, t.v0.z, t.v1.x, t.v1.y, t.v1.z, t.v2.x, t.v2.y, t.v2.z};
  memcpy(out9, v, sizeof(v));
  *instance = S->tri_instance[i];
}

// TraceRay over explicit rays. brute!=0: all-triangles loop. counters (optional): [0]=BVH2 nodes, [1]=tris tested.
void orc_trace_rays(void* s, const PtRay* rays, uint64_t n, PtHit* hits, int brute, uint64_t* counters) {
  const Scene& S = *(const Scene*)s;
  uint64_t cn = 0, ct = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : cn, ct)
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    const PtRay& r = rays[i];
    TraceCounters tc;
    hits[i] = brute ? trace_brute(S, ld3(r.origin), r.tmin, ld3(r.dir), r.tmax)
                    : trace_bvh(S, ld3(r.origin), r.tmin, ld3(r.dir), r.tmax, counters ? &tc : nullptr);
    cn += tc.nodes; ct += tc.tris;
  }
  if (counters) { counters[0] = cn; counters[1] = ct; }
}

// RayGen lines 43-71 only: the primary ray of (pixel, sample) and its closest hit.
void orc_primary_hits(void* s, const PtRtConsts* c, uint32_t w, uint32_t h, uint32_t sample, PtHit* hits, int brute,
                      PtRay* rays_out /*optional*/) {
  const Scene& S = *(const Scene*)s;
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = 0; y < (int)h; ++y)
    for (uint32_t x = 0; x < w; ++x) {
      uint4 rng = InitRNG(x, (uint32_t)y, sample);
      float jx = GetRand01(rng), jy = GetRand01(rng);
      float2 pixel = {clampf((float)x + jx, 0.0f, (float)w), clampf((float)y + jy, 0.0f, (float)h)};
      float3 o, d;
      GetPrimaryRay(pixel, w, h, ld3(c->myNearPlaneCorner), ld3(c->myXAxis), ld3(c->myYAxis), ld3(c->myCameraPos), o, d);
      size_t i = (size_t)y * w + x;
      hits[i] = brute ? trace_brute(S, o, 0.0f, d, 10000.0f) : trace_bvh(S, o, 0.0f, d, 10000.0f);
      if (rays_out) {
        PtRay r = {{o.x, o.y, o.z}, 0.0f, {d.x, d.y, d.z}, 10000.0f};
        rays_out[i] = r;
      }
    }
}

// Test probe: the ray every path of sample index `sample` carries into iteration `bounce` of the RayGen loop
// (PathTracing.hlsl:63-71; bounce 0 = the primary ray, bounce b = origin myHitPos / direction sampled at hit b-1, :126-127),
// i.e. the content of the product's bounce-b ray queue, one row per pixel in row order. alive[i] = 0 for a path that ended
// before (sky miss, last bounce, roulette); state (optional, 8 floats per pixel) = transmission rgb, 0, luminance rgb, 0 at
// that point. What it isolates is ClosestHit (:16-38) and the loaders behind it: myHitPos, the interpolated and flipped
// normal (through the sampled direction), colour and emission (through the state).
void orc_bounce_rays(void* s, const PtRtConsts* c, const uint16_t* lut, uint32_t w, uint32_t h, uint32_t sample,
                     uint32_t bounce, int brute, PtRay* rays, uint8_t* alive, float* state /*optional*/) {
  const Scene& S = *(const Scene*)s;
#pragma omp parallel for schedule(dynamic, 2)
  for (int y = 0; y < (int)h; ++y)
    for (uint32_t x = 0; x < w; ++x) {
      SampleStats st; int ox, oy;
      PathProbe pr;
      pr.bounce = bounce;
      RayGenSample(S, *c, lut, x, (uint32_t)y, w, h, sample, 0, brute, st, ox, oy, false, &pr);
      const size_t i = (size_t)y * w + x;
      alive[i] = pr.reached ? 1 : 0;
      PtRay r = {{0.0f, 0.0f, 0.0f}, 0.0f, {0.0f, 0.0f, 0.0f}, 0.0f};
      if (pr.reached) r = PtRay{{pr.origin.x, pr.origin.y, pr.origin.z}, pr.tmin, {pr.dir.x, pr.dir.y, pr.dir.z}, pr.tmax};
      rays[i] = r;
      if (state) {
        float* q = state + i * 8;
        q[0] = pr.reached ? pr.transmission.x : 0.0f; q[1] = pr.reached ? pr.transmission.y : 0.0f; q[2] = pr.reached ? pr.transmission.z : 0.0f; q[3] = 0.0f;
        q[4] = pr.reached ? pr.luminance.x : 0.0f; q[5] = pr.reached ? pr.luminance.y : 0.0f; q[6] = pr.reached ? pr.luminance.z : 0.0f; q[7] = 0.0f;
      }
    }
}

// Whole RenderRT equivalent on the CPU: samples [first_sample, first_sample+num_samples) of every pixel.
//   accum_rgba : w*h*4 floats, in/out.
//   accum_mode 0: accum += (L,0)  (sum; caller divides by the count — the product's target format)
//   accum_mode 1: accum = (accum*N + (L,0)) / (N+1), N = n_prev + i  — the reference's literal running mean
//                 (PathTracing.hlsl:134-137, myNumAccumulationFrames PathTracer.cpp:764)
//   per_sample_rgb (optional): w*h*num_samples*3 floats, [sample][y][x][rgb] (jitter_mode 0 only)
//   stats (optional, 8 x u64): rays, shaded hits, sky misses, samples, bvh2 nodes, tris tested, ns elapsed, threads
void orc_render(void* s, const PtRtConsts* c, const uint16_t* lut, uint32_t w, uint32_t h, uint32_t first_sample,
                uint32_t num_samples, float* accum_rgba, uint64_t n_prev, int accum_mode, int jitter_mode, int brute,
                int count_traversal, uint64_t* stats, float* per_sample_rgb) {
  const Scene& S = *(const Scene*)s;
  uint64_t rays = 0, hits = 0, sky = 0, nodes = 0, tris = 0;
  auto t0 = std::chrono::steady_clock::now();
  for (uint32_t si = 0; si < num_samples; ++si) {  // one reference "frame" per sample index
    const uint32_t seed = first_sample + si;
    const float N = (float)(n_prev + si);
    if (jitter_mode == 1) {
      // literal reference indexing scatters to neighbouring texels: run serially (it is a data race on the GPU, Q1)
      for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
          SampleStats st; int ox, oy;
          float3 L = RayGenSample(S, *c, lut, x, y, w, h, seed, 1, brute, st, ox, oy, count_traversal != 0);
          rays += st.rays; hits += st.hits; sky += st.sky; nodes += st.nodes; tris += st.tris;
          if (ox < 0 || oy < 0 || ox >= (int)w || oy >= (int)h) continue;  // out-of-bounds UAV write is dropped
          float* a = &accum_rgba[((size_t)oy * w + ox) * 4];
          const float Lv[4] = {L.x, L.y, L.z, 0.0f};
          for (int k = 0; k < 4; ++k) a[k] = accum_mode == 1 ? (a[k] * N + Lv[k]) / (N + 1.0f) : a[k] + Lv[k];
        }
      continue;
    }
#pragma omp parallel for schedule(dynamic, 2) reduction(+ : rays, hits, sky, nodes, tris)
    for (int y = 0; y < (int)h; ++y)
      for (uint32_t x = 0; x < w; ++x) {
        SampleStats st; int ox, oy;
        float3 L = RayGenSample(S, *c, lut, x, (uint32_t)y, w, h, seed, 0, brute, st, ox, oy, count_traversal != 0);
        rays += st.rays; hits += st.hits; sky += st.sky; nodes += st.nodes; tris += st.tris;
        float* a = &accum_rgba[((size_t)y * w + x) * 4];
        const float Lv[4] = {L.x, L.y, L.z, 0.0f};
        for (int k = 0; k < 4; ++k) a[k] = accum_mode == 1 ? (a[k] * N + Lv[k]) / (N + 1.0f) : a[k] + Lv[k];
        if (per_sample_rgb) {
          float* p = &per_sample_rgb[(((size_t)si * h + y) * w + x) * 3];
          p[0] = L.x; p[1] = L.y; p[2] = L.z;
        }
      }
  }
  auto t1 = std::chrono::steady_clock::now();
  if (stats) {
    stats[0] = rays; stats[1] = hits; stats[2] = sky; stats[3] = (uint64_t)w * h * num_samples;
    stats[4] = nodes; stats[5] = tris;
    stats[6] = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
#ifdef _OPENMP
    stats[7] = (uint64_t)omp_get_max_threads();
#else
    stats[7] = 1;
#endif
  }
}

// The AO pipeline (Ao.hlsl) over samples [first_sample, first_sample+num_samples): accum += (L,0). No NaN scrub in the
// reference's AO raygen (:99-102), and none here. stats as orc_render.
void orc_render_ao(void* s, const PtRtConsts* c, const uint16_t* lut, uint32_t w, uint32_t h, uint32_t first_sample,
                   uint32_t num_samples, float* accum_rgba, int brute, uint64_t* stats) {
  const Scene& S = *(const Scene*)s;
  uint64_t rays = 0, hits = 0, sky = 0;
  auto t0 = std::chrono::steady_clock::now();
  for (uint32_t si = 0; si < num_samples; ++si) {
    const uint32_t seed = first_sample + si;
#pragma omp parallel for schedule(dynamic, 2) reduction(+ : rays, hits, sky)
    for (int y = 0; y < (int)h; ++y)
      for (uint32_t x = 0; x < w; ++x) {
        SampleStats st;
        float3 L = RayGenAoSample(S, *c, lut, x, (uint32_t)y, w, h, seed, brute, st);
        rays += st.rays; hits += st.hits; sky += st.sky;
        float* a = &accum_rgba[((size_t)y * w + x) * 4];
        a[0] += L.x; a[1] += L.y; a[2] += L.z;
      }
  }
  auto t1 = std::chrono::steady_clock::now();
  if (stats) {
    stats[0] = rays; stats[1] = hits; stats[2] = sky; stats[3] = (uint64_t)w * h * num_samples;
    stats[4] = 0; stats[5] = 0;
    stats[6] = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
#ifdef _OPENMP
    stats[7] = (uint64_t)omp_get_max_threads();
#else
    stats[7] = 1;
#endif
  }
}

void orc_set_russian_roulette(int on) { g_russian_roulette = on ? 1 : 0; }

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void orc_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

}  // extern "C"
