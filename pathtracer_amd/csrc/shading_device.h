// shading_device.h — device restatement of the reference's RNG, camera ray, Fresnel, Lambert and modified-Phong
// code (resources/shaders/raytracing/{Random,Common,brdfSampling}.hlsl). Citations name the HLSL lines.
#pragma once
#include "device_math.h"

namespace ptd {

// ---- Random.hlsl ----
struct Rng { uint32_t x, y, z, w; };

// pcg4d — Random.hlsl:45-62 (only .x is consumed by GetRand01; the last w update is dead and dropped)
__device__ __forceinline__ uint32_t pcg4d_x(Rng v) {
  v.x = v.x * 1664525u + 1013904223u;
  v.y = v.y * 1664525u + 1013904223u;
  v.z = v.z * 1664525u + 1013904223u;
  v.w = v.w * 1664525u + 1013904223u;
  v.x += v.y * v.w;
  v.y += v.z * v.x;
  v.z += v.x * v.y;
  v.w += v.y * v.z;
  v.x ^= v.x >> 16u;
  v.y ^= v.y >> 16u;
  v.z ^= v.z >> 16u;
  v.w ^= v.w >> 16u;
  v.x += v.y * v.w;
  return v.x;
}
// uintToFloat — Random.hlsl:84-86
__device__ __forceinline__ float uintToFloat(uint32_t x) { return __uint_as_float(0x3f800000u | (x >> 9)) - 1.0f; }
// InitRNG — Random.hlsl:91-93 (frameNumber = sample index)
__device__ __forceinline__ Rng InitRNG(uint32_t px, uint32_t py, uint32_t frame) { Rng r; r.x = px; r.y = py; r.z = frame; r.w = 0u; return r; }
// GetRand01 — Random.hlsl:96-99
__device__ __forceinline__ float GetRand01(Rng& s) {
  s.w++;
  return uintToFloat(pcg4d_x(s));
}

// ---- Common.hlsl ----
// GetPrimaryRay — Common.hlsl:183-189
__device__ __forceinline__ void GetPrimaryRay(v2 pixel, uint32_t resX, uint32_t resY, v3 corner, v3 xAxis, v3 yAxis, v3 camPos,
                                              v3& origin, v3& dir) {
  float vx = pixel.x / (float)resX;
  float vy = pixel.y / (float)resY;
  vy = 1.0f - vy;
  origin = corner + xAxis * vx + yAxis * vy;
  dir = normalize(origin - camPos);
}
// RayGen's ray set-up for path sample p of a batch (PathTracing.hlsl:40-61 with the Q1 deviation: position = uPixel +
// jitter). One function for k_generate, the fused fetch of k_extend and bounce 0 of k_shade, so all three produce the
// same bits. p = sample_in_batch * w*h + y*w + x; returns the RNG draw counter after the two jitter draws.
// Path ids of a batch of k samples per pixel are PIXEL-MAJOR: p = (y*w + x) * k + sample. The k samples of a pixel are
// neighbours in every queue, so a wave of the coherent-primaries kernel holds one pixel's samples (k >= 64) or a few
// adjacent pixels', secondary rays of a wave start from almost the same point, and the path-state accesses of consecutive
// queue slots stay contiguous. (k = 1, the AO pipeline and pt_primary_hits: p = pixel.)
struct PrimaryFrame { v3 corner, xAxis, yAxis, camPos; uint32_t w, h, wh, first_sample, k; };
__device__ __forceinline__ void SplitPathId(uint32_t p, uint32_t k, uint32_t& si, uint32_t& pix) {
  if ((k & (k - 1u)) == 0u) {   // wave-uniform: powers of two by shift
    si = p & (k - 1u);
    pix = p >> (31 - __clz((int)k));
  } else {
    pix = p / k;
    si = p - pix * k;
  }
}
__device__ __forceinline__ uint32_t PrimaryRayOfPixel(uint32_t px, uint32_t py, uint32_t si, const PrimaryFrame& f, v3& origin, v3& dir) {
  Rng rng = InitRNG(px, py, f.first_sample + si);
  v2 pixel;
  pixel.x = (float)px;
  pixel.y = (float)py;
  const float jx = GetRand01(rng);
  const float jy = GetRand01(rng);
  pixel.x += jx;
  pixel.y += jy;
  pixel.x = clampf(pixel.x, 0.0f, (float)f.w);
  pixel.y = clampf(pixel.y, 0.0f, (float)f.h);
  GetPrimaryRay(pixel, f.w, f.h, f.corner, f.xAxis, f.yAxis, f.camPos, origin, dir);
  return rng.w;
}
__device__ __forceinline__ uint32_t PrimaryRayOfPath(uint32_t p, const PrimaryFrame& f, v3& origin, v3& dir) {
  uint32_t pix, si, px, py;
  SplitPathId(p, f.k, si, pix);
  FastDiv(f.w, f.wh).divmod(pix, py, px);
  return PrimaryRayOfPixel(px, py, si, f, origin, dir);
}
// Fused primaries only (no k_generate pass, so nothing else sees the order): slot j of the bounce-0 queue enumerates the
// PIXELS in 8x8-pixel tile order (sample-minor as the path ids), so the 64 consecutive slots a wave fetches are the samples
// of one pixel or of a few pixels of one tile: a footprint of a pixel or two instead of a 64x1 strip (fewer distinct BVH
// nodes per box test; what k_extend_packet lives on). The path id -- what the RNG, the path state and k_resolve are keyed
// on -- keeps its meaning, so the target bits do not change. Frames whose width or height is not a multiple of 8 keep the
// row order of pixels (wave-uniform test). Returns the path id.
#ifndef PT_PRIMARY_TILES
#define PT_PRIMARY_TILES 1
#endif
__device__ __forceinline__ uint32_t PrimaryRayOfSlot(uint32_t j, const PrimaryFrame& f, v3& origin, v3& dir) {
  uint32_t r, si, px, py;
  SplitPathId(j, f.k, si, r);
  if (PT_PRIMARY_TILES && ((f.w | f.h) & 7u) == 0u) {
    uint32_t ty, tx;
    FastDiv(f.w >> 3, f.wh >> 6).divmod(r >> 6, ty, tx);
    px = tx * 8u + (r & 7u);
    py = ty * 8u + ((r >> 3) & 7u);
  } else {
    FastDiv(f.w, f.wh).divmod(r, py, px);
  }
  (void)PrimaryRayOfPixel(px, py, si, f, origin, dir);
  return (py * f.w + px) * f.k + si;
}
// GetLuminance — Common.hlsl:191-194
__device__ __forceinline__ float GetLuminance(v3 c) { return dot(c, mk3(0.2126f, 0.7152f, 0.0722f)); }
// GetFresnelSchlick — Common.hlsl:196-201
__device__ __forceinline__ float GetFresnelSchlick(v3 N, v3 V) {
  const float f0 = 0.04f;
  float cosTheta = fmaxf(0.0f, dot(N, V));
  return saturate(f0 + (1.0f - f0) * hlsl_pow5(1.0f - cosTheta));
}
// GetUniformRandomDirectionInSphere — Common.hlsl:129-135
// theta = 2 acos(sqrt(1 - u.y)) is evaluated through the half-angle identities: cos(theta) = 2 (1 - u.y) - 1 = 1 - 2 u.y,
// sin(theta) = 2 sqrt(u.y) sqrt(1 - u.y) — the same real-valued function without acos and two of the four sin/cos (and
// without the cancellation acos has near 1); phi's sin/cos come from the hardware unit (device_math.h).
__device__ __forceinline__ v3 GetUniformRandomDirectionInSphere(v2 u) {
  if (PT_SHADE_LIBM_FORMS) {
    float phi = 2.0f * PT_PI * u.x;
    float theta = 2.0f * acosf(sqrtf(1.0f - u.y));
    return mk3(sinf(theta) * cosf(phi), sinf(theta) * sinf(phi), cosf(theta));
  }
  const float sinTheta = 2.0f * sqrtf(u.y * (1.0f - u.y));
  const float cosTheta = 1.0f - 2.0f * u.y;
  return mk3(sinTheta * cos_2pi(u.x), sinTheta * sin_2pi(u.x), cosTheta);
}
// GetCosineWeightedHemisphereDirection — Common.hlsl:137-141
__device__ __forceinline__ v3 GetCosineWeightedHemisphereDirection(v2 u, v3 N) {
  return normalize(N + GetUniformRandomDirectionInSphere(u));
}
// GetCoordinateFrame + TransformToNormalFrame — Common.hlsl:150-157,168-176 (rows of tbn dotted with dir, Q6)
__device__ __forceinline__ v3 TransformToNormalFrame(v3 N, v3 dir) {
  v3 side = fabsf(N.x) < 0.999f ? mk3(1.0f, 0.0f, 0.0f) : mk3(0.0f, 0.0f, 1.0f);
  v3 z = normalize(cross(-N, -side));
  v3 x = cross(z, N);
  return mk3(dot(x, dir), dot(N, dir), dot(z, dir));
}
// GetHemisphereDirection — Common.hlsl:159-166 (used by Ao.hlsl only); tangent/bitangent as in GetCoordinateFrame
__device__ __forceinline__ v3 GetHemisphereDirection(v2 rand11, v3 N) {
  v3 side = fabsf(N.x) < 0.999f ? mk3(1.0f, 0.0f, 0.0f) : mk3(0.0f, 0.0f, 1.0f);
  v3 z = normalize(cross(-N, -side));
  v3 x = cross(z, N);
  return normalize(N + x * rand11.x + z * rand11.y);
}

// ---- brdfSampling.hlsl ----
// GetLambertianBRDF / GetLambertianPDF — brdfSampling.hlsl:6-14
__device__ __forceinline__ v3 GetLambertianBRDF(v3 albedo, v3 N, v3 L) { return albedo * fmaxf(0.0f, dot(N, L)) * (1.0f / PT_PI); }
__device__ __forceinline__ float GetLambertianPDF(v3 N, v3 L) { return fmaxf(0.0f, dot(N, L)) * (1.0f / PT_PI); }
// EvaluateModifiedPhong — brdfSampling.hlsl:16-22
__device__ __forceinline__ float EvaluateModifiedPhong(v3 N, v3 L, v3 V, float specularStrength, float specularPower) {
  v3 R = normalize(reflect(-L, N));
  float cosAlpha = fmaxf(0.0f, dot(R, V));
  return specularStrength * ((specularPower + 2.0f) / PT_TWO_PI) * hlsl_pow_hw(cosAlpha, specularPower) * fmaxf(0.0f, dot(N, L));
}
// SampleModifiedPhong — brdfSampling.hlsl:24-39
__device__ __forceinline__ v3 SampleModifiedPhong(v2 u, v3 N, float specularPower, float& pdf) {
  float cosTheta = hlsl_pow_hw(1.0f - u.x, 1.0f / (2.0f + specularPower));
  float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
  pdf = ((specularPower + 2.0f) / PT_TWO_PI) * hlsl_pow_hw(cosTheta, specularPower);
  v3 d = mk3(cos_2pi(u.y) * sinTheta, cosTheta, sin_2pi(u.y) * sinTheta);   // phi = 2 pi u.y
  return TransformToNormalFrame(N, d);
}
// EstimateSpecularRayProbability — brdfSampling.hlsl:41-49
__device__ __forceinline__ float EstimateSpecularRayProbability(float specStrength, v3 baseColor, float fresnel) {
  float spec = fresnel * specStrength;
  float diffuse = (1.0f - fresnel) * GetLuminance(baseColor);
  float sum = spec + diffuse;
  float p = sum > 0.001f ? spec / sum : 0.0f;
  return clampf(p, 0.1f, 0.9f);
}
// Decode_Unorm_RGBA (FANCY Encoding.h, absent): byte / 255, R in the low byte.
__device__ __forceinline__ v3 DecodeUnormRGB(uint32_t c) {
  return mk3((float)(c & 0xffu) / 255.0f, (float)((c >> 8) & 0xffu) / 255.0f, (float)((c >> 16) & 0xffu) / 255.0f);
}

}  // namespace ptd
