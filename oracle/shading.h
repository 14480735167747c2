// ORACLE — TEST INFRASTRUCTURE ONLY (see hlsl_math.h). Scalar restatement of the reference's
// RNG, camera-ray, BRDF and sampling HLSL. Each function cites the reference lines it follows
// (paths relative to /root/reference/resources/shaders/raytracing/).
#pragma once
#include "hlsl_math.h"

namespace orc {

// ---------------------------------------------------------------- RNG (Random.hlsl) ----
// pcg4d — Random.hlsl:45-62. Pure u32 arithmetic: exact.
static inline uint4 pcg4d(uint4 v) {
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
  v.y += v.z * v.x;
  v.z += v.x * v.y;
  v.w += v.y * v.z;
  return v;
}
// uintToFloat — Random.hlsl:84-86
static inline float uintToFloat(uint32_t x) { return asfloat(0x3f800000u | (x >> 9)) - 1.0f; }
// InitRNG (PCG version) — Random.hlsl:91-93. resolution is unused by the PCG variant.
static inline uint4 InitRNG(uint32_t px, uint32_t py, uint32_t frameNumber) { return uint4{px, py, frameNumber, 0u}; }
// GetRand01 (PCG version) — Random.hlsl:96-99
static inline float GetRand01(uint4& rngState) {
  rngState.w++;
  return uintToFloat(pcg4d(rngState).x);
}

// ---------------------------------------------------------------- Common.hlsl ----
// GetPrimaryRay — Common.hlsl:183-189
static inline void GetPrimaryRay(float2 pixel, uint32_t resX, uint32_t resY, float3 nearPlaneCorner, float3 xAxis,
                                 float3 yAxis, float3 cameraPos, float3& origin, float3& dir) {
  float2 vpLerp = {pixel.x / (float)resX, pixel.y / (float)resY};
  vpLerp.y = 1.0f - vpLerp.y;
  origin = nearPlaneCorner + xAxis * vpLerp.x + yAxis * vpLerp.y;
  dir = normalize(origin - cameraPos);
}
// GetLuminance — Common.hlsl:191-194
static inline float GetLuminance(float3 radiance) { return dot(radiance, f3(0.2126f, 0.7152f, 0.0722f)); }
// GetFresnelSchlick — Common.hlsl:196-201
static inline float GetFresnelSchlick(float3 aNormal, float3 aView) {
  const float f0 = 0.04f;
  float cosTheta = hmax(0.0f, dot(aNormal, aView));
  return saturate(f0 + (1.0f - f0) * hlsl_pow(1.0f - cosTheta, 5.0f));
}
// GetUniformRandomDirectionInSphere — Common.hlsl:129-135
static inline float3 GetUniformRandomDirectionInSphere(float2 aRand01) {
  float phi = 2.0f * PI * aRand01.x;
  float theta = 2.0f * acosf(sqrtf(1.0f - aRand01.y));
  return f3(sinf(theta) * cosf(phi), sinf(theta) * sinf(phi), cosf(theta));
}
// GetCosineWeightedHemisphereDirection — Common.hlsl:137-141 (aPoint is unused by the reference)
static inline float3 GetCosineWeightedHemisphereDirection(float2 aRand01, float3 aNormal) {
  return normalize(aNormal + GetUniformRandomDirectionInSphere(aRand01));
}
// GetCoordinateFrame — Common.hlsl:150-157
static inline void GetCoordinateFrame(float3 aNormal, float3& tangentOut, float3& bitangentOut) {
  float3 side = fabsf(aNormal.x) < 0.999f ? f3(1, 0, 0) : f3(0, 0, 1);
  float3 z = normalize(cross(-aNormal, -side));
  float3 x = cross(z, aNormal);
  tangentOut = x;
  bitangentOut = z;
}
// TransformToNormalFrame — Common.hlsl:168-176. float3x3(t,n,b) has those vectors as ROWS, and
// mul(matrix, vector) dots each row with the vector (this is the transposed rotation; kept).
static inline float3 TransformToNormalFrame(float3 aNormal, float3 aDir) {
  float3 tangent, bitangent;
  GetCoordinateFrame(aNormal, tangent, bitangent);
  return f3(dot(tangent, aDir), dot(aNormal, aDir), dot(bitangent, aDir));
}
// GetHemisphereDirection — Common.hlsl:159-166 (used by Ao.hlsl only)
static inline float3 GetHemisphereDirection(float2 aRand11, float3 aNormal) {
  float3 tangent, bitangent;
  GetCoordinateFrame(aNormal, tangent, bitangent);
  return normalize(aNormal + tangent * aRand11.x + bitangent * aRand11.y);
}

// ---------------------------------------------------------------- brdfSampling.hlsl ----
// GetLambertianBRDF — brdfSampling.hlsl:6-9
static inline float3 GetLambertianBRDF(float3 diffuseReflectance, float3 N, float3 L) {
  return diffuseReflectance * hmax(0.0f, dot(N, L)) * (1.0f / PI);
}
// GetLambertianPDF — brdfSampling.hlsl:11-14
static inline float GetLambertianPDF(float3 N, float3 L) { return hmax(0.0f, dot(N, L)) * (1.0f / PI); }
// EvaluateModifiedPhong — brdfSampling.hlsl:16-22
static inline float EvaluateModifiedPhong(float3 N, float3 L, float3 V, float specularStrength, float specularPower) {
  float3 R = normalize(reflect(-L, N));
  float cosAlpha = hmax(0.0f, dot(R, V));
  float brdf = specularStrength * ((specularPower + 2.0f) / TWO_PI) * hlsl_pow(cosAlpha, specularPower) *
               hmax(0.0f, dot(N, L));
  return brdf;
}
// SampleModifiedPhong — brdfSampling.hlsl:24-39
static inline float3 SampleModifiedPhong(float2 aRand01, float3 aNormal, float specularPower, float& pdf) {
  float cosTheta = hlsl_pow(1.0f - aRand01.x, 1.0f / (2.0f + specularPower));
  float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
  float phi = TWO_PI * aRand01.y;
  pdf = ((specularPower + 2.0f) / TWO_PI) * hlsl_pow(cosTheta, specularPower);
  float3 sampleDir = f3(cosf(phi) * sinTheta, cosTheta, sinf(phi) * sinTheta);
  return TransformToNormalFrame(aNormal, sampleDir);
}
// EstimateSpecularRayProbability — brdfSampling.hlsl:41-49
static inline float EstimateSpecularRayProbability(float aMaterialSpecularStrength, float3 aMaterialBaseColor,
                                                   float aFresnel) {
  float spec = aFresnel * aMaterialSpecularStrength;
  float diffuse = (1.0f - aFresnel) * GetLuminance(aMaterialBaseColor);
  float specAndDiffuse = spec + diffuse;
  float specRayProbability = specAndDiffuse > 0.001f ? spec / specAndDiffuse : 0.0f;
  specRayProbability = clampf(specRayProbability, 0.1f, 0.9f);
  return specRayProbability;
}

// Decode_Unorm_RGBA — FANCY Encoding.h is absent (parity unpinned):
// adopted encode = round(clamp(c,0,1)*255) per channel, R in the low byte; decode = byte / 255.
static inline float3 Decode_Unorm_RGB(uint32_t c) {
  return f3((float)(c & 0xffu) / 255.0f, (float)((c >> 8) & 0xffu) / 255.0f, (float)((c >> 16) & 0xffu) / 255.0f);
}

}  // namespace orc
