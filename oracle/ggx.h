// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU restatement of the material extension for BASELINE configs C4/C5 (GGX + metallic/roughness + procedural albedo
// patterns). There is NO counterpart in the reference (its materials are untextured Lambert + modified Phong,
// PathTracing.hlsl:83-106); the definition lives in pathtracer_amd/csrc/ggx_device.h and this file follows it
// operation by operation. Parity of this path is therefore GPU <-> oracle only ("parity unpinned" against any
// reference output).
#pragma once
#include "shading.h"

namespace orc {

static const uint32_t PT_MAT_GGX_BIT = 1u;

static inline float lattice01(int ix, int iy, uint32_t seed) {
  uint4 r = {(uint32_t)ix, (uint32_t)iy, seed, 1u};
  return uintToFloat(pcg4d(r).x);
}

static inline float AlbedoPattern(uint32_t pattern, float scale, float2 uv) {
  if (pattern == 0u) return 1.0f;
  const float px = uv.x * scale, py = uv.y * scale;
  const float fx = floorf(px), fy = floorf(py);
  const int ix = (int)fx, iy = (int)fy;
  if (pattern == 1u) return ((ix + iy) & 1) ? 1.0f : 0.35f;
  const float tx = px - fx, ty = py - fy;
  const float sx = tx * tx * (3.0f - 2.0f * tx), sy = ty * ty * (3.0f - 2.0f * ty);
  const float v00 = lattice01(ix, iy, 0x9E3779B9u), v10 = lattice01(ix + 1, iy, 0x9E3779B9u);
  const float v01 = lattice01(ix, iy + 1, 0x9E3779B9u), v11 = lattice01(ix + 1, iy + 1, 0x9E3779B9u);
  const float a = v00 + (v10 - v00) * sx, b = v01 + (v11 - v01) * sx;
  return 0.35f + 0.65f * (a + (b - a) * sy);
}

// Albedo pattern 3 (ggx_device.h SampleAlbedoTexture, operation by operation): RGBA8 image, repeat addressing, bilinear
// filtering of the decoded texels, texel centres at half-integer coordinates.
static inline float3 TexelRgb(const uint32_t* tex, uint32_t w, int x, int y) {
  const uint32_t c = tex[(size_t)y * w + (uint32_t)x];
  return float3{(float)(c & 0xffu) / 255.0f, (float)((c >> 8) & 0xffu) / 255.0f, (float)((c >> 16) & 0xffu) / 255.0f};
}
static inline int WrapTexel(int i, int n) { const int m = i % n; return m < 0 ? m + n : m; }
static inline float3 SampleAlbedoTexture(const uint32_t* tex, uint32_t w, uint32_t h, float scale, float2 uv) {
  if (tex == nullptr || w == 0u || h == 0u) return f3(1.0f);
  const float px = uv.x * scale * (float)w - 0.5f, py = uv.y * scale * (float)h - 0.5f;
  if (!(fabsf(px) < 1.0e9f) || !(fabsf(py) < 1.0e9f)) return f3(1.0f);
  const float fx = floorf(px), fy = floorf(py);
  const float tx = px - fx, ty = py - fy;
  const int x0 = WrapTexel((int)fx, (int)w), y0 = WrapTexel((int)fy, (int)h);
  const int x1 = x0 + 1 == (int)w ? 0 : x0 + 1, y1 = y0 + 1 == (int)h ? 0 : y0 + 1;
  const float3 c00 = TexelRgb(tex, w, x0, y0), c10 = TexelRgb(tex, w, x1, y0), c01 = TexelRgb(tex, w, x0, y1), c11 = TexelRgb(tex, w, x1, y1);
  const float3 a = c00 + (c10 - c00) * tx, b = c01 + (c11 - c01) * tx;
  return a + (b - a) * ty;
}

static inline float3 SchlickF(float3 F0, float c) {
  const float m = 1.0f - saturate(c);
  const float m2 = m * m;
  const float m5 = m2 * m2 * m;
  return F0 + (f3(1.0f) - F0) * m5;
}

static inline float SmithG1(float3 w, float alpha) {
  const float t2 = (w.x * w.x + w.y * w.y) / (w.z * w.z);
  return 2.0f / (1.0f + sqrtf(1.0f + alpha * alpha * t2));
}

static inline bool SampleGgxMaterial(float3 base, float roughness, float metallic, float3 hitNormal, float3 V, uint4& rng,
                                     float3& transmission, float3& newDir) {
  const float3 n = normalize(hitNormal);
  const float3 side = fabsf(n.x) < 0.999f ? f3(1.0f, 0.0f, 0.0f) : f3(0.0f, 0.0f, 1.0f);
  const float3 tz = normalize(cross(-n, -side));
  const float3 tx = cross(tz, n);
  const float alpha = fmaxf(roughness * roughness, 1e-3f);
  const float3 F0 = f3(0.04f) + (base - f3(0.04f)) * metallic;
  const float3 diffuse = base * (1.0f - metallic);
  const float NdotV = fmaxf(dot(n, V), 1e-4f);
  const float lumF = GetLuminance(SchlickF(F0, NdotV));
  const float lumD = GetLuminance(diffuse) * (1.0f - lumF);
  const float pSpec = clampf(lumF / (lumF + lumD), 0.1f, 0.9f);
  if (GetRand01(rng) < pSpec) {
    const float u1 = GetRand01(rng);
    const float u2 = GetRand01(rng);
    const float3 Ve = f3(dot(tx, V), dot(tz, V), NdotV);
    const float3 Vh = normalize(f3(alpha * Ve.x, alpha * Ve.y, Ve.z));
    const float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    const float3 T1 = lensq > 0.0f ? f3(-Vh.y, Vh.x, 0.0f) / sqrtf(lensq) : f3(1.0f, 0.0f, 0.0f);
    const float3 T2 = cross(Vh, T1);
    const float r = sqrtf(u1);
    const float phi = TWO_PI * u2;
    const float t1 = r * cosf(phi);
    float t2 = r * sinf(phi);
    const float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * sqrtf(fmaxf(0.0f, 1.0f - t1 * t1)) + s * t2;
    const float3 Nh = t1 * T1 + t2 * T2 + sqrtf(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
    const float3 Ne = normalize(f3(alpha * Nh.x, alpha * Nh.y, fmaxf(0.0f, Nh.z)));
    const float VdotH = dot(Ve, Ne);
    const float3 L = 2.0f * VdotH * Ne - Ve;
    if (!(L.z > 0.0f)) return false;
    const float3 weight = SchlickF(F0, VdotH) * SmithG1(L, alpha);
    transmission = transmission * weight;
    transmission = transmission / pSpec;
    newDir = tx * L.x + tz * L.y + n * L.z;
  } else {
    float2 u;
    u.x = GetRand01(rng);
    u.y = GetRand01(rng);
    const float3 weight = diffuse * (f3(1.0f) - SchlickF(F0, NdotV));
    transmission = transmission * weight;
    transmission = transmission / (1.0f - pSpec);
    newDir = GetCosineWeightedHemisphereDirection(u, n);
  }
  return true;
}

}  // namespace orc
