// ggx_device.h — material extension for BASELINE configs C4/C5: GGX microfacet specular + Lambert
// diffuse with a metallic/roughness parametrisation and procedural albedo patterns driven by the interpolated uv.
// NOT in the reference (its materials are untextured Lambert + modified Phong, PathTracing.hlsl:83-106): this file is
// the definition, oracle/ggx.h restates it on the CPU, and only materials flagged PT_MAT_GGX take this path.
//   F0 = lerp(0.04, base, metallic), diffuse colour = base * (1 - metallic), alpha = max(roughness^2, 1e-3)
//   lobe choice p = clamp(lum(F) / (lum(F) + lum(diffuse) * (1 - lum(F))), 0.1, 0.9), F = Schlick(F0, N.V)
//   specular: half vector from the distribution of visible normals (Heitz 2018), weight F(V.H) * G1(L) (separable
//             Smith), direction below the surface -> the path ends;   diffuse: cosine-weighted, weight diffuse*(1-F)
// RNG use is the reference's: one draw for the lobe, two for the direction.
#pragma once
#include "shading_device.h"

namespace ptd {

#define PT_MAT_GGX_BIT 1u

// value in [0,1) on the integer lattice: the RNG's own hash, so CPU and GPU agree bit for bit
__device__ __forceinline__ float lattice01(int ix, int iy, uint32_t seed) {
  Rng r; r.x = (uint32_t)ix; r.y = (uint32_t)iy; r.z = seed; r.w = 1u;
  return uintToFloat(pcg4d_x(r));
}

// procedural albedo factor: 0 none, 1 checker, 2 smooth value noise (3 = image: SampleAlbedoTexture below)
__device__ __forceinline__ float AlbedoPattern(uint32_t pattern, float scale, v2 uv) {
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

// Albedo pattern 3: an RGBA8 image (pt_scene_set_albedo_texture), repeat addressing, bilinear filtering of the decoded
// texels (c / 255, R in the low byte as Decode_Unorm_RGBA), texel centres at half-integer coordinates; uv is multiplied by
// the material's pattern_scale first. Plain fp32 multiply / add in this order, so the oracle reproduces it bit for bit.
// No image set, or non-finite coordinates: factor 1.
__device__ __forceinline__ v3 TexelRgb(const uint32_t* __restrict__ tex, uint32_t w, int x, int y) {
  const uint32_t c = tex[(size_t)y * w + (uint32_t)x];
  return mk3((float)(c & 0xffu) / 255.0f, (float)((c >> 8) & 0xffu) / 255.0f, (float)((c >> 16) & 0xffu) / 255.0f);
}
__device__ __forceinline__ int WrapTexel(int i, int n) { const int m = i % n; return m < 0 ? m + n : m; }
__device__ __forceinline__ v3 SampleAlbedoTexture(const uint32_t* __restrict__ tex, uint32_t w, uint32_t h, float scale, v2 uv) {
  if (tex == nullptr || w == 0u || h == 0u) return mk3(1.0f);
  const float px = uv.x * scale * (float)w - 0.5f, py = uv.y * scale * (float)h - 0.5f;
  if (!(fabsf(px) < 1.0e9f) || !(fabsf(py) < 1.0e9f)) return mk3(1.0f);
  const float fx = floorf(px), fy = floorf(py);
  const float tx = px - fx, ty = py - fy;
  const int x0 = WrapTexel((int)fx, (int)w), y0 = WrapTexel((int)fy, (int)h);
  const int x1 = x0 + 1 == (int)w ? 0 : x0 + 1, y1 = y0 + 1 == (int)h ? 0 : y0 + 1;
  const v3 c00 = TexelRgb(tex, w, x0, y0), c10 = TexelRgb(tex, w, x1, y0), c01 = TexelRgb(tex, w, x0, y1), c11 = TexelRgb(tex, w, x1, y1);
  const v3 a = c00 + (c10 - c00) * tx, b = c01 + (c11 - c01) * tx;
  return a + (b - a) * ty;
}

__device__ __forceinline__ v3 SchlickF(v3 F0, float c) {
  const float m = 1.0f - saturate(c);
  const float m2 = m * m;
  const float m5 = m2 * m2 * m;
  return F0 + (mk3(1.0f) - F0) * m5;
}

__device__ __forceinline__ float SmithG1(v3 w, float alpha) {   // w in the local frame, z = normal
  const float t2 = (w.x * w.x + w.y * w.y) / (w.z * w.z);
  return 2.0f / (1.0f + sqrtf(1.0f + alpha * alpha * t2));
}

// One bounce of the extended material. Returns false when the path ends (sampled direction below the surface).
// hitNormal: interpolated, flipped towards the viewer, not normalised (as ClosestHit leaves it).
__device__ __forceinline__ bool SampleGgxMaterial(v3 base, float roughness, float metallic, v3 hitNormal, v3 V, Rng& rng,
                                                  v3& transmission, v3& newDir) {
  const v3 n = normalize(hitNormal);
  const v3 side = fabsf(n.x) < 0.999f ? mk3(1.0f, 0.0f, 0.0f) : mk3(0.0f, 0.0f, 1.0f);
  const v3 tz = normalize(cross(-n, -side));   // GetCoordinateFrame (Common.hlsl:150-157) on the unit normal
  const v3 tx = cross(tz, n);
  const float alpha = fmaxf(roughness * roughness, 1e-3f);
  const v3 F0 = mk3(0.04f) + (base - mk3(0.04f)) * metallic;
  const v3 diffuse = base * (1.0f - metallic);
  const float NdotV = fmaxf(dot(n, V), 1e-4f);
  const float lumF = GetLuminance(SchlickF(F0, NdotV));
  const float lumD = GetLuminance(diffuse) * (1.0f - lumF);
  const float pSpec = clampf(lumF / (lumF + lumD), 0.1f, 0.9f);
  if (GetRand01(rng) < pSpec) {
    const float u1 = GetRand01(rng);
    const float u2 = GetRand01(rng);
    const v3 Ve = mk3(dot(tx, V), dot(tz, V), NdotV);
    const v3 Vh = normalize(mk3(alpha * Ve.x, alpha * Ve.y, Ve.z));
    const float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    const v3 T1 = lensq > 0.0f ? mk3(-Vh.y, Vh.x, 0.0f) / sqrtf(lensq) : mk3(1.0f, 0.0f, 0.0f);
    const v3 T2 = cross(Vh, T1);
    const float r = sqrtf(u1);
    const float phi = PT_TWO_PI * u2;
    const float t1 = r * cosf(phi);
    float t2 = r * sinf(phi);
    const float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * sqrtf(fmaxf(0.0f, 1.0f - t1 * t1)) + s * t2;
    const v3 Nh = t1 * T1 + t2 * T2 + sqrtf(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
    const v3 Ne = normalize(mk3(alpha * Nh.x, alpha * Nh.y, fmaxf(0.0f, Nh.z)));
    const float VdotH = dot(Ve, Ne);
    const v3 L = 2.0f * VdotH * Ne - Ve;
    if (!(L.z > 0.0f)) return false;
    const v3 weight = SchlickF(F0, VdotH) * SmithG1(L, alpha);
    transmission = transmission * weight;
    transmission = transmission / pSpec;
    newDir = tx * L.x + tz * L.y + n * L.z;
  } else {
    v2 u;
    u.x = GetRand01(rng);
    u.y = GetRand01(rng);
    const v3 weight = diffuse * (mk3(1.0f) - SchlickF(F0, NdotV));
    transmission = transmission * weight;
    transmission = transmission / (1.0f - pSpec);
    newDir = GetCosineWeightedHemisphereDirection(u, n);
  }
  return true;
}

}  // namespace ptd
