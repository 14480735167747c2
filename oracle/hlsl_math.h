// ORACLE — TEST INFRASTRUCTURE ONLY. Nothing under pathtracer_amd/ may include, link or call this.
//
// Minimal float3 / HLSL-intrinsic layer for the scalar CPU restatement of the reference's HLSL.
// Every operation is plain IEEE fp32 evaluated left to right, with NO fused multiply-add
// (compile with -ffp-contract=off), so that +,-,*,/,sqrt results are bit-reproducible on the
// device side too. Transcendentals (pow/exp/sin/cos/acos) are libm fp32; the HLSL originals are
// driver-precision, so anything downstream of them is tolerance-checked, never bit-checked.
//
// HLSL semantics adopted ("parity unpinned" – the DXC/driver is not here):
//   dot(a,b)      = a.x*b.x + a.y*b.y + a.z*b.z           (left to right, no fma)
//   cross(a,b)    = (a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x)
//   length(v)     = sqrt(dot(v,v));  normalize(v) = v / length(v)   (true divide, not rsqrt)
//   reflect(i,n)  = i - 2*n*dot(i,n)
//   saturate(x)   = min(max(x,0),1) with NaN -> 0 (fmaxf/fminf return the non-NaN operand)
//   lerp(a,b,t)   = a + (b-a)*t
//   pow(x,y)      = NaN for x<0; 0 for x==0,y>0; else powf   (HLSL doc for pow)
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct uint4  { uint32_t x, y, z, w; };

static inline float3 f3(float x, float y, float z) { return float3{x, y, z}; }
static inline float3 f3(float s) { return float3{s, s, s}; }

static inline float3 operator+(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline float3 operator-(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline float3 operator*(float3 a, float3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline float3 operator/(float3 a, float3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
static inline float3 operator*(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline float3 operator*(float s, float3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline float3 operator/(float3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
static inline float3 operator-(float3 a) { return {-a.x, -a.y, -a.z}; }
static inline float3& operator+=(float3& a, float3 b) { a = a + b; return a; }
static inline float3& operator*=(float3& a, float3 b) { a = a * b; return a; }
static inline float3& operator/=(float3& a, float s) { a = a / s; return a; }

static inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float3 cross(float3 a, float3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline float length(float3 v) { return sqrtf(dot(v, v)); }
static inline float3 normalize(float3 v) { return v / length(v); }
static inline float3 reflect(float3 i, float3 n) { return i - 2.0f * n * dot(i, n); }
static inline float hmax(float a, float b) { return fmaxf(a, b); }
static inline float hmin(float a, float b) { return fminf(a, b); }
static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline float saturate(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
static inline float lerpf(float a, float b, float t) { return a + (b - a) * t; }
static inline float3 max3f(float s, float3 v) { return {fmaxf(s, v.x), fmaxf(s, v.y), fmaxf(s, v.z)}; }
static inline float3 exp3(float3 v) { return {expf(v.x), expf(v.y), expf(v.z)}; }
static inline float hlsl_pow(float x, float y) {
  if (x < 0.0f) return NAN;
  if (x == 0.0f) return y > 0.0f ? 0.0f : (y < 0.0f ? INFINITY : 1.0f);
  return powf(x, y);
}
static inline float asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

// FANCY common_types.h is absent: PI / TWO_PI adopted as float(pi), float(2pi).
static const float PI = 3.14159265358979323846f;
static const float TWO_PI = 6.28318530717958647692f;

// IEEE binary16 round-to-nearest-even, subnormals kept (the transmittance LUT is RGBA16F, Sky.cpp:73).
static inline uint16_t float_to_half(float f) {
  uint32_t x = asuint(f);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t mant = x & 0x007fffffu;
  int32_t exp = (int32_t)((x >> 23) & 0xff);
  if (exp == 0xff) return (uint16_t)(sign | 0x7c00u | (mant ? 0x200u : 0));
  int32_t e = exp - 127 + 15;
  if (e >= 31) return (uint16_t)(sign | 0x7c00u);
  if (e <= 0) {
    if (e < -10) return (uint16_t)sign;
    mant |= 0x00800000u;
    uint32_t shift = (uint32_t)(14 - e);
    uint32_t hm = mant >> shift;
    uint32_t rem = mant & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1u))) hm++;
    return (uint16_t)(sign | hm);
  }
  uint32_t hm = mant >> 13;
  uint32_t rem = mant & 0x1fffu;
  uint32_t h = sign | ((uint32_t)e << 10) | hm;
  if (rem > 0x1000u || (rem == 0x1000u && (hm & 1u))) h++;
  return (uint16_t)h;
}
static inline float half_to_float(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1f;
  uint32_t mant = h & 0x3ffu;
  if (exp == 0) {
    if (mant == 0) return asfloat(sign);
    float v = (float)mant * 5.9604644775390625e-08f;  // 2^-24
    return sign ? -v : v;
  }
  if (exp == 31) return asfloat(sign | 0x7f800000u | (mant << 13));
  return asfloat(sign | ((exp - 15 + 127) << 23) | (mant << 13));
}

}  // namespace orc
