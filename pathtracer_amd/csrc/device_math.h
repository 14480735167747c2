// device_math.h — fp32 vector helpers and the HLSL intrinsic semantics the kernels adopt.
// All arithmetic is plain IEEE fp32, left to right, no implicit fused multiply-add (the library is built with
// -ffp-contract=off; fused ops appear only where written as __builtin_fmaf, i.e. in the conservative box test).
// dot/cross/normalize/reflect/saturate/lerp/pow follow HLSL as used by resources/shaders/raytracing/*.hlsl.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptd {

struct v3 { float x, y, z; };
struct v2 { float x, y; };

__device__ __forceinline__ v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 mk3(float s) { return mk3(s, s, s); }
__device__ __forceinline__ v3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator*(v3 a, v3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3 operator/(v3 a, v3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 operator*(float s, v3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ v3 operator/(v3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ v3 operator-(v3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ v3 cross(v3 a, v3 b) {
  return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float length(v3 v) { return sqrtf(dot(v, v)); }
__device__ __forceinline__ v3 normalize(v3 v) { return v / length(v); }
__device__ __forceinline__ v3 reflect(v3 i, v3 n) { return i - 2.0f * n * dot(i, n); }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
__device__ __forceinline__ float saturate(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
__device__ __forceinline__ float lerpf(float a, float b, float t) { return a + (b - a) * t; }
__device__ __forceinline__ v3 exp3(v3 v) { return mk3(expf(v.x), expf(v.y), expf(v.z)); }
// HLSL pow: NaN for x<0, 0 for x==0 (y>0), powf otherwise.
__device__ __forceinline__ float hlsl_pow(float x, float y) {
  if (x < 0.0f) return __builtin_nanf("");
  if (x == 0.0f) return y > 0.0f ? 0.0f : (y < 0.0f ? __builtin_inff() : 1.0f);
  return powf(x, y);
}

#define PT_PI 3.14159265358979323846f
#define PT_TWO_PI 6.28318530717958647692f

}  // namespace ptd
