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

// ---- the hardware transcendental unit -----------------------------------------------------------------------------------
// HLSL sin / cos / pow compile to the GPU's transcendental instructions on the reference's target too; here the hot
// shading path uses CDNA's directly instead of OCML's software routines (range reduction with a Payne-Hanek slow path,
// extended-precision pow: some 50-120 instructions each). Measured exhaustively over all 2^23 values GetRand01 can
// return (tools/hwmath/hwmath.hip): v_sin_f32 / v_cos_f32 abs error <= 1.26e-7; exp2(y*log2(x)) rel error <= 7.9e-6
// for results >= 1e-30. tests: test_shading_functions_match_oracle_and_fixture (rtol 2e-5 .. 5e-5 vs the libm oracle).
// PT_SHADE_LIBM_FORMS=1 builds the OCML forms back in (bisecting).
#ifndef PT_SHADE_LIBM_FORMS
#define PT_SHADE_LIBM_FORMS 0
#endif
// sin(2 pi u), cos(2 pi u): v_sin_f32 / v_cos_f32 take their argument in revolutions
__device__ __forceinline__ float sin_2pi(float u) { return PT_SHADE_LIBM_FORMS ? sinf(6.28318530717958647692f * u) : __builtin_amdgcn_sinf(u); }
__device__ __forceinline__ float cos_2pi(float u) { return PT_SHADE_LIBM_FORMS ? cosf(6.28318530717958647692f * u) : __builtin_amdgcn_cosf(u); }
// hlsl_pow with the same edge cases, core = v_exp_f32(y * v_log_f32(x))
__device__ __forceinline__ float hlsl_pow_hw(float x, float y) {
  if (PT_SHADE_LIBM_FORMS) return hlsl_pow(x, y);
  float r = __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));   // x < 0: NaN; x == 0, y > 0: exp2(-inf) = 0; y < 0: inf
  r = y == 0.0f ? 1.0f : r;                                          // 0 * -inf
  return x < 0.0f ? __builtin_nanf("") : r;
}
// hlsl_pow(x, 5) by multiplication (4 roundings; NaN for x < 0 as the intrinsic)
__device__ __forceinline__ float hlsl_pow5(float x) {
  if (PT_SHADE_LIBM_FORMS) return hlsl_pow(x, 5.0f);
  const float x2 = x * x;
  const float r = x2 * x2 * x;
  return x < 0.0f ? __builtin_nanf("") : r;
}

// n / d and n % d for a wave-uniform divisor through one float multiply and one correction step; replaces four
// ~25-instruction 32-bit divisions per path. The float estimate is within 1 of the quotient while the quotient stays
// below 2^20 (float(n) and 1/d carry 2^-24 each); n_max is the caller's bound on n, and a divisor too small for that
// (1-pixel-wide targets, 1x1 targets with millions of samples) takes the integer division instead (wave-uniform).
struct FastDiv {
  uint32_t d;
  float inv;
  bool exact_path;
  __device__ __forceinline__ FastDiv(uint32_t div, uint32_t n_max) : d(div), inv(1.0f / (float)div), exact_path(((uint64_t)div << 20) <= (uint64_t)n_max) {}
  __device__ __forceinline__ void divmod(uint32_t n, uint32_t& q, uint32_t& r) const {
    if (exact_path) { q = n / d; r = n - q * d; return; }
    q = (uint32_t)((float)n * inv);
    r = n - q * d;
    if ((int32_t)r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
  }
};

#define PT_PI 3.14159265358979323846f
#define PT_TWO_PI 6.28318530717958647692f

}  // namespace ptd
