// sky_device.h — device restatement of the reference atmosphere: resources/shaders/sky/Common.hlsl (cited as
// sky/Common.hlsl:line), sky/compute_transmittance_lut.hlsl and raytracing/SampleSky.hlsl. Two instantiations of
// IntegrateScatteredLuminance are reachable from the path-tracing pipeline and both are written out here:
//   OPTICAL_DEPTH_ONLY / 40 fixed steps  -> transmittance LUT texel  (Sky.cpp:62-63,127-148)
//   full / ground=false / 4..14 variable steps / Mie+Rayleigh phase -> sky radiance on a miss (SampleSky.hlsl:7-47)
// The LUT is RGBA16F (Sky.cpp:73), sampled bilinear + clamp with float weights (D3D sampler precision is
// hardware-defined; see DESIGN.md "adopted definitions").
#pragma once
#include "device_math.h"
#include "../../include/pt_api.h"

namespace ptd {

#define PT_TLUT_W 256
#define PT_TLUT_H 64
#define PT_PLANET_RADIUS_OFFSET 10.0f   // sky/Common.hlsl:153

struct Atmo {  // sky/Common.hlsl:7-43, same 112-byte layout as PtAtmosphere
  v3 RayleighScattering; float RayleighDensityExpScale;
  v3 AbsorptionExtinction; float BottomRadius;
  v3 GroundAlbedo; float TopRadius;
  v3 MieScattering; float MieDensityExpScale;
  v3 MieExtinction; float MiePhaseG;
  v3 MieAbsorption; float AbsorptionDensity0LayerWidth;
  float AbsorptionDensity0ConstantTerm, AbsorptionDensity0LinearTerm;
  float AbsorptionDensity1ConstantTerm, AbsorptionDensity1LinearTerm;
};
static_assert(sizeof(Atmo) == sizeof(PtAtmosphere), "atmosphere layout");

// The per-miss march (IntegrateSkyLuminance) takes exp and the divisions that come AFTER the sample heights are formed
// from the hardware unit: v_exp_f32 on x*log2(e) and v_rcp_f32 (1 ulp each) instead of OCML's expf and the IEEE division
// sequence — about a quarter of the march's instructions. Everything upstream of a height (ray/sphere hits, sample
// positions, |P|) keeps the exact forms: at |P| = 6.36e6 m one fp32 ulp is half a metre, so a different rounding there
// moves a density by up to 4e-4, while these change results by ~1e-7 (test_sky_function_matches_oracle: median < 1e-5
// of the brightest channel). The transmittance-LUT build keeps expf (its halfs must match the oracle's).
// -DPT_SKY_LIBM_FORMS=1 builds the exact forms back in.
#ifndef PT_SKY_LIBM_FORMS
#define PT_SKY_LIBM_FORMS 0
#endif
template <bool FAST>
__device__ __forceinline__ float sky_exp(float x) {
  return (FAST && !PT_SKY_LIBM_FORMS) ? __builtin_amdgcn_exp2f(x * 1.44269504088896340736f) : expf(x);
}
template <bool FAST>
__device__ __forceinline__ float sky_sqrt(float x) {
  return (FAST && !PT_SKY_LIBM_FORMS) ? __builtin_amdgcn_sqrtf(x) : sqrtf(x);
}
template <bool FAST>
__device__ __forceinline__ float sky_div(float a, float b) {
  return (FAST && !PT_SKY_LIBM_FORMS) ? a * __builtin_amdgcn_rcpf(b) : a / b;
}

// raySphereIntersectNearest — sky/Common.hlsl:75-101
__device__ __forceinline__ float raySphereIntersectNearest(v3 r0, v3 rd, v3 s0, float sR) {
  float a = dot(rd, rd);
  v3 s0_r0 = r0 - s0;
  float b = 2.0f * dot(rd, s0_r0);
  float c = dot(s0_r0, s0_r0) - (sR * sR);
  float delta = b * b - 4.0f * a * c;
  if (delta < 0.0f || a == 0.0f) return -1.0f;
  float sq = sqrtf(delta);
  float sol0 = (-b - sq) / (2.0f * a);
  float sol1 = (-b + sq) / (2.0f * a);
  if (sol0 < 0.0f && sol1 < 0.0f) return -1.0f;
  if (sol0 < 0.0f) return fmaxf(0.0f, sol1);
  else if (sol1 < 0.0f) return fmaxf(0.0f, sol0);
  return fmaxf(0.0f, fminf(sol0, sol1));
}

// "raySphereIntersectNearest(...) >= 0" as a predicate (the march's earth-shadow test, sky/Common.hlsl:519-521, only asks
// whether there is a hit): with a = |rd|^2 > 0 the sign of each root is the sign of its numerator, so the two divisions
// by 2a are not needed. Same delta and square root as the function above.
__device__ __forceinline__ bool raySphereHits(v3 r0, v3 rd, v3 s0, float sR) {
  float a = dot(rd, rd);
  v3 s0_r0 = r0 - s0;
  float b = 2.0f * dot(rd, s0_r0);
  float c = dot(s0_r0, s0_r0) - (sR * sR);
  float delta = b * b - 4.0f * a * c;
  if (delta < 0.0f || a == 0.0f) return false;
  float sq = sky_sqrt<true>(delta);
  return !((-b - sq) < 0.0f && (-b + sq) < 0.0f);
}

// LutTransmittanceParamsToUv — sky/Common.hlsl:103-117
template <bool FAST>
__device__ __forceinline__ v2 LutTransmittanceParamsToUv(float bottomR, float topR, float viewHeight, float mu) {
  float H = sqrtf(fmaxf(0.0f, topR * topR - bottomR * bottomR));
  float rho = sky_sqrt<FAST>(fmaxf(0.0f, viewHeight * viewHeight - bottomR * bottomR));
  float discriminant = viewHeight * viewHeight * (mu * mu - 1.0f) + topR * topR;
  float d = fmaxf(0.0f, (-viewHeight * mu + sky_sqrt<FAST>(discriminant)));
  float d_min = topR - viewHeight;
  float d_max = rho + H;
  v2 uv;
  uv.x = sky_div<FAST>(d - d_min, d_max - d_min);
  uv.y = sky_div<FAST>(rho, H);
  return uv;
}

// UvToLutTransmittanceParams — sky/Common.hlsl:273-288
__device__ __forceinline__ void UvToLutTransmittanceParams(const Atmo& A, float& viewHeight, float& mu, v2 uv) {
  float x_mu = uv.x, x_r = uv.y;
  float H = sqrtf(A.TopRadius * A.TopRadius - A.BottomRadius * A.BottomRadius);
  float rho = H * x_r;
  viewHeight = sqrtf(rho * rho + A.BottomRadius * A.BottomRadius);
  float d_min = A.TopRadius - viewHeight;
  float d_max = rho + H;
  float d = d_min + x_mu * (d_max - d_min);
  mu = d == 0.0f ? 1.0f : (H * H - rho * rho - d * d) / (2.0f * viewHeight * d);
  mu = clampf(mu, -1.0f, 1.0f);
}

struct Medium { v3 scattering, extinction, scatteringMie, scatteringRay; };
// sampleMediumRGB — sky/Common.hlsl:191-221
template <bool FAST>
__device__ __forceinline__ Medium sampleMediumRGB(v3 P, const Atmo& A) {
  const float viewHeight = length(P) - A.BottomRadius;
  const float densityMie = sky_exp<FAST>(A.MieDensityExpScale * viewHeight);
  const float densityRay = sky_exp<FAST>(A.RayleighDensityExpScale * viewHeight);
  const float densityOzo = saturate(viewHeight < A.AbsorptionDensity0LayerWidth
                                        ? A.AbsorptionDensity0LinearTerm * viewHeight + A.AbsorptionDensity0ConstantTerm
                                        : A.AbsorptionDensity1LinearTerm * viewHeight + A.AbsorptionDensity1ConstantTerm);
  Medium s;
  v3 scatteringMie = densityMie * A.MieScattering;
  v3 extinctionMie = densityMie * A.MieExtinction;
  v3 scatteringRay = densityRay * A.RayleighScattering;
  v3 extinctionRay = scatteringRay + mk3(0.0f);
  v3 scatteringOzo = mk3(0.0f);
  v3 absorptionOzo = densityOzo * A.AbsorptionExtinction;
  v3 extinctionOzo = scatteringOzo + absorptionOzo;
  s.scatteringMie = scatteringMie;
  s.scatteringRay = scatteringRay;
  s.scattering = scatteringMie + scatteringRay + scatteringOzo;
  s.extinction = extinctionMie + extinctionRay + extinctionOzo;
  return s;
}

// hgPhase == CornetteShanksMiePhaseFunction — sky/Common.hlsl:246-255
__device__ __forceinline__ float hgPhase(float g, float cosTheta) {
  float k = 3.0f / (8.0f * PT_PI) * (1.0f - g * g) / (2.0f + g * g);
  const float x = 1.0f + g * g - 2.0f * g * -cosTheta;
  // pow(x, 1.5) = x * sqrt(x) (x >= (1 - |g|)^2 >= 0; a negative x gives NaN either way): two roundings instead of OCML's powf
  const float p15 = PT_SKY_LIBM_FORMS ? hlsl_pow(x, 1.5f) : x * sqrtf(x);
  return k * (1.0f + cosTheta * cosTheta) / p15;
}
// RayleighPhase — sky/Common.hlsl:267-271
__device__ __forceinline__ float RayleighPhase(float cosTheta) {
  float factor = 3.0f / (16.0f * PT_PI);
  return factor * (1.0f + cosTheta * cosTheta);
}

// MoveToTopAtmosphere — sky/Common.hlsl:320-339
__device__ __forceinline__ bool MoveToTopAtmosphere(v3& P, v3 viewDir, float topR) {
  float viewHeight = length(P);
  if (viewHeight > topR) {
    float tTop = raySphereIntersectNearest(P, viewDir, mk3(0.0f), topR);
    if (tTop >= 0.0f) {
      v3 UpVector = P / viewHeight;
      v3 UpOffset = UpVector * -PT_PLANET_RADIUS_OFFSET;
      P = P + viewDir * tTop + UpOffset;
    } else {
      return false;
    }
  }
  return true;
}

__device__ __forceinline__ float half_bits_to_float(uint32_t h) {
  _Float16 f;
  unsigned short s = (unsigned short)h;
  __builtin_memcpy(&f, &s, 2);
  return (float)f;
}
__device__ __forceinline__ uint32_t float_to_half_bits(float x) {
  _Float16 f = (_Float16)x;  // v_cvt_f16_f32, round to nearest even, subnormals kept
  unsigned short s;
  __builtin_memcpy(&s, &f, 2);
  return (uint32_t)s;
}

// SampleLevel(linearClamp, uv, 0).rgb of the RGBA16F transmittance LUT — sky/Common.hlsl:510-511
__device__ __forceinline__ v3 SampleTransmittanceLut(const uint2* __restrict__ lut, v2 uv) {
  float x = clampf(uv.x * (float)PT_TLUT_W - 0.5f, 0.0f, (float)(PT_TLUT_W - 1));
  float y = clampf(uv.y * (float)PT_TLUT_H - 0.5f, 0.0f, (float)(PT_TLUT_H - 1));
  int x0 = (int)floorf(x), y0 = (int)floorf(y);
  float fx = x - (float)x0, fy = y - (float)y0;
  int x1 = x0 + 1 < PT_TLUT_W ? x0 + 1 : PT_TLUT_W - 1;
  int y1 = y0 + 1 < PT_TLUT_H ? y0 + 1 : PT_TLUT_H - 1;
  uint2 t00 = lut[y0 * PT_TLUT_W + x0], t10 = lut[y0 * PT_TLUT_W + x1];
  uint2 t01 = lut[y1 * PT_TLUT_W + x0], t11 = lut[y1 * PT_TLUT_W + x1];
  float gx = 1.0f - fx, gy = 1.0f - fy;
  v3 r;
  {
    float top = half_bits_to_float(t00.x & 0xffffu) * gx + half_bits_to_float(t10.x & 0xffffu) * fx;
    float bot = half_bits_to_float(t01.x & 0xffffu) * gx + half_bits_to_float(t11.x & 0xffffu) * fx;
    r.x = top * gy + bot * fy;
  }
  {
    float top = half_bits_to_float(t00.x >> 16) * gx + half_bits_to_float(t10.x >> 16) * fx;
    float bot = half_bits_to_float(t01.x >> 16) * gx + half_bits_to_float(t11.x >> 16) * fx;
    r.y = top * gy + bot * fy;
  }
  {
    float top = half_bits_to_float(t00.y & 0xffffu) * gx + half_bits_to_float(t10.y & 0xffffu) * fx;
    float bot = half_bits_to_float(t01.y & 0xffffu) * gx + half_bits_to_float(t11.y & 0xffffu) * fx;
    r.z = top * gy + bot * fy;
  }
  return r;
}

// Sphere hits that bound the march — sky/Common.hlsl:388-411,428. Returns false when the ray leaves without
// touching planet or atmosphere (the function returns a zero result there).
__device__ __forceinline__ bool sky_tmax(v3 P, v3 dir, const Atmo& A, float& tMax) {
  v3 earthO = mk3(0.0f);
  float tBottom = raySphereIntersectNearest(P, dir, earthO, A.BottomRadius);
  float tTop = raySphereIntersectNearest(P, dir, earthO, A.TopRadius);
  tMax = 0.0f;
  if (tBottom < 0.0f) {
    if (tTop < 0.0f) return false;
    tMax = tTop;
  } else {
    if (tTop > 0.0f) tMax = fminf(tTop, tBottom);
  }
  tMax = fminf(tMax, 9000000.0f);
  return true;
}

// IntegrateScatteredLuminance, OPTICAL_DEPTH_ONLY instantiation — sky/Common.hlsl:368-589 with
// VariableSampleCount=false, SampleCountIni=40 (compute_transmittance_lut.hlsl:30-56).
__device__ __forceinline__ v3 IntegrateOpticalDepth(v3 P0, v3 dir, const Atmo& A) {
  v3 OpticalDepth = mk3(0.0f);
  float tMax;
  if (!sky_tmax(P0, dir, A, tMax)) return OpticalDepth;
  const float SampleCount = 40.0f;
  float t = 0.0f;
  const float SampleSegmentT = 0.3f;
  for (float s = 0.0f; s < SampleCount; s += 1.0f) {
    float NewT = tMax * (s + SampleSegmentT) / SampleCount;
    float dt = NewT - t;
    t = NewT;
    v3 P = P0 + t * dir;
    Medium medium = sampleMediumRGB<false>(P, A);
    v3 SampleOpticalDepth = medium.extinction * dt;
    OpticalDepth = OpticalDepth + SampleOpticalDepth;
  }
  return OpticalDepth;
}

// compute_transmittance_lut.hlsl main — :15-62; returns the RGBA16F texel packed in a uint2.
__device__ __forceinline__ uint2 TransmittanceLutTexel(const Atmo& A, uint32_t ix, uint32_t iy) {
  v2 uv;
  uv.x = ((float)ix + 0.5f) / (float)PT_TLUT_W;
  uv.y = ((float)iy + 0.5f) / (float)PT_TLUT_H;
  float viewHeight, mu;
  UvToLutTransmittanceParams(A, viewHeight, mu, uv);
  v3 pos = mk3(0.0f, viewHeight, 0.0f);
  v3 dir = mk3(0.0f, mu, sqrtf(1.0f - mu * mu));
  v3 od = IntegrateOpticalDepth(pos, dir, A);
  v3 tr = exp3(-od);
  uint2 r;
  r.x = float_to_half_bits(tr.x) | (float_to_half_bits(tr.y) << 16);
  r.y = float_to_half_bits(tr.z) | (float_to_half_bits(1.0f) << 16);
  return r;
}

// IntegrateScatteredLuminance, full instantiation as called from SampleSkyLuminance — sky/Common.hlsl:368-589 with
// ground=false, VariableSampleCount=true, MieRayPhase=true, DepthBufferValue=-1 (SampleSky.hlsl:21-41).
__device__ __forceinline__ v3 IntegrateSkyLuminance(v3 P0, v3 dir, v3 sunDir, const Atmo& A, v2 minMaxSPP, v3 sunIlluminance,
                                                    const uint2* __restrict__ lut) {
  v3 L = mk3(0.0f);
  float tMax;
  if (!sky_tmax(P0, dir, A, tMax)) return L;
  float SampleCount = lerpf(minMaxSPP.x, minMaxSPP.y, saturate(tMax * 0.01f));
  float SampleCountFloor = floorf(SampleCount);
  float tMaxFloor = tMax * SampleCountFloor / SampleCount;

  float cosTheta = dot(sunDir, dir);
  float MiePhaseValue = hgPhase(A.MiePhaseG, -cosTheta);
  float RayleighPhaseValue = RayleighPhase(cosTheta);
  v3 globalL = sunIlluminance;
  v3 throughput = mk3(1.0f);
  const float SampleSegmentT = 0.3f;
  v3 earthO = mk3(0.0f);
  float t1_lin = 0.0f / SampleCountFloor;   // (s + 1) / SampleCountFloor of one step is s / SampleCountFloor of the next: one division per step
  for (float s = 0.0f; s < SampleCount; s += 1.0f) {
    float t0 = t1_lin;
    float t1 = (s + 1.0f) / SampleCountFloor;
    t1_lin = t1;
    t0 = t0 * t0;
    t1 = t1 * t1;
    t0 = tMaxFloor * t0;
    if (t1 > 1.0f) t1 = tMax;
    else t1 = tMaxFloor * t1;
    float t = t0 + (t1 - t0) * SampleSegmentT;
    float dt = t1 - t0;
    v3 P = P0 + t * dir;

    Medium medium = sampleMediumRGB<true>(P, A);
    v3 SampleOpticalDepth = medium.extinction * dt;
    v3 SampleTransmittance = mk3(sky_exp<true>(-SampleOpticalDepth.x), sky_exp<true>(-SampleOpticalDepth.y), sky_exp<true>(-SampleOpticalDepth.z));

    float pHeight = length(P);
    // the unit up vector feeds the LUT coordinate and the 10 m lift of the shadow sphere, not a height: reciprocal form
    const float rcpHeight = sky_div<true>(1.0f, pHeight);
    v3 UpVector = PT_SKY_LIBM_FORMS ? P / pHeight : P * rcpHeight;
    float SunZenithCosAngle = dot(sunDir, UpVector);
    v2 uv = LutTransmittanceParamsToUv<true>(A.BottomRadius, A.TopRadius, pHeight, SunZenithCosAngle);
    v3 TransmittanceToSun = SampleTransmittanceLut(lut, uv);

    v3 PhaseTimesScattering = medium.scatteringMie * MiePhaseValue + medium.scatteringRay * RayleighPhaseValue;

    float earthShadow = raySphereHits(P, sunDir, earthO + PT_PLANET_RADIUS_OFFSET * UpVector, A.BottomRadius) ? 0.0f : 1.0f;
    v3 multiScatteredLuminance = mk3(0.0f);
    float shadow = 1.0f;
    v3 S = globalL * (earthShadow * shadow * TransmittanceToSun * PhaseTimesScattering + multiScatteredLuminance * medium.scattering);
    const v3 dS = S - S * SampleTransmittance;
    v3 Sint = mk3(sky_div<true>(dS.x, medium.extinction.x), sky_div<true>(dS.y, medium.extinction.y), sky_div<true>(dS.z, medium.extinction.z));
    L = L + throughput * Sint;
    throughput = throughput * SampleTransmittance;
  }
  return L;
}

// SampleSkyLuminance — SampleSky.hlsl:7-47
__device__ __forceinline__ v3 SampleSkyLuminance(v3 viewPos, v3 viewDir, const PtRtConsts& c, const uint2* __restrict__ lut) {
  if (!c.mySampleSky) return ld3(c.mySkyFallbackEmission);
  Atmo A;
  __builtin_memcpy(&A, &c.mySkyConsts.myAtmosphere, sizeof(A));
  v3 worldPos = viewPos + mk3(0.0f, A.BottomRadius, 0.0f);
  v3 sky = mk3(0.0f);
  if (MoveToTopAtmosphere(worldPos, viewDir, A.TopRadius)) {
    v2 mm; mm.x = c.mySkyConsts.myRayMarchMinMaxSPP[0]; mm.y = c.mySkyConsts.myRayMarchMinMaxSPP[1];
    sky = IntegrateSkyLuminance(worldPos, viewDir, ld3(c.mySkyConsts.mySunDirection), A, mm, ld3(c.mySkyConsts.mySunIlluminance), lut);
  }
  return sky;
}

// ---------------------------------------------------------------------------------------------------------------
// Sky-view LUT (extension, PT_OPT_SKY_MODE = 1): the marched sky radiance of SampleSkyLuminance tabulated over view
// direction for ONE reference position (the camera), so a miss costs a bilinear fetch instead of a 14-step march.
// Parametrisation after the reference's raster-path LUT (SkyViewLutParamsToUv / UvToSkyViewLutParams,
// sky/Common.hlsl:119-149,290-318; compute_skyView_lut.hlsl:27-91; render_sky.hlsl:22-54): u = sqrt of the azimuth
// term relative to the sun, v = non-linear zenith angle, split at the horizon. Differences, all deliberate: texel i
// maps to unit coordinate i/(N-1) (the reference's half-texel helpers are not mutually inverse); the sun is placed in
// the LUT's local frame (the reference marches with the world-space sun, which is only right for suns in the +X
// azimuth); fp32 texels. The origin of a miss ray is replaced by the camera position: scene extents (tens of
// metres) are negligible against the atmosphere's scale heights (1.2 / 8 km).
// Known limit: the marched function is discontinuous in elevation wherever one of its 14 sample points crosses
// 10 m of altitude (the earth-shadow sphere is lifted by PLANET_RADIUS_OFFSET, sky/Common.hlsl:153,519-521, so a
// sample below 10 m counts as shadowed). For a camera near the ground that happens at ~0.3 deg and ~0 deg of
// elevation; the table smooths each step over one row (<= 0.1 deg there). Everywhere else it tracks the march to
// < 0.1 % (tests/test_gpu_parity.py::test_sky_view_lut_tracks_marched_sky).
#define PT_SKYVIEW_W 256
#define PT_SKYVIEW_H 256

struct SkyViewFrame {
  float viewHeight;      // |camera + (0,R,0)|, clamped above the ground
  v3 up;                 // local zenith
  float beta;            // angle between horizon and nadir-side: acos(Vhorizon / h)
  float zenithHorizonAngle;
  float sunZenithCos;
  v3 sunFwd;             // unit horizontal direction towards the sun (arbitrary when the sun is at the zenith)
  bool sunHasAzimuth;
};

__device__ __forceinline__ SkyViewFrame MakeSkyViewFrame(const PtRtConsts& c) {
  SkyViewFrame f;
  const float R = c.mySkyConsts.myAtmosphere.BottomRadius;
  v3 worldPos = ld3(c.myCameraPos) + mk3(0.0f, R, 0.0f);
  float h = length(worldPos);
  if (!(h > R + 1.0f)) { worldPos = mk3(0.0f, R + 1.0f, 0.0f); h = R + 1.0f; }
  f.viewHeight = h;
  f.up = worldPos / h;
  const float Vhorizon = sqrtf(fmaxf(0.0f, (h - R) * (h + R)));
  f.beta = acosf(clampf(Vhorizon / h, -1.0f, 1.0f));
  f.zenithHorizonAngle = PT_PI - f.beta;
  const v3 sun = ld3(c.mySkyConsts.mySunDirection);
  f.sunZenithCos = dot(f.up, sun);
  const v3 horiz = sun - f.up * f.sunZenithCos;
  const float hl = length(horiz);
  f.sunHasAzimuth = hl > 1e-4f;
  f.sunFwd = f.sunHasAzimuth ? horiz / hl : mk3(1.0f, 0.0f, 0.0f);
  return f;
}

// texel -> radiance (one LUT entry). Local frame: y = up, x = towards the sun's azimuth.
__device__ __forceinline__ v3 SkyViewLutTexel(const PtRtConsts& c, const uint2* __restrict__ tlut, uint32_t ix, uint32_t iy) {
  const SkyViewFrame f = MakeSkyViewFrame(c);
  Atmo A;
  __builtin_memcpy(&A, &c.mySkyConsts.myAtmosphere, sizeof(A));
  // two charts of PT_SKYVIEW_H/2 rows each: rows [0, H/2) zenith -> horizon, rows [H/2, H) horizon -> nadir; the
  // first and last row of a chart sit exactly on its end points
  const int half = PT_SKYVIEW_H / 2;
  const float u = (float)ix / (float)(PT_SKYVIEW_W - 1);
  float viewZenithCos;
  if ((int)iy < half) {
    float coord = (float)iy / (float)(half - 1);
    coord = 1.0f - coord;
    coord *= coord;
    coord = 1.0f - coord;
    viewZenithCos = cosf(f.zenithHorizonAngle * coord);
  } else {
    float coord = (float)((int)iy - half) / (float)(half - 1);
    coord *= coord;
    viewZenithCos = cosf(f.zenithHorizonAngle + f.beta * coord);
  }
  const float lightViewCos = -(u * u * 2.0f - 1.0f);
  const float vs = sqrtf(fmaxf(0.0f, 1.0f - viewZenithCos * viewZenithCos));
  const v3 dir = mk3(vs * lightViewCos, viewZenithCos, vs * sqrtf(fmaxf(0.0f, 1.0f - lightViewCos * lightViewCos)));
  const float ss = sqrtf(fmaxf(0.0f, 1.0f - f.sunZenithCos * f.sunZenithCos));
  const v3 sunLocal = mk3(ss, f.sunZenithCos, 0.0f);
  v3 P = mk3(0.0f, f.viewHeight, 0.0f);
  v3 L = mk3(0.0f);
  if (MoveToTopAtmosphere(P, dir, A.TopRadius)) {
    v2 mm; mm.x = c.mySkyConsts.myRayMarchMinMaxSPP[0]; mm.y = c.mySkyConsts.myRayMarchMinMaxSPP[1];
    L = IntegrateSkyLuminance(P, dir, sunLocal, A, mm, ld3(c.mySkyConsts.mySunIlluminance), tlut);
  }
  return L;
}

// direction -> radiance by bilinear lookup (the stand-in for SampleSkyLuminance in sky mode 1)
__device__ __forceinline__ v3 SampleSkyViewLut(const float4* __restrict__ lut, const SkyViewFrame& f, float bottomR, v3 viewDir) {
  const float vzc = clampf(dot(viewDir, f.up), -1.0f, 1.0f);
  float lvc = 1.0f;
  if (f.sunHasAzimuth) {
    const v3 hv = viewDir - f.up * vzc;
    const float hl = length(hv);
    if (hl > 1e-6f) lvc = clampf(dot(hv, f.sunFwd) / hl, -1.0f, 1.0f);
  }
  const float ang = acosf(vzc);
  const int half = PT_SKYVIEW_H / 2;
  float coord;
  int row0;
  if (ang <= f.zenithHorizonAngle) {      // above the horizon (does not intersect the ground): upper chart
    coord = 1.0f - sqrtf(fmaxf(0.0f, 1.0f - ang / f.zenithHorizonAngle));
    row0 = 0;
  } else {
    coord = sqrtf(clampf((ang - f.zenithHorizonAngle) / f.beta, 0.0f, 1.0f));
    row0 = half;
  }
  const float u = sqrtf(clampf(-lvc * 0.5f + 0.5f, 0.0f, 1.0f));
  const float x = clampf(u * (float)(PT_SKYVIEW_W - 1), 0.0f, (float)(PT_SKYVIEW_W - 1));
  const float y = clampf(coord * (float)(half - 1), 0.0f, (float)(half - 1));   // never interpolates across the charts
  const int x0 = (int)floorf(x), yy = (int)floorf(y);
  const float fx = x - (float)x0, fy = y - (float)yy;
  const int x1 = x0 + 1 < PT_SKYVIEW_W ? x0 + 1 : PT_SKYVIEW_W - 1;
  const int y0 = row0 + yy, y1 = row0 + (yy + 1 < half ? yy + 1 : half - 1);
  const float4 t00 = lut[y0 * PT_SKYVIEW_W + x0], t10 = lut[y0 * PT_SKYVIEW_W + x1];
  const float4 t01 = lut[y1 * PT_SKYVIEW_W + x0], t11 = lut[y1 * PT_SKYVIEW_W + x1];
  const float gx = 1.0f - fx, gy = 1.0f - fy;
  (void)bottomR;
  return mk3((t00.x * gx + t10.x * fx) * gy + (t01.x * gx + t11.x * fx) * fy,
             (t00.y * gx + t10.y * fx) * gy + (t01.y * gx + t11.y * fx) * fy,
             (t00.z * gx + t10.z * fx) * gy + (t01.z * gx + t11.z * fx) * fy);
}

// SampleSkyLuminance with the march replaced by the table. The march's domain tests are kept, evaluated at the real
// ray origin: in fp32, viewPos.y + 6360000 rounds to the planet radius for y < 0.25 m, raySphereIntersectNearest then
// returns tBottom = 0, tMax = 0 and the sky is black (SampleSky.hlsl:14, sky/Common.hlsl:388-411) — floor-level path
// vertices see no sky in the reference, and neither do they here.
__device__ __forceinline__ v3 SampleSkyLuminanceViewLut(v3 viewPos, v3 viewDir, const PtRtConsts& c, const float4* __restrict__ skyview,
                                                        const SkyViewFrame& f) {
  if (!c.mySampleSky) return ld3(c.mySkyFallbackEmission);
  Atmo A;
  __builtin_memcpy(&A, &c.mySkyConsts.myAtmosphere, sizeof(A));
  v3 worldPos = viewPos + mk3(0.0f, A.BottomRadius, 0.0f);
  if (!MoveToTopAtmosphere(worldPos, viewDir, A.TopRadius)) return mk3(0.0f);
  float tMax;
  if (!sky_tmax(worldPos, viewDir, A, tMax) || !(tMax > 0.0f)) return mk3(0.0f);
  return SampleSkyViewLut(skyview, f, A.BottomRadius, viewDir);
}

}  // namespace ptd
