// ORACLE — TEST INFRASTRUCTURE ONLY (see hlsl_math.h). Scalar restatement of the reference's
// atmosphere code: resources/shaders/sky/Common.hlsl (cited as sky/Common.hlsl:line),
// sky/compute_transmittance_lut.hlsl and raytracing/SampleSky.hlsl. Only the two instantiations of
// IntegrateScatteredLuminance that the path-tracing pipeline reaches are restated:
//   (A) OPTICAL_DEPTH_ONLY, fixed 40 samples          -> transmittance LUT   (Sky.cpp:62-63,127-148)
//   (B) full, ground=false, variable 4..14 samples, Mie+Rayleigh phase -> sky radiance on a miss
// MULTISCATAPPROX_ENABLED / SHADOWMAP_ENABLED / ILLUMINANCE_IS_ONE are undefined in the reference build.
//
// Texture sampling adopted (D3D12 sampler precision is hardware-defined, parity unpinned):
// RGBA16F texels (RNE conversion), bilinear with float weights, texel centres at (i+0.5)/N, clamp addressing.
#pragma once
#include "hlsl_math.h"
#include "../include/pt_api.h"

namespace orc {

enum { TRANSMITTANCE_TEXTURE_WIDTH = 256, TRANSMITTANCE_TEXTURE_HEIGHT = 64 };  // Sky.cpp:14-15

struct Atmosphere {  // sky/Common.hlsl:7-43
  float3 RayleighScattering; float RayleighDensityExpScale;
  float3 AbsorptionExtinction; float BottomRadius;
  float3 GroundAlbedo; float TopRadius;
  float3 MieScattering; float MieDensityExpScale;
  float3 MieExtinction; float MiePhaseG;
  float3 MieAbsorption; float AbsorptionDensity0LayerWidth;
  float AbsorptionDensity0ConstantTerm, AbsorptionDensity0LinearTerm;
  float AbsorptionDensity1ConstantTerm, AbsorptionDensity1LinearTerm;
};
static_assert(sizeof(Atmosphere) == 112 && sizeof(PtAtmosphere) == 112, "AtmosphereParameters is 112 bytes");

// SetupEarthAtmosphere — PathTracer/Sky.cpp:32-56 (values are metres despite the "km" comments)
static inline void SetupEarthAtmosphere(Atmosphere& p) {
  const float EarthBottomRadius = 6360000.0f;
  const float EarthTopRadius = 6460000.0f;
  const float EarthRayleighScaleHeight = 8000.0f;
  const float EarthMieScaleHeight = 1200.0f;
  p.RayleighScattering = f3(0.000005802f, 0.000013558f, 0.000033100f);
  p.RayleighDensityExpScale = -1.0f / EarthRayleighScaleHeight;
  p.AbsorptionExtinction = f3(0.000000650f, 0.000001881f, 0.0000000085f);
  p.BottomRadius = EarthBottomRadius;
  p.GroundAlbedo = f3(0.0f, 0.0f, 0.0f);
  p.TopRadius = EarthTopRadius;
  p.MieScattering = f3(0.000003996f, 0.000003996f, 0.000003996f);
  p.MieDensityExpScale = -1.0f / EarthMieScaleHeight;
  p.MieExtinction = f3(0.000004440f, 0.000004440f, 0.000004440f);
  p.MiePhaseG = 0.8f;
  p.MieAbsorption = max3f(0.0f, p.MieExtinction - p.MieScattering);
  p.AbsorptionDensity0LayerWidth = 25000.0f;
  p.AbsorptionDensity0ConstantTerm = -2.0f / 3.0f;
  p.AbsorptionDensity0LinearTerm = 1.0f / 15.0f;
  p.AbsorptionDensity1ConstantTerm = 8.0f / 3.0f;
  p.AbsorptionDensity1LinearTerm = -1.0f / 15.0f;
}

// raySphereIntersectNearest — sky/Common.hlsl:75-101
static inline float raySphereIntersectNearest(float3 r0, float3 rd, float3 s0, float sR) {
  float a = dot(rd, rd);
  float3 s0_r0 = r0 - s0;
  float b = 2.0f * dot(rd, s0_r0);
  float c = dot(s0_r0, s0_r0) - (sR * sR);
  float delta = b * b - 4.0f * a * c;
  if (delta < 0.0f || a == 0.0f) return -1.0f;
  float sol0 = (-b - sqrtf(delta)) / (2.0f * a);
  float sol1 = (-b + sqrtf(delta)) / (2.0f * a);
  if (sol0 < 0.0f && sol1 < 0.0f) return -1.0f;
  if (sol0 < 0.0f) return hmax(0.0f, sol1);
  else if (sol1 < 0.0f) return hmax(0.0f, sol0);
  return hmax(0.0f, hmin(sol0, sol1));
}

// LutTransmittanceParamsToUv — sky/Common.hlsl:103-117
static inline float2 LutTransmittanceParamsToUv(float bottomR, float topR, float viewHeight, float viewZenithCosAngle) {
  float H = sqrtf(hmax(0.0f, topR * topR - bottomR * bottomR));
  float rho = sqrtf(hmax(0.0f, viewHeight * viewHeight - bottomR * bottomR));
  float discriminant = viewHeight * viewHeight * (viewZenithCosAngle * viewZenithCosAngle - 1.0f) + topR * topR;
  float d = hmax(0.0f, (-viewHeight * viewZenithCosAngle + sqrtf(discriminant)));
  float d_min = topR - viewHeight;
  float d_max = rho + H;
  float x_mu = (d - d_min) / (d_max - d_min);
  float x_r = rho / H;
  return float2{x_mu, x_r};
}

// UvToLutTransmittanceParams — sky/Common.hlsl:273-288
static inline void UvToLutTransmittanceParams(const Atmosphere& A, float& viewHeight, float& viewZenithCosAngle, float2 uv) {
  float x_mu = uv.x;
  float x_r = uv.y;
  float H = sqrtf(A.TopRadius * A.TopRadius - A.BottomRadius * A.BottomRadius);
  float rho = H * x_r;
  viewHeight = sqrtf(rho * rho + A.BottomRadius * A.BottomRadius);
  float d_min = A.TopRadius - viewHeight;
  float d_max = rho + H;
  float d = d_min + x_mu * (d_max - d_min);
  viewZenithCosAngle = d == 0.0f ? 1.0f : (H * H - rho * rho - d * d) / (2.0f * viewHeight * d);
  viewZenithCosAngle = clampf(viewZenithCosAngle, -1.0f, 1.0f);
}

struct MediumSampleRGB {  // the members IntegrateScatteredLuminance reads (sky/Common.hlsl:168-187)
  float3 scattering, extinction, scatteringMie, scatteringRay;
};
// sampleMediumRGB — sky/Common.hlsl:191-221
static inline MediumSampleRGB sampleMediumRGB(float3 aPosWorld, const Atmosphere& A) {
  const float viewHeight = length(aPosWorld) - A.BottomRadius;
  const float densityMie = expf(A.MieDensityExpScale * viewHeight);
  const float densityRay = expf(A.RayleighDensityExpScale * viewHeight);
  const float densityOzo = saturate(viewHeight < A.AbsorptionDensity0LayerWidth
                                        ? A.AbsorptionDensity0LinearTerm * viewHeight + A.AbsorptionDensity0ConstantTerm
                                        : A.AbsorptionDensity1LinearTerm * viewHeight + A.AbsorptionDensity1ConstantTerm);
  MediumSampleRGB s;
  float3 scatteringMie = densityMie * A.MieScattering;
  float3 extinctionMie = densityMie * A.MieExtinction;
  float3 scatteringRay = densityRay * A.RayleighScattering;
  float3 absorptionRay = f3(0.0f);
  float3 extinctionRay = scatteringRay + absorptionRay;
  float3 scatteringOzo = f3(0.0f);
  float3 absorptionOzo = densityOzo * A.AbsorptionExtinction;
  float3 extinctionOzo = scatteringOzo + absorptionOzo;
  s.scatteringMie = scatteringMie;
  s.scatteringRay = scatteringRay;
  s.scattering = scatteringMie + scatteringRay + scatteringOzo;
  s.extinction = extinctionMie + extinctionRay + extinctionOzo;
  return s;
}

// CornetteShanksMiePhaseFunction / hgPhase — sky/Common.hlsl:246-255 (hgPhase forwards to Cornette-Shanks)
static inline float hgPhase(float g, float cosTheta) {
  float k = 3.0f / (8.0f * PI) * (1.0f - g * g) / (2.0f + g * g);
  return k * (1.0f + cosTheta * cosTheta) / hlsl_pow(1.0f + g * g - 2.0f * g * -cosTheta, 1.5f);
}
// RayleighPhase — sky/Common.hlsl:267-271
static inline float RayleighPhase(float cosTheta) {
  float factor = 3.0f / (16.0f * PI);
  return factor * (1.0f + cosTheta * cosTheta);
}

// MoveToTopAtmosphere — sky/Common.hlsl:320-339   (PLANET_RADIUS_OFFSET 10.0, :153)
static const float PLANET_RADIUS_OFFSET = 10.0f;
static inline bool MoveToTopAtmosphere(float3& aPosWorld, float3 aaViewDir, float AtmosphereTopRadius) {
  float viewHeight = length(aPosWorld);
  if (viewHeight > AtmosphereTopRadius) {
    float tTop = raySphereIntersectNearest(aPosWorld, aaViewDir, f3(0.0f), AtmosphereTopRadius);
    if (tTop >= 0.0f) {
      float3 UpVector = aPosWorld / viewHeight;
      float3 UpOffset = UpVector * -PLANET_RADIUS_OFFSET;
      aPosWorld = aPosWorld + aaViewDir * tTop + UpOffset;
    } else {
      return false;
    }
  }
  return true;
}

// theTextures2D[lut].SampleLevel(linearClamp, uv, 0).rgb — sky/Common.hlsl:510-511, adopted sampler (header).
static inline float3 SampleTransmittanceLut(const uint16_t* lut /*256*64*4 half*/, float2 uv) {
  const int W = TRANSMITTANCE_TEXTURE_WIDTH, Hh = TRANSMITTANCE_TEXTURE_HEIGHT;
  // Clamp addressing == clamping the texel-space coordinate to [0, N-1] before splitting it
  // (both taps collapse onto the border texel outside). fmaxf(NaN,0) = 0, so NaN reads texel 0.
  float x = clampf(uv.x * (float)W - 0.5f, 0.0f, (float)(W - 1));
  float y = clampf(uv.y * (float)Hh - 0.5f, 0.0f, (float)(Hh - 1));
  int x0 = (int)floorf(x), y0 = (int)floorf(y);
  float fx = x - (float)x0, fy = y - (float)y0;
  int x1 = x0 + 1 < W ? x0 + 1 : W - 1;
  int y1 = y0 + 1 < Hh ? y0 + 1 : Hh - 1;
  float c[3];
  for (int k = 0; k < 3; ++k) {
    float t00 = half_to_float(lut[(y0 * W + x0) * 4 + k]);
    float t10 = half_to_float(lut[(y0 * W + x1) * 4 + k]);
    float t01 = half_to_float(lut[(y1 * W + x0) * 4 + k]);
    float t11 = half_to_float(lut[(y1 * W + x1) * 4 + k]);
    float top = t00 * (1.0f - fx) + t10 * fx;
    float bot = t01 * (1.0f - fx) + t11 * fx;
    c[k] = top * (1.0f - fy) + bot * fy;
  }
  return f3(c[0], c[1], c[2]);
}

struct SingleScatteringResult { float3 L, OpticalDepth, Transmittance; };  // sky/Common.hlsl:357-366 (members used)

// IntegrateScatteredLuminance — sky/Common.hlsl:368-589.
// OPTICAL_DEPTH_ONLY==true restates instantiation (A), false restates (B). DepthBufferValue is -1 at both
// call sites (the :413-427 block is dead), ground is false at both (:567-583 dead), tMaxMax defaults to 9000000.
template <bool OPTICAL_DEPTH_ONLY>
static inline SingleScatteringResult IntegrateScatteredLuminance(float3 aPosWorld, float3 aViewDir, float3 aSunDir,
                                                                 const Atmosphere& A, float SampleCountIni,
                                                                 bool VariableSampleCount, bool MieRayPhase,
                                                                 float2 aRayMarchMinMaxSPP, float3 aSunIlluminance,
                                                                 const uint16_t* lut) {
  SingleScatteringResult result;
  result.L = f3(0.0f); result.OpticalDepth = f3(0.0f); result.Transmittance = f3(0.0f);
  const float tMaxMax = 9000000.0f;

  float3 earthO = f3(0.0f);
  float tBottom = raySphereIntersectNearest(aPosWorld, aViewDir, earthO, A.BottomRadius);
  float tTop = raySphereIntersectNearest(aPosWorld, aViewDir, earthO, A.TopRadius);
  float tMax = 0.0f;
  if (tBottom < 0.0f) {
    if (tTop < 0.0f) {
      tMax = 0.0f;
      return result;
    } else {
      tMax = tTop;
    }
  } else {
    if (tTop > 0.0f) tMax = hmin(tTop, tBottom);
  }
  tMax = hmin(tMax, tMaxMax);

  float SampleCount = SampleCountIni;
  float SampleCountFloor = SampleCountIni;
  float tMaxFloor = tMax;
  if (VariableSampleCount) {
    SampleCount = lerpf(aRayMarchMinMaxSPP.x, aRayMarchMinMaxSPP.y, saturate(tMax * 0.01f));
    SampleCountFloor = floorf(SampleCount);
    tMaxFloor = tMax * SampleCountFloor / SampleCount;
  }
  float dt = tMax / SampleCount;

  const float uniformPhase = 1.0f / (4.0f * PI);
  const float3 wi = aSunDir;
  const float3 wo = aViewDir;
  float cosTheta = dot(wi, wo);
  float MiePhaseValue = hgPhase(A.MiePhaseG, -cosTheta);
  float RayleighPhaseValue = RayleighPhase(cosTheta);

  float3 globalL = aSunIlluminance;

  float3 L = f3(0.0f);
  float3 throughput = f3(1.0f);
  float3 OpticalDepth = f3(0.0f);
  float t = 0.0f;
  const float SampleSegmentT = 0.3f;
  for (float s = 0.0f; s < SampleCount; s += 1.0f) {
    if (VariableSampleCount) {
      float t0 = (s) / SampleCountFloor;
      float t1 = (s + 1.0f) / SampleCountFloor;
      t0 = t0 * t0;
      t1 = t1 * t1;
      t0 = tMaxFloor * t0;
      if (t1 > 1.0f) t1 = tMax;
      else t1 = tMaxFloor * t1;
      t = t0 + (t1 - t0) * SampleSegmentT;
      dt = t1 - t0;
    } else {
      float NewT = tMax * (s + SampleSegmentT) / SampleCount;
      dt = NewT - t;
      t = NewT;
    }
    float3 P = aPosWorld + t * aViewDir;

    MediumSampleRGB medium = sampleMediumRGB(P, A);
    const float3 SampleOpticalDepth = medium.extinction * dt;
    const float3 SampleTransmittance = exp3(-SampleOpticalDepth);
    OpticalDepth += SampleOpticalDepth;

    if (!OPTICAL_DEPTH_ONLY) {
      float pHeight = length(P);
      const float3 UpVector = P / pHeight;
      float SunZenithCosAngle = dot(aSunDir, UpVector);
      float2 uv = LutTransmittanceParamsToUv(A.BottomRadius, A.TopRadius, pHeight, SunZenithCosAngle);
      float3 TransmittanceToSun = SampleTransmittanceLut(lut, uv);

      float3 PhaseTimesScattering;
      if (MieRayPhase) PhaseTimesScattering = medium.scatteringMie * MiePhaseValue + medium.scatteringRay * RayleighPhaseValue;
      else PhaseTimesScattering = medium.scattering * uniformPhase;

      float tEarth = raySphereIntersectNearest(P, aSunDir, earthO + PLANET_RADIUS_OFFSET * UpVector, A.BottomRadius);
      float earthShadow = tEarth >= 0.0f ? 0.0f : 1.0f;

      float3 multiScatteredLuminance = f3(0.0f);
      float shadow = 1.0f;
      float3 S = globalL * (earthShadow * shadow * TransmittanceToSun * PhaseTimesScattering +
                            multiScatteredLuminance * medium.scattering);

      float3 Sint = (S - S * SampleTransmittance) / medium.extinction;
      L += throughput * Sint;
      throughput *= SampleTransmittance;
    }
  }

  result.L = L;
  result.OpticalDepth = OpticalDepth;
  result.Transmittance = throughput;
  return result;
}

// compute_transmittance_lut.hlsl main — :15-62. One texel.
static inline void TransmittanceLutTexel(const Atmosphere& A, uint32_t ix, uint32_t iy, uint16_t out_rgba[4]) {
  float2 pixPos = {(float)ix + 0.5f, (float)iy + 0.5f};
  float2 uv = {pixPos.x / (float)TRANSMITTANCE_TEXTURE_WIDTH, pixPos.y / (float)TRANSMITTANCE_TEXTURE_HEIGHT};
  float viewHeight, viewZenithCosAngle;
  UvToLutTransmittanceParams(A, viewHeight, viewZenithCosAngle, uv);
  float3 posPlanetSpace = f3(0.0f, viewHeight, 0.0f);
  float3 worldDir = f3(0.0f, viewZenithCosAngle, sqrtf(1.0f - viewZenithCosAngle * viewZenithCosAngle));
  float3 opticalDepth = IntegrateScatteredLuminance<true>(posPlanetSpace, worldDir, f3(0.0f), A, 40.0f, false, false,
                                                          float2{0.0f, 0.0f}, f3(0.0f), nullptr).OpticalDepth;
  float3 transmittance = exp3(-opticalDepth);
  out_rgba[0] = float_to_half(transmittance.x);
  out_rgba[1] = float_to_half(transmittance.y);
  out_rgba[2] = float_to_half(transmittance.z);
  out_rgba[3] = float_to_half(1.0f);
}

// SampleSkyLuminance — raytracing/SampleSky.hlsl:7-47
static inline float3 SampleSkyLuminance(float3 viewPos, float3 viewDir, const PtRtConsts& c, const uint16_t* lut) {
  if (!c.mySampleSky) return f3(c.mySkyFallbackEmission[0], c.mySkyFallbackEmission[1], c.mySkyFallbackEmission[2]);
  Atmosphere A;
  memcpy(&A, &c.mySkyConsts.myAtmosphere, sizeof(A));
  float3 worldPos = viewPos + f3(0.0f, A.BottomRadius, 0.0f);
  float3 skyLuminance = f3(0.0f);
  if (MoveToTopAtmosphere(worldPos, viewDir, A.TopRadius)) {
    const float sampleCountIni = 30.0f;
    float3 sunDir = f3(c.mySkyConsts.mySunDirection[0], c.mySkyConsts.mySunDirection[1], c.mySkyConsts.mySunDirection[2]);
    float3 sunIll = f3(c.mySkyConsts.mySunIlluminance[0], c.mySkyConsts.mySunIlluminance[1], c.mySkyConsts.mySunIlluminance[2]);
    float2 mm = {c.mySkyConsts.myRayMarchMinMaxSPP[0], c.mySkyConsts.myRayMarchMinMaxSPP[1]};
    SingleScatteringResult ss = IntegrateScatteredLuminance<false>(worldPos, viewDir, sunDir, A, sampleCountIni, true,
                                                                   true, mm, sunIll, lut);
    skyLuminance = ss.L;
  }
  return skyLuminance;
}

}  // namespace orc
