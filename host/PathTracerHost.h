// PathTracerHost.h — headless C++ host mirror of the reference's application classes for the path-tracing hot path.
// The reference is compiled C++ (PathTracer/PathTracer.{h,cpp}, Sky.{h,cpp}); this keeps its class surface — member
// names, defaults, call order — and routes everything through the C-ABI (include/pt_api.h). No window, UI or raster.
//   PathTracer  <- PathTracer/PathTracer.h:58-135
//   Sky         <- PathTracer/Sky.h:68-99, Sky.cpp:32-56,127-148
//   Camera      <- FANCY Common/Camera.h (absent; semantics inferred from PathTracer.cpp:134-148,686-687,747-750)
//   SceneData   <- FANCY IO/Scene.h as consumed by InitRtScene (PathTracer.cpp:171-339)
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/pt_api.h"

namespace pth {

struct float3 { float x = 0, y = 0, z = 0; };

struct MeshData {
  std::vector<PtVertex> myVertexData;   // interleaved pos/normal/uv (PathTracer.cpp:116-120)
  std::vector<uint32_t> myIndexData;
};
struct SceneMeshInstance {
  uint32_t myMeshIndex = 0, myMaterialIndex = 0;
  std::array<float, 16> myTransform{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
};
struct MaterialDesc {
  float myEmission[3] = {0, 0, 0};
  float myColor[4] = {1, 1, 1, 1};
  // material extension as read from MTL (not in the reference): Pr / Pm select the GGX model, map_Kd [-s u v w] names the
  // albedo image, pt_albedo_pattern <id> <scale> a procedural pattern
  bool myHasPbr = false;
  float myRoughness = 0.5f, myMetallic = 0.0f;
  uint32_t myAlbedoPattern = 0;
  float myPatternScale = 1.0f;
  std::string myAlbedoMap;                 // path of the map_Kd file ("" = none)
  float myAlbedoMapScale = 1.0f;
};
struct SceneData {
  std::vector<MeshData> myMeshes;
  std::vector<SceneMeshInstance> myInstances;
  std::vector<MaterialDesc> myMaterials;
  // filled by ImportObj when the MTL carries PBR statements; InitRtScene hands them to pt_scene_set_material_ext /
  // pt_scene_set_albedo_texture (one image per scene: the first map_Kd file; materials naming another keep their Kd)
  std::vector<PtMaterialExt> myMaterialsExt;
  std::vector<uint8_t> myAlbedoTexture;    // RGBA8, row 0 = top
  uint32_t myAlbedoWidth = 0, myAlbedoHeight = 0;
  std::vector<std::string> mySkippedTextures;
};

// Texture file -> RGBA8 texels: 8-bit non-interlaced PNG (every colour type, all row filters) and binary PPM / PGM.
bool ReadImageRgba8(const std::string& aPath, std::vector<uint8_t>& someTexelsOut, uint32_t& aWidthOut, uint32_t& aHeightOut,
                    std::string* anErrorOut = nullptr);

// Minimal OBJ/MTL import: one mesh + instance per (o/g, usemtl) run in file order (stand-in for MeshImporter::Import).
bool ImportObj(const std::string& aPath, SceneData& aSceneOut, std::string* anErrorOut = nullptr);
uint32_t Encode_Unorm_RGBA(const float aColor[4]);

class Camera {
public:
  float3 myPosition;
  float myOrientation[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};   // rows: right, up, forward
  float myFovDeg = 60.0f, myNear = 1.0f, myFar = 10000.0f;
  float myWidth = 1280.0f, myHeight = 720.0f;
  bool myIsOrtho = false;
  void LookAt(float3 anEye, float3 aTarget, float3 anUp = {0, 1, 0});
  void GetVerticesOnNearPlane(float3 someVerticesOut[4]) const;   // v0 BL, v1 BR, v2 TR, v3 TL
  // myViewProj as CameraHasChanged reads it (PathTracer.cpp:446-453): proj * view, row-major [row][col]; adopted definition
  // (FANCY's UpdateView / UpdateProjection are absent): camera axes as the view rotation, left-handed D3D perspective
  void GetViewProj(float aMatOut[4][4]) const;
};

class PtException : public std::runtime_error {
public:
  PtException(int aCode, const std::string& aMsg) : std::runtime_error(aMsg), myCode(aCode) {}
  int myCode;
};

class Sky {
public:
  explicit Sky(pt_ctx* aCtx);
  void ComputeTranmittanceLut();           // Sky.cpp:127-148
  float myMultiScatteringFactor = 0.0f;    // Sky.h:80-83
  float3 mySunDir{0, 1, 0};
  float3 mySunIlluminance{1000.0f, 1000.0f, 1000.0f};
  PtAtmosphere myAtmosphereParams;
private:
  pt_ctx* myCtx;
};
void SetupEarthAtmosphere(PtAtmosphere& someParams);   // Sky.cpp:32-56

class PathTracer {
public:
  PathTracer(int aDevice, uint32_t aWindowWidth = 1280, uint32_t aWindowHeight = 720);
  ~PathTracer();
  PathTracer(const PathTracer&) = delete;
  PathTracer& operator=(const PathTracer&) = delete;

  void LoadScene(const char* aPath, const float3& aCamPos);            // PathTracer.cpp:115-149
  void LoadScene(const SceneData& aScene, const float3& aCamPos);
  void InitSky();                                                      // :151-154
  void InitRtScene(const SceneData& aScene);                           // :171-339
  void OnWindowResized(uint32_t aWidth, uint32_t aHeight);             // :439-444
  void Update();                                                       // :557-571: restart when !myAccumulate, camera moved, sky changed
  bool CameraHasChanged() const;                                       // :446-453 (epsilon 1e-4 on the view-projection matrix)
  void SkySettingsChanged();                                           // Sky_Imgui::HaveSettingsChanged -> RestartAccumulation (:566-568)
  void Render(uint32_t aNumFrames = 1) { RenderRT(aNumFrames); }       // :573-595, RT branch
  void RenderRT(uint32_t aNumFrames = 1);                              // :659-787
  void UpdateOutputTexture();                                          // :834-878
  void RestartAccumulation();                                          // :912-915
  PtRtConsts BuildRtConsts() const;                                    // :689-768
  void ReadOutput(std::vector<float>& anRgbaMeanOut);                  // content of the RGBA32F light texture
  static void Tonemap(std::vector<float>& anRgbaInOut);                // tonemap_composit.hlsl:37-44
  bool WritePfm(const char* aPath, bool aTonemap = false);

  pt_ctx* myCtx = nullptr;
  Sky* mySky = nullptr;
  Camera myCamera;
  uint32_t myWindowWidth, myWindowHeight, myOutputWidth = 0, myOutputHeight = 0;
  // PathTracer.h:112-131
  float myAoDistance = 1.0f;
  uint32_t myNumAccumulationFrames = 0u;
  bool myAccumulationNeedsClear = true;
  bool myRenderRaster = false, myRenderAo = false, myAccumulate = true, myHalfResRender = true, mySampleSky = true;
  float mySkyFallbackIntensity = 100.0f;
  int myMaxRecursionDepth = 4;
  int myLightInstanceIdx = 4;
  bool myLightEnabled = true;
  float3 myLightColor{1.0f, 1.0f, 1.0f};
  float myLightStrength = 100.0f;
  float myPhongSpecularPower = 10.0f;
  bool mySkyDirty = true;
  float myLastViewMat[4][4];                                           // PathTracer.h:115

private:
  void Check(int aCode) const;
  struct SkySnapshot { float3 mySunDir, mySunIlluminance; PtAtmosphere myAtmosphere; };
  SkySnapshot myLastSky;
  SkySnapshot SnapshotSky() const;
};

}  // namespace pth
