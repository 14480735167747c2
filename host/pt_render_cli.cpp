// pt_render_cli — headless replacement of the reference's WinMain loop (PathTracer/PathTracer_main.cpp:26-61) for the
// path-tracing mode: load an OBJ, accumulate N frames, write a PFM.
//   pt_render_cli scene.obj out.pfm [--spp N] [--width W] [--height H] [--cam x y z] [--depth D] [--no-sky]
//                 [--fallback I] [--light-instance K] [--full-res] [--tonemap] [--device G] [--ao DIST]
//   pt_render_cli --check-obj scene.obj      (parse only, no GPU: prints mesh / triangle / material counts)
#include <string>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "PathTracerHost.h"

int main(int argc, char** argv) {
  if (argc == 3 && strcmp(argv[1], "--check-obj") == 0) {   // importer only: needs no GPU
    pth::SceneData sd;
    std::string err;
    if (!pth::ImportObj(argv[2], sd, &err)) { fprintf(stderr, "Failed importing scene: %s\n", err.c_str()); return 1; }
    size_t tris = 0;
    for (const auto& m : sd.myMeshes) tris += m.myIndexData.size() / 3;
    printf("meshes %zu triangles %zu materials %zu\n", sd.myMeshes.size(), tris, sd.myMaterials.size());
    for (size_t i = 0; i < sd.myMaterialsExt.size(); ++i) {   // material extension read from the MTL, one line per material
      const PtMaterialExt& e = sd.myMaterialsExt[i];
      printf("ext %zu roughness %.9g metallic %.9g pattern %u scale %.9g flags %u\n", i, e.roughness, e.metallic, e.albedo_pattern,
             e.pattern_scale, e.flags);
    }
    if (!sd.myAlbedoTexture.empty()) {
      uint32_t sum = 0;   // Adler-style checksum of the texels, for comparison with the Python host's decoder
      uint32_t a = 1, b = 0;
      for (uint8_t t : sd.myAlbedoTexture) { a = (a + t) % 65521u; b = (b + a) % 65521u; }
      sum = (b << 16) | a;
      printf("texture %ux%u adler32 %u skipped %zu\n", sd.myAlbedoWidth, sd.myAlbedoHeight, sum, sd.mySkippedTextures.size());
    }
    return 0;
  }
  if (argc < 3) {
    fprintf(stderr, "usage: %s scene.obj out.pfm [--spp N] [--width W] [--height H] [--cam x y z] [--depth D] [--no-sky] "
                    "[--fallback I] [--light-instance K] [--full-res] [--tonemap] [--device G] [--ao DIST]\n", argv[0]);
    return 2;
  }
  uint32_t spp = 64, w = 1280, h = 720;   // window 1280x720 (PathTracer_main.cpp:41-42)
  pth::float3 cam{1.0f, 102.0f, -30.0f};  // sceneLoadInfos[0] (PathTracer.cpp:40)
  int depth = 4, device = 0, light = 4;
  bool sky = true, fullres = false, tonemap = false, ao = false;
  float fallback = 100.0f, ao_dist = 1.0f;
  for (int i = 3; i < argc; ++i) {
    auto is = [&](const char* s) { return strcmp(argv[i], s) == 0; };
    if (is("--spp") && i + 1 < argc) spp = (uint32_t)atoi(argv[++i]);
    else if (is("--width") && i + 1 < argc) w = (uint32_t)atoi(argv[++i]);
    else if (is("--height") && i + 1 < argc) h = (uint32_t)atoi(argv[++i]);
    else if (is("--depth") && i + 1 < argc) depth = atoi(argv[++i]);
    else if (is("--device") && i + 1 < argc) device = atoi(argv[++i]);
    else if (is("--light-instance") && i + 1 < argc) light = atoi(argv[++i]);
    else if (is("--fallback") && i + 1 < argc) fallback = (float)atof(argv[++i]);
    else if (is("--cam") && i + 3 < argc) { cam.x = (float)atof(argv[++i]); cam.y = (float)atof(argv[++i]); cam.z = (float)atof(argv[++i]); }
    else if (is("--ao") && i + 1 < argc) { ao = true; ao_dist = (float)atof(argv[++i]); }
    else if (is("--no-sky")) sky = false;
    else if (is("--full-res")) fullres = true;
    else if (is("--tonemap")) tonemap = true;
    else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
  }
  try {
    pth::PathTracer app(device, w, h);
    app.myHalfResRender = !fullres;
    app.UpdateOutputTexture();
    app.LoadScene(argv[1], cam);
    app.mySampleSky = sky;
    app.mySkyFallbackIntensity = fallback;
    app.myMaxRecursionDepth = depth;
    app.myLightInstanceIdx = light;
    app.myRenderAo = ao;          // "Render AO" checkbox + "Ao Distance" (PathTracer.cpp:505-523)
    app.myAoDistance = ao_dist;
    auto t0 = std::chrono::steady_clock::now();
    app.Update();
    app.Render(spp);
    pt_synchronize(app.myCtx);
    auto t1 = std::chrono::steady_clock::now();
    if (!app.WritePfm(argv[2], tonemap)) { fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
    PtStats st;
    st.struct_size = (uint32_t)sizeof(st);   // versioned by size (include/pt_api.h)
    pt_stats_get(app.myCtx, &st);
    const double s = std::chrono::duration<double>(t1 - t0).count();
    printf("%ux%u, %u frames, %llu rays in %.3f s (%.1f Mrays/s) -> %s\n", app.myOutputWidth, app.myOutputHeight, spp,
           (unsigned long long)st.rays, s, st.rays / s / 1e6, argv[2]);
  } catch (const pth::PtException& e) {
    fprintf(stderr, "error %d: %s\n", e.myCode, e.what());
    return 1;
  }
  return 0;
}
