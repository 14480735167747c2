// PathTracerHost.cpp — see PathTracerHost.h. Reference: PathTracer/PathTracer.cpp, PathTracer/Sky.cpp.
#include "PathTracerHost.h"

#include <cctype>
#include <cerrno>
#include <climits>
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <tuple>

#include <zlib.h>

namespace pth {

static float3 sub(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static float3 add(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static float3 mul(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static float3 cross(float3 a, float3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static float3 norm(float3 a) { float l = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); return {a.x / l, a.y / l, a.z / l}; }

uint32_t Encode_Unorm_RGBA(const float c[4]) {
  uint32_t r = 0;
  for (int k = 0; k < 4; ++k) {
    float v = std::min(std::max(c[k], 0.0f), 1.0f);
    r |= (uint32_t)std::floor(v * 255.0f + 0.5f) << (8 * k);
  }
  return r;
}

// ------------------------------------------------------------------------------------------------ OBJ / MTL
static bool IsNumber(const std::string& t) {
  char* end = nullptr;
  std::strtod(t.c_str(), &end);
  return !t.empty() && end && *end == '\0';
}

static std::map<std::string, MaterialDesc> ParseMtl(const std::string& aPath, std::vector<std::string>& anOrderOut) {
  std::map<std::string, MaterialDesc> mats;
  std::ifstream f(aPath);
  std::string line, cur;
  const std::string dir = aPath.find_last_of("/\\") == std::string::npos ? "" : aPath.substr(0, aPath.find_last_of("/\\") + 1);
  while (std::getline(f, line)) {
    std::istringstream ss(line);
    std::string t;
    ss >> t;
    if (t == "newmtl") { ss >> cur; mats[cur] = MaterialDesc(); anOrderOut.push_back(cur); }
    else if (!cur.empty() && t == "Kd") { ss >> mats[cur].myColor[0] >> mats[cur].myColor[1] >> mats[cur].myColor[2]; }
    else if (!cur.empty() && t == "Ke") { ss >> mats[cur].myEmission[0] >> mats[cur].myEmission[1] >> mats[cur].myEmission[2]; }
    else if (!cur.empty() && t == "Pr") { ss >> mats[cur].myRoughness; mats[cur].myHasPbr = true; }
    else if (!cur.empty() && t == "Pm") { ss >> mats[cur].myMetallic; mats[cur].myHasPbr = true; }
    else if (!cur.empty() && t == "pt_albedo_pattern") { ss >> mats[cur].myAlbedoPattern >> mats[cur].myPatternScale; }
    else if (!cur.empty() && t == "map_Kd") {
      // options ("-s u v w", "-o …", "-clamp on", …) come first, the file name last
      std::vector<std::string> tok;
      std::string x;
      while (ss >> x) tok.push_back(x);
      size_t k = 0;
      while (k < tok.size()) {
        if (tok[k].size() > 1 && tok[k][0] == '-' && !std::isdigit((unsigned char)tok[k][1]) && k + 1 < tok.size()) {
          const std::string opt = tok[k++];
          std::vector<std::string> vals;
          while (k + 1 < tok.size() && (tok[k] == "on" || tok[k] == "off" || IsNumber(tok[k]))) vals.push_back(tok[k++]);
          if (opt == "-s" && !vals.empty() && IsNumber(vals[0])) mats[cur].myAlbedoMapScale = (float)std::atof(vals[0].c_str());
        } else {
          std::string name = tok[k];
          for (size_t j = k + 1; j < tok.size(); ++j) name += " " + tok[j];
          std::replace(name.begin(), name.end(), '\\', '/');
          mats[cur].myAlbedoMap = dir + name;
          break;
        }
      }
    }
  }
  return mats;
}

// ------------------------------------------------------------------------------------------------ texture files
static uint32_t Be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static bool ReadPng(const std::vector<uint8_t>& d, const std::string& aPath, std::vector<uint8_t>& out, uint32_t& w, uint32_t& h, std::string* err) {
  auto bad = [&](const std::string& what) { if (err) *err = aPath + ": " + what; return false; };
  size_t pos = 8;
  bool haveHdr = false;
  uint32_t depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte, trns;
  while (pos + 12 <= d.size()) {
    const uint32_t n = Be32(&d[pos]);
    if ((uint64_t)pos + 12 + n > d.size()) return bad("truncated chunk");
    const uint8_t* tag = &d[pos + 4];
    const uint8_t* body = &d[pos + 8];
    if (Be32(&d[pos + 8 + n]) != (uint32_t)crc32(crc32(0L, Z_NULL, 0), tag, n + 4)) return bad("CRC mismatch");
    const std::string t((const char*)tag, 4);
    if (t == "IHDR" && n == 13) { w = Be32(body); h = Be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; haveHdr = true; }
    else if (t == "PLTE") plte.assign(body, body + n);
    else if (t == "tRNS") trns.assign(body, body + n);
    else if (t == "IDAT") idat.insert(idat.end(), body, body + n);
    else if (t == "IEND") break;
    pos += 12 + (size_t)n;
  }
  const uint32_t ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!haveHdr || idat.empty()) return bad("no IHDR / IDAT");
  if (depth != 8 || ch == 0 || interlace != 0 || w == 0 || h == 0 || (uint64_t)w * h > (1u << 28)) return bad("only 8-bit non-interlaced PNG is read");
  const size_t stride = (size_t)w * ch;
  std::vector<uint8_t> raw((stride + 1) * h);
  uLongf rawLen = (uLongf)raw.size();
  if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size()) return bad("bad image data");
  std::vector<uint8_t> px(stride * h), zero(stride, 0);
  for (uint32_t y = 0; y < h; ++y) {
    const uint8_t ft = raw[(stride + 1) * y];
    const uint8_t* line = &raw[(stride + 1) * y + 1];
    const uint8_t* up = y ? &px[stride * (y - 1)] : zero.data();
    uint8_t* cur = &px[stride * y];
    if (ft > 4) return bad("bad row filter");
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= ch ? cur[i - ch] : 0, b = up[i], c = i >= ch ? up[i - ch] : 0;
      int pred = 0;
      if (ft == 1) pred = a;
      else if (ft == 2) pred = b;
      else if (ft == 3) pred = (a + b) >> 1;
      else if (ft == 4) { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
      cur[i] = (uint8_t)(line[i] + pred);
    }
  }
  out.assign((size_t)w * h * 4, 255);
  for (size_t i = 0; i < (size_t)w * h; ++i) {
    const uint8_t* s = &px[i * ch];
    uint8_t* o = &out[i * 4];
    if (ctype == 0) { o[0] = o[1] = o[2] = s[0]; }
    else if (ctype == 4) { o[0] = o[1] = o[2] = s[0]; o[3] = s[1]; }
    else if (ctype == 2) { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; }
    else if (ctype == 6) { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3]; }
    else {
      if ((size_t)s[0] * 3 + 2 >= plte.size()) return bad("palette index out of range");
      o[0] = plte[s[0] * 3]; o[1] = plte[s[0] * 3 + 1]; o[2] = plte[s[0] * 3 + 2];
      if (s[0] < trns.size()) o[3] = trns[s[0]];
    }
  }
  return true;
}

static bool ReadPnm(const std::vector<uint8_t>& d, const std::string& aPath, std::vector<uint8_t>& out, uint32_t& w, uint32_t& h, std::string* err) {
  auto bad = [&](const std::string& what) { if (err) *err = aPath + ": " + what; return false; };
  std::vector<std::string> tok;
  size_t pos = 0;
  while (tok.size() < 4) {
    while (pos < d.size() && std::isspace(d[pos])) ++pos;
    if (pos < d.size() && d[pos] == '#') { while (pos < d.size() && d[pos] != '\n') ++pos; continue; }
    size_t end = pos;
    while (end < d.size() && !std::isspace(d[end])) ++end;
    if (end == pos) return bad("truncated PNM header");
    tok.emplace_back((const char*)&d[pos], end - pos);
    pos = end;
  }
  ++pos;
  if ((tok[0] != "P5" && tok[0] != "P6") || tok[3] != "255" || !IsNumber(tok[1]) || !IsNumber(tok[2])) return bad("only binary P5 / P6 with maxval 255 is read");
  const long lw = std::atol(tok[1].c_str()), lh = std::atol(tok[2].c_str());
  const size_t ch = tok[0] == "P6" ? 3 : 1;
  if (lw <= 0 || lh <= 0 || (uint64_t)lw * lh > (1u << 28) || pos > d.size() || d.size() - pos < (size_t)lw * lh * ch) return bad("pixel data shorter than the header says");
  w = (uint32_t)lw; h = (uint32_t)lh;
  out.assign((size_t)w * h * 4, 255);
  for (size_t i = 0; i < (size_t)w * h; ++i)
    for (size_t k = 0; k < 3; ++k) out[i * 4 + k] = d[pos + i * ch + (ch == 3 ? k : 0)];
  return true;
}

bool ReadImageRgba8(const std::string& aPath, std::vector<uint8_t>& out, uint32_t& w, uint32_t& h, std::string* err) {
  std::ifstream f(aPath, std::ios::binary);
  if (!f) { if (err) *err = "cannot open " + aPath; return false; }
  std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  if (d.size() >= 8 && memcmp(d.data(), sig, 8) == 0) return ReadPng(d, aPath, out, w, h, err);
  if (d.size() >= 2 && d[0] == 'P' && (d[1] == '5' || d[1] == '6')) return ReadPnm(d, aPath, out, w, h, err);
  if (err) *err = aPath + ": unsupported image format (8-bit PNG and binary PPM / PGM are read)";
  return false;
}

bool ImportObj(const std::string& aPath, SceneData& out, std::string* err) {
  std::ifstream f(aPath);
  if (!f) { if (err) *err = "cannot open " + aPath; return false; }
  std::vector<float3> pos, nrm;
  std::vector<std::array<float, 2>> tex;
  std::map<std::string, MaterialDesc> mtl;
  std::vector<std::string> matOrder;
  struct Group { std::string mat; std::vector<std::array<std::array<int, 3>, 3>> faces; };
  std::vector<Group> groups;
  bool open = false;
  std::string curMat, line;
  size_t lineNo = 0;
  auto bad = [&](const std::string& what) { if (err) *err = aPath + ":" + std::to_string(lineNo) + ": " + what; return false; };
  const std::string dir = aPath.find_last_of("/\\") == std::string::npos ? "" : aPath.substr(0, aPath.find_last_of("/\\") + 1);
  while (std::getline(f, line)) {
    ++lineNo;
    std::istringstream ss(line);
    std::string t;
    ss >> t;
    if (t == "v") { float3 p; ss >> p.x >> p.y >> p.z; pos.push_back(p); }
    else if (t == "vn") { float3 p; ss >> p.x >> p.y >> p.z; nrm.push_back(p); }
    else if (t == "vt") { std::array<float, 2> p{{0, 0}}; ss >> p[0] >> p[1]; tex.push_back(p); }
    else if (t == "mtllib") { std::string n; ss >> n; auto m = ParseMtl(dir + n, matOrder); mtl.insert(m.begin(), m.end()); }
    else if (t == "o" || t == "g") { open = false; }
    else if (t == "usemtl") { ss >> curMat; open = false; }
    else if (t == "f") {
      if (!open) { groups.push_back(Group{curMat, {}}); open = true; }
      std::vector<std::array<int, 3>> corners;
      std::string c;
      while (ss >> c) {
        std::array<int, 3> k{{0, 0, 0}};
        int part = 0;
        std::string num;
        for (size_t i = 0; i <= c.size(); ++i) {
          if (i == c.size() || c[i] == '/') {
            if (!num.empty() && part < 3) {   // strtol with full checks: a malformed token is an error, not an exception
              char* end = nullptr;
              errno = 0;
              const long v = std::strtol(num.c_str(), &end, 10);
              if (errno != 0 || end == num.c_str() || *end != '\0' || v > INT32_MAX || v < -INT32_MAX) return bad("malformed face index '" + c + "'");
              k[part] = (int)v;
            }
            num.clear();
            ++part;
          } else num.push_back(c[i]);
        }
        if (k[0] == 0) return bad("face corner '" + c + "' has no position index (OBJ indices start at 1)");
        k[0] = k[0] > 0 ? k[0] - 1 : (int)pos.size() + k[0];
        k[1] = k[1] > 0 ? k[1] - 1 : (k[1] < 0 ? (int)tex.size() + k[1] : -1);
        k[2] = k[2] > 0 ? k[2] - 1 : (k[2] < 0 ? (int)nrm.size() + k[2] : -1);
        // every resolved index must name an element read so far (relative indices count back from the current end)
        if (k[0] < 0 || k[0] >= (int)pos.size()) return bad("position index out of range in '" + c + "'");
        if (k[1] < -1 || k[1] >= (int)tex.size()) return bad("texture-coordinate index out of range in '" + c + "'");
        if (k[2] < -1 || k[2] >= (int)nrm.size()) return bad("normal index out of range in '" + c + "'");
        corners.push_back(k);
      }
      for (size_t i = 1; i + 1 < corners.size(); ++i) groups.back().faces.push_back({{corners[0], corners[i], corners[i + 1]}});
    }
  }
  for (auto& g : groups)
    if (!g.mat.empty() && !mtl.count(g.mat)) { mtl[g.mat] = MaterialDesc(); matOrder.push_back(g.mat); }
  if (matOrder.empty()) { matOrder.push_back("default"); mtl["default"] = MaterialDesc(); }
  out = SceneData();
  for (auto& n : matOrder) out.myMaterials.push_back(mtl[n]);
  for (auto& g : groups) {
    MeshData m;
    std::map<std::tuple<int, int, int>, uint32_t> vmap;
    for (auto& tri : g.faces) {
      float3 fn{0, 0, 1};
      { float3 a = pos[tri[0][0]], b = pos[tri[1][0]], c = pos[tri[2][0]]; float3 n = cross(sub(b, a), sub(c, a)); float l = std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z); if (l > 0) fn = mul(n, 1.0f / l); }
      for (auto& k : tri) {
        auto key = std::make_tuple(k[0], k[1], k[2]);
        auto it = vmap.find(key);
        if (it == vmap.end()) {
          PtVertex v;
          float3 p = pos[k[0]], n = k[2] >= 0 ? nrm[k[2]] : fn;
          v.pos[0] = p.x; v.pos[1] = p.y; v.pos[2] = p.z;
          v.normal[0] = n.x; v.normal[1] = n.y; v.normal[2] = n.z;
          v.uv[0] = k[1] >= 0 ? tex[k[1]][0] : 0.0f; v.uv[1] = k[1] >= 0 ? tex[k[1]][1] : 0.0f;
          it = vmap.emplace(key, (uint32_t)m.myVertexData.size()).first;
          m.myVertexData.push_back(v);
        }
        m.myIndexData.push_back(it->second);
      }
    }
    SceneMeshInstance inst;
    inst.myMeshIndex = (uint32_t)out.myMeshes.size();
    auto mi = std::find(matOrder.begin(), matOrder.end(), g.mat);
    inst.myMaterialIndex = mi == matOrder.end() ? 0u : (uint32_t)(mi - matOrder.begin());
    out.myMeshes.push_back(std::move(m));
    out.myInstances.push_back(inst);
  }
  // material extension from the MTL's PBR statements (none: the reference's material model, nothing attached)
  bool anyExt = false;
  for (auto& md : out.myMaterials) anyExt = anyExt || md.myHasPbr || md.myAlbedoPattern != 0 || !md.myAlbedoMap.empty();
  if (anyExt) {
    std::string texPath;
    for (auto& md : out.myMaterials) {
      PtMaterialExt e;
      memset(&e, 0, sizeof(e));
      e.roughness = md.myRoughness;
      e.metallic = md.myMetallic;
      e.albedo_pattern = md.myAlbedoPattern;
      e.pattern_scale = md.myPatternScale;
      e.flags = md.myHasPbr ? PT_MAT_GGX : 0u;
      if (!md.myAlbedoMap.empty()) {
        if (texPath.empty()) texPath = md.myAlbedoMap;
        if (md.myAlbedoMap == texPath) { e.albedo_pattern = 3; e.pattern_scale = md.myAlbedoMapScale; }
        else out.mySkippedTextures.push_back(md.myAlbedoMap);
      }
      out.myMaterialsExt.push_back(e);
    }
    if (!texPath.empty() && !ReadImageRgba8(texPath, out.myAlbedoTexture, out.myAlbedoWidth, out.myAlbedoHeight, err)) return false;
  }
  return true;
}

// ------------------------------------------------------------------------------------------------ Camera
void Camera::LookAt(float3 eye, float3 target, float3 up) {
  float3 f = norm(sub(target, eye)), r = norm(cross(up, f)), u = cross(f, r);
  myPosition = eye;
  const float3 rows[3] = {r, u, f};
  for (int i = 0; i < 3; ++i) { myOrientation[i][0] = rows[i].x; myOrientation[i][1] = rows[i].y; myOrientation[i][2] = rows[i].z; }
}

void Camera::GetVerticesOnNearPlane(float3 v[4]) const {
  const float3 right{myOrientation[0][0], myOrientation[0][1], myOrientation[0][2]};
  const float3 up{myOrientation[1][0], myOrientation[1][1], myOrientation[1][2]};
  const float3 fwd{myOrientation[2][0], myOrientation[2][1], myOrientation[2][2]};
  const float halfH = (float)(myNear * std::tan(myFovDeg * 3.14159265358979323846 / 180.0 * 0.5));
  const float halfW = halfH * (myWidth / myHeight);
  const float3 c = add(myPosition, mul(fwd, myNear));
  v[0] = sub(sub(c, mul(right, halfW)), mul(up, halfH));
  v[1] = sub(add(c, mul(right, halfW)), mul(up, halfH));
  v[2] = add(add(c, mul(right, halfW)), mul(up, halfH));
  v[3] = add(sub(c, mul(right, halfW)), mul(up, halfH));
}

void Camera::GetViewProj(float m[4][4]) const {
  double view[4][4] = {{0}}, proj[4][4] = {{0}};
  const double pos[3] = {myPosition.x, myPosition.y, myPosition.z};
  for (int r = 0; r < 3; ++r) {
    double t = 0.0;
    for (int c = 0; c < 3; ++c) { view[r][c] = myOrientation[r][c]; t += (double)myOrientation[r][c] * pos[c]; }
    view[r][3] = -t;
  }
  view[3][3] = 1.0;
  const double f = 1.0 / std::tan(myFovDeg * 3.14159265358979323846 / 180.0 * 0.5), n = myNear, fr = myFar;
  proj[0][0] = f / ((double)myWidth / myHeight);
  proj[1][1] = f;
  proj[2][2] = fr / (fr - n);
  proj[2][3] = -n * fr / (fr - n);
  proj[3][2] = 1.0;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      double a = 0.0;
      for (int k = 0; k < 4; ++k) a += proj[r][k] * view[k][c];
      m[r][c] = (float)a;
    }
}

// ------------------------------------------------------------------------------------------------ Sky
void SetupEarthAtmosphere(PtAtmosphere& p) {
  const float EarthBottomRadius = 6360000.0f, EarthTopRadius = 6460000.0f;
  const float EarthRayleighScaleHeight = 8000.0f, EarthMieScaleHeight = 1200.0f;
  const float rs[3] = {0.000005802f, 0.000013558f, 0.000033100f}, ae[3] = {0.000000650f, 0.000001881f, 0.0000000085f};
  for (int i = 0; i < 3; ++i) {
    p.RayleighScattering[i] = rs[i];
    p.AbsorptionExtinction[i] = ae[i];
    p.GroundAlbedo[i] = 0.0f;
    p.MieScattering[i] = 0.000003996f;
    p.MieExtinction[i] = 0.000004440f;
    p.MieAbsorption[i] = std::max(0.0f, p.MieExtinction[i] - p.MieScattering[i]);
  }
  p.RayleighDensityExpScale = -1.0f / EarthRayleighScaleHeight;
  p.BottomRadius = EarthBottomRadius;
  p.TopRadius = EarthTopRadius;
  p.MieDensityExpScale = -1.0f / EarthMieScaleHeight;
  p.MiePhaseG = 0.8f;
  p.AbsorptionDensity0LayerWidth = 25000.0f;
  p.AbsorptionDensity0ConstantTerm = -2.0f / 3.0f;
  p.AbsorptionDensity0LinearTerm = 1.0f / 15.0f;
  p.AbsorptionDensity1ConstantTerm = 8.0f / 3.0f;
  p.AbsorptionDensity1LinearTerm = -1.0f / 15.0f;
}

Sky::Sky(pt_ctx* aCtx) : myCtx(aCtx) { SetupEarthAtmosphere(myAtmosphereParams); }
void Sky::ComputeTranmittanceLut() {
  int rc = pt_sky_set(myCtx, &myAtmosphereParams);
  if (rc != PT_OK) throw PtException(rc, pt_last_error(myCtx));
}

// ------------------------------------------------------------------------------------------------ PathTracer
void PathTracer::Check(int rc) const {
  if (rc != PT_OK) throw PtException(rc, pt_last_error(myCtx));
}

PathTracer::PathTracer(int aDevice, uint32_t w, uint32_t h) : myWindowWidth(w), myWindowHeight(h) {
  int rc = pt_create(aDevice, &myCtx);
  if (rc != PT_OK) throw PtException(rc, pt_last_error(nullptr));
  myCamera.myWidth = (float)w;
  myCamera.myHeight = (float)h;
  UpdateOutputTexture();
  InitSky();
  myCamera.GetViewProj(myLastViewMat);
  myLastSky = SnapshotSky();
}

PathTracer::~PathTracer() {
  delete mySky;
  if (myCtx) pt_destroy(myCtx);
}

void PathTracer::InitSky() {
  delete mySky;
  mySky = new Sky(myCtx);
  mySkyDirty = true;
}

void PathTracer::LoadScene(const char* aPath, const float3& aCamPos) {
  SceneData scene;
  std::string err;
  if (!ImportObj(aPath, scene, &err)) throw PtException(PT_ERR_INVALID_ARG, "Failed importing scene " + std::string(aPath) + ": " + err);
  LoadScene(scene, aCamPos);
}

void PathTracer::LoadScene(const SceneData& aScene, const float3& aCamPos) {
  InitRtScene(aScene);
  myCamera = Camera();
  myCamera.myPosition = aCamPos;
  myCamera.myFovDeg = 60.0f;
  myCamera.myNear = 1.0f;
  myCamera.myFar = 10000.0f;
  myCamera.myWidth = (float)myWindowWidth;
  myCamera.myHeight = (float)myWindowHeight;
  RestartAccumulation();
}

void PathTracer::InitRtScene(const SceneData& aScene) {
  std::vector<PtMeshDesc> meshes(aScene.myMeshes.size());
  for (size_t i = 0; i < meshes.size(); ++i) {
    meshes[i].vertices = aScene.myMeshes[i].myVertexData.data();
    meshes[i].num_vertices = (uint32_t)aScene.myMeshes[i].myVertexData.size();
    meshes[i].indices = aScene.myMeshes[i].myIndexData.data();
    meshes[i].num_triangles = (uint32_t)(aScene.myMeshes[i].myIndexData.size() / 3);
  }
  std::vector<PtInstanceDesc> inst(aScene.myInstances.size());
  for (size_t i = 0; i < inst.size(); ++i) {
    inst[i].myMeshIndex = aScene.myInstances[i].myMeshIndex;
    inst[i].myMaterialIndex = aScene.myInstances[i].myMaterialIndex;
    memcpy(inst[i].myTransform, aScene.myInstances[i].myTransform.data(), sizeof(float) * 16);
  }
  std::vector<PtMaterial> mats(aScene.myMaterials.size());
  for (size_t i = 0; i < mats.size(); ++i) {
    memcpy(mats[i].myEmission, aScene.myMaterials[i].myEmission, sizeof(float) * 3);
    mats[i].myColor = Encode_Unorm_RGBA(aScene.myMaterials[i].myColor);   // PathTracer.cpp:328
  }
  Check(pt_scene_create(myCtx, meshes.data(), (uint32_t)meshes.size(), inst.data(), (uint32_t)inst.size(), mats.data(),
                        (uint32_t)mats.size(), PT_SCENE_DEFAULT));
  if (!aScene.myMaterialsExt.empty()) {   // material extension read from the MTL (not in the reference)
    if (aScene.myMaterialsExt.size() != aScene.myMaterials.size()) throw PtException(PT_ERR_INVALID_ARG, "one PtMaterialExt per material");
    Check(pt_scene_set_material_ext(myCtx, aScene.myMaterialsExt.data(), (uint32_t)aScene.myMaterialsExt.size()));
    if (!aScene.myAlbedoTexture.empty())
      Check(pt_scene_set_albedo_texture(myCtx, aScene.myAlbedoTexture.data(), aScene.myAlbedoWidth, aScene.myAlbedoHeight));
  }
}

void PathTracer::UpdateOutputTexture() {
  uint32_t w = myWindowWidth, h = myWindowHeight;
  if (myHalfResRender) { w = (w + 1) / 2; h = (h + 1) / 2; }
  myOutputWidth = w;
  myOutputHeight = h;
  Check(pt_accum_reset(myCtx, w, h));
  myNumAccumulationFrames = 0;
  myAccumulationNeedsClear = false;
}

void PathTracer::RestartAccumulation() {
  myNumAccumulationFrames = 0u;
  myAccumulationNeedsClear = true;
}

void PathTracer::OnWindowResized(uint32_t w, uint32_t h) {
  myWindowWidth = w;
  myWindowHeight = h;
  myCamera.myWidth = (float)w;
  myCamera.myHeight = (float)h;
  UpdateOutputTexture();
  RestartAccumulation();
}

bool PathTracer::CameraHasChanged() const {
  float m[4][4];
  myCamera.GetViewProj(m);
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < 4; ++k)
      if (std::fabs(m[i][k] - myLastViewMat[i][k]) > 0.0001f) return true;
  return false;
}

PathTracer::SkySnapshot PathTracer::SnapshotSky() const { return SkySnapshot{mySky->mySunDir, mySky->mySunIlluminance, mySky->myAtmosphereParams}; }

void PathTracer::SkySettingsChanged() {
  mySkyDirty = true;
  RestartAccumulation();
}

void PathTracer::Update() {
  if (!myAccumulate) RestartAccumulation();   // always render frame 0 only (PathTracer.cpp:562-564)
  const SkySnapshot now = SnapshotSky();
  if (memcmp(&now, &myLastSky, sizeof(now)) != 0) {   // mySky_Imgui.HaveSettingsChanged() (:566)
    myLastSky = now;
    SkySettingsChanged();
  }
  if (CameraHasChanged()) RestartAccumulation();      // (:566-568)
  myCamera.GetViewProj(myLastViewMat);                // (:570)
}

PtRtConsts PathTracer::BuildRtConsts() const {
  PtRtConsts c;
  memset(&c, 0, sizeof(c));
  float3 v[4];
  myCamera.GetVerticesOnNearPlane(v);
  auto put = [](float* d, float3 s) { d[0] = s.x; d[1] = s.y; d[2] = s.z; };
  put(c.myNearPlaneCorner, v[0]);
  c.myAoDistance = myAoDistance;
  put(c.myXAxis, sub(v[1], v[0]));
  put(c.myYAxis, sub(v[3], v[0]));
  put(c.myCameraPos, myCamera.myPosition);
  c.myLightInstanceId = (uint32_t)myLightInstanceIdx;
  c.myNumHaltonSamples = 8192;
  put(c.myLightEmission, myLightEnabled ? mul(myLightColor, myLightStrength) : float3{0, 0, 0});
  c.mySampleSky = mySampleSky ? 1u : 0u;
  put(c.mySkyFallbackEmission, float3{mySkyFallbackIntensity, mySkyFallbackIntensity, mySkyFallbackIntensity});
  c.myPhongSpecularPower = myPhongSpecularPower;
  c.myFrameRandomSeed = myNumAccumulationFrames;
  c.myNumAccumulationFrames = myNumAccumulationFrames;
  c.myMaxRecursionDepth = (uint32_t)myMaxRecursionDepth;
  c.mySkyConsts.myAtmosphere = mySky->myAtmosphereParams;
  put(c.mySkyConsts.mySunDirection, mySky->mySunDir);
  put(c.mySkyConsts.mySunIlluminance, mySky->mySunIlluminance);
  c.mySkyConsts.myRayMarchMinMaxSPP[0] = 4.0f;   // PathTracer.cpp:707
  c.mySkyConsts.myRayMarchMinMaxSPP[1] = 14.0f;
  return c;
}

void PathTracer::RenderRT(uint32_t aNumFrames) {
  if (myAccumulationNeedsClear || !myAccumulate) {
    Check(pt_accum_reset(myCtx, myOutputWidth, myOutputHeight));   // clear + frame counter = 0 (PathTracer.cpp:668-678)
    myAccumulationNeedsClear = false;
    myNumAccumulationFrames = 0;
  }
  if (mySampleSky && mySkyDirty) {
    mySky->ComputeTranmittanceLut();   // PathTracer.cpp:580
    mySkyDirty = false;
  }
  const PtRtConsts c = BuildRtConsts();
  // PathTracer.cpp:680-683: myRenderAo swaps in the AO pipeline for the same dispatch
  Check((myRenderAo ? pt_render_ao : pt_render)(myCtx, &c, myOutputWidth, myOutputHeight, myNumAccumulationFrames, aNumFrames));
  myNumAccumulationFrames += aNumFrames;
}

void PathTracer::ReadOutput(std::vector<float>& out) {
  out.resize((size_t)myOutputWidth * myOutputHeight * 4);
  Check(pt_accum_read(myCtx, out.data()));
}

void PathTracer::Tonemap(std::vector<float>& rgba) {
  for (size_t i = 0; i + 3 < rgba.size(); i += 4) {
    for (int k = 0; k < 3; ++k) rgba[i + k] = rgba[i + k] / (rgba[i + k] + 1.0f);
    rgba[i + 3] = 1.0f;
  }
}

bool PathTracer::WritePfm(const char* aPath, bool aTonemap) {
  std::vector<float> img;
  ReadOutput(img);
  if (aTonemap) Tonemap(img);
  FILE* f = fopen(aPath, "wb");
  if (!f) return false;
  fprintf(f, "PF\n%u %u\n-1.0\n", myOutputWidth, myOutputHeight);
  std::vector<float> row((size_t)myOutputWidth * 3);
  for (uint32_t y = 0; y < myOutputHeight; ++y) {   // PFM stores the bottom row first
    const float* src = &img[(size_t)(myOutputHeight - 1 - y) * myOutputWidth * 4];
    for (uint32_t x = 0; x < myOutputWidth; ++x) { row[3 * x] = src[4 * x]; row[3 * x + 1] = src[4 * x + 1]; row[3 * x + 2] = src[4 * x + 2]; }
    fwrite(row.data(), sizeof(float), row.size(), f);
  }
  fclose(f);
  return true;
}

}  // namespace pth
