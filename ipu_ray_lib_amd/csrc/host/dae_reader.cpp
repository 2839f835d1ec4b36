// dae_reader.cpp — Collada 1.4 scene import: the product's counterpart of importScene()
// (reference src/scene_utils.cpp:152-317), which goes through assimp with
// PreTransformVertices | Triangulate | JoinIdenticalVertices | ... and then re-interprets materials.
//
// What is reproduced (rendering depends on nothing else):
//   * geometry: <triangles>, <polylist> and <polygons> primitives (polygons of more than three corners are cut into a fan
//     from their first corner - what assimp's Triangulate step does for convex faces; holes, <ph>, are refused), every
//     primitive group a mesh of its own with its own material, per-corner position/normal indices, node <matrix> transforms
//     and the Z_UP -> Y_UP root rotation baked into the vertices (normals by the inverse transpose); a group of more than
//     65 536 distinct vertices is split into several meshes (scene_types.hpp appendSplitMeshes: Triangle indices are 16 bit);
//   * camera: first <instance_camera>; horizontal fov = xfov in radians; the view matrix assimp's
//     aiCamera::GetCameraMatrix builds from the node-transformed position/lookAt/up; then the
//     reference's "camera to origin + swap handedness" v = (-p.x, p.y, -p.z) (scene_utils.cpp:300-309);
//   * material heuristics (scene_utils.cpp:214-282): diffuse -> albedo, emission (x shininess, which
//     assimp defaults to 10 when the effect has none), index_of_refraction -> ior, a name containing
//     "glass" -> Refractive, reflectivity > 0 -> Specular (checked last, so it wins).
// What is NOT pinned (assimp internals, SURVEY.md §8c iii): vertex welding/order and how meshes are
// grouped/numbered; here every <instance_geometry> becomes one mesh, in visual-scene order, and
// materials are numbered in <library_materials> order.
#include <cmath>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>

#include "scene_types.hpp"

namespace mi::host {

namespace {

struct Xml {
  std::string name, text;
  std::map<std::string, std::string> attr;
  std::vector<std::unique_ptr<Xml>> kids;

  const Xml* child(const std::string& n) const { for (auto& k : kids) if (k->name == n) return k.get(); return nullptr; }
  std::vector<const Xml*> children(const std::string& n) const { std::vector<const Xml*> r; for (auto& k : kids) if (k->name == n) r.push_back(k.get()); return r; }
  const Xml* path(std::initializer_list<const char*> p) const { const Xml* c = this; for (auto n : p) { if (!c) return nullptr; c = c->child(n); } return c; }
  std::string get(const std::string& a, const std::string& def = "") const { auto it = attr.find(a); return it == attr.end() ? def : it->second; }
  void all(const std::string& n, std::vector<const Xml*>& out) const { for (auto& k : kids) { if (k->name == n) out.push_back(k.get()); k->all(n, out); } }
};

class XmlParser {
 public:
  explicit XmlParser(const std::string& s) : s(s) {}
  std::unique_ptr<Xml> parse() {
    skipProlog();
    return element();
  }

 private:
  const std::string& s;
  size_t p = 0;
  void ws() { while (p < s.size() && isspace((unsigned char)s[p])) ++p; }
  void skipProlog() {
    for (;;) {
      ws();
      if (s.compare(p, 2, "<?") == 0) { p = s.find("?>", p); if (p == std::string::npos) fail(); p += 2; }
      else if (s.compare(p, 4, "<!--") == 0) { p = s.find("-->", p); if (p == std::string::npos) fail(); p += 3; }
      else if (s.compare(p, 2, "<!") == 0) { p = s.find('>', p); if (p == std::string::npos) fail(); p += 1; }
      else break;
    }
  }
  [[noreturn]] void fail() { throw std::runtime_error("dae: malformed XML near offset " + std::to_string(p)); }
  std::string ident() { size_t b = p; while (p < s.size() && (isalnum((unsigned char)s[p]) || s[p] == '_' || s[p] == ':' || s[p] == '-' || s[p] == '.')) ++p; return s.substr(b, p - b); }
  std::unique_ptr<Xml> element() {
    if (p >= s.size() || s[p] != '<') fail();
    ++p;
    auto e = std::make_unique<Xml>();
    e->name = ident();
    for (;;) {
      ws();
      if (p >= s.size()) fail();
      if (s[p] == '/') { p += 2; return e; }
      if (s[p] == '>') { ++p; break; }
      std::string a = ident();
      ws(); if (s[p] != '=') fail(); ++p; ws();
      const char q = s[p++];
      size_t e2 = s.find(q, p); if (e2 == std::string::npos) fail();
      e->attr[a] = s.substr(p, e2 - p);
      p = e2 + 1;
    }
    for (;;) {
      size_t lt = s.find('<', p);
      if (lt == std::string::npos) fail();
      e->text.append(s, p, lt - p);
      p = lt;
      if (s.compare(p, 4, "<!--") == 0) { p = s.find("-->", p); if (p == std::string::npos) fail(); p += 3; continue; }
      if (s.compare(p, 2, "</") == 0) { p = s.find('>', p); if (p == std::string::npos) fail(); ++p; return e; }
      e->kids.push_back(element());
    }
  }
};

std::vector<float> floats(const std::string& t) {
  std::vector<float> v;
  const char* c = t.c_str();
  char* end = nullptr;
  for (;;) {
    const float f = strtof(c, &end);
    if (end == c) break;
    v.push_back(f); c = end;
  }
  return v;
}
std::vector<uint32_t> uints(const std::string& t) {
  std::vector<uint32_t> v;
  const char* c = t.c_str();
  char* end = nullptr;
  for (;;) {
    const unsigned long u = strtoul(c, &end, 10);
    if (end == c) break;
    v.push_back((uint32_t)u); c = end;
  }
  return v;
}
std::string stripHash(const std::string& u) { return (!u.empty() && u[0] == '#') ? u.substr(1) : u; }

struct M4 { float m[16]; };   // row-major, as in the file
M4 ident() { M4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f; return r; }
M4 mul(const M4& a, const M4& b) {
  M4 r{};
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { float s = 0.f; for (int k = 0; k < 4; ++k) s += a.m[4 * i + k] * b.m[4 * k + j]; r.m[4 * i + j] = s; }
  return r;
}
f3 xformPoint(const M4& m, f3 v) {
  return mk(m.m[0] * v.x + m.m[1] * v.y + m.m[2] * v.z + m.m[3], m.m[4] * v.x + m.m[5] * v.y + m.m[6] * v.z + m.m[7],
            m.m[8] * v.x + m.m[9] * v.y + m.m[10] * v.z + m.m[11]);
}
f3 xformDir(const M4& m, f3 v) {
  return mk(m.m[0] * v.x + m.m[1] * v.y + m.m[2] * v.z, m.m[4] * v.x + m.m[5] * v.y + m.m[6] * v.z, m.m[8] * v.x + m.m[9] * v.y + m.m[10] * v.z);
}
// inverse transpose of the upper 3x3 (what assimp applies to normals in PretransformVertices)
M4 inverseTranspose3(const M4& a) {
  const float* m = a.m;
  const double c00 = (double)m[5] * m[10] - (double)m[6] * m[9], c01 = (double)m[6] * m[8] - (double)m[4] * m[10], c02 = (double)m[4] * m[9] - (double)m[5] * m[8];
  const double c10 = (double)m[2] * m[9] - (double)m[1] * m[10], c11 = (double)m[0] * m[10] - (double)m[2] * m[8], c12 = (double)m[1] * m[8] - (double)m[0] * m[9];
  const double c20 = (double)m[1] * m[6] - (double)m[2] * m[5], c21 = (double)m[2] * m[4] - (double)m[0] * m[6], c22 = (double)m[0] * m[5] - (double)m[1] * m[4];
  const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  if (det == 0.0) return a;
  M4 r = ident();   // cofactor matrix / det == inverse transpose
  r.m[0] = (float)(c00 / det); r.m[1] = (float)(c01 / det); r.m[2] = (float)(c02 / det);
  r.m[4] = (float)(c10 / det); r.m[5] = (float)(c11 / det); r.m[6] = (float)(c12 / det);
  r.m[8] = (float)(c20 / det); r.m[9] = (float)(c21 / det); r.m[10] = (float)(c22 / det);
  return r;
}

struct Source { std::vector<float> data; uint32_t stride = 3; };

struct ImportedCamera { bool found = false; float xfovDeg = 45.f; M4 world = ident(); };

}  // namespace

SceneDescription importColladaScene(const std::string& path, bool loadNormals) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("Could not load scene file.");
  std::stringstream ss; ss << f.rdbuf();
  const std::string text = ss.str();
  std::unique_ptr<Xml> root = XmlParser(text).parse();
  if (root->name != "COLLADA") throw std::runtime_error("dae: not a COLLADA document");

  SceneDescription scene;

  // ---- up axis (assimp ColladaLoader: root transform) ----
  M4 rootM = ident();
  if (const Xml* up = root->path({"asset", "up_axis"})) {
    const std::string u = up->text.substr(up->text.find_first_not_of(" \n\t"), 4);
    if (u == "Z_UP") { const float z[16] = {1, 0, 0, 0, 0, 0, 1, 0, 0, -1, 0, 0, 0, 0, 0, 1}; memcpy(rootM.m, z, sizeof z); }
    else if (u == "X_UP") { const float x[16] = {0, -1, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; memcpy(rootM.m, x, sizeof x); }
  }

  // ---- effects + materials ----
  std::map<std::string, const Xml*> effects;
  if (const Xml* lib = root->child("library_effects")) for (auto* e : lib->children("effect")) effects[e->get("id")] = e;
  std::map<std::string, uint32_t> materialIndex;
  if (const Xml* lib = root->child("library_materials")) {
    for (auto* m : lib->children("material")) {
      mi_material mat;
      memset(&mat, 0, sizeof mat);
      mat.albedo = {0.6f, 0.6f, 0.6f};       // assimp's default diffuse when the effect has none
      mat.ior = 1.f;                          // Collada::Effect default refraction index
      mat.type = 0;
      float shininess = 10.f;                 // Collada::Effect default
      float reflectivity = 0.f;
      const std::string name = m->get("name", m->get("id"));
      const Xml* ie = m->child("instance_effect");
      const Xml* eff = ie ? effects[stripHash(ie->get("url"))] : nullptr;
      if (eff) {
        std::vector<const Xml*> tech; eff->all("technique", tech);
        const Xml* shader = nullptr;
        for (auto* t : tech) for (const char* sh : {"lambert", "phong", "blinn", "constant"}) if (!shader && t->child(sh)) shader = t->child(sh);
        if (shader) {
          auto color = [&](const char* n, mi_vec3& out) { if (const Xml* c = shader->path({n, "color"})) { auto v = floats(c->text); if (v.size() >= 3) out = {v[0], v[1], v[2]}; } };
          auto scalar = [&](const char* n, float& out) { if (const Xml* c = shader->path({n, "float"})) { auto v = floats(c->text); if (!v.empty()) out = v[0]; } };
          color("diffuse", mat.albedo);
          color("emission", mat.emission);
          scalar("index_of_refraction", mat.ior);
          scalar("shininess", shininess);
          scalar("reflectivity", reflectivity);
        }
      }
      mat.emissive = (mat.emission.x != 0.f || mat.emission.y != 0.f || mat.emission.z != 0.f) ? 1 : 0;
      if (mat.emissive) { mat.emission.x *= shininess; mat.emission.y *= shininess; mat.emission.z *= shininess; }   // scene_utils.cpp:247-256
      if (name.find("glass") != std::string::npos) mat.type = 2;                                                       // :267-270
      if (reflectivity > 0.f) mat.type = 1;                                                                             // :272-281
      materialIndex[m->get("id")] = (uint32_t)scene.materials.size();
      scene.materials.push_back(mat);
    }
  }

  // ---- geometries ----
  std::map<std::string, const Xml*> geoms;
  if (const Xml* lib = root->child("library_geometries")) for (auto* g : lib->children("geometry")) geoms[g->get("id")] = g;
  std::map<std::string, float> cameraFov;
  if (const Xml* lib = root->child("library_cameras"))
    for (auto* c : lib->children("camera")) {
      float fov = 45.f;
      if (const Xml* x = c->path({"optics", "technique_common", "perspective", "xfov"})) { auto v = floats(x->text); if (!v.empty()) fov = v[0]; }
      cameraFov[c->get("id")] = fov;
    }

  ImportedCamera cam;

  // ---- visual scene walk ----
  std::function<void(const Xml*, const M4&)> walk = [&](const Xml* node, const M4& parent) {
    M4 local = ident();
    for (auto& k : node->kids) {   // transforms apply in document order
      if (k->name == "matrix") { auto v = floats(k->text); if (v.size() == 16) { M4 m; memcpy(m.m, v.data(), sizeof m.m); local = mul(local, m); } }
      else if (k->name == "translate") { auto v = floats(k->text); if (v.size() == 3) { M4 m = ident(); m.m[3] = v[0]; m.m[7] = v[1]; m.m[11] = v[2]; local = mul(local, m); } }
      else if (k->name == "scale") { auto v = floats(k->text); if (v.size() == 3) { M4 m = ident(); m.m[0] = v[0]; m.m[5] = v[1]; m.m[10] = v[2]; local = mul(local, m); } }
      else if (k->name == "rotate") {
        auto v = floats(k->text);
        if (v.size() == 4) {
          const float a = v[3] * (float)(3.14159265358979323846 / 180.0), c = cosf(a), s = sinf(a), t = 1.f - c;
          const float len = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
          const float x = v[0] / len, y = v[1] / len, z = v[2] / len;
          M4 m = ident();
          m.m[0] = t * x * x + c; m.m[1] = t * x * y - s * z; m.m[2] = t * x * z + s * y;
          m.m[4] = t * x * y + s * z; m.m[5] = t * y * y + c; m.m[6] = t * y * z - s * x;
          m.m[8] = t * x * z - s * y; m.m[9] = t * y * z + s * x; m.m[10] = t * z * z + c;
          local = mul(local, m);
        }
      }
    }
    const M4 world = mul(parent, local);
    for (auto* ic : node->children("instance_camera")) {
      if (!cam.found) { cam.found = true; cam.world = world; auto it = cameraFov.find(stripHash(ic->get("url"))); if (it != cameraFov.end()) cam.xfovDeg = it->second; }
    }
    for (auto* ig : node->children("instance_geometry")) {
      auto git = geoms.find(stripHash(ig->get("url")));
      if (git == geoms.end()) continue;
      const Xml* mesh = git->second->child("mesh");
      if (!mesh) continue;
      std::map<std::string, std::string> bind;   // symbol -> material id
      { std::vector<const Xml*> ims; ig->all("instance_material", ims); for (auto* im : ims) bind[im->get("symbol")] = stripHash(im->get("target")); }
      std::map<std::string, Source> sources;
      for (auto* s : mesh->children("source")) {
        Source src;
        if (const Xml* fa = s->child("float_array")) src.data = floats(fa->text);
        if (const Xml* acc = s->path({"technique_common", "accessor"})) src.stride = (uint32_t)std::stoul(acc->get("stride", "3"));
        sources[s->get("id")] = std::move(src);
      }
      std::string posSource;
      if (const Xml* v = mesh->child("vertices")) for (auto* in : v->children("input")) if (in->get("semantic") == "POSITION") posSource = stripHash(in->get("source"));
      const M4 nrmM = inverseTranspose3(world);
      std::vector<const Xml*> prims = mesh->children("triangles");
      for (auto* pl : mesh->children("polylist")) prims.push_back(pl);
      for (auto* pl : mesh->children("polygons")) prims.push_back(pl);
      for (auto* tri : prims) {
        uint32_t stride = 0, vOff = 0, nOff = ~0u;
        std::string nSource;
        for (auto* in : tri->children("input")) {
          const uint32_t off = (uint32_t)std::stoul(in->get("offset", "0"));
          stride = std::max(stride, off + 1);
          if (in->get("semantic") == "VERTEX") vOff = off;
          if (in->get("semantic") == "NORMAL") { nOff = off; nSource = stripHash(in->get("source")); }
        }
        if (!stride) continue;
        // the primitive group as a list of polygons: `idx` holds `stride` indices per corner, poly[i] corners per polygon
        std::vector<uint32_t> idx, poly;
        if (tri->name == "polygons") {
          if (tri->child("ph")) throw std::runtime_error("dae: polygons with holes (<ph>) are not supported");
          for (auto* pe : tri->children("p")) { const std::vector<uint32_t> one = uints(pe->text); poly.push_back((uint32_t)(one.size() / stride)); idx.insert(idx.end(), one.begin(), one.begin() + one.size() / stride * stride); }
        } else {
          const Xml* pe = tri->child("p");
          if (!pe) continue;
          idx = uints(pe->text);
          if (tri->name == "polylist") { if (const Xml* vc = tri->child("vcount")) poly = uints(vc->text); }
          if (poly.empty()) poly.assign(idx.size() / stride / 3, 3u);
        }
        const Source& ps = sources[posSource];
        const Source* ns = (nOff != ~0u && loadNormals) ? &sources[nSource] : nullptr;
        std::vector<f3> verts, norms;
        std::vector<uint32_t> tris;
        std::map<std::pair<uint32_t, uint32_t>, uint32_t> weld;   // (position index, normal index) -> vertex
        auto corner = [&](size_t c) -> uint32_t {
          const uint32_t pi = idx[c * stride + vOff], ni = ns ? idx[c * stride + nOff] : 0u;
          auto key = std::make_pair(pi, ni);
          auto it = weld.find(key);
          if (it == weld.end()) {
            if ((size_t)pi * ps.stride + 2 >= ps.data.size()) throw std::runtime_error("dae: position index out of range");
            const f3 pos = mk(ps.data[pi * ps.stride], ps.data[pi * ps.stride + 1], ps.data[pi * ps.stride + 2]);
            verts.push_back(xformPoint(world, pos));
            if (ns) {
              if ((size_t)ni * ns->stride + 2 >= ns->data.size()) throw std::runtime_error("dae: normal index out of range");
              const f3 n = mk(ns->data[ni * ns->stride], ns->data[ni * ns->stride + 1], ns->data[ni * ns->stride + 2]);
              norms.push_back(normalized(xformDir(nrmM, n)));
            }
            it = weld.emplace(key, (uint32_t)(verts.size() - 1)).first;
          }
          return it->second;
        };
        size_t at = 0;
        for (uint32_t cnt : poly) {
          if (at + cnt > idx.size() / stride) throw std::runtime_error("dae: polygon list runs past its index data");
          if (cnt >= 3) {
            const uint32_t c0 = corner(at);
            uint32_t prev = corner(at + 1);
            for (uint32_t k = 2; k < cnt; ++k) { const uint32_t cur = corner(at + k); tris.push_back(c0); tris.push_back(prev); tris.push_back(cur); prev = cur; }
          }
          at += cnt;       // (points and lines - fewer than three corners - are dropped, as assimp's SortByPType does for the reference)
        }
        if (tris.empty()) continue;
        const std::string matId = bind.count(tri->get("material")) ? bind[tri->get("material")] : tri->get("material");
        const size_t pieces = appendSplitMeshes(scene.meshes, verts, norms, tris);
        for (size_t k = 0; k < pieces; ++k) scene.matIDs.push_back(materialIndex.count(matId) ? materialIndex[matId] : 0u);
      }
    }
    for (auto* ch : node->children("node")) walk(ch, world);
  };

  const Xml* vscenes = root->child("library_visual_scenes");
  if (!vscenes || !vscenes->child("visual_scene")) throw std::runtime_error("dae: no visual scene");
  for (auto* n : vscenes->child("visual_scene")->children("node")) walk(n, rootM);

  if (!cam.found) throw std::runtime_error("No camera found in scene file.");   // scene_utils.cpp:177-180
  if (scene.materials.empty()) { mi_material m; memset(&m, 0, sizeof m); m.albedo = {0.6f, 0.6f, 0.6f}; m.ior = 1.f; scene.materials.push_back(m); }
  scene.horizontalFov = cam.xfovDeg * (float)(3.14159265358979323846 / 180.0);

  // aiCamera::GetCameraMatrix on the node-transformed camera (position 0, lookAt -z, up +y)
  const f3 pos = xformPoint(cam.world, mk(0, 0, 0));
  const f3 zaxis = normalized(xformDir(cam.world, mk(0, 0, -1)));
  const f3 yaxis = normalized(xformDir(cam.world, mk(0, 1, 0)));
  const f3 xaxis = normalized(cross(yaxis, zaxis));
  M4 view = ident();
  view.m[0] = xaxis.x; view.m[1] = xaxis.y; view.m[2] = xaxis.z; view.m[3] = -dot(xaxis, pos);
  view.m[4] = yaxis.x; view.m[5] = yaxis.y; view.m[6] = yaxis.z; view.m[7] = -dot(yaxis, pos);
  view.m[8] = zaxis.x; view.m[9] = zaxis.y; view.m[10] = zaxis.z; view.m[11] = -dot(zaxis, pos);
  for (auto& m : scene.meshes) {
    for (auto& v : m.vertices) { const f3 p = xformPoint(view, v); v = mk(-p.x, p.y, -p.z); }
    for (auto& n : m.normals) { const f3 p = xformDir(view, n); n = mk(-p.x, p.y, -p.z); }
  }
  return scene;
}

}  // namespace mi::host
