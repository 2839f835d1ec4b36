// scene_builtin.cpp — the reference CLI's built-in scenes and the array packing that feeds the
// renderers. The geometry constants are DATA transcribed from the reference so both render the
// same scene (src/scene_utils.cpp:319-597); packing follows src/app_utils.cpp:291-345.
#include <cmath>
#include <stdexcept>

#include "scene_types.hpp"

namespace mi::host {

const float* sinTable() {
  static const uint32_t bits[92] = {MI_SIN_TABLE_BITS};
  static float tbl[92];
  static bool init = false;
  if (!init) { memcpy(tbl, bits, sizeof tbl); init = true; }
  return tbl;
}

void TriMesh::addQuad(f3 a, f3 b, f3 c, f3 d) {
  const uint16_t base = (uint16_t)vertices.size();
  vertices.push_back(a); vertices.push_back(b); vertices.push_back(c); vertices.push_back(d);
  const uint16_t tri[6] = {0, 1, 2, 2, 3, 0};
  for (uint16_t t : tri) indices.push_back((uint16_t)(base + t));
}

namespace {

struct Quad { float v[4][3]; };

void addQuads(TriMesh& m, const Quad* q, size_t n) {
  for (size_t i = 0; i < n; ++i)
    m.addQuad(mk(q[i].v[0][0], q[i].v[0][1], q[i].v[0][2]), mk(q[i].v[1][0], q[i].v[1][1], q[i].v[1][2]),
              mk(q[i].v[2][0], q[i].v[2][1], q[i].v[2][2]), mk(q[i].v[3][0], q[i].v[3][1], q[i].v[3][2]));
}

// Cornell box measurements (scene_utils.cpp:319-456). Mesh order: light, white (floor, ceiling,
// back wall), red (left wall), green (right wall), short block, tall block.
const Quad kLight[] = {{{{343, 548.7998f, 227}, {343, 548.7998f, 332}, {213, 548.7998f, 332}, {213, 548.7998f, 227}}}};
const Quad kWhite[] = {
    {{{552.8f, 0, 0}, {0, 0, 0}, {0, 0, 559.2f}, {549.6f, 0, 559.2f}}},
    {{{556, 548.8f, 0}, {556, 548.8f, 559.2f}, {0, 548.8f, 559.2f}, {0, 548.8f, 0}}},
    {{{549.6f, 0, 559.2f}, {0, 0, 559.2f}, {0, 548.8f, 559.2f}, {556, 548.8f, 559.2f}}}};
const Quad kGreen[] = {{{{0, 0, 559.2f}, {0, 0, 0}, {0, 548.8f, 0}, {0, 548.8f, 559.2f}}}};
const Quad kRed[] = {{{{552.8f, 0, 0}, {549.6f, 0, 559.2f}, {556, 548.8f, 559.2f}, {556, 548.8f, 0}}}};
const Quad kShort[] = {
    {{{130, 165, 65}, {82, 165, 225}, {240, 165, 272}, {290, 165, 114}}},
    {{{290, 0, 114}, {290, 165, 114}, {240, 165, 272}, {240, 0, 272}}},
    {{{130, 0, 65}, {130, 165, 65}, {290, 165, 114}, {290, 0, 114}}},
    {{{82, 0, 225}, {82, 165, 225}, {130, 165, 65}, {130, 0, 65}}},
    {{{240, 0, 272}, {240, 165, 272}, {82, 165, 225}, {82, 0, 225}}}};
const Quad kTall[] = {
    {{{423, 330, 247}, {265, 330, 296}, {314, 330, 456}, {472, 330, 406}}},
    {{{423, 0, 247}, {423, 330, 247}, {472, 330, 406}, {472, 0, 406}}},
    {{{472, 0, 406}, {472, 330, 406}, {314, 330, 456}, {314, 0, 456}}},
    {{{314, 0, 456}, {314, 330, 456}, {265, 330, 296}, {265, 0, 296}}},
    {{{265, 0, 296}, {265, 330, 296}, {423, 330, 247}, {423, 0, 247}}}};

mi_material material(f3 albedo, f3 emission, int type) {
  mi_material m;
  memset(&m, 0, sizeof m);
  m.albedo = {albedo.x, albedo.y, albedo.z};
  m.ior = 1.52f;                                   // Material.hpp:27
  m.emission = {emission.x, emission.y, emission.z};
  m.type = type;
  m.emissive = (emission.x != 0.f || emission.y != 0.f || emission.z != 0.f) ? 1 : 0;
  return m;
}

// importMesh (scene_utils.cpp:102-150): each imported mesh is scaled so its own bounding-box
// diagonal is 175, turned to face the camera and put on top of the short block.
void placeImportedMesh(TriMesh& m) {
  const Bounds b = m.bounds();
  const f3 diag = b.hi - b.lo;
  const float scale = 175.f / sqrtf(sqnorm(diag));
  for (auto& v : m.vertices) {
    v.x = -v.x;
    v.z = -v.z;
    v = v * scale;
    v = v + mk(210.f, 165.f, 160.f);
  }
  for (auto& n : m.normals) { n.x = -n.x; n.z = -n.z; }
}

}  // namespace

SceneDescription makeCornellBoxScene(const std::string& meshFile, bool boxOnly) {
  SceneDescription s;
  s.meshes.resize(6);
  addQuads(s.meshes[0], kLight, 1);
  addQuads(s.meshes[1], kWhite, 3);
  addQuads(s.meshes[2], kRed, 1);
  addQuads(s.meshes[3], kGreen, 1);
  addQuads(s.meshes[4], kShort, 5);
  addQuads(s.meshes[5], kTall, 5);

  if (!boxOnly) {
    s.spheres.push_back({450.f, 37.f, 90.f, 37.f});
    s.spheres.push_back({350.f, 37.f, 90.f, 37.f});
    s.discs.push_back({1.f, 0.f, 0.f, 60.f, 0.0002f, 300.f, 250.f});
    if (meshFile.empty()) throw std::runtime_error("scene 'box' needs the monkey-bust mesh file (assets/monkey_bust.glb)");
    for (auto& m : loadGlbMeshes(meshFile, /*loadNormals=*/false)) {   // scene_utils.cpp:118
      placeImportedMesh(m);
      s.meshes.push_back(std::move(m));
    }
  }

  // Camera to the origin and handedness flip (scene_utils.cpp:478-511)
  const f3 cam = mk(278.f, 273.f, -800.f);
  for (auto& m : s.meshes)
    for (auto& v : m.vertices) { v = v - cam; v.x = -v.x; v.z = -v.z; }
  for (auto& sp : s.spheres) {
    sp.x -= cam.x; sp.y -= cam.y; sp.z -= cam.z;
    sp.x = -sp.x; sp.z = -sp.z;
  }
  for (auto& d : s.discs) {
    d.cx -= cam.x; d.cy -= cam.y; d.cz -= cam.z;
    d.cx = -d.cx; d.cz = -d.cz;
    d.nx = -d.nx; d.nz = -d.nz;
  }

  const f3 black = mk(0, 0, 0), red = mk(.66f, 0, 0), green = mk(0, .48f, 0), blue = mk(.4f, .4f, .85f);
  const f3 blueLight = mk(.4f, .7f, .92f) * 2.f;
  const f3 white = mk(.75f, .75f, .75f), grey = mk(.4f, .4f, .4f), lightR = mk(.78f, .78f, .78f);
  const f3 lightE = mk((100.f * 15.6f + 100.f * 18.4f) / 255.f, (100.f * 8.f + 74.5f * 15.6f) / 255.f, (57.3f * 8.f) / 255.f);
  s.materials = {material(white, black, 0), material(red, black, 0), material(green, black, 0),
                 material(blue, black, 2),  material(lightR, lightE, 0), material(grey, black, 1),
                 material(blue, blueLight, 0), material(blue, black, 0)};
  // light, white parts, left wall, right wall, short box, tall box | loaded meshes | sphere, sphere, disc
  s.matIDs = {4, 0, 1, 2, 0, 5, 0, 0, 3, 7, 6};
  const size_t prims = s.meshes.size() + s.spheres.size() + s.discs.size();
  if (s.matIDs.size() < prims) throw std::logic_error("All primitives must be assigned a material.");
  s.horizontalFov = (float)(3.14159265358979323846264338327950288 / 4.0);
  return s;
}

SceneDescription makePrimitiveScene() {
  SceneDescription s;
  s.horizontalFov = (float)(3.14159265358979323846264338327950288 / 2.0);
  s.spheres = {{-1.8575f, -0.98714f, -3.6f, 0.6f},
               {0.74795f, -0.55f, -4.3816f, 1.05f},
               {1.9929f, -1.08666f, (float)-3.23, 0.5f},
               {(float)-0.19931, -1.183f, -2.75f, 0.4f},
               {(float)-0.19931, -1.183f, -2.75f, 0.4010f}};
  s.discs = {{0.f, 1.f, 0.f, 3.5f, 0.f, -1.6f, -5.22f}};
  const f3 zero = mk(0, 0, 0), one = mk(1, 1, 1);
  s.materials = {material(mk(1.f, .89f, .55f), zero, 0), material(one, zero, 1),
                 material(mk(.75f, .75f, .75f), zero, 2), material(mk(.8f, .06f, .391f), zero, 0),
                 material(one, zero, 2),                  material(mk(.98f, .76f, .66f), zero, 0)};
  s.matIDs = {0, 1, 2, 3, 4, 5};
  return s;
}

// BASELINE config 5 names `assets/monkey_bust.glb` + the NIF environment. The reference cannot load
// that file as a scene (it has no camera, scene_utils.cpp:177-180), so the scene is defined here: the
// bust placed exactly as importMesh places it in the box scene (scene_utils.cpp:122-143), seen from the
// Cornell camera, with NO box around it: every path ends in the environment. Both meshes are white diffuse.
SceneDescription makeMonkeyScene(const std::string& meshFile) {
  if (meshFile.empty()) throw std::runtime_error("scene 'monkey' needs the monkey-bust mesh file (assets/monkey_bust.glb)");
  SceneDescription s;
  for (auto& m : loadGlbMeshes(meshFile, /*loadNormals=*/false)) { placeImportedMesh(m); s.meshes.push_back(std::move(m)); }
  const f3 cam = mk(278.f, 273.f, -800.f);
  for (auto& m : s.meshes)
    for (auto& v : m.vertices) { v = v - cam; v.x = -v.x; v.z = -v.z; }
  s.materials = {material(mk(.75f, .75f, .75f), mk(0, 0, 0), 0)};
  s.matIDs.assign(s.meshes.size(), 0u);
  s.horizontalFov = (float)(3.14159265358979323846264338327950288 / 4.0);
  return s;
}

// buildSceneData (src/app_utils.cpp:291-371) + makeBuildPrimitivesForEmbree (:145-188)
PackedScene packScene(const SceneDescription& scene) {
  PackedScene d;
  for (const auto& m : scene.meshes) {
    d.meshInfo.push_back({(uint32_t)(d.meshTris.size() / 3), (uint32_t)d.meshVerts.size(),
                          (uint32_t)(m.indices.size() / 3), (uint32_t)m.vertices.size()});
    d.meshTris.insert(d.meshTris.end(), m.indices.begin(), m.indices.end());
    for (auto& v : m.vertices) d.meshVerts.push_back({v.x, v.y, v.z});
    for (auto& n : m.normals) d.meshNormals.push_back({n.x, n.y, n.z});
  }
  for (size_t i = 0; i < scene.meshes.size(); ++i) d.geometry.push_back({(uint16_t)i, 0, 0});
  for (size_t i = 0; i < scene.spheres.size(); ++i) d.geometry.push_back({(uint16_t)i, 1, 0});
  for (size_t i = 0; i < scene.discs.size(); ++i) d.geometry.push_back({(uint16_t)i, 2, 0});
  d.materials = scene.materials;
  d.matIDs = scene.matIDs;
  d.spheres = scene.spheres;
  d.discs = scene.discs;
  d.horizontalFov = scene.horizontalFov;

  // One build primitive per triangle, one per sphere/disc; geomID = position in `geometry`.
  std::vector<BuildPrim> prims;
  for (size_t g = 0; g < d.geometry.size(); ++g) {
    const auto& ref = d.geometry[g];
    if (ref.type == 0) {
      const auto& m = scene.meshes[ref.index];
      for (uint32_t t = 0; t < m.indices.size() / 3; ++t) {
        BuildPrim bp; bp.geomID = (uint16_t)g; bp.primID = t;
        bp.box.grow(m.vertices[m.indices[3 * t]]);
        bp.box.grow(m.vertices[m.indices[3 * t + 1]]);
        bp.box.grow(m.vertices[m.indices[3 * t + 2]]);
        prims.push_back(bp);
      }
    } else if (ref.type == 1) {
      const auto& s = scene.spheres[ref.index];
      BuildPrim bp; bp.geomID = (uint16_t)g; bp.primID = 0;
      bp.box.lo = mk(s.x - s.radius, s.y - s.radius, s.z - s.radius);   // Primitives.hpp:53-56
      bp.box.hi = mk(s.x + s.radius, s.y + s.radius, s.z + s.radius);
      prims.push_back(bp);
    } else {
      const auto& c = scene.discs[ref.index];
      BuildPrim bp; bp.geomID = (uint16_t)g; bp.primID = 0;
      bp.box.lo = mk(c.cx - c.r, c.cy - c.r, c.cz - c.r);               // Primitives.hpp:77-81
      bp.box.hi = mk(c.cx + c.r, c.cy + c.r, c.cz + c.r);
      prims.push_back(bp);
    }
  }
  buildCompactBvh(prims, d.bvhNodes, d.bvhMaxDepth);
  return d;
}

}  // namespace mi::host
