// scene_types.hpp — host-side scene description (the product's counterpart of the reference's
// SceneDescription / SceneData, include/scene_utils.hpp:30-44 and include/Scene.hpp:36-48).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/mi_raylib.h"
#include "../ray_math.h"

namespace mi::host {

struct Bounds {
  f3 lo{kInf, kInf, kInf};
  f3 hi{-kInf, -kInf, -kInf};
  void grow(f3 p) {
    lo.x = p.x < lo.x ? p.x : lo.x; lo.y = p.y < lo.y ? p.y : lo.y; lo.z = p.z < lo.z ? p.z : lo.z;
    hi.x = p.x > hi.x ? p.x : hi.x; hi.y = p.y > hi.y ? p.y : hi.y; hi.z = p.z > hi.z ? p.z : hi.z;
  }
  void grow(const Bounds& b) { grow(b.lo); grow(b.hi); }
};

struct TriMesh {
  std::vector<uint16_t> indices;   // 3 per triangle
  std::vector<f3> vertices;
  std::vector<f3> normals;         // empty, or one per vertex
  Bounds bounds() const { Bounds b; for (auto& v : vertices) b.grow(v); return b; }
  void addQuad(f3 a, f3 b, f3 c, f3 d);   // two triangles (0,1,2) (2,3,0): scene_utils.cpp:31-46
};

struct SceneDescription {
  std::vector<TriMesh> meshes;
  std::vector<mi_sphere> spheres;
  std::vector<mi_disc> discs;
  std::vector<mi_material> materials;
  std::vector<uint32_t> matIDs;
  float horizontalFov = 0.f;
};

// The packed arrays handed to the renderers (SceneData, Scene.hpp:36-48)
struct PackedScene {
  std::vector<mi_geom_ref> geometry;
  std::vector<mi_mesh_info> meshInfo;
  std::vector<uint16_t> meshTris;
  std::vector<mi_vec3> meshVerts;
  std::vector<mi_vec3> meshNormals;
  std::vector<uint32_t> matIDs;
  std::vector<mi_material> materials;
  std::vector<mi_bvh_node> bvhNodes;
  uint32_t bvhMaxDepth = 0;
  std::vector<mi_sphere> spheres;
  std::vector<mi_disc> discs;
  float horizontalFov = 0.f;
};

struct BuildPrim { Bounds box; uint16_t geomID; uint32_t primID; };

// bvh_sah.cpp
void buildCompactBvh(const std::vector<BuildPrim>& prims, std::vector<mi_bvh_node>& nodes, uint32_t& maxDepth);

// glb_reader.cpp: meshes of a glTF-binary file with node transforms baked in, file order kept
std::vector<TriMesh> loadGlbMeshes(const std::string& path, bool loadNormals);

// scene_builtin.cpp
SceneDescription makeCornellBoxScene(const std::string& meshFile, bool boxOnly);
SceneDescription makePrimitiveScene();
SceneDescription makeMonkeyScene(const std::string& meshFile);   // monkey bust in an open environment (BASELINE config 5)

// dae_reader.cpp: importScene() for Collada files
SceneDescription importColladaScene(const std::string& path, bool loadNormals);
PackedScene packScene(const SceneDescription& scene);

const float* sinTable();   // 92-entry table for sincos_deg_table on the host

}  // namespace mi::host
