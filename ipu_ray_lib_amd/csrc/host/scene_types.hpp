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

// An indexed triangle list with 32-bit indices -> TriMeshes of at most 65 536 vertices each (`Triangle` indices are 16 bit,
// include/Primitives.hpp:21-25; the reference hands assimp's 32-bit face indices to that constructor unchecked, so a larger
// mesh comes out garbled there: splitting is this importer's answer). Greedy in triangle order: a piece is closed when
// the next triangle's new vertices would not fit; a vertex shared across the cut is duplicated. Returns the pieces added.
inline size_t appendSplitMeshes(std::vector<TriMesh>& out, const std::vector<f3>& vertices, const std::vector<f3>& normals,
                                const std::vector<uint32_t>& indices, size_t maxVertices = 65536) {
  size_t pieces = 0;
  TriMesh cur;
  std::vector<uint32_t> remap(vertices.size(), 0xFFFFFFFFu);
  std::vector<uint32_t> used;      // vertices of the current piece (to reset their remap entries)
  auto flush = [&] {
    if (cur.indices.empty()) return;
    out.push_back(std::move(cur)); ++pieces;
    cur = TriMesh();
    for (uint32_t v : used) remap[v] = 0xFFFFFFFFu;
    used.clear();
  };
  for (size_t t = 0; t + 2 < indices.size(); t += 3) {
    size_t fresh = 0;
    for (int k = 0; k < 3; ++k) {
      const uint32_t v = indices[t + k];
      if (remap[v] == 0xFFFFFFFFu && (k < 1 || indices[t] != v) && (k < 2 || indices[t + 1] != v)) ++fresh;
    }
    if (cur.vertices.size() + fresh > maxVertices) flush();
    for (int k = 0; k < 3; ++k) {
      const uint32_t v = indices[t + k];
      if (remap[v] == 0xFFFFFFFFu) {
        remap[v] = (uint32_t)cur.vertices.size(); used.push_back(v);
        cur.vertices.push_back(vertices[v]);
        if (!normals.empty()) cur.normals.push_back(normals[v]);
      }
      cur.indices.push_back((uint16_t)remap[v]);
    }
  }
  flush();
  return pieces;
}

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
