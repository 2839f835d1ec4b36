// glb_reader.cpp — minimal glTF-2.0 binary (.glb) mesh reader.
//
// Stands in for the assimp import the reference uses for assets/monkey_bust.glb
// (src/scene_utils.cpp:102-150, flags PreTransformVertices | Triangulate | ...): every mesh
// primitive of the default scene - a mesh may have several - becomes one TriMesh with its node's TRS transform baked into
// the positions (and the rotation into the normals); indices of 8, 16 or 32 bits; a primitive of more than 65 536
// vertices is split into several TriMeshes (scene_types.hpp appendSplitMeshes: Triangle indices are 16 bit). Vertex order is the file's order (assimp's
// JoinIdenticalVertices re-indexing is not reproduced; only triangle vertex POSITIONS matter to
// the renderer and those are unchanged by welding).
#include <cstring>
#include <fstream>
#include <sstream>

#include "json_min.hpp"
#include "scene_types.hpp"

namespace mi::host {

namespace {

struct Mat4 { float m[16]; };   // row-major

Mat4 identity() { Mat4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f; return r; }

Mat4 mul(const Mat4& a, const Mat4& b) {
  Mat4 r{};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float s = 0.f;
      for (int k = 0; k < 4; ++k) s += a.m[4 * i + k] * b.m[4 * k + j];
      r.m[4 * i + j] = s;
    }
  return r;
}

Mat4 nodeLocalMatrix(const json::Value& node) {
  if (node.has("matrix")) {   // column-major in glTF
    Mat4 r{};
    for (int c = 0; c < 4; ++c) for (int rI = 0; rI < 4; ++rI) r.m[4 * rI + c] = (float)node.at("matrix").at(4 * c + rI).number();
    return r;
  }
  float t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
  if (node.has("translation")) for (int i = 0; i < 3; ++i) t[i] = (float)node.at("translation").at(i).number();
  if (node.has("rotation")) for (int i = 0; i < 4; ++i) q[i] = (float)node.at("rotation").at(i).number();
  if (node.has("scale")) for (int i = 0; i < 3; ++i) s[i] = (float)node.at("scale").at(i).number();
  const float x = q[0], y = q[1], z = q[2], w = q[3];
  Mat4 r = identity();
  r.m[0] = (1 - 2 * (y * y + z * z)) * s[0]; r.m[1] = (2 * (x * y - z * w)) * s[1]; r.m[2] = (2 * (x * z + y * w)) * s[2]; r.m[3] = t[0];
  r.m[4] = (2 * (x * y + z * w)) * s[0]; r.m[5] = (1 - 2 * (x * x + z * z)) * s[1]; r.m[6] = (2 * (y * z - x * w)) * s[2]; r.m[7] = t[1];
  r.m[8] = (2 * (x * z - y * w)) * s[0]; r.m[9] = (2 * (y * z + x * w)) * s[1]; r.m[10] = (1 - 2 * (x * x + y * y)) * s[2]; r.m[11] = t[2];
  return r;
}

struct Glb {
  json::ValuePtr doc;
  std::vector<uint8_t> bin;
};

Glb readGlb(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("Could not open mesh file '" + path + "'");
  std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  auto u32 = [&](size_t off) { uint32_t v; if (off + 4 > data.size()) throw std::runtime_error("glb: truncated"); memcpy(&v, &data[off], 4); return v; };
  if (data.size() < 20 || u32(0) != 0x46546C67u) throw std::runtime_error("glb: bad magic in '" + path + "'");
  if (u32(4) != 2) throw std::runtime_error("glb: only version 2 is supported");
  Glb g;
  size_t off = 12;
  while (off + 8 <= data.size()) {
    const uint32_t len = u32(off), type = u32(off + 4);
    off += 8;
    if (off + len > data.size()) throw std::runtime_error("glb: chunk overruns file");
    if (type == 0x4E4F534Au) g.doc = json::parse(std::string((const char*)&data[off], len));
    else if (type == 0x004E4942u && g.bin.empty()) g.bin.assign(data.begin() + off, data.begin() + off + len);
    off += (len + 3u) & ~3u;
  }
  if (!g.doc) throw std::runtime_error("glb: no JSON chunk");
  return g;
}

struct AccessorView { const uint8_t* base; size_t count; size_t stride; int componentType; int numComp; };

AccessorView accessor(const Glb& g, size_t index) {
  const auto& acc = g.doc->at("accessors").at(index);
  const auto& bv = g.doc->at("bufferViews").at((size_t)acc.at("bufferView").number());
  if ((size_t)bv.at("buffer").number() != 0) throw std::runtime_error("glb: only the embedded buffer is supported");
  AccessorView v;
  v.componentType = (int)acc.at("componentType").number();
  const std::string& type = acc.at("type").string();
  v.numComp = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : 0;
  if (!v.numComp) throw std::runtime_error("glb: unsupported accessor type " + type);
  const size_t compSize = (v.componentType == 5126 || v.componentType == 5125) ? 4 : (v.componentType == 5123 || v.componentType == 5122) ? 2 : 1;
  size_t off = (bv.has("byteOffset") ? (size_t)bv.at("byteOffset").number() : 0) + (acc.has("byteOffset") ? (size_t)acc.at("byteOffset").number() : 0);
  v.stride = bv.has("byteStride") ? (size_t)bv.at("byteStride").number() : compSize * v.numComp;
  v.count = (size_t)acc.at("count").number();
  if (off + (v.count ? (v.count - 1) * v.stride + compSize * v.numComp : 0) > g.bin.size()) throw std::runtime_error("glb: accessor overruns buffer");
  v.base = g.bin.data() + off;
  return v;
}

void collect(const Glb& g, size_t nodeIndex, const Mat4& parent, bool loadNormals, std::vector<TriMesh>& out, std::vector<uint8_t>& onPath) {
  // the node hierarchy must be a forest: a node that is its own ancestor would recurse for ever
  if (nodeIndex >= onPath.size()) throw std::runtime_error("glb: node index out of range");
  if (onPath[nodeIndex]) throw std::runtime_error("glb: node hierarchy contains a cycle");
  onPath[nodeIndex] = 1;
  const auto& node = g.doc->at("nodes").at(nodeIndex);
  const Mat4 world = mul(parent, nodeLocalMatrix(node));
  if (node.has("mesh")) {
    const auto& mesh = g.doc->at("meshes").at((size_t)node.at("mesh").number());
    for (size_t p = 0; p < mesh.at("primitives").size(); ++p) {
      const auto& prim = mesh.at("primitives").at(p);
      if (prim.has("mode") && (int)prim.at("mode").number() != 4) continue;   // triangles only (SortByPType)
      std::vector<f3> verts, norms;
      std::vector<uint32_t> tris;
      const AccessorView pos = accessor(g, (size_t)prim.at("attributes").at("POSITION").number());
      if (pos.componentType != 5126 || pos.numComp != 3) throw std::runtime_error("glb: POSITION must be float VEC3");
      verts.reserve(pos.count);
      for (size_t i = 0; i < pos.count; ++i) {
        float v[3]; memcpy(v, pos.base + i * pos.stride, 12);
        const float* m = world.m;
        verts.push_back(mk(m[0] * v[0] + m[1] * v[1] + m[2] * v[2] + m[3],
                           m[4] * v[0] + m[5] * v[1] + m[6] * v[2] + m[7],
                           m[8] * v[0] + m[9] * v[1] + m[10] * v[2] + m[11]));
      }
      if (loadNormals && prim.at("attributes").has("NORMAL")) {
        const AccessorView nrm = accessor(g, (size_t)prim.at("attributes").at("NORMAL").number());
        if (nrm.componentType != 5126 || nrm.numComp != 3) throw std::runtime_error("glb: NORMAL must be float VEC3");
        if (nrm.count != pos.count) throw std::runtime_error("glb: NORMAL count differs from POSITION count");
        for (size_t i = 0; i < nrm.count; ++i) {
          float v[3]; memcpy(v, nrm.base + i * nrm.stride, 12);
          const float* m = world.m;   // rigid node transforms only: rotation part applies to normals
          norms.push_back(normalized(mk(m[0] * v[0] + m[1] * v[1] + m[2] * v[2],
                                        m[4] * v[0] + m[5] * v[1] + m[6] * v[2],
                                        m[8] * v[0] + m[9] * v[1] + m[10] * v[2])));
        }
      }
      if (prim.has("indices")) {
        const AccessorView idx = accessor(g, (size_t)prim.at("indices").number());
        tris.reserve(idx.count);
        for (size_t i = 0; i < idx.count; ++i) {
          uint32_t v = 0;
          const uint8_t* src = idx.base + i * idx.stride;
          if (idx.componentType == 5123) { uint16_t t; memcpy(&t, src, 2); v = t; }
          else if (idx.componentType == 5125) { memcpy(&v, src, 4); }
          else if (idx.componentType == 5121) { v = *src; }
          else throw std::runtime_error("glb: unsupported index type");
          if (v >= pos.count) throw std::runtime_error("glb: triangle index " + std::to_string(v) + " out of range (mesh has " + std::to_string(pos.count) + " vertices)");
          tris.push_back(v);
        }
      } else {
        for (size_t i = 0; i < pos.count; ++i) tris.push_back((uint32_t)i);
      }
      if (tris.size() % 3) throw std::runtime_error("Only triangle meshes are supported.");
      if (pos.count <= 65536) {
        // (the common case keeps the file's vertex order and its unreferenced vertices, as before)
        TriMesh tm;
        tm.vertices = std::move(verts); tm.normals = std::move(norms);
        tm.indices.reserve(tris.size());
        for (uint32_t v : tris) tm.indices.push_back((uint16_t)v);
        out.push_back(std::move(tm));
      } else {
        appendSplitMeshes(out, verts, norms, tris);
      }
    }
  }
  if (node.has("children"))
    for (size_t c = 0; c < node.at("children").size(); ++c)
      collect(g, (size_t)node.at("children").at(c).number(), world, loadNormals, out, onPath);
  onPath[nodeIndex] = 0;
}

}  // namespace

std::vector<TriMesh> loadGlbMeshes(const std::string& path, bool loadNormals) {
  const Glb g = readGlb(path);
  std::vector<TriMesh> meshes;
  const size_t sceneIndex = g.doc->has("scene") ? (size_t)g.doc->at("scene").number() : 0;
  const auto& roots = g.doc->at("scenes").at(sceneIndex).at("nodes");
  std::vector<uint8_t> onPath(g.doc->at("nodes").size(), 0);
  for (size_t i = 0; i < roots.size(); ++i) collect(g, (size_t)roots.at(i).number(), identity(), loadNormals, meshes, onPath);
  return meshes;
}

}  // namespace mi::host
