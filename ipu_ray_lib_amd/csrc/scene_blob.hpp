// scene_blob.hpp — the reference's serialised-scene wire format, host side, header-only.
//
// The reference moves the whole scene to the device as ONE byte stream written by Serialiser<16>
// (include/serialisation/Serialiser.hpp:19-60, serialisation.hpp:11-53; caller src/IpuScene.cpp:31,51-53)
// and read back IN PLACE by Deserialiser<16> (Deserialiser.hpp:13-80, deserialisation.hpp:21-59), so
// arrays inside the blob are aliased, not copied. Format, restated:
//   * an object of alignment A is preceded by (A - (BaseAlign + offset) % A) % A padding bytes, where
//     offset is its distance from the first byte and the first byte is assumed BaseAlign-aligned;
//   * a fundamental value is its little-endian bytes; an array is a u32 element count followed by the
//     elements (padded to the element alignment) — raw struct bytes, no per-field encoding;
//   * SceneRef = geometry, meshInfo, meshTris, meshVerts, meshNormals, matIDs, materials, bvhNodes
//     (arrays, in this order), then maxLeafDepth u32, imageWidth f32, imageHeight f32, fovRadians f32,
//     antiAliasScale f32, maxPathLength u32, rouletteStartDepth u32, samplesPerPixel u32.
//     rngSeed, the crop window and pathTrace are NOT in the blob (they travel as separate tensors).
// Element alignments are alignof() of the reference structs (Serialiser::write(const T*, n) pads to alignof(T),
// Serialiser.hpp:33-60; deserialiseArrayRef<T> skips the same, deserialisation.hpp:31-38):
//   GeomRef 2 (Scene.hpp:29-34: u16 + 2 x u8)        MeshInfo 4 (Mesh.hpp:15-20: 4 x u32)
//   Triangle 2 (Primitives.hpp:21-25: packed, aligned(alignof(u16)))
//   Vec3fa 4 (embree_utils/geometry.hpp:25-27: VEC3_ALIGN 4)      u32 4
//   Material 4 (Material.hpp:8-35: Vec3fa, float, enum, bool)
//   CompactBVH2Node 8 (CompactBVH2Node.hpp:52-53: __attribute__((aligned(8))), 24 bytes) - so the node array
//   is preceded by 4 pad bytes whenever the byte after its u32 count falls on (BaseAlign + offset) % 8 == 4.
// Parity note: the reference's Serialiser cannot be compiled here (boost::alignment + Eigen::half are
// absent), so this format is pinned by the properties its unit tests check (tests/test.cpp:38-237),
// restated in tests/test_scene_blob.py, and by an independent numpy packer in that test.
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "../../include/mi_raylib.h"

namespace mi::blob {

constexpr uint32_t kNodeAlign = 8;     // alignof(CompactBVH2Node), CompactBVH2Node.hpp:52-53

inline uint32_t padding(uint32_t baseAlign, size_t offset, uint32_t align) {
  const size_t rem = (baseAlign + offset) % align;
  return rem ? (uint32_t)(align - rem) : 0u;
}

class Writer {
 public:
  explicit Writer(uint32_t baseAlign = 16, size_t capacity = 0) : base(baseAlign) { bytes.reserve(capacity); }
  template <typename T> uint32_t write(const T& v, uint32_t align = alignof(T)) {
    const uint32_t pad = padding(base, bytes.size(), align);
    bytes.resize(bytes.size() + pad + sizeof(T), 0);
    std::memcpy(bytes.data() + bytes.size() - sizeof(T), &v, sizeof(T));
    return pad + (uint32_t)sizeof(T);
  }
  // u32 count + elements
  void writeArray(const void* src, uint32_t count, uint32_t elemSize, uint32_t elemAlign) {
    write<uint32_t>(count);
    const uint32_t pad = padding(base, bytes.size(), elemAlign);
    const size_t n = (size_t)count * elemSize;
    bytes.resize(bytes.size() + pad + n, 0);
    if (n) std::memcpy(bytes.data() + bytes.size() - n, src, n);
  }
  std::vector<uint8_t> bytes;
  const uint32_t base;
};

class Reader {
 public:
  Reader(const uint8_t* p, size_t n, uint32_t baseAlign = 16) : begin(p), size(n), base(baseAlign) {}
  template <typename T> void read(T& v, uint32_t align = alignof(T)) {
    const uint32_t pad = padding(base, off, align);
    need((size_t)pad + sizeof(T));
    off += pad;
    std::memcpy(&v, begin + off, sizeof(T));
    off += sizeof(T);
  }
  // In-place array view (deserialiseArrayRef, deserialisation.hpp:33-41)
  const void* arrayRef(uint32_t& count, uint32_t elemSize, uint32_t elemAlign) {
    read<uint32_t>(count);
    const uint32_t pad = padding(base, off, elemAlign);
    need((size_t)pad + (size_t)count * elemSize);
    off += pad;
    const void* p = begin + off;
    off += (size_t)count * elemSize;
    return p;
  }
  size_t offset() const { return off; }

 private:
  void need(size_t n) const {
    if (off + n > size) throw std::runtime_error("Deserialiser encountered end of byte stream.");   // Deserialiser.hpp:74
  }
  const uint8_t* begin;
  size_t size, off = 0;
  const uint32_t base;
};

inline void serialiseScene(Writer& w, const mi_scene_desc& d) {
  w.writeArray(d.geometry, d.num_geometry, sizeof(mi_geom_ref), 2);
  w.writeArray(d.mesh_info, d.num_meshes, sizeof(mi_mesh_info), 4);
  w.writeArray(d.mesh_tris, d.num_tris, 6, 2);
  w.writeArray(d.mesh_verts, d.num_verts, sizeof(mi_vec3), 4);
  w.writeArray(d.mesh_normals, d.num_normals, sizeof(mi_vec3), 4);
  w.writeArray(d.mat_ids, d.num_mat_ids, 4, 4);
  w.writeArray(d.materials, d.num_materials, sizeof(mi_material), 4);
  w.writeArray(d.bvh_nodes, d.num_nodes, sizeof(mi_bvh_node), kNodeAlign);
  w.write<uint32_t>(d.max_leaf_depth);
  w.write<float>(d.image_width);
  w.write<float>(d.image_height);
  w.write<float>(d.fov_radians);
  w.write<float>(d.anti_alias_scale);
  w.write<uint32_t>(d.max_path_length);
  w.write<uint32_t>(d.roulette_start_depth);
  w.write<uint32_t>(d.samples_per_pixel);
}

// Fills the array views (pointing into the blob) and the eight scalars; everything else in `d` is untouched.
inline size_t deserialiseScene(const uint8_t* bytes, size_t n, mi_scene_desc& d, uint32_t baseAlign = 16) {
  Reader r(bytes, n, baseAlign);
  d.geometry = (const mi_geom_ref*)r.arrayRef(d.num_geometry, sizeof(mi_geom_ref), 2);
  d.mesh_info = (const mi_mesh_info*)r.arrayRef(d.num_meshes, sizeof(mi_mesh_info), 4);
  d.mesh_tris = (const uint16_t*)r.arrayRef(d.num_tris, 6, 2);
  d.mesh_verts = (const mi_vec3*)r.arrayRef(d.num_verts, sizeof(mi_vec3), 4);
  d.mesh_normals = (const mi_vec3*)r.arrayRef(d.num_normals, sizeof(mi_vec3), 4);
  d.mat_ids = (const uint32_t*)r.arrayRef(d.num_mat_ids, 4, 4);
  d.materials = (const mi_material*)r.arrayRef(d.num_materials, sizeof(mi_material), 4);
  d.bvh_nodes = (const mi_bvh_node*)r.arrayRef(d.num_nodes, sizeof(mi_bvh_node), kNodeAlign);
  r.read<uint32_t>(d.max_leaf_depth);
  r.read<float>(d.image_width);
  r.read<float>(d.image_height);
  r.read<float>(d.fov_radians);
  r.read<float>(d.anti_alias_scale);
  r.read<uint32_t>(d.max_path_length);
  r.read<uint32_t>(d.roulette_start_depth);
  r.read<uint32_t>(d.samples_per_pixel);
  if (d.num_normals == 0) d.mesh_normals = nullptr;
  return r.offset();
}

}  // namespace mi::blob
