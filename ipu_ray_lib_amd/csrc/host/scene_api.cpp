// scene_api.cpp — C ABI of the host scene plumbing (include/mi_scene_host.h).
#include <algorithm>
#include <cstring>
#include <vector>
#include <limits>
#include <stdexcept>
#include <string>

#include "../../../include/mi_scene_host.h"
#include "scene_types.hpp"
#include "../scene_blob.hpp"
#include "../ray_shard.hpp"

using namespace mi;
using namespace mi::host;

struct mi_host_scene {
  PackedScene packed;
};

namespace {
thread_local std::string g_err;

template <class F>
int guarded(F&& f) {
  try { f(); g_err.clear(); return MI_OK; }
  catch (const std::invalid_argument& e) { g_err = e.what(); return MI_ERR_INVALID_ARG; }
  catch (const std::exception& e) { g_err = e.what(); return MI_ERR_IO; }
}
}  // namespace

extern "C" {

const char* mi_host_last_error(void) { return g_err.c_str(); }

int mi_host_scene_builtin(const char* scene_name, const char* mesh_file, mi_host_scene** out) {
  if (!scene_name || !out) { g_err = "null argument"; return MI_ERR_INVALID_ARG; }
  return guarded([&] {
    const std::string name(scene_name);
    SceneDescription desc;
    if (name == "box-simple" || name == "box") desc = makeCornellBoxScene(mesh_file ? mesh_file : "", name == "box-simple");
    else if (name == "spheres") desc = makePrimitiveScene();
    else if (name == "monkey") desc = makeMonkeyScene(mesh_file ? mesh_file : "");
    else throw std::invalid_argument("Invalid scene selection: '" + name + "'");   // src/app_utils.cpp:268-270
    auto* hs = new mi_host_scene;
    hs->packed = packScene(desc);
    *out = hs;
  });
}

int mi_host_scene_import(const char* file, int load_normals, mi_host_scene** out) {
  if (!file || !out) { g_err = "null argument"; return MI_ERR_INVALID_ARG; }
  return guarded([&] {
    const std::string path(file);
    const std::string ext = path.size() >= 4 ? path.substr(path.size() - 4) : "";
    SceneDescription desc;
    if (ext == ".dae" || ext == ".DAE") desc = importColladaScene(path, load_normals != 0);
    else if (ext == ".glb") { loadGlbMeshes(path, false); throw std::runtime_error("No camera found in scene file."); }   // scene_utils.cpp:177-180
    else throw std::runtime_error("Could not load scene file.");
    auto* hs = new mi_host_scene;
    hs->packed = packScene(desc);
    *out = hs;
  });
}

int mi_host_scene_from_arrays(const mi_scene_desc* g, mi_host_scene** out) {
  if (!g || !out) { g_err = "null argument"; return MI_ERR_INVALID_ARG; }
  return guarded([&] {
    SceneDescription d;
    for (uint32_t m = 0; m < g->num_meshes; ++m) {
      const mi_mesh_info& mi_ = g->mesh_info[m];
      TriMesh tm;
      tm.indices.assign(g->mesh_tris + 3 * (size_t)mi_.first_index, g->mesh_tris + 3 * (size_t)(mi_.first_index + mi_.num_triangles));
      for (uint32_t v = 0; v < mi_.num_vertices; ++v) {
        const mi_vec3& p = g->mesh_verts[mi_.first_vertex + v];
        tm.vertices.push_back(mk(p.x, p.y, p.z));
        if (g->num_normals) { const mi_vec3& n = g->mesh_normals[mi_.first_vertex + v]; tm.normals.push_back(mk(n.x, n.y, n.z)); }
      }
      d.meshes.push_back(std::move(tm));
    }
    d.spheres.assign(g->spheres, g->spheres + g->num_spheres);
    d.discs.assign(g->discs, g->discs + g->num_discs);
    d.materials.assign(g->materials, g->materials + g->num_materials);
    d.matIDs.assign(g->mat_ids, g->mat_ids + g->num_mat_ids);
    if (d.matIDs.size() < d.meshes.size() + d.spheres.size() + d.discs.size())
      throw std::invalid_argument("All primitives must be assigned a material.");
    d.horizontalFov = g->fov_radians;
    auto* hs = new mi_host_scene;
    hs->packed = packScene(d);
    *out = hs;
  });
}

int mi_host_scene_fill_desc(const mi_host_scene* scene, mi_scene_desc* desc) {
  if (!scene || !desc) { g_err = "null argument"; return MI_ERR_INVALID_ARG; }
  const PackedScene& p = scene->packed;
  memset(desc, 0, sizeof *desc);
  desc->geometry = p.geometry.data();       desc->num_geometry = (uint32_t)p.geometry.size();
  desc->mesh_info = p.meshInfo.data();      desc->num_meshes = (uint32_t)p.meshInfo.size();
  desc->mesh_tris = p.meshTris.data();      desc->num_tris = (uint32_t)(p.meshTris.size() / 3);
  desc->mesh_verts = p.meshVerts.data();    desc->num_verts = (uint32_t)p.meshVerts.size();
  desc->mesh_normals = p.meshNormals.data();desc->num_normals = (uint32_t)p.meshNormals.size();
  desc->mat_ids = p.matIDs.data();          desc->num_mat_ids = (uint32_t)p.matIDs.size();
  desc->materials = p.materials.data();     desc->num_materials = (uint32_t)p.materials.size();
  desc->bvh_nodes = p.bvhNodes.data();      desc->num_nodes = (uint32_t)p.bvhNodes.size();
  desc->max_leaf_depth = p.bvhMaxDepth;
  desc->spheres = p.spheres.data();         desc->num_spheres = (uint32_t)p.spheres.size();
  desc->discs = p.discs.data();             desc->num_discs = (uint32_t)p.discs.size();
  // CLI defaults, trace.cpp:343-366
  desc->image_width = 768.f; desc->image_height = 432.f;
  desc->fov_radians = p.horizontalFov;
  desc->anti_alias_scale = .25f;
  desc->max_path_length = 10; desc->roulette_start_depth = 3; desc->samples_per_pixel = 256;
  desc->rng_seed = 1442;
  desc->window_w = 768; desc->window_h = 432; desc->window_c = 0; desc->window_r = 0;
  desc->path_trace = 1;
  desc->device = 0;
  return MI_OK;
}

void mi_host_scene_destroy(mi_host_scene* scene) { delete scene; }

int mi_build_compact_bvh(const float* lower, const float* upper, const uint16_t* geom_ids,
                         const uint32_t* prim_ids, uint32_t n,
                         mi_bvh_node* nodes, uint32_t* num_nodes, uint32_t* max_leaf_depth) {
  if (!lower || !upper || !geom_ids || !prim_ids || !nodes || !num_nodes || !max_leaf_depth) { g_err = "null argument"; return MI_ERR_INVALID_ARG; }
  return guarded([&] {
    std::vector<BuildPrim> prims(n);
    for (uint32_t i = 0; i < n; ++i) {
      prims[i].box.lo = mk(lower[3 * i], lower[3 * i + 1], lower[3 * i + 2]);
      prims[i].box.hi = mk(upper[3 * i], upper[3 * i + 1], upper[3 * i + 2]);
      prims[i].geomID = geom_ids[i]; prims[i].primID = prim_ids[i];
    }
    std::vector<mi_bvh_node> out;
    uint32_t depth = 0;
    buildCompactBvh(prims, out, depth);
    memcpy(nodes, out.data(), out.size() * sizeof(mi_bvh_node));
    *num_nodes = (uint32_t)out.size();
    *max_leaf_depth = depth;
  });
}

// initPerspectiveRayStream with gen == nullptr, then zeroRgb (src/app_utils.cpp:19-53)
int mi_init_ray_stream(const mi_scene_desc* d, mi_trace_result* rays, size_t capacity) {
  if (!d || !rays) { g_err = "null argument"; return MI_ERR_INVALID_ARG; }
  if (d->window_w < 0 || d->window_h < 0 || capacity < (size_t)d->window_w * (size_t)d->window_h) { g_err = "ray buffer smaller than the render window"; return MI_ERR_INVALID_ARG; }
  float s, c;
  sincos_deg_table(d->fov_radians / 2.f, sinTable(), s, c);
  const float tanTheta = s / c;
  size_t i = 0;
  for (uint32_t r = (uint32_t)d->window_r; r < (uint32_t)(d->window_r + d->window_h); ++r)
    for (uint32_t col = (uint32_t)d->window_c; col < (uint32_t)(d->window_c + d->window_w); ++col) {
      const f3 dir = pixel_to_ray_dir((float)col, (float)r, d->image_width, d->image_height, tanTheta);
      mi_trace_result& t = rays[i++];
      memset(&t, 0, sizeof t);
      t.u = (float)r; t.v = (float)col;
      t.h.r.t_min = 0.f; t.h.r.t_max = std::numeric_limits<float>::infinity();
      t.h.r.direction = {dir.x, dir.y, dir.z};
      t.h.prim_id = MI_INVALID_PRIM;
      t.h.normal = {0.f, 0.f, 1.f};
      t.h.geom_id = MI_INVALID_GEOM;
      t.h.flags = 0;
    }
  return MI_OK;
}

void mi_scale_rgb(mi_trace_result* rays, size_t n, float scale) {
  for (size_t i = 0; i < n; ++i) { rays[i].rgb.x *= scale; rays[i].rgb.y *= scale; rays[i].rgb.z *= scale; }
}

size_t mi_scene_blob_size(const mi_scene_desc* d) {
  if (!d) return 0;
  mi::blob::Writer w(16);
  mi::blob::serialiseScene(w, *d);
  return w.bytes.size();
}

int mi_scene_serialise(const mi_scene_desc* d, uint8_t* out, size_t capacity, size_t* written) {
  return guarded([&] {
    if (!d || !out) throw std::runtime_error("null argument");
    mi::blob::Writer w(16, 600 * 1024);                            // the reference reserves 600 KiB (src/IpuScene.cpp:31)
    mi::blob::serialiseScene(w, *d);
    if (w.bytes.size() > capacity) throw std::runtime_error("serialised scene needs " + std::to_string(w.bytes.size()) + " bytes");
    std::memcpy(out, w.bytes.data(), w.bytes.size());
    if (written) *written = w.bytes.size();
  });
}

int mi_scene_deserialise(const uint8_t* blob, size_t size, mi_scene_desc* d, size_t* consumed) {
  return guarded([&] {
    if (!blob || !d) throw std::runtime_error("null argument");
    if ((uintptr_t)blob % 16) throw std::runtime_error("serialised scene must be 16-byte aligned to be aliased in place");
    const size_t used = mi::blob::deserialiseScene(blob, size, *d, 16);
    if (consumed) *consumed = used;
  });
}

size_t mi_shard_band_rays(size_t n, uint32_t window_w) { return mi::shard::band_rays(n, window_w); }

size_t mi_shard_count(size_t n, size_t band, uint32_t replicas, uint32_t r) { return mi::shard::replica_count(n, band, replicas, r); }

int mi_shard_stream_index(size_t n, size_t band, uint32_t replicas, uint32_t r, uint64_t* out, size_t capacity) {
  if (!out || band == 0 || replicas == 0 || r >= replicas) { g_err = "mi_shard_stream_index: bad argument"; return MI_ERR_INVALID_ARG; }
  if (capacity < mi::shard::replica_count(n, band, replicas, r)) { g_err = "mi_shard_stream_index: output too small"; return MI_ERR_INVALID_ARG; }
  size_t k = 0;
  for (size_t b = r; b * band < n; b += replicas)
    for (size_t i = b * band; i < std::min(n, (b + 1) * band); ++i) out[k++] = i;
  return MI_OK;
}

int mi_shard_frame_index(size_t n, size_t band, uint32_t replicas, uint64_t* out) {
  if (!out || band == 0 || replicas == 0) { g_err = "mi_shard_frame_index: bad argument"; return MI_ERR_INVALID_ARG; }
  std::vector<size_t> offset(replicas + 1, 0);
  for (uint32_t r = 0; r < replicas; ++r) offset[r + 1] = offset[r] + mi::shard::replica_count(n, band, replicas, r);
  for (size_t i = 0; i < n; ++i) {
    uint32_t r; size_t pos;
    mi::shard::locate(band, replicas, i, r, pos);
    out[i] = offset[r] + pos;
  }
  return MI_OK;
}

uint32_t mi_blob_padding(uint32_t base_align, size_t offset, uint32_t align) {
  return align ? mi::blob::padding(base_align, offset, align) : 0u;
}

}  // extern "C"
