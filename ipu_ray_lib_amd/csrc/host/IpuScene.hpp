// IpuScene.hpp — header-only C++ driver object with the reference's IpuScene surface
// (include/IpuScene.hpp:22-56, caller trace.cpp:270-336), implemented over the C ABI of
// libmi_raylib.so. Same constructor arguments, same setters, same in-place overwrite of the caller's
// ray stream, rgb returned as the SUM over samples; run() plays the part of
// ipu_utils::GraphManager().run(scene): errors are logged and turned into EXIT_FAILURE
// (include/ipu_utils.hpp:590-595).
#pragma once

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../../../include/mi_raylib.h"
#include "../../../include/mi_scene_host.h"

namespace mi {

// The reference's SceneRef (include/Scene.hpp:50-74): non-owning views + render parameters.
// mi_scene_desc already is that struct field for field; spheres/discs travel separately as in the
// reference constructor.
using SceneRef = mi_scene_desc;

struct RuntimeConfig {       // subset of ipu_utils::RuntimeConfig that has a meaning on a GPU node (trace.cpp:297-309)
  uint32_t numGpus = 1;      // numIpus: devices device .. device + numGpus - 1 take part
  uint32_t numReplicas = 1;  // scene replicas the ray stream is dealt to (>= numGpus; replica i runs on device + i % numGpus)
  int device = 0;            // first device ordinal
  int transport = 0;         // mi_group_create: 0 = RCCL when more than one device takes part, 1 = RCCL always, 2 = peer copies
};

class IpuScene {
 public:
  using RayCallbackFn = std::function<void(std::size_t, const std::vector<mi_trace_result>&)>;

  IpuScene(const std::vector<mi_sphere>& spheres, const std::vector<mi_disc>& discs, SceneRef& sceneRef,
           std::vector<mi_trace_result>& results, std::size_t raysPerWorker, RayCallbackFn* fn = nullptr)
      : data(sceneRef), rayStream(results), rayFunc(fn), maxRaysPerWorker(raysPerWorker) {
    data.spheres = spheres.data(); data.num_spheres = (uint32_t)spheres.size();
    data.discs = discs.data(); data.num_discs = (uint32_t)discs.size();
  }
  ~IpuScene() { release(); if (nif) mi_host_nif_destroy(nif); }
  IpuScene(const IpuScene&) = delete;
  IpuScene& operator=(const IpuScene&) = delete;

  void setRuntimeConfig(const RuntimeConfig& c) { config = c; }

  // Reference: reads <assetPath>/nif_metadata.txt and <assetPath>/converted.hdf5 (src/IpuScene.cpp:174-187).
  // Same here, through libmi_scene_host.so (mi_host_nif_load: Keras-H5 via the HDF5 plugin, or the flat
  // nif_weights.bin dump when no .hdf5 is present). Returns false (after logging) on any failure,
  // exactly like the reference.
  bool loadNifModel(const std::string& assetPath) {
    mi_host_nif* loaded = nullptr;
    if (mi_host_nif_load(assetPath.c_str(), &loaded) != 0) {
      std::fprintf(stderr, "[error] %s\n", mi_host_nif_last_error());
      return false;
    }
    if (nif) mi_host_nif_destroy(nif);
    nif = loaded;
    mi_nif_desc d{};
    mi_host_nif_describe(nif, &d);
    std::fprintf(stderr, "[info] Loaded NIF model '%s' from '%s' (%u layers, %s weights)\n", d.name, d.source, d.num_layers,
                 d.weights_are_half ? "float16" : "float32");
    return true;
  }

  void setHdriRotation(float degrees) { hdriRotationDegrees = degrees; }
  void setAvailableMemoryProportion(float) {}                      // poplin planning knob: no GPU meaning
  void setMaxNifBatchSize(std::size_t raysPerBatch) { nifMaxRaysPerBatch = raysPerBatch; }
  double getTraceTimeSecs() const { return traceTimeSecs; }
  RayCallbackFn* getRayCallback() { return rayFunc; }

  // GraphManager().run(*this): build (scene upload) + execute (trace the ray stream in place). With more than one GPU
  // or replica the stream is dealt to the replicas in 8-row bands and the frame is assembled on the first device with
  // one RCCL group call (mi_group_render), as the reference's IpuScene spreads its ray batches over the replicas
  // (src/IpuScene.cpp:676-684, 699-732).
  int run() {
    release();
    const uint32_t gpus = config.numGpus ? config.numGpus : 1u;
    const uint32_t replicas = config.numReplicas > gpus ? config.numReplicas : gpus;
    const int mode = data.path_trace ? MI_MODE_PATH_TRACE : MI_MODE_SHADOW_TRACE;
    // batch = 1440 compute tiles x 6 workers x raysPerWorker rays, as on one IPU (src/IpuScene.cpp:360-361)
    const size_t batch = rayFunc ? (size_t)1440 * 6 * (maxRaysPerWorker ? maxRaysPerWorker : 1) : 0;
    if (replicas > 1) {
      std::vector<int32_t> devices(replicas);
      for (uint32_t i = 0; i < replicas; ++i) devices[i] = config.device + (int32_t)(i % gpus);
      if (mi_group_create(&data, devices.data(), replicas, config.transport, &group) != MI_OK) return fail("scene group creation");
      for (uint32_t i = 0; i < replicas; ++i) if (!configure(mi_group_scene(group, i))) return EXIT_FAILURE;
      mi_group_set_ray_batch(group, batch);
      if (mi_group_render(group, mode, rayStream.data(), rayStream.size(), rayFunc ? &IpuScene::trampoline : nullptr, this) != MI_OK)
        return fail("render");
      traceTimeSecs = mi_group_trace_time_secs(group);
      return EXIT_SUCCESS;
    }
    data.device = config.device;
    if (mi_scene_create(&data, &scene) != MI_OK) return fail("scene creation");
    if (!configure(scene)) return EXIT_FAILURE;
    if (rayFunc) mi_scene_set_ray_batch(scene, batch);
    if (mi_render(scene, mode, rayStream.data(), rayStream.size(), rayFunc ? &IpuScene::trampoline : nullptr, this) != MI_OK)
      return fail("render");
    traceTimeSecs = mi_trace_time_secs(scene);
    return EXIT_SUCCESS;
  }

  uint64_t rayCasts() const {
    uint64_t c[4] = {0, 0, 0, 0};
    if (group) mi_group_get_counters(group, c); else if (scene) mi_get_counters(scene, c);
    return c[0];
  }
  // RCCL send/recv pairs, peer copies and bands of the last multi-replica render (zeros for a single scene)
  void lastTransfer(uint64_t info[5]) const { for (int i = 0; i < 5; ++i) info[i] = 0; if (group) mi_group_last_transfer(group, info); }

 private:
  static void trampoline(void* user, size_t batch, const mi_trace_result* rays, size_t count) {
    auto* self = static_cast<IpuScene*>(user);
    std::vector<mi_trace_result> v(rays, rays + count);
    (*self->rayFunc)(batch, v);
  }
  void release() {
    if (scene) { mi_scene_destroy(scene); scene = nullptr; }
    if (group) { mi_group_destroy(group); group = nullptr; }
  }
  bool configure(mi_scene* s) {
    if (nif) {
      mi_nif_desc d{};
      mi_host_nif_describe(nif, &d);
      if (mi_scene_set_nif(s, d.num_layers, d.kernels, d.biases, d.rows, d.cols, d.relu, d.embedding_dimension,
                           d.max_value, d.mean, d.log_tonemap) != MI_OK) { fail("NIF upload"); return false; }
    }
    mi_scene_set_hdri_rotation(s, hdriRotationDegrees);
    mi_scene_set_max_nif_batch(s, nifMaxRaysPerBatch);
    return true;
  }
  int fail(const char* what) {
    std::fprintf(stderr, "[error] %s failed: %s\n", what, mi_last_error());
    return EXIT_FAILURE;
  }

  SceneRef data;
  std::vector<mi_trace_result>& rayStream;
  RayCallbackFn* rayFunc;
  std::size_t maxRaysPerWorker;
  RuntimeConfig config;
  mi_scene* scene = nullptr;
  mi_group* group = nullptr;
  double traceTimeSecs = 0.0;
  float hdriRotationDegrees = 0.f;
  std::size_t nifMaxRaysPerBatch = 0;
  mi_host_nif* nif = nullptr;
};

}  // namespace mi
