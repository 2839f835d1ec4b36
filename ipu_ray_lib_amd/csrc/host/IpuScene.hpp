// IpuScene.hpp — header-only C++ driver object with the reference's IpuScene surface
// (include/IpuScene.hpp:22-56, caller trace.cpp:270-336), implemented over the C ABI of
// libmi_raylib.so. Same constructor arguments, same setters, same in-place overwrite of the caller's
// ray stream, rgb returned as the SUM over samples; run() plays the part of
// ipu_utils::GraphManager().run(scene): errors are logged and turned into EXIT_FAILURE
// (include/ipu_utils.hpp:590-595).
#pragma once

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <sstream>
#include <string>
#include <vector>

#include "../../../include/mi_raylib.h"
#include "json_min.hpp"

namespace mi {

// The reference's SceneRef (include/Scene.hpp:50-74): non-owning views + render parameters.
// mi_scene_desc already is that struct field for field; spheres/discs travel separately as in the
// reference constructor.
using SceneRef = mi_scene_desc;

struct RuntimeConfig {       // subset of ipu_utils::RuntimeConfig that has a meaning on a GPU node
  uint32_t numGpus = 1;      // numIpus
  uint32_t numReplicas = 1;
  int device = 0;            // first device ordinal
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
  ~IpuScene() { if (scene) mi_scene_destroy(scene); }
  IpuScene(const IpuScene&) = delete;
  IpuScene& operator=(const IpuScene&) = delete;

  void setRuntimeConfig(const RuntimeConfig& c) { config = c; }

  // Reference: reads <assetPath>/nif_metadata.txt and <assetPath>/converted.hdf5. The Keras-H5 reader
  // is not reproduced; weights are read from <assetPath>/nif_weights.bin instead:
  // u32 numLayers, then per layer u32 rows, u32 cols, u8 relu, u8 hasBias, f32 kernel[rows*cols]
  // (row-major, Keras kernel:0 order), f32 bias[cols]. Returns false (after logging) on any failure,
  // exactly like the reference (src/IpuScene.cpp:174-187).
  bool loadNifModel(const std::string& assetPath) {
    try {
      std::ifstream meta(assetPath + "/nif_metadata.txt");
      if (!meta) throw std::runtime_error("cannot open nif_metadata.txt");
      std::stringstream ss; ss << meta.rdbuf();
      auto doc = json::parse(ss.str());
      nifEmbedding = (uint32_t)doc->at("embedding_dimension").number();
      const auto& enc = doc->at("encode_params");
      const float eps = (float)enc.at("eps").number();
      nifLogTonemap = enc.at("log_tone_map").b;
      nifMax = (float)enc.at("max").number();
      for (int i = 0; i < 3; ++i) nifMean[i] = (float)enc.at("mean").at(i).number();
      if (nifLogTonemap) for (float& m : nifMean) m -= eps;            // NifMetaData.cpp:48-53
      std::ifstream w(assetPath + "/nif_weights.bin", std::ios::binary);
      if (!w) throw std::runtime_error("cannot open nif_weights.bin (the reference's converted.hdf5 is not readable here)");
      uint32_t n = 0; w.read((char*)&n, 4);
      if (!w || n == 0 || n > 16) throw std::runtime_error("bad layer count");
      nifKernels.assign(n, {}); nifBiases.assign(n, {}); nifRows.assign(n, 0); nifCols.assign(n, 0); nifRelu.assign(n, 0);
      for (uint32_t l = 0; l < n; ++l) {
        uint8_t relu = 0, hasBias = 0;
        w.read((char*)&nifRows[l], 4); w.read((char*)&nifCols[l], 4); w.read((char*)&relu, 1); w.read((char*)&hasBias, 1);
        nifRelu[l] = relu;
        nifKernels[l].resize((size_t)nifRows[l] * nifCols[l]);
        w.read((char*)nifKernels[l].data(), nifKernels[l].size() * 4);
        if (hasBias) { nifBiases[l].resize(nifCols[l]); w.read((char*)nifBiases[l].data(), nifCols[l] * 4); }
        if (!w) throw std::runtime_error("truncated weights file");
      }
      nifLoaded = true;
      std::fprintf(stderr, "[info] Loaded NIF model from '%s'\n", assetPath.c_str());
      return true;
    } catch (const std::exception& e) {
      std::fprintf(stderr, "[error] Could not load NIF model from '%s'. Exception: %s\n", assetPath.c_str(), e.what());
    }
    return false;
  }

  void setHdriRotation(float degrees) { hdriRotationDegrees = degrees; }
  void setAvailableMemoryProportion(float) {}                      // poplin planning knob: no GPU meaning
  void setMaxNifBatchSize(std::size_t raysPerBatch) { nifMaxRaysPerBatch = raysPerBatch; }
  double getTraceTimeSecs() const { return traceTimeSecs; }
  RayCallbackFn* getRayCallback() { return rayFunc; }

  // GraphManager().run(*this): build (scene upload) + execute (trace the ray stream in place).
  int run() {
    data.device = config.device;
    if (mi_scene_create(&data, &scene) != MI_OK) return fail("scene creation");
    if (nifLoaded) {
      std::vector<const float*> kp, bp;
      for (size_t l = 0; l < nifKernels.size(); ++l) { kp.push_back(nifKernels[l].data()); bp.push_back(nifBiases[l].empty() ? nullptr : nifBiases[l].data()); }
      if (mi_scene_set_nif(scene, (uint32_t)kp.size(), kp.data(), bp.data(), nifRows.data(), nifCols.data(), nifRelu.data(),
                           nifEmbedding, nifMax, nifMean, nifLogTonemap ? 1 : 0) != MI_OK) return fail("NIF upload");
    }
    mi_scene_set_hdri_rotation(scene, hdriRotationDegrees);
    mi_scene_set_max_nif_batch(scene, nifMaxRaysPerBatch);
    // batch = 1440 compute tiles x 6 workers x raysPerWorker rays, as on one IPU (src/IpuScene.cpp:360-361)
    if (rayFunc) mi_scene_set_ray_batch(scene, (size_t)1440 * 6 * (maxRaysPerWorker ? maxRaysPerWorker : 1));
    const int mode = data.path_trace ? MI_MODE_PATH_TRACE : MI_MODE_SHADOW_TRACE;
    if (mi_render(scene, mode, rayStream.data(), rayStream.size(), rayFunc ? &IpuScene::trampoline : nullptr, this) != MI_OK)
      return fail("render");
    traceTimeSecs = mi_trace_time_secs(scene);
    return EXIT_SUCCESS;
  }

  uint64_t rayCasts() const { uint64_t c[4] = {0, 0, 0, 0}; if (scene) mi_get_counters(scene, c); return c[0]; }

 private:
  static void trampoline(void* user, size_t batch, const mi_trace_result* rays, size_t count) {
    auto* self = static_cast<IpuScene*>(user);
    std::vector<mi_trace_result> v(rays, rays + count);
    (*self->rayFunc)(batch, v);
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
  double traceTimeSecs = 0.0;
  float hdriRotationDegrees = 0.f;
  std::size_t nifMaxRaysPerBatch = 0;
  bool nifLoaded = false, nifLogTonemap = true;
  uint32_t nifEmbedding = 0;
  float nifMax = 1.f, nifMean[3] = {0, 0, 0};
  std::vector<std::vector<float>> nifKernels, nifBiases;
  std::vector<uint32_t> nifRows, nifCols;
  std::vector<uint8_t> nifRelu;
};

}  // namespace mi
