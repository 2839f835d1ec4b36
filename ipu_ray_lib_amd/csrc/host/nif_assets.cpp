// nif_assets.cpp — NIF asset loading: nif_metadata.txt (JSON) + converted.hdf5 (Keras "Functional"
// model saved as HDF5). Restates what the reference does in src/neural_networks/NifMetaData.cpp:11-71,
// src/keras/Hdf5Model.cpp:8-96 and NifModel::Data::setupModel (src/neural_networks/NifModel.cpp:51-86),
// with the results handed over as plain float arrays for mi_scene_set_nif.
//
// HDF5 itself is reached through the plugin libmi_nif_h5.so (nif_h5.c), dlopen'ed from the directory
// this library lives in; when the plugin (or libhdf5) is missing, loading a .hdf5 model fails with a
// message saying so — there is no silent fallback. <assetPath>/nif_weights.bin (a flat dump, format in
// include/mi_scene_host.h) is read instead only when <assetPath>/converted.hdf5 does not exist.
#include <dlfcn.h>

#include <cstdint>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/mi_scene_host.h"
#include "json_min.hpp"

namespace {

struct H5Plugin {
  void* so = nullptr;
  int (*open)(const char*, void**, char*, size_t) = nullptr;
  void (*close)(void*) = nullptr;
  void (*free_)(void*) = nullptr;
  int (*readAttr)(void*, const char*, char**, char*, size_t) = nullptr;
  int (*readData)(void*, const char*, float**, uint64_t*, int*, int*, char*, size_t) = nullptr;
};

std::string ownDirectory() {
  Dl_info info{};
  if (dladdr((void*)&ownDirectory, &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    const auto slash = p.find_last_of('/');
    return slash == std::string::npos ? "." : p.substr(0, slash);
  }
  return ".";
}

H5Plugin& h5Plugin() {
  static H5Plugin p;
  if (p.so) return p;
  const char* overridePath = std::getenv("MI_NIF_H5_PLUGIN");
  const std::string path = overridePath ? overridePath : ownDirectory() + "/libmi_nif_h5.so";
  void* so = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!so) {
    const char* why = dlerror();
    throw std::runtime_error("HDF5 support is unavailable: cannot load '" + path + "' (" + (why ? why : "?") + ")");
  }
  auto sym = [&](const char* n) {
    void* s = dlsym(so, n);
    if (!s) { dlclose(so); throw std::runtime_error(std::string("HDF5 plugin lacks symbol ") + n); }
    return s;
  };
  p.open = (decltype(p.open))sym("mi_h5_open");
  p.close = (decltype(p.close))sym("mi_h5_close");
  p.free_ = (decltype(p.free_))sym("mi_h5_free");
  p.readAttr = (decltype(p.readAttr))sym("mi_h5_read_string_attr");
  p.readData = (decltype(p.readData))sym("mi_h5_read_float_dataset");
  p.so = so;
  return p;
}

struct H5File {
  H5Plugin& p;
  void* h = nullptr;
  explicit H5File(const std::string& file) : p(h5Plugin()) {
    char err[512] = {0};
    if (p.open(file.c_str(), &h, err, sizeof err) != 0) throw std::runtime_error(err);
  }
  ~H5File() { if (h) p.close(h); }
  std::string attr(const char* name) {
    char err[512] = {0};
    char* s = nullptr;
    if (p.readAttr(h, name, &s, err, sizeof err) != 0) throw std::runtime_error(err);
    std::string out = s ? s : "";
    p.free_(s);
    return out;
  }
  std::vector<float> dataset(const std::string& path, std::vector<uint64_t>& shape, bool& isHalf) {
    char err[512] = {0};
    float* d = nullptr;
    uint64_t dims[4] = {0, 0, 0, 0};
    int nd = 0, eb = 0;
    if (p.readData(h, path.c_str(), &d, dims, &nd, &eb, err, sizeof err) != 0) throw std::runtime_error(err);
    shape.assign(dims, dims + nd);
    size_t n = 1;
    for (auto v : shape) n *= (size_t)v;
    std::vector<float> out(d, d + n);
    p.free_(d);
    isHalf = eb == 2;
    return out;
  }
};

bool fileExists(const std::string& p) { std::ifstream f(p); return (bool)f; }

}  // namespace

struct mi_host_nif {
  std::string name;
  uint32_t embedding = 0, hiddenSize = 0;
  float eps = 0.f, maxValue = 1.f, mean[3] = {0, 0, 0};
  bool logToneMap = true, anyHalf = false;
  std::vector<uint32_t> imageShape;
  std::vector<std::string> layerNames;
  std::vector<std::vector<float>> kernels, biases;
  std::vector<const float*> kernelPtrs, biasPtrs;
  std::vector<uint32_t> rows, cols;
  std::vector<uint8_t> relu;
  std::string source;
};

namespace {

// NifMetaData::NifMetaData (src/neural_networks/NifMetaData.cpp:11-71)
void readMetaData(const std::string& file, mi_host_nif& m) {
  std::ifstream in(file);
  if (!in) throw std::runtime_error("cannot open '" + file + "'");
  std::stringstream ss; ss << in.rdbuf();
  mi::json::ValuePtr doc;
  try { doc = mi::json::parse(ss.str()); }
  catch (const std::exception& e) { throw std::runtime_error(std::string("Error reading property: ") + e.what() + " from file: '" + file + "'"); }
  try {
    m.embedding = (uint32_t)doc->at("embedding_dimension").number();
    m.name = doc->at("name").string();
    if (doc->has("original_image_shape"))
      for (size_t i = 0; i < doc->at("original_image_shape").size(); ++i)
        m.imageShape.push_back((uint32_t)doc->at("original_image_shape").at(i).number());
    const auto& enc = doc->at("encode_params");
    m.eps = (float)enc.at("eps").number();
    m.logToneMap = enc.at("log_tone_map").b;
    m.maxValue = (float)enc.at("max").number();
    const auto& mean = enc.at("mean");
    if (mean.size() < 3) throw std::runtime_error("encode_params.mean needs 3 entries");
    for (int i = 0; i < 3; ++i) m.mean[i] = (float)mean.at(i).number();
    if (m.logToneMap) for (float& v : m.mean) v -= m.eps;           // inverse eps folded into the mean (:48-53)
    if (doc->has("train_command")) {                                 // hidden size = argument after --layer-size (:56-64)
      const auto& cmd = doc->at("train_command");
      bool next = false;
      for (size_t i = 0; i < cmd.size(); ++i) {
        const auto& v = cmd.at(i);
        const std::string s = v.kind == mi::json::Value::String ? v.str : std::string();
        if (next) { m.hiddenSize = (uint32_t)std::atoi(s.c_str()); next = false; }
        if (s == "--layer-size") next = true;
      }
    }
  } catch (const std::exception& e) {
    throw std::runtime_error(std::string("Error reading property: ") + e.what() + " from file: '" + file + "'");
  }
}

// Hdf5Model::Hdf5Model + parseJsonModel (src/keras/Hdf5Model.cpp:8-86)
void readKerasH5(const std::string& file, mi_host_nif& m) {
  H5File h5(file);
  const std::string config = h5.attr("model_config");
  auto doc = mi::json::parse(config);
  if (doc->at("class_name").string() != "Functional") throw std::runtime_error("Expected a Keras 'Functional' Model");
  const auto& layers = doc->at("config").at("layers");
  for (size_t i = 0; i < layers.size(); ++i) {
    const auto& l = layers.at(i);
    const std::string cls = l.at("class_name").string();
    if (cls == "InputLayer" || cls == "Concatenate") continue;      // hard-wired in the NIF evaluator (:40-42)
    if (cls != "Dense") throw std::runtime_error("Layer class: '" + cls + "' not supported by Hdf5Model loader.");
    const auto& cfg = l.at("config");
    const std::string name = cfg.at("name").string();
    const std::string act = cfg.at("activation").string();
    const uint32_t units = (uint32_t)cfg.at("units").number();
    const bool useBias = cfg.at("use_bias").b;
    if (act != "relu" && act != "linear") throw std::runtime_error("Dense layer '" + name + "': activation '" + act + "' is not supported (relu|linear)");

    std::vector<uint64_t> shape;
    bool half = false;
    auto kernel = h5.dataset("/model_weights/" + name + "/" + name + "/kernel:0", shape, half);
    if (shape.size() != 2 || shape[1] != units) throw std::runtime_error("Dense layer '" + name + "': kernel shape does not match units");
    m.anyHalf = m.anyHalf || half;
    m.rows.push_back((uint32_t)shape[0]);
    m.cols.push_back((uint32_t)shape[1]);
    m.kernels.push_back(std::move(kernel));
    if (useBias) {
      std::vector<uint64_t> bshape;
      auto bias = h5.dataset("/model_weights/" + name + "/" + name + "/bias:0", bshape, half);
      if (bshape.size() != 1 || bshape[0] != units) throw std::runtime_error("Dense layer '" + name + "': bias shape does not match units");
      m.biases.push_back(std::move(bias));
    } else {
      m.biases.emplace_back();
    }
    m.relu.push_back(act == "relu" ? 1 : 0);                        // "linear" ≙ none (NifModel.cpp:75-77)
    m.layerNames.push_back(name);
  }
  if (m.kernels.empty()) throw std::runtime_error("model has no Dense layers");
  m.source = file;
}

// Flat dump: u32 numLayers, then per layer u32 rows, u32 cols, u8 relu, u8 hasBias,
// f32 kernel[rows*cols] (row-major, Keras kernel:0 order), f32 bias[cols] when hasBias.
void readFlatWeights(const std::string& file, mi_host_nif& m) {
  std::ifstream w(file, std::ios::binary);
  if (!w) throw std::runtime_error("cannot open '" + file + "'");
  uint32_t n = 0;
  w.read((char*)&n, 4);
  if (!w || n == 0 || n > 64) throw std::runtime_error("bad layer count in '" + file + "'");
  for (uint32_t l = 0; l < n; ++l) {
    uint32_t r = 0, c = 0; uint8_t relu = 0, hasBias = 0;
    w.read((char*)&r, 4); w.read((char*)&c, 4); w.read((char*)&relu, 1); w.read((char*)&hasBias, 1);
    if (!w || r == 0 || c == 0 || (uint64_t)r * c > (1u << 28)) throw std::runtime_error("bad layer header in '" + file + "'");
    std::vector<float> k((size_t)r * c), b(hasBias ? c : 0);
    w.read((char*)k.data(), k.size() * 4);
    if (hasBias) w.read((char*)b.data(), b.size() * 4);
    if (!w) throw std::runtime_error("truncated weights file '" + file + "'");
    m.rows.push_back(r); m.cols.push_back(c); m.relu.push_back(relu);
    m.kernels.push_back(std::move(k)); m.biases.push_back(std::move(b));
    m.layerNames.push_back("layer_" + std::to_string(l));
  }
  m.source = file;
}

thread_local std::string g_nifErr;

}  // namespace

extern "C" {

const char* mi_host_nif_last_error(void) { return g_nifErr.c_str(); }

int mi_host_nif_load(const char* asset_path, mi_host_nif** out) {
  if (!asset_path || !out) { g_nifErr = "null argument"; return 1; }
  try {
    auto m = std::make_unique<mi_host_nif>();
    const std::string dir = asset_path;
    readMetaData(dir + "/nif_metadata.txt", *m);                    // IpuScene::loadNifModel (src/IpuScene.cpp:176-178)
    const std::string h5 = dir + "/converted.hdf5";
    if (fileExists(h5)) readKerasH5(h5, *m);
    else if (fileExists(dir + "/nif_weights.bin")) readFlatWeights(dir + "/nif_weights.bin", *m);
    else throw std::runtime_error("neither converted.hdf5 nor nif_weights.bin found in '" + dir + "'");
    for (size_t i = 0; i < m->kernels.size(); ++i) {
      m->kernelPtrs.push_back(m->kernels[i].data());
      m->biasPtrs.push_back(m->biases[i].empty() ? nullptr : m->biases[i].data());
    }
    *out = m.release();
    return 0;
  } catch (const std::exception& e) {
    g_nifErr = std::string("Could not load NIF model from '") + asset_path + "'. Exception: " + e.what();
    return 2;
  }
}

int mi_host_nif_describe(const mi_host_nif* m, mi_nif_desc* d) {
  if (!m || !d) { g_nifErr = "null argument"; return 1; }
  d->num_layers = (uint32_t)m->kernels.size();
  d->kernels = m->kernelPtrs.data();
  d->biases = m->biasPtrs.data();
  d->rows = m->rows.data();
  d->cols = m->cols.data();
  d->relu = m->relu.data();
  d->embedding_dimension = m->embedding;
  d->hidden_size = m->hiddenSize;
  d->max_value = m->maxValue;
  for (int i = 0; i < 3; ++i) d->mean[i] = m->mean[i];
  d->log_tonemap = m->logToneMap ? 1 : 0;
  d->weights_are_half = m->anyHalf ? 1 : 0;
  d->name = m->name.c_str();
  d->source = m->source.c_str();
  return 0;
}

void mi_host_nif_destroy(mi_host_nif* m) { delete m; }

}  // extern "C"
