// nif_regs_pack.hpp — host side of K3r (nif_regs_kernel.hpp): shape check and packing of a NIF model's weights into the
// stream of 1-KiB A fragments the register-resident MLP kernel consumes. Included by nif_kernels.hpp ahead of NifDevice,
// which owns one NifRegsDevice next to the packed weights of nif_mlp_kernel.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace mi {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
constexpr uint32_t kNifMaxLayers = 16;

constexpr uint32_t kRegRows = 256;                             // rays per workgroup pass (8 waves x 2 ray tiles of 16)
constexpr uint32_t kRegSlots = 3;                              // weight ring: slots of one chunk each
constexpr uint32_t kRegFeatBytes = 32u * 1024u;                // the pass's Fourier features as B fragments: [wave][chunk][ray tile][lane x 16 B]
constexpr uint32_t kRegPassCounters = 64u;                     // K3a / K3b: pass counters, one per launch in flight (nif_asm_kernel.hpp)
constexpr uint32_t kRegMaxLdsBytes = 160u * 1024u;             // one workgroup per compute unit may take all of its LDS

enum : uint32_t { RL_FIRST = 0, RL_PLAIN = 1, RL_CONCAT = 2, RL_LAST = 3, RL_LAST_CONCAT = 4 };

// what the kernel reads once per pass or per layer sits in device memory, not in its scalar registers
struct NifRegsCold {
  uint32_t numLayers, embedDim;
  float maxValue, mean[3];
  int32_t logTonemap;
  uint32_t chunksPerPass, biasFloats;
};
struct NifRegsLayer { uint32_t kind, relu, biasBase, pad; };

struct NifRegsDevice {
  NifRegsCold cold{};
  uint32_t ht = 0;                  // hidden width / 32
  h8* d_stream = nullptr;           // the weight fragments in consumption order (1 KiB each), cut into chunks of 4 ht fragments (the final layer: one short chunk)
  uint2* d_chunks = nullptr;        // per chunk of a pass: {first fragment, fragments (a multiple of 8: every wave fetches fragments / 8 of them)}
  NifRegsLayer* d_layers = nullptr;
  NifRegsCold* d_cold = nullptr;
  float* d_bias = nullptr;          // all layers' biases, 16 per output tile (zero padded)
  uint32_t* d_passCtr = nullptr;    // kRegPassCounters zeroed words (K3a / K3b draw their passes from one per launch)
  mutable uint32_t ctrNext = 0;
  std::vector<NifRegsLayer> layersHost;      // the layer table as uploaded (K3a's launch checks the layer sequence against the one its body was generated for)
  bool ok = false;

  void release() {
    if (d_stream) (void)hipFree(d_stream);
    if (d_chunks) (void)hipFree(d_chunks);
    if (d_layers) (void)hipFree(d_layers);
    if (d_cold) (void)hipFree(d_cold);
    if (d_bias) (void)hipFree(d_bias);
    if (d_passCtr) (void)hipFree(d_passCtr);
    d_passCtr = nullptr;
    d_stream = nullptr; d_chunks = nullptr; d_layers = nullptr; d_cold = nullptr; d_bias = nullptr; ok = false;
    layersHost.clear();
  }

  static bool widthSupported(uint32_t hidden) { return hidden == 64 || hidden == 128 || hidden == 256 || hidden == 320; }
  size_t ldsBytes() const { return (size_t)kRegSlots * 4u * ht * 1024u + (size_t)cold.biasFloats * sizeof(float) + kRegFeatBytes; }

  // The k order of a layer's input as the B operand holds it. Activations of k-step ks (32 features = two output tiles of
  // the previous layer): lane group g, element e -> feature 32 ks + (e < 4 ? 4 g + e : 16 + 4 g + e - 4), i.e. the four
  // accumulator values of the lane in tile 2 ks followed by the four in tile 2 ks + 1. Feature k-steps (chunk c of the 64
  // padded Fourier features): lane group g, element e -> feature 32 c + 8 g + e.
  static uint32_t actK(uint32_t ks, uint32_t g, uint32_t e) { return 32u * ks + (e < 4u ? 4u * g + e : 16u + 4u * g + (e - 4u)); }

  // Returns false (and leaves the object unloaded) when the network is not of a supported shape: the caller then runs
  // nif_mlp_kernel. Throws only on allocation failure.
  bool load(uint32_t numLayers, const float* const* kernels, const float* const* biases, const uint32_t* rows, const uint32_t* cols,
            const uint8_t* relu, uint32_t embedDim, float maxValue, const float mean[3], int32_t logTonemap) {
    release();
    const uint32_t F = 4u * embedDim;
    if (numLayers < 2 || numLayers > kNifMaxLayers || embedDim == 0 || F > 64u) return false;
    const uint32_t H = cols[0];
    if (!widthSupported(H) || rows[0] != F || cols[numLayers - 1] != 3u) return false;
    for (uint32_t l = 1; l + 1 < numLayers; ++l) if (cols[l] != H || (rows[l] != H && rows[l] != H + F)) return false;
    if (rows[numLayers - 1] != H && rows[numLayers - 1] != H + F) return false;
    const uint32_t HT = H / 32u, CH = 4u * HT;
    NifRegsCold P{};
    P.numLayers = numLayers; P.embedDim = embedDim;
    P.maxValue = maxValue; P.mean[0] = mean[0]; P.mean[1] = mean[1]; P.mean[2] = mean[2]; P.logTonemap = logTonemap;
    std::vector<_Float16> stream;
    std::vector<uint2> chunks;
    std::vector<NifRegsLayer> layers(numLayers);
    std::vector<float> bias;
    // one fragment: tile nt (16 output features), k-step ks of the layer's input; lane l = 16 g + r holds, for output feature
    // 16 nt + r, the 8 weights of its k-values
    auto fragment = [&](uint32_t l, uint32_t nt, uint32_t ks, bool featuresOnly, bool concat) {
      const uint32_t K = rows[l], N = cols[l];
      const uint32_t actSteps = featuresOnly ? 0u : HT;
      for (uint32_t lane = 0; lane < 64; ++lane)
        for (uint32_t e = 0; e < 8; ++e) {
          const uint32_t g = lane >> 4, n = nt * 16u + (lane & 15u);
          uint32_t k;
          bool real;
          if (ks < actSteps) { k = actK(ks, g, e); real = true; }
          else { const uint32_t f = 32u * (ks - actSteps) + 8u * g + e; real = f < F && (featuresOnly || concat); k = (featuresOnly ? 0u : H) + f; }
          const float w = (real && k < K && n < N) ? kernels[l][(size_t)k * N + n] : 0.f;
          stream.push_back((_Float16)w);
        }
    };
    // a layer's fragments [first, now) as chunks of CH (every chunk a multiple of 8 fragments: zero fragments fill up)
    auto closeLayer = [&](uint32_t first) {
      uint32_t n = (uint32_t)(stream.size() / 512u) - first;
      while (n % 8u) { stream.insert(stream.end(), 512u, (_Float16)0.f); ++n; }
      for (uint32_t at = 0; at < n; at += CH) chunks.push_back(make_uint2(first + at, std::min(CH, n - at)));
    };
    for (uint32_t l = 0; l < numLayers; ++l) {
      const bool first = l == 0, last = l + 1 == numLayers;
      const bool concat = !first && rows[l] == H + F;
      layers[l].kind = first ? RL_FIRST : last ? (concat ? RL_LAST_CONCAT : RL_LAST) : (concat ? RL_CONCAT : RL_PLAIN);
      layers[l].relu = relu[l] ? 1u : 0u;
      layers[l].biasBase = (uint32_t)bias.size();
      layers[l].pad = 0;
      const uint32_t tiles = last ? 1u : 2u * HT;
      for (uint32_t n = 0; n < tiles * 16u; ++n) bias.push_back((biases && biases[l] && n < cols[l]) ? biases[l][n] : 0.f);
      const uint32_t KS = first ? 2u : (concat ? HT + 2u : HT);
      const uint32_t c0 = (uint32_t)(stream.size() / 512u);
      if (last) {
        for (uint32_t ks = 0; ks < KS; ++ks) fragment(l, 0, ks, false, concat);         // the final layer: its one tile, k-step by k-step
      } else {
        // a hidden layer: output-tile PAIR j, k-step ks, tile 2j then 2j + 1
        for (uint32_t j = 0; j < HT; ++j)
          for (uint32_t ks = 0; ks < KS; ++ks) { fragment(l, 2 * j, ks, first, concat); fragment(l, 2 * j + 1, ks, first, concat); }
      }
      closeLayer(c0);
    }
    P.chunksPerPass = (uint32_t)chunks.size();
    P.biasFloats = (uint32_t)bias.size();
    cold = P; ht = HT;
    if (ldsBytes() > kRegMaxLdsBytes) return false;
    auto up = [&](auto*& d, const auto& v) {
      if (hipMalloc(&d, v.size() * sizeof(v[0])) != hipSuccess) throw std::runtime_error("NIF: hipMalloc failed");
      (void)hipMemcpy(d, v.data(), v.size() * sizeof(v[0]), hipMemcpyHostToDevice);
    };
    if (hipMalloc(&d_stream, stream.size() * sizeof(_Float16)) != hipSuccess) throw std::runtime_error("NIF: hipMalloc failed");
    (void)hipMemcpy(d_stream, stream.data(), stream.size() * sizeof(_Float16), hipMemcpyHostToDevice);
    up(d_chunks, chunks); up(d_layers, layers); up(d_bias, bias);
    layersHost = layers;
    if (hipMalloc(&d_passCtr, kRegPassCounters * sizeof(uint32_t)) != hipSuccess) throw std::runtime_error("NIF: hipMalloc failed");
    (void)hipMemset(d_passCtr, 0, kRegPassCounters * sizeof(uint32_t));
    if (hipMalloc(&d_cold, sizeof(NifRegsCold)) != hipSuccess) throw std::runtime_error("NIF: hipMalloc failed");
    (void)hipMemcpy(d_cold, &P, sizeof P, hipMemcpyHostToDevice);
    ok = true;
    return true;
  }
};

}  // namespace mi
