// nif_kernels.hpp — K3: the NIF neural environment light on gfx950 matrix cores.
//
//   escaped_uv_kernel   == PreProcessEscapedRays  (codelets/TraceCodelets.cpp:321-358) + wave-ballot
//                          compaction of the escaped rays, so the MLP only runs on rays that need it
//   nif_mlp_kernel      == NifModel::buildInference's execModel (src/neural_networks/NifModel.cpp:
//                          186-246 encode/decode, :300-327 dense stack) with the env add of
//                          PostProcessEscapedRays (codelets :361-382) fused into its epilogue
//
// Numerics follow the reference's fp16 model (--fp16 training, NifModel.cpp:212-216,
// src/IpuScene.cpp:256-262): Fourier features are evaluated on binary16-rounded phases and rounded
// to binary16; weights and inter-layer activations are binary16; products accumulate in binary32
// on v_mfma_f32_16x16x32_f16 (the IPU accumulates in half: ours is the more accurate of the two).
//
// Kernel shape: a 256-thread workgroup owns 96 rays (two workgroups per CU); wave w evaluates
// output-feature tiles {w, w+4, ...} for all 96 rays, so every packed weight fragment is fetched once per
// workgroup and feeds 6 MFMAs. Activations live in LDS as [rays][stride] binary16; the dense layers are
// evaluated transposed, Y^T = W^T · X^T, so that the MFMA result fragment of a lane is 4 consecutive output
// features of ONE ray and goes back to LDS as one 8-byte store. W^T is pre-packed on the host in exact
// A-fragment order, so every weight load is a fully coalesced 1 KiB wave read served from L2. Both operand
// streams are software-pipelined with inline-asm loads whose completion is counted by hand (hipcc sinks its own
// loads to their first use at this register pressure): weights three k-steps deep in registers, activation
// fragments three deep. The 8-wave shapes (64 / 96 rays per row group) are kept selectable (MI_RAYLIB_NIF_SHAPE)
// for widths whose LDS image does not fit and for the measurements in DESIGN.md.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <stdexcept>
#include <string>
#include <vector>

#include "ray_math.h"
#include "../../include/mi_raylib.h"

#ifndef MI_NIF_STAMPS
#define MI_NIF_STAMPS 0
#endif

namespace mi {

#if MI_NIF_STAMPS
// diagnostic build only (tools/nif_stamps.py): where one wave's cycles go. [0] k-loops, [1] wait at the barrier after a
// k-loop, [2] epilogue body, [3] wait at the barrier after it, [4] staging, [5] kernel total, [6] waves counted
__device__ unsigned long long nif_stamps[8];
__device__ __forceinline__ unsigned long long nif_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define MI_STAMP(var) const unsigned long long var = nif_now()
#define MI_STAMP_ADD(slot, a, b) stampSum[slot] += (b) - (a)
#else
#define MI_STAMP(var)
#define MI_STAMP_ADD(slot, a, b)
#endif

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));

// Workgroup shapes of the same kernel (template parameters MT = 16-ray tiles per wave, RG = row groups):
//   w6 (default) : 4 waves x 96 rays, two workgroups per CU; every packed weight fragment is fetched once per
//                  workgroup and feeds 6 MFMAs. Needs 96 x stride x 2 B of LDS per workgroup.
//   t6, t4       : 8 waves = 4 output-feature groups x 2 row groups of 96 / 64 rays (t4: any width up to 384).
constexpr uint32_t kNifMaxLdsBytes = 160 * 1024 - 2048;   // dynamic LDS: what is left beside the static coordinate staging (2 x 192 floats)
constexpr uint32_t kNifMaxLayers = 16;
constexpr uint32_t kNifMaxTilesPerWave = 6;   // output-feature tiles (of 16) per wave: supports widths up to 384

struct NifLayerDesc {
  uint32_t kSteps;        // ceil(K/32)
  uint32_t nTiles;        // ceil(N/16)
  uint32_t n;             // real output width
  uint32_t inBase;        // first LDS column of the layer's input
  uint32_t wOffset;       // offset (in h8 units) of the packed weights
  uint32_t bOffset;       // offset (floats) of the bias, 0xFFFFFFFF = none
  uint32_t relu;
};

struct NifParams {
  uint32_t numLayers, embedDim, featBase, stride;   // stride in halves
  float maxValue, mean[3];
  int32_t logTonemap;
  NifLayerDesc layers[kNifMaxLayers];
};

struct NifDevice {
  NifParams p{};
  h8* d_weights = nullptr;
  float* d_bias = nullptr;
  uint32_t* d_count = nullptr;     // compaction counter
  uint32_t* d_index = nullptr;     // compacted ray indices
  size_t indexCap = 0;
  bool ok = false;

  bool loaded() const { return ok; }
  void release() {
    if (d_weights) (void)hipFree(d_weights);
    if (d_bias) (void)hipFree(d_bias);
    if (d_count) (void)hipFree(d_count);
    if (d_index) (void)hipFree(d_index);
    d_weights = nullptr; d_bias = nullptr; d_count = nullptr; d_index = nullptr; indexCap = 0; ok = false;
  }

  // Packs W (Keras [rows=K][cols=N], y = x·W) into A-fragment order of v_mfma_f32_16x16x32_f16 for the
  // transposed product: fragment (nt, ks), lane l, element j = W[ks*32 + 8*(l>>4) + j][nt*16 + (l&15)].
  void load(uint32_t numLayers, const float* const* kernels, const float* const* biases, const uint32_t* rows,
            const uint32_t* cols, const uint8_t* relu, uint32_t embedDim, float maxValue, const float mean[3], int32_t logTonemap) {
    release();
    if (numLayers == 0 || numLayers > kNifMaxLayers) throw std::invalid_argument("NIF: unsupported number of layers");
    if (embedDim == 0 || embedDim > 16) throw std::invalid_argument("NIF: embedding dimension must be 1..16");
    const uint32_t F = 4 * embedDim;
    uint32_t hidden = 0;
    for (uint32_t l = 0; l + 1 < numLayers; ++l) hidden = cols[l] > hidden ? cols[l] : hidden;
    if (numLayers == 1) hidden = 0;
    const uint32_t featBase = (hidden + 31u) & ~31u;
    if (cols[numLayers - 1] != 3) throw std::invalid_argument("NIF: last layer must have 3 outputs (BGR)");
    NifParams P{};
    P.numLayers = numLayers; P.embedDim = embedDim; P.featBase = featBase;
    const uint32_t kPadMax = ((featBase + F + 31u) & ~31u);
    P.stride = kPadMax + 16;
                     // +16 halves: rows start 8 banks apart -> the ds_read_b128 lane groups
                                                  // (rows l&15, k-chunk l>>4) hit 16 disjoint 4-bank slots (measured: +8 gave 2-way conflicts)
    P.maxValue = maxValue; P.mean[0] = mean[0]; P.mean[1] = mean[1]; P.mean[2] = mean[2]; P.logTonemap = logTonemap;
    std::vector<_Float16> packed;
    std::vector<float> bias;
    uint32_t width = F;
    bool inputIsFeatures = true;
    for (uint32_t l = 0; l < numLayers; ++l) {
      if (!kernels[l]) throw std::invalid_argument("NIF: null kernel");
      const uint32_t K = rows[l], N = cols[l];
      NifLayerDesc& L = P.layers[l];
      if (inputIsFeatures) {
        if (K != F) throw std::invalid_argument("NIF: first layer must take the " + std::to_string(F) + " Fourier features");
        L.inBase = featBase;
      } else if (K == width) {
        L.inBase = 0;
      } else if (K == width + F && width == featBase) {
        L.inBase = 0;                              // activations followed by the features: the concat of NifModel.cpp:306-309
      } else {
        throw std::invalid_argument("NIF: layer " + std::to_string(l) + " input width does not match (concat needs a uniform hidden width that is a multiple of 32)");
      }
      if (N > 16 * kNifMaxTilesPerWave * 4) throw std::invalid_argument("NIF: layer too wide");
      if (l + 1 < numLayers && N > featBase) throw std::invalid_argument("NIF: internal width error");
      L.kSteps = (K + 31) / 32; L.nTiles = (((N + 15) / 16) + 3u) & ~3u; L.n = N; L.relu = relu[l] ? 1u : 0u;   // tiles padded to a multiple of 4: every wave owns nTiles/4 of them
      L.wOffset = (uint32_t)(packed.size() / 8);
      for (uint32_t nt = 0; nt < L.nTiles; ++nt) {
        for (uint32_t ks = 0; ks < L.kSteps; ++ks)
          for (uint32_t lane = 0; lane < 64; ++lane)
            for (uint32_t j = 0; j < 8; ++j) {
              const uint32_t k = ks * 32 + 8 * (lane >> 4) + j, n = nt * 16 + (lane & 15);
              const float w = (k < K && n < N) ? kernels[l][(size_t)k * N + n] : 0.f;
              packed.push_back((_Float16)w);
            }
        // fragment kSteps of the tile is not weights: lane l's 16 bytes are the four binary32 bias values of the
        // output features its accumulator holds (16nt + 4(l>>4) + 0..3; zeros for a layer without bias). The weight
        // pipeline's loads past the last k-step fetch it, so the bias arrives with the stream, for free.
        for (uint32_t lane = 0; lane < 64; ++lane)
          for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t n = nt * 16 + 4 * (lane >> 4) + q;
            const float b = (biases && biases[l] && n < N) ? biases[l][n] : 0.f;
            _Float16 two[2];
            memcpy(two, &b, 4);
            packed.push_back(two[0]); packed.push_back(two[1]);
          }
      }
      if (biases && biases[l]) { L.bOffset = (uint32_t)bias.size(); for (uint32_t n = 0; n < N; ++n) bias.push_back(biases[l][n]); for (uint32_t n = N; n < L.nTiles * 16; ++n) bias.push_back(0.f); }
      else L.bOffset = 0xFFFFFFFFu;
      width = N;
      inputIsFeatures = false;
    }
    if (hipMalloc(&d_weights, packed.size() * sizeof(_Float16)) != hipSuccess) throw std::runtime_error("NIF: hipMalloc failed");
    (void)hipMemcpy(d_weights, packed.data(), packed.size() * sizeof(_Float16), hipMemcpyHostToDevice);
    if (!bias.empty()) {
      if (hipMalloc(&d_bias, bias.size() * sizeof(float)) != hipSuccess) throw std::runtime_error("NIF: hipMalloc failed");
      (void)hipMemcpy(d_bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice);
    }
    if (hipMalloc(&d_count, sizeof(uint32_t)) != hipSuccess) throw std::runtime_error("NIF: hipMalloc failed");
    p = P;
    ok = true;
  }

  void ensureIndex(size_t n) {
    if (indexCap >= n) return;
    if (d_index) (void)hipFree(d_index);
    d_index = nullptr; indexCap = 0;
    if (hipMalloc(&d_index, n * sizeof(uint32_t)) != hipSuccess) throw std::runtime_error("NIF: hipMalloc failed");
    indexCap = n;
  }
};

// sin and cos of a phase that is a binary16 value (|p| <= 65504). Cody-Waite reduction to
// r = p - k*2pi (k = nearest integer of p/2pi, 2pi split in three parts so the products are exact for the
// <= 11-bit significand of p and |k| < 2^14), then the hardware v_sin/v_cos on r/2pi. Absolute error
// ~1e-6, far below the binary16 rounding (4.9e-4) applied to the result (NifModel.cpp:212-216).
__device__ __forceinline__ void sincos_half_phase(float p, float& sn, float& cs) {
  const float k = rintf(p * 0.15915494309189535f);
  float r = fmaf(-k, 6.28125f, p);                       // 2pi = 6.28125 + 1.9350051879882812e-3 + 3.0199159819e-7
  r = fmaf(-k, 1.9350051879882812e-3f, r);
  r = fmaf(-k, 3.0199159819e-7f, r);
  const float rev = r * 0.15915494309189535f;            // |rev| <= 0.5
  sn = __builtin_amdgcn_sinf(rev);
  cs = __builtin_amdgcn_cosf(rev);
}

// Column swizzle of the activation image X[ray][column]: rows whose index has bit 2 set keep every pair of 16-byte
// chunks swapped (column ^ 8 halves). Row starts are 8 banks apart (stride = 32k + 16 halves), which is what
// ds_read_b128's lane groups want (MI355X_MICROARCH.md, LDS table: 16 disjoint 4-bank slots per group) but puts rows
// r, r+4, r+8, r+12 of a ds_write_b64 group (16 consecutive lanes = 16 rows, one column) on the same banks mod 32:
// the epilogue's stores were 4-way conflicts (SQ_LDS_BANK_CONFLICT = 12 extra cycles per store). With the swap they
// are 2-way (8 array cycles against the 6 the instruction needs anyway), the reads stay conflict-free, and the
// swizzle is a per-lane constant in every address.
__device__ __forceinline__ uint32_t nif_swizzle(uint32_t row) { return ((row >> 2) & 1u) << 3; }

// PreProcessEscapedRays + compaction. u/v are written for EVERY ray (0 for rays that did not escape,
// as the reference does); `index[0..*count)` receives the escaped rays' indices, one atomic per wave.
__global__ void __launch_bounds__(256) escaped_uv_kernel(const mi_trace_result* rays, uint32_t n, float azimuthRotation,
                                                         float* u, float* v, uint32_t* index, uint32_t* count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool escaped = false;
  float uu = 0.f, vv = 0.f;
  if (i < n) {
    const mi_hit_record& h = rays[i].h;
    if (h.flags & MI_FLAG_ESCAPED) {
      escaped = true;
      const float twoPi = (float)(2.0 * 3.14159265358979323846264338327950288);
      const float invPi = (float)(1.0 / 3.14159265358979323846264338327950288);
      const float inv2Pi = (float)(1.0 / (2.0 * 3.14159265358979323846264338327950288));
      const float theta = acosf(h.r.direction.y);
      float phi = atan2f(h.r.direction.z, h.r.direction.x) + azimuthRotation;
      if (phi < 0.f) phi += twoPi;
      else if (phi > twoPi) phi -= twoPi;
      uu = theta * invPi;
      vv = phi * inv2Pi;
    }
    u[i] = uu; v[i] = vv;
  }
  const unsigned long long mask = __ballot(escaped);
  const uint32_t lane = threadIdx.x & 63;
  uint32_t base = 0;
  if (lane == 0 && mask) base = atomicAdd(count, (uint32_t)__popcll(mask));
  base = __shfl(base, 0);
  if (escaped) index[base + __popcll(mask & ((1ull << lane) - 1ull))] = i;
}

// A RUN of dense layers [l0, l1) for one wave: all of them give the wave TN output-feature tiles (nt = ng + 4a) x MT ray
// tiles, fully unrolled, with their own accumulators. Per layer: the k-loop (the next k-steps' weight fragments in
// flight while the current ones feed the matrix cores), then the epilogue (bias, ReLU, binary16 store - or, for the
// network's final layer, decode + environment add).
//
// The loop is ROTATED: an iteration asks for its layer's k-step-0 fragments, THEN runs the previous layer's epilogue
// (with its two barriers), THEN its own k-loop; the last layer's epilogue follows the loop. So the L2 latency of a
// layer's first fragments is spent under the previous layer's epilogue instead of in front of its first MFMA, and no
// register with a load in flight is carried around the loop's back edge: what IS carried are the accumulators and the
// bias, ordinary values. (An earlier form issued the fragments at the END of the previous iteration; hipcc gave them
// registers of their own and copied those into the k-loop's at the loop header - before they had landed.
// tests/test_asm_pipeline_audit.py now follows the control flow and catches that.)
// LAST: the run is the network's final layer alone (decode + global stores in its epilogue).
template <uint32_t TN, uint32_t MT, bool LAST>
__device__ __forceinline__ void nif_dense_layers(const NifParams& P, uint32_t l0, uint32_t l1, _Float16* X, uint32_t stride,
                                                 uint32_t rowBase, uint32_t ng, uint32_t lane, const h8* __restrict__ weights,
                                                 const float* __restrict__ bias, uint32_t row0, uint32_t total,
                                                 const uint32_t* __restrict__ idx, float* __restrict__ bgrOut, mi_trace_result* rays, bool scatter
#if MI_NIF_STAMPS
                                                 , unsigned long long (&stampSum)[8]
#endif
                                                 ) {
  constexpr bool last = LAST;
  f4v acc[TN][MT];
  f4v bv[TN];
  const f4v zero4 = {0.f, 0.f, 0.f, 0.f};
  // Fragment (nt, ks) is 1 KiB at weights + wOffset + (nt*(kSteps+1) + ks)*64 (+ lane): a wave-uniform base, so the
  // loads use the SGPR-base + VGPR-offset form and need ONE address VGPR (lane*16) for all of them. Fragment kSteps of
  // a tile is its bias (NifDevice::load).
  const uint32_t laneOff = lane * 16u;
  const uint32_t ngU = (uint32_t)__builtin_amdgcn_readfirstlane((int)ng);

  // bias / ReLU / binary16 store of layer EL (or decode + environment add), between the two barriers that separate
  // its k-loop's reads of X from these writes and these writes from the next k-loop's reads
  const uint32_t laneCol = (4 * (lane >> 4)) ^ nif_swizzle(lane);   // row = lane mod 16 (+ multiples of 16): the swizzle is a lane constant
  auto epilogue = [&](const NifLayerDesc& EL) {
    MI_STAMP(tE0);
    __syncthreads();   // every wave has read its inputs: X[.., 0..N) may be overwritten
    MI_STAMP(tE1);
#pragma unroll
    for (uint32_t a = 0; a < TN; ++a) {
      const uint32_t nt = ng + 4 * a;
      if (16 * nt < EL.n) {
        // D fragment: rows (= output features) 16nt + 4(lane>>4) + reg, column (= ray) 16m + (lane&15)
        const uint32_t f0 = 16 * nt + laneCol;
#pragma unroll
        for (uint32_t m = 0; m < MT; ++m) {
          f4v y = acc[a][m] + bv[a];
          const uint32_t r = rowBase + 16 * m + (lane & 15);
          if (!last) {
            // ReLU on the rounded halves (two packed max): rounding is monotone and keeps the sign, so
            // max(round(y), 0) == round(max(y, 0))
            h4 yh = {(_Float16)y[0], (_Float16)y[1], (_Float16)y[2], (_Float16)y[3]};
            if (EL.relu) yh = __builtin_elementwise_max(yh, (h4){(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f});
            *reinterpret_cast<h4*>(&X[r * stride + f0]) = yh;
          } else if (nt == 0 && (lane >> 4) == 0 && row0 + r < total) {
            // decode (NifModel.cpp:222-246): y*max + mean, exp for log-tonemapped models
            if (EL.relu) {
#pragma unroll
              for (int q = 0; q < 4; ++q) y[q] = y[q] > 0.f ? y[q] : 0.f;
            }
            float o[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              o[c] = y[c] * P.maxValue + P.mean[c];
              if (P.logTonemap) o[c] = expf(o[c]);
            }
            const uint32_t row = row0 + r;
            if (bgrOut) { const size_t dst = scatter ? (size_t)idx[row] : (size_t)row; bgrOut[3 * dst] = o[0]; bgrOut[3 * dst + 1] = o[1]; bgrOut[3 * dst + 2] = o[2]; }
            if (rays) {
              mi_trace_result* res = rays + (idx ? idx[row] : row);
              const mi_vec3 tp = res->h.throughput;
              res->rgb.x += tp.x * o[2];          // BGR -> RGB (codelets/TraceCodelets.cpp:376)
              res->rgb.y += tp.y * o[1];
              res->rgb.z += tp.z * o[0];
            }
          }
        }
      }
    }
    MI_STAMP(tE2);
    __syncthreads();
    MI_STAMP(tE3);
    MI_STAMP_ADD(1, tE0, tE1); MI_STAMP_ADD(2, tE1, tE2); MI_STAMP_ADD(3, tE2, tE3);
  };

  for (uint32_t l = l0; l < l1; ++l) {
    const NifLayerDesc L = P.layers[l];
    const uint32_t kSteps = L.kSteps;
    // Weight stream: three named fragment sets (k-steps 3j, 3j+1, 3j+2), each loaded two k-steps before it is used.
    // hipcc sinks ordinary loads down to their first use here (it minimises live ranges at this register
    // pressure: the .s showed "load, s_waitcnt vmcnt(0), mfma" every k-step), so the loads are inline asm, invisible
    // to its scheduler, and their completion is counted by hand (cdna_hip_programming.md §5.7 form ii): loads return
    // in order, every step issues exactly TN of them, so "all but the newest 2*TN have landed" is the set about to be
    // used. Every destination is named in the wait ("; landed vN" for the audit), which keeps the consumers below it.
    // The counts assume that only LOADS join the queue meanwhile (they return in order, so a younger one only makes a
    // wait stricter); tests/test_asm_pipeline_audit.py checks the generated code for stores / atomics / scratch in
    // flight together with these loads and for any touch of a destination before it has landed.
    h8 wA[TN], wB[TN], wC[TN];
    // One scalar base per tile, fixed for the layer (all-scalar arithmetic on kernel arguments + ngU: SALU results, which
    // a VMEM instruction may read without wait states - a v_readfirstlane result would need 5: §5.7 item 2), and ONE
    // vector offset per k-step shared by the TN loads: lane*16 + 1 KiB * step. The address arithmetic used to be five
    // scalar instructions per load; a lone wave issues about one instruction per four cycles beside its MFMAs
    // (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'), and those 25 per k-step were paid in full.
    uint64_t wTile[TN];
#pragma unroll
    for (uint32_t a = 0; a < TN; ++a) {
      const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((ngU + 4 * a) * (kSteps + 1)) << 10));
      wTile[a] = (uint64_t)(uintptr_t)(weights + L.wOffset) + off;
    }
    auto loadW = [&](h8 (&w)[TN], uint32_t ks) {
      ks = ks < kSteps ? ks : kSteps;                        // past the end: the tile's bias fragment (keeps the count uniform)
      const uint32_t voff = laneOff + (ks << 10);
#pragma unroll
      for (uint32_t a = 0; a < TN; ++a)
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(w[a]) : "v"(voff), "s"(wTile[a]));
    };
    auto landed = [&](h8 (&w)[TN], auto outstanding) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(outstanding)::value));
#pragma unroll
      for (uint32_t a = 0; a < TN; ++a) asm volatile("; landed %0" : "+v"(w[a]));
    };
    // k-step 0 first: it is in flight while the previous layer's epilogue runs
    loadW(wA, 0);
    if (l > l0) epilogue(P.layers[l - 1]);

    // Activation fragments come from LDS through inline-asm reads as well: hipcc otherwise emits "ds_read; s_waitcnt
    // lgkmcnt(0); TN mfma" per ray tile, exposing the LDS latency MT times per k-step. Same hand-counted scheme: reads
    // return in order, so lgkmcnt(2) = all but the newest two. The three-slot ring runs two ray tiles ahead of the MFMAs
    // ACROSS k-steps: the last two tiles of a step already ask for the first two fragments of the next one (stamps
    // showed a wave that restarts the ring every k-step, waiting out one LDS round trip in front of 30 MFMAs, keeping
    // the matrix pipe only 56 % busy when it has it to itself). Tile m of every step uses slot m % kRing, and kRing
    // divides MT (3 slots for 6 tiles, 4 for 4), so the slots - the registers - of a tile are the same in every step.
    const uint32_t xAddr0 = (uint32_t)(uintptr_t)X + ((rowBase + (lane & 15)) * stride + L.inBase + ((8 * (lane >> 4)) ^ nif_swizzle(lane))) * 2u;
    const uint32_t mStep = 16u * stride * 2u;
    constexpr uint32_t kRing = (MT % 3u == 0u) ? 3u : 4u;
    static_assert(MT % kRing == 0 && MT >= 2, "the activation ring runs two tiles ahead and must divide the tiles of a step");
    h8 xb[kRing];
    auto readX = [&](h8& x, uint32_t addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(x) : "v"(addr)); };
    auto step = [&](const h8 (&w)[TN], uint32_t ks, auto first) {
      // B fragment: X^T[k][ray] = X[ray = rowBase + 16m + (lane&15)][k = 32ks + 8(lane>>4) + j]
      const uint32_t a0 = xAddr0 + 64u * ks;
      const uint32_t aNext = ks + 1 < kSteps ? a0 + 64u : a0;             // behind the last step: a harmless re-read, retired by the drain
#pragma unroll
      for (uint32_t m = 0; m < MT; ++m) {
        if (m + 2 < MT) readX(xb[(m + 2) % kRing], a0 + (m + 2) * mStep);
        else readX(xb[(m + 2) % kRing], aNext + (m + 2 - MT) * mStep);
        asm volatile("s_waitcnt lgkmcnt(2)\n\t; landed %0" : "+v"(xb[m % kRing]));
        // the first k-step's MFMAs take C = 0; the bias is added in the epilogue. It is not fetched by a load of its
        // own (a compiler-issued load in front of the hand-counted ones made hipcc drain the whole queue before the
        // first MFMA of every layer): it rides the weight stream as the fragment behind each tile's last k-step.
#pragma unroll
        for (uint32_t a = 0; a < TN; ++a)
          acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[a], xb[m % kRing], decltype(first)::value ? zero4 : acc[a][m], 0, 0, 0);
      }
    };
    using TwoSets = std::integral_constant<int, 2 * TN>;
    MI_STAMP(tK0);
    readX(xb[0], xAddr0);
    readX(xb[1], xAddr0 + mStep);
    loadW(wB, 1);
    loadW(wC, 2); landed(wA, TwoSets{}); step(wA, 0, std::true_type{});
    loadW(wA, 3); landed(wB, TwoSets{}); if (1 < kSteps) step(wB, 1, std::false_type{});
    loadW(wB, 4); landed(wC, TwoSets{}); if (2 < kSteps) step(wC, 2, std::false_type{});
    for (uint32_t ks = 3; ks < kSteps; ks += 3) {
      loadW(wC, ks + 2); landed(wA, TwoSets{}); step(wA, ks, std::false_type{});
      loadW(wA, ks + 3); landed(wB, TwoSets{}); if (ks + 1 < kSteps) step(wB, ks + 1, std::false_type{});
      loadW(wB, ks + 4); landed(wC, TwoSets{}); if (ks + 2 < kSteps) step(wC, ks + 2, std::false_type{});
    }
    // drain: the two weight sets and the two activation fragments still in flight own their registers until they land
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
#pragma unroll
    for (uint32_t a = 0; a < TN; ++a) { asm volatile("; landed %0" : "+v"(wA[a])); asm volatile("; landed %0" : "+v"(wB[a])); asm volatile("; landed %0" : "+v"(wC[a])); }
#pragma unroll
    for (uint32_t q = 0; q < kRing; ++q) asm volatile("; landed %0" : "+v"(xb[q]));
    // the last load into wA was past the end for every kSteps >= 1: it holds the tiles' bias fragments
#pragma unroll
    for (uint32_t a = 0; a < TN; ++a) bv[a] = __builtin_bit_cast(f4v, wA[a]);
    MI_STAMP(tK1);
    MI_STAMP_ADD(0, tK0, tK1);
  }
  epilogue(P.layers[l1 - 1]);
}

// The MLP. rows: `numRows` (or *countPtr when countPtr != nullptr) entries; entry r reads
// u[idx ? idx[r] : r], v[...]; result goes to bgrOut[3*r..] (stand-alone) and/or is added to
// rays[idx[r]].rgb as throughput * (b,g,r)->(r,g,b) (PostProcessEscapedRays).
// TILES = most output-feature tiles any layer gives a wave; layers with fewer run their own instantiation.
template <uint32_t TILES, uint32_t MT, uint32_t RG>
__global__ void __launch_bounds__(256 * RG, (MT == 6 && RG == 1) ? 2 : 1) nif_mlp_kernel(NifParams P, const h8* __restrict__ weights, const float* __restrict__ bias,
                                                      const float* __restrict__ u, const float* __restrict__ v,
                                                      const uint32_t* __restrict__ idx, const uint32_t* __restrict__ countPtr,
                                                      uint32_t numRows, float* __restrict__ bgrOut, mi_trace_result* rays, bool scatter) {
  constexpr uint32_t kNifRows = 16u * MT * RG;                    // rays per workgroup pass
  extern __shared__ __attribute__((aligned(16))) _Float16 X[];   // [kNifRows][P.stride]
  __shared__ float uvS[2 * kNifRows];                            // the pass's environment coordinates, fetched once
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t ng = wave & 3u, rowBase = 16u * MT * (wave >> 2);   // output-feature group, row group
  const uint32_t total = countPtr ? *countPtr : numRows;
  const uint32_t stride = P.stride;
  const uint32_t E = P.embedDim, F = 4 * E;
#if MI_NIF_STAMPS
  unsigned long long stampSum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long tBegin = nif_now();
#endif

  // The K-padding columns behind the features are zero and nothing ever writes them again: once per workgroup, not
  // once per pass. Items are numbered column-major (e = column * kNifRows + row) so the row/column split divides by a
  // compile-time constant.
  {
    const uint32_t padCols = stride - P.featBase - F;
    for (uint32_t e = tid; e < kNifRows * padCols; e += blockDim.x) {
      const uint32_t r = e % kNifRows, c = e / kNifRows;
      X[r * stride + ((P.featBase + F + c) ^ nif_swizzle(r))] = (_Float16)0.f;
    }
  }
  // Two row groups (RG == 2) run HALF A PHASE APART: the second group passes one extra barrier here and the first one
  // at the very end, so every workgroup barrier still sees all eight waves but group 1's n-th phase runs beside group
  // 0's (n+1)-th. The phases of a group alternate between a k-loop (matrix cores) and an epilogue / staging phase
  // (VALU + LDS), and each SIMD holds one wave of either group: with the offset a SIMD's matrix pipe always has
  // exactly one wave feeding it while its partner converts and stores, instead of both waves competing for the pipe
  // and then both leaving it idle (MI355X_MICROARCH.md, "Two waves per SIMD", item 5: a rendezvous pays when the
  // paired intervals are complementary). The groups share no rows, so every phase touches only its own group's part
  // of X and uvS and the barriers order exactly what they ordered before.
  constexpr uint32_t kGroupRows = 16u * MT, kGroupThreads = 256u;
  const uint32_t gtid = tid & (kGroupThreads - 1u);
  if (RG == 2) {
    __syncthreads();
    if (wave >= 4) __builtin_amdgcn_s_barrier();
  }
  for (uint32_t row0 = blockIdx.x * kNifRows; row0 < total; row0 += gridDim.x * kNifRows) {
    MI_STAMP(tS0);
    __syncthreads();   // previous pass finished with X
    // every row's (u, v) is needed by 2E feature columns: fetch it once (index, then coordinate: two dependent global
    // loads per row instead of per feature) and hand it round through LDS
    // (each row group stages its own rows with its own 256 threads)
    for (uint32_t rr = gtid; rr < kGroupRows; rr += kGroupThreads) {
      const uint32_t r = rowBase + rr, row = row0 + r;
      float cu = 0.f, cv = 0.f;
      if (row < total) { const uint32_t src = idx ? idx[row] : row; cu = u[src]; cv = v[src]; }
      uvS[r] = cu; uvS[kNifRows + r] = cv;
    }
    __syncthreads();
    for (uint32_t e = gtid; e < kGroupRows * E * 2; e += kGroupThreads) {
      const uint32_t r = rowBase + e % kGroupRows, q = e / kGroupRows, isV = q >= E ? 1u : 0u, j = q - isV * E;
      const float coord = uvS[isV * kNifRows + r];
      const float nrm = (coord - 1.f) * 2.f;                                  // NifModel.cpp:203-205
      const float phase = (float)(_Float16)(nrm * (float)(1u << j));           // cast to HALF before sin/cos (:212)
      float fs, fc;
      sincos_half_phase(phase, fs, fc);
      const _Float16 sn = (_Float16)fs, cs = (_Float16)fc;
      // feature order [sin u | sin v | cos u | cos v] (NifModel.cpp:216)
      X[r * stride + ((P.featBase + q) ^ nif_swizzle(r))] = sn;
      X[r * stride + ((P.featBase + 2 * E + q) ^ nif_swizzle(r))] = cs;
    }
    __syncthreads();
    MI_STAMP(tS1);
    MI_STAMP_ADD(4, tS0, tS1);

    for (uint32_t l = 0; l < P.numLayers;) {
      // consecutive layers that give a wave the same number of output-feature tiles run as one pipelined loop; the
      // network's final layer (decode + global stores in its epilogue) is always a run of its own
      const uint32_t tilesLayer = P.layers[l].nTiles >> 2;            // (wave-uniform)
      const bool finalLayer = l + 1 == P.numLayers;
      uint32_t l1 = l + 1;
      while (!finalLayer && l1 + 1 < P.numLayers && (P.layers[l1].nTiles >> 2) == tilesLayer) ++l1;
#if MI_NIF_STAMPS
#define MI_NIF_STAMP_ARG , stampSum
#else
#define MI_NIF_STAMP_ARG
#endif
#define MI_NIF_RUN(TN) do { if (finalLayer) nif_dense_layers<TN, MT, true>(P, l, l1, X, stride, rowBase, ng, lane, weights, bias, row0, total, idx, bgrOut, rays, scatter MI_NIF_STAMP_ARG); \
                            else nif_dense_layers<TN, MT, false>(P, l, l1, X, stride, rowBase, ng, lane, weights, bias, row0, total, idx, bgrOut, rays, scatter MI_NIF_STAMP_ARG); } while (0)
      if (tilesLayer == TILES) MI_NIF_RUN(TILES);
      if constexpr (TILES > 1) { if (tilesLayer == 1) MI_NIF_RUN(1); }
      if constexpr (TILES > 2) { if (tilesLayer == 2) MI_NIF_RUN(2); }
      if constexpr (TILES > 3) { if (tilesLayer == 3) MI_NIF_RUN(3); }
      if constexpr (TILES > 4) { if (tilesLayer == 4) MI_NIF_RUN(4); }
      if constexpr (TILES > 5) { if (tilesLayer == 5) MI_NIF_RUN(5); }
#undef MI_NIF_RUN
      l = l1;
    }
  }
  if (RG == 2 && wave < 4) __builtin_amdgcn_s_barrier();
#if MI_NIF_STAMPS
  stampSum[5] = nif_now() - tBegin; stampSum[6] = 1;
  if (lane == 0 && (wave & 3u) == 1u)
    for (int q = 0; q < 7; ++q) atomicAdd(&nif_stamps[q], stampSum[q]);
#endif
}

// bgrOut: result of row r at bgrOut[3r..] - or, with `scatter`, at bgrOut[3*idx[r]..] (the row's own slot)
inline void nif_launch_mlp(const NifDevice& nif, const float* u, const float* v, const uint32_t* idx, const uint32_t* countPtr,
                           uint32_t numRows, float* bgrOut, mi_trace_result* rays, hipStream_t stream, bool scatter = false, uint32_t shape = 0) {
  if (numRows == 0) return;
  uint32_t maxTiles = 1;
  for (uint32_t l = 0; l < nif.p.numLayers; ++l) maxTiles = std::max(maxTiles, nif.p.layers[l].nTiles / 4);
  // shape: MT ray tiles per wave x RG row groups. Scene option "nif_shape" = w6|t6|t4 (`shape` 0|1|2) overrides the default (w6).
  // (4 waves x 64 / 128 / 192 rays were measured too - DESIGN.md §6 - and are not kept: at their register
  // pressure hipcc spills around the hand-counted asm loads, which the .s audit in tests/ flags.)
  uint32_t mt = 6, rg = 1;
  if (shape != 0) { rg = 2; mt = (shape == 1) ? 6 : 4; }
  if (!(mt == 4 && rg == 2) && ((size_t)16 * mt * rg * nif.p.stride * sizeof(_Float16) > kNifMaxLdsBytes || maxTiles > 5)) { mt = 4; rg = 2; }
  auto launch = [&](auto kern, uint32_t rowsPerPass, uint32_t threads) {
    size_t lds = (size_t)rowsPerPass * nif.p.stride * sizeof(_Float16);
#if MI_NIF_STAMPS
    if (getenv("MI_NIF_DIAG_ONE_WG")) lds = kNifMaxLdsBytes;   // diagnostic: one workgroup per CU, i.e. one wave per SIMD for the 4-wave shape
#endif
    uint32_t blocks = (numRows + rowsPerPass - 1) / rowsPerPass;
    if (blocks > 256 * 8) blocks = 256 * 8;          // grid-stride beyond 8 passes per CU
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kNifMaxLdsBytes);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, stream, nif.p, nif.d_weights, nif.d_bias, u, v, idx, countPtr, numRows, bgrOut, rays, scatter);
  };
  if (mt == 6 && rg == 1) {
    if (maxTiles <= 2) launch(nif_mlp_kernel<2, 6, 1>, 96, 256);
    else launch(nif_mlp_kernel<5, 6, 1>, 96, 256);
  } else if (mt == 6) {
    if (maxTiles <= 2) launch(nif_mlp_kernel<2, 6, 2>, 192, 512);
    else launch(nif_mlp_kernel<5, 6, 2>, 192, 512);
  } else {
    if (maxTiles <= 2) launch(nif_mlp_kernel<2, 4, 2>, 128, 512);
    else if (maxTiles <= 4) launch(nif_mlp_kernel<4, 4, 2>, 128, 512);
    else if (maxTiles <= 5) launch(nif_mlp_kernel<5, 4, 2>, 128, 512);
    else launch(nif_mlp_kernel<6, 4, 2>, 128, 512);
  }
}

// Replays the reference's per-sample order for one launch of `samples` slots per pixel (trace_wavefront.hpp,
// WaveExtras): rgb += color of the sample's path (codelets/TraceCodelets.cpp:262), then, if it escaped,
// rgb += throughput * env with BGR -> RGB (PostProcessEscapedRays, codelets :361-382) - within each segment of
// 2^segShift samples; the segments' partial sums are added in segment order, the render's segment 0 accumulating
// onto the incoming rgb directly (ray_math.h segment_samples; a launch starts at a segment boundary).
__global__ void __launch_bounds__(256) nif_accumulate_kernel(mi_trace_result* rays, uint32_t n, uint32_t samples, uint32_t segShift, uint32_t firstSegment,
                                                             const float* __restrict__ color, const float* __restrict__ tp, const float* __restrict__ u,
                                                             const float* __restrict__ bgr) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  mi_vec3 total = rays[i].rgb;
  for (uint32_t s0 = 0; s0 < samples; s0 += 1u << segShift) {
    const bool first = firstSegment == 0 && s0 == 0;
    mi_vec3 rgb = first ? total : mi_vec3{0.f, 0.f, 0.f};
    const uint32_t s1 = min(samples, s0 + (1u << segShift));
    for (uint32_t s = s0; s < s1; ++s) {
      const size_t q = (size_t)s * n + i;
      rgb.x += color[3 * q]; rgb.y += color[3 * q + 1]; rgb.z += color[3 * q + 2];
      // u == -1 is the "did not escape" mark of the trace kernel; an escaped ray's u is theta / pi in [0, 1] - or NaN when
      // the direction's y rounded to just outside [-1, 1] (acosf), and such a ray still takes its environment term, as
      // in the reference's per-sample form (found by tests/fuzz_parity.py nif, seed 20261005 case 8486)
      if (u[q] != -1.f) {
        rgb.x += tp[3 * q] * bgr[3 * q + 2];
        rgb.y += tp[3 * q + 1] * bgr[3 * q + 1];
        rgb.z += tp[3 * q + 2] * bgr[3 * q];
      }
    }
    if (first) total = rgb; else { total.x += rgb.x; total.y += rgb.y; total.z += rgb.z; }
  }
  rays[i].rgb = total;
}

// Sample-at-a-time NIF renders: at a segment boundary the finished segments' sum moves to `total` and rgb restarts
// from zero (segment 0 had started from the incoming rgb); `finish` adds the last segment's partial sum back.
__global__ void __launch_bounds__(256) nif_segment_roll_kernel(mi_trace_result* rays, float* __restrict__ total, uint32_t n, uint32_t segment, uint32_t finish) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const mi_vec3 rgb = rays[i].rgb;
  mi_vec3 t = rgb;
  if (segment > 1 || finish) { t.x = total[3 * i] + rgb.x; t.y = total[3 * i + 1] + rgb.y; t.z = total[3 * i + 2] + rgb.z; }
  if (finish) { rays[i].rgb = t; return; }
  total[3 * i] = t.x; total[3 * i + 1] = t.y; total[3 * i + 2] = t.z;
  rays[i].rgb = {0.f, 0.f, 0.f};
}

// mi_nif_infer_device: every row is evaluated (no compaction)
inline void nif_infer(NifDevice& nif, const float* d_u, const float* d_v, float* d_bgr, size_t n, size_t maxBatch, hipStream_t stream, uint32_t shape = 0) {
  const size_t chunk = maxBatch ? maxBatch : n;
  for (size_t off = 0; off < n; off += chunk) {
    const size_t cnt = (n - off < chunk) ? (n - off) : chunk;
    nif_launch_mlp(nif, d_u + off, d_v + off, nullptr, nullptr, (uint32_t)cnt, d_bgr + 3 * off, nullptr, stream, false, shape);
  }
}

// One sample's environment pass over the whole ray stream (src/IpuScene.cpp:571-583)
inline void nif_env_pass(NifDevice& nif, mi_trace_result* d_rays, uint32_t n, float azimuthRadians, float* d_u, float* d_v,
                         float* d_bgr, size_t /*maxBatch*/, hipStream_t stream, uint32_t shape = 0) {
  nif.ensureIndex(n);
  (void)hipMemsetAsync(nif.d_count, 0, sizeof(uint32_t), stream);
  hipLaunchKernelGGL(escaped_uv_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_rays, n, azimuthRadians, d_u, d_v, nif.d_index, nif.d_count);
  nif_launch_mlp(nif, d_u, d_v, nif.d_index, nif.d_count, n, nullptr, d_rays, stream, false, shape);
}

}  // namespace mi
