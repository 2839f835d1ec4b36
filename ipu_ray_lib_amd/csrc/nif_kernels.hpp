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
// workgroup and feeds 6 MFMAs. Activations live in LDS as a k-chunk-major binary16 image (nif_x_byte); the dense
// layers are evaluated transposed, Y^T = W^T · X^T, so that the MFMA result fragment of a lane is 4 consecutive
// output features of ONE ray and goes back to LDS as one 8-byte store. W^T is pre-packed on the host in exact
// A-fragment order, one stream per 16-feature tile headed by the tile's bias, so every weight load is a fully
// coalesced 1 KiB wave read served from L2. The k-loop is written instruction by instruction (asm MFMAs, loads, LDS
// reads and hand-counted waits placed between them: nif_dense_layers); three weight fragment sets and a three-slot
// activation ring are in flight. The 8-wave shapes (64 / 96 rays per row group) are kept selectable
// (MI_RAYLIB_NIF_SHAPE) for widths whose LDS image does not fit and for the measurements in DESIGN.md.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <stdexcept>
#include <string>
#include <vector>

#include "ray_math.h"
#include "nif_regs_pack.hpp"
#include "../../include/mi_raylib.h"

#ifndef MI_RAYLIB_VARIANTS
#define MI_RAYLIB_VARIANTS 0
#endif
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

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));   // a 16-byte fragment as the asm statements of the k-loop see it (legal as an in/out operand; 8 halves are not)

// Workgroup shapes of the same kernel (template parameters MT = 16-ray tiles per wave, RG = row groups):
//   w6 (default) : 4 waves x 96 rays, two workgroups per CU; every packed weight fragment is fetched once per
//                  workgroup and feeds 6 MFMAs. Needs 96 x (stride + 32) x 2 B of LDS per workgroup.
//   t6, t4       : 8 waves = 4 output-feature groups x 2 row groups of 96 / 64 rays (t4: any width up to 384).
constexpr uint32_t kNifMaxLdsBytes = 160 * 1024 - 2048;   // dynamic LDS: what is left beside the static coordinate staging (2 x 192 floats)
constexpr uint32_t kNifMaxTilesPerWave = 6;   // output-feature tiles (of 16) per wave: supports widths up to 384

struct NifLayerDesc {
  uint32_t kSteps;        // ceil(K/32)
  uint32_t nTiles;        // ceil(N/16)
  uint32_t n;             // real output width
  uint32_t inBase;        // first LDS column of the layer's input
  uint32_t wOffset;       // offset (in h8 units) of the packed weights
  uint32_t relu;
};

struct NifParams {
  uint32_t numLayers, embedDim, featBase, stride;   // stride: halves per ray in the LDS image (features included, padded to 32)
  float maxValue, mean[3];
  int32_t logTonemap;
  NifLayerDesc layers[kNifMaxLayers];
};

struct NifDevice {
  NifParams p{};
  h8* d_weights = nullptr;
  uint32_t* d_count = nullptr;     // compaction counter
  uint32_t* d_index = nullptr;     // compacted ray indices
  unsigned long long* d_clock = nullptr;      // K3a: {shader cycles, 100-MHz ticks} one wave spent in the last launch (mi_get_nif_clock)
  size_t indexCap = 0;
  bool ok = false;
  NifRegsDevice regs;              // the same model packed as a stream of 1-KiB A fragments in consumption order for the register-resident kernels: K3a
                                   // (nif_asm_kernel.hpp) and, in the variants build, K3r (nif_regs_kernel.hpp); regs.ok only for the shapes they cover

  bool loaded() const { return ok; }
  void release() {
    regs.release();
    if (d_weights) (void)hipFree(d_weights);
    if (d_count) (void)hipFree(d_count);
    if (d_index) (void)hipFree(d_index);
    if (d_clock) (void)hipFree(d_clock);
    d_weights = nullptr; d_count = nullptr; d_index = nullptr; d_clock = nullptr; indexCap = 0; ok = false;
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
    P.stride = kPadMax;   // halves per ray (a multiple of 32); the image is k-chunk major (nif_x_byte)
    P.maxValue = maxValue; P.mean[0] = mean[0]; P.mean[1] = mean[1]; P.mean[2] = mean[2]; P.logTonemap = logTonemap;
    std::vector<_Float16> packed;
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
        // fragment 0 of the tile is not weights: lane l's 16 bytes are the four binary32 bias values of the output
        // features its accumulator holds (16nt + 4(l>>4) + 0..3; zeros for a layer without bias). It heads the
        // tile's stream, lands with the first weight fragments and is the C operand of the first k-step's MFMAs.
        for (uint32_t lane = 0; lane < 64; ++lane)
          for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t n = nt * 16 + 4 * (lane >> 4) + q;
            const float b = (biases && biases[l] && n < N) ? biases[l][n] : 0.f;
            _Float16 two[2];
            memcpy(two, &b, 4);
            packed.push_back(two[0]); packed.push_back(two[1]);
          }
        for (uint32_t ks = 0; ks < L.kSteps; ++ks)
          for (uint32_t lane = 0; lane < 64; ++lane)
            for (uint32_t j = 0; j < 8; ++j) {
              const uint32_t k = ks * 32 + 8 * (lane >> 4) + j, n = nt * 16 + (lane & 15);
              const float w = (k < K && n < N) ? kernels[l][(size_t)k * N + n] : 0.f;
              packed.push_back((_Float16)w);
            }
      }
      width = N;
      inputIsFeatures = false;
    }
    if (hipMalloc(&d_weights, packed.size() * sizeof(_Float16)) != hipSuccess) throw std::runtime_error("NIF: hipMalloc failed");
    (void)hipMemcpy(d_weights, packed.data(), packed.size() * sizeof(_Float16), hipMemcpyHostToDevice);
    if (hipMalloc(&d_count, sizeof(uint32_t)) != hipSuccess) throw std::runtime_error("NIF: hipMalloc failed");
    if (hipMalloc(&d_clock, 2 * sizeof(unsigned long long)) != hipSuccess) throw std::runtime_error("NIF: hipMalloc failed");
    (void)hipMemset(d_clock, 0, 2 * sizeof(unsigned long long));
    (void)regs.load(numLayers, kernels, biases, rows, cols, relu, embedDim, maxValue, mean, logTonemap);
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

// The activation image in LDS, k-chunk major: X[kc][ray][32 halves], kc = column / 32, and inside a ray's 64 bytes the four
// 16-byte pieces are stored at piece ^ S[(ray >> 2) & 3], S = {0, 2, 3, 1}.
//   * the B fragment of (k-step ks, ray tile m) for lane (ray r = lane & 15, piece g = lane >> 4) is at
//       laneBase + ks * (ROWS * 64) + m * 1024
//     so the whole k-loop addresses its ds_read_b128 with ONE VGPR and compile-time offsets (a wave issues about one
//     instruction per four cycles beside its MFMAs; with a row-major image every read cost a v_add of its own);
//   * ds_read_b128 serves a wave in four 16-lane groups, {0-3,12-15,20-27} ... (MI355X_MICROARCH.md, LDS table): with
//     the swap every group's 16 lanes hit 16 different 4-bank slots (checked over all groups / tiles / k-steps by
//     enumeration, tests/test_host_and_abi.py) - conflict-free without padding, so the image is 12 x 96 x 64 B = 72 KiB;
//   * the epilogue's ds_write_b64 (16 consecutive lanes = 16 rays, one column) is 2-way: 8 LDS cycles against the 6
//     the instruction needs anyway (row-major with 8-bank row offsets it was 4-way: SQ_LDS_BANK_CONFLICT = 12 extra
//     cycles per store).
// An index-list entry that names no ray: the trace kernel reserves room in the escaped-slot list 2 048 entries at a time and marks the whole
// 256-entry blocks a wave leaves unused (trace_wavefront.hpp kEnvHole). Every MLP kernel skips such rows (no coordinate load, no result);
// K3a / K3b skip a pass of 256 holes altogether.
constexpr uint32_t kNifHole = 0xFFFFFFFFu;
constexpr uint32_t kNifGenerations = 64;      // workgroups launched per resident slot (nif_launch_mlp)

template <uint32_t ROWS>
__device__ __forceinline__ uint32_t nif_x_byte(uint32_t ray, uint32_t col) {
  const uint32_t sw = (0x78u >> (2u * ((ray >> 2) & 3u))) & 3u;            // S = {0, 2, 3, 1}, two bits each: 0b01'11'10'00
  return (col >> 5) * (ROWS * 64u) + ray * 64u + ((((col >> 3) & 3u) ^ sw) << 4) + (col & 7u) * 2u;
}
template <typename F, uint32_t... I>
__device__ __forceinline__ void nif_unroll(F&& f, std::integer_sequence<uint32_t, I...>) { (f(std::integral_constant<uint32_t, I>{}), ...); }

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
// tiles, fully unrolled, with their own accumulators. Per layer: the k-loop, then the epilogue (ReLU, binary16 store -
// or, for the network's final layer, decode + environment add).
//
// The loop is ROTATED: an iteration asks for its layer's bias and k-step-0 fragments, THEN runs the previous layer's
// epilogue (with its two barriers), THEN its own k-loop; the last layer's epilogue follows the loop. So the L2 latency
// of a layer's first fragments is spent under the previous layer's epilogue instead of in front of its first MFMA, and
// no register with a load in flight is carried around the loop's back edge: what IS carried are the accumulators,
// ordinary values. (An earlier form issued the fragments at the END of the previous iteration; hipcc gave them
// registers of their own and copied those into the k-loop's at the loop header - before they had landed.
// tests/test_asm_pipeline_audit.py follows the control flow and catches that.)
// LAST: the run is the network's final layer alone (decode + global stores in its epilogue).
template <uint32_t TN, uint32_t MT, uint32_t ROWS, bool LAST>
__device__ __forceinline__ void nif_dense_layers(const NifParams& P, uint32_t l0, uint32_t l1, _Float16* X,
                                                 uint32_t rowBase, uint32_t ng, uint32_t lane, const h8* __restrict__ weights,
                                                 uint32_t row0, uint32_t total,
                                                 const uint32_t* __restrict__ idx, float* __restrict__ bgrOut, mi_trace_result* rays, bool scatter
#if MI_NIF_STAMPS
                                                 , unsigned long long (&stampSum)[8]
#endif
                                                 ) {
  constexpr bool last = LAST;
  f4v acc[TN][MT];
  // Fragment f of tile nt is 1 KiB at weights + wOffset + (nt*(kSteps+1) + f)*64 (+ lane): fragment 0 is the tile's
  // bias, fragment 1 + ks the weights of k-step ks (NifDevice::load). A wave-uniform base per tile, so the loads use
  // the SGPR-base + VGPR-offset form with ONE address VGPR (lane*16 + 1 KiB * fragment) for the TN loads of a step.
  const uint32_t laneOff = lane * 16u;
  const uint32_t ngU = (uint32_t)__builtin_amdgcn_readfirstlane((int)ng);

  // ReLU / binary16 store of layer EL (or decode + environment add), between the two barriers that separate its
  // k-loop's reads of X from these writes and these writes from the next k-loop's reads. The bias is already in the
  // accumulators (C operand of the first k-step).
  // store address of the lane's 4 features of tile nt = ng + 4a, ray tile m: nif_x_byte(rowBase + 16m + (lane&15),
  // 16nt + 4(lane>>4)) = laneStore + a * (2 * ROWS * 64) + m * 1024 - the parity of nt is the wave's (ng & 1)
  const uint32_t laneStore = nif_x_byte<ROWS>(rowBase + (lane & 15), 16 * ng + 4 * (lane >> 4));
  auto epilogue = [&](const NifLayerDesc& EL) {
    MI_STAMP(tE0);
    __syncthreads();   // every wave has read its inputs: X[.., 0..N) may be overwritten
    MI_STAMP(tE1);
    // one copy of the store loop per activation, chosen by a wave-uniform branch (as a select per value the choice
    // cost two v_cndmask per accumulator)
    auto body = [&](auto reluC) {
      constexpr bool relu = decltype(reluC)::value;
#pragma unroll
      for (uint32_t a = 0; a < TN; ++a) {
        const uint32_t nt = ng + 4 * a;
        if (16 * nt < EL.n) {
          // D fragment: rows (= output features) 16nt + 4(lane>>4) + reg, column (= ray) 16m + (lane&15)
#pragma unroll
          for (uint32_t m = 0; m < MT; ++m) {
            f4v y = acc[a][m];
            const uint32_t r = rowBase + 16 * m + (lane & 15);
            if (!last) {
              // ReLU on the rounded halves (two packed max): rounding is monotone and keeps the sign, so
              // max(round(y), 0) == round(max(y, 0))
              h4 yh = {(_Float16)y[0], (_Float16)y[1], (_Float16)y[2], (_Float16)y[3]};
              if (relu) yh = __builtin_elementwise_max(yh, (h4){(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f});
              *reinterpret_cast<h4*>(reinterpret_cast<char*>(X) + laneStore + a * (2u * ROWS * 64u) + m * 1024u) = yh;
            } else if (nt == 0 && (lane >> 4) == 0 && row0 + r < total) {
              // decode (NifModel.cpp:222-246): y*max + mean, exp for log-tonemapped models
              if (relu) {
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
              const uint32_t src = idx ? idx[row] : row;
              if (src == kNifHole) continue;          // (a hole of the escaped-slot list: no ray)
              if (bgrOut) { const size_t dst = scatter ? (size_t)src : (size_t)row; bgrOut[3 * dst] = o[0]; bgrOut[3 * dst + 1] = o[1]; bgrOut[3 * dst + 2] = o[2]; }
              if (rays) {
                mi_trace_result* res = rays + src;
                const mi_vec3 tp = res->h.throughput;
                res->rgb.x += tp.x * o[2];          // BGR -> RGB (codelets/TraceCodelets.cpp:376)
                res->rgb.y += tp.y * o[1];
                res->rgb.z += tp.z * o[0];
              }
            }
          }
        }
      }
    };
    if (EL.relu) body(std::true_type{}); else body(std::false_type{});
    MI_STAMP(tE2);
    __syncthreads();
    MI_STAMP(tE3);
    MI_STAMP_ADD(1, tE0, tE1); MI_STAMP_ADD(2, tE1, tE2); MI_STAMP_ADD(3, tE2, tE3);
  };

  for (uint32_t l = l0; l < l1; ++l) {
    const NifLayerDesc L = P.layers[l];
    const uint32_t kSteps = L.kSteps;
    // the network's final layer has 3 outputs: one real tile, padded to four so that every wave has one. The three
    // waves whose tile is padding skip its k-loop (60 MFMAs and 10 KiB of weights each) and only keep the barriers
    if (LAST && 16 * ng >= L.n) continue;
    // The k-loop is written instruction by instruction (every statement below is a volatile asm, which hipcc keeps in
    // program order; it only allocates the registers and adds the little address arithmetic that is left). Why: a
    // wave issues about one instruction per four cycles, a v_mfma_f32_16x16x32_f16 occupies the matrix pipe for 16 and
    // the issue port for 8 of them (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'), so loads, LDS reads and
    // waits are free only while they sit BETWEEN MFMAs. Stamps (tools/nif_stamps.py) showed a lone wave's k-loop at
    // 1.7x its MFMA time with the compiler's placement (loads and five scalar instructions of address arithmetic each
    // bunched in front of a k-step, a v_add per LDS read, s_nop pads behind every "landed" marker).
    //   weight stream : three fragment sets; a step asks for its OWN set's next fragments (k-step ks+3) in its last ray
    //                   tile, each load in the shadow of the MFMA that was the last to read that fragment (a
    //                   global_load_dwordx4 holds the issue port for ~16 cycles: five of them in front of a step cost 80
    //                   of its 480); "all but the newest 2*TN loads have landed" (loads return in order) is the set
    //                   about to be used - and, in step 0, the bias. A skipped tail step skips its loads too; the
    //                   waits in front of the steps are unconditional, so every set a later step could read has landed
    //                   on every path hipcc's branch structure contains.
    //   activations   : a ring of LDS reads two ray tiles ahead, across k-steps (the last two tiles of a step ask for
    //                   the first two fragments of the next); tile m of every step uses slot m % kRing, kRing divides MT.
    // Every destination is named behind its wait ("; landed vN") for tests/test_asm_pipeline_audit.py, which checks
    // that nothing touches a destination in flight and that no store / atomic / scratch access joins the counted queue.
    u4v wA[TN], wB[TN], wC[TN];
    f4v bv[TN];
    uint64_t wTile[TN];
#pragma unroll
    for (uint32_t a = 0; a < TN; ++a) {
      // all-scalar arithmetic on kernel arguments + ngU: SALU results, which a VMEM instruction may read without wait
      // states (a v_readfirstlane result would need 5: cdna_hip_programming.md §5.7 item 2)
      const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((ngU + 4 * a) * (kSteps + 1)) << 10));
      wTile[a] = (uint64_t)(uintptr_t)(weights + L.wOffset) + off;
    }
#if defined(MI_NIF_KO) && (MI_NIF_KO & 1)
    auto loadOne = [&](u4v& w, uint32_t voff, uint32_t a) { asm volatile("; no load %0, %1, %2" : "+v"(w) : "v"(voff), "s"(wTile[a])); };
#else
    auto loadOne = [&](u4v& w, uint32_t voff, uint32_t a) { asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(w) : "v"(voff), "s"(wTile[a])); };
#endif
    auto fragOff = [&](uint32_t ks) { return laneOff + ((1u + (ks < kSteps ? ks : kSteps - 1u)) << 10); };   // past the end: the last fragment again (keeps the count uniform)
    auto loadW = [&](u4v (&w)[TN], uint32_t ks) {
      const uint32_t voff = fragOff(ks);
#pragma unroll
      for (uint32_t a = 0; a < TN; ++a) loadOne(w[a], voff, a);
    };
    // bias and k-step 0 first: they are in flight while the previous layer's epilogue runs
#pragma unroll
    for (uint32_t a = 0; a < TN; ++a) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(bv[a]) : "v"(laneOff), "s"(wTile[a]));
    loadW(wA, 0);
    if (l > l0) epilogue(P.layers[l - 1]);

    constexpr uint32_t kStepBytes = ROWS * 64u;      // one k-chunk of the image
    const uint32_t xLane = (uint32_t)(uintptr_t)X + nif_x_byte<ROWS>(rowBase + (lane & 15), L.inBase + 8 * (lane >> 4));
    constexpr uint32_t kRing = (MT % 3u == 0u) ? 3u : 4u;
    static_assert(MT % kRing == 0 && MT >= 2, "the activation ring runs two tiles ahead and must divide the tiles of a step");
    u4v xb[kRing];
#if defined(MI_NIF_KO) && (MI_NIF_KO & 2)
    auto readX = [&](u4v& x, uint32_t addr, auto off) { asm volatile("; no read %0, %1" : "+v"(x) : "v"(addr)); };
#else
    auto readX = [&](u4v& x, uint32_t addr, auto off) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x) : "v"(addr), "n"(decltype(off)::value)); };
#endif
    // "all but the newest `younger` loads have landed": the set about to be used (and, the first time, the bias behind it)
    auto landedBut = [&](u4v (&w)[TN], auto younger) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(younger)::value));
#pragma unroll
      for (uint32_t a = 0; a < TN; ++a) asm volatile("; landed %0" : "+v"(w[a]));
    };
    auto landed = [&](u4v (&w)[TN]) { landedBut(w, std::integral_constant<int, 2 * TN>{}); };
    // One k-step on the fragment set w. On entry tile 0's fragment has landed and tile 1's is in flight. Per ray tile:
    //   MFMA 0 | ds_read of tile m+2 (of the next step behind the last two tiles) | MFMAs 1..TN-2 |
    //   s_waitcnt lgkmcnt(1) = tile m+1 has landed | MFMA TN-1
    // so the read and the wait issue in the shadow of an MFMA (8 of its 16 cycles leave the issue port free).
    // Behind the last step the reads fetch the k-chunk after the layer's input: inside the image, or its one chunk of
    // slack (nif_launch_mlp); the drain retires them.
    auto step = [&](u4v (&w)[TN], uint32_t ks, auto first) {
      constexpr bool isFirst = decltype(first)::value;
      const uint32_t a0 = xLane + ks * kStepBytes;
      const uint32_t voff3 = fragOff(ks + 3);
      nif_unroll([&](auto mc) {
        constexpr uint32_t m = decltype(mc)::value;
        constexpr uint32_t off = (m + 2 < MT) ? (m + 2) * 1024u : kStepBytes + (m + 2 - MT) * 1024u;
#pragma unroll
        for (uint32_t a = 0; a < TN; ++a) {
          if (a + 1 == TN && TN > 1) asm volatile("s_waitcnt lgkmcnt(1)\n\t; landed %0" : "+v"(xb[(m + 1) % kRing]));
          if (isFirst) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(acc[a][m]) : "v"(w[a]), "v"(xb[m % kRing]), "v"(bv[a]));
          else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[a][m]) : "v"(w[a]), "v"(xb[m % kRing]));
          // the step's last MFMA on fragment a frees it: its successor three k-steps on is asked for at once, in that
          // MFMA's shadow ("+v": the same registers, so no copy where a conditional step's branches meet)
          if (!isFirst && m + 1 == MT) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(w[a]) : "v"(voff3), "s"(wTile[a]));
          if (a == 0) readX(xb[(m + 2) % kRing], a0, std::integral_constant<uint32_t, off>{});
          if (TN == 1) asm volatile("s_waitcnt lgkmcnt(1)\n\t; landed %0" : "+v"(xb[(m + 1) % kRing]));
        }
      }, std::make_integer_sequence<uint32_t, MT>{});
      if (isFirst) {
        // the bias registers were SrcC of MFMAs hipcc does not know about: keep them allocated until those have read them
        asm volatile("s_nop 7");
#pragma unroll
        for (uint32_t a = 0; a < TN; ++a) asm volatile("; keep %0" ::"v"(bv[a]));
      }
    };
    MI_STAMP(tK0);
    readX(xb[0], xLane, std::integral_constant<uint32_t, 0>{});
    readX(xb[1], xLane, std::integral_constant<uint32_t, 1024>{});
    asm volatile("s_waitcnt lgkmcnt(1)\n\t; landed %0" : "+v"(xb[0]));
    loadW(wB, 1);
    // (the loads and the wait in front of a k-step are issued whether or not the step exists - past the end the last
    // fragment again - so which registers have a load in flight never depends on the path taken: hipcc chains the
    // conditional steps with flag registers, and its branch structure contains paths the source rules out. Early exits
    // instead of conditional steps make it spill ~2000 registers - 120 accumulators x 6 exits meet at the drain - and an
    // else-branch that issues a skipped step's loads makes it copy sets that are in flight where the branches meet.)
    // (step 0 runs with two sets and the bias in registers; the third set is asked for once the bias is spent)
    landedBut(wA, std::integral_constant<int, TN>{});
#pragma unroll
    for (uint32_t a = 0; a < TN; ++a) asm volatile("; landed %0" : "+v"(bv[a]));
    step(wA, 0, std::true_type{});
    loadW(wC, 2);
    loadW(wA, 3);
    // from here on every step asks for its own set's next fragments (k-step ks + 3) behind its last ray tile's MFMAs
    landed(wB); if (1 < kSteps) step(wB, 1, std::false_type{});
    landed(wC); if (2 < kSteps) step(wC, 2, std::false_type{});
    for (uint32_t ks = 3; ks < kSteps; ks += 3) {
      landed(wA); step(wA, ks, std::false_type{});
      landed(wB); if (ks + 1 < kSteps) step(wB, ks + 1, std::false_type{});
      landed(wC); if (ks + 2 < kSteps) step(wC, ks + 2, std::false_type{});
    }
    // drain: the weight sets and activation fragments still in flight own their registers until they land; the pads
    // are the wait states between the last MFMAs and the first VALU read of an accumulator (the hazard recogniser does
    // not see MFMAs inside asm statements)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15");
#pragma unroll
    for (uint32_t a = 0; a < TN; ++a) { asm volatile("; landed %0" : "+v"(wA[a])); asm volatile("; landed %0" : "+v"(wB[a])); asm volatile("; landed %0" : "+v"(wC[a])); }
#pragma unroll
    for (uint32_t q = 0; q < kRing; ++q) asm volatile("; landed %0" : "+v"(xb[q]));
#pragma unroll
    for (uint32_t a = 0; a < TN; ++a)
#pragma unroll
      for (uint32_t m = 0; m < MT; ++m) asm volatile("; result %0" : "+v"(acc[a][m]));
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
__global__ void __launch_bounds__(256 * RG, (MT == 6 && RG == 1) ? 2 : 1) nif_mlp_kernel(NifParams P, const h8* __restrict__ weights,
                                                      const float* __restrict__ u, const float* __restrict__ v,
                                                      const uint32_t* __restrict__ idx, const uint32_t* __restrict__ countPtr,
                                                      uint32_t numRows, float* __restrict__ bgrOut, mi_trace_result* rays, bool scatter) {
  constexpr uint32_t kNifRows = 16u * MT * RG;                    // rays per workgroup pass
  extern __shared__ __attribute__((aligned(16))) _Float16 X[];   // nif_x_byte<kNifRows>: [P.stride / 32][kNifRows][32] (+ one chunk of slack)
  __shared__ float uvS[2 * kNifRows];                            // the pass's environment coordinates, fetched once
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t ng = wave & 3u, rowBase = 16u * MT * (wave >> 2);   // output-feature group, row group
  const uint32_t total = countPtr ? *countPtr : numRows;
  char* const Xb = reinterpret_cast<char*>(X);
  const uint32_t E = P.embedDim, F = 4 * E;
#if MI_NIF_STAMPS
  unsigned long long stampSum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long tBegin = nif_now();
#endif

  // The K-padding columns behind the features are zero and nothing ever writes them again: once per workgroup, not
  // once per pass. Items are numbered column-major (e = column * kNifRows + row) so the row/column split divides by a
  // compile-time constant.
  {
    const uint32_t padCols = P.stride - P.featBase - F;
    for (uint32_t e = tid; e < kNifRows * padCols; e += blockDim.x) {
      const uint32_t r = e % kNifRows, c = e / kNifRows;
      *reinterpret_cast<_Float16*>(Xb + nif_x_byte<kNifRows>(r, P.featBase + F + c)) = (_Float16)0.f;
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
      if (row < total) { const uint32_t src = idx ? idx[row] : row; if (src != kNifHole) { cu = u[src]; cv = v[src]; } }
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
      *reinterpret_cast<_Float16*>(Xb + nif_x_byte<kNifRows>(r, P.featBase + q)) = sn;
      *reinterpret_cast<_Float16*>(Xb + nif_x_byte<kNifRows>(r, P.featBase + 2 * E + q)) = cs;
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
#define MI_NIF_RUN(TN) do { if (finalLayer) nif_dense_layers<TN, MT, kNifRows, true>(P, l, l1, X, rowBase, ng, lane, weights, row0, total, idx, bgrOut, rays, scatter MI_NIF_STAMP_ARG); \
                            else nif_dense_layers<TN, MT, kNifRows, false>(P, l, l1, X, rowBase, ng, lane, weights, row0, total, idx, bgrOut, rays, scatter MI_NIF_STAMP_ARG); } while (0)
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

__device__ __forceinline__ uint32_t nif_uniform(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

inline bool nif_asm_covers(const NifRegsDevice& nr);                                                               // nif_asm_kernel.hpp
inline void nif_asm_launch(const NifRegsDevice& nr, const float* u, const float* v, const uint32_t* idx, const uint32_t* countPtr,
                           uint32_t numRows, float* bgrOut, mi_trace_result* rays, hipStream_t stream, bool scatter, uint32_t numCUs, unsigned long long* clockOut, uint32_t which);
inline void nif_regs_launch(const NifRegsDevice& nr, const float* u, const float* v, const uint32_t* idx, const uint32_t* countPtr,
                            uint32_t numRows, float* bgrOut, mi_trace_result* rays, hipStream_t stream, bool scatter, uint32_t numCUs, uint32_t variant);      // nif_regs_kernel.hpp

// bgrOut: result of row r at bgrOut[3r..] - or, with `scatter`, at bgrOut[3*idx[r]..] (the row's own slot)
// shape (scene option "nif_shape"): 0 = w6 (default), 1 = t6, 2 = t4 (nif_mlp_kernel's workgroup shapes); 4 = r8, 5 = r8s = K3r, the
// register-resident kernel of nif_regs_kernel.hpp (eight waves in lock-step / waves 4-7 staggered by a quarter chunk), for the
// network shapes it covers - measured 6 % (r8) and 15 % (r8s) SLOWER than w6 (profiles/r04_k3r_attempt.txt), so never the default
// and compiled into the variants build only; a network it does not cover runs w6
inline void nif_launch_mlp(const NifDevice& nif, const float* u, const float* v, const uint32_t* idx, const uint32_t* countPtr,
                           uint32_t numRows, float* bgrOut, mi_trace_result* rays, hipStream_t stream, bool scatter = false, uint32_t shape = 0, uint32_t numCUs = 256,
                           uint32_t generations = kNifGenerations) {
  if (numRows == 0) return;
  // shape 7 ("auto", the default) = K3a, the hand-scheduled register-resident kernel (nif_asm_kernel.hpp), for the network shape its
  // body was generated for (the reference's 6 x 320 with the features re-concatenated in the middle), nif_mlp_kernel's w6 for every
  // other network; 6 ("a8") asks for K3a by name (and still falls back when the network is not covered)
  // 8 ("b4") = K3b: the same dataflow with four waves of 64 rays, one per SIMD (half the LDS bytes per MFMA)
  if ((shape == 6 || shape == 7 || shape == 8) && nif_asm_covers(nif.regs)) { nif_asm_launch(nif.regs, u, v, idx, countPtr, numRows, bgrOut, rays, stream, scatter, numCUs, nif.d_clock, shape == 8 ? 4u : 2u); return; }
#if MI_RAYLIB_VARIANTS
  if (shape >= 4 && shape <= 5 && nif.regs.ok) { nif_regs_launch(nif.regs, u, v, idx, countPtr, numRows, bgrOut, rays, stream, scatter, numCUs, shape == 5 ? 0u : 1u); return; }
#endif
  if (shape >= 3) shape = 0;
  uint32_t maxTiles = 1;
  for (uint32_t l = 0; l < nif.p.numLayers; ++l) maxTiles = std::max(maxTiles, nif.p.layers[l].nTiles / 4);
  // shape: MT ray tiles per wave x RG row groups. Scene option "nif_shape" = w6|t6|t4 (`shape` 0|1|2) overrides the default (w6).
  // (4 waves x 64 / 128 / 192 rays were measured too - DESIGN.md §6 - and are not kept: at their register
  // pressure hipcc spills around the hand-counted asm loads, which the .s audit in tests/ flags.)
  uint32_t mt = 6, rg = 1;
  if (shape != 0) { rg = 2; mt = (shape == 1) ? 6 : 4; }
  // the image plus one k-chunk of slack: the activation ring of a layer's last k-step reads one chunk past its input
  auto imageBytes = [&](uint32_t rows) { return (size_t)rows * (nif.p.stride + 32u) * sizeof(_Float16); };
  if (!(mt == 4 && rg == 2) && (imageBytes(16 * mt * rg) > kNifMaxLdsBytes || maxTiles > 5)) { mt = 4; rg = 2; }
  auto launch = [&](auto kern, uint32_t rowsPerPass, uint32_t threads) {
    size_t lds = imageBytes(rowsPerPass);
#if MI_NIF_STAMPS
    if (getenv("MI_NIF_DIAG_ONE_WG")) lds = kNifMaxLdsBytes;   // diagnostic: one workgroup per CU, i.e. one wave per SIMD for the 4-wave shape
#endif
    // grid-stride: kNifGenerations times the workgroups that stay resident (two per compute unit for the 4-wave shape - the LDS
    // image sets that -, one for the 8-wave shapes). A workgroup is the unit the hardware balances: with 4 generations the 1440^2
    // launch's workgroups ran 10 or 11 passes of 96 rays each and the last quarter of the chip's slots idled through the last
    // pass; with 64 a workgroup runs one or two passes, a slot that comes free takes the next, and beside a trace launch
    // (NIF renders) the two kernels' workgroups interleave finely: K3 alone -1.3 % (1.837 -> 1.813 ms), config 5 -4.5 %
    // (profiles/r04_nif_generations_ab.txt). Scene option "nif_generations" overrides the constant for such measurements.
    uint32_t blocks = (numRows + rowsPerPass - 1) / rowsPerPass;
    const uint32_t cap = numCUs * (threads == 256 ? 2u : 1u) * generations;
    if (blocks > cap) blocks = cap;
    if (const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kNifMaxLdsBytes); e != hipSuccess)
      throw std::runtime_error(std::string("NIF: hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, stream, nif.p, nif.d_weights, u, v, idx, countPtr, numRows, bgrOut, rays, scatter);
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
    // Four samples' slots are fetched before the first is added (ten dwords each, the throughput and the environment value whether
    // the path escaped or not: the slots exist either way): the adds stay in sample order, the loads no longer wait for one another
    // through the escape test (one thread per pixel walks 128 slots; with a load, a compare and then two more loads per sample the
    // pass ran at 1.5 TB/s).
    struct P3 { float x, y, z; };
    for (uint32_t s = s0; s < s1; s += 4u) {
      P3 c[4], t[4], e[4];
      float uu[4];
#pragma unroll
      for (uint32_t k = 0; k < 4u; ++k) {
        const size_t q = (size_t)min(s + k, s1 - 1u) * n + i;
        uu[k] = u[q];
        c[k] = reinterpret_cast<const P3*>(color)[q]; t[k] = reinterpret_cast<const P3*>(tp)[q]; e[k] = reinterpret_cast<const P3*>(bgr)[q];
      }
#pragma unroll
      for (uint32_t k = 0; k < 4u; ++k) {
        if (s + k >= s1) break;
        rgb.x += c[k].x; rgb.y += c[k].y; rgb.z += c[k].z;
        // u == -1 is the "did not escape" mark of the trace kernel; an escaped ray's u is theta / pi in [0, 1] - or NaN when
        // the direction's y rounded to just outside [-1, 1] (acosf), and such a ray still takes its environment term, as
        // in the reference's per-sample form (found by tests/fuzz_parity.py nif, seed 20261005 case 8486)
        if (uu[k] != -1.f) {
          rgb.x += t[k].x * e[k].z;
          rgb.y += t[k].y * e[k].y;
          rgb.z += t[k].z * e[k].x;
        }
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
inline void nif_infer(NifDevice& nif, const float* d_u, const float* d_v, float* d_bgr, size_t n, size_t maxBatch, hipStream_t stream, uint32_t shape = 0, uint32_t numCUs = 256,
                      uint32_t generations = kNifGenerations) {
  const size_t chunk = maxBatch ? maxBatch : n;
  for (size_t off = 0; off < n; off += chunk) {
    const size_t cnt = (n - off < chunk) ? (n - off) : chunk;
    nif_launch_mlp(nif, d_u + off, d_v + off, nullptr, nullptr, (uint32_t)cnt, d_bgr + 3 * off, nullptr, stream, false, shape, numCUs, generations);
  }
}

// One sample's environment pass over the whole ray stream (src/IpuScene.cpp:571-583)
inline void nif_env_pass(NifDevice& nif, mi_trace_result* d_rays, uint32_t n, float azimuthRadians, float* d_u, float* d_v,
                         float* d_bgr, size_t /*maxBatch*/, hipStream_t stream, uint32_t shape = 0, uint32_t numCUs = 256, uint32_t generations = kNifGenerations) {
  nif.ensureIndex(n);
  (void)hipMemsetAsync(nif.d_count, 0, sizeof(uint32_t), stream);
  hipLaunchKernelGGL(escaped_uv_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_rays, n, azimuthRadians, d_u, d_v, nif.d_index, nif.d_count);
  nif_launch_mlp(nif, d_u, d_v, nif.d_index, nif.d_count, n, nullptr, d_rays, stream, false, shape, numCUs, generations);
}

}  // namespace mi
