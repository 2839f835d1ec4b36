// nif_regs_kernel.hpp — K3r: the NIF MLP with the ACTIVATIONS IN REGISTERS and the WEIGHTS STREAMED THROUGH LDS.
// Round 4's attempt at the MLP kernel; compiled into the VARIANTS build only (libmi_raylib_variants.so, scene option nif_shape
// = r8 | r8s): it reproduces the oracle in every test and runs 6 % slower than nif_mlp_kernel on fast boxes (1.6 % faster on a
// clock-limited one) - the builds, PMC rows and knock-outs are in profiles/r04_k3r_attempt.txt, the reading in DESIGN.md §13.
//
// Same mathematics as nif_mlp_kernel (nif_kernels.hpp; reference: src/neural_networks/NifModel.cpp:186-246 encode /
// decode, :300-327 dense stack; numerics of the reference's fp16 model: binary16 features, weights and inter-layer
// activations, binary32 accumulation on v_mfma_f32_16x16x32_f16), another dataflow:
//
//   * nif_mlp_kernel keeps a 96-ray activation image in LDS and streams every weight fragment from L2 once per 96 rays:
//     1.09 MB per 96 rays = 23.5 GB per 1440^2-ray launch = 11 TB/s out of L2, which is what holds its clock at 1.7 GHz
//     (DESIGN.md §6 K3), and every layer ends in a convert + LDS store + two workgroup barriers.
//   * here a WAVE owns 32 rays for the whole network and keeps their activations in its registers. The layers are
//     evaluated transposed, Y^T = W^T X^T: the MFMA result of an output-feature tile is, per lane, 4 consecutive
//     features of ONE ray - and two such tiles side by side are exactly the 8 k-values per lane that the B operand of
//     the NEXT layer's k-step wants, if that layer's weights are packed with the matching k order (the order of the
//     terms of a dot product is free; NifRegsDevice::pack does it on the host). So a layer's output goes from the
//     accumulators to the next layer's operand registers through a convert, a ReLU and nothing else: no activation
//     ever touches LDS, no barrier separates the layers.
//   * what goes through LDS instead is the weight stream, ONCE per workgroup pass of 256 rays (8 waves x 32): 1.09 MB
//     per 256 rays = 8.8 GB per launch, 0.38 x the L2 traffic. A layer's fragments are one straight run in consumption
//     order, cut into chunks of 4 HT fragments of 1 KiB (40 for width 320; a hidden layer is a whole number of chunks),
//     brought in by LDS-DMA (global_load_lds_dwordx4: no staging registers) into a ring of three slots, two chunks ahead
//     of the MFMAs, the pieces issued one at a time between MFMAs; one workgroup barrier per chunk (80 MFMAs per wave).
//     Every A fragment a wave reads from LDS (ds_read_b128, conflict-free: 64 lanes x 16 contiguous bytes) feeds two
//     MFMAs; eight waves at full matrix rate read 128 B/clk/CU, half of the LDS's 256 (MI355X_MICROARCH.md, LDS table).
//
// A workgroup is 8 waves = 2 per SIMD, 1 workgroup per CU (ring 120 KiB + a 32-KiB image of the pass's Fourier features as B
// fragments, one 4-KiB part per wave, + the biases). Registers per lane: input 80 + output 80 (10 k-steps x 2 ray tiles x 4)
// + accumulators 16 + a four-deep fragment ring 16 + bias 8 - hipcc wants ~30 more than the 256 there are and spills a few
// finished output values per layer; what matters is that NO scratch access sits between LDS-DMA issues (see below).
//
// Shapes: hidden width H a multiple of 32 up to 320 (instantiated for H = 64, 128, 256, 320), every hidden layer H wide,
// inputs of F = 4 * embedding <= 64 Fourier features, a layer's input either the previous layer's output or that
// followed by the features (the concat of NifModel.cpp:306-309), last layer 3 outputs. Everything else runs
// nif_mlp_kernel as before.
#pragma once

#include "nif_kernels.hpp"

// Timing-only builds (tools/k3r_knockouts.sh; results are wrong): MI_NIF_REGS_KO bit 0 = no LDS-DMA, bit 1 = no wait and no
// barrier at a chunk's start, bit 2 = no fragment reads from LDS, bit 3 = no convert / ReLU epilogue.
#ifndef MI_NIF_REGS_KO
#define MI_NIF_REGS_KO 0
#endif

namespace mi {

// One 1-KiB piece of the weight stream, global -> LDS, no register in between (cdna_hip_programming.md, LDS-DMA recipe):
// lane l's 16 bytes land at ldsDst + 16 l. Counted in vmcnt like a load.
// Source = the stream's base (scalar registers, never modified) + a 32-bit byte offset per lane (piece + 16 lane): no
// 64-bit address is kept or computed; the LDS destination goes through readfirstlane, so whatever register class hipcc
// gives the ring's bookkeeping, M0 gets a scalar.
__device__ __forceinline__ void nif_glds16(const char* gbase, uint32_t byteOff, uint32_t ldsDst) {
  ldsDst = (uint32_t)__builtin_amdgcn_readfirstlane((int)ldsDst);
#if MI_NIF_REGS_KO & 1
  return;
#endif
  uint32_t keep;
  // (s_nop 2: with the two s_mov in front, the five wait states a VMEM base needs behind a v_readfirstlane that wrote it)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(byteOff), "s"(gbase), "s"(ldsDst) : "memory");
}

// Everything the hot loop needs of the ring, in scalar registers.
struct NifRegsState {
  const char* stream; const uint2* chunkTab;
  uint32_t chunksPerPass;
  uint32_t ringBase;             // LDS byte address of slot 0
  uint32_t rdSlot;               // slot (0..2) of the chunk this wave consumes next
  uint32_t wrSlot;               // slot the next chunk to be fetched goes to
  uint32_t fetchCi;              // its index in the pass's chunk table
  uint32_t fetchLeft;            // chunks of the workgroup's passes not yet fetched
  uint32_t wave, lag;            // lag: waves 4.. of the workgroup pass the chunk barrier a quarter into the chunk (STAGGER)
  // LDS-DMA pieces this wave still has to issue for the chunk being fetched
  uint32_t pendSrc, pendDst, pendF, pendN;      // pendSrc: byte offset of the chunk in the stream (the stream is < 4 GiB)
};

__device__ __forceinline__ h8 nif_lds_h8(uint32_t addr) {
#if MI_NIF_REGS_KO & 4
  h8 x; asm volatile("; no read %0, %1" : "=v"(x) : "v"(addr)); return x;
#else
  return *reinterpret_cast<const __attribute__((address_space(3))) h8*>((uintptr_t)addr);
#endif
}

// The weight ring's protocol. Barrier instance k separates "chunk k - 1 is no longer read by anybody" from "the pieces of
// chunk k + 2 may be written" and makes chunk k + 1, whose pieces every wave has waited for before arriving, visible to
// all. With STAGGER the two waves that share a SIMD reach an instance at DIFFERENT points of their own work: waves 0-3 in
// front of chunk k, waves 4-7 a quarter into chunk k (they run a quarter of a chunk ahead; chunk k became visible to them
// at instance k - 1), so that one partner's epilogues, ring restarts and barrier waits fall into the other's MFMA runs.
// The pieces of chunk k + 2 are issued one at a time between the following fragments' MFMAs (an LDS-DMA instruction holds
// the issue port for ~60 cycles, MI355X_MICROARCH.md), never in a burst.
// NOTHING in the hot loop may touch scratch: a scratch reload is a vector-memory load, counted in vmcnt IN ORDER behind
// the LDS-DMA pieces in flight - its s_waitcnt waits for every one of them to land, an L2 round trip per chunk (the
// first build of this kernel lost 17 % there).
template <uint32_t W>
__device__ __forceinline__ void nif_regs_issue_one(NifRegsState& st, uint32_t lane16) {
  if (nif_uniform(st.pendF) < nif_uniform(st.pendN)) {
    nif_glds16(st.stream, lane16 + st.pendSrc + st.pendF * 1024u, st.pendDst + st.pendF * 1024u);
    st.pendF = nif_uniform(st.pendF + W);
  }
}
template <uint32_t W, uint32_t SLOT_BYTES>
__device__ __forceinline__ void nif_regs_sync(NifRegsState& st, uint32_t lane16) {
  while (nif_uniform(st.pendF) < nif_uniform(st.pendN)) nif_regs_issue_one<W>(st, lane16);      // (normally empty by now)
#if !(MI_NIF_REGS_KO & 2)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the chunks in flight are in LDS
  __syncthreads();                                       // ... and so are everybody's; nobody reads the chunk before the current one any more
#endif
  st.pendN = 0; st.pendF = 0;
  if (nif_uniform(st.fetchLeft)) {
    const uint2 cdv = st.chunkTab[st.fetchCi];           // (uniform index: a scalar load)
    const uint2 cd = make_uint2(nif_uniform(cdv.x), nif_uniform(cdv.y));
    st.pendDst = nif_uniform(st.ringBase + st.wrSlot * SLOT_BYTES);
    st.pendSrc = cd.x * 1024u;
    st.pendN = cd.y; st.pendF = st.wave;
    st.fetchLeft = nif_uniform(st.fetchLeft - 1u);
    st.fetchCi = nif_uniform((st.fetchCi + 1u == st.chunksPerPass) ? 0u : st.fetchCi + 1u);
    st.wrSlot = nif_uniform((st.wrSlot + 1u == kRegSlots) ? 0u : st.wrSlot + 1u);
  }
}
// the lane's LDS byte address in the slot of the chunk it consumes next
template <uint32_t SLOT_BYTES>
__device__ __forceinline__ uint32_t nif_regs_next_chunk(NifRegsState& st, uint32_t lane16) {
  const uint32_t addr = nif_uniform(st.ringBase + st.rdSlot * SLOT_BYTES) + lane16;
  st.rdSlot = nif_uniform((st.rdSlot + 1u == kRegSlots) ? 0u : st.rdSlot + 1u);
  return addr;
}

// One hidden layer (or the first): `in` (+ the features, read from the wave's LDS image at featAddr) -> `out`, all in
// registers. KIND: RL_FIRST (input = the features), RL_PLAIN (input = in), RL_CONCAT (input = in followed by the features).
// The layer is ONE straight run of T = 2 KS HT fragments in consumption order - pair of output tiles, k-step, tile of the
// pair -, cut into chunks of CH = 4 HT wherever that falls, read through a ring of D fragment registers: fragment f + D is
// asked for as soon as fragment f's MFMAs have issued; hipcc counts the lgkmcnt waits.
template <uint32_t HT, uint32_t KIND, uint32_t W, uint32_t MT, uint32_t D, bool STAGGER>
__device__ __forceinline__ void nif_regs_hidden(NifRegsState& st, uint32_t lane16, const h8 (&in)[HT][MT], uint32_t featAddr, h8 (&out)[HT][MT],
                                                uint32_t biasAddr, bool relu) {
  constexpr uint32_t KS = KIND == RL_FIRST ? 2u : (KIND == RL_CONCAT ? HT + 2u : HT);
  constexpr uint32_t ACT = KIND == RL_FIRST ? 0u : HT;           // k-steps that read `in`; the others read the features
  constexpr uint32_t F2 = 2u * KS;                              // fragments per pair of output tiles
  constexpr uint32_t T = HT * F2, CH = 4u * HT, SLOT = CH * 1024u;
  constexpr uint32_t QS = STAGGER ? ((CH / 4u) & ~1u) : 0u;     // where waves 4.. pass the barrier
  static_assert(T % CH == 0, "a hidden layer is a whole number of chunks");
  auto ldsF4 = [](uint32_t a) { return *reinterpret_cast<const __attribute__((address_space(3))) f4v*>((uintptr_t)a); };
  uint32_t addr = 0;
  h8 ring[D];
  f4v acc[2][MT];
  h8 fb[MT];                                                      // the feature k-step about to be used (B operands from the LDS image)
  // the pair's bias is the C operand of its first MFMAs; the NEXT pair's is asked for right behind them
  f4v bias0 = ldsF4(biasAddr), bias1 = ldsF4(biasAddr + 64u);
  if (ACT == 0) {
#pragma unroll
    for (uint32_t m = 0; m < MT; ++m) fb[m] = nif_lds_h8(featAddr + m * 1024u);
  }
#pragma unroll
  for (uint32_t c = 0; c < T / CH; ++c)
#pragma unroll
  for (uint32_t off = 0; off < CH; ++off) {
    const uint32_t f = c * CH + off, j = f / F2, r = f % F2, ks = r >> 1, tt = r & 1u;
    if (off == 0) {
      if (!STAGGER || !nif_uniform(st.lag)) nif_regs_sync<W, SLOT>(st, lane16);
      addr = nif_regs_next_chunk<SLOT>(st, lane16);
#pragma unroll
      for (uint32_t q = 0; q < D; ++q) if (q < CH) ring[(f + q) % D] = nif_lds_h8(addr + q * 1024u);
    }
    if (STAGGER && off == QS && nif_uniform(st.lag)) nif_regs_sync<W, SLOT>(st, lane16);
    const h8 a = ring[f % D];
#pragma unroll
    for (uint32_t m = 0; m < MT; ++m) {
      const h8 b = ks < ACT ? in[ks < ACT ? ks : 0][m] : fb[m];
      acc[tt][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, ks == 0 ? (tt ? bias1 : bias0) : acc[tt][m], 0, 0, 0);
    }
    if (off + D < CH) ring[f % D] = nif_lds_h8(addr + (off + D) * 1024u);
    if (r == 1 && j + 1 < HT) { bias0 = ldsF4(biasAddr + 128u * (j + 1u)); bias1 = ldsF4(biasAddr + 128u * (j + 1u) + 64u); }
    // the feature k-step that comes next, asked for as soon as its registers are free (behind the last MFMAs that read the
    // previous one - or, for its first use in a pair, one k-step ahead)
    if (tt == 1) {
      const uint32_t nks = (ks + 1u == KS) ? 0u : ks + 1u;       // the k-step of the next fragment pair (of the next pair of tiles after the last)
      const bool more = ks + 1u < KS || j + 1u < HT;
      if (more && nks >= ACT) {
#pragma unroll
        for (uint32_t m = 0; m < MT; ++m) fb[m] = nif_lds_h8(featAddr + ((nks - ACT) * MT + m) * 1024u);
      }
    }
    if ((off & 3u) == 2u) nif_regs_issue_one<W>(st, lane16);      // one LDS-DMA piece every four fragments, behind their MFMAs
    __builtin_amdgcn_sched_barrier(0);                            // (keeps the reads D fragments ahead: hipcc otherwise sinks them to their use)
    if (r == F2 - 1u) {
      // the pair's 32 output features of this lane's rays: k-step j of the next layer's input (NifRegsDevice::actK)
#pragma unroll
      for (uint32_t m = 0; m < MT; ++m) {
        const f4v y0 = acc[0][m], y1 = acc[1][m];
#if MI_NIF_REGS_KO & 8
        h8 v0; { union { f4v f; h8 h; } cv; cv.f = y0 + y1; v0 = cv.h; } out[j][m] = v0; continue;
#endif
        h8 v = {(_Float16)y0[0], (_Float16)y0[1], (_Float16)y0[2], (_Float16)y0[3], (_Float16)y1[0], (_Float16)y1[1], (_Float16)y1[2], (_Float16)y1[3]};
        // ReLU on the rounded halves: rounding is monotone and keeps the sign, so max(round(y), 0) == round(max(y, 0))
        if (relu) v = __builtin_elementwise_max(v, (h8){0, 0, 0, 0, 0, 0, 0, 0});
        out[j][m] = v;
      }
    }
  }
}

// (the tables are read with scalar loads: kernel parameters marked __restrict__ const, so hipcc knows the kernel's own stores
// cannot change them; a value that arrives in a vector register would drag the whole ring bookkeeping into the VALU)

// The final layer: one tile of 3 real outputs; decode + store / environment add (nif_kernels.hpp epilogue).
template <uint32_t HT, uint32_t W, uint32_t MT>
__device__ __forceinline__ void nif_regs_final(NifRegsState& st, uint32_t lane16, const h8 (&src)[HT][MT], uint32_t featAddr, uint32_t biasBase,
                                               const NifRegsLayer L, const NifRegsCold& C1, uint32_t row0, uint32_t total, uint32_t wave, uint32_t lane,
                                               const uint32_t* __restrict__ idx, float* __restrict__ bgrOut, mi_trace_result* rays, uint32_t scatter) {
  constexpr uint32_t SLOT = 4u * HT * 1024u;
  const uint32_t g = lane >> 4;
  const bool concat = nif_uniform(L.kind) == RL_LAST_CONCAT;
  nif_regs_sync<W, SLOT>(st, lane16);                       // (the final layer's short chunk: both halves of the workgroup meet in front of it)
  const uint32_t addr = nif_regs_next_chunk<SLOT>(st, lane16);
  while (nif_uniform(st.pendF) < nif_uniform(st.pendN)) nif_regs_issue_one<W>(st, lane16);
  f4v acc[MT];
  const f4v b0 = *reinterpret_cast<const __attribute__((address_space(3))) f4v*>((uintptr_t)(biasBase + 4u * nif_uniform(L.biasBase) + 16u * g));
#pragma unroll
  for (uint32_t m = 0; m < MT; ++m) acc[m] = b0;
#pragma unroll
  for (uint32_t ks = 0; ks < HT; ++ks) {
    const h8 a0 = nif_lds_h8(addr + ks * 1024u);
#pragma unroll
    for (uint32_t m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, src[ks][m], acc[m], 0, 0, 0);
  }
  if (concat) {
#pragma unroll
    for (uint32_t c = 0; c < 2; ++c) {
      const h8 a0 = nif_lds_h8(addr + (HT + c) * 1024u);
#pragma unroll
      for (uint32_t m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, nif_lds_h8(featAddr + (c * MT + m) * 1024u), acc[m], 0, 0, 0);
    }
  }
#pragma unroll
  for (uint32_t m = 0; m < MT; ++m) {
    const uint32_t r = wave * (16u * MT) + 16u * m + (lane & 15u);
    if (g == 0 && row0 + r < total) {
      f4v y = acc[m];
      if (nif_uniform(L.relu)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) y[q] = y[q] > 0.f ? y[q] : 0.f;
      }
      float o[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        o[c] = y[c] * C1.maxValue + C1.mean[c];                 // decode (NifModel.cpp:222-246)
        if (C1.logTonemap) o[c] = expf(o[c]);
      }
      const uint32_t row = row0 + r;
      const uint32_t src = idx ? idx[row] : row;
      if (src == 0xFFFFFFFFu) continue;          // (kNifHole: a hole of the escaped-slot list)
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

// W waves of MT 16-ray tiles each (W * 16 * MT = 256 rays per pass): 8 waves x 2 tiles, two waves per SIMD at 256 registers.
template <uint32_t HT, uint32_t W, uint32_t MT, uint32_t D, bool STAGGER>
__global__ void __launch_bounds__(64 * W) nif_regs_kernel(NifRegsCold C0, const h8* __restrict__ streamG, const uint2* __restrict__ chunkTabG,
                                                         const NifRegsLayer* __restrict__ layerTabG, const float* __restrict__ biasG,
                                                         const float* __restrict__ uG, const float* __restrict__ vG,
                                                         const uint32_t* __restrict__ idxG, const uint32_t* __restrict__ countPtr,
                                                         uint32_t numRows, float* __restrict__ bgrOutG, mi_trace_result* raysG, uint32_t scatter) {
  constexpr uint32_t kRows = W * 16u * MT;                                        // rays per workgroup pass
  constexpr uint32_t CH = 4u * HT, SLOT = CH * 1024u;
  extern __shared__ __attribute__((aligned(1024))) unsigned char ringLds[];      // [kRegSlots][SLOT] weights, [W][2][MT][1 KiB] features, then the biases
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t lane16 = lane * 16u, g = lane >> 4;
  const uint32_t total = nif_uniform(countPtr ? *countPtr : numRows);
  const uint32_t firstRow = blockIdx.x * kRows;
  if (firstRow >= total) return;                               // (whole workgroup: nothing issued yet)
  const uint32_t ringBase = (uint32_t)(uintptr_t)ringLds;
  const uint32_t featBase = ringBase + kRegSlots * SLOT;
  const uint32_t biasBase = featBase + kRegFeatBytes;
  {
    float* const biasS = reinterpret_cast<float*>(ringLds + kRegSlots * SLOT + kRegFeatBytes);
    for (uint32_t k = tid; k < C0.biasFloats; k += blockDim.x) biasS[k] = biasG[k];
  }
  const uint32_t passes = (total - firstRow + gridDim.x * kRows - 1u) / (gridDim.x * kRows);
  NifRegsState st;
  st.stream = reinterpret_cast<const char*>(streamG); st.chunkTab = chunkTabG;
  st.chunksPerPass = C0.chunksPerPass;
  st.ringBase = ringBase;
  st.rdSlot = 0;
  st.wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
  st.lag = (STAGGER && st.wave >= W / 2u) ? 1u : 0u;
  st.pendSrc = 0; st.pendDst = 0; st.pendF = 0; st.pendN = 0;
  // prologue: the first two chunks, visible to every wave before anybody starts (with STAGGER waves 4.. read chunk 0 ahead of instance 0)
  const uint32_t chunksTotal = passes * C0.chunksPerPass;
  for (uint32_t c = 0; c < 2u && c < chunksTotal; ++c) {
    const uint2 cd = chunkTabG[c % C0.chunksPerPass];
    for (uint32_t f = st.wave; f < cd.y; f += W) nif_glds16(st.stream, lane16 + (cd.x + f) * 1024u, ringBase + c * SLOT + f * 1024u);
  }
  st.fetchLeft = chunksTotal > 2u ? chunksTotal - 2u : 0u;
  st.fetchCi = 2u % C0.chunksPerPass;
  st.wrSlot = 2u;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const uint32_t featAddr = featBase + st.wave * (2u * MT * 1024u) + lane16;
  for (uint32_t row0 = firstRow; row0 < total; row0 += gridDim.x * kRows) {
    // ---- this lane's Fourier features of its rays, as B fragments in the wave's LDS image: chunk c, ray tile m, lane: the 8
    // halves feature 32 c + 8 g + e of [sin u | sin v | cos u | cos v] (NifModel.cpp:203-216), zero beyond F. Only this wave
    // reads them (in order behind these stores: no barrier). ----
    {
      const uint32_t E = C0.embedDim, F = 4u * E;
#pragma unroll
      for (uint32_t m = 0; m < MT; ++m) {
        const uint32_t row = row0 + wave * (16u * MT) + 16u * m + (lane & 15u);
        float cu = 0.f, cv = 0.f;
        if (row < total) { const uint32_t src = idxG ? idxG[row] : row; if (src != 0xFFFFFFFFu) { cu = uG[src]; cv = vG[src]; } }
#pragma unroll
        for (uint32_t c = 0; c < 2; ++c) {
          h8 fv;
#pragma unroll
          for (uint32_t e = 0; e < 8; ++e) {
            const uint32_t f = 32u * c + 8u * g + e;
            const bool isCos = f >= 2u * E;
            const uint32_t q = isCos ? f - 2u * E : f;
            const bool isV = q >= E;
            const uint32_t j = isV ? q - E : q;
            const float nrm = ((isV ? cv : cu) - 1.f) * 2.f;                          // NifModel.cpp:203-205
            const float phase = (float)(_Float16)(nrm * (float)(1u << (j & 15u)));     // cast to HALF before sin/cos (:212)
            float fs, fc;
            sincos_half_phase(phase, fs, fc);
            fv[e] = f < F ? (_Float16)(isCos ? fc : fs) : (_Float16)0.f;
          }
          *reinterpret_cast<__attribute__((address_space(3))) h8*>((uintptr_t)(featAddr + (c * MT + m) * 1024u)) = fv;
        }
      }
    }

    // The layers' activations alternate between two register arrays: layer 0 (features -> xa), layer 1 (xa -> xb), layer 2
    // (xb -> xa), ...: no copy between layers, each kind of hidden layer is compiled once per direction.
    h8 xa[HT][MT], xb[HT][MT];
#pragma unroll
    for (uint32_t k = 0; k < HT; ++k)
#pragma unroll
      for (uint32_t m = 0; m < MT; ++m) xb[k][m] = (h8){0, 0, 0, 0, 0, 0, 0, 0};
    {
      const NifRegsLayer L0 = layerTabG[0];
      nif_regs_hidden<HT, RL_FIRST, W, MT, D, STAGGER>(st, lane16, xb, featAddr, xa, biasBase + 4u * nif_uniform(L0.biasBase) + 16u * g, nif_uniform(L0.relu) != 0u);
    }
    // (spelled out, not lambdas: everything of the ring's state must stay in scalar registers, and a closure that takes it by
    // reference made hipcc treat it as memory)
#define MI_NIF_HIDDEN(l, src, dst) do { \
      const NifRegsLayer L = layerTabG[l]; \
      const uint32_t biasAddr = biasBase + 4u * nif_uniform(L.biasBase) + 16u * g; \
      if (nif_uniform(L.kind) == RL_PLAIN) nif_regs_hidden<HT, RL_PLAIN, W, MT, D, STAGGER>(st, lane16, src, featAddr, dst, biasAddr, nif_uniform(L.relu) != 0u); \
      else nif_regs_hidden<HT, RL_CONCAT, W, MT, D, STAGGER>(st, lane16, src, featAddr, dst, biasAddr, nif_uniform(L.relu) != 0u); \
    } while (0)
#define MI_NIF_FINAL(src) nif_regs_final<HT, W, MT>(st, lane16, src, featAddr, biasBase, layerTabG[C0.numLayers - 1u], C0, row0, total, wave, lane, idxG, bgrOutG, raysG, scatter)
    {
      uint32_t l = 1;
      bool inA = true;                                        // which array holds the layer's input
      for (; l + 1u < C0.numLayers; ++l) {
        MI_NIF_HIDDEN(l, xa, xb);
#pragma unroll
        for (uint32_t k = 0; k < HT; ++k)
#pragma unroll
          for (uint32_t m = 0; m < MT; ++m) xa[k][m] = xb[k][m];
      }
      (void)inA;
      MI_NIF_FINAL(xa);
    }
#undef MI_NIF_HIDDEN
#undef MI_NIF_FINAL
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA may outlive the workgroup's LDS
}

// variant (scene option "nif_shape"): 0 = r8s (waves 4-7 staggered by a quarter chunk), 1 = r8 (all eight waves in lock-step: the faster of the two)
inline void nif_regs_launch(const NifRegsDevice& nr, const float* u, const float* v, const uint32_t* idx, const uint32_t* countPtr,
                            uint32_t numRows, float* bgrOut, mi_trace_result* rays, hipStream_t stream, bool scatter, uint32_t numCUs, uint32_t variant) {
  if (numRows == 0) return;
  const size_t lds = nr.ldsBytes();
  auto launch = [&](auto kern) {
    uint32_t blocks = (numRows + kRegRows - 1) / kRegRows;
    if (blocks > numCUs) blocks = numCUs;             // one workgroup per compute unit (the ring takes most of its LDS), grid-stride over passes
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRegMaxLdsBytes);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * 8), lds, stream, nr.cold, nr.d_stream, nr.d_chunks, nr.d_layers, nr.d_bias, u, v, idx, countPtr, numRows, bgrOut, rays, scatter ? 1u : 0u);
  };
#define MI_NIF_REGS_GO(HT) do { if (variant == 1) launch(nif_regs_kernel<HT, 8, 2, 4, false>); else launch(nif_regs_kernel<HT, 8, 2, 4, true>); } while (0)
#ifdef MI_NIF_REGS_ONLY10      // (tools/isa_k3r.sh: one instantiation, for a quick look at its code)
  MI_NIF_REGS_GO(10);
#else
  switch (nr.ht) {
    case 2: MI_NIF_REGS_GO(2); break;
    case 4: MI_NIF_REGS_GO(4); break;
    case 8: MI_NIF_REGS_GO(8); break;
    default: MI_NIF_REGS_GO(10); break;
  }
#endif
#undef MI_NIF_REGS_GO
}

}  // namespace mi
