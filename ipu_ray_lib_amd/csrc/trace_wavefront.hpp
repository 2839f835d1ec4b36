// trace_wavefront.hpp — K1w: the path-trace hot loop as a persistent, phase-scheduled kernel.
//
// Why (profiles/r01_v0_*: the one-thread-one-pixel nested-loop kernel keeps 21 % of the VALU lanes
// busy): a path tracer's lanes diverge three ways — traversal length, leaf tests, path length. Here
// every lane is a small state machine that is always in exactly one PHASE
//     NODE  one box test of the stackless BVH walk          (CompactBvh.hpp:103-135, CompactBVH2Node.cpp:5-22)
//     LEAF  one primitive test at a leaf whose box was hit   (Mesh.cpp:6-104, Primitives.cpp:24-67)
//     SHADE traversal finished: hit update, BxDF, roulette   (codelets/TraceCodelets.cpp:214-257)
//     GEN   next sample's camera ray                          (codelets/TraceCodelets.cpp:142-164)
//     FETCH work unit finished: write it back, take another one from the global work counter
// and each loop iteration the wave votes (ballot + popcount, scalar) and runs the phase most of its
// lanes are waiting in (SHADE and GEN are served by ONE turn: template parameter MERGE). A lane that
// finishes a path starts its next sample in the turn that ended the path, a lane that
// finishes a work unit - a 4- to 64-sample segment of a pixel (ray_math.h) - pulls a new one, so no lane
// idles until the frame runs out of work; units are small on purpose, the drain at the end of a frame is paid
// per unit. The kernel is bound by instruction issue over half-empty waves and by the texture addresser that serves the
// node gathers, so the scheduling below is about few, short chains and few instructions: several
// box tests per vote, whole 8x8 tiles per wave, one copy of the cast set-up, cold state in LDS so that six waves fit a
// SIMD without a spill. Each lane
// still performs exactly the reference's sequence of operations for its rays, in the reference's
// order, with the same arithmetic — only the interleaving between lanes changes — so results stay
// bit-identical to the nested-loop kernel and to the CPU oracle.
#pragma once

#include "trace_kernels.hpp"

namespace mi {

enum : uint32_t { PH_NODE = 0, PH_LEAF = 1, PH_SHADE = 2, PH_GEN = 3, PH_FETCH = 4, PH_DONE = 5 };

// Scheduling weights (quarter units, NODE/TRAVERSE weigh 4) and traversal-burst limits; the defaults
// {8, 16, 24, 48, 3} (dbl 4, maxExtra 6) are the measured optimum over the box and the Collada scene (+-1 % plateau):
//   leafAt   inside a traversal burst, LEAF runs when cL*leafAt > cN*4
//   shadeAt, genAt   top-level vote: SHADE/GEN run when their weighted population exceeds (cN+cL)*4
//   burst    at most this many NODE/LEAF steps before the wave re-votes
//   keep8    ... or earlier, once fewer than keep8/8 of the lanes that started the burst still traverse
//   dbl, maxExtra   a NODE turn runs 1 + min(lanes / dbl, maxExtra) box tests before the wave votes again
//   leafThenNode    a box test follows every LEAF turn at once
//   prio     1: waves run their traversal turns at s_setprio 1 (short dependent steps win VALU arbitration over
//            another wave's long SHADE/GEN blocks: +1 %), 0: no priorities
//   leafP    (SPEC build) a LEAF turn also runs once this many lanes hold a pending primitive test
//   probe    (runtime-weights build only; results unchanged) knock-ins that price a resource: bit 0 = every box test issues one
//            more 16-byte load of its node (vector-memory path), bit 1 = eight more independent VALU instructions per box test
struct WaveTune { uint32_t leafAt, shadeAt, genAt, burst, keep8, dbl = 4, maxExtra = 6, leafThenNode = 1, prio = 1, leafP = 40, probe = 0;
  bool operator==(const WaveTune& o) const { return leafAt == o.leafAt && shadeAt == o.shadeAt && genAt == o.genAt && burst == o.burst && keep8 == o.keep8 && dbl == o.dbl && maxExtra == o.maxExtra && leafThenNode == o.leafThenNode && prio == o.prio && leafP == o.leafP && probe == o.probe; } };

// Per-launch extras for renders with the NIF environment. The reference traces ONE sample, evaluates the
// environment for the rays that escaped, adds it, and repeats (src/IpuScene.cpp:571-583). One sample per launch
// leaves the GPU waiting for the longest path of every wave (0.6 ms per launch on one MI355X whatever the pixel
// count), so sampleCount samples are traced per launch and each path leaves a SLOT q = sample * n + pixel:
// its radiance `color`, and - if it escaped - throughput and environment coordinates (PreProcessEscapedRays,
// codelets/TraceCodelets.cpp:321-358; u = -1 marks "did not escape") plus q appended to the compacted list the
// MLP consumes. A per-pixel pass then replays the reference's order exactly: for each sample rgb += color, then
// rgb += throughput * env, segment by segment (nif_accumulate_kernel). A launch holds whole segments, each its own
// work atom and RNG stream, so no generator state is carried from launch to launch.
// All-null extras = the plain multi-sample launch (rgb accumulated in registers).
struct WaveExtras {
  uint32_t sampleCount = 0;      // samples of this launch (whole segments, except at the end of the render); 0 = scene's samplesPerPixel
  float* u = nullptr; float* v = nullptr;          // [sampleCount][n]
  float* slotColor = nullptr; float* slotTp = nullptr;   // [sampleCount][n][3]
  uint32_t* index = nullptr; uint32_t* count = nullptr;
  float azimuthRotation = 0.f;
  uint32_t fetchChunk = 0;       // work indices taken per global atomic (multiple of 64; 0 = 64)
  // Segmented pixels (ray_math.h segment_samples): the work atom is (pixel, segment), work index = segment * n + i;
  // every atom leaves its partial rgb sum in segPart[segment][pixel] (segment 0 starts from the incoming rgb) and
  // segment_combine_kernel adds them in segment order afterwards; the last segment writes the hit record.
  // A launch may cover only the segments [segBase, segBase + segments) of every pixel (the host cuts long renders so
  // that the partial buffer stays within its budget); segment_combine_kernel then continues the running sum.
  // NIF launches use the same atoms (segments/segBase describe the launch's samples; slots instead of partial sums):
  // every atom seeds its own segment stream, so nothing but the slots is carried from launch to launch.
  float* segPart = nullptr;      // [segments][n][3]
  uint32_t segments = 1;
  uint32_t segBase = 0;
  // The stream's pixel coordinates (TraceResult::u, ::v) as a compact array [n] of float pairs, gathered once per launch
  // (pixel_coords_kernel): an atom's FETCH then reads 8 contiguous bytes instead of pulling a 64-byte sector of the 84-byte
  // record out of HBM - sixteen times per pixel at 1000 spp, which was most of the frame's HBM traffic.
  const float2* coords = nullptr;
  uint32_t firstInSetup = 0;     // slot-mode launches: 1 = a cast's first box test runs in the turn that sets the cast up (scene option "nif_first_test") instead of in a NODE turn
};

// SPEC: a lane whose walk reaches a primitive whose box it hits does not wait for the LEAF turn: it notes the primitive
// (pend1, found at node pend1Node) and walks on as if the test were going to leave the closest hit unchanged - which
// is what the reference's walk does in that case. When the LEAF turn has run the test, either the guess was right
// (no closer hit: the box tests made meanwhile used the right hit distance and stand) or the lane goes back to node
// pend1Node + 1 with the new, closer hit and repeats the walk from there, exactly as the reference continues. A second
// primitive found while the first is still pending makes the lane wait as before. Nothing is ever skipped or reordered
// in what a path finally takes from the walk: only work that turns out to be unnecessary is added, in lanes that
// would have idled.
// SLOTS: 0 = plain launches only (rgb in registers / partial sums), 1 = NIF launches only (slots), 2 = decided at run time
// from ex.slotColor. FIXED_TUNE: the scheduling weights are the compile-time defaults (kDefaultTune) instead of the
// `tune` argument. The two default-path instantiations (<.., 0, true> and <.., 1, true>) carry neither the other mode's
// code nor the ten weights in scalar registers: no scalar spills (33 before), -2.5 % frame time.
// FAST tier, per cast: the constant terms of the box test's FMAs (-o/d) and the pad of its far side. An axis the ray runs
// parallel to (v_rcp_f32 of a zero component: infinite) would make them inf - inf = NaN, and every box would then read as
// hit (fminf / fmaxf drop a NaN): correct, but such a ray walked the whole BVH. A large finite stand-in for 1/d gives what
// the slab test means for a parallel ray - no constraint when the origin lies between the planes (-huge, +huge), a miss
// otherwise - through the same FMAs, and that axis is left out of the pad (its cancellation error is beside the point: the
// products are huge either way). The shear keeps the true reciprocal (make_shear_fast is called before this).
__device__ __forceinline__ void fast_box_setup(f3 o, f3& inv, f3& oi, float& slabPad) {
  const float big = 1e18f;
  const bool px = !(fabsf(inv.x) < big), py = !(fabsf(inv.y) < big), pz = !(fabsf(inv.z) < big);
  inv = mk(px ? copysignf(big, inv.x) : inv.x, py ? copysignf(big, inv.y) : inv.y, pz ? copysignf(big, inv.z) : inv.z);
  oi = mk(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
  slabPad = 4.8e-7f * fmaxf(fmaxf(px ? 0.f : fabsf(oi.x), py ? 0.f : fabsf(oi.y)), pz ? 0.f : fabsf(oi.z));
}

// How many lanes below this one are set in `mask` (v_mbcnt: no per-lane bit mask held in registers across the kernel).
__device__ __forceinline__ uint32_t lane_rank(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// entries of the escaped-slot list a wave reserves at a time (slot mode; the host sizes the list for the padding: raylib.hip)
constexpr uint32_t kEnvChunk = 2048;     // entries of the escaped-slot list a wave reserves per atomic (a multiple of kEnvBlock)
constexpr uint32_t kEnvBlock = 256;      // what a wave has left of its last reservation when it runs out of work: the block it stands in is padded with a slot it wrote
                                         // before, every further whole block is filled with kEnvHole - entries no MLP kernel evaluates (nif_kernels.hpp kNifHole)
constexpr uint32_t kEnvHole = 0xFFFFFFFFu;

constexpr WaveTune kDefaultTune = {8, 16, 24, 48, 3, 4, 6, 1, 1, 40, 0};      // re-swept on the round-3 kernel on two scenes (profiles/r03_kernel_ab.txt): a cheaper box test favours one more of them per vote
// DF: the reference's ALLOW_DOUBLE_FALLBACK=1 build of the triangle test (trace_kernels.hpp). FAST: the tolerance tier
// (scene option "fast"): the box test as three pairs of FMAs on (plane, 1/d, -o/d), the triangle test contracted, no
// literal NaN-exact fallback - results within a stated tolerance of the exact tier's, not bit-identical.
// BARY = false (round 5): for scenes WITHOUT vertex normals. The barycentrics of the closest hit only ever feed the interpolated
// normal (Mesh.hpp:115-120); a scene without vertex normals shades with the face normal, so the three products per primitive test,
// the three selects of the closest-hit update and three registers of the walk are dead weight there. Bit-identical by construction
// (nothing reads them); launchWavefront picks it by the scene's hasNormals.
template <bool STATS, bool LDS_NODES, int BLOCK, int WAVES_PER_SIMD = 4, bool SPEC = false, int SLOTS = 2, bool FIXED_TUNE = false, bool DF = false, bool FAST = false, bool MERGE = true,
          bool BARY = true, bool ROT = false>
__global__ void __launch_bounds__(BLOCK, (BLOCK == 256 && WAVES_PER_SIMD > 4) ? WAVES_PER_SIMD : 1) path_trace_wavefront_kernel(DeviceScene sc, mi_trace_result* rays, uint32_t n,
                                                                   uint32_t* workCounter, uint32_t ldsNodeCount, WaveTune tuneArg, uint32_t tileStreamW, WaveExtras ex) {
  const WaveTune tune = FIXED_TUNE ? kDefaultTune : tuneArg;
  const bool slots = (SLOTS == 2) ? (ex.slotColor != nullptr) : (SLOTS == 1);
  constexpr bool kFirstInSetup = SLOTS != 0 && MERGE && !SPEC && !FAST && !LDS_NODES && !DF;      // (see the cast set-up of the merged SHADE / GEN turn)
  __shared__ float sinTbl[92];
  // the materials a hit is shaded with, when the scene has few (the built-in scenes have 8, test_scene.dae 9): SHADE
  // otherwise waits for two dependent global loads, leaf record then material
  constexpr uint32_t kMatCache = 16;
  struct __attribute__((aligned(16))) CachedMaterial { mi_material m; uint32_t pad[3]; };
  __shared__ CachedMaterial matS[kMatCache];
  const bool matsInLds = sc.numMaterials <= kMatCache;
  if (matsInLds) {
    for (uint32_t k = threadIdx.x; k < sc.numMaterials * 9u; k += blockDim.x)
      reinterpret_cast<uint32_t*>(&matS[k / 9u].m)[k % 9u] = reinterpret_cast<const uint32_t*>(sc.materials)[k];
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char dynLds[];
  load_sin_table(sinTbl);
  if (LDS_NODES) {
    // stage the first ldsNodeCount nodes (preorder prefix) once per workgroup, 16 B per lane per step
    const uint4* src = reinterpret_cast<const uint4*>(sc.nodes);
    uint4* dst = reinterpret_cast<uint4*>(dynLds);
    for (uint32_t k = threadIdx.x; k < ldsNodeCount * 2; k += blockDim.x) dst[k] = src[k];
    __syncthreads();
  }

  const uint32_t lane = threadIdx.x & 63;
  const uint32_t fetchChunk = ex.fetchChunk ? ex.fetchChunk : 64u;
  uint32_t chunkNext = 0, chunkEnd = 0;          // wave-uniform: the local range of work indices not handed out yet
  const uint32_t tiledCount = tileStreamW ? (n / (8u * tileStreamW)) * (8u * tileStreamW) : 0u;
  // `node` (and the successors stored in a device node) are BYTE offsets into the node array, so a box test's load needs no
  // shift; numNodes is the array's end in the same unit.
  const uint32_t numNodes = sc.numNodes << 5;
  const uint32_t spp = ex.sampleCount ? ex.sampleCount : sc.samplesPerPixel;
  const bool segd = ex.segPart != nullptr || slots;          // (pixel, segment) work atoms
  const uint32_t segShift = segment_shift(sc.samplesPerPixel), segMask = (1u << segShift) - 1u;
  const uint32_t segs = segd ? ex.segments : 1u;
  const uint32_t items = n * segs;                 // (host checks that this fits 32 bits)

  // ---- lane state ----
  uint32_t ph = PH_FETCH;
  uint32_t sample = 0, bounce = 0, node = 0, pendLeaf = 0;
  uint32_t pend1 = 0xFFFFFFFFu, pend1Node = 0, specNodes = 0;       // SPEC: the primitive test the lane has walked past, where it was found, box tests made since (STATS)
  float prow = 0.f, pcol = 0.f;
  Rng rng; rng.s0 = rng.s1 = 0;
  f3 color = mk(0, 0, 0), tp = mk(1, 1, 1);
  f3 o = mk(0, 0, 0), d = mk(0, 0, -1), nrm = mk(0, 0, 1), inv = mk(0, 0, 0);
  f3 oi = mk(0, 0, 0);           // FAST only: -o * inv, the constant term of the box test's FMAs
  float slabPad = 0.f;           // FAST only: what the far side is widened by (below)
  Shear sh; sh.kz = 2; sh.sx = sh.sy = 0.f; sh.sz = 1.f;
  Hit hit; hit.t = kInf; hit.leaf = 0xFFFFFFFFu; hit.geomID = 0xFFFFu; hit.b0 = hit.b1 = hit.b2 = 0.f;
  uint32_t oFlags = 0;
  bool exactSlab = false;
  // Cold per-lane state lives in LDS (23 dwords per lane, [word][thread] so a wave's accesses are conflict-free): the
  // pixel's stream index (0), its (row, col) (1, 2) and its running rgb sum (3..5) are touched once per path or per
  // pixel, and holding them in VGPRs made the 96-register build spill inside the traversal loop.
  // A build for seven waves per SIMD must fit seven workgroups' LDS into a compute unit: it keeps 21 words, not 23 - the pixel's
  // (row, col) (1, 2) are fetched again from the stream's compact copy by every GEN instead (8 bytes, an L2 hit: the lane's 64
  // samples of one (pixel, segment) unit read the same address).
  constexpr bool kCoordsFromStream = WAVES_PER_SIMD >= 7;
  constexpr uint32_t kColdWords = kCoordsFromStream ? 21u : 23u;
  __shared__ uint32_t coldLds[kColdWords * BLOCK];   // + the last hit's leaf and distance (6, 7) + the path state (8..22)
  auto coldU = [&](uint32_t w) -> uint32_t& { return coldLds[((kCoordsFromStream && w > 2u) ? w - 2u : w) * BLOCK + threadIdx.x]; };
  auto coldF = [&](uint32_t w) -> float& { return reinterpret_cast<float*>(coldLds)[((kCoordsFromStream && w > 2u) ? w - 2u : w) * BLOCK + threadIdx.x]; };

  auto getPix = [&]() -> uint32_t { return coldU(0); };
  // Path state that only SHADE / GEN / FETCH touch - the RNG, radiance, throughput, normal, bounce and sample
  // counters - is loaded at the top of those phases and stored at their end, so it holds no registers while the
  // wave traverses (words 8..22).
  auto pathLoad = [&]() {
    rng.s0 = (uint64_t)coldU(8) | ((uint64_t)coldU(9) << 32); rng.s1 = (uint64_t)coldU(10) | ((uint64_t)coldU(11) << 32);
    color = mk(coldF(12), coldF(13), coldF(14)); tp = mk(coldF(15), coldF(16), coldF(17)); nrm = mk(coldF(18), coldF(19), coldF(20));
    { const uint32_t w = coldU(21); bounce = w >> 2; oFlags = w & 3u; } sample = coldU(22);     // (MI_FLAG_ERROR | MI_FLAG_ESCAPED ride in the bounce word: the host sends path lengths of 2^30 and more to K1)
  };
  auto pathStore = [&]() {
    coldU(8) = (uint32_t)rng.s0; coldU(9) = (uint32_t)(rng.s0 >> 32); coldU(10) = (uint32_t)rng.s1; coldU(11) = (uint32_t)(rng.s1 >> 32);
    coldF(12) = color.x; coldF(13) = color.y; coldF(14) = color.z; coldF(15) = tp.x; coldF(16) = tp.y; coldF(17) = tp.z;
    coldF(18) = nrm.x; coldF(19) = nrm.y; coldF(20) = nrm.z;
    coldU(21) = (bounce << 2) | oFlags; coldU(22) = sample;
  };
  CastStats cs = {0, 0};
  uint32_t casts = 0, paths = 0;       // wave-uniform: counted per turn from the turn's ballots, so they live in scalar registers
  // STATS only: per-wave phase executions and the lanes that were active in them (wave-uniform values)
  uint32_t itN = 0, itL = 0, itS = 0, itG = 0, lnN = 0, lnL = 0, lnS = 0, lnG = 0;
  uint32_t itQ = 0, lnQ = 0;      // (instrumented build) primitive-test turns that hold a sphere or disc lane, and those lanes
  unsigned long long tTrav = 0, tShade = 0, tGen = 0, tLoop0 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;   // STATS: shader cycles per phase

  // Slot mode: the escaped paths of a SHADE turn append their slots to the list the MLP consumes (ex.index, ex.count), with
  // their environment coordinates (PreProcessEscapedRays, codelets/TraceCodelets.cpp:321-358, same arithmetic as
  // escaped_uv_kernel) and throughput. The wave takes room in the list kEnvChunk entries at a time: one atomic on the ONE
  // counter every wave of the launch shares per chunk, not per turn - in an open scene nearly every SHADE turn has an escape,
  // and the waves queued behind that address (profiles/r04_nif_trace_ab.txt; the atomics execute at the memory side, ~14 ns apart:
  // with 256 entries per atomic they were still 18 ms of a 21-ms launch, hence 2 048 - profiles/r05_config5_launch_ab.txt). What a
  // wave has left of its last chunk when it runs out of work: the 256-entry block it stands in is padded with a slot it wrote before
  // (the MLP evaluates that ray once more into the same slot, which changes nothing - a row's result depends on the row alone), the
  // whole blocks behind it are marked as holes no MLP kernel evaluates (kEnvHole).
  uint32_t envNext = 0, envEnd = 0, envFill = 0;      // wave-uniform
  auto pushEscaped = [&](bool envRay, uint32_t envSlot) {
    const unsigned long long mE = __ballot(envRay);
    if (!mE) return;
    const uint32_t firstE = (uint32_t)__ffsll((long long)mE) - 1u;
    const uint32_t nE = (uint32_t)__popcll(mE), room = envEnd - envNext;
    uint32_t fresh = 0;
    if (nE > room) {
      uint32_t b = 0;
      if (lane == firstE) b = atomicAdd(ex.count, kEnvChunk);
      fresh = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(b, firstE));
    }
    if (envRay) {
      const float twoPi = (float)(2.0 * 3.14159265358979323846264338327950288);
      const float invPi = (float)(1.0 / 3.14159265358979323846264338327950288);
      const float inv2Pi = (float)(1.0 / (2.0 * 3.14159265358979323846264338327950288));
      const float theta = acosf(d.y);
      float phi = atan2f(d.z, d.x) + ex.azimuthRotation;
      if (phi < 0.f) phi += twoPi;
      else if (phi > twoPi) phi -= twoPi;
      ex.u[envSlot] = theta * invPi;
      ex.v[envSlot] = phi * inv2Pi;
      ex.slotTp[3 * (size_t)envSlot] = tp.x; ex.slotTp[3 * (size_t)envSlot + 1] = tp.y; ex.slotTp[3 * (size_t)envSlot + 2] = tp.z;
      const uint32_t r = lane_rank(mE);
      ex.index[r < room ? envNext + r : fresh + (r - room)] = envSlot;      // (the old chunk is filled to its end first)
    }
    envFill = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(envSlot, firstE));
    if (nE > room) { envNext = fresh + (nE - room); envEnd = fresh + kEnvChunk; }
    else envNext += nE;
  };

  for (;;) {
    // ---------------- FETCH: cheap, always served first ----------------
    // Work is taken from the global counter in chunks (ex.fetchChunk indices, a multiple of 64 = whole 8x8 tiles)
    // and handed out to the lanes from the wave's local range: one global atomic per chunk, not per service.
    for (;;) {
      const unsigned long long mF = __ballot(ph == PH_FETCH);
      if (!mF) break;
      if (chunkNext >= chunkEnd) {
        const uint32_t firstF = (uint32_t)__ffsll((long long)mF) - 1u;
        uint32_t base = 0;
        if (lane == firstF) base = atomicAdd(workCounter, fetchChunk);
        chunkNext = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(base, firstF));
        chunkEnd = chunkNext + fetchChunk;
      }
      const uint32_t avail = chunkEnd - chunkNext, rankF = lane_rank(mF);
      const uint32_t chunkBase = chunkNext;
      chunkNext += min((uint32_t)__popcll(mF), avail);
      if (ph == PH_FETCH && rankF < avail) {
        const uint32_t idx = chunkBase + rankF;
        if (idx < items) {
          // When the stream is made of full rows of width tileStreamW (a multiple of 8), consecutive work indices
          // walk 8x8 pixel tiles over each complete group of 8 rows (the remainder keeps stream order), so the
          // 64 pixels a wave starts with are a compact tile whose primary rays traverse alike. The map is a
          // bijection on [0, n) and any order gives the same image: every pixel owns its RNG stream.
          uint32_t seg = 0, pidx = idx;
          if (segd) { const uint32_t local = idx / n; pidx = idx - local * n; seg = ex.segBase + local; }
          uint32_t entry = pidx;
          if (tileStreamW && pidx < tiledCount) {
            const uint32_t t = pidx >> 6, within = pidx & 63u, perRow = tileStreamW >> 3;
            entry = ((t / perRow) * 8u + (within >> 3)) * tileStreamW + (t % perRow) * 8u + (within & 7u);
          }
          const mi_trace_result* res = rays + entry;
          if (ex.coords) { const float2 pc = ex.coords[entry]; prow = pc.x; pcol = pc.y; }
          else { prow = res->u; pcol = res->v; }
          coldU(0) = entry; if (!kCoordsFromStream) { coldF(1) = prow; coldF(2) = pcol; }
          if (seg == 0) { coldF(3) = res->rgb.x; coldF(4) = res->rgb.y; coldF(5) = res->rgb.z; }
          else { coldF(3) = 0.f; coldF(4) = 0.f; coldF(5) = 0.f; }           // a later segment's own partial sum
          rng_seed_pixel_segment(rng, sc.rngSeed, prow, pcol, seg);
          sample = (slots ? seg - ex.segBase : seg) << segShift;      // slots are numbered within the launch
          pathStore();                 // (GEN initialises the rest)
          ph = PH_GEN;
        } else {
          ph = PH_DONE;
        }
      }
    }

    // ---------------- vote ----------------
    uint32_t cN = (uint32_t)__popcll(__ballot(ph == PH_NODE)), cL = (uint32_t)__popcll(__ballot(ph == PH_LEAF));
    uint32_t cS = (uint32_t)__popcll(__ballot(ph == PH_SHADE)), cG = (uint32_t)__popcll(__ballot(ph == PH_GEN));
    if ((cN | cL | cS | cG) == 0) break;            // every ray DONE (FETCH lanes were just served)
    // Top-level vote: TRAVERSE (the NODE and LEAF populations together) against SHADE and GEN, by weighted
    // population (a phase cannot use more than 64 lanes). Inside TRAVERSE a two-way mini-vote (two ballots)
    // alternates box tests and primitive tests, so the expensive vote is only paid when the wave leaves traversal.
    // run: 0 = TRAVERSE, 2 = SHADE, 3 = GEN
    uint32_t run;
    {
      const uint32_t pT = cN + cL, pS = cS, pG = cG;
      const uint32_t wT = pT * 4u, wS = pS * tune.shadeAt, wG = pG * tune.genAt;
      if (MERGE) run = (pT > 0 && wT >= wS + wG) ? 0 : 2;        // (one turn serves both: the weights add; the frame is flat in them, +-0.3 %)
      else if (pT > 0 && wT >= max(wS, wG)) run = 0;
      else run = (wS >= wG) ? 2 : 3;
    }
    if (run == 0) {
      // ---------------- TRAVERSE: NODE and LEAF steps under a two-way mini-vote ----------------
      const unsigned long long tq0 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
      if (tune.prio == 1) __builtin_amdgcn_s_setprio(1); else if (tune.prio == 2) __builtin_amdgcn_s_setprio(0);
      const uint32_t startT = cN + cL;
      uint32_t steps = 0;
      // lanes only change rays in SHADE/GEN, so "some lane needs the literal box test" is a
      // per-burst fact
      bool anyExact = !FAST && __ballot(exactSlab) != 0ull;
      // One box test of a lane that is in the NODE phase; returns whether the lane is still in it afterwards, so that
      // back-to-back tests narrow the exec mask from that condition directly instead of re-reading `ph`.
      // deferTag (the spelled-out run of box tests only): the lane's phase is not updated test by test; the phase of
      // every lane that took part is derived once, behind the run, from the node value it stopped with.
      auto nodeBodyT = [&](auto exactTag, auto deferTag) -> bool {
        {
          GNode nd;
          typedef uint32_t ProbeVec __attribute__((ext_vector_type(4)));
          ProbeVec probeLoad = {0u, 0u, 0u, 0u};
          if (!FIXED_TUNE && (tune.probe & 1u)) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(probeLoad) : "v"(node), "s"(sc.nodes) : "memory");
          // (uniform base + 32-bit byte offset: the load takes the scalar-base form, one shift instead of 64-bit address math)
          if (LDS_NODES && node < (ldsNodeCount << 5)) nd = *reinterpret_cast<const GNode*>(dynLds + node);
          else nd = *reinterpret_cast<const GNode*>(reinterpret_cast<const char*>(sc.nodes) + node);
          if (STATS) cs.nodes++;
          // Box test (CompactBVH2Node.cpp:5-22, intersectRaySlab CompactBVH2Node.hpp:14-50).
          // Fast form: with finite origin and finite inverse direction no slab product can be NaN, and for
          // non-NaN values the reference's ordered compare/selects ARE min/max: swap(tmin,tmax) = (min,max),
          // "t0 = tmin > t0 ? tmin : t0" = max, "t1 = tmax < t1 ? tmax : t1" = min, in any axis order; the
          // sign of a zero never reaches the result (only t0 > t1 is used). Lanes whose ray has a zero /
          // denormal direction component or a non-finite origin (exactSlab) redo the test with the
          // reference's literal compare/select sequence below, so NaN cases stay bit-identical too.
          // The far side is scaled ONCE: x -> fl(x * kSlabScale) is monotone non-decreasing, so
          // min(fl(bx*s), fl(by*s), fl(bz*s)) == fl(min(bx, by, bz) * s) bit for bit (no NaNs on this path).
          float ax, bx, ay, by, az, bz;
          if constexpr (FAST) {
            ax = __builtin_fmaf(nd.minx, inv.x, oi.x); bx = __builtin_fmaf(nd.maxx, inv.x, oi.x);
            ay = __builtin_fmaf(nd.miny, inv.y, oi.y); by = __builtin_fmaf(nd.maxy, inv.y, oi.y);
            az = __builtin_fmaf(nd.minz, inv.z, oi.z); bz = __builtin_fmaf(nd.maxz, inv.z, oi.z);
          } else {
            ax = (nd.minx - o.x) * inv.x; bx = (nd.maxx - o.x) * inv.x;
            ay = (nd.miny - o.y) * inv.y; by = (nd.maxy - o.y) * inv.y;
            az = (nd.minz - o.z) * inv.z; bz = (nd.maxz - o.z) * inv.z;
          }
          float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.f));
          // FAST: fma(plane, 1/d, -o/d) carries an ABSOLUTE error of about eps * |o/d| (the two terms cancel), far more than
          // the eps * |t| of the exact tier's (plane - o) * (1/d) and more than the 1 + 2 gamma(3) scale covers, so a thin box
          // far from the origin could be missed falsely. The far side is therefore widened by 8 eps * max |o/d| (slabPad, per
          // cast) - inside the instruction that applies the scale, so the test stays as cheap and errs on the side of
          // visiting. (On the box scene the pad changes nothing measurable: the tier's differences from the exact one come
          // from the reference's own knife-edge self-intersections)
          float t1 = FAST ? fminf(__builtin_fmaf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)), kSlabScale, slabPad), hit.t)
                          : fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)) * kSlabScale, hit.t);
          if constexpr (decltype(exactTag)::value) {
            if (exactSlab) {
              t0 = 0.f; t1 = hit.t;
              { float tmin = ax, tmax = bx; if (tmin > tmax) { const float s = tmin; tmin = tmax; tmax = s; } tmax *= kSlabScale; t0 = tmin > t0 ? tmin : t0; t1 = tmax < t1 ? tmax : t1; }
              { float tmin = ay, tmax = by; if (tmin > tmax) { const float s = tmin; tmin = tmax; tmax = s; } tmax *= kSlabScale; t0 = tmin > t0 ? tmin : t0; t1 = tmax < t1 ? tmax : t1; }
              { float tmin = az, tmax = bz; if (tmin > tmax) { const float s = tmin; tmin = tmax; tmax = s; } tmax *= kSlabScale; t0 = tmin > t0 ? tmin : t0; t1 = tmax < t1 ? tmax : t1; }
            }
          }
          const bool boxHit = !(t0 > t1);
          if (!FIXED_TUNE && (tune.probe & 1u)) asm volatile("s_waitcnt vmcnt(0)" : "+v"(probeLoad) : : "memory");      // (the register is the load's until it has landed)
          if (!FIXED_TUNE && (tune.probe & 2u)) {
            float q0 = ax, q1 = bx, q2 = ay, q3 = by, q4 = az, q5 = bz, q6 = t0, q7 = t1;
            asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2\n\tv_mov_b32 %3, %3\n\tv_mov_b32 %4, %4\n\tv_mov_b32 %5, %5\n\tv_mov_b32 %6, %6\n\tv_mov_b32 %7, %7"
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7));
          }
          if (SPEC) {
            const bool isLeaf = node_is_leaf(nd);
            const uint32_t here = node >> 5;                       // (leaves[] is indexed by node)
            if (STATS) { if (pend1 != 0xFFFFFFFFu) { cs.nodes--; specNodes++; } }
            node = (boxHit && !isLeaf) ? node + 32u : nd.link;    // (a lane always stands at the node BEHIND a primitive it waits for; pend1Node likewise)
            if (boxHit && isLeaf) {
              if (pend1 != 0xFFFFFFFFu) { pendLeaf = here; ph = PH_LEAF; return false; }     // a second one: wait
              pend1 = here; pend1Node = node;                                                // the first one: walk on
            }
            if (node >= numNodes) {
              if (pend1 != 0xFFFFFFFFu) { pendLeaf = 0xFFFFFFFFu; ph = PH_LEAF; } else ph = PH_SHADE;   // walked to the end with a test pending: wait for it
              return false;
            }
            return true;
          }
          // One select, one compare (GNode: both successors are in the node; a leaf's hit successor carries kLeafFlag and
          // so reads as "stop"). A lane that stops at a primitive stands at node = leaf + 1 once the flag is taken off:
          // the record it waits for is leaves[node - 1], nothing is carried from this step to the LEAF turn.
          node = boxHit ? nd.hit : nd.link;
          if constexpr (decltype(deferTag)::value) {
            return node < numNodes;                              // the phase is derived once, behind the run of box tests
          } else {
            if (node & kLeafFlag) { node &= ~kLeafFlag; ph = PH_LEAF; return false; }
            if (node >= numNodes) { ph = PH_SHADE; return false; }
            return true;
          }
        }
      };
      auto nodeBody = [&]() -> bool { return nodeBodyT(std::false_type{}, std::true_type{}); };               // the common case: no lane needs the literal test
      auto nodeStep = [&]() { if (ph == PH_NODE) (void)(anyExact ? nodeBodyT(std::true_type{}, std::false_type{}) : nodeBodyT(std::false_type{}, std::false_type{})); };
      for (;;) {
        const uint32_t stay = cN;
        const uint32_t cP = SPEC ? (uint32_t)__popcll(__ballot(pend1 != 0xFFFFFFFFu)) : 0u;
        if (cN * 4u >= cL * tune.leafAt && cN > 0 && !(SPEC && cP >= tune.leafP)) {
          // NODE: one box test per lane. With many lanes in the walk two box tests run back to back before the wave
          // votes again (tune.dbl): a vote costs a ballot-popcount-branch chain whose latency the second test hides;
          // lanes that reached a leaf in the first simply sit the second out.
          if (STATS) { itN++; lnN += stay; }
          const uint32_t extra = min(stay / tune.dbl, tune.maxExtra);       // wave-uniform
          if (STATS) {
            // instrumented build: the same tests, one exec region each, counted
            nodeStep();
            for (uint32_t e = 0; e < extra; ++e) { itN++; lnN += (uint32_t)__popcll(__ballot(ph == PH_NODE)); nodeStep(); }
          } else if (anyExact) {
            // (a lane of this burst needs the literal compare/select box test: the rolled form carries it)
            nodeStep();
            for (uint32_t e = 0; e < extra; ++e) nodeStep();
          } else if (ph == PH_NODE) {
            // spelled out rather than looped: straight-line code, and each further test runs under the previous
            // one's "still walking" condition instead of re-reading `ph`
            bool go = nodeBody();
            if (extra >= 1 && go) { go = nodeBody();
              if (extra >= 2 && go) { go = nodeBody();
                if (extra >= 3 && go) { go = nodeBody();
                  if (extra >= 4 && go) { go = nodeBody();
                    if (extra >= 5 && go) { go = nodeBody();
                      if (extra >= 6 && go) { go = nodeBody();
                        if (extra >= 7 && go) (void)nodeBody();
                      }
                    }
                  }
                }
              }
            }
            if (!SPEC) {        // (the SPEC form keeps its own phase bookkeeping, test by test)
              ph = (node & kLeafFlag) ? PH_LEAF : ((node >= numNodes) ? PH_SHADE : PH_NODE);
              node &= ~kLeafFlag;
            }
          }
          steps += extra;
        } else {
          // LEAF: one primitive test per lane
          if (STATS) { itL++; lnL += SPEC ? cP : cL; }
          if (SPEC) {
            if (pend1 != 0xFFFFFFFFu) {
              if (STATS) cs.leaves++;
              const GLeaf L = sc.leaves[pend1];
              float t, b0 = 0.f, b1 = 0.f, b2 = 0.f;
              bool cand;
              const uint32_t kind = leaf_kind(L);
              if (kind == LEAF_TRI) {
                t = FAST ? intersect_triangle_fast(mk(L.f[0], L.f[1], L.f[2]), mk(L.f[3], L.f[4], L.f[5]), mk(L.f[6], L.f[7], L.f[8]), o, sh, b0, b1, b2)
                       : intersect_triangle<DF>(mk(L.f[0], L.f[1], L.f[2]), mk(L.f[3], L.f[4], L.f[5]), mk(L.f[6], L.f[7], L.f[8]), o, sh, b0, b1, b2);
                cand = t > 0.f && t < kInf;
              } else if (kind == LEAF_SPHERE) {
                t = intersect_sphere(L, o, d, 0.f);
                cand = true;
              } else {
                t = intersect_disc(L, o, d);
                cand = true;
              }
              if (cand && t > 0.f && t < hit.t) {
                // a closer hit: everything walked since is void, the walk resumes behind the primitive's node with it
                hit.t = t; hit.leaf = pend1; hit.b0 = b0; hit.b1 = b1; hit.b2 = b2;
                node = pend1Node;
                pend1 = 0xFFFFFFFFu;
                if (STATS) specNodes = 0;
                ph = (node >= numNodes) ? PH_SHADE : PH_NODE;
              } else {
                // the guess was right: the box tests made meanwhile stand
                if (STATS) { cs.nodes += specNodes; specNodes = 0; }
                pend1 = 0xFFFFFFFFu;
                if (ph == PH_LEAF) {
                  if (pendLeaf != 0xFFFFFFFFu) {             // the second primitive becomes the pending one, the walk goes on
                    pend1 = pendLeaf; pend1Node = node;
                    if (node >= numNodes) pendLeaf = 0xFFFFFFFFu; else ph = PH_NODE;
                  } else ph = PH_SHADE;                      // the walk had already reached the end
                }
              }
            }
          } else
          if (ph == PH_LEAF) {
            if (STATS) cs.leaves++;
            const uint32_t atLeaf = (node >> 5) - 1u;   // the leaf the lane stopped at (its link is the node after it)
            // (uniform base + 32-bit byte offset, as for the nodes: a 64-byte record per 32-byte node)
            GLeaf L;
            if constexpr (ROT) {
              // the block of the cast's shear axis: the same three loads (16 + 16 + 8 bytes), the vertices arrive rotated
              const GLeafBlock B = *reinterpret_cast<const GLeafBlock*>(reinterpret_cast<const char*>(sc.leavesRot) + ((node - 32u) << 2) + sh.kz * 40u);
              L.type = B.type;
#pragma unroll
              for (int q = 0; q < 9; ++q) L.f[q] = B.f[q];
            } else {
              L = *reinterpret_cast<const GLeaf*>(reinterpret_cast<const char*>(sc.leaves) + ((node - 32u) << 1));
            }
            float t, b0 = 0.f, b1 = 0.f, b2 = 0.f;
            bool cand;
            const uint32_t kind = leaf_kind(L);
            if (STATS) { const unsigned long long qm = __ballot(kind != LEAF_TRI); itQ += qm ? 1u : 0u; lnQ += (uint32_t)__popcll(qm); }
            if (kind == LEAF_TRI) {
              if constexpr (ROT) t = intersect_triangle<DF, true>(mk(L.f[0], L.f[1], L.f[2]), mk(L.f[3], L.f[4], L.f[5]), mk(L.f[6], L.f[7], L.f[8]), permute_kz(o, sh.kz), sh, b0, b1, b2);
              else
              t = FAST ? intersect_triangle_fast(mk(L.f[0], L.f[1], L.f[2]), mk(L.f[3], L.f[4], L.f[5]), mk(L.f[6], L.f[7], L.f[8]), o, sh, b0, b1, b2)
                       : intersect_triangle<DF>(mk(L.f[0], L.f[1], L.f[2]), mk(L.f[3], L.f[4], L.f[5]), mk(L.f[6], L.f[7], L.f[8]), o, sh, b0, b1, b2);
              cand = t > 0.f && t < kInf;
            } else if (kind == LEAF_SPHERE) {
              t = intersect_sphere(L, o, d, 0.f);
              cand = true;
            } else {
              t = intersect_disc(L, o, d);
              cand = true;
            }
            // (five selects on one mask: as an if-block hipcc copies the five values out, branches, and copies them back)
            const bool closer = cand & (t > 0.f) & (t < hit.t);
            hit.t = closer ? t : hit.t; hit.leaf = closer ? atLeaf : hit.leaf;
            if constexpr (BARY) { hit.b0 = closer ? b0 : hit.b0; hit.b1 = closer ? b1 : hit.b1; hit.b2 = closer ? b2 : hit.b2; }
            ph = (node >= numNodes) ? PH_SHADE : PH_NODE;
          }
          // every lane that waited for a primitive test is walking again: the next vote would pick NODE anyway,
          // so a box test follows at once (tune.leafThenNode) and the vote after it sees its outcome
          if (tune.leafThenNode) {
            if (STATS) { itN++; lnN += (uint32_t)__popcll(__ballot(ph == PH_NODE)); }
            nodeStep();
            ++steps;
          }
        }
        cN = (uint32_t)__popcll(__ballot(ph == PH_NODE));
        cL = (uint32_t)__popcll(__ballot(ph == PH_LEAF));
        if (++steps >= tune.burst || (cN + cL) * 8u < startT * tune.keep8 || (cN + cL) == 0) break;
      }
      if (tune.prio == 1) __builtin_amdgcn_s_setprio(0); else if (tune.prio == 2) __builtin_amdgcn_s_setprio(1);
      if (STATS) tTrav += __builtin_amdgcn_s_memtime() - tq0;
    } else if (MERGE) {
      // ---------------- SHADE and GEN in ONE turn (MERGE, the default since round 4) ----------------
      // A lane whose path ends with samples left draws its next camera ray in the same turn, lanes that wait in GEN (they come
      // from FETCH) join, and the cast set-up (reciprocal direction, shear, root start) runs once for both populations: no GEN
      // turns and their votes, one copy of the set-up code - the 80-VGPR build of this form has no spills, so six waves fit a
      // SIMD and, unlike with the two-turn form, pay (profiles/r04_k1w_merged_turn_ab.txt: 368 -> 348 ms, with six waves 328).
      if (STATS) { itS++; lnS += cS + cG; }
      const unsigned long long tq1 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
      bool envRay = false, pathEnd = false, genNow = false, setup = false;
      uint32_t envSlot = 0;
      const bool wasShade = ph == PH_SHADE, wasGen = ph == PH_GEN;
      if (wasShade || wasGen) pathLoad();
      if (wasShade) {
        bool terminated = false;
        if (hit.leaf != 0xFFFFFFFFu) {
          const GLeaf L = sc.leaves[hit.leaf];
          hit.geomID = leaf_geom(L);
          coldU(6) = hit.leaf; coldF(7) = hit.t;
          o = o + d * hit.t;
          nrm = hit_normal(sc, hit, o);
          // (two loads under a wave-uniform branch: as a select of the two addresses hipcc made FLAT loads of it, which go through the
          // texture addresser as well as the LDS and wait on both counters)
          mi_material mat;
          if (matsInLds) { mat = matS[L.matIndex].m; asm volatile("" ::: "memory"); } else mat = sc.materials[L.matIndex];      // (the empty asm keeps the two loads apart)
          const f3 albedo = mk(mat.albedo.x, mat.albedo.y, mat.albedo.z);
          if (mat.emissive) color = color + tp * mk(mat.emission.x, mat.emission.y, mat.emission.z);
          if (mat.type == 0) {
            const float u1 = rng_uniform01(rng);
            const float u2 = rng_uniform01(rng);
            d = sample_diffuse(nrm, u1, u2, sinTbl);
            tp = tp * albedo;
          } else if (mat.type == 1) {
            d = reflect_dir(d, nrm);
            tp = tp * albedo;
          } else if (mat.type == 2) {
            const float u1 = rng_uniform01(rng);
            f3 nd2;
            const bool refracted = dielectric(d, nrm, mat.ior, u1, nd2);
            d = nd2;
            if (refracted) tp = tp * albedo;
          } else {
            const float qn = __builtin_nanf("");
            coldF(3) = coldF(3) * qn; coldF(4) = coldF(4) * qn; coldF(5) = coldF(5) * qn;
            oFlags |= MI_FLAG_ERROR;
          }
        } else {
          coldF(7) = kInf;
          oFlags |= MI_FLAG_ESCAPED;
          terminated = true;
        }
        if (!terminated && bounce > sc.rouletteStartDepth) {
          const float u1 = rng_uniform01(rng);
          if (roulette_stop(u1, tp)) terminated = true;
        }
        bounce++;
        if (bounce >= sc.maxPathLength) terminated = true;
        if (terminated) {
          const uint32_t pixNow = getPix();
          mi_trace_result* res = rays + pixNow;
          if (slots) {
            const size_t q = (size_t)sample * n + pixNow;
            ex.slotColor[3 * q] = color.x; ex.slotColor[3 * q + 1] = color.y; ex.slotColor[3 * q + 2] = color.z;
            envRay = (oFlags & MI_FLAG_ESCAPED) != 0;
            envSlot = (uint32_t)q;
            if (!envRay) ex.u[q] = -1.f;
          } else { coldF(3) = coldF(3) + color.x; coldF(4) = coldF(4) + color.y; coldF(5) = coldF(5) + color.z; }
          pathEnd = true;
          ++sample;
          const bool more = segd ? ((sample & segMask) != 0u && sample < spp) : (sample < spp);
          if (more) genNow = true;
          else if (segd && sample < spp) {
            if (!slots && ex.segPart) {
              float* part = ex.segPart + 3 * ((size_t)(((sample - 1u) >> segShift) - ex.segBase) * n + pixNow);
              part[0] = coldF(3); part[1] = coldF(4); part[2] = coldF(5);
            }
            ph = PH_FETCH;
          } else {
            if (!slots && ex.segPart) {
              float* part = ex.segPart + 3 * ((size_t)(((sample - 1u) >> segShift) - ex.segBase) * n + pixNow);
              part[0] = coldF(3); part[1] = coldF(4); part[2] = coldF(5);
            } else
            if (!slots) res->rgb = {coldF(3), coldF(4), coldF(5)};
            uint32_t oPrim = MI_INVALID_PRIM, oGeom = MI_INVALID_GEOM;
            const uint32_t lastLeaf = coldU(6);
            if (lastLeaf != 0xFFFFFFFFu) { const GLeaf LL = sc.leaves[lastLeaf]; oPrim = LL.primID; oGeom = leaf_geom(LL); }
            mi_hit_record hr;
            hr.r.origin = {o.x, o.y, o.z}; hr.r.t_min = 0.f;
            hr.r.direction = {d.x, d.y, d.z}; hr.r.t_max = coldF(7);
            hr.prim_id = oPrim;
            hr.normal = {nrm.x, nrm.y, nrm.z};
            hr.throughput = {tp.x, tp.y, tp.z};
            hr.geom_id = (uint16_t)oGeom; hr.flags = (uint16_t)oFlags;
            res->h = hr;
            ph = PH_FETCH;
          }
        } else {
          o = offset_origin(o, d, nrm);
          setup = true;
        }
      }
      if (slots) pushEscaped(envRay, envSlot);
      if (genNow || wasGen) {
        if constexpr (kCoordsFromStream) {
          const uint32_t e = coldU(0);
          if (ex.coords) { const float2 pc = ex.coords[e]; prow = pc.x; pcol = pc.y; } else { prow = rays[e].u; pcol = rays[e].v; }
        } else { prow = coldF(1); pcol = coldF(2); }
        float g0, g1;
        rng_gauss2(rng, sinTbl, g0, g1);
        const float jr = prow + sc.antiAliasScale * g0, jc = pcol + sc.antiAliasScale * g1;
        d = pixel_to_ray_dir(jc, jr, sc.imageWidth, sc.imageHeight, sc.tanTheta);
        o = mk(0.f, 0.f, 0.f);
        nrm = mk(0.f, 0.f, 1.f);
        coldU(6) = 0xFFFFFFFFu; coldF(7) = kInf;
        oFlags = 0;
        tp = mk(1.f, 1.f, 1.f);
        color = mk(0.f, 0.f, 0.f);
        bounce = 0;
        o = offset_origin(o, d, nrm);
        setup = true;
      }
      if (setup) {
        if (FAST) { inv = fast_inverse(d); sh = make_shear_fast(d, inv); }
        else {
          inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
          exactSlab = !(fabsf(inv.x) < kInf && fabsf(inv.y) < kInf && fabsf(inv.z) < kInf && fabsf(o.x) < kInf && fabsf(o.y) < kInf && fabsf(o.z) < kInf);
          sh = make_shear(d, inv);
        }
        if (FAST) fast_box_setup(o, inv, oi, slabPad);
        hit.t = kInf; hit.leaf = 0xFFFFFFFFu;
        { uint32_t seen; node = root_start(sc, o, seen) << 5; if (STATS) cs.nodes += seen; }
        ph = (numNodes > 0) ? PH_NODE : PH_SHADE;
        // Slot-mode launches (NIF renders), option nif_first_test: the cast's FIRST box test runs here, in the turn that set the cast up,
        // instead of in a NODE turn. In the open scenes an environment light is for, most casts miss the root's box: they take a NODE
        // turn each for that one test - next to the few lanes that walk the mesh, 11 NODE turns per 64 casts at 20 % occupancy in
        // config 5 - and with the option go from set-up to SHADE without one, while the walking lanes collect until a NODE turn pays.
        // The same test on the same values as nodeBodyT below, the walk continues from its outcome: nothing a path computes changes.
        // Measured neutral (the walking lanes sit out the SHADE turns instead: 2.1 of them per 64 casts at half occupancy against 1.2):
        // off by default.
        if constexpr (kFirstInSetup) {
          if (slots && ex.firstInSetup && ph == PH_NODE) {
            const GNode nd = *reinterpret_cast<const GNode*>(reinterpret_cast<const char*>(sc.nodes) + node);
            if (STATS) cs.nodes++;
            const float ax = (nd.minx - o.x) * inv.x, bx = (nd.maxx - o.x) * inv.x;
            const float ay = (nd.miny - o.y) * inv.y, by = (nd.maxy - o.y) * inv.y;
            const float az = (nd.minz - o.z) * inv.z, bz = (nd.maxz - o.z) * inv.z;
            float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.f));
            float t1 = fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)) * kSlabScale, hit.t);
            if (exactSlab) {      // (the reference's literal compare / select sequence: NaN cases stay bit-identical)
              t0 = 0.f; t1 = hit.t;
              { float tmin = ax, tmax = bx; if (tmin > tmax) { const float w = tmin; tmin = tmax; tmax = w; } tmax *= kSlabScale; t0 = tmin > t0 ? tmin : t0; t1 = tmax < t1 ? tmax : t1; }
              { float tmin = ay, tmax = by; if (tmin > tmax) { const float w = tmin; tmin = tmax; tmax = w; } tmax *= kSlabScale; t0 = tmin > t0 ? tmin : t0; t1 = tmax < t1 ? tmax : t1; }
              { float tmin = az, tmax = bz; if (tmin > tmax) { const float w = tmin; tmin = tmax; tmax = w; } tmax *= kSlabScale; t0 = tmin > t0 ? tmin : t0; t1 = tmax < t1 ? tmax : t1; }
            }
            node = !(t0 > t1) ? nd.hit : nd.link;
            ph = (node & kLeafFlag) ? PH_LEAF : ((node >= numNodes) ? PH_SHADE : PH_NODE);
            node &= ~kLeafFlag;
          }
        }
      }
      if (wasShade || wasGen) pathStore();
      paths += (uint32_t)__popcll(__ballot(pathEnd));
      casts += (uint32_t)__popcll(__ballot(setup));
      if (STATS) tShade += __builtin_amdgcn_s_memtime() - tq1;
    } else if (run == 2) {
      // ---------------- SHADE: traversal of bounce `bounce` is complete ----------------
      if (STATS) { itS++; lnS += cS; }
      const unsigned long long tq1 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
      bool envRay = false;          // this lane's path ended in this SHADE step by escaping (slot mode)
      bool pathEnd = false;
      uint32_t envSlot = 0;
      if (ph == PH_SHADE) {
        pathLoad();
        bool terminated = false;
        if (hit.leaf != 0xFFFFFFFFu) {
          const GLeaf L = sc.leaves[hit.leaf];
          hit.geomID = leaf_geom(L);
          coldU(6) = hit.leaf; coldF(7) = hit.t;
          o = o + d * hit.t;                                          // updateHit, Render.hpp:15-23
          nrm = hit_normal(sc, hit, o);
          mi_material mat;                                                                              // = materials[matIDs[geomID]]
          if (matsInLds) { mat = matS[L.matIndex].m; asm volatile("" ::: "memory"); } else mat = sc.materials[L.matIndex];      // (the empty asm keeps the two loads apart)
          const f3 albedo = mk(mat.albedo.x, mat.albedo.y, mat.albedo.z);
          if (mat.emissive) color = color + tp * mk(mat.emission.x, mat.emission.y, mat.emission.z);
          if (mat.type == 0) {
            const float u1 = rng_uniform01(rng);
            const float u2 = rng_uniform01(rng);
            d = sample_diffuse(nrm, u1, u2, sinTbl);
            tp = tp * albedo;
          } else if (mat.type == 1) {
            d = reflect_dir(d, nrm);
            tp = tp * albedo;
          } else if (mat.type == 2) {
            const float u1 = rng_uniform01(rng);
            f3 nd2;
            const bool refracted = dielectric(d, nrm, mat.ior, u1, nd2);
            d = nd2;
            if (refracted) tp = tp * albedo;
          } else {
            const float qn = __builtin_nanf("");
            coldF(3) = coldF(3) * qn; coldF(4) = coldF(4) * qn; coldF(5) = coldF(5) * qn;
            oFlags |= MI_FLAG_ERROR;
          }
        } else {
          coldF(7) = kInf;
          oFlags |= MI_FLAG_ESCAPED;
          terminated = true;
        }
        if (!terminated && bounce > sc.rouletteStartDepth) {
          const float u1 = rng_uniform01(rng);
          if (roulette_stop(u1, tp)) terminated = true;
        }
        bounce++;
        if (bounce >= sc.maxPathLength) terminated = true;
        if (terminated) {
          const uint32_t pixNow = getPix();
          mi_trace_result* res = rays + pixNow;
          if (slots) {
            const size_t q = (size_t)sample * n + pixNow;
            ex.slotColor[3 * q] = color.x; ex.slotColor[3 * q + 1] = color.y; ex.slotColor[3 * q + 2] = color.z;
            envRay = (oFlags & MI_FLAG_ESCAPED) != 0;
            envSlot = (uint32_t)q;
            if (!envRay) ex.u[q] = -1.f;
          } else { coldF(3) = coldF(3) + color.x; coldF(4) = coldF(4) + color.y; coldF(5) = coldF(5) + color.z; }
          pathEnd = true;
          ++sample;
          const bool more = segd ? ((sample & segMask) != 0u && sample < spp) : (sample < spp);
          if (more) ph = PH_GEN;
          else if (segd && sample < spp) {
            // a segment other than the last is complete: its partial sum (or its slots) is all it leaves
            if (!slots && ex.segPart) {
              float* part = ex.segPart + 3 * ((size_t)(((sample - 1u) >> segShift) - ex.segBase) * n + pixNow);
              part[0] = coldF(3); part[1] = coldF(4); part[2] = coldF(5);
            }
            ph = PH_FETCH;
          } else {
            // pixel complete: rgb sum + the LAST sample's hit record
            if (!slots && ex.segPart) {
              float* part = ex.segPart + 3 * ((size_t)(((sample - 1u) >> segShift) - ex.segBase) * n + pixNow);
              part[0] = coldF(3); part[1] = coldF(4); part[2] = coldF(5);
            } else
            if (!slots) res->rgb = {coldF(3), coldF(4), coldF(5)};
            uint32_t oPrim = MI_INVALID_PRIM, oGeom = MI_INVALID_GEOM;
            const uint32_t lastLeaf = coldU(6);
            if (lastLeaf != 0xFFFFFFFFu) { const GLeaf LL = sc.leaves[lastLeaf]; oPrim = LL.primID; oGeom = leaf_geom(LL); }
            mi_hit_record hr;
            hr.r.origin = {o.x, o.y, o.z}; hr.r.t_min = 0.f;
            hr.r.direction = {d.x, d.y, d.z}; hr.r.t_max = coldF(7);
            hr.prim_id = oPrim;
            hr.normal = {nrm.x, nrm.y, nrm.z};
            hr.throughput = {tp.x, tp.y, tp.z};
            hr.geom_id = (uint16_t)oGeom; hr.flags = (uint16_t)oFlags;
            res->h = hr;
            ph = PH_FETCH;
          }
        } else {
          // next bounce: offsetRay + cast set-up (codelets :207-211)
          o = offset_origin(o, d, nrm);
          if (FAST) { inv = fast_inverse(d); sh = make_shear_fast(d, inv); }
          else {
            inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
            exactSlab = !(fabsf(inv.x) < kInf && fabsf(inv.y) < kInf && fabsf(inv.z) < kInf && fabsf(o.x) < kInf && fabsf(o.y) < kInf && fabsf(o.z) < kInf);
            sh = make_shear(d, inv);
          }
          if (FAST) fast_box_setup(o, inv, oi, slabPad);
          hit.t = kInf; hit.leaf = 0xFFFFFFFFu;
          { uint32_t seen; node = root_start(sc, o, seen) << 5; if (STATS) cs.nodes += seen; }
          ph = (numNodes > 0) ? PH_NODE : PH_SHADE;
        }
        pathStore();
      }
      { const uint32_t ended = (uint32_t)__popcll(__ballot(pathEnd)); paths += ended; casts += cS - ended; }    // every SHADE lane either ends its path or casts again
      if (slots) pushEscaped(envRay, envSlot);
      if (STATS) tShade += __builtin_amdgcn_s_memtime() - tq1;
    } else {
      // ---------------- GEN: camera ray of the next sample ----------------
      if (STATS) { itG++; lnG += cG; }
      const unsigned long long tq2 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
      if (ph == PH_GEN) {
        pathLoad();
        if constexpr (kCoordsFromStream) {
          const uint32_t e = coldU(0);
          if (ex.coords) { const float2 pc = ex.coords[e]; prow = pc.x; pcol = pc.y; } else { prow = rays[e].u; pcol = rays[e].v; }
        } else { prow = coldF(1); pcol = coldF(2); }
        float g0, g1;
        rng_gauss2(rng, sinTbl, g0, g1);
        const float jr = prow + sc.antiAliasScale * g0, jc = pcol + sc.antiAliasScale * g1;
        d = pixel_to_ray_dir(jc, jr, sc.imageWidth, sc.imageHeight, sc.tanTheta);
        o = mk(0.f, 0.f, 0.f);
        nrm = mk(0.f, 0.f, 1.f);                         // HitRecord ctor, geometry.hpp:236-242
        coldU(6) = 0xFFFFFFFFu; coldF(7) = kInf;
        oFlags = 0;
        tp = mk(1.f, 1.f, 1.f);
        color = mk(0.f, 0.f, 0.f);
        bounce = 0;
        o = offset_origin(o, d, nrm);
        if (FAST) { inv = fast_inverse(d); sh = make_shear_fast(d, inv); }
        else {
          inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
          exactSlab = !(fabsf(inv.x) < kInf && fabsf(inv.y) < kInf && fabsf(inv.z) < kInf && fabsf(o.x) < kInf && fabsf(o.y) < kInf && fabsf(o.z) < kInf);
          sh = make_shear(d, inv);
        }
        if (FAST) fast_box_setup(o, inv, oi, slabPad);
        hit.t = kInf; hit.leaf = 0xFFFFFFFFu;
        { uint32_t seen; node = root_start(sc, o, seen) << 5; if (STATS) cs.nodes += seen; }
        ph = (numNodes > 0) ? PH_NODE : PH_SHADE;
        pathStore();
      }
      casts += cG;
      if (STATS) tGen += __builtin_amdgcn_s_memtime() - tq2;
    }
  }
  if (slots) {      // the rest of the wave's last reservation: its block padded, the blocks behind it marked as holes
    const uint32_t padEnd = min(envEnd, (envNext + kEnvBlock - 1u) & ~(kEnvBlock - 1u));
    for (uint32_t k = envNext + lane; k < envEnd; k += 64u) ex.index[k] = k < padEnd ? envFill : kEnvHole;
  }
  flush_stats(sc, lane == 0 ? casts : 0u, cs, lane == 0 ? paths : 0u);
  if (STATS && lane == 0) {
    atomicAdd(&sc.counters[4], (unsigned long long)itN); atomicAdd(&sc.counters[5], (unsigned long long)lnN);
    atomicAdd(&sc.counters[6], (unsigned long long)itL); atomicAdd(&sc.counters[7], (unsigned long long)lnL);
    atomicAdd(&sc.counters[16], (unsigned long long)itQ); atomicAdd(&sc.counters[17], (unsigned long long)lnQ);      // (read through mi_get_pool_stats: slots 0 and 1; the pool kernel is not this kernel)
    atomicAdd(&sc.counters[8], (unsigned long long)itS); atomicAdd(&sc.counters[9], (unsigned long long)lnS);
    atomicAdd(&sc.counters[10], (unsigned long long)itG); atomicAdd(&sc.counters[11], (unsigned long long)lnG);
    atomicAdd(&sc.counters[12], tTrav); atomicAdd(&sc.counters[13], tShade); atomicAdd(&sc.counters[14], tGen);
    atomicAdd(&sc.counters[15], __builtin_amdgcn_s_memtime() - tLoop0);
  }
}

// WaveExtras::coords of a stream
__global__ void __launch_bounds__(256) pixel_coords_kernel(const mi_trace_result* rays, uint32_t n, float2* __restrict__ coords) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) coords[i] = make_float2(rays[i].u, rays[i].v);
}

// rgb of a segmented pixel: ((segment 0, which started from the incoming rgb) + segment 1) + ... in segment order.
__global__ void __launch_bounds__(256) segment_combine_kernel(mi_trace_result* rays, uint32_t n, uint32_t segments, const float* __restrict__ part, uint32_t continues) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // the first launch's segment 0 already holds rgb_in + its samples; a later launch continues the running sum
  f3 rgb = mk(part[3 * (size_t)i], part[3 * (size_t)i + 1], part[3 * (size_t)i + 2]);
  if (continues) { const mi_vec3 acc = rays[i].rgb; rgb = mk(acc.x, acc.y, acc.z) + rgb; }
  for (uint32_t s = 1; s < segments; ++s) {
    const size_t q = 3 * ((size_t)s * n + i);
    rgb = rgb + mk(part[q], part[q + 1], part[q + 2]);
  }
  rays[i].rgb = {rgb.x, rgb.y, rgb.z};
}

}  // namespace mi
