// trace_pool.hpp — K1p: the path-trace hot loop with the paths DECOUPLED from the lanes.
//
// Why (profiles/r01_final_pmc.csv, tools/phase_probe.py on K1w): in the phase-scheduled kernel every lane owns one
// path, so a NODE turn runs with the 56 % of the lanes whose path happens to be walking the BVH; the others wait for
// a SHADE turn (46 % of the idle lanes), a primitive test (27 %) or a GEN turn (6 %), and SHADE / GEN themselves run
// with 70 % / 32 % of the lanes. Here a WORKGROUP owns a pool of PWG path slots, more than it has lanes, and its
// lanes are workers:
//   * a slot's hot state lives in LDS (ray 11 words - overwritten by the 5-word hit record when the traversal
//     retires -, RNG 4, radiance 3, throughput 3, counters 1; [word][slot]); what is touched once per path or less
//     (pixel index and coordinates, the work unit's rgb sum, the last hit's normal and leaf for the AOVs) lives in an
//     L2-resident scratch array. Both are shared by the waves of the workgroup, which all run on one CU: LDS
//     operations of a wave execute in order, and workgroup-scope release / acquire fences order the rest (on gfx950
//     they cost an s_waitcnt lgkmcnt(0): the CU's vector L1 is shared, no cache maintenance, no vmcnt wait);
//   * four rings of slot numbers in LDS - READY (ray waits for a traversal lane), SHADE (traversal finished), GEN
//     (next sample's camera ray), FETCH (work unit finished). A wave pushes with one ds_add on the ring's tail and
//     plain stores, pops with one compare-and-swap on its head; an entry that has been reserved but not written yet
//     reads as EMPTY and is simply re-read;
//   * a lane that finishes a traversal RETIRES it (5 words to the slot, slot number to the SHADE ring) and REFILLS
//     itself from the READY ring (9 words): NODE turns stay nearly full. Whichever wave finds a full batch on the
//     SHADE or GEN ring (and has fewer lanes walking than the batch holds) serves it with all 64 lanes; the rays its
//     lanes were walking just stay in their registers meanwhile.
// Every path still performs exactly the reference's sequence of operations, in the reference's order, with the same
// arithmetic (the blocks below are those of trace_wavefront.hpp, cited there line by line); only WHICH lane executes
// a step, and when, changes - so every byte of every TraceResult stays equal to the oracle's.
#pragma once

#include "trace_wavefront.hpp"

namespace mi {

// Scheduling of one wave (all counts in lanes / slots; weights are quarter units, a walking lane weighs 4):
//   leafAt        inside a traversal burst a LEAF turn runs when cL * leafAt > cN * 4
//   burst         at most this many NODE/LEAF steps before the wave looks at the rings again
//   retireAt      ... or earlier, once this many lanes have finished their traversal
//   refillMin     free lanes are refilled from the READY ring when at least this many can be served
//   shadeW, genW  vote: SHADE / GEN are served when min(ring, 64) * weight exceeds (lanes walking) * 4
//   dbl, maxExtra, leafThenNode, prio   as in WaveTune
struct PoolTune { uint32_t leafAt = 4, burst = 48, retireAt = 16, refillMin = 8, shadeW = 4, genW = 4, dbl = 4, maxExtra = 5, leafThenNode = 1, prio = 1; };

enum : uint32_t { PP_NODE = 0, PP_LEAF = 1, PP_FIN = 2, PP_FREE = 3 };

// LDS words of a slot
enum : uint32_t {
  PW_O = 0, PW_D = 3, PW_INV = 6, PW_SX = 9, PW_SY = 10,                  // the ray as the traversal wants it
  PW_HT = 6, PW_HLEAF = 7, PW_HB0 = 8, PW_HB1 = 9, PW_HB2 = 10,           // ... overwritten by the hit when it retires
  PW_RNG = 11, PW_COLOR = 15, PW_TP = 18, PW_CNT = 21, PW_WORDS = 22
};
// scratch words of a slot (global memory, [word][slot of the whole grid])
enum : uint32_t { PG_PIX = 0, PG_ROW = 1, PG_COL = 2, PG_RGB = 3, PG_NRM = 6, PG_LEAF = 9, PG_WORDS = 10 };
// PW_CNT: sample 0..18 | bounce 19..26 | flags 27..28 | kz 29..30 | exactSlab 31
constexpr uint32_t kPoolMaxSamples = (1u << 19) - 1u, kPoolMaxBounces = 255u;
enum : uint32_t { RING_READY = 0, RING_SHADE = 1, RING_GEN = 2, RING_FETCH = 3 };
constexpr uint32_t kRingEmpty = 0xFFFFu;

// LDS of a workgroup: [PW_WORDS][PWG] u32 | 4 rings x RCAP u16 | ring heads and tails | flags | sin table
__host__ __device__ constexpr uint32_t pool_ring_cap(uint32_t pwg) { return pwg <= 512u ? 512u : (pwg <= 1024u ? 1024u : 2048u); }
__host__ __device__ constexpr size_t pool_lds_bytes(uint32_t pwg) { return (size_t)PW_WORDS * pwg * 4u + 4u * pool_ring_cap(pwg) * 2u + 16u * 4u; }

template <bool STATS, int WAVES, int PWG, int WAVES_PER_SIMD>
__global__ void __launch_bounds__(64 * WAVES, WAVES_PER_SIMD) path_trace_pool_kernel(DeviceScene sc, mi_trace_result* rays, uint32_t n, uint32_t* workCounter, PoolTune tune,
                                                                                    uint32_t tileStreamW, WaveExtras ex, uint32_t* scratch, uint32_t scratchStride) {
  static_assert(PWG >= 64 * WAVES && PWG <= 2048, "pool size");
  constexpr uint32_t RCAP = pool_ring_cap(PWG), RM = RCAP - 1;
  __shared__ float sinTbl[92];
  extern __shared__ __attribute__((aligned(16))) unsigned char dynLds[];
  uint32_t* const wp = reinterpret_cast<uint32_t*>(dynLds);
  volatile uint16_t* const ringEnt = reinterpret_cast<volatile uint16_t*>(dynLds + (size_t)PW_WORDS * PWG * 4u);          // [4][RCAP]
  uint32_t* const ctl = reinterpret_cast<uint32_t*>(dynLds + (size_t)PW_WORDS * PWG * 4u + 4u * RCAP * 2u);               // head[4], tail[4], drained
  volatile uint32_t* const vctl = ctl;
  {
    for (uint32_t i = threadIdx.x; i < 4u * RCAP; i += blockDim.x) ringEnt[i] = (uint16_t)((i >= RING_FETCH * RCAP && i < RING_FETCH * RCAP + (uint32_t)PWG) ? (i - RING_FETCH * RCAP) : kRingEmpty);
    if (threadIdx.x < 16u) ctl[threadIdx.x] = (threadIdx.x == 4u + RING_FETCH) ? (uint32_t)PWG : 0u;      // every slot starts on the FETCH ring
  }
  load_sin_table(sinTbl);            // (ends with __syncthreads)

  const uint32_t lane = threadIdx.x & 63;
  const uint32_t gslotBase = blockIdx.x * (uint32_t)PWG;
  auto SU = [&](uint32_t w, uint32_t s) -> uint32_t& { return wp[w * (uint32_t)PWG + s]; };
  auto SF = [&](uint32_t w, uint32_t s) -> float& { return reinterpret_cast<float*>(wp)[w * (uint32_t)PWG + s]; };
  auto GU = [&](uint32_t w, uint32_t s) -> uint32_t& { return scratch[(size_t)w * scratchStride + gslotBase + s]; };
  auto GF = [&](uint32_t w, uint32_t s) -> float& { return reinterpret_cast<float*>(scratch)[(size_t)w * scratchStride + gslotBase + s]; };
  auto lanesBelow = [&](unsigned long long m) -> uint32_t { return (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); };
  auto uni = [&](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };

  // ---- rings ----
  // number of entries on a ring as this wave sees it now (reserved entries included)
  // (head first, then tail, each read exactly once: the tail can only have grown in between, so the difference never wraps
  // downwards; other waves may push and claim between the two reads, hence the clamp to the ring's capacity)
  auto ringCount = [&](uint32_t r) -> uint32_t {
    const uint32_t head = vctl[r];
    const uint32_t tail = vctl[4 + r];
    const uint32_t c = tail - head;
    return c > (uint32_t)PWG ? (uint32_t)PWG : c;
  };
  // push the slots of the lanes with p: everything the slot needs (LDS words, scratch) was written before
  auto ringPush = [&](uint32_t r, bool p, uint32_t s) {
    const unsigned long long m = __ballot(p);
    if (m == 0ull) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    uint32_t base = 0;
    if (lane == (uint32_t)__ffsll((long long)m) - 1u) base = atomicAdd(&ctl[4 + r], (uint32_t)__popcll(m));
    base = uni(__shfl(base, (int)((uint32_t)__ffsll((long long)m) - 1u)));
    if (p) {
      volatile uint16_t* e = ringEnt + r * RCAP + ((base + lanesBelow(m)) & RM);
      while (*e != kRingEmpty) __builtin_amdgcn_s_sleep(1);      // (the entry's previous occupant has been claimed but not read yet: never seen in practice; its reader needs the LDS port this poll would occupy)
      *e = (uint16_t)s;
    }
  };
  // claim up to kmax entries; lanes [0, k) then read entry (base + lane)
  auto ringClaim = [&](uint32_t r, uint32_t kmax, uint32_t& base) -> uint32_t {
    uint32_t k = 0, h = 0;
    if (lane == 0) {
      for (;;) {
        h = vctl[r];
        k = min(vctl[4 + r] - h, kmax);
        if (k == 0 || atomicCAS(&ctl[r], h, h + k) == h) break;
      }
    }
    base = uni(h);
    return uni(k);
  };
  auto ringTake = [&](uint32_t r, uint32_t idx) -> uint32_t {
    volatile uint16_t* e = ringEnt + r * RCAP + (idx & RM);
    uint32_t s;
    for (;;) { s = *e; if (s != kRingEmpty) break; __builtin_amdgcn_s_sleep(1); }        // reserved by a pushing wave, not written yet: back off while it writes
    *e = (uint16_t)kRingEmpty;
    return s;
  };
  auto acquire = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); };

  const uint32_t tiledCount = tileStreamW ? (n / (8u * tileStreamW)) * (8u * tileStreamW) : 0u;
  const uint32_t numNodes = sc.numNodes;
  const uint32_t spp = ex.sampleCount ? ex.sampleCount : sc.samplesPerPixel;
  const bool segd = ex.segPart != nullptr || ex.slotColor != nullptr;
  const uint32_t segShift = segment_shift(sc.samplesPerPixel), segMask = (1u << segShift) - 1u;
  const uint32_t segs = segd ? ex.segments : 1u;
  const uint32_t items = n * segs;

  // ---- the lane's own traversal (its ray stays in registers across SHADE / GEN / FETCH turns) ----
  uint32_t ph = PP_FREE, slot = 0, node = 0, pendLeaf = 0;
  f3 o = mk(0, 0, 0), inv = mk(0, 0, 0);
  Shear sh; sh.kz = 2; sh.sx = sh.sy = 0.f; sh.sz = 1.f;
  Hit hit; hit.t = kInf; hit.leaf = 0xFFFFFFFFu; hit.geomID = 0xFFFFu; hit.b0 = hit.b1 = hit.b2 = 0.f;
  bool exactSlab = false;

  CastStats cs = {0, 0};
  uint32_t casts = 0, paths = 0;
  uint32_t itN = 0, itL = 0, itS = 0, itG = 0, lnN = 0, lnL = 0, lnS = 0, lnG = 0;
  unsigned long long tTrav = 0, tShade = 0, tGen = 0, tLoop0 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
  uint32_t itLoop = 0, itRefill = 0, lnRefill = 0, itIdle = 0, itFail = 0, itBurst = 0, lnBurst = 0;      // STATS: scheduler bookkeeping
  unsigned long long tRefill = 0;

  for (;;) {
    if (STATS) itLoop++;
    bool drained = vctl[8] != 0u;                  // some wave has seen the work counter pass the end of the launch
    // ---------------- FETCH: slots whose work unit is finished take the next (pixel, segment) unit ----------------
    while (!drained && ringCount(RING_FETCH) > 0) {
      // One global atomic per turn hands out exactly as many work indices as slots were claimed, so no wave ever
      // holds indices it might not use; the first index past the end tells every wave to stop fetching.
      uint32_t fbase;
      const uint32_t take = ringClaim(RING_FETCH, 64u, fbase);
      if (take == 0) break;
      const bool mine = lane < take;
      const uint32_t fs = mine ? ringTake(RING_FETCH, fbase + lane) : 0u;
      uint32_t ibase = 0;
      if (lane == 0) ibase = atomicAdd(workCounter, take);
      const uint32_t idx = uni(ibase) + lane;
      const bool live = mine && idx < items;
      if (live) {
        // (the 8x8 tile walk of trace_wavefront.hpp: a bijection on [0, n), any order gives the same image)
        uint32_t seg = 0, pidx = idx;
        if (segd) { const uint32_t local = idx / n; pidx = idx - local * n; seg = ex.segBase + local; }
        uint32_t entry = pidx;
        if (tileStreamW && pidx < tiledCount) {
          const uint32_t t = pidx >> 6, within = pidx & 63u, perRow = tileStreamW >> 3;
          entry = ((t / perRow) * 8u + (within >> 3)) * tileStreamW + (t % perRow) * 8u + (within & 7u);
        }
        const mi_trace_result* res = rays + entry;
        const float prow = res->u, pcol = res->v;
        GU(PG_PIX, fs) = entry; GF(PG_ROW, fs) = prow; GF(PG_COL, fs) = pcol;
        if (seg == 0) { GF(PG_RGB, fs) = res->rgb.x; GF(PG_RGB + 1, fs) = res->rgb.y; GF(PG_RGB + 2, fs) = res->rgb.z; }
        else { GF(PG_RGB, fs) = 0.f; GF(PG_RGB + 1, fs) = 0.f; GF(PG_RGB + 2, fs) = 0.f; }
        Rng rng;
        rng_seed_pixel_segment(rng, sc.rngSeed, prow, pcol, seg);
        SU(PW_RNG, fs) = (uint32_t)rng.s0; SU(PW_RNG + 1, fs) = (uint32_t)(rng.s0 >> 32);
        SU(PW_RNG + 2, fs) = (uint32_t)rng.s1; SU(PW_RNG + 3, fs) = (uint32_t)(rng.s1 >> 32);
        SU(PW_CNT, fs) = (ex.slotColor ? seg - ex.segBase : seg) << segShift;       // sample index; bounce, flags = 0
      }
      ringPush(RING_GEN, live, fs);
      if (__ballot(mine && !live)) { drained = true; if (lane == 0) vctl[8] = 1u; }     // indices past the end: no more work (the dead slots are dropped)
    }

    // ---------------- what is there to do ----------------
    uint32_t cN = (uint32_t)__popcll(__ballot(ph == PP_NODE)), cL = (uint32_t)__popcll(__ballot(ph == PP_LEAF));
    const uint32_t cT = cN + cL;
    const uint32_t nR = ringCount(RING_READY), nS = ringCount(RING_SHADE), nG = ringCount(RING_GEN);
    if ((cT | nR | nS | nG) == 0 && drained) break;       // nothing queued, nothing to fetch: the other waves finish what their lanes hold

    // ---------------- REFILL: free lanes take rays from the READY ring ----------------
    {
      const uint32_t want = min(nR, 64u - cT);
      if (want > 0 && (want >= tune.refillMin || cT == 0 || (nS | nG) == 0)) {
        const unsigned long long tqr = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
        uint32_t rbase;
        const uint32_t got = ringClaim(RING_READY, want, rbase);
        if (STATS) { itRefill++; lnRefill += got; }
        const unsigned long long mF = __ballot(ph == PP_FREE);
        const uint32_t rank = lanesBelow(mF);
        if (ph == PP_FREE && rank < got) {
          const uint32_t s = ringTake(RING_READY, rbase + rank);
          acquire();
          slot = s;
          o = mk(SF(PW_O, s), SF(PW_O + 1, s), SF(PW_O + 2, s));
          inv = mk(SF(PW_INV, s), SF(PW_INV + 1, s), SF(PW_INV + 2, s));
          sh.sx = SF(PW_SX, s); sh.sy = SF(PW_SY, s);
          const uint32_t cnt = SU(PW_CNT, s);
          sh.kz = (cnt >> 29) & 3u;
          exactSlab = (cnt >> 31) != 0u;
          sh.sz = comp(inv, sh.kz);                       // 1 / d[kz]: the same IEEE division (make_shear(d, inv))
          hit.t = kInf; hit.leaf = 0xFFFFFFFFu;
          node = 0;
          ph = (numNodes > 0) ? PP_NODE : PP_FIN;
        }
        if (STATS) tRefill += __builtin_amdgcn_s_memtime() - tqr;
        if (got > 0) continue;
      }
    }

    // ---------------- vote: walk, or serve a ring ----------------
    uint32_t run;          // 0 = TRAVERSE, 2 = SHADE, 3 = GEN, 4 = nothing to do right now (another wave holds the work)
    {
      const uint32_t wT = cT * 4u, wS = min(nS, 64u) * tune.shadeW, wG = min(nG, 64u) * tune.genW;
      if (cT > 0 && wT >= max(wS, wG)) run = 0;
      else run = (nS > 0 && wS >= wG) ? 2 : (nG > 0 ? 3 : (nS > 0 ? 2 : (cT > 0 ? 0 : 4)));
    }
    if (run == 4) { if (STATS) itIdle++; __builtin_amdgcn_s_sleep(8); continue; }

    if (run == 0) {
      // ---------------- TRAVERSE: NODE and LEAF steps under a two-way mini-vote (trace_wavefront.hpp) ----------------
      const unsigned long long tq0 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
      if (tune.prio == 1) __builtin_amdgcn_s_setprio(1);
      if (STATS) { itBurst++; lnBurst += cT; }
      uint32_t steps = 0;
      const bool anyExact = __ballot(exactSlab && ph <= PP_LEAF) != 0ull;
      auto nodeBodyT = [&](auto exactTag) -> bool {
        GNode nd = *reinterpret_cast<const GNode*>(reinterpret_cast<const char*>(sc.nodes) + (node << 5));
        if (STATS) cs.nodes++;
        // Box test (CompactBVH2Node.cpp:5-22, intersectRaySlab CompactBVH2Node.hpp:14-50); min/max form and the
        // literal fallback exactly as in trace_wavefront.hpp
        const float ax = (nd.minx - o.x) * inv.x, bx = (nd.maxx - o.x) * inv.x;
        const float ay = (nd.miny - o.y) * inv.y, by = (nd.maxy - o.y) * inv.y;
        const float az = (nd.minz - o.z) * inv.z, bz = (nd.maxz - o.z) * inv.z;
        float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.f));
        float t1 = fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)) * kSlabScale, hit.t);
        if constexpr (decltype(exactTag)::value) {
          if (exactSlab) {
            t0 = 0.f; t1 = hit.t;
            { float tmin = ax, tmax = bx; if (tmin > tmax) { const float s = tmin; tmin = tmax; tmax = s; } tmax *= kSlabScale; t0 = tmin > t0 ? tmin : t0; t1 = tmax < t1 ? tmax : t1; }
            { float tmin = ay, tmax = by; if (tmin > tmax) { const float s = tmin; tmin = tmax; tmax = s; } tmax *= kSlabScale; t0 = tmin > t0 ? tmin : t0; t1 = tmax < t1 ? tmax : t1; }
            { float tmin = az, tmax = bz; if (tmin > tmax) { const float s = tmin; tmin = tmax; tmax = s; } tmax *= kSlabScale; t0 = tmin > t0 ? tmin : t0; t1 = tmax < t1 ? tmax : t1; }
          }
        }
        const bool boxHit = !(t0 > t1);
        const bool isLeaf = node_is_leaf(nd);
        pendLeaf = node;                                          // (leaves[] is indexed by node)
        node = (boxHit && !isLeaf) ? node + 1 : (nd.link >> 5);          // (a lane that waits for a primitive test already stands at the node behind it)
        if (boxHit && isLeaf) { ph = PP_LEAF; return false; }
        if (node >= numNodes) { ph = PP_FIN; return false; }
        return true;
      };
      auto nodeBody = [&]() -> bool { return nodeBodyT(std::false_type{}); };
      auto nodeStep = [&]() { if (ph == PP_NODE) (void)(anyExact ? nodeBodyT(std::true_type{}) : nodeBodyT(std::false_type{})); };
      for (;;) {
        const uint32_t stay = cN;
        if (cN * 4u >= cL * tune.leafAt && cN > 0) {
          if (STATS) { itN++; lnN += stay; }
          const uint32_t extra = min(stay / tune.dbl, tune.maxExtra);
          if (STATS) {
            nodeStep();
            for (uint32_t e = 0; e < extra; ++e) { itN++; lnN += (uint32_t)__popcll(__ballot(ph == PP_NODE)); nodeStep(); }
          } else if (anyExact) {
            nodeStep();
            for (uint32_t e = 0; e < extra; ++e) nodeStep();
          } else if (ph == PP_NODE) {
            bool go = nodeBody();
            if (extra >= 1 && go) { go = nodeBody();
              if (extra >= 2 && go) { go = nodeBody();
                if (extra >= 3 && go) { go = nodeBody();
                  if (extra >= 4 && go) { go = nodeBody();
                    if (extra >= 5 && go) (void)nodeBody();
                  }
                }
              }
            }
          }
          steps += extra;
        } else {
          if (STATS) { itL++; lnL += cL; }
          if (ph == PP_LEAF) {
            if (STATS) cs.leaves++;
            const GLeaf L = sc.leaves[pendLeaf];
            float t, b0 = 0.f, b1 = 0.f, b2 = 0.f;
            bool cand;
            const uint32_t kind = leaf_kind(L);
            if (kind == LEAF_TRI) {
              t = intersect_triangle(mk(L.f[0], L.f[1], L.f[2]), mk(L.f[3], L.f[4], L.f[5]), mk(L.f[6], L.f[7], L.f[8]), o, sh, b0, b1, b2);
              cand = t > 0.f && t < kInf;
            } else {
              const f3 d = mk(SF(PW_D, slot), SF(PW_D + 1, slot), SF(PW_D + 2, slot));      // (only spheres and discs need the direction itself)
              if (kind == LEAF_SPHERE) t = intersect_sphere(L, o, d, 0.f);
              else t = intersect_disc(L, o, d);
              cand = true;
            }
            if (cand && t > 0.f && t < hit.t) { hit.t = t; hit.leaf = pendLeaf; hit.b0 = b0; hit.b1 = b1; hit.b2 = b2; }
            ph = (node >= numNodes) ? PP_FIN : PP_NODE;
          }
          if (tune.leafThenNode) {
            if (STATS) { itN++; lnN += (uint32_t)__popcll(__ballot(ph == PP_NODE)); }
            nodeStep();
            ++steps;
          }
        }
        cN = (uint32_t)__popcll(__ballot(ph == PP_NODE));
        cL = (uint32_t)__popcll(__ballot(ph == PP_LEAF));
        if (++steps >= tune.burst || (cN + cL) == 0 || (cT - (cN + cL)) >= tune.retireAt) break;
      }
      if (tune.prio == 1) __builtin_amdgcn_s_setprio(0);
      // ---- RETIRE: finished traversals hand their hit to the slot and queue it for shading ----
      {
        const bool fin = ph == PP_FIN;
        if (fin) {
          SF(PW_HT, slot) = hit.t; SU(PW_HLEAF, slot) = hit.leaf;
          SF(PW_HB0, slot) = hit.b0; SF(PW_HB1, slot) = hit.b1; SF(PW_HB2, slot) = hit.b2;
          ph = PP_FREE;
        }
        ringPush(RING_SHADE, fin, slot);
      }
      if (STATS) tTrav += __builtin_amdgcn_s_memtime() - tq0;
    } else if (run == 2) {
      // ---------------- SHADE: up to 64 finished traversals (codelets/TraceCodelets.cpp:214-257) ----------------
      const unsigned long long tq1 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
      uint32_t sbase;
      const uint32_t k = ringClaim(RING_SHADE, 64u, sbase);
      if (k == 0) { if (STATS) itFail++; continue; }                        // another wave was faster
      if (STATS) { itS++; lnS += k; }
      const bool mine = lane < k;
      const uint32_t s = mine ? ringTake(RING_SHADE, sbase + lane) : 0u;
      acquire();
      bool toReady = false, toGen = false, toFetch = false;
      bool envRay = false;
      uint32_t envSlot = 0;
      f3 envDir = mk(0, 0, 0), envTp = mk(0, 0, 0);
      if (mine) {
        Hit h2; h2.t = SF(PW_HT, s); h2.leaf = SU(PW_HLEAF, s); h2.b0 = SF(PW_HB0, s); h2.b1 = SF(PW_HB1, s); h2.b2 = SF(PW_HB2, s); h2.geomID = 0xFFFFu;
        f3 so = mk(SF(PW_O, s), SF(PW_O + 1, s), SF(PW_O + 2, s));
        f3 sd = mk(SF(PW_D, s), SF(PW_D + 1, s), SF(PW_D + 2, s));
        Rng rng;
        rng.s0 = (uint64_t)SU(PW_RNG, s) | ((uint64_t)SU(PW_RNG + 1, s) << 32); rng.s1 = (uint64_t)SU(PW_RNG + 2, s) | ((uint64_t)SU(PW_RNG + 3, s) << 32);
        f3 color = mk(SF(PW_COLOR, s), SF(PW_COLOR + 1, s), SF(PW_COLOR + 2, s));
        f3 tp = mk(SF(PW_TP, s), SF(PW_TP + 1, s), SF(PW_TP + 2, s));
        const uint32_t cnt = SU(PW_CNT, s);
        // the unit's running rgb sum and pixel are only used when the path ends; asked for now, they arrive meanwhile
        const f3 rgbIn = mk(GF(PG_RGB, s), GF(PG_RGB + 1, s), GF(PG_RGB + 2, s));
        const uint32_t pixNow = GU(PG_PIX, s);
        uint32_t sample = cnt & kPoolMaxSamples, bounce = (cnt >> 19) & 0xFFu, oFlags = (cnt >> 27) & 3u;
        const bool lastSample = sample + 1u >= spp;          // this path's final state is the pixel's AOV record
        f3 nrm = mk(0.f, 0.f, 1.f);                          // HitRecord ctor, geometry.hpp:236-242
        uint32_t lastLeaf = 0xFFFFFFFFu;
        bool terminated = false, rgbNaN = false;
        const bool gotHit = h2.leaf != 0xFFFFFFFFu;
        if (gotHit) {
          const GLeaf L = sc.leaves[h2.leaf];
          h2.geomID = leaf_geom(L);
          lastLeaf = h2.leaf;
          so = so + sd * h2.t;                                          // updateHit, Render.hpp:15-23
          nrm = hit_normal(sc, h2, so);
          const mi_material mat = sc.materials[L.matIndex];
          const f3 albedo = mk(mat.albedo.x, mat.albedo.y, mat.albedo.z);
          if (mat.emissive) color = color + tp * mk(mat.emission.x, mat.emission.y, mat.emission.z);
          if (mat.type == 0) {
            const float u1 = rng_uniform01(rng);
            const float u2 = rng_uniform01(rng);
            sd = sample_diffuse(nrm, u1, u2, sinTbl);
            tp = tp * albedo;
          } else if (mat.type == 1) {
            sd = reflect_dir(sd, nrm);
            tp = tp * albedo;
          } else if (mat.type == 2) {
            const float u1 = rng_uniform01(rng);
            f3 nd2;
            const bool refracted = dielectric(sd, nrm, mat.ior, u1, nd2);
            sd = nd2;
            if (refracted) tp = tp * albedo;
          } else {
            rgbNaN = true;                                   // rgb *= NaN (codelets :240-244)
            oFlags |= MI_FLAG_ERROR;
          }
        } else {
          oFlags |= MI_FLAG_ESCAPED;
          terminated = true;
          if (lastSample && bounce > 0u) {                   // the AOVs keep the last HIT's normal and primitive
            nrm = mk(GF(PG_NRM, s), GF(PG_NRM + 1, s), GF(PG_NRM + 2, s));
            lastLeaf = GU(PG_LEAF, s);
          }
        }
        f3 sum = rgbIn;
        if (rgbNaN) { const float qn = __builtin_nanf(""); sum = mk(sum.x * qn, sum.y * qn, sum.z * qn); }
        if (!terminated && bounce > sc.rouletteStartDepth) {
          const float u1 = rng_uniform01(rng);
          if (roulette_stop(u1, tp)) terminated = true;
        }
        bounce++;
        if (bounce >= sc.maxPathLength) terminated = true;
        if (terminated) {
          mi_trace_result* res = rays + pixNow;
          if (ex.slotColor) {
            const size_t q = (size_t)sample * n + pixNow;
            ex.slotColor[3 * q] = color.x; ex.slotColor[3 * q + 1] = color.y; ex.slotColor[3 * q + 2] = color.z;
            envRay = (oFlags & MI_FLAG_ESCAPED) != 0;
            envSlot = (uint32_t)q;
            envDir = sd; envTp = tp;
            if (!envRay) ex.u[q] = -1.f;
          } else {
            sum = mk(sum.x + color.x, sum.y + color.y, sum.z + color.z);
          }
          ++paths;
          ++sample;
          const bool more = segd ? ((sample & segMask) != 0u && sample < spp) : (sample < spp);
          if (more) {
            if (!ex.slotColor) { GF(PG_RGB, s) = sum.x; GF(PG_RGB + 1, s) = sum.y; GF(PG_RGB + 2, s) = sum.z; }
            SU(PW_RNG, s) = (uint32_t)rng.s0; SU(PW_RNG + 1, s) = (uint32_t)(rng.s0 >> 32);
            SU(PW_RNG + 2, s) = (uint32_t)rng.s1; SU(PW_RNG + 3, s) = (uint32_t)(rng.s1 >> 32);
            SU(PW_CNT, s) = sample;
            toGen = true;
          } else if (segd && sample < spp) {
            // a segment other than the last is complete: its partial sum (or its slots) is all it leaves
            if (ex.segPart) {
              float* part = ex.segPart + 3 * ((size_t)(((sample - 1u) >> segShift) - ex.segBase) * n + pixNow);
              part[0] = sum.x; part[1] = sum.y; part[2] = sum.z;
            }
            toFetch = true;
          } else {
            // pixel complete: rgb sum + the LAST sample's hit record
            if (ex.segPart) {
              float* part = ex.segPart + 3 * ((size_t)(((sample - 1u) >> segShift) - ex.segBase) * n + pixNow);
              part[0] = sum.x; part[1] = sum.y; part[2] = sum.z;
            } else if (!ex.slotColor) res->rgb = {sum.x, sum.y, sum.z};
            uint32_t oPrim = MI_INVALID_PRIM, oGeom = MI_INVALID_GEOM;
            if (lastLeaf != 0xFFFFFFFFu) { const GLeaf LL = sc.leaves[lastLeaf]; oPrim = LL.primID; oGeom = leaf_geom(LL); }
            mi_hit_record hr;
            hr.r.origin = {so.x, so.y, so.z}; hr.r.t_min = 0.f;
            hr.r.direction = {sd.x, sd.y, sd.z}; hr.r.t_max = gotHit ? h2.t : kInf;
            hr.prim_id = oPrim;
            hr.normal = {nrm.x, nrm.y, nrm.z};
            hr.throughput = {tp.x, tp.y, tp.z};
            hr.geom_id = (uint16_t)oGeom; hr.flags = (uint16_t)oFlags;
            res->h = hr;
            toFetch = true;
          }
        } else {
          // next bounce: offsetRay + cast set-up (codelets :207-211)
          if (rgbNaN) { GF(PG_RGB, s) = sum.x; GF(PG_RGB + 1, s) = sum.y; GF(PG_RGB + 2, s) = sum.z; }
          if (lastSample) { GF(PG_NRM, s) = nrm.x; GF(PG_NRM + 1, s) = nrm.y; GF(PG_NRM + 2, s) = nrm.z; GU(PG_LEAF, s) = lastLeaf; }
          so = offset_origin(so, sd, nrm);
          const f3 si = mk(1.f / sd.x, 1.f / sd.y, 1.f / sd.z);
          const bool ex2 = !(fabsf(si.x) < kInf && fabsf(si.y) < kInf && fabsf(si.z) < kInf && fabsf(so.x) < kInf && fabsf(so.y) < kInf && fabsf(so.z) < kInf);
          const Shear s2 = make_shear(sd, si);
          SF(PW_O, s) = so.x; SF(PW_O + 1, s) = so.y; SF(PW_O + 2, s) = so.z;
          SF(PW_D, s) = sd.x; SF(PW_D + 1, s) = sd.y; SF(PW_D + 2, s) = sd.z;
          SF(PW_INV, s) = si.x; SF(PW_INV + 1, s) = si.y; SF(PW_INV + 2, s) = si.z;
          SF(PW_SX, s) = s2.sx; SF(PW_SY, s) = s2.sy;
          SU(PW_RNG, s) = (uint32_t)rng.s0; SU(PW_RNG + 1, s) = (uint32_t)(rng.s0 >> 32);
          SU(PW_RNG + 2, s) = (uint32_t)rng.s1; SU(PW_RNG + 3, s) = (uint32_t)(rng.s1 >> 32);
          SF(PW_COLOR, s) = color.x; SF(PW_COLOR + 1, s) = color.y; SF(PW_COLOR + 2, s) = color.z;
          SF(PW_TP, s) = tp.x; SF(PW_TP + 1, s) = tp.y; SF(PW_TP + 2, s) = tp.z;
          SU(PW_CNT, s) = sample | (bounce << 19) | (oFlags << 27) | (s2.kz << 29) | (ex2 ? 0x80000000u : 0u);
          ++casts;
          toReady = true;
        }
      }
      if (ex.slotColor) {
        const unsigned long long mE = __ballot(envRay);
        if (mE) {
          const uint32_t firstE = (uint32_t)__ffsll((long long)mE) - 1u;
          uint32_t baseE = 0;
          if (lane == firstE) baseE = atomicAdd(ex.count, (uint32_t)__popcll(mE));
          baseE = __shfl(baseE, firstE);
          if (envRay) {
            // PreProcessEscapedRays (codelets/TraceCodelets.cpp:321-358), same arithmetic as escaped_uv_kernel
            const float twoPi = (float)(2.0 * 3.14159265358979323846264338327950288);
            const float invPi = (float)(1.0 / 3.14159265358979323846264338327950288);
            const float inv2Pi = (float)(1.0 / (2.0 * 3.14159265358979323846264338327950288));
            const float theta = acosf(envDir.y);
            float phi = atan2f(envDir.z, envDir.x) + ex.azimuthRotation;
            if (phi < 0.f) phi += twoPi;
            else if (phi > twoPi) phi -= twoPi;
            ex.u[envSlot] = theta * invPi;
            ex.v[envSlot] = phi * inv2Pi;
            ex.slotTp[3 * (size_t)envSlot] = envTp.x; ex.slotTp[3 * (size_t)envSlot + 1] = envTp.y; ex.slotTp[3 * (size_t)envSlot + 2] = envTp.z;
            ex.index[baseE + lanesBelow(mE)] = envSlot;
          }
        }
      }
      ringPush(RING_READY, toReady, s);
      ringPush(RING_GEN, toGen, s);
      if (vctl[8] == 0u) ringPush(RING_FETCH, toFetch, s);      // (after the end of the launch a finished slot simply dies)
      if (STATS) tShade += __builtin_amdgcn_s_memtime() - tq1;
    } else {
      // ---------------- GEN: camera rays of up to 64 next samples (codelets/TraceCodelets.cpp:142-164) ----------------
      const unsigned long long tq2 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
      uint32_t gbase;
      const uint32_t k = ringClaim(RING_GEN, 64u, gbase);
      if (k == 0) { if (STATS) itFail++; continue; }
      if (STATS) { itG++; lnG += k; }
      const bool mine = lane < k;
      const uint32_t s = mine ? ringTake(RING_GEN, gbase + lane) : 0u;
      acquire();
      if (mine) {
        Rng rng;
        rng.s0 = (uint64_t)SU(PW_RNG, s) | ((uint64_t)SU(PW_RNG + 1, s) << 32); rng.s1 = (uint64_t)SU(PW_RNG + 2, s) | ((uint64_t)SU(PW_RNG + 3, s) << 32);
        const uint32_t sample = SU(PW_CNT, s) & kPoolMaxSamples;
        const float prow = GF(PG_ROW, s), pcol = GF(PG_COL, s);
        float g0, g1;
        rng_gauss2(rng, sinTbl, g0, g1);
        const float jr = prow + sc.antiAliasScale * g0, jc = pcol + sc.antiAliasScale * g1;
        const f3 sd = pixel_to_ray_dir(jc, jr, sc.imageWidth, sc.imageHeight, sc.tanTheta);
        const f3 so = offset_origin(mk(0.f, 0.f, 0.f), sd, mk(0.f, 0.f, 1.f));
        const f3 si = mk(1.f / sd.x, 1.f / sd.y, 1.f / sd.z);
        const bool ex2 = !(fabsf(si.x) < kInf && fabsf(si.y) < kInf && fabsf(si.z) < kInf && fabsf(so.x) < kInf && fabsf(so.y) < kInf && fabsf(so.z) < kInf);
        const Shear s2 = make_shear(sd, si);
        SF(PW_O, s) = so.x; SF(PW_O + 1, s) = so.y; SF(PW_O + 2, s) = so.z;
        SF(PW_D, s) = sd.x; SF(PW_D + 1, s) = sd.y; SF(PW_D + 2, s) = sd.z;
        SF(PW_INV, s) = si.x; SF(PW_INV + 1, s) = si.y; SF(PW_INV + 2, s) = si.z;
        SF(PW_SX, s) = s2.sx; SF(PW_SY, s) = s2.sy;
        SU(PW_RNG, s) = (uint32_t)rng.s0; SU(PW_RNG + 1, s) = (uint32_t)(rng.s0 >> 32);
        SU(PW_RNG + 2, s) = (uint32_t)rng.s1; SU(PW_RNG + 3, s) = (uint32_t)(rng.s1 >> 32);
        SF(PW_COLOR, s) = 0.f; SF(PW_COLOR + 1, s) = 0.f; SF(PW_COLOR + 2, s) = 0.f;
        SF(PW_TP, s) = 1.f; SF(PW_TP + 1, s) = 1.f; SF(PW_TP + 2, s) = 1.f;
        SU(PW_CNT, s) = sample | (s2.kz << 29) | (ex2 ? 0x80000000u : 0u);             // bounce 0, flags 0
        ++casts;
      }
      ringPush(RING_READY, mine, s);
      if (STATS) tGen += __builtin_amdgcn_s_memtime() - tq2;
    }
  }
  flush_stats(sc, casts, cs, paths);
  if (STATS && lane == 0) {
    atomicAdd(&sc.counters[4], (unsigned long long)itN); atomicAdd(&sc.counters[5], (unsigned long long)lnN);
    atomicAdd(&sc.counters[6], (unsigned long long)itL); atomicAdd(&sc.counters[7], (unsigned long long)lnL);
    atomicAdd(&sc.counters[8], (unsigned long long)itS); atomicAdd(&sc.counters[9], (unsigned long long)lnS);
    atomicAdd(&sc.counters[10], (unsigned long long)itG); atomicAdd(&sc.counters[11], (unsigned long long)lnG);
    atomicAdd(&sc.counters[12], tTrav); atomicAdd(&sc.counters[13], tShade); atomicAdd(&sc.counters[14], tGen);
    atomicAdd(&sc.counters[15], __builtin_amdgcn_s_memtime() - tLoop0);
    atomicAdd(&sc.counters[16], (unsigned long long)itLoop); atomicAdd(&sc.counters[17], (unsigned long long)itRefill);
    atomicAdd(&sc.counters[18], (unsigned long long)lnRefill); atomicAdd(&sc.counters[19], (unsigned long long)itIdle);
    atomicAdd(&sc.counters[20], (unsigned long long)itFail); atomicAdd(&sc.counters[21], (unsigned long long)itBurst);
    atomicAdd(&sc.counters[22], (unsigned long long)lnBurst); atomicAdd(&sc.counters[23], tRefill);
  }
}

}  // namespace mi
