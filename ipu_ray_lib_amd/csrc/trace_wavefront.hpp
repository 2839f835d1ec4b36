// trace_wavefront.hpp — K1w: the path-trace hot loop as a persistent, phase-scheduled kernel.
//
// Why (profiles/r01_v0_*: the one-thread-one-pixel nested-loop kernel keeps 21 % of the VALU lanes
// busy): a path tracer's lanes diverge three ways — traversal length, leaf tests, path length. Here
// every lane is a small state machine that is always in exactly one PHASE
//     NODE  one box test of the stackless BVH walk          (CompactBvh.hpp:103-135, CompactBVH2Node.cpp:5-22)
//     LEAF  one primitive test at a leaf whose box was hit   (Mesh.cpp:6-104, Primitives.cpp:24-67)
//     SHADE traversal finished: hit update, BxDF, roulette   (codelets/TraceCodelets.cpp:214-257)
//     GEN   next sample's camera ray                          (codelets/TraceCodelets.cpp:142-164)
//     FETCH pixel finished: write it back, take another one from the global work counter
// and each loop iteration the wave votes (ballot + popcount, scalar) and runs the phase most of its
// lanes are waiting in. A lane that finishes a path immediately starts its next sample, a lane that
// finishes a pixel pulls a new one, so no lane idles until the frame runs out of pixels. Each lane
// still performs exactly the reference's sequence of operations for its rays, in the reference's
// order, with the same arithmetic — only the interleaving between lanes changes — so results stay
// bit-identical to the nested-loop kernel and to the CPU oracle.
#pragma once

#include "trace_kernels.hpp"

namespace mi {

enum : uint32_t { PH_NODE = 0, PH_LEAF = 1, PH_SHADE = 2, PH_GEN = 3, PH_FETCH = 4, PH_DONE = 5 };

// Scheduling thresholds: a waiting phase runs as soon as this many lanes are parked in it; below the
// thresholds NODE runs while it has any lane, and when nothing traverses the fullest phase runs.
// Derivation of the defaults (a batch/occupancy trade-off under the 64-lane budget) is in DESIGN.md §6.
struct WaveTune { uint32_t leafAt, shadeAt, genAt, burst, keep8; };

template <bool STATS, bool LDS_NODES, int BLOCK>
__global__ void __launch_bounds__(BLOCK) path_trace_wavefront_kernel(DeviceScene sc, mi_trace_result* rays, uint32_t n,
                                                                   uint32_t* workCounter, uint32_t ldsNodeCount, WaveTune tune) {
  __shared__ float sinTbl[92];
  extern __shared__ __attribute__((aligned(16))) unsigned char dynLds[];
  load_sin_table(sinTbl);
  const GNode* ldsNodes = reinterpret_cast<const GNode*>(dynLds);
  if (LDS_NODES) {
    // stage the first ldsNodeCount nodes (preorder prefix) once per workgroup, 8 B per lane per step
    const uint2* src = reinterpret_cast<const uint2*>(sc.nodes);
    uint2* dst = reinterpret_cast<uint2*>(dynLds);
    for (uint32_t k = threadIdx.x; k < ldsNodeCount * 3; k += blockDim.x) dst[k] = src[k];
    __syncthreads();
  }

  const uint32_t lane = threadIdx.x & 63;
  const uint32_t numNodes = sc.numNodes;
  const uint32_t spp = sc.samplesPerPixel;

  // ---- lane state ----
  uint32_t ph = PH_FETCH;
  uint32_t pix = 0, sample = 0, bounce = 0, node = 0, pendLeaf = 0;
  float prow = 0.f, pcol = 0.f;
  Rng rng; rng.s0 = rng.s1 = 0;
  f3 rgb = mk(0, 0, 0), color = mk(0, 0, 0), tp = mk(1, 1, 1);
  f3 o = mk(0, 0, 0), d = mk(0, 0, -1), nrm = mk(0, 0, 1), inv = mk(0, 0, 0);
  Shear sh; sh.kz = 2; sh.sx = sh.sy = 0.f; sh.sz = 1.f;
  Hit hit; hit.t = kInf; hit.leaf = 0xFFFFFFFFu; hit.geomID = 0xFFFFu; hit.b0 = hit.b1 = hit.b2 = 0.f;
  uint32_t oFlags = 0, oPrim = MI_INVALID_PRIM, oGeom = MI_INVALID_GEOM;
  bool exactSlab = false;
  float oTmax = kInf;
  CastStats cs = {0, 0};
  uint32_t casts = 0, paths = 0;
  // STATS only: per-wave phase executions and the lanes that were active in them (wave-uniform values)
  uint32_t itN = 0, itL = 0, itS = 0, itG = 0, lnN = 0, lnL = 0, lnS = 0, lnG = 0;

  for (;;) {
    // ---------------- FETCH: cheap, always served first ----------------
    const unsigned long long mF = __ballot(ph == PH_FETCH);
    if (mF) {
      uint32_t base = 0;
      if (lane == (uint32_t)__ffsll((long long)mF) - 1) base = atomicAdd(workCounter, (uint32_t)__popcll(mF));
      base = __shfl(base, __ffsll((long long)mF) - 1);
      if (ph == PH_FETCH) {
        const uint32_t idx = base + (uint32_t)__popcll(mF & ((1ull << lane) - 1ull));
        if (idx < n) {
          pix = idx;
          const mi_trace_result* res = rays + idx;
          prow = res->u; pcol = res->v;
          rgb = mk(res->rgb.x, res->rgb.y, res->rgb.z);
          rng_seed_pixel(rng, sc.rngSeed, prow, pcol);
          sample = 0;
          ph = PH_GEN;
        } else {
          ph = PH_DONE;
        }
      }
    }

    // ---------------- vote ----------------
    const uint32_t cN = (uint32_t)__popcll(__ballot(ph == PH_NODE));
    const uint32_t cL = (uint32_t)__popcll(__ballot(ph == PH_LEAF));
    const uint32_t cS = (uint32_t)__popcll(__ballot(ph == PH_SHADE));
    const uint32_t cG = (uint32_t)__popcll(__ballot(ph == PH_GEN));
    if ((cN | cL | cS | cG) == 0) break;            // every lane DONE (FETCH lanes were just served)
    // 0 = NODE, 1 = LEAF, 2 = SHADE, 3 = GEN
    // weighted populations (quarter units): the phase with the largest weighted population runs
    uint32_t run;
    {
      const uint32_t wN = cN * 4u, wL = cL * tune.leafAt, wS = cS * tune.shadeAt, wG = cG * tune.genAt;
      const uint32_t other = max(wL, max(wS, wG));
      if (cN > 0 && wN >= other) run = 0;
      else run = (wL >= wS && wL >= wG) ? 1 : (wS >= wG ? 2 : 3);
    }

    if (run == 0) {
      // ---------------- NODE: one box test per lane ----------------
      // NODE steps run in a short burst: re-voting costs about as much as a box test, so the wave keeps
      // stepping while at least 3/4 of the lanes that started the burst are still traversing (<= 4 steps).
      uint32_t stay = cN, burst = 0;
      do {
        if (STATS) { itN++; lnN += stay; }
        if (ph == PH_NODE) {
          GNode nd;
          if (LDS_NODES && node < ldsNodeCount) nd = ldsNodes[node];
          else nd = sc.nodes[node];
          if (STATS) cs.nodes++;
          // Box test (CompactBVH2Node.cpp:5-22, intersectRaySlab CompactBVH2Node.hpp:14-50).
          // Fast form: with finite origin and finite inverse direction no slab product can be NaN, and for
          // non-NaN values the reference's ordered compare/selects ARE min/max: swap(tmin,tmax) = (min,max),
          // "t0 = tmin > t0 ? tmin : t0" = max, "t1 = tmax < t1 ? tmax : t1" = min, in any axis order; the
          // sign of a zero never reaches the result (only t0 > t1 is used). Lanes whose ray has a zero /
          // denormal direction component or a non-finite origin (exactSlab) redo the test with the
          // reference's literal compare/select sequence below, so NaN cases stay bit-identical too.
          const float maxx = nd.minx + half_bits_to_float(nd.hx);
          const float maxy = nd.miny + half_bits_to_float(nd.hy);
          const float maxz = nd.minz + half_bits_to_float(nd.hz);
          const float ax = (nd.minx - o.x) * inv.x, bx = (maxx - o.x) * inv.x;
          const float ay = (nd.miny - o.y) * inv.y, by = (maxy - o.y) * inv.y;
          const float az = (nd.minz - o.z) * inv.z, bz = (maxz - o.z) * inv.z;
          float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.f));
          float t1 = fminf(fminf(fmaxf(ax, bx) * kSlabScale, fmaxf(ay, by) * kSlabScale), fminf(fmaxf(az, bz) * kSlabScale, hit.t));
          if (__ballot(exactSlab)) {
            if (exactSlab) {
              t0 = 0.f; t1 = hit.t;
              { float tmin = ax, tmax = bx; if (tmin > tmax) { const float s = tmin; tmin = tmax; tmax = s; } tmax *= kSlabScale; t0 = tmin > t0 ? tmin : t0; t1 = tmax < t1 ? tmax : t1; }
              { float tmin = ay, tmax = by; if (tmin > tmax) { const float s = tmin; tmin = tmax; tmax = s; } tmax *= kSlabScale; t0 = tmin > t0 ? tmin : t0; t1 = tmax < t1 ? tmax : t1; }
              { float tmin = az, tmax = bz; if (tmin > tmax) { const float s = tmin; tmin = tmax; tmax = s; } tmax *= kSlabScale; t0 = tmin > t0 ? tmin : t0; t1 = tmax < t1 ? tmax : t1; }
            }
          }
          const bool boxHit = !(t0 > t1);
          const bool isLeaf = nd.geomID != 0xFFFFu;
          if (boxHit && isLeaf) {
            pendLeaf = nd.link;
            ph = PH_LEAF;
          } else {
            node = (boxHit || isLeaf) ? node + 1 : nd.link;
            if (node >= numNodes) ph = PH_SHADE;
          }
        }
        stay = (uint32_t)__popcll(__ballot(ph == PH_NODE));
      } while (++burst < tune.burst && stay * 8u >= cN * tune.keep8 && stay > 0);
    } else if (run == 1) {
      // ---------------- LEAF: one primitive test per lane ----------------
      if (STATS) { itL++; lnL += cL; }
      if (ph == PH_LEAF) {
        if (STATS) cs.leaves++;
        const GLeaf L = sc.leaves[pendLeaf];
        float t, b0 = 0.f, b1 = 0.f, b2 = 0.f;
        bool cand;
        const uint32_t kind = leaf_kind(L);
        if (kind == LEAF_TRI) {
          t = intersect_triangle(mk(L.f[0], L.f[1], L.f[2]), mk(L.f[3], L.f[4], L.f[5]), mk(L.f[6], L.f[7], L.f[8]), o, sh, b0, b1, b2);
          cand = t > 0.f && t < kInf;
        } else if (kind == LEAF_SPHERE) {
          t = intersect_sphere(L, o, d, 0.f);
          cand = true;
        } else {
          t = intersect_disc(L, o, d);
          cand = true;
        }
        if (cand && t > 0.f && t < hit.t) { hit.t = t; hit.leaf = pendLeaf; hit.b0 = b0; hit.b1 = b1; hit.b2 = b2; }
        node = node + 1;
        ph = (node >= numNodes) ? PH_SHADE : PH_NODE;
      }
    } else if (run == 2) {
      // ---------------- SHADE: traversal of bounce `bounce` is complete ----------------
      if (STATS) { itS++; lnS += cS; }
      if (ph == PH_SHADE) {
        bool terminated = false;
        if (hit.leaf != 0xFFFFFFFFu) {
          const GLeaf L = sc.leaves[hit.leaf];
          hit.geomID = leaf_geom(L);
          oGeom = hit.geomID; oPrim = L.primID; oTmax = hit.t;
          o = o + d * hit.t;                                          // updateHit, Render.hpp:15-23
          nrm = hit_normal(sc, hit, o);
          const mi_material mat = sc.materials[sc.matIDs[hit.geomID]];
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
            rgb = rgb * __builtin_nanf("");
            oFlags |= MI_FLAG_ERROR;
          }
        } else {
          oTmax = kInf;
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
          rgb = rgb + color;
          ++paths;
          ++sample;
          if (sample < spp) ph = PH_GEN;
          else {
            // pixel complete: rgb sum + the LAST sample's hit record
            mi_trace_result* res = rays + pix;
            res->rgb = {rgb.x, rgb.y, rgb.z};
            mi_hit_record hr;
            hr.r.origin = {o.x, o.y, o.z}; hr.r.t_min = 0.f;
            hr.r.direction = {d.x, d.y, d.z}; hr.r.t_max = oTmax;
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
          inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
          exactSlab = !(fabsf(inv.x) < kInf && fabsf(inv.y) < kInf && fabsf(inv.z) < kInf && fabsf(o.x) < kInf && fabsf(o.y) < kInf && fabsf(o.z) < kInf);
          sh = make_shear(d);
          hit.t = kInf; hit.leaf = 0xFFFFFFFFu;
          node = 0;
          ++casts;
          ph = (numNodes > 0) ? PH_NODE : PH_SHADE;
        }
      }
    } else {
      // ---------------- GEN: camera ray of the next sample ----------------
      if (STATS) { itG++; lnG += cG; }
      if (ph == PH_GEN) {
        float g0, g1;
        rng_gauss2(rng, sinTbl, g0, g1);
        const float jr = prow + sc.antiAliasScale * g0, jc = pcol + sc.antiAliasScale * g1;
        d = pixel_to_ray_dir(jc, jr, sc.imageWidth, sc.imageHeight, sc.tanTheta);
        o = mk(0.f, 0.f, 0.f);
        nrm = mk(0.f, 0.f, 1.f);                         // HitRecord ctor, geometry.hpp:236-242
        oPrim = MI_INVALID_PRIM; oGeom = MI_INVALID_GEOM; oFlags = 0; oTmax = kInf;
        tp = mk(1.f, 1.f, 1.f);
        color = mk(0.f, 0.f, 0.f);
        bounce = 0;
        o = offset_origin(o, d, nrm);
        inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
        exactSlab = !(fabsf(inv.x) < kInf && fabsf(inv.y) < kInf && fabsf(inv.z) < kInf && fabsf(o.x) < kInf && fabsf(o.y) < kInf && fabsf(o.z) < kInf);
        sh = make_shear(d);
        hit.t = kInf; hit.leaf = 0xFFFFFFFFu;
        node = 0;
        ++casts;
        ph = (numNodes > 0) ? PH_NODE : PH_SHADE;
      }
    }
  }
  flush_stats(sc, casts, cs, paths);
  if (STATS && lane == 0) {
    atomicAdd(&sc.counters[4], (unsigned long long)itN); atomicAdd(&sc.counters[5], (unsigned long long)lnN);
    atomicAdd(&sc.counters[6], (unsigned long long)itL); atomicAdd(&sc.counters[7], (unsigned long long)lnL);
    atomicAdd(&sc.counters[8], (unsigned long long)itS); atomicAdd(&sc.counters[9], (unsigned long long)lnS);
    atomicAdd(&sc.counters[10], (unsigned long long)itG); atomicAdd(&sc.counters[11], (unsigned long long)lnG);
  }
}

}  // namespace mi
