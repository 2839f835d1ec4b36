// trace_kernels.hpp — gfx950 kernels for the ray-parallel hot path:
//   K2 shadow_trace_kernel  == ShadowTrace vertex  (codelets/TraceCodelets.cpp:269-316, Render.hpp:37-72)
//   K1 path_trace_kernel    == PathTrace vertex    (codelets/TraceCodelets.cpp:170-264 == trace.cpp:115-188)
//
// Traversal design. The reference walks the depth-first CompactBVH2Node array with a
// per-ray stack, always first child first (CompactBvh.hpp:80-139). Because the array is in
// preorder, "pop the next entry" is always the node that follows the current node's subtree, so the
// same visit order is produced by a STACKLESS walk: box hit on an interior node -> its first child,
// anything else -> next(i), the node that follows i's subtree in preorder. The upload step gives every
// device node that `next` explicitly (interior nodes: instead of secondChildIndex; leaves: beside their
// primitive's index); no per-wavefront stack, no LDS or scratch traffic for it, and every lane's
// sequence of box tests, primitive tests and closest-hit updates is exactly the reference's.
#pragma once

#include <hip/hip_runtime.h>

#include "ray_math.h"
#include "../../include/mi_raylib.h"

namespace mi {

// ---- device scene ---------------------------------------------------------------------------------
// 32 B device node = two aligned 16-byte loads. The box is stored as the six floats the reference's box
// test derives from a CompactBVH2Node on every visit: min and max = min + (float)extent
// (CompactBVH2Node.cpp:8-10; the binary16 -> binary32 conversion is exact and the add is one rounded
// binary32 operation, so doing it once at upload gives the same bits). Interleaved (min,max) per axis so
// a 64-bit register pair feeds the packed subtract/multiply directly.
struct __attribute__((aligned(16))) GNode {
  float minx, maxx, miny, maxy, minz, maxz;
  // Both successors are BYTE offsets into the node array (index << 5), ready to be added to the array's base:
  uint32_t link;                                  // where the walk goes on when the box is MISSED: next(i), the node after i's subtree in preorder (numNodes << 5 = the walk ends)
  uint32_t hit;                                   // where it goes on when the box is HIT: interior node: i + 1 (its first child); leaf: link | kLeafFlag -
                                                  // "stop: the primitive of the node BEFORE `link` is to be tested" (a leaf's link is i + 1)
};
// Both successors precomputed: a box test ends in ONE select (hit ? nd.hit : nd.link), and one unsigned compare with
// numNodes tells whether the lane walks on - a flagged value is >= 2^31 > numNodes, so "stopped at a primitive" and
// "walked off the end" both read as "stop" and are told apart once, after a run of box tests, from the flag.
constexpr uint32_t kLeafFlag = 0x80000000u;
__host__ __device__ __forceinline__ bool node_is_leaf(const GNode& nd) { return (nd.hit & kLeafFlag) != 0u; }
static_assert(sizeof(GNode) == 32, "GNode must stay 32 bytes");

enum : uint32_t { LEAF_TRI = 0, LEAF_SPHERE = 1, LEAF_DISC = 2 };

struct __attribute__((aligned(16))) GLeaf {       // 64 B pre-resolved primitive (the primitive test reads the first 40); leaves[i] belongs to leaf NODE i
  uint32_t type;     // LEAF_* in bits 0..15, geomID in bits 16..31. FIRST, so that it arrives with the first 16-byte load of
                     // the record (the test branches on it) and a triangle is three loads (16 + 16 + 8 bytes), not four
  float f[9];        // tri: p0,p1,p2 | sphere: cx,cy,cz,radius,radius2 | disc: nx,ny,nz,cx,cy,cz,r2
  uint32_t primID;   // value reported in the hit record
  uint32_t triBase;  // tri: index of the triangle's first u16 in meshTris (for vertex normals)
  float n[3];        // tri: the face normal normalise(cross(p1-p0, p2-p0)) (Mesh.hpp:112-114), evaluated once at
                     // upload with the same binary32 operations the reference performs per hit (no contraction,
                     // correctly rounded sqrt and divide on both sides), so SHADE just reads it
  uint32_t matIndex; // matIDs[geomID], resolved at upload (one dependent load less when a hit is shaded)
};
static_assert(sizeof(GLeaf) == 64, "GLeaf must stay 64 bytes");
__host__ __device__ __forceinline__ uint32_t leaf_kind(const GLeaf& L) { return L.type & 0xFFFFu; }
__host__ __device__ __forceinline__ uint32_t leaf_geom(const GLeaf& L) { return L.type >> 16; }

// Round 5 (scene option "leaf_rot", default on: profiles/r05_k1w_leaf_ab.txt): the primitive-test part of a leaf record once per
// shear axis. Block kz = {type, nine floats}: a triangle's vertices with their components rotated so that component kz
// comes last - (p - o) permuted equals permuted p - permuted o component for component, so the test's 18 selects on the vertices
// become 6 on the origin -, a sphere's or disc's floats as they are. 128 bytes per node (indexed like leaves[]).
struct __attribute__((aligned(8))) GLeafBlock { uint32_t type; float f[9]; };
struct __attribute__((aligned(128))) GLeafRot { GLeafBlock b[3]; uint32_t pad[2]; };
static_assert(sizeof(GLeafBlock) == 40 && sizeof(GLeafRot) == 128, "GLeafRot: three 40-byte blocks in a 128-byte line");

struct DeviceScene {
  const GNode* nodes;        uint32_t numNodes;
  const GLeaf* leaves;       uint32_t numLeaves;
  const GLeafRot* leavesRot;     // the same records' primitive-test part, pre-rotated per shear axis
  const uint32_t* matIDs;    // per geomID
  const mi_material* materials; uint32_t numMaterials;
  // vertex normals (only when the scene was loaded with normals)
  const uint16_t* meshTris; const mi_vec3* meshNormals; const uint32_t* geomFirstVertex; uint32_t hasNormals;
  const float* leafNormals;  // [numLeaves][9]: a triangle leaf's three vertex normals, gathered at upload (zeros for other leaves)
  float imageWidth, imageHeight, tanTheta, antiAliasScale;
  uint32_t maxPathLength, rouletteStartDepth, samplesPerPixel;
  uint64_t rngSeed;
  unsigned long long* counters;   // [casts, nodes visited, leaf tests, paths]
  // The root's box, when the root is an interior node (a BVH of more than one node): a ray that starts strictly inside
  // it hits it, so the walk can start at node 1 (root_start below). rootInterior = 0 switches the shortcut off.
  float rootLoX, rootLoY, rootLoZ, rootHiX, rootHiY, rootHiZ;
  uint32_t rootInterior;
};

// First node of a closest-hit walk with tMin = 0 and tMax = +inf (every cast of the path-trace loop). The reference
// starts at node 0 (CompactBvh.hpp:95-101) and tests the root's box. For an origin STRICTLY inside that box the test's
// outcome is known without evaluating it: per axis (min - o) < 0 < (max - o) exactly (the difference of two distinct
// binary32 numbers never rounds to zero), so whatever the reciprocal direction is - finite, denormal, zero, infinite or
// NaN - the slab gives tmin <= 0 <= tmax or leaves t0 / t1 untouched (CompactBVH2Node.hpp:14-50: every update is an
// ordered compare, false for NaN), hence t0 = 0 <= t1 and the box is hit; the root of a BVH with more than one node is an
// interior node, so the reference's next visit is node 1. Returns that node, and the number of box tests it stands for.
__device__ __forceinline__ uint32_t root_start(const DeviceScene& sc, f3 o, uint32_t& visited) {
  const bool inside = (o.x > sc.rootLoX) & (o.x < sc.rootHiX) & (o.y > sc.rootLoY) & (o.y < sc.rootHiY) & (o.z > sc.rootLoZ) & (o.z < sc.rootHiZ) & (sc.rootInterior != 0u);
  visited = inside ? 1u : 0u;
  return inside ? 1u : 0u;
}

struct Shear { uint32_t kz; float sx, sy, sz; };

// Primitives.cpp:5-22. Pure function of the ray, so it is evaluated once per cast instead of once
// per leaf (Mesh.hpp:89); identical values.
__device__ __forceinline__ Shear make_shear(f3 d) {
  Shear s;
  s.kz = min_index(d);
  uint32_t kx = s.kz + 1; if (kx == 3) kx = 0;
  uint32_t ky = kx + 1; if (ky == 3) ky = 0;
  const float dx = comp(d, kx), dy = comp(d, ky), dz = comp(d, s.kz);
  s.sx = -dx / dz;
  s.sy = -dy / dz;
  s.sz = 1.f / dz;
  return s;
}
// The same with the ray's reciprocal direction at hand: 1 / d[kz] is one of its components (the same IEEE division).
__device__ __forceinline__ Shear make_shear(f3 d, f3 inv) {
  Shear s;
  s.kz = min_index(d);
  uint32_t kx = s.kz + 1; if (kx == 3) kx = 0;
  uint32_t ky = kx + 1; if (ky == 3) ky = 0;
  const float dx = comp(d, kx), dy = comp(d, ky), dz = comp(d, s.kz);
  s.sx = -dx / dz;
  s.sy = -dy / dz;
  s.sz = comp(inv, s.kz);
  return s;
}

// The tolerance tier's cast set-up: 1/d by v_rcp_f32 (1 ulp) and the shear as products with it instead of two more
// correctly rounded divisions (five division sequences of ~10 instructions per cast become three instructions).
__device__ __forceinline__ f3 fast_inverse(f3 d) { return mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z)); }
__device__ __forceinline__ Shear make_shear_fast(f3 d, f3 inv) {
  Shear s;
  s.kz = min_index(d);
  uint32_t kx = s.kz + 1; if (kx == 3) kx = 0;
  uint32_t ky = kx + 1; if (ky == 3) ky = 0;
  s.sz = comp(inv, s.kz);
  s.sx = -comp(d, kx) * s.sz;
  s.sy = -comp(d, ky) * s.sz;
  return s;
}

__device__ __forceinline__ f3 permute_kz(f3 p, uint32_t kz) {
  // (kx,ky,kz) is the cyclic rotation that puts component kz last. Written as selects: as three early returns
  // hipcc built divergent regions (exec-mask juggling, ~50 scalar instructions in dependent chains per
  // primitive test) out of what is six v_cndmask per vector.
  const bool r1 = kz == 0;   // (1, 2, 0)
  const bool r2 = kz == 1;   // (2, 0, 1)
  return mk(r1 ? p.y : (r2 ? p.z : p.x), r1 ? p.z : (r2 ? p.x : p.y), r1 ? p.x : (r2 ? p.y : p.z));
}

// Mesh.cpp:6-104, tFar = inf as passed by Mesh.hpp:92. Returns t (0 = miss). DF = the reference's compile-time variant
// ALLOW_DOUBLE_FALLBACK=1 (CMakeLists.txt:13,34-41; Mesh.cpp:38-51): when an edge function is exactly zero all three are
// recomputed from products and differences in binary64 and narrowed to binary32 (v_mul_f64 / v_add_f64 / v_cvt_f32_f64
// are IEEE operations: the same bits as the host's doubles). DF = false is the reference default.
// Written without the reference's early returns: every lane runs every operation and the verdicts are combined at the
// end. Identical values for every input, NaNs included: each early return of Mesh.cpp is a predicate that is evaluated
// here on the same operands (a lane that would have returned early computes on - with whatever its operands give, an
// infinite or NaN 1/det included - and is then told 0). With tFar = +inf (Mesh.hpp:90-92) tFar*det is -inf (det<0) or
// +inf (det>0), and "tScaled < -inf" / "tScaled > +inf" are false for every float including NaN, so those two
// comparisons of Mesh.cpp:62-66 drop out exactly. Measured against the early-return form: -1.4 % frame time (hipcc
// turns early returns into nested exec regions whose merges cost 45 register moves per test).
// PRE: p0, p1, p2 and o arrive with their components already rotated for sh.kz (GLeafRot)
template <bool DF = false, bool PRE = false>
__device__ __forceinline__ float intersect_triangle(f3 p0, f3 p1, f3 p2, f3 o, const Shear& sh, float& b0, float& b1, float& b2) {
  f3 p0t = PRE ? p0 - o : permute_kz(p0 - o, sh.kz), p1t = PRE ? p1 - o : permute_kz(p1 - o, sh.kz), p2t = PRE ? p2 - o : permute_kz(p2 - o, sh.kz);
  p0t.x += sh.sx * p0t.z; p0t.y += sh.sy * p0t.z;
  p1t.x += sh.sx * p1t.z; p1t.y += sh.sy * p1t.z;
  p2t.x += sh.sx * p2t.z; p2t.y += sh.sy * p2t.z;
  float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
  float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
  float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
  if constexpr (DF) {
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {             // Mesh.cpp:40-50, operation for operation
      const double p2txp1ty = (double)p2t.x * (double)p1t.y;
      const double p2typ1tx = (double)p2t.y * (double)p1t.x;
      e0 = (float)(p2typ1tx - p2txp1ty);
      const double p0txp2ty = (double)p0t.x * (double)p2t.y;
      const double p0typ2tx = (double)p0t.y * (double)p2t.x;
      e1 = (float)(p0typ2tx - p0txp2ty);
      const double p1txp0ty = (double)p1t.x * (double)p0t.y;
      const double p1typ0tx = (double)p1t.y * (double)p0t.x;
      e2 = (float)(p1typ0tx - p1txp0ty);
    }
  }
  // (bitwise, not short-circuit: with || hipcc rebuilds the early returns as nested exec regions; -0.4 % frame time)
#define MI_OR |
#define MI_AND &
  bool miss = ((e0 < 0) MI_OR (e1 < 0) MI_OR (e2 < 0)) MI_AND ((e0 > 0) MI_OR (e1 > 0) MI_OR (e2 > 0));
  const float det = e0 + e1 + e2;
  miss = miss MI_OR (det == 0);
  p0t.z *= sh.sz; p1t.z *= sh.sz; p2t.z *= sh.sz;
  const float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
  miss = miss MI_OR ((det < 0.f) MI_AND (tScaled >= 0.f)) MI_OR ((det > 0.f) MI_AND (tScaled <= 0.f));
  const float invDet = 1 / det;
  b0 = e0 * invDet; b1 = e1 * invDet; b2 = e2 * invDet;
  const float t = tScaled * invDet;
  const float maxZt = min_comp(abs3(mk(p0t.z, p1t.z, p2t.z)));
  const float deltaZ = gamma_n(3) * maxZt;
  const float maxXt = min_comp(abs3(mk(p0t.x, p1t.x, p2t.x)));
  const float maxYt = min_comp(abs3(mk(p0t.y, p1t.y, p2t.y)));
  const float deltaX = gamma_n(5) * (maxXt + maxZt);
  const float deltaY = gamma_n(5) * (maxYt + maxZt);
  const float deltaE = 2 * (gamma_n(2) * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
  const float maxE = min_comp(abs3(mk(e0, e1, e2)));
  const float deltaT = 3 * (gamma_n(3) * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * fabsf(invDet);
  miss = miss MI_OR (t <= deltaT);
  return miss ? 0.f : t;
}

// The TOLERANCE tier's triangle test (scene option "fast" = 1; never the default, never the headline): the same test
// with floating-point contraction allowed (the shear, the edge functions, tScaled and the error bound fuse into FMAs)
// and the determinant's reciprocal taken with v_rcp_f32 (1 ulp) instead of the correctly rounded division. t, the
// barycentrics and the accept / reject verdict then differ from the exact tier's by a few ulp at most - stated and
// tested as a tolerance (tests/test_gpu_parity.py, test_fast_tier_*), not as parity.
__device__ __forceinline__ float intersect_triangle_fast(f3 p0, f3 p1, f3 p2, f3 o, const Shear& sh, float& b0, float& b1, float& b2) {
#pragma clang fp contract(fast)
  f3 p0t = permute_kz(p0 - o, sh.kz), p1t = permute_kz(p1 - o, sh.kz), p2t = permute_kz(p2 - o, sh.kz);
  p0t.x = __builtin_fmaf(sh.sx, p0t.z, p0t.x); p0t.y = __builtin_fmaf(sh.sy, p0t.z, p0t.y);
  p1t.x = __builtin_fmaf(sh.sx, p1t.z, p1t.x); p1t.y = __builtin_fmaf(sh.sy, p1t.z, p1t.y);
  p2t.x = __builtin_fmaf(sh.sx, p2t.z, p2t.x); p2t.y = __builtin_fmaf(sh.sy, p2t.z, p2t.y);
  const float e0 = __builtin_fmaf(p1t.x, p2t.y, -(p1t.y * p2t.x));
  const float e1 = __builtin_fmaf(p2t.x, p0t.y, -(p2t.y * p0t.x));
  const float e2 = __builtin_fmaf(p0t.x, p1t.y, -(p0t.y * p1t.x));
  bool miss = ((e0 < 0) MI_OR (e1 < 0) MI_OR (e2 < 0)) MI_AND ((e0 > 0) MI_OR (e1 > 0) MI_OR (e2 > 0));
  const float det = e0 + e1 + e2;
  miss = miss MI_OR (det == 0);
  p0t.z *= sh.sz; p1t.z *= sh.sz; p2t.z *= sh.sz;
  const float tScaled = __builtin_fmaf(e0, p0t.z, __builtin_fmaf(e1, p1t.z, e2 * p2t.z));
  miss = miss MI_OR ((det < 0.f) MI_AND (tScaled >= 0.f)) MI_OR ((det > 0.f) MI_AND (tScaled <= 0.f));
  const float invDet = __builtin_amdgcn_rcpf(det);
  b0 = e0 * invDet; b1 = e1 * invDet; b2 = e2 * invDet;
  const float t = tScaled * invDet;
  const float maxZt = min_comp(abs3(mk(p0t.z, p1t.z, p2t.z)));
  const float deltaZ = gamma_n(3) * maxZt;
  const float maxXt = min_comp(abs3(mk(p0t.x, p1t.x, p2t.x)));
  const float maxYt = min_comp(abs3(mk(p0t.y, p1t.y, p2t.y)));
  const float deltaX = gamma_n(5) * (maxXt + maxZt);
  const float deltaY = gamma_n(5) * (maxYt + maxZt);
  const float deltaE = 2 * __builtin_fmaf(gamma_n(2) * maxXt, maxYt, __builtin_fmaf(deltaY, maxXt, deltaX * maxYt));
  const float maxE = min_comp(abs3(mk(e0, e1, e2)));
  const float deltaT = 3 * __builtin_fmaf(gamma_n(3) * maxE, maxZt, __builtin_fmaf(deltaE, maxZt, deltaZ * maxE)) * fabsf(invDet);
  miss = miss MI_OR (t <= deltaT);
#undef MI_OR
#undef MI_AND
  return miss ? 0.f : t;
}

// Primitives.cpp:24-47
__device__ __forceinline__ float intersect_sphere(const GLeaf& L, f3 o, f3 d, float tMin) {
  const float radius2 = L.f[4];
  const f3 f = mk(L.f[0], L.f[1], L.f[2]) - o;
  const float rd2 = 1.f / sqnorm(d);
  const float tca = dot(f, d) * rd2;
  if (tca < 0.f) return 0.f;
  const f3 l = f - d * tca;
  const float l2 = sqnorm(l);
  if (l2 > radius2) return 0.f;
  const float td = sqrtf(radius2 - l2) * rd2;
  float t0 = tca - td, t1 = tca + td;
  if (t0 > t1) { const float tmp = t0; t0 = t1; t1 = tmp; }
  if (t0 < tMin) {
    t0 = t1;
    if (t0 < tMin) return 0.f;
  }
  return t0;
}

// Primitives.cpp:49-67
__device__ __forceinline__ float intersect_disc(const GLeaf& L, f3 o, f3 d) {
  const f3 n = mk(L.f[0], L.f[1], L.f[2]), c = mk(L.f[3], L.f[4], L.f[5]);
  const float r2 = L.f[6];
  const float angle = dot(n, d);
  if (angle != 0.f) {
    const float dd = fabsf(dot(c, n));
    const float t = -(dot(n, o) + dd) / angle;
    if (t > kMachineEps) {
      const f3 hp = o + d * t;
      const float d2 = sqnorm(hp - c);
      if (d2 < r2) return t;
    }
  }
  return 0.f;
}

struct Hit {
  float t;            // closest t so far (starts at ray.tMax)
  uint32_t leaf;      // 0xFFFFFFFF = none
  uint32_t geomID;
  float b0, b1, b2;   // barycentrics of the closest triangle hit (vertex-normal scenes)
};

struct CastStats { uint32_t nodes, leaves; };

// CompactBvh::intersect (ANY_HIT=false, CompactBvh.hpp:80-139) / ::occluded (ANY_HIT=true, :33-78).
// Box test: CompactBVH2Node.cpp:5-22 + intersectRaySlab (CompactBVH2Node.hpp:14-50). All three
// slabs are evaluated before the single t0>t1 test; since t0 only grows and t1 only shrinks across
// the axes, that is the same predicate as the reference's per-axis early outs.
template <bool ANY_HIT, bool STATS, bool DF = false>
__device__ __forceinline__ bool traverse(const DeviceScene& sc, f3 o, f3 d, float tMin, float tMax, Hit& hit, CastStats& cs) {
  const f3 inv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
  const Shear sh = make_shear(d);
  hit.t = tMax; hit.leaf = 0xFFFFFFFFu; hit.geomID = 0xFFFFu; hit.b0 = hit.b1 = hit.b2 = 0.f;
  const uint32_t numNodes = sc.numNodes;
  uint32_t i = 0;
  while (i < numNodes) {
    const GNode nd = sc.nodes[i];
    if (STATS) cs.nodes++;
    float t0 = tMin, t1 = hit.t;
    {
      float tmin = (nd.minx - o.x) * inv.x, tmax = (nd.maxx - o.x) * inv.x;
      if (tmin > tmax) { const float s = tmin; tmin = tmax; tmax = s; }
      tmax *= kSlabScale;
      t0 = tmin > t0 ? tmin : t0;
      t1 = tmax < t1 ? tmax : t1;
    }
    {
      float tmin = (nd.miny - o.y) * inv.y, tmax = (nd.maxy - o.y) * inv.y;
      if (tmin > tmax) { const float s = tmin; tmin = tmax; tmax = s; }
      tmax *= kSlabScale;
      t0 = tmin > t0 ? tmin : t0;
      t1 = tmax < t1 ? tmax : t1;
    }
    {
      float tmin = (nd.minz - o.z) * inv.z, tmax = (nd.maxz - o.z) * inv.z;
      if (tmin > tmax) { const float s = tmin; tmin = tmax; tmax = s; }
      tmax *= kSlabScale;
      t0 = tmin > t0 ? tmin : t0;
      t1 = tmax < t1 ? tmax : t1;
    }
    const bool boxHit = !(t0 > t1);
    const bool isLeaf = node_is_leaf(nd);
    if (boxHit && isLeaf) {
      if (STATS) cs.leaves++;
      const GLeaf L = sc.leaves[i];               // (leaves[] is indexed by node)
      float t, b0 = 0.f, b1 = 0.f, b2 = 0.f;
      bool cand;
      if (leaf_kind(L) == LEAF_TRI) {
        t = intersect_triangle<DF>(mk(L.f[0], L.f[1], L.f[2]), mk(L.f[3], L.f[4], L.f[5]), mk(L.f[6], L.f[7], L.f[8]), o, sh, b0, b1, b2);
        cand = t > 0.f && t < kInf;                 // Mesh.hpp:93
      } else if (leaf_kind(L) == LEAF_SPHERE) {
        t = intersect_sphere(L, o, d, tMin);
        cand = true;                                // Failed() carries t = 0, rejected by t > tMin below
      } else {
        t = intersect_disc(L, o, d);
        cand = true;
      }
      if (cand && t > tMin && t < hit.t) {          // CompactBvh.hpp:124 / :60 (hit.t == ray.tMax for any-hit)
        if (ANY_HIT) return true;
        hit.t = t; hit.leaf = i; hit.geomID = leaf_geom(L); hit.b0 = b0; hit.b1 = b1; hit.b2 = b2;
      }
    }
    // next node in the reference's visit order
    i = (boxHit && !isLeaf) ? i + 1 : (nd.link >> 5);
  }
  return hit.leaf != 0xFFFFFFFFu;
}

// Primitive::normal for the closest hit (Mesh.hpp:107-121, Primitives.hpp:48-50,72). `hp` is the
// advanced ray origin (Render.hpp:21-22).
__device__ __forceinline__ f3 hit_normal(const DeviceScene& sc, const Hit& h, f3 hp) {
  const GLeaf L = sc.leaves[h.leaf];
  if (leaf_kind(L) == LEAF_TRI) {
    if (!sc.hasNormals) return mk(L.n[0], L.n[1], L.n[2]);
    // normals[firstVertex + tris[triBase + k]], k = 0..2 (Mesh.hpp:115-120), gathered per leaf at upload: one load
    // that depends on the leaf's index alone instead of three hops (first vertex, vertex indices, normals)
    const float* q = sc.leafNormals + 9 * (size_t)h.leaf;
    return normalized((mk(q[0], q[1], q[2]) * h.b0) + (mk(q[3], q[4], q[5]) * h.b1) + (mk(q[6], q[7], q[8]) * h.b2));
  }
  if (leaf_kind(L) == LEAF_SPHERE) return normalized(hp - mk(L.f[0], L.f[1], L.f[2]));
  return mk(L.f[0], L.f[1], L.f[2]);
}

__device__ __forceinline__ void flush_stats(const DeviceScene& sc, uint32_t casts, const CastStats& cs, uint32_t paths) {
  // one atomic per wave and counter
  unsigned long long c = casts, n = cs.nodes, l = cs.leaves, p = paths;
  for (int off = 32; off > 0; off >>= 1) {
    c += __shfl_down(c, off); n += __shfl_down(n, off); l += __shfl_down(l, off); p += __shfl_down(p, off);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&sc.counters[0], c);
    if (n) atomicAdd(&sc.counters[1], n);
    if (l) atomicAdd(&sc.counters[2], l);
    if (p) atomicAdd(&sc.counters[3], p);
  }
}

__constant__ uint32_t kSinBits[92] = {MI_SIN_TABLE_BITS};

__device__ __forceinline__ void load_sin_table(float* tbl) {
  for (uint32_t k = threadIdx.x; k < 92; k += blockDim.x) tbl[k] = __uint_as_float(kSinBits[k]);
  __syncthreads();
}

// ---- K2: shadow trace ---------------------------------------------------------------------------------
template <bool STATS, bool DF = false>
__global__ void __launch_bounds__(256) shadow_trace_kernel(DeviceScene sc, mi_trace_result* rays, uint32_t n, float ambient, f3 lightPos) {
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  CastStats cs = {0, 0};
  uint32_t casts = 0;
  if (idx < n) {
    mi_trace_result* res = rays + idx;
    const mi_hit_record h = res->h;
    f3 o = mk(h.r.origin.x, h.r.origin.y, h.r.origin.z), d = mk(h.r.direction.x, h.r.direction.y, h.r.direction.z);
    Hit hit;
    ++casts;
    if (traverse<false, STATS, DF>(sc, o, d, h.r.t_min, h.r.t_max, hit, cs)) {
      // updateHit, Render.hpp:15-23
      const f3 hp = o + d * hit.t;
      const f3 nrm = hit_normal(sc, hit, hp);
      const mi_material mat = sc.materials[sc.matIDs[hit.geomID]];
      const f3 lightOffset = lightPos - hp;
      const f3 sd = normalized(lightOffset);
      const f3 so = offset_origin(hp, sd, nrm);
      const float sTmax = sqrtf(sqnorm(lightOffset));
      const f3 albedo = mk(mat.albedo.x, mat.albedo.y, mat.albedo.z);
      f3 color = albedo * ambient;
      Hit shadowHit;
      ++casts;
      if (!traverse<true, STATS, DF>(sc, so, sd, 0.f, sTmax, shadowHit, cs)) color = color + albedo * dot(sd, nrm);
      res->rgb = {color.x, color.y, color.z};
      res->h.r.origin = {hp.x, hp.y, hp.z};
      res->h.r.t_max = hit.t;
      res->h.prim_id = sc.leaves[hit.leaf].primID;
      res->h.normal = {nrm.x, nrm.y, nrm.z};
      res->h.geom_id = (uint16_t)hit.geomID;
    } else {
      res->h.flags = h.flags | MI_FLAG_ESCAPED;
    }
  }
  flush_stats(sc, casts, cs, idx < n ? 1u : 0u);
}

// ---- K1: path trace -----------------------------------------------------------------------------------
// One lane owns one pixel's ray-stream entry and runs all samplesPerPixel samples of it, like a
// worker of the PathTrace vertex runs the sample loop inside the vertex (codelets :191). Random
// numbers come from the lane's own xoroshiro128** stream.
struct PathState {
  f3 o, d, n, tp;
  float tMax;
  uint32_t primID, geomID, flags;
};

template <bool STATS, bool DF = false>
__global__ void __launch_bounds__(256) path_trace_kernel(DeviceScene sc, mi_trace_result* rays, uint32_t n, uint32_t firstSample, uint32_t numSamples, Rng* rngStates) {
  __shared__ float sinTbl[92];
  load_sin_table(sinTbl);
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  CastStats cs = {0, 0};
  uint32_t casts = 0, paths = 0;
  if (idx < n) {
    mi_trace_result* res = rays + idx;
    const float prow = res->u, pcol = res->v;
    Rng rng;
    // sample-at-a-time form (NIF renders): the stream lives in rngStates between launches and is re-seeded at
    // every segment boundary of the render (the host rolls the partial sums there, nif_segment_roll_kernel)
    const uint32_t segLenRender = segment_samples(sc.samplesPerPixel);
    if (rngStates && (firstSample % segLenRender) != 0) rng = rngStates[idx];
    else rng_seed_pixel_segment(rng, sc.rngSeed, prow, pcol, rngStates ? firstSample / segLenRender : 0u);
    f3 rgb = mk(res->rgb.x, res->rgb.y, res->rgb.z);
    PathState ps;
    ps.flags = 0; ps.primID = MI_INVALID_PRIM; ps.geomID = MI_INVALID_GEOM; ps.tMax = kInf;
    ps.o = mk(0, 0, 0); ps.d = mk(0, 0, -1); ps.n = mk(0, 0, 1); ps.tp = mk(1, 1, 1);
    f3 total = rgb;                          // segments (ray_math.h, segment_samples): sum of the finished ones
    const uint32_t segLen = segment_samples(numSamples);
    const bool segmented = rngStates == nullptr;       // (the sample-at-a-time form is one sample per launch: the host rolls its segments)
    for (uint32_t s = 0; s < numSamples; ++s) {
      if (segmented && s != 0 && (s % segLen) == 0) {
        // a new segment: own stream, own partial sum; segment 0 accumulates onto the incoming rgb directly
        const uint32_t segment = s / segLen;
        if (segment > 1) total = total + rgb; else total = rgb;
        rgb = mk(0.f, 0.f, 0.f);
        rng_seed_pixel_segment(rng, sc.rngSeed, prow, pcol, segment);
      }
      // sampleCameraRays, codelets/TraceCodelets.cpp:142-164
      float g0, g1;
      rng_gauss2(rng, sinTbl, g0, g1);
      const float jr = prow + sc.antiAliasScale * g0, jc = pcol + sc.antiAliasScale * g1;
      ps.d = pixel_to_ray_dir(jc, jr, sc.imageWidth, sc.imageHeight, sc.tanTheta);
      ps.o = mk(0.f, 0.f, 0.f);
      ps.n = mk(0.f, 0.f, 1.f);                       // HitRecord ctor, geometry.hpp:236-242
      ps.primID = MI_INVALID_PRIM; ps.geomID = MI_INVALID_GEOM; ps.flags = 0; ps.tMax = kInf;
      ps.tp = mk(1.f, 1.f, 1.f);
      f3 color = mk(0.f, 0.f, 0.f);
      for (uint32_t i = 0; i < sc.maxPathLength; ++i) {
        ps.o = offset_origin(ps.o, ps.d, ps.n);
        Hit hit;
        ++casts;
        if (traverse<false, STATS, DF>(sc, ps.o, ps.d, 0.f, kInf, hit, cs)) {
          // updateHit
          ps.geomID = hit.geomID;
          ps.primID = sc.leaves[hit.leaf].primID;
          ps.tMax = hit.t;
          ps.o = ps.o + ps.d * hit.t;
          ps.n = hit_normal(sc, hit, ps.o);
          const mi_material mat = sc.materials[sc.matIDs[hit.geomID]];
          const f3 albedo = mk(mat.albedo.x, mat.albedo.y, mat.albedo.z);
          if (mat.emissive) color = color + ps.tp * mk(mat.emission.x, mat.emission.y, mat.emission.z);
          if (mat.type == 0) {
            const float u1 = rng_uniform01(rng);
            const float u2 = rng_uniform01(rng);
            ps.d = sample_diffuse(ps.n, u1, u2, sinTbl);
            ps.tp = ps.tp * albedo;
          } else if (mat.type == 1) {
            ps.d = reflect_dir(ps.d, ps.n);
            ps.tp = ps.tp * albedo;
          } else if (mat.type == 2) {
            const float u1 = rng_uniform01(rng);
            f3 nd;
            const bool refracted = dielectric(ps.d, ps.n, mat.ior, u1, nd);
            ps.d = nd;
            if (refracted) ps.tp = ps.tp * albedo;
          } else {
            rgb = rgb * __builtin_nanf("");
            ps.flags |= MI_FLAG_ERROR;
          }
        } else {
          ps.tMax = kInf;
          ps.flags |= MI_FLAG_ESCAPED;
          break;
        }
        if (i > sc.rouletteStartDepth) {
          const float u1 = rng_uniform01(rng);
          if (roulette_stop(u1, ps.tp)) break;
        }
      }
      rgb = rgb + color;
      ++paths;
    }
    if (rngStates) rngStates[idx] = rng;
    if (segmented && numSamples > segLen) rgb = total + rgb;     // + the last segment's partial sum
    if (numSamples) {
      res->rgb = {rgb.x, rgb.y, rgb.z};
      mi_hit_record hr;
      hr.r.origin = {ps.o.x, ps.o.y, ps.o.z}; hr.r.t_min = 0.f;
      hr.r.direction = {ps.d.x, ps.d.y, ps.d.z}; hr.r.t_max = ps.tMax;
      hr.prim_id = ps.primID;
      hr.normal = {ps.n.x, ps.n.y, ps.n.z};
      hr.throughput = {ps.tp.x, ps.tp.y, ps.tp.z};
      hr.geom_id = (uint16_t)ps.geomID; hr.flags = (uint16_t)ps.flags;
      res->h = hr;
    }
  }
  flush_stats(sc, casts, cs, paths);
}

}  // namespace mi
