// bvh_sah.cpp — host BVH2 builder emitting the reference's CompactBVH2Node array.
//
// The reference drives Embree's rtcBuildBVH (branching factor 2, one primitive per leaf, SAH;
// include/embree_utils/bvh.hpp:47-69) and flattens the pointer tree depth-first
// (src/CompactBvhBuild.cpp:34-56). Embree is not available here and its tree topology is not
// part of the format, so this is an own full-sweep SAH builder followed by an insertion-based
// optimisation of the tree (below); what IS contractual and reproduced exactly is the node
// encoding (src/CompactBvhBuild.cpp:5-32):
//   * node i's first child is node i+1, the second child index is stored in the node;
//   * interior nodes carry geomID 0xFFFF and the union of their children's boxes;
//   * extents are stored as binary16 rounded UP (precision_utils.hpp:39-47), and an extent
//     above 65504 is an error;
//   * maxDepth counts levels with the root at depth 1 and bounds the traversal stack.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <numeric>
#include <queue>
#include <stdexcept>

#include "scene_types.hpp"

namespace mi::host {

namespace {

struct TreeNode {
  Bounds box;
  int child[2] = {-1, -1};
  int prim = -1;   // index into the BuildPrim array for leaves
};

double halfArea(const Bounds& b) {
  const double dx = (double)b.hi.x - b.lo.x, dy = (double)b.hi.y - b.lo.y, dz = (double)b.hi.z - b.lo.z;
  return dx * dy + dy * dz + dz * dx;
}

struct Builder {
  const std::vector<BuildPrim>& prims;
  std::vector<TreeNode> tree;
  std::vector<f3> centroid;

  explicit Builder(const std::vector<BuildPrim>& p) : prims(p) {
    centroid.reserve(p.size());
    for (auto& bp : p) centroid.push_back((bp.box.lo + bp.box.hi) * .5f);
    tree.reserve(2 * p.size());
  }

  int build(std::vector<uint32_t>& ids, size_t begin, size_t end) {
    const int me = (int)tree.size();
    tree.emplace_back();
    const size_t n = end - begin;
    if (n == 1) {
      tree[me].prim = (int)ids[begin];
      tree[me].box = prims[ids[begin]].box;
      return me;
    }

    // Full sweep along each axis over centroid-sorted primitives; cost = A_l*N_l + A_r*N_r
    // (Embree's traversal/intersection costs scale both candidates alike when every leaf holds
    // exactly one primitive, so they do not change the arg-min).
    double bestCost = std::numeric_limits<double>::infinity();
    int bestAxis = -1;
    size_t bestSplit = 0;
    std::vector<uint32_t> order(ids.begin() + begin, ids.begin() + end);
    std::vector<double> rightArea(n);
    auto sortAlong = [&](int axis) {
      std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        const float ca = comp(centroid[a], axis), cb = comp(centroid[b], axis);
        if (ca != cb) return ca < cb;
        return a < b;   // total order: the result does not depend on the incoming permutation
      });
    };
    for (int axis = 0; axis < 3; ++axis) {
      sortAlong(axis);
      Bounds acc;
      for (size_t i = n; i-- > 1;) { acc.grow(prims[order[i]].box); rightArea[i] = halfArea(acc); }
      acc = Bounds();
      for (size_t i = 1; i < n; ++i) {
        acc.grow(prims[order[i - 1]].box);
        const double cost = halfArea(acc) * (double)i + rightArea[i] * (double)(n - i);
        if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestSplit = i; }
      }
    }
    if (bestAxis < 0) { bestSplit = n / 2; }   // degenerate (NaN boxes): split in the middle, keep going
    else sortAlong(bestAxis);
    std::copy(order.begin(), order.end(), ids.begin() + begin);
    const int l = build(ids, begin, begin + bestSplit);
    const int r = build(ids, begin + bestSplit, end);
    tree[me].child[0] = l;
    tree[me].child[1] = r;
    // InnerNode::setBounds: union of the two child boxes (embree_utils/node.hpp:56-70)
    Bounds u;
    u.grow(tree[l].box);
    u.grow(tree[r].box);
    tree[me].box = u;
    return me;
  }
};

// ---- tree optimisation by reinsertion ---------------------------------------------------------------------------
// The sweep builder is greedy, top down. What a ray pays during traversal is one box test per child of every interior
// node whose box it hits, i.e. (for uniformly distributed rays) the sum of the interior nodes' surface areas. That sum
// is lowered after the build by taking subtrees out of the tree and putting each back at the position that increases
// it least (Bittner, Hapala, Havran: "Fast insertion-based optimization of bounding volume hierarchies", 2013): a
// node X is removed together with its parent (the sibling takes the parent's place), the best new sibling Y is found
// by branch and bound over the induced area increase of Y's ancestors, and the freed parent becomes the common
// parent of X and Y. The node format, the one-primitive-per-leaf rule and the depth-first layout are untouched; only
// the topology - which the reference leaves to Embree and is not part of its format - changes.
struct Reinserter {
  std::vector<TreeNode>& t;
  std::vector<int> parent;
  int root;

  Reinserter(std::vector<TreeNode>& tree, int r) : t(tree), parent(tree.size(), -1), root(r) {
    for (int i = 0; i < (int)t.size(); ++i)
      for (int c : t[i].child) if (c >= 0) parent[c] = i;
  }
  static Bounds merged(const Bounds& a, const Bounds& b) { Bounds u = a; u.grow(b); return u; }
  void refitUp(int n) {
    for (; n >= 0; n = parent[n]) t[n].box = merged(t[t[n].child[0]].box, t[t[n].child[1]].box);
  }
  double interiorArea() const {
    double a = 0;
    for (const TreeNode& n : t) if (n.prim < 0) a += halfArea(n.box);
    return a;
  }
  // one reinsertion attempt of subtree x; returns the change of the interior-area sum (<= 0 when it moved)
  bool reinsert(int x) {
    const int p = parent[x];
    if (p < 0 || parent[p] < 0) return false;                 // the root and its children stay
    const int g = parent[p];
    const int s = t[p].child[0] == x ? t[p].child[1] : t[p].child[0];
    // take x and p out: s takes p's place
    t[g].child[t[g].child[0] == p ? 0 : 1] = s;
    parent[s] = g;
    const double before = halfArea(t[p].box);
    std::vector<std::pair<int, Bounds>> saved;                // ancestors' boxes, to undo cheaply
    for (int a = g; a >= 0; a = parent[a]) { saved.emplace_back(a, t[a].box); t[a].box = merged(t[t[a].child[0]].box, t[t[a].child[1]].box); }
    double removedGain = before;
    for (auto& sv : saved) removedGain += halfArea(sv.second) - halfArea(t[sv.first].box);
    // branch and bound for the best sibling y
    const Bounds& xb = t[x].box;
    const double xArea = halfArea(xb);
    struct Cand { double induced; int node; bool operator<(const Cand& o) const { return induced > o.induced; } };
    std::priority_queue<Cand> q;
    q.push({0.0, root});
    double bestCost = std::numeric_limits<double>::infinity();
    int best = -1;
    while (!q.empty()) {
      const Cand c = q.top(); q.pop();
      if (c.induced + xArea >= bestCost) break;
      const double direct = halfArea(merged(t[c.node].box, xb));
      const double total = c.induced + direct;
      if (total < bestCost) { bestCost = total; best = c.node; }
      if (t[c.node].prim < 0) {
        const double childInduced = total - halfArea(t[c.node].box);
        if (childInduced + xArea < bestCost) { q.push({childInduced, t[c.node].child[0]}); q.push({childInduced, t[c.node].child[1]}); }
      }
    }
    if (best < 0 || bestCost >= removedGain - 1e-12 * removedGain) {
      // no better place: put everything back
      t[g].child[t[g].child[0] == s ? 0 : 1] = p;
      parent[p] = g; parent[s] = p;
      for (auto& sv : saved) t[sv.first].box = sv.second;
      return false;
    }
    // p becomes the parent of (best, x) where best stood
    const int bp = parent[best];
    if (bp >= 0) t[bp].child[t[bp].child[0] == best ? 0 : 1] = p; else root = p;
    parent[p] = bp;
    t[p].child[0] = best; t[p].child[1] = x;
    parent[best] = p; parent[x] = p;
    refitUp(p);
    return true;
  }
  int run(int passes) {
    int moved = 0;
    for (int pass = 0; pass < passes; ++pass) {
      std::vector<int> order;
      for (int i = 0; i < (int)t.size(); ++i) if (i != root) order.push_back(i);
      // largest boxes first: they are the ones whose misplacement costs most
      std::vector<double> area(t.size());
      for (int i : order) area[i] = halfArea(t[i].box);
      std::sort(order.begin(), order.end(), [&](int a, int b) { return area[a] != area[b] ? area[a] > area[b] : a < b; });
      int movedPass = 0;
      for (int x : order) if (reinsert(x)) ++movedPass;
      moved += movedPass;
      if (movedPass == 0) break;
    }
    return moved;
  }
};

mi_bvh_node toCompact(const TreeNode& t, const std::vector<BuildPrim>& prims) {
  mi_bvh_node c;
  c.min_x = t.box.lo.x; c.min_y = t.box.lo.y; c.min_z = t.box.lo.z;
  const float dx = t.box.hi.x - t.box.lo.x, dy = t.box.hi.y - t.box.lo.y, dz = t.box.hi.z - t.box.lo.z;
  const float maxHalf = 65504.f;
  if (dx > maxHalf || dy > maxHalf || dz > maxHalf)
    throw std::runtime_error("Cannot compress BVH bounds into fp16 (half)");
  c.dx = half_not_smaller(dx); c.dy = half_not_smaller(dy); c.dz = half_not_smaller(dz);
  if (t.prim >= 0) { c.geom_id = prims[t.prim].geomID; c.prim_or_second_child = prims[t.prim].primID; }
  else { c.geom_id = MI_INVALID_GEOM; c.prim_or_second_child = 0; }
  return c;
}

uint32_t flatten(const Builder& b, int node, std::vector<mi_bvh_node>& out, uint32_t depth, uint32_t& maxDepth) {
  const uint32_t my = (uint32_t)out.size();
  out.push_back(toCompact(b.tree[node], b.prims));
  if (depth > maxDepth) maxDepth = depth;
  if (b.tree[node].prim < 0) {
    flatten(b, b.tree[node].child[0], out, depth + 1, maxDepth);
    const uint32_t second = flatten(b, b.tree[node].child[1], out, depth + 1, maxDepth);
    out[my].prim_or_second_child = second;
  }
  return my;
}

}  // namespace

void buildCompactBvh(const std::vector<BuildPrim>& prims, std::vector<mi_bvh_node>& nodes, uint32_t& maxDepth) {
  nodes.clear();
  maxDepth = 0;
  if (prims.empty()) return;
  Builder b(prims);
  std::vector<uint32_t> ids(prims.size());
  std::iota(ids.begin(), ids.end(), 0u);
  int root = b.build(ids, 0, ids.size());
  {
    // MI_BVH_REINSERT=<passes> overrides the default of 4 (0 = the plain sweep tree). Measured with the oracle's
    // counters, 160x160 x 8 spp: box scene 21.08 -> 18.85 box tests and 2.55 -> 2.35 primitive tests per cast,
    // test_scene.dae 30.9 -> 28.0 and 3.43 -> 3.22 (converged after 3 passes).
    const char* e = std::getenv("MI_BVH_REINSERT");
    const int passes = e ? std::atoi(e) : 4;
    if (passes > 0 && prims.size() > 2) {
      Reinserter r(b.tree, root);
      const double a0 = r.interiorArea();
      const int moved = r.run(passes);
      root = r.root;
      if (std::getenv("MI_BVH_VERBOSE")) std::fprintf(stderr, "[bvh] reinsertion: %d moves, interior area %.6g -> %.6g\n", moved, a0, r.interiorArea());
    }
  }
  {
    // Child order. The walk always takes the first child first (CompactBvh.hpp:132-133), so the first child should
    // be the one more likely to hold the closest hit: what is found there prunes the other. The reference moves every
    // scene so that the camera sits at the origin (src/scene_utils.cpp:58-120), a third of all casts are primary
    // rays from there, and the rest start on surfaces those rays reach; "the child whose box centre is nearer to the
    // origin first" measured best of six static rules (oracle counters, 160x160 x 8 spp): box scene 18.85 -> 18.79
    // box tests and 2.35 -> 2.26 primitive tests per cast, test_scene.dae 27.9 -> 26.1 and 3.22 -> 2.98, monkey bust
    // 2.64 -> 2.37 and 0.150 -> 0.112 (larger-area-first: 18.83 / 2.33; smaller-area-first, farther-first and
    // fewer-primitives-first are worse than no rule). MI_BVH_ORDER=0 keeps the builder's order.
    const char* e = std::getenv("MI_BVH_ORDER");
    if (!(e && e[0] == '0')) {
      auto dist2 = [](const Bounds& x) { const f3 c = (x.lo + x.hi) * .5f; return (double)c.x * c.x + (double)c.y * c.y + (double)c.z * c.z; };
      for (auto& n : b.tree)
        if (n.prim < 0 && dist2(b.tree[n.child[1]].box) < dist2(b.tree[n.child[0]].box)) std::swap(n.child[0], n.child[1]);
    }
  }
  nodes.reserve(2 * prims.size() - 1);
  flatten(b, root, nodes, 1, maxDepth);
}

}  // namespace mi::host
