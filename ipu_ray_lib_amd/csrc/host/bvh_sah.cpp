// bvh_sah.cpp — host BVH2 builder emitting the reference's CompactBVH2Node array.
//
// The reference drives Embree's rtcBuildBVH (branching factor 2, one primitive per leaf, SAH;
// include/embree_utils/bvh.hpp:47-69) and flattens the pointer tree depth-first
// (src/CompactBvhBuild.cpp:34-56). Embree is not available here and its tree topology is not
// part of the format, so this is an own full-sweep SAH builder; what IS contractual and
// reproduced exactly is the node encoding (src/CompactBvhBuild.cpp:5-32):
//   * node i's first child is node i+1, the second child index is stored in the node;
//   * interior nodes carry geomID 0xFFFF and the union of their children's boxes;
//   * extents are stored as binary16 rounded UP (precision_utils.hpp:39-47), and an extent
//     above 65504 is an error;
//   * maxDepth counts levels with the root at depth 1 and bounds the traversal stack.
#include <algorithm>
#include <limits>
#include <numeric>
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
  const int root = b.build(ids, 0, ids.size());
  nodes.reserve(2 * prims.size() - 1);
  flatten(b, root, nodes, 1, maxDepth);
}

}  // namespace mi::host
