// bvh.h — host-side BVH2 builder (binned SAH, bounded depth) over triangles + spheres.
// Replaces the Embree scene build of the reference (src/scene.cpp:20-31: RTC_BUILD_QUALITY_HIGH).
#pragma once
#include "../device_scene.h"
#include <vector>

namespace gdpt {

struct PrimBounds {
    float bmin[3], bmax[3];
};

struct BvhBuildResult {
    std::vector<DevBvhNode> nodes;     // nodes[0] is the root (always an inner node when num prims > 0)
    std::vector<uint32_t> order;       // order[i] = input primitive index stored at leaf slot i
    int depth = 0;                     // number of inner-node levels on the longest root->leaf path
};

// `bounds[i]` must enclose primitive i (conservatively). Leaves hold 1..GDPT_LEAF_MAX_PRIMS primitives.
BvhBuildResult build_bvh(const std::vector<PrimBounds> &bounds);

// Collapses a BVH2 into nodes of up to `max_children` (2..4) children: the child with the largest surface area is
// replaced by its own two children until the node is full. Nodes are emitted breadth-first (the top of the tree is
// contiguous). The traversal stack bound stays within `stack_slots` (arity drops locally along over-deep paths; the
// BVH2's own depth must fit); `*stack_need` receives the bound actually reached.
std::vector<DevBvh4Node> collapse_bvh4(const std::vector<DevBvhNode> &nodes, int max_children, int stack_slots, int *stack_need);

// The same collapse to up to 8 children, written as DevBvh8Node (child boxes quantised to 8 bits on a per-node grid,
// always enclosing the boxes they replace). Same stack rule.
std::vector<DevBvh8Node> collapse_bvh8(const std::vector<DevBvhNode> &nodes, int stack_slots, int *stack_need);

// DevBvh4QNode form of a BVH4 (same topology and indices; child boxes on a per-node 8-bit grid that encloses them, checked in
// double). Throws if a box cannot be represented.
std::vector<DevBvh4QNode> quantise_bvh4(const std::vector<DevBvh4Node> &nodes);

// The wide trees the device walks (stack bounds <= GDPT_BVH_MAX_DEPTH slots); arity = widest BVH4 node present.
struct WideBvh { std::vector<DevBvh4Node> nodes; std::vector<DevBvh8Node> nodes8; int arity = 0, stack_need = 0, stack_need8 = 0; };
// with_bvh8: also the quantised 8-wide form (the host check and the GDPT_HBM_BVH8 A/B build; product uploads do not need it)
WideBvh collapse_for_traversal(const std::vector<DevBvhNode> &nodes, bool with_bvh8);

// The same tree form built with spatial splits (host/sbvh.cpp): a triangle that straddles a chosen split plane is referenced from
// both children. `tri_verts`: 9 floats per triangle for the first tri_verts.size()/9 entries of `bounds` (spheres follow, never
// split). `budget` = extra references allowed as a fraction of the primitive count. On return ref_prim[order[i]] = primitive stored
// at leaf slot i (order.size() = references >= primitives).
BvhBuildResult build_sbvh(const std::vector<PrimBounds> &bounds, const std::vector<float> &tri_verts, double budget, std::vector<uint32_t> *ref_prim);

} // namespace gdpt

namespace gdpt {
// Early split clipping of large triangles before the build: a triangle whose box is large is referenced from several
// smaller boxes (each the box of the triangle clipped to a cell of its own box, rounded outward), so the SAH build can
// separate geometry that long or diagonal triangles would otherwise glue together. References never change what a ray
// hits — every piece points at the same primitive, the pieces cover the triangle — only how many boxes it visits.
// `tri_verts`: 9 floats per triangle for the first tri_verts.size()/9 entries of `bounds` (spheres follow unsplit).
// `budget` = extra references allowed as a fraction of the primitive count. ref_prim[r] = primitive of reference r.
void presplit_triangles(const std::vector<PrimBounds> &bounds, const std::vector<float> &tri_verts, double budget,
                        std::vector<PrimBounds> *refs, std::vector<uint32_t> *ref_prim);
} // namespace gdpt
