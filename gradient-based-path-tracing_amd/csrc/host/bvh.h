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

} // namespace gdpt
