// render_kernels.h — launch interface of the GDPT render kernels (host side of render_kernels.hip).
#pragma once
#include "../device_scene.h"
#include <hip/hip_runtime.h>

namespace gdpt {

struct RenderCounters {            // device-resident, zeroed per render
    unsigned long long rays, bounces, nonfinite, nodes, prims;
};

struct RenderLaunch {
    int spp;
    int rng_scheme;                // GDPT_RNG_*
    int row_begin, row_end;
    int max_depth;                 // effective (scene value or override)
    double *img, *cx0, *cy0, *cx1, *cy1;   // device, W*H*3 each
    RenderCounters *counters;      // device
    bool count_traversal;          // counting build: BVH nodes / primitives per ray
};

// Enqueues the five-buffer render on `stream`. Throws std::runtime_error on a launch failure.
void launch_render(const DevSceneView &sv, const RenderLaunch &rl, hipStream_t stream);
// Name of the dominant kernel of the last launch configuration (for rocprof matching).
const char *render_kernel_name(int rng_scheme);

} // namespace gdpt
