// Eager evaluator: any material set (needed when a two-sided lobe — DisneyGlass, DisneyBSDF — is present).
#define GDPT_BUILD_EAGER 1
#include "render_device.h"
namespace gdpt {
void launch_eager(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, hipStream_t stream) {
    hipLaunchKernelGGL(gd::gdpt_render_eager, grid, dim3(gd::kBlock), 0, stream, sv, a);
}
void launch_tile_eager(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, int ntx, int nty, hipStream_t stream) {
    hipLaunchKernelGGL(gd::gdpt_render_tile_stream_eager, grid, dim3(64), 0, stream, sv, a, ntx, nty);
}
} // namespace gdpt
