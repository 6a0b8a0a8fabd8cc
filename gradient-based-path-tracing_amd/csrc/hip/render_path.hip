// render_path.hip — launchers of Integrator::Path (kernels: render_path.h) + its non-template kernels.
#define GDPT_BUILD_PATH_MISC 1
#include "render_path.h"

namespace gdpt {
void launch_path(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, hipStream_t stream) {
    hipLaunchKernelGGL(gd::gdpt_path_eager, grid, dim3(gd::kBlock), 0, stream, sv, a);
}
void launch_path_persistent(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, bool lds, bool lambert, bool plain, hipStream_t stream) {
    if (lambert) launch_path_persistent_lambert(sv, a, grid, lds, plain, stream);
    else launch_path_persistent_general(sv, a, grid, lds, stream);
    const long long nslots = a.num_slots;
    hipLaunchKernelGGL(gd::gdpt_path_reduce, dim3((unsigned)((nslots * 4 + 255) / 256)), dim3(256), 0, stream, a, sv.cam.width);
}
void launch_tile_path(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, int ntx, int nty, hipStream_t stream) {
    hipLaunchKernelGGL(gd::gdpt_path_tile_stream, grid, dim3(64), 0, stream, sv, a, ntx, nty);
}
} // namespace gdpt
