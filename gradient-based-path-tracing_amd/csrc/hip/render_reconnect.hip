// render_reconnect.hip — GDPT_SHIFT_RECONNECT, general materials (kernels: render_reconnect.h) + the launcher.
#include "render_reconnect.h"

namespace gdpt {
void launch_reconnect_lambert(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, bool lds, hipStream_t stream);
void launch_reconnect(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, bool lds, bool lambert, hipStream_t stream) {
    if (lambert) launch_reconnect_lambert(sv, a, grid, lds, stream);
    else hipLaunchKernelGGL((gd::gdpt_render_reconnect<false, false>), grid, dim3(gd::kBlock), 0, stream, sv, a);
}
} // namespace gdpt
