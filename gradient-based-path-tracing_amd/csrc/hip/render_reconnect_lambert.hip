// render_reconnect_lambert.hip — GDPT_SHIFT_RECONNECT for Lambertian-only scenes (cosine lobe inlined).
#include "render_reconnect.h"

namespace gdpt {
void launch_reconnect_lambert(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, bool lds, hipStream_t stream) {
    if (lds) hipLaunchKernelGGL((gd::gdpt_render_reconnect<true, true>), grid, dim3(gd::kBlock), 0, stream, sv, a);
    else hipLaunchKernelGGL((gd::gdpt_render_reconnect<true, false>), grid, dim3(gd::kBlock), 0, stream, sv, a);
}
} // namespace gdpt
