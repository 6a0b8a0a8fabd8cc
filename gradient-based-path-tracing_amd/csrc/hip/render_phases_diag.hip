// Diagnostic build of the Lambertian lane machine with in-kernel cycle stamps (render_device.h: Stamps<true>). Selected
// only by the test-only knob "stamps" (include/gdpt_debug.h); its run time is never quoted — its segment SHARES are.
#include "render_device.h"
namespace gdpt {
void launch_phases_lambert_stamped(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, bool lds, bool plain, hipStream_t stream) {
    if (lds && plain) hipLaunchKernelGGL((gd::gdpt_render_phases<true, true, true, true, true, gd::kPlainBoth>), grid, dim3(gd::kBlock), 0, stream, sv, a);
    else if (lds) hipLaunchKernelGGL((gd::gdpt_render_phases<true, true, true, true, true>), grid, dim3(gd::kBlock), 0, stream, sv, a);
    else hipLaunchKernelGGL((gd::gdpt_render_phases<true, false, true, true, true>), grid, dim3(gd::kBlock), gd::hbm_dynamic_lds(a), stream, sv, a);
}
} // namespace gdpt
