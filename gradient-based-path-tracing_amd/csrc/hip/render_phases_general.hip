// One-sided Disney lobes (diffuse, metal, clearcoat, sheen) mixed with Lambertian: phase-machine kernels
// with the full material switch.
#include "render_device.h"
namespace gdpt {
void launch_phases_general(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, bool lds, bool lds_wide, hipStream_t stream) {
    if (lds && lds_wide) hipLaunchKernelGGL((gd::gdpt_render_phases<false, true, true, true>), grid, dim3(gd::kBlock), 0, stream, sv, a);
    else if (lds) hipLaunchKernelGGL((gd::gdpt_render_phases<false, true, false, false>), grid, dim3(gd::kBlock), 0, stream, sv, a);
    else hipLaunchKernelGGL((gd::gdpt_render_phases<false, false, true, true>), grid, dim3(gd::kBlock), gd::hbm_dynamic_lds(a), stream, sv, a);
}
void launch_tile_phases_general(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, int ntx, int nty, hipStream_t stream) {
    hipLaunchKernelGGL((gd::gdpt_render_tile_stream_phases<false>), grid, dim3(64), 0, stream, sv, a, ntx, nty);
}
} // namespace gdpt
