// Lambertian-only scenes (cbox, sponza): phase-machine kernels with the cosine lobe inlined.
#define GDPT_BUILD_REDUCE 1
#include "render_device.h"
namespace gdpt {
void launch_phases_lambert(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, bool lds, bool lds_wide, hipStream_t stream) {
    if (lds && lds_wide) hipLaunchKernelGGL((gd::gdpt_render_phases<true, true, true, true>), grid, dim3(gd::kBlock), 0, stream, sv, a);
    else if (lds) hipLaunchKernelGGL((gd::gdpt_render_phases<true, true, false, false>), grid, dim3(gd::kBlock), 0, stream, sv, a);
    else hipLaunchKernelGGL((gd::gdpt_render_phases<true, false, true, true>), grid, dim3(gd::kBlock), gd::hbm_dynamic_lds(a), stream, sv, a);
}
void launch_reduce_partials(const DevSceneView &sv, const gd::KernelArgs &a, hipStream_t stream) {
    const long long nslots = a.num_slots;
    hipLaunchKernelGGL(gd::gdpt_reduce_partials, dim3((unsigned)((nslots * 16 + 255) / 256)), dim3(256), 0, stream, a, sv.cam.width);
}
void launch_tile_phases_lambert(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, int ntx, int nty, hipStream_t stream) {
    hipLaunchKernelGGL((gd::gdpt_render_tile_stream_phases<true>), grid, dim3(64), 0, stream, sv, a, ntx, nty);
}
} // namespace gdpt
