// Lambertian-only scenes made of triangles: the lane machine built without the sphere test and the sphere shading frame
// and, where every texture is constant (cbox), without the image / checkerboard lookups (device_trace.h: PLAIN). Same
// arithmetic on every path such a scene can take; the LDS variant for cbox needs 217 instead of 254 VGPRs and spills 34
// instead of 87 SGPRs.
#include "render_device.h"
namespace gdpt {
void launch_phases_lambert_plain(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, bool lds, bool const_tex, hipStream_t stream) {
    if (lds && const_tex) hipLaunchKernelGGL((gd::gdpt_render_phases<true, true, true, true, false, gd::kPlainBoth>), grid, dim3(gd::kBlock), 0, stream, sv, a);
    else if (lds) hipLaunchKernelGGL((gd::gdpt_render_phases<true, true, true, true, false, gd::kPlainNoSpheres>), grid, dim3(gd::kBlock), 0, stream, sv, a);
    else if (const_tex) hipLaunchKernelGGL((gd::gdpt_render_phases<true, false, true, true, false, gd::kPlainBoth>), grid, dim3(gd::kBlock), gd::hbm_dynamic_lds(a), stream, sv, a);
    else hipLaunchKernelGGL((gd::gdpt_render_phases<true, false, true, true, false, gd::kPlainNoSpheres>), grid, dim3(gd::kBlock), gd::hbm_dynamic_lds(a), stream, sv, a);
}
} // namespace gdpt
