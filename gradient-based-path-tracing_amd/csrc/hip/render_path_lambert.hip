// Integrator::Path lane machine, Lambertian-only scenes (cosine lobe inlined).
#include "render_path.h"
namespace gdpt {
template <bool LDS>
static void launch_env(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, hipStream_t stream) {
    if (sv.has_envmap) hipLaunchKernelGGL((gd::gdpt_path_persistent<true, LDS, true>), grid, dim3(gd::kBlock), 0, stream, sv, a);
    else hipLaunchKernelGGL((gd::gdpt_path_persistent<true, LDS, false>), grid, dim3(gd::kBlock), 0, stream, sv, a);
}
void launch_path_persistent_lambert(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, bool lds, bool plain, hipStream_t stream) {
    // triangles only, constant textures, no environment map, LDS-resident (cbox): the kernel without sphere / texture code
    if (lds && plain && !sv.has_envmap) hipLaunchKernelGGL((gd::gdpt_path_persistent<true, true, false, gd::kPlainBoth>), grid, dim3(gd::kBlock), 0, stream, sv, a);
    else if (lds) launch_env<true>(sv, a, grid, stream); else launch_env<false>(sv, a, grid, stream);
}
} // namespace gdpt
