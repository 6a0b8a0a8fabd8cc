// GradPath lane machine for scenes with two-sided lobes (DisneyGlass, DisneyBSDF): offsets replayed from a bounce log.
// The full material switch lives here; scenes whose materials fit one of the small sets go to the kernels built for it
// (render_phases_twosided_sets.hip; HBM scenes only — an LDS-sized scene with a Disney lobe has not been met yet).
#include "render_twosided.h"
namespace gdpt {
size_t twosided_log_bytes(unsigned blocks) { return (size_t)blocks * gd::kBlock * gd::kLogCap * sizeof(gd::BounceLog); }
void launch_phases_twosided(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, bool lds, unsigned material_mask, void *bounce_log, hipStream_t stream) {
    if (!lds && (material_mask & ~gd::kSetGlass) == 0) { launch_phases_twosided_glass(sv, a, grid, bounce_log, stream); return; }
    if (lds) hipLaunchKernelGGL((gd::gdpt_render_twosided<true>), grid, dim3(gd::kBlock), 0, stream, sv, a, (gd::BounceLog *)bounce_log);
    else hipLaunchKernelGGL((gd::gdpt_render_twosided<false>), grid, dim3(gd::kBlock), gd::hbm_dynamic_lds(a), stream, sv, a, (gd::BounceLog *)bounce_log);
}
} // namespace gdpt
