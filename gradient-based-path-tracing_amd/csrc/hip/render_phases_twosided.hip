// GradPath lane machine for scenes with two-sided lobes (DisneyGlass, DisneyBSDF): offsets replayed from a bounce log.
#include "render_twosided.h"
namespace gdpt {
size_t twosided_log_bytes(unsigned blocks) { return (size_t)blocks * gd::kBlock * gd::kLogCap * sizeof(gd::BounceLog); }
void launch_phases_twosided(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, bool lds, void *bounce_log, hipStream_t stream) {
    if (lds) hipLaunchKernelGGL((gd::gdpt_render_twosided<true>), grid, dim3(gd::kBlock), 0, stream, sv, a, (gd::BounceLog *)bounce_log);
    else hipLaunchKernelGGL((gd::gdpt_render_twosided<false>), grid, dim3(gd::kBlock), 0, stream, sv, a, (gd::BounceLog *)bounce_log);
}
} // namespace gdpt
