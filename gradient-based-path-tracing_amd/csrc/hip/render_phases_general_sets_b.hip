// {Lambertian, DisneyClearcoat} and {Lambertian, DisneySheen}: see render_phases_general_sets.h
#include "render_phases_general_sets.h"
namespace gdpt {
bool launch_phases_general_set_b(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, unsigned material_mask, hipStream_t stream) {
    if ((material_mask & ~(kSetLambert | 1u << GDPT_MAT_DISNEY_CLEARCOAT)) == 0) { launch_phases_set<kSetLambert | 1u << GDPT_MAT_DISNEY_CLEARCOAT>(sv, a, grid, stream); return true; }
    if ((material_mask & ~(kSetLambert | 1u << GDPT_MAT_DISNEY_SHEEN)) == 0) { launch_phases_set<kSetLambert | 1u << GDPT_MAT_DISNEY_SHEEN>(sv, a, grid, stream); return true; }
    return false;
}
} // namespace gdpt
