// {Lambertian, DisneyDiffuse} and {Lambertian, DisneyMetal}: see render_phases_general_sets.h
#include "render_phases_general_sets.h"
namespace gdpt {
bool launch_phases_general_set_a(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, unsigned material_mask, hipStream_t stream) {
    if ((material_mask & ~(kSetLambert | 1u << GDPT_MAT_DISNEY_DIFFUSE)) == 0) { launch_phases_set<kSetLambert | 1u << GDPT_MAT_DISNEY_DIFFUSE>(sv, a, grid, stream); return true; }
    if ((material_mask & ~(kSetLambert | 1u << GDPT_MAT_DISNEY_METAL)) == 0) { launch_phases_set<kSetLambert | 1u << GDPT_MAT_DISNEY_METAL>(sv, a, grid, stream); return true; }
    return false;
}
} // namespace gdpt
