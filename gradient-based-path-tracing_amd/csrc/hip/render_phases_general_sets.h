// One-sided lane machine built for the material set of the scene (scenes walked from HBM, triangles only): the Disney
// single-lobe scenes hold {Lambertian, one Disney lobe}, and a kernel whose material switch has only those two arms — and no
// sphere code — needs fewer registers than the one built for every one-sided lobe (88 spilled VGPRs). The set travels in the
// bits of the PLAIN flag above kPlainSetShift (render_device.h: lane_consume); same arithmetic on every path such a scene
// can take, so the buffers are bit-identical to the full switch's (debug knob full_material_switch, tests/test_gpu_render_parity.py).
#pragma once
#include "render_device.h"
namespace gdpt {
template <unsigned SET>
inline void launch_phases_set(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, hipStream_t stream) {
    constexpr int P = (int)(SET << gd::kPlainSetShift) | gd::kPlainNoSpheres;
    hipLaunchKernelGGL((gd::gdpt_render_phases<false, false, true, true, false, P>), grid, dim3(gd::kBlock), gd::hbm_dynamic_lds(a), stream, sv, a);
}
constexpr unsigned kSetLambert = 1u << GDPT_MAT_LAMBERTIAN;
} // namespace gdpt
