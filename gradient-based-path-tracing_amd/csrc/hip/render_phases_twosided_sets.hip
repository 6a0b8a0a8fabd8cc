// Two-sided lane machine built for a small material set (render_twosided.h: kSetGlass), scenes walked from HBM.
// (A {Lambertian, DisneyBSDF} build was measured too: 197 instead of 318 spilled VGPRs and 6 % SLOWER than the full switch
// on the same box, tests/ab_twosided.py — spills are not what binds these kernels — so DisneyBSDF scenes keep the full one.
// Round 3, with the shared BSDF block: 229 instead of 278 spilled VGPRs, 56 k instead of 92 k instructions, and again 6-7 % slower
// (226-228 against 239-244 Msamples/s, same process, profiles/r03_ab_bsdf_set_kernel.txt); instruction-cache misses of the full
// kernel: 0.9 % of its fetches. Neither code size nor spill count predicts these kernels.)
// (The same kernels without the sphere test and the sphere shading frame, for scenes made of triangles: 40 / 217 instead of
// 43 / 225 spilled VGPRs, glass within noise, DisneyBSDF +1..3 % in a same-process A/B — not kept.)
#include "render_twosided.h"
namespace gdpt {
void launch_phases_twosided_glass(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, void *bounce_log, hipStream_t stream) {
    hipLaunchKernelGGL((gd::gdpt_render_twosided<false, gd::kSetGlass>), grid, dim3(gd::kBlock), gd::hbm_dynamic_lds(a), stream, sv, a, (gd::BounceLog *)bounce_log);
}
} // namespace gdpt
