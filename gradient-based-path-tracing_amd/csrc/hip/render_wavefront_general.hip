// Wavefront pipeline (render_wavefront.h): step kernel with the full one-sided material switch.
#include "render_wavefront.h"
namespace gdpt {
void launch_wf_step_general(const DevSceneView &sv, const gd::KernelArgs &a, const gd::WfBuf &w, hipStream_t stream) {
    hipLaunchKernelGGL((gd::gdpt_wf_step<false>), dim3((unsigned)(w.n / gd::kBlock)), dim3(gd::kBlock), 0, stream, sv, a, w);
}
} // namespace gdpt
