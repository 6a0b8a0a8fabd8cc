// Wavefront pipeline (render_wavefront.h), Lambertian-only scenes: step kernel with the cosine lobe inlined, the
// trace kernel and the slot initialiser.
#define GDPT_BUILD_WF_TRACE 1
#include "render_wavefront.h"
namespace gd {
// every slot starts idle and without an item
__global__ __launch_bounds__(kBlock) void gdpt_wf_init(WfBuf w) {
    const long long slot = (long long)blockIdx.x * kBlock + threadIdx.x;
    w.state[(long long)WF_I0 * w.n + slot] = pack2((unsigned)S_DONE, 0u);
    w.state[(long long)WF_ITEM * w.n + slot] = pack2(0xFFFFFFFFu, 0xFFFFFFFFu);
}
} // namespace gd
namespace gdpt {
void launch_wf_init(const gd::WfBuf &w, hipStream_t stream) {
    hipLaunchKernelGGL(gd::gdpt_wf_init, dim3((unsigned)(w.n / gd::kBlock)), dim3(gd::kBlock), 0, stream, w);
}
void launch_wf_step_lambert(const DevSceneView &sv, const gd::KernelArgs &a, const gd::WfBuf &w, hipStream_t stream) {
    hipLaunchKernelGGL((gd::gdpt_wf_step<true>), dim3((unsigned)(w.n / gd::kBlock)), dim3(gd::kBlock), 0, stream, sv, a, w);
}
void launch_wf_trace(const DevSceneView &sv, const gd::KernelArgs &a, const gd::WfBuf &w, unsigned blocks, hipStream_t stream) {
    hipLaunchKernelGGL(gd::gdpt_wf_trace, dim3(blocks), dim3(gd::kBlock), 0, stream, sv, a, w);
}
} // namespace gdpt
