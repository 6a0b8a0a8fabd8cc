// Wavefront pipeline (render_wavefront.h), Lambertian-only scenes: step kernel with the cosine lobe inlined, the
// sort and trace kernels and the slot initialiser.
#define GDPT_BUILD_WF_TRACE 1
#include "render_wavefront.h"
namespace gd {
// every slot starts idle and without an item
__global__ __launch_bounds__(kBlock) void gdpt_wf_init(WfBuf w) {
    const long long slot = (long long)blockIdx.x * kBlock + threadIdx.x;
    w.state[(long long)WF_I0 * w.n + slot] = pack2((unsigned)S_DONE, 0u);
    w.state[(long long)WF_ITEM * w.n + slot] = 0xFFFFFFFFull;
    w.keys[slot] = make_uint2(kWfNoKey, 0u);
}
} // namespace gd
namespace gdpt {
void launch_wf_init(const gd::WfBuf &w, hipStream_t stream) {
    hipLaunchKernelGGL(gd::gdpt_wf_init, dim3((unsigned)(w.n / gd::kBlock)), dim3(gd::kBlock), 0, stream, w);
}
void launch_wf_step_lambert(const DevSceneView &sv, const gd::KernelArgs &a, const gd::WfBuf &w, hipStream_t stream) {
    hipLaunchKernelGGL((gd::gdpt_wf_step<true>), dim3((unsigned)(w.n / gd::kBlock)), dim3(gd::kBlock), 0, stream, sv, a, w);
}
// totals of the generation (always) + the counting sort's scan and scatter (sorted queues)
void launch_wf_sort(const gd::WfBuf &w, hipStream_t stream) {
    hipLaunchKernelGGL(gd::gdpt_wf_scan, dim3(1), dim3(gd::kWfScanBlock), 0, stream, w);
    if (w.sort != gd::WF_SORT_NONE) hipLaunchKernelGGL(gd::gdpt_wf_scatter, dim3((unsigned)(w.n / gd::kBlock)), dim3(gd::kBlock), 0, stream, w);
}
void launch_wf_trace(const gd::WfTrace &t, bool spheres, unsigned blocks, hipStream_t stream) {
    const size_t lds = (size_t)gd::kWfLdsLevels * gd::kWfTraceBlock * sizeof(int);
    if (t.count_stats) {
        if (spheres) hipLaunchKernelGGL((gd::gdpt_wf_trace<true, true>), dim3(blocks), dim3(gd::kWfTraceBlock), lds, stream, t);
        else hipLaunchKernelGGL((gd::gdpt_wf_trace<false, true>), dim3(blocks), dim3(gd::kWfTraceBlock), lds, stream, t);
    } else if (spheres) hipLaunchKernelGGL((gd::gdpt_wf_trace<true, false>), dim3(blocks), dim3(gd::kWfTraceBlock), lds, stream, t);
    else hipLaunchKernelGGL((gd::gdpt_wf_trace<false, false>), dim3(blocks), dim3(gd::kWfTraceBlock), lds, stream, t);
}
} // namespace gdpt
