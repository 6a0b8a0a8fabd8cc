// scene_internal.h — the device-resident scene handle and the render entry shared by the C-ABI translation units
// (capi_device.hip: single-device entry points; multi_gpu.hip: the row-band host of several devices).
#pragma once
#include "../../../include/gdpt.h"
#include "../device_scene.h"
#include "render_kernels.h"

#include <hip/hip_runtime.h>

#include <stdexcept>
#include <string>
#include <vector>

namespace gdpt {
inline void ck(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
} // namespace gdpt

struct GdptScene {
    int device = 0;
    DevSceneView view{};
    int bvh_depth = 0;
    int wide_stack_need = 0;       // stack bound of the BVH4 (LDS-resident scenes)
    int wide8_stack_need = 0;      // stack bound of the BVH8 (scenes walked from HBM)
    bool has_envmap = false;
    int scene_spp = 0;             // <sampler sampleCount> of the description (default spp)
    bool one_sided = true, lambert_only = true;
    unsigned material_mask = 0;    // bit t = a material of type t is present
    int plan_take_pct = 0;         // work-item plan: share of the unassigned samples a chunk takes (0 = default 55; 40 where a refractive lobe is present)
    bool has_rough = false;        // RoughPlastic / RoughDielectric present: GradPath uses the evaluator built with those lobes
    std::vector<void *> allocations;
    // cached output/work buffers for the host-pointer entry points
    double *d_buf[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t buf_elems = 0;
    gdpt::RenderCounters *d_counters = nullptr;
    gdpt::RenderCounters *h_counters = nullptr; // pinned
    void *d_bounce_log = nullptr; size_t bounce_log_bytes = 0;   // per-lane bounce log of the two-sided lane machine
    double *d_partials = nullptr; size_t partials_doubles = 0;   // work-item partial sums of the persistent render kernel
    unsigned long long *d_queue = nullptr;
    // wavefront pipeline (render_wavefront.h): path state, live list, generation counters
    unsigned long long *d_wf_state = nullptr; unsigned *d_wf_live = nullptr, *d_wf_counters = nullptr, *h_wf_word = nullptr;
    void *d_wf_aux = nullptr;      // ray / hit records, sort keys and histogram, overflow stacks (render_kernels.hip: wf_aux_layout)
    int wf_slots = 0;
    float bounds[6] = {0, 0, 0, 0, 0, 0};   // fp32 scene bounds (min xyz, max xyz), as get_intersection_epsilon sees them
    hipEvent_t wf_event = nullptr;
    int num_cus = 256;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    template <class T>
    T *keep(T *p) { if (p) allocations.push_back((void *)p); return p; }

    void ensure_buffers(size_t elems) {
        if (elems <= buf_elems) return;
        for (auto &b : d_buf) { if (b) hipFree(b); b = nullptr; }
        for (auto &b : d_buf) gdpt::ck(hipMalloc((void **)&b, elems * sizeof(double)), "hipMalloc(image buffers)");
        buf_elems = elems;
    }
    ~GdptScene() {
        hipSetDevice(device);
        for (void *p : allocations) hipFree(p);
        for (auto &b : d_buf) if (b) hipFree(b);
        if (d_counters) hipFree(d_counters);
        if (d_partials) hipFree(d_partials);
        if (d_bounce_log) hipFree(d_bounce_log);
        if (d_queue) hipFree(d_queue);
        if (d_wf_state) hipFree(d_wf_state);
        if (d_wf_live) hipFree(d_wf_live);
        if (d_wf_aux) hipFree(d_wf_aux);
        if (d_wf_counters) hipFree(d_wf_counters);
        if (h_wf_word) hipHostFree(h_wf_word);
        if (wf_event) hipEventDestroy(wf_event);
        if (h_counters) hipHostFree(h_counters);
        if (ev0) hipEventDestroy(ev0);
        if (ev1) hipEventDestroy(ev1);
    }
};


namespace gdpt {
// Own BVH build + HBM upload of a flattened scene (replaces Scene::Scene, src/scene.cpp:4-53).
void build_scene(const GdptSceneDesc *desc, int device, GdptScene *sc);
// Enqueues one five-buffer render of rows [params->row_begin, row_end) on `stream`; waits only when `stats` is given.
void render_device_impl(GdptScene *sc, const GdptRenderParams *params, int scene_spp,
                        double *img, double *cx0, double *cy0, double *cx1, double *cy1,
                        hipStream_t stream, GdptRenderStats *stats);
} // namespace gdpt
