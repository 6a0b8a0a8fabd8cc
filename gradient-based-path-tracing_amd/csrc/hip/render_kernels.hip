// render_kernels.hip — launch dispatch of the GDPT render kernels (device code: render_device.h).
#include "render_device.h"

#include <stdexcept>
#include <string>

namespace gdpt {

const char *render_kernel_name(int rng_scheme) {
    return rng_scheme == GDPT_RNG_TILE ? "gdpt_render_tile_stream_phases" : "gdpt_render_phases";
}

void launch_render(const DevSceneView &sv, const RenderLaunch &rl, hipStream_t stream) {
    gd::KernelArgs a{};
    a.spp = rl.spp; a.row_begin = rl.row_begin; a.row_end = rl.row_end; a.max_depth = rl.max_depth;
    a.img = rl.img; a.cx0 = rl.cx0; a.cy0 = rl.cy0; a.cx1 = rl.cx1; a.cy1 = rl.cy1; a.counters = rl.counters;
    // trace phase is left when this fraction (/256) of the rays that entered it is still unfinished
    a.thresh_a = rl.thresh_a >= 0 ? (rl.thresh_a > 255 ? 255 : rl.thresh_a) : 64;
    // ... and its inner-node loop when this fraction of the live rays is still looking for the next leaf
    a.thresh_c = rl.thresh_c >= 0 ? (rl.thresh_c > 255 ? 255 : rl.thresh_c) : 112;
    a.count = rl.count_traversal ? 1 : 0;
    const int W = sv.cam.width, rows = rl.row_end - rl.row_begin;
    if (W <= 0 || rows <= 0 || rl.spp <= 0) throw std::runtime_error("launch_render: empty image band or spp <= 0");
    const bool phases = (rl.one_sided_materials || rl.two_sided_machine) && !rl.force_eager;
    if (rl.shift_mode == GDPT_SHIFT_RECONNECT) {
        // reconnection shift (render_reconnect.hip): straight per-sample loop, K = 2^log2k lanes per pixel
        if (rl.rng_scheme != GDPT_RNG_SAMPLE) throw std::runtime_error("launch_render: GDPT_SHIFT_RECONNECT needs GDPT_RNG_SAMPLE");
        long long pixels = (long long)W * rows;
        int log2k = 0;
        while ((1 << (log2k + 1)) <= rl.spp && log2k < 6 && (pixels << log2k) < (1LL << 19)) log2k++;
        if (rl.force_log2k >= 0) { log2k = rl.force_log2k; while (log2k > 0 && (1 << log2k) > rl.spp) log2k--; }
        a.log2k = log2k;
        int ppb = gd::kBlock >> log2k;
        a.tile_w = ppb >= 16 ? 16 : ppb;
        a.tile_h = ppb / a.tile_w;
        a.tiles_x = (W + a.tile_w - 1) / a.tile_w;
        int tiles_y = (rows + a.tile_h - 1) / a.tile_h;
        launch_reconnect(sv, a, dim3((unsigned)(a.tiles_x * tiles_y)), rl.scene_fits_lds && rl.lds_wide, rl.lambert_only, stream);
    } else if (rl.rng_scheme == GDPT_RNG_TILE) {
        int ntx = (W + 15) / 16, nty = (sv.cam.height + 15) / 16;
        dim3 grid((unsigned)((ntx * nty + 63) / 64));
        if (!phases) launch_tile_eager(sv, a, grid, ntx, nty, stream);
        else if (rl.lambert_only) launch_tile_phases_lambert(sv, a, grid, ntx, nty, stream);
        else launch_tile_phases_general(sv, a, grid, ntx, nty, stream);
    } else if (rl.rng_scheme == GDPT_RNG_SAMPLE) {
        if (!phases) {
            // eager evaluator: static mapping, K = 2^log2k lanes per pixel
            long long pixels = (long long)W * rows;
            int log2k = 0;
            while ((1 << (log2k + 1)) <= rl.spp && log2k < 6 && (pixels << log2k) < (1LL << 19)) log2k++;
            if (rl.force_log2k >= 0) { log2k = rl.force_log2k; while (log2k > 0 && (1 << log2k) > rl.spp) log2k--; }
            a.log2k = log2k;
            int ppb = gd::kBlock >> log2k;                 // pixels per block
            a.tile_w = ppb >= 16 ? 16 : ppb;
            a.tile_h = ppb / a.tile_w;
            a.tiles_x = (W + a.tile_w - 1) / a.tile_w;
            int tiles_y = (rows + a.tile_h - 1) / a.tile_h;
            launch_eager(sv, a, dim3((unsigned)(a.tiles_x * tiles_y)), stream);
        } else {
            // persistent lanes pulling (pixel, chunk) items: >= 4 samples per item, at most 8 items per pixel
            a.log2c = render_log2_chunks(rl.spp, rl.force_log2k, (long long)W * rows);
            a.tiles_x = (W + 15) / 16;
            const long long tiles = (long long)a.tiles_x * ((rows + 15) / 16);
            a.num_items = (tiles * 256) << a.log2c;
            if (a.num_items >= (1LL << 32)) throw std::runtime_error("launch_render: image band too large for the 32-bit work queue");
            a.partials = rl.partials; a.queue_head = rl.queue_head;
            if (!a.partials || !a.queue_head) throw std::runtime_error("launch_render: work-queue buffers missing");
            hipError_t me = hipMemsetAsync(a.queue_head, 0, sizeof(unsigned long long), stream);
            if (me != hipSuccess) throw std::runtime_error("launch_render: queue reset failed");
            const unsigned blocks = persistent_blocks(rl, a.num_items);   // 2 resident blocks per CU (LDS-bound)
            dim3 grid(blocks);
            if (rl.two_sided_machine) {
                if (!rl.bounce_log || rl.bounce_log_bytes < twosided_log_bytes(blocks)) throw std::runtime_error("launch_render: bounce log missing");
                launch_phases_twosided(sv, a, grid, rl.scene_fits_lds && rl.lds_wide, rl.bounce_log, stream);
            } else if (rl.lambert_only && rl.stamped && (!rl.scene_fits_lds || rl.lds_wide)) launch_phases_lambert_stamped(sv, a, grid, rl.scene_fits_lds, stream);
            else if (rl.lambert_only) launch_phases_lambert(sv, a, grid, rl.scene_fits_lds, rl.lds_wide, stream);
            else launch_phases_general(sv, a, grid, rl.scene_fits_lds, rl.lds_wide, stream);
            launch_reduce_partials(sv, a, stream);
        }
    } else {
        throw std::runtime_error("launch_render: unknown rng_scheme");
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) throw std::runtime_error(std::string("render kernel launch failed: ") + hipGetErrorString(e));
}

// Integrator::Path (render_path.hip): img only.
void launch_path_render(const DevSceneView &sv, const RenderLaunch &rl, hipStream_t stream) {
    gd::KernelArgs a{};
    a.spp = rl.spp; a.row_begin = rl.row_begin; a.row_end = rl.row_end; a.max_depth = rl.max_depth;
    a.img = rl.img; a.counters = rl.counters;
    a.count = rl.count_traversal ? 1 : 0;
    const int W = sv.cam.width, rows = rl.row_end - rl.row_begin;
    if (W <= 0 || rows <= 0 || rl.spp <= 0) throw std::runtime_error("launch_path_render: empty image band or spp <= 0");
    if (rl.rng_scheme == GDPT_RNG_TILE) {
        int ntx = (W + 15) / 16, nty = (sv.cam.height + 15) / 16;
        launch_tile_path(sv, a, dim3((unsigned)((ntx * nty + 63) / 64)), ntx, nty, stream);
    } else if (rl.rng_scheme == GDPT_RNG_SAMPLE && !rl.force_eager) {
        // persistent lanes pulling (pixel, chunk) items, as the GradPath kernel does
        a.thresh_a = rl.thresh_a >= 0 ? (rl.thresh_a > 255 ? 255 : rl.thresh_a) : 64;
        a.thresh_c = rl.thresh_c >= 0 ? (rl.thresh_c > 255 ? 255 : rl.thresh_c) : 112;
        a.log2c = render_log2_chunks(rl.spp, rl.force_log2k, (long long)W * rows);
        a.tiles_x = (W + 15) / 16;
        const long long tiles = (long long)a.tiles_x * ((rows + 15) / 16);
        a.num_items = (tiles * 256) << a.log2c;
        if (a.num_items >= (1LL << 32)) throw std::runtime_error("launch_path_render: image band too large for the 32-bit work queue");
        a.partials = rl.partials; a.queue_head = rl.queue_head;
        if (!a.partials || !a.queue_head) throw std::runtime_error("launch_path_render: work-queue buffers missing");
        hipError_t me = hipMemsetAsync(a.queue_head, 0, sizeof(unsigned long long), stream);
        if (me != hipSuccess) throw std::runtime_error("launch_path_render: queue reset failed");
        long long waves_needed = (a.num_items + 63) / 64;
        long long blocks = (long long)rl.num_cus * (rl.blocks_per_cu > 0 ? rl.blocks_per_cu : 2);
        if (blocks > (waves_needed + 3) / 4) blocks = (waves_needed + 3) / 4;
        if (blocks < 1) blocks = 1;
        launch_path_persistent(sv, a, dim3((unsigned)blocks), rl.scene_fits_lds, rl.lambert_only, stream);
    } else if (rl.rng_scheme == GDPT_RNG_SAMPLE) {
        // straight per-sample loop (A/B checks): static mapping, K = 2^log2k lanes per pixel
        long long pixels = (long long)W * rows;
        int log2k = 0;
        while ((1 << (log2k + 1)) <= rl.spp && log2k < 6 && (pixels << log2k) < (1LL << 19)) log2k++;
        if (rl.force_log2k >= 0) { log2k = rl.force_log2k; while (log2k > 0 && (1 << log2k) > rl.spp) log2k--; }
        a.log2k = log2k;
        int ppb = gd::kBlock >> log2k;
        a.tile_w = ppb >= 16 ? 16 : ppb;
        a.tile_h = ppb / a.tile_w;
        a.tiles_x = (W + a.tile_w - 1) / a.tile_w;
        int tiles_y = (rows + a.tile_h - 1) / a.tile_h;
        launch_path(sv, a, dim3((unsigned)(a.tiles_x * tiles_y)), stream);
    } else {
        throw std::runtime_error("launch_path_render: unknown rng_scheme");
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) throw std::runtime_error(std::string("path kernel launch failed: ") + hipGetErrorString(e));
}

// Work items per pixel = 2^log2c chunks of its sample range: at least 4 samples per item, at most 8 items per pixel on
// large films; small bands with many samples (multi-GPU row bands) are cut finer, towards ~2^20 items per launch, so
// that every lane still sees several items and the drain at the end of the kernel stays one short item long.
unsigned persistent_blocks(const RenderLaunch &rl, long long num_items) {
    long long waves_needed = (num_items + 63) / 64;
    long long blocks = (long long)rl.num_cus * (rl.blocks_per_cu > 0 ? rl.blocks_per_cu : 2);
    if (blocks > (waves_needed + 3) / 4) blocks = (waves_needed + 3) / 4;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

int render_log2_chunks(int spp, int force_log2k, long long pixels) {
    int by_spp = 0;
    while ((8 << by_spp) <= spp) by_spp++;                      // 2^by_spp <= spp / 4
    int target = 0;
    while (target < 8 && (pixels << target) < (1LL << 20)) target++;
    int log2c = by_spp < (target > 3 ? target : 3) ? by_spp : (target > 3 ? target : 3);
    if (force_log2k >= 0) { log2c = force_log2k; while (log2c > 0 && (1 << log2c) > spp) log2c--; }
    return log2c;
}

size_t render_partials_doubles(int width, int rows, int spp, int force_log2k) {
    const long long tiles = (long long)((width + 15) / 16) * ((rows + 15) / 16);
    return (size_t)16 * (size_t)((tiles * 256) << render_log2_chunks(spp, force_log2k, (long long)width * rows));
}

bool scene_fits_lds(int num_nodes, int num_prims, int num_tris, int num_materials, int bvh_depth) {
    size_t bytes = (size_t)num_nodes * sizeof(DevBvhNode) + (size_t)num_prims * sizeof(DevPrim) +
                   (size_t)num_tris * sizeof(DevTriShade) + (size_t)num_materials * sizeof(GdptMaterial);
    return bytes <= (size_t)gd::kLdsSceneBytes && bvh_depth <= gd::kLdsSceneLevels && num_nodes > 0;
}

bool scene_fits_lds_wide(int num_nodes4, int num_prims, int num_tris, int num_materials, int wide_stack_need) {
    size_t bytes = (size_t)num_nodes4 * sizeof(DevBvh4Node) + (size_t)num_prims * sizeof(DevPrim) +
                   (size_t)num_tris * sizeof(DevTriShade) + (size_t)num_materials * sizeof(GdptMaterial);
    return bytes <= (size_t)gd::kLdsSceneBytes && wide_stack_need <= gd::kLdsSceneLevels && num_nodes4 > 0;
}

} // namespace gdpt
