// render_kernels.hip — launch dispatch of the GDPT render kernels (device code: render_device.h).
#include "render_wavefront.h"

#include "../capi_common.h"
#include <stdexcept>
#include <string>

namespace gdpt {

const char *render_kernel_name(int rng_scheme) {
    return rng_scheme == GDPT_RNG_TILE ? "gdpt_render_tile_stream_phases" : "gdpt_render_phases";
}

static long long resident_lanes(const RenderLaunch &rl) { return (long long)rl.num_cus * (rl.blocks_per_cu > 0 ? rl.blocks_per_cu : 2) * gd::kBlock; }

// item layout of the persistent kernels (render_device.h: item_to_pixel)
// The plan is made for a band of rl.plan_rows rows (default: the whole film), whatever band is rendered: a pixel's samples
// are cut (and its partial sums merged) the same way by every render that names the same plan_rows, so a sharded render
// equals the unsharded one with that plan bit for bit. (Round 2 always planned for the whole film: a 64-row band of the
// 512x512x256 film, 1/8 of the work, then held 32 k items of 128 samples and took 9.3 ms instead of 3.8 —
// profiles/r03_band_costs.txt.)
static void set_chunks(gd::KernelArgs &a, const RenderLaunch &rl, int W, int rows) {
    const ChunkPlan plan = make_chunk_plan(rl.spp, rl.force_log2k, (long long)W * rl.plan_rows, resident_lanes(rl), rl.plan_take_pct);
    a.num_chunks = plan.n;
    for (int c = 0; c <= plan.n; c++) a.chunk_begin[c] = plan.begin[c];
    a.tiles_x = (W + 15) / 16;
    a.num_slots = (long long)a.tiles_x * ((rows + 15) / 16) * 256;
    a.num_items = a.num_slots * plan.n;
}

int wf_words() { return gd::WF_WORDS; }
int wf_max_generations() { return gd::kWfMaxGen; }
int wf_slot_count(long long num_items) {
    long long n = num_items < (1LL << 21) ? num_items : (1LL << 21);       // 2 M slots = 0.75 GB of path state
    n = (n + gd::kBlock - 1) / gd::kBlock * gd::kBlock;
    return (int)(n < gd::kBlock ? gd::kBlock : n);
}
// layout of the auxiliary block: rays | hits | keys | hist | offsets | overflow stacks (all 16-byte aligned)
namespace {
struct WfAux { size_t rays, hits, keys, hist, offsets, blocks, ovf, total; };
WfAux wf_aux_layout(int slots) {
    WfAux l{};
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
    l.rays = take((size_t)slots * 32); l.hits = take((size_t)slots * 32); l.keys = take((size_t)slots * 8);
    l.hist = take((size_t)gd::kWfBins * 4); l.offsets = take((size_t)gd::kWfBins * 4);
    l.blocks = take((size_t)(slots / gd::kBlock) * 16);
    l.ovf = take((size_t)slots * (size_t)(GDPT_BVH_MAX_DEPTH - gd::kWfLdsLevels) * 4);
    l.total = off;
    return l;
}
}
size_t wf_aux_bytes(int slots) { return wf_aux_layout(slots).total; }

// Generations are enqueued in chunks; the host reads the number of slots that still need a step one chunk behind the
// launches (the speculative chunk behind a finished render finds nothing to do: its kernels return at once).
static void run_wavefront(const DevSceneView &sv, const gd::KernelArgs &a, const RenderLaunch &rl, hipStream_t stream) {
    if (!rl.wf_state || !rl.wf_live || !rl.wf_counters || !rl.wf_host || !rl.wf_aux || rl.wf_slots <= 0) throw std::runtime_error("launch_render: wavefront buffers missing");
    static_assert(GDPT_BVH_MAX_DEPTH > gd::kWfLdsLevels, "overflow stack levels");
    const WfAux lay = wf_aux_layout(rl.wf_slots);
    char *aux = (char *)rl.wf_aux;
    gd::WfBuf w{};
    w.state = rl.wf_state; w.live = rl.wf_live; w.counters = rl.wf_counters; w.n = rl.wf_slots; w.gen = 0;
    w.rays = (float4 *)(aux + lay.rays); w.hits = (float4 *)(aux + lay.hits); w.keys = (uint2 *)(aux + lay.keys);
    w.hist = (unsigned *)(aux + lay.hist); w.offsets = (unsigned *)(aux + lay.offsets);
    w.block_counts = (uint4 *)(aux + lay.blocks); w.totals = rl.counters;
    w.sort = rl.wf_sort; w.tiles_x = (sv.cam.width + 15) / 16;
    for (int k = 0; k < 3; k++) {
        const float lo = rl.wf_bounds[k], hi = rl.wf_bounds[3 + k];
        w.bmin[k] = lo;
        w.cell_scale[k] = (hi > lo) ? 16.0f / (hi - lo) : 0.0f;
    }
    gd::WfTrace t{};
    t.nodes4 = sv.nodes4; t.nodes4q = sv.nodes4q; t.prims = sv.prims; t.spheres = sv.spheres; t.rays = w.rays; t.hits = w.hits; t.live = w.live;
    t.ovf = (int *)(aux + lay.ovf); t.ovf_stride = (unsigned)rl.wf_slots; t.counters = rl.counters;
    t.num_tris = sv.num_tris; t.num_nodes4 = sv.num_nodes4; t.num_spheres = sv.num_spheres; t.search_frac = a.thresh_c; t.count_stats = a.count;
    auto ckh = [](hipError_t e, const char *what) { if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e)); };
    ckh(hipMemsetAsync(rl.wf_counters, 0, sizeof(unsigned) * 3 * gd::kWfMaxGen, stream), "hipMemsetAsync(wavefront counters)");
    ckh(hipMemsetAsync(w.hist, 0, sizeof(unsigned) * gd::kWfBins, stream), "hipMemsetAsync(wavefront histogram)");
    launch_wf_init(w, stream);
    // The trace grid covers the slots that can still hold a ray: all of them until the host has seen a generation with
    // fewer active slots (from then on the work queue is empty and the number only falls). The kernel strides over the
    // queue, so a grid that is too small would still be correct.
    unsigned active_bound = (unsigned)rl.wf_slots;
    const int chunk = 8;
    int gen = 0;
    auto enqueue_chunk = [&]() {
        for (int k = 0; k < chunk; k++, gen++) {
            if (gen >= gd::kWfMaxGen) throw std::runtime_error("launch_render: wavefront generation limit reached");
            w.gen = gen;
            if (rl.lambert_only) launch_wf_step_lambert(sv, a, w, stream); else launch_wf_step_general(sv, a, w, stream);
            launch_wf_sort(w, stream);
            t.count = rl.wf_counters + gen;
            launch_wf_trace(t, sv.num_spheres > 0, (active_bound + gd::kWfTraceBlock - 1) / gd::kWfTraceBlock, stream);
        }
    };
    enqueue_chunk();
    for (;;) {
        ckh(hipMemcpyAsync(rl.wf_host, rl.wf_counters + 2 * gd::kWfMaxGen + (gen - 1), sizeof(unsigned), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(wavefront live count)");
        ckh(hipEventRecord(rl.wf_event, stream), "hipEventRecord");
        const bool room = gen + chunk <= gd::kWfMaxGen;
        if (room) enqueue_chunk();
        ckh(hipEventSynchronize(rl.wf_event), "hipEventSynchronize");
        if (*rl.wf_host == 0u) break;
        if (*rl.wf_host < active_bound) active_bound = *rl.wf_host;
        if (!room) throw std::runtime_error("launch_render: wavefront generation limit reached");
    }
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
            set_chunks(a, rl, W, rows);
            if (a.num_items >= (1LL << 32)) throw std::runtime_error("launch_render: image band too large for the 32-bit work queue");
            // scenes walked from HBM: stack slots = the tree's own bound (host-verified at upload)
            a.stack_levels = rl.wide_stack_need > 0 ? rl.wide_stack_need : GDPT_BVH_MAX_DEPTH;
            a.replay_per_step = rl.replay_per_step >= 1 ? rl.replay_per_step : 4;       // render_twosided.h: kReplayPerStep
            if (a.stack_levels > GDPT_BVH_MAX_DEPTH) throw std::runtime_error("launch_render: traversal stack bound exceeds the builder's maximum");
            a.partials = rl.partials; a.queue_head = rl.queue_head;
            if (!a.partials || !a.queue_head) throw std::runtime_error("launch_render: work-queue buffers missing");
            hipError_t me = hipMemsetAsync(a.queue_head, 0, sizeof(unsigned long long), stream);
            if (me != hipSuccess) throw std::runtime_error("launch_render: queue reset failed");
            const unsigned blocks = persistent_blocks(rl, a.num_items);   // 2 resident blocks per CU (LDS-bound)
            dim3 grid(blocks);
            if (rl.wavefront && !rl.two_sided_machine && !rl.scene_fits_lds) {
                run_wavefront(sv, a, rl, stream);
            } else if (rl.two_sided_machine) {
                if (!rl.bounce_log || rl.bounce_log_bytes < twosided_log_bytes(blocks)) throw std::runtime_error("launch_render: bounce log missing");
                launch_phases_twosided(sv, a, grid, rl.scene_fits_lds && rl.lds_wide, rl.material_mask, rl.bounce_log, stream);
            } else if (rl.lambert_only && rl.stamped && (!rl.scene_fits_lds || rl.lds_wide)) launch_phases_lambert_stamped(sv, a, grid, rl.scene_fits_lds, rl.no_spheres && rl.const_textures, stream);
            else if (rl.lambert_only && rl.no_spheres && (!rl.scene_fits_lds || rl.lds_wide)) launch_phases_lambert_plain(sv, a, grid, rl.scene_fits_lds, rl.const_textures, stream);
            else if (rl.lambert_only) launch_phases_lambert(sv, a, grid, rl.scene_fits_lds, rl.lds_wide, stream);
            else if (!rl.scene_fits_lds && rl.no_spheres && (launch_phases_general_set_a(sv, a, grid, rl.material_mask, stream) || launch_phases_general_set_b(sv, a, grid, rl.material_mask, stream))) {}   // kernel built for the scene's material set
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
        set_chunks(a, rl, W, rows);
        if (a.num_items >= (1LL << 32)) throw std::runtime_error("launch_path_render: image band too large for the 32-bit work queue");
        a.partials = rl.partials; a.queue_head = rl.queue_head;
        if (!a.partials || !a.queue_head) throw std::runtime_error("launch_path_render: work-queue buffers missing");
        hipError_t me = hipMemsetAsync(a.queue_head, 0, sizeof(unsigned long long), stream);
        if (me != hipSuccess) throw std::runtime_error("launch_path_render: queue reset failed");
        long long waves_needed = (a.num_items + 63) / 64;
        long long blocks = (long long)rl.num_cus * (rl.blocks_per_cu > 0 ? rl.blocks_per_cu : 2);
        if (blocks > (waves_needed + 3) / 4) blocks = (waves_needed + 3) / 4;
        if (blocks < 1) blocks = 1;
        launch_path_persistent(sv, a, dim3((unsigned)blocks), rl.scene_fits_lds, rl.lambert_only, rl.no_spheres && rl.const_textures, stream);
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

unsigned persistent_blocks(const RenderLaunch &rl, long long num_items) {
    long long waves_needed = (num_items + 63) / 64;
    long long blocks = (long long)rl.num_cus * (rl.blocks_per_cu > 0 ? rl.blocks_per_cu : 2);
    if (blocks > (waves_needed + 3) / 4) blocks = (waves_needed + 3) / 4;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

// Work items per pixel. A lane processes an item's samples one after the other, so when the queue runs dry the kernel
// still needs as long as its longest unfinished item: with equal chunks of 4+ samples that drain was 0.44 ms of a 2.5 ms
// launch on cbox 512x512x16 (tests/prof_drain.py; bounding the path length barely changed it, so it is the ITEM, not the
// longest path). Chunks therefore shrink along the queue — each takes about 55 % of the samples still unassigned, the
// last ones are single samples — and the queue hands out chunk 0 of every pixel, then chunk 1, ...: long items start
// early, the tail of the launch consists of one-sample items. The rule is applied per ROUND of items, not per chunk (below):
// 16 spp on a 512^2 film -> 5,5,2,2,1,1 (8,5,2,1 until round 3); 64 -> 18,18,8,8,3,3,2,2,1,1; a 64-row band of that film at 128 spp
// -> 8 x 8, 5 x 8, 2 x 8, 1 x 8. Small bands with many samples (multi-GPU row bands) cap the chunk size so that every resident lane
// still sees several items. One 128-byte partial record per item; gdpt_reduce_partials merges a pixel's records in chunk order,
// so the result depends neither on which lane ran what nor on when.
ChunkPlan make_chunk_plan(int spp, int force_log2k, long long pixels, long long lanes, int take_pct_arg) {
    ChunkPlan p{};
    if (spp < 1) spp = 1;
    if (force_log2k >= 0) {                                    // tests: 2^k equal chunks
        int k = force_log2k;
        while (k > 0 && ((1 << k) > spp || (1 << k) > kMaxChunks)) k--;
        p.n = 1 << k;
        for (int c = 0; c <= p.n; c++) p.begin[c] = (int)(((long long)c * spp) >> k);
        return p;
    }
    {   // test knob plan_digits: the chunk sizes spelled out as decimal digits (552211 = 5,5,2,2,1,1), used if they add up to spp
        long long digits = (long long)gdpt::debug_knob("plan_digits", 0.0);
        int sz[18], m = 0, sum = 0;
        for (; digits > 0 && m < 18; digits /= 10) { sz[m] = (int)(digits % 10); sum += sz[m]; if (sz[m] == 0) sum = -1000000; m++; }
        if (m > 0 && sum == spp) {
            p.n = m; p.begin[0] = 0;
            for (int c = 0; c < m; c++) p.begin[c + 1] = p.begin[c] + sz[m - 1 - c];
            return p;
        }
    }
    // at least ~4 items per resident lane, as far as the sample count allows
    long long cap = (long long)spp * pixels / (lanes > 0 ? lanes * 4 : 1);
    if (cap < 1) cap = 1;
    // The queue is chunk-major, so one ROUND of items — one per resident lane — spans lanes / pixels chunks. What is in flight when the
    // queue runs dry is the last round and the stragglers of the ones before it, so the sizes have to fall off per ROUND, not per
    // chunk: a 64-row band of the 512x512 film at 128 spp (pixels = lanes / 4) with the tail "..., 8, 5, 2, 1" had 8-, 5-, 2- and
    // 1-sample items in its last round together — queue dry after 1.95 ms, then 1.0 ms of drain, against 0.39 ms for the whole film at
    // 16 spp (profiles/r03_prof_band.txt). The plan is therefore made for spp / q samples and every chunk of it is laid down q times,
    // q = ceil(rounds * lanes / pixels): every size of the shrinking tail then lasts `rounds` rounds. rounds = 4 (measured: the band
    // above 2.67 -> 2.23 ms, the speed of the whole film; the 512x512 film itself, q = 2: 16 spp -> 5,5,2,2,1,1 instead of 8,5,2,1,
    // +1.3 % at 16 spp and +3 % at 256 spp, profiles/r03_ab_plan_rounds.txt; 6 and 8 rounds no better).
    long long rounds = gdpt::debug_knob_int("plan_rounds", 4);
    rounds = rounds <= 0 ? 4 : std::min(8LL, rounds);                                                    // (test knob; 0 = the default)
    long long q = (pixels > 0 && rounds * lanes > pixels) ? (rounds * lanes + pixels - 1) / pixels : 1;
    if (q > 8) q = 8;
    if ((long long)spp <= q) q = 1;                             // (nothing to shape: the rule below makes single samples)
    const long long slots = kMaxChunks / q;                     // chunks the virtual plan may have (q = 1: kMaxChunks)
    const long long head = q > 1 ? (slots > 4 ? slots - 4 : 1) : kMaxChunks - 8;      // of which for cap-sized chunks (the rest: the shrinking tail)
    const int v = (int)(((long long)spp + q - 1) / q);          // samples of the virtual plan; q * v - spp < q are taken back below
    if ((long long)v > cap * head) cap = ((long long)v + head - 1) / head;             // never more than kMaxChunks chunks
    // share of the unassigned samples a chunk takes, in percent: 55, or what the caller asks for (capi_device.hip: 40 for scenes whose
    // paths are long-tailed — DisneyGlass: an item's time varies more, smaller items balance it, +9..15 %, profiles/r03_sweep_plan_twosided.txt)
    long long take_pct = gdpt::debug_knob_int("plan_shrink", 0);                                         // (test knob; 0 = no override)
    if (take_pct <= 0) take_pct = take_pct_arg > 0 ? take_pct_arg : 55;
    take_pct = std::min(90LL, std::max(10LL, take_pct));
    int sizes[kMaxChunks];
    int rem = v, m = 0;
    while (rem > 0) {
        long long sz = ((long long)rem * take_pct + 99) / 100; // ceil(0.55 * rem)
        if (rem <= 2) sz = 1;                                  // the tail of the queue: single samples
        if (sz > cap) sz = cap;
        if (sz < 1) sz = 1;
        if (m == (int)slots - 1) sz = rem;
        rem -= (int)sz;
        sizes[m++] = (int)sz;
    }
    int excess = (int)(q * v - spp);                            // < q <= 8 samples too many: the last copies of the smallest size above 1 give
    int shrink = -1;                                            // one back each (the sizes stay in descending order); all single: chunks dropped
    for (int c = 0; c < m; c++) if (sizes[c] > 1) shrink = c;
    int n = 0;
    p.begin[0] = 0;
    for (int c = 0; c < m; c++)
        for (int r = 0; r < (int)q; r++) {
            int sz = sizes[c];
            if (c == shrink && r >= (int)q - excess) sz--;
            p.begin[n + 1] = p.begin[n] + sz;
            n++;
        }
    if (shrink < 0) n -= excess;                                // (every chunk a single sample: q * v - spp of them too many)
    p.n = n;
    return p;
}

size_t render_partials_doubles(int width, int rows, int plan_rows, int spp, int force_log2k, long long lanes, int take_pct) {
    const long long tiles = (long long)((width + 15) / 16) * ((rows + 15) / 16);
    return (size_t)16 * (size_t)(tiles * 256) * (size_t)make_chunk_plan(spp, force_log2k, (long long)width * plan_rows, lanes, take_pct).n;
}

bool scene_fits_lds(int num_nodes, int num_prims, int num_tris, int num_materials, int num_lights, int bvh_depth) {
    size_t bytes = (size_t)num_nodes * sizeof(DevBvhNode) + (size_t)num_prims * sizeof(DevPrim) +
                   (size_t)num_tris * sizeof(DevTriShade) + (size_t)num_materials * sizeof(GdptMaterial) + 8 + (size_t)num_lights * 24;
    return bytes <= (size_t)gd::kLdsSceneBytes && bvh_depth <= gd::kLdsSceneLevels && num_nodes > 0;
}

bool scene_fits_lds_wide(int num_nodes4, int num_prims, int num_tris, int num_materials, int num_lights, int wide_stack_need) {
    size_t bytes = (size_t)num_nodes4 * sizeof(DevBvh4Node) + (size_t)num_prims * sizeof(DevPrim) +
                   (size_t)num_tris * sizeof(DevTriShade) + (size_t)num_materials * sizeof(GdptMaterial) + 8 + (size_t)num_lights * 24;
    return bytes <= (size_t)gd::kLdsSceneBytes && wide_stack_need <= gd::kLdsSceneLevels && num_nodes4 > 0;
}

} // namespace gdpt
