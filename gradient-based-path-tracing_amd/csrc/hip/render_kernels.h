// render_kernels.h — launch interface of the GDPT render kernels (host side of render_kernels.hip).
#pragma once
#include "../device_scene.h"
#include <hip/hip_runtime.h>

namespace gdpt {

// A pixel's samples are cut into work items ("chunks"): chunk c covers samples [begin[c], begin[c+1]).
constexpr int kMaxChunks = 64;
struct ChunkPlan { int n; int begin[kMaxChunks + 1]; };

struct RenderCounters {            // device-resident, zeroed per render
    unsigned long long rays, bounces, nonfinite, nodes, prims;
    unsigned long long wave_node_trips, wave_leaf_trips, wave_steps, lane_steps;   // counting builds: SIMT utilisation
    unsigned long long stamps[16];  // diagnostic build (knob "stamps"): wave cycles per segment, render_device.h SEG_*
};

struct RenderLaunch {
    int spp;
    int rng_scheme;                // GDPT_RNG_*
    int row_begin, row_end;
    int plan_rows;                 // the work-item plan is made for a band of this many rows (GdptRenderParams::plan_rows, resolved)
    int max_depth;                 // effective (scene value or override)
    double *img, *cx0, *cy0, *cx1, *cy1;   // device, W*H*3 each
    RenderCounters *counters;      // device
    bool count_traversal;          // counting build: BVH nodes / primitives per ray
    int shift_mode;                // GDPT_SHIFT_*: 0 = the reference's offsets (parity mode), 1 = reconnection shift
    // scene classification (decided at upload) and tuning knobs
    bool one_sided_materials;      // no DisneyGlass / DisneyBSDF: the phase machine with lazy offsets is exact
    bool lambert_only;             // every material is Lambertian
    bool scene_fits_lds;           // BVH nodes + primitive records fit the block's LDS copy
    bool force_eager;              // run the eager evaluator regardless (checks / A-B runs)
    int thresh_a, thresh_c;        // trace-phase exit fractions /256 (unfinished rays; lanes still searching a leaf), -1 = default
    int force_log2k;               // lanes per pixel = 2^force_log2k (-1 = automatic)
    bool lds_wide;                 // LDS-resident scene walked in its BVH4 form
    int wide_stack_need;           // traversal-stack bound of the tree the HBM kernels walk (host-verified)
    int num_materials;
    unsigned material_mask;        // bit t = some material of the scene has type t (selects kernels built for small sets)
    bool two_sided_machine;        // two-sided lobes present (and no rough ones): lane machine with replayed offsets
    void *bounce_log;              // its per-lane log (device), sized by twosided_log_bytes(blocks)
    size_t bounce_log_bytes;
    int num_cus;                   // compute units of the device (persistent grid size)
    int blocks_per_cu;             // persistent blocks per CU (0 = default 2)
    // wavefront pipeline (render_wavefront.h): scenes walked from HBM, one-sided lobes, SAMPLE streams
    bool wavefront;
    unsigned long long *wf_state;  // device, wf_words() * wf_slots 8-byte words
    unsigned *wf_live;             // device, wf_slots
    void *wf_aux;                  // device, wf_aux_bytes(wf_slots): ray / hit records, sort keys, histogram, overflow stacks
    int wf_sort;                   // 0 = queue in slot order, 1 = sorted (octant, origin cell), 2 = sorted (origin cell, octant)
    float wf_bounds[6];            // scene bounds (min xyz, max xyz) for the origin cells of the sort key
    unsigned *wf_counters;         // device, 3 * wf_max_generations()
    unsigned *wf_host;             // pinned, 1 word (live-count read-back)
    hipEvent_t wf_event;
    int wf_slots;
    int replay_per_step;           // two-sided lane machine (render_twosided.h), 0 = default
    bool no_spheres, const_textures;   // triangles only / every texture constant: kernels built without sphere / texture code
    bool stamped;                  // diagnostic build with in-kernel cycle stamps (test-only knob "stamps")
    int plan_take_pct;             // work-item plan: share of the unassigned samples a chunk takes, percent (0 = default; scenes of long-tailed paths take less)
    double *partials;              // device, >= 15 * W * rows * 8 doubles (work-item partial sums)
    unsigned long long *queue_head;// device, work-queue head
};
bool scene_fits_lds(int num_nodes, int num_prims, int num_tris, int num_materials, int num_lights, int bvh_depth);
bool scene_fits_lds_wide(int num_nodes4, int num_prims, int num_tris, int num_materials, int num_lights, int wide_stack_need);

// Doubles the `partials` buffer must hold for a band of `pixels` pixels at `spp`.
size_t twosided_log_bytes(unsigned blocks);
unsigned persistent_blocks(const RenderLaunch &rl, long long num_items);
void launch_path_render(const DevSceneView &sv, const RenderLaunch &rl, hipStream_t stream);
// Chunk sizes shrink along the queue (about 40 % of what is left each time, ending in single samples) unless
// force_log2k >= 0 asks for 2^k equal chunks (tests). `lanes` = resident lanes of the persistent grid.
ChunkPlan make_chunk_plan(int spp, int force_log2k, long long pixels, long long lanes, int take_pct = 0);    // take_pct: share of the unassigned samples a chunk takes (0 = the default 55)
size_t render_partials_doubles(int width, int rows, int plan_rows, int spp, int force_log2k, long long lanes, int take_pct = 0);

int wf_words();
int wf_max_generations();
// Path slots of the wavefront pipeline for a band of `num_items` work items.
int wf_slot_count(long long num_items);
size_t wf_aux_bytes(int slots);
// Enqueues the five-buffer render on `stream`. Throws std::runtime_error on a launch failure.
void launch_render(const DevSceneView &sv, const RenderLaunch &rl, hipStream_t stream);
// Name of the dominant kernel of the last launch configuration (for rocprof matching).
const char *render_kernel_name(int rng_scheme);

} // namespace gdpt
