/* gdpt_debug.h — test instrument, NOT part of the drop-in boundary of include/gdpt.h.
 *
 * The product entry points take every scheduling decision (which evaluator, how a pixel's samples are cut into work
 * items, whether a small scene is copied to LDS, BVH leaf policy ...) from the scene and the render parameters alone;
 * no environment variable is read anywhere in libgdpt.so. The parity tests, however, must be able to force the
 * alternatives (straight-loop evaluator vs lane machine, 1 / 2 / 8 work items per pixel, HBM walk of an LDS-sized
 * scene, a second BVH over the same triangles) to show that they agree. These two calls are that switchboard: a
 * process-global table of overrides, empty unless a test fills it. Nothing in lajolla, bench.py's timed region or the
 * library itself calls them.
 *
 * Knob names (value semantics in parentheses; an unknown name is an error):
 *   force_eager          (0/1)  straight per-sample loop instead of the lane machine
 *   log2k                (int)  work items (or lanes) per pixel = 2^value
 *   keep_frac            (0..255) trace phase is left when this fraction /256 of its rays is unfinished
 *   search_frac          (0..255) node loop is left when this fraction /256 of the live lanes still searches a leaf
 *   blocks_per_cu        (int)  persistent blocks per compute unit
 *   no_lds_scene         (0/1)  walk an LDS-sized scene from HBM
 *   lds_wide             (0/1)  LDS-resident scenes in BVH4 form (default 1)
 *   no_twosided_machine  (0/1)  two-sided lobes on the straight loop instead of the replay machine
 *   presplit             (real) triangle pre-split budget, extra references per primitive (scene upload)
 *   presplit_floor       (real) pre-split priority floor (scene upload)
 *   bvh_leaf_max         (1..4) SAH builder: primitives per leaf (scene upload)
 *   bvh_leaf_factor      (real) SAH builder: leaf cost factor (scene upload)
 *   sbvh                 (real) SAH builder with spatial splits: extra references allowed per primitive, 0 = object splits only (scene upload)
 *   plan_rounds          (1..8) work-item plan: rounds of items (one per resident lane) every size of the plan's shrinking tail lasts (default 4)
 *   plan_shrink          (10..90) work-item plan: percentage of the samples still unassigned that the next chunk takes (default 55)
 *   plan_digits          (int) work-item plan spelled out: decimal digits = chunk sizes (552211 = 5,5,2,2,1,1), used when they add up to the spp
 *   sbvh_alpha           (real) ... spatial splits are tried where the object split's children overlap by more than this fraction of the root's area
 *   wavefront            (0/1)  scenes walked from HBM, one-sided lobes: 1 = wavefront pipeline (step + trace kernels, path
 *                               state in HBM), 0 = lane machine; the two give bit-identical images
 *   wf_slots             (int)  wavefront pipeline: at most this many path slots (forces slots to run several items)
 *   wf_sort              (0/1/2) wavefront pipeline: ray queue in slot order / sorted by (octant, origin cell) / by (origin cell, octant)
 *   multi_fail_band / multi_fail_stage (int)  gdpt_multi_*: band `multi_fail_band` throws in stage 1 (render), 2 (halo +
 *                               assembly), 3 (all-gather) or 4 (solve): the other bands must stand down, not hang
 *   dct_bk (16 / 32), dct_bm (32 / 64)   GDPT_SOLVER_DCT_MFMA: force a shape of the folded GEMM (default: by grid size)
 *   replay_per_step (n >= 1)    two-sided lane machine: replay iterations of an offset per wave step (default 4; 1 = one per step)
 *   no_plain_kernel (0/1)       one-sided lane machine: the kernel with sphere and texture code even for a triangles-only, constant-texture scene
 *   full_material_switch (0/1)  lane machines: the kernel with the full material switch even when the scene fits a small set
 *   stamps               (0/1)  Lambertian lane machine: the diagnostic build with in-kernel cycle stamps; a render with
 *                               stats then leaves its per-segment wave cycles for gdpt_debug_get_stamps
 */
#ifndef GDPT_DEBUG_H
#define GDPT_DEBUG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Sets one override; returns 0, or non-zero (message in gdpt_last_error) for an unknown name. */
int gdpt_debug_knob_set(const char *name, double value);
/* Wave cycles per segment of the last stamped render (summed over waves): [0] loop head remainder, [1] traversal,
 * [2] hit-vertex rebuild, [3] state arms, [4] BSDF block, [5] offset / finish arm, [6] camera-ray block, [7] wave steps
 * (a count), [8] publishing finished items, [9] work-queue take, [10] item -> pixel mapping, [11] unused; wall clock
 * (s_memrealtime, 100 MHz ticks): [12] first wave started, [13] first wave found the queue empty, [14] last wave ended. */
void gdpt_debug_get_stamps(double out[16]);
/* The work-item plan of the persistent render kernels (make_chunk_plan, csrc/hip/render_kernels.hip): chunk c of a pixel
 * covers samples [begin[c], begin[c+1]). Returns the number of chunks (begin[] gets n + 1 entries), -1 if `capacity` is too
 * small. force_log2k < 0: the product plan. Host only. */
int gdpt_debug_chunk_plan(int spp, int force_log2k, long long film_pixels, long long resident_lanes, int32_t *begin, int capacity);
/* Removes every override: the library is back on its product path. */
void gdpt_debug_knobs_reset(void);

#ifdef __cplusplus
}
#endif
#endif /* GDPT_DEBUG_H */
