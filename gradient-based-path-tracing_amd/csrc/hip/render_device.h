// render_device.h — the GDPT sample evaluator and tile loop for gfx950.
//
// Restates gradient_path_render's tile loop (src/render.cpp:277-331) and grad_path_tracing
// (src/path_tracing.h:354-1050): one base path plus four pixel-offset paths per sample, BSDF sampling
// only (no NEE/MIS on this integrator), Russian roulette from rr_depth, per-pixel accumulation of
// img, cx0, cy0, cx1, cy1. The four undefined end-of-bounce reads (src/path_tracing.h:1007-1010) use
// the "A-semantics" of SURVEY.md §8(a) G2: an offset keeps its primary hit for the whole path.
//
// Two evaluators live here:
//
//  * phase machine (scenes whose materials are all one-sided lobes: Lambertian, DisneyDiffuse/Metal/
//    Clearcoat/Sheen — cbox, sponza). A sample is cut into three phases:
//        A  base primary ray + the first bounce iteration (2 rays)
//        B  one later bounce iteration (1 ray)
//        C  the four offset paths (4 rays) + accumulation of the sample's record
//    Offsets never feed back into the base path or the RNG stream (they reuse the base's numbers), and for
//    one-sided lobes an offset that survives bounce 1 is retired by bounce 2 (its next sample_bsdf sees
//    dir_in below the surface, or the material test fails). So an offset is observable only when the base
//    path leaves the loop during bounce 1, or at the p2<=0 exit of bounce 2, and phase C runs only then —
//    with bounce-1 random numbers re-derived from the sample's PCG stream. Each lane owns (pixel, chunk of
//    samples); a wave runs a phase when enough lanes wait for it (ballot counts), so lanes whose path has
//    ended pick up their next sample instead of idling behind the longest path of the wave.
//    Exactness: identical to the eager evaluation except when dot(geometric_normal, dir) of a surviving
//    offset is exactly 0 while its pdf is > 0 (then the reference would carry it one more bounce).
//
//  * eager evaluator (any scene; used when a two-sided lobe — DisneyGlass, DisneyBSDF — is present, and by
//    the checks): the straight loop with offset state carried in private memory.
//
// Traversal is fp32 with a per-lane stack in LDS (and, for small scenes, BVH nodes + primitive records
// resident in LDS); shading is fp64 with per-lane PCG32 state in registers. K lanes per pixel sit next to each
// other in a wave and are combined with a fixed-order xor-shuffle tree: one writer per pixel, no atomics,
// run-to-run deterministic.
#pragma once
#include "render_kernels.h"
#include "device_trace.h"

namespace gd {

constexpr int kBlock = 256;
constexpr int kLdsSceneBytes = 20 * 1024;     // nodes + prims + shading table + materials of a "small" scene
constexpr int kLdsSceneLevels = 12;           // stack slots per lane when the scene is LDS-resident

struct LaneCounters { unsigned rays, bounces, nonfinite; };

// Diagnostic builds only (render_phases_diag.hip, selected by the test-only knob "stamps", include/gdpt_debug.h): where a
// wave's cycles go, segment by segment. s_memtime is read by the wave (scalar), so a segment's share includes what the
// lanes that sit it out wait for — the point of the exercise. The product kernels are built with ON = false: every call
// below compiles to nothing. Stamp values only ever reach RenderCounters::stamps, never an output image.
enum { SEG_QUEUE = 0, SEG_TRACE = 1, SEG_VERTEX = 2, SEG_CONSUME = 3, SEG_BSDF = 4, SEG_FINISH = 5, SEG_CAMERA = 6, SEG_STEPS = 7, SEG_PUBLISH = 8, SEG_TAKE = 9, SEG_ITEM = 10, SEG_COUNT = 12, SEG_T_START = 12, SEG_T_DRY = 13, SEG_T_END = 14, SEG_SLOTS = 16 };
// The accumulators belong to the WAVE, not to a lane: they live in LDS (`slot` = the wave's row of 13 words) and every mark
// is booked by the first ACTIVE lane. (Per-lane accumulators, as first built, book a step that a lane sits out onto that
// lane's next mark: read from lane 0 they showed 13 % of the wave's time in "publish", which really takes 5 %.)
template <bool ON> struct Stamps {
    unsigned long long *slot;                     // [SEG_COUNT] accumulators + [SEG_COUNT] = time of the previous mark
    GD void start(unsigned long long *wave_row) {
        if (ON) { slot = wave_row; if ((threadIdx.x & 63) == 0) { for (int i = 0; i < SEG_COUNT; i++) slot[i] = 0; slot[SEG_COUNT] = __builtin_amdgcn_s_memtime(); } }
    }
    GD void mark(int seg) {                       // time since the previous mark belongs to `seg`
        if (ON) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the stamp has landed
            if ((int)(threadIdx.x & 63) == __ffsll((unsigned long long)__ballot(1)) - 1) { slot[seg] += t - slot[SEG_COUNT]; slot[SEG_COUNT] = t; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    GD void tick(int seg) { if (ON) { if ((int)(threadIdx.x & 63) == __ffsll((unsigned long long)__ballot(1)) - 1) slot[seg] += 1; } }
    GD unsigned long long get(int seg) const { return ON ? slot[seg] : 0ull; }
};
struct Accum { D3 r, dx0, dy0, dx1, dy1; };

struct KernelArgs {
    int spp, log2k, tile_w, tile_h, tiles_x;
    int row_begin, row_end, max_depth;
    int thresh_a, thresh_c;
    int count;
    int stack_levels;                  // scenes walked from HBM: traversal-stack slots per lane in dynamic LDS (= the tree's bound)
    int replay_per_step;               // two-sided lane machine: replay iterations of an offset per wave step (>= 1)
    int num_chunks;                    // work items per pixel: chunk c covers samples [chunk_begin[c], chunk_begin[c+1])
    long long num_slots;               // pixel slots of the band: tiles * 256 (ragged edge tiles keep all 256)
    long long num_items;               // num_slots * num_chunks (< 2^32); item = chunk * num_slots + slot ("tier-major")
    double *partials;                  // [num_items][16]: 15 sums (r, dx0, dy0, dx1, dy1 as xyz) + pad, 128-B records
    int chunk_begin[gdpt::kMaxChunks + 1];
    unsigned long long *queue_head;    // work-queue head (zeroed per launch)
    double *img, *cx0, *cy0, *cx1, *cy1;
    gdpt::RenderCounters *counters;
};

GD bool loop_allows(int max_depth, int num_vertices) { return max_depth == -1 || num_vertices <= max_depth + 1; }   // :515

// ------------------------------------------------------------------------------------------------
// accumulation of one GraidentPTRadiance into the pixel sums (src/render.cpp:311-318)
// ------------------------------------------------------------------------------------------------
// Two homes for a lane's 15 running sums: registers (serial kernels) or the block's LDS, one private slot per lane
// and component, updated with return-less ds_add_f64 (frees 30 VGPRs; one writer per slot, so still deterministic).
struct AccReg {
    Accum a;
    GD void init() { a.r = a.dx0 = a.dy0 = a.dx1 = a.dy1 = splat(0); }
    GD void add(int which, D3 v) {     // which: 0 r, 1 dx0, 2 dy0, 3 dx1, 4 dy1
        if (which == 0) a.r = a.r + v; else if (which == 1) a.dx0 = a.dx0 + v; else if (which == 2) a.dy0 = a.dy0 + v;
        else if (which == 3) a.dx1 = a.dx1 + v; else a.dy1 = a.dy1 + v;
    }
    GD Accum result() { return a; }
};
struct AccLds {
    double *slot;   // &s_acc[0][tid]; component c lives at slot[c * stride]
    int stride;
    GD void init() { for (int c = 0; c < 15; c++) slot[c * stride] = 0.0; }
    GD void add(int which, D3 v) {
        double *p = slot + which * 3 * stride;
        __hip_atomic_fetch_add(p, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(p + stride, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(p + 2 * stride, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    GD Accum result() {
        Accum a;
        a.r = mk(slot[0], slot[stride], slot[2 * stride]);
        a.dx0 = mk(slot[3 * stride], slot[4 * stride], slot[5 * stride]);
        a.dy0 = mk(slot[6 * stride], slot[7 * stride], slot[8 * stride]);
        a.dx1 = mk(slot[9 * stride], slot[10 * stride], slot[11 * stride]);
        a.dy1 = mk(slot[12 * stride], slot[13 * stride], slot[14 * stride]);
        return a;
    }
};

template <class ACC>
GD void acc_base(ACC &a, D3 radiance, double prob, double spp, LaneCounters &lc) {
    if (!(isfinite(prob) && isfinite(radiance.x + radiance.y + radiance.z))) lc.nonfinite++;
    if (prob > 0.0) a.add(0, radiance / spp);
}
// k: 0 = x0, 1 = x1, 2 = y0, 3 = y1
template <class ACC>
GD void acc_offset(ACC &a, int k, D3 contrib, D3 cX, double w, double prob, double spp, LaneCounters &lc, bool &flagged) {
    if (!(isfinite(cX.x + cX.y + cX.z) && isfinite(w) && isfinite(contrib.x + contrib.y + contrib.z))) { if (!flagged) lc.nonfinite++; flagged = true; }
    if (!(prob > 0.0)) return;
    double f = w / (prob * spp);
    if (k == 0) a.add(1, (contrib - cX) * f);
    else if (k == 1) a.add(3, (cX - contrib) * f);
    else if (k == 2) a.add(2, (contrib - cX) * f);
    else a.add(4, (cX - contrib) * f);
}
// a finished sample whose four offsets are all dead: contribX = 0, wX = 1 (struct defaults, src/intersection.h:65-77)
template <class ACC>
GD void acc_no_offsets(ACC &a, D3 radiance, D3 contrib, double prob, double spp, LaneCounters &lc) {
    bool finite = isfinite(prob) && isfinite(radiance.x + radiance.y + radiance.z) && isfinite(contrib.x + contrib.y + contrib.z);
    if (!finite) lc.nonfinite++;
    if (prob > 0.0) {
        a.add(0, radiance / spp);
        D3 t = contrib * (1.0 / (prob * spp));
        a.add(1, t); a.add(2, t);
        a.add(3, -t); a.add(4, -t);
    }
}

// ------------------------------------------------------------------------------------------------
// scene memory view: BVH nodes / primitive records from HBM or from the block's LDS copy
// ------------------------------------------------------------------------------------------------
struct TraceCtx {      // (every field is set by setup_trace or, for the wavefront kernels, by hand: keep the two in step)
    const DevBvhNode *nodes;
    const DevBvh4Node *nodes4;
    const DevBvh8Node *nodes8;
    const DevBvh4QNode *nodes4q;
    const DevPrim *prims;
    const DevTriShade *tris;
    const GdptMaterial *materials;
    const double *lights;    // area-light intensities
    int *stack;
    int stride;
    bool count;       // wave-uniform: count BVH nodes / primitives (bench roofline accounting)
    bool need_uv;     // wave-uniform: some texture is not constant (uv / footprint are observable)
};

// One BVH4 node: tests the four child boxes and returns the hit children sorted near to far. key = entry distance
// (>= 0, so its bit pattern orders like the float) with the child slot in the low two bits; 0xFFFFFFFF = missed.
struct WideVisit { unsigned key[4]; int ch[4]; };    // ch[i]: child behind key[i]
constexpr unsigned kMissKey = 0xFFFFFFFFu;
GD int pick4(unsigned k, int a0, int a1, int a2, int a3) {     // three v_cndmask, no branches
    const int lo = (k & 1u) ? a1 : a0, hi = (k & 1u) ? a3 : a2;
    return (k & 2u) ? hi : lo;
}
// SIGNED (scenes resident in LDS): the plane a ray enters a slab through is known from the sign of d, so the four near
// bounds and the four far bounds of an axis are read from the row that holds them (lo or hi) instead of ordering the two
// distances afterwards — same distances, same keys (rounding is monotonic), a third fewer instructions per child: cbox
// +3.3 % (same-box A/B). On scenes walked from HBM the six address selects cost more than they save (16 instead of 4
// spilled VGPRs in the general Lambertian kernel, sponza -10 %, the Disney scenes -1..-3 %), so those keep the ordered form.
template <bool SIGNED>
GD void visit_wide(const DevBvh4Node &n, const float oi[3], const float inv[3], float tnear, float tb, WideVisit &w) {
    const int c0 = n.child[0], c1 = n.child[1], c2 = n.child[2], c3 = n.child[3];
    const char *base = (const char *)&n;
    float nr[3][4], fr[3][4];
    if (SIGNED) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const bool neg = inv[k] < 0.0f;                      // NaN (d = 0): either row, the distances are NaN and drop out
            const float4 a = *(const float4 *)(base + (neg ? 48 + 16 * k : 16 * k)), b = *(const float4 *)(base + (neg ? 16 * k : 48 + 16 * k));
            nr[k][0] = a.x; nr[k][1] = a.y; nr[k][2] = a.z; nr[k][3] = a.w;
            fr[k][0] = b.x; fr[k][1] = b.y; fr[k][2] = b.z; fr[k][3] = b.w;
        }
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        float t0 = tnear, t1 = tb;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (SIGNED) { t0 = fmaxf(t0, fmaf(nr[k][c], inv[k], -oi[k])); t1 = fminf(t1, fmaf(fr[k][c], inv[k], -oi[k])); }
            else {
                float a = fmaf(n.lo[k][c], inv[k], -oi[k]), b = fmaf(n.hi[k][c], inv[k], -oi[k]);
                t0 = fmaxf(t0, fminf(a, b)); t1 = fminf(t1, fmaxf(a, b));       // NaN (inf - inf, 0*inf) is dropped by fmin/fmax
            }
        }
        const int ch = c == 0 ? c0 : (c == 1 ? c1 : (c == 2 ? c2 : c3));
        const bool h = (ch != GDPT_CHILD_EMPTY) && (t0 <= t1);   // see box_hit
        w.key[c] = h ? ((__float_as_uint(t0) & ~3u) | (unsigned)c) : kMissKey;
    }
#define GDPT_CSWAP(i, j) { unsigned lo_ = min(w.key[i], w.key[j]), hi_ = max(w.key[i], w.key[j]); w.key[i] = lo_; w.key[j] = hi_; }
    GDPT_CSWAP(0, 1) GDPT_CSWAP(2, 3) GDPT_CSWAP(0, 2) GDPT_CSWAP(1, 3) GDPT_CSWAP(1, 2)
#undef GDPT_CSWAP
#pragma unroll
    for (int i = 0; i < 4; i++) w.ch[i] = pick4(w.key[i], c0, c1, c2, c3);
}

// One BVH4 node with quantised boxes (DevBvh4QNode: scenes walked from HBM, GDPT_HBM_Q4 builds): four 16-byte loads instead
// of seven. Distances are formed as fma(q, A, B) with A = scale / d, B = (org - o) / d per axis — conservative by the
// argument given at visit_wide8 (the grid boxes enclose the boxes the host padded). The plane a ray enters a slab through
// is picked by the sign of d on the whole dword of four bounds (one v_cndmask per axis and side), so a child costs six
// conversions, six fmas and a max3 / min3 pair: the instruction count of the fp32 node's ordered test.
GD void visit_wide_q4(const DevBvh4QNode &n, const float oi[3], const float inv[3], float tnear, float tb, WideVisit &w) {
    const uint4 *qp = (const uint4 *)&n;
    const uint4 q0 = qp[0], q1 = qp[1], q2 = qp[2], q3 = qp[3];
    const float org[3] = {__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z)};
    const float scale[3] = {__uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y)};
    const unsigned lo[3] = {q1.z, q1.w, q2.x}, hi[3] = {q2.y, q2.z, q2.w};
    const int c0 = (int)q3.x, c1 = (int)q3.y, c2 = (int)q3.z, c3 = (int)q3.w;
    float A[3], B[3];
    unsigned nq[3], fq[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        A[k] = scale[k] * inv[k];
        B[k] = fmaf(org[k], inv[k], -oi[k]);
        const bool neg = inv[k] < 0.0f;
        nq[k] = neg ? hi[k] : lo[k]; fq[k] = neg ? lo[k] : hi[k];
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        float t0 = tnear, t1 = tb;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float qa = (float)((nq[k] >> (8 * c)) & 0xffu), qb = (float)((fq[k] >> (8 * c)) & 0xffu);
            t0 = fmaxf(t0, fmaf(qa, A[k], B[k])); t1 = fminf(t1, fmaf(qb, A[k], B[k]));     // NaN (d = 0) is dropped by fmin/fmax
        }
        const int ch = c == 0 ? c0 : (c == 1 ? c1 : (c == 2 ? c2 : c3));
        const bool h = (ch != GDPT_CHILD_EMPTY) && (t0 <= t1);
        w.key[c] = h ? ((__float_as_uint(t0) & ~3u) | (unsigned)c) : kMissKey;
    }
#define GDPT_CSWAP(i, j) { unsigned lo_ = min(w.key[i], w.key[j]), hi_ = max(w.key[i], w.key[j]); w.key[i] = lo_; w.key[j] = hi_; }
    GDPT_CSWAP(0, 1) GDPT_CSWAP(2, 3) GDPT_CSWAP(0, 2) GDPT_CSWAP(1, 3) GDPT_CSWAP(1, 2)
#undef GDPT_CSWAP
#pragma unroll
    for (int i = 0; i < 4; i++) w.ch[i] = pick4(w.key[i], c0, c1, c2, c3);
}

// One BVH8 node (scenes walked from HBM): eight child boxes, one byte per bound on the node's grid org + q * scale.
// The slab distances are formed as fma(q, A, B) with A = scale / d and B = (org - o) / d per axis. A = fl(scale * fl(1/d))
// and B = fma(org, fl(1/d), -fl(o/d)) round once each on top of 1/d and o/d, the final fma once: with e = 2^-24 a
// computed distance differs from ((org + q scale) - o) / d by at most e (|q A| + |B| + |t|) beyond what 1/d and o/d
// contribute (box_hit: e (|t| + |o/d|)), in total <= e (2E + 2E + 2E + 2E + E) / |d| = 5.4e-7 E / |d| — inside the
// P / |d| = 1e-6 E / |d| by which the host's padding moved every plane before the boxes were put on the grid (rounded
// outward there, verified in double).
// The near plane of an axis is picked by the sign of d on whole dwords of four bounds (two v_cndmask per axis and
// half), so a child costs six conversions, six fmas and two max3 / min3 pairs. Returns the hit children as keys sorted
// near to far (entry distance with the slot in the low three bits).
struct Wide8Visit { unsigned key[8]; };
GD int pick8(unsigned k, const int c[8]) {     // seven v_cndmask
    const int a0 = (k & 1u) ? c[1] : c[0], a1 = (k & 1u) ? c[3] : c[2], a2 = (k & 1u) ? c[5] : c[4], a3 = (k & 1u) ? c[7] : c[6];
    const int b0 = (k & 2u) ? a1 : a0, b1 = (k & 2u) ? a3 : a2;
    return (k & 4u) ? b1 : b0;
}
GD void visit_wide8(const DevBvh8Node &n, const float oi[3], const float inv[3], float tnear, float tb, int ch[8], Wide8Visit &w) {
    const uint4 *qp = (const uint4 *)&n.qlo[0][0];
    const uint4 q0 = qp[0], q1 = qp[1], q2 = qp[2];     // qlo x, y | qlo z, qhi x | qhi y, z
    const unsigned lo[3][2] = {{q0.x, q0.y}, {q0.z, q0.w}, {q1.x, q1.y}};
    const unsigned hi[3][2] = {{q1.z, q1.w}, {q2.x, q2.y}, {q2.z, q2.w}};
    float A[3], B[3];
    unsigned nq[3][2], fq[3][2];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        A[k] = n.scale[k] * inv[k];
        B[k] = fmaf(n.org[k], inv[k], -oi[k]);
        const bool neg = inv[k] < 0.0f;
#pragma unroll
        for (int h = 0; h < 2; h++) { nq[k][h] = neg ? hi[k][h] : lo[k][h]; fq[k][h] = neg ? lo[k][h] : hi[k][h]; }
    }
#pragma unroll
    for (int c = 0; c < 8; c++) ch[c] = n.child[c];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        float t0 = tnear, t1 = tb;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float qa = (float)((nq[k][c >> 2] >> (8 * (c & 3))) & 0xffu), qb = (float)((fq[k][c >> 2] >> (8 * (c & 3))) & 0xffu);
            t0 = fmaxf(t0, fmaf(qa, A[k], B[k])); t1 = fminf(t1, fmaf(qb, A[k], B[k]));     // NaN (d = 0) is dropped by fmin/fmax
        }
        const bool h = (ch[c] != GDPT_CHILD_EMPTY) && (t0 <= t1);
        w.key[c] = h ? ((__float_as_uint(t0) & ~7u) | (unsigned)c) : kMissKey;
    }
    if (!GDPT_BVH8_SORT) return;
#define GDPT_CSWAP(i, j) { unsigned lo_ = min(w.key[i], w.key[j]), hi_ = max(w.key[i], w.key[j]); w.key[i] = lo_; w.key[j] = hi_; }
    // 19-comparator network for eight keys
    GDPT_CSWAP(0, 1) GDPT_CSWAP(2, 3) GDPT_CSWAP(4, 5) GDPT_CSWAP(6, 7)
    GDPT_CSWAP(0, 2) GDPT_CSWAP(1, 3) GDPT_CSWAP(4, 6) GDPT_CSWAP(5, 7)
    GDPT_CSWAP(1, 2) GDPT_CSWAP(5, 6) GDPT_CSWAP(0, 4) GDPT_CSWAP(3, 7)
    GDPT_CSWAP(1, 5) GDPT_CSWAP(2, 6)
    GDPT_CSWAP(1, 4) GDPT_CSWAP(3, 6)
    GDPT_CSWAP(2, 4) GDPT_CSWAP(3, 5)
    GDPT_CSWAP(3, 4)
#undef GDPT_CSWAP
}

// A leaf holds 1..4 primitive records. All of them are fetched before the first test (indices clamped to the leaf, so
// short leaves re-read their last record): the leaf then costs one memory latency instead of one per primitive.
// (Tried: not fetching the slots a leaf does not fill — exec-masked loads instead of clamped indices, a third fewer requests per
// leaf: sponza -2.7 %, the Disney scenes -6..-10 %, profiles/r03_ab_leaf_loads.txt; the mask bookkeeping costs more than the
// re-reads of a line the lane has just fetched.)
template <bool FLAT, bool SPHERES = true>
GD void test_leaf(const DevSceneView &sv, const TraceCtx &tx, int cur, const float o[3], const float d[3], float tnear, float tfar,
                  Hit &best, TraceCounters &tc) {
    const unsigned packed = ~(unsigned)cur;
    const unsigned first = packed >> 2, last = packed & 3u;      // last = count - 1
    const DevPrim p0 = tx.prims[first];
    const DevPrim p1 = tx.prims[first + min(1u, last)];
    const DevPrim p2 = tx.prims[first + min(2u, last)];
    const DevPrim p3 = tx.prims[first + last];
    if (tx.count) { tc.prims += last + 1u; if (wave_leader()) tc.leaf_trips++; }
    // FLAT (scenes walked from HBM, incoherent rays): four branch-free triangle tests. Otherwise (small LDS-resident
    // scenes, where whole waves miss a triangle together) the early-outs of tri_hit pay.
    if (!FLAT || (SPHERES && ((p0.gid | p1.gid | p2.gid | p3.gid) & GDPT_SPHERE_FLAG))) {     // (a sphere in the leaf: general test)
        test_prim<SPHERES>(sv, p0, o, d, tnear, tfar, best);
        if (last >= 1u) test_prim<SPHERES>(sv, p1, o, d, tnear, tfar, best);
        if (last >= 2u) test_prim<SPHERES>(sv, p2, o, d, tnear, tfar, best);
        if (last >= 3u) test_prim<SPHERES>(sv, p3, o, d, tnear, tfar, best);
    } else {
        test_tri_flat(p0, o, d, tnear, tfar, true, best);
        test_tri_flat(p1, o, d, tnear, tfar, last >= 1u, best);
        test_tri_flat(p2, o, d, tnear, tfar, last >= 2u, best);
        test_tri_flat(p3, o, d, tnear, tfar, last >= 3u, best);
    }
}

// Resumable closest-hit traversal (definition of "closest": device_trace.h). The walk's whole state is (cur, sp, best)
// plus the lane's stack column in LDS, so a wave can leave the loop while some rays are unfinished, shade and re-arm the
// lanes that are done, and come back: `stop_below` = number of unfinished rays at or below which the loop is left
// (0 = run every ray to completion).
// WIDE: walk the BVH4 form (half the dependent node fetches of the BVH2). WW: "while-while" loop order — all lanes
// first walk inner nodes until each holds a leaf (or has finished), then the leaves are intersected together (scenes
// walked from HBM) — vs. one node or leaf per trip (scenes resident in LDS).
constexpr int kTravDone = INT32_MIN;      // cur: >= 0 inner node, < 0 leaf (~cur = first << 2 | count-1), kTravDone finished
struct Trav { Hit best; int cur, sp; int ovf[GDPT_STACK_OVERFLOW]; };     // ovf: stack slots past the LDS column (BVH8 only; never touched otherwise)
GD void trav_init(const DevSceneView &sv, Trav &tv, double tfar) {
    tv.best.gid = -1; tv.best.t = (float)tfar; tv.best.u = tv.best.v = 0; tv.best.ngx = tv.best.ngy = tv.best.ngz = 0;
    tv.sp = 0; tv.cur = (sv.num_nodes == 0) ? kTravDone : 0;
}
// (What was tried on this walk and dropped — top of the tree in LDS, distance-tagged stack entries, the top of the stack in a
// register — is recorded with its measurements in DESIGN.md 7, "Tried and dropped".)
// OVF: the tree's stack bound may exceed the LDS column (BVH8): slots from GDPT_BVH_MAX_DEPTH on live in the lane's private array.
template <bool OVF = false>
GD void trav_pop(const TraceCtx &tx, int &cur, int &sp, int *ovf = nullptr) {
    if (sp > 0) {
        sp--;
        if (OVF && sp >= GDPT_BVH_MAX_DEPTH) cur = ovf[sp - GDPT_BVH_MAX_DEPTH];
        else cur = tx.stack[sp * tx.stride];
    } else cur = kTravDone;
}
template <bool OVF = false>
GD void trav_push(const TraceCtx &tx, int &sp, int x, int *ovf = nullptr) {
    if (OVF && sp >= GDPT_BVH_MAX_DEPTH) ovf[sp - GDPT_BVH_MAX_DEPTH] = x;
    else tx.stack[sp * tx.stride] = x;
    sp++;
}
template <bool WIDE, bool HBM = false>
GD void trav_node(const TraceCtx &tx, const float oi[3], const float inv[3], float tnear, float tb, int &cur, int &sp, int *ovf) {
    if (WIDE && HBM && GDPT_HBM_BVH8) {
        Wide8Visit w;
        int ch[8];
        visit_wide8(tx.nodes8[cur], oi, inv, tnear, tb, ch, w);
        if (GDPT_BVH8_SORT) {
            if (w.key[0] != kMissKey) {
#pragma unroll
                for (int i = 7; i >= 1; i--) if (w.key[i] != kMissKey) trav_push<true>(tx, sp, pick8(w.key[i], ch), ovf);
                cur = pick8(w.key[0], ch);
            } else trav_pop<true>(tx, cur, sp, ovf);
        } else {
            unsigned near = w.key[0];
#pragma unroll
            for (int c = 1; c < 8; c++) near = min(near, w.key[c]);
            if (near != kMissKey) {
#pragma unroll
                for (int c = 0; c < 8; c++) if (w.key[c] != kMissKey && w.key[c] != near) trav_push<true>(tx, sp, ch[c], ovf);
                cur = pick8(near, ch);
            } else trav_pop<true>(tx, cur, sp, ovf);
        }
    } else if (WIDE) {
        WideVisit w;
#ifdef GDPT_EXTRA_LOOKUPS     // experiment: seven more L1 lookups per node visit (hits), results unchanged
        if (HBM) {
            const char *np = (const char *)&tx.nodes4[cur];
            unsigned d0, d1, d2, d3, d4, d5, d6;
            asm volatile("global_load_dword %0, %7, off\n global_load_dword %1, %7, off offset:16\n global_load_dword %2, %7, off offset:32\n"
                         "global_load_dword %3, %7, off offset:48\n global_load_dword %4, %7, off offset:64\n global_load_dword %5, %7, off offset:80\n"
                         "global_load_dword %6, %7, off offset:96\n s_waitcnt vmcnt(0)"
                         : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6) : "v"(np) : "memory");
        }
#endif
        if (HBM && GDPT_HBM_Q4) visit_wide_q4(tx.nodes4q[cur], oi, inv, tnear, tb, w);
        else visit_wide<!HBM>(tx.nodes4[cur], oi, inv, tnear, tb, w);
        if (w.key[0] != kMissKey) {
            if (w.key[3] != kMissKey) trav_push(tx, sp, w.ch[3]);
            if (w.key[2] != kMissKey) trav_push(tx, sp, w.ch[2]);
            if (w.key[1] != kMissKey) trav_push(tx, sp, w.ch[1]);
            cur = w.ch[0];
        } else trav_pop(tx, cur, sp);
    } else {
        const DevBvhNode &n = tx.nodes[cur];
        float tl, tr;
        bool hl = (n.left != GDPT_CHILD_EMPTY) && box_hit(n.lmin, n.lmax, oi, inv, tnear, tb, tl);
        bool hr = (n.right != GDPT_CHILD_EMPTY) && box_hit(n.rmin, n.rmax, oi, inv, tnear, tb, tr);
        if (hl && hr) {
            int nearc = n.left, farc = n.right;
            if (tr < tl) { nearc = n.right; farc = n.left; }
            trav_push(tx, sp, farc);
            cur = nearc;
        } else if (hl) cur = n.left;
        else if (hr) cur = n.right;
        else trav_pop(tx, cur, sp);
    }
}
// Compile-time traversal configuration of a kernel: loop order, tree form, leaf test style.
// SPEC (while-while order only): a lane that reaches a leaf sets it aside and goes on looking for its next one.
template <bool WW_, bool WIDE_, bool FLAT_, bool SPHERES_ = true, bool SPEC_ = false> struct TraceCfg { static constexpr bool WW = WW_, WIDE = WIDE_, FLAT = FLAT_, SPHERES = SPHERES_, SPEC = SPEC_ || GDPT_SPEC_LEAF; };
using TraceHbm = TraceCfg<true, true, true>;       // scenes walked from HBM

// Called by the lanes whose ray is unfinished (tv.cur != kTravDone); the others of the wave sit it out.
template <class TC>
GD void trav_run(const DevSceneView &sv, const TraceCtx &tx, D3 org, D3 dir, float tnear, float tfar, Trav &tv, int stop_below, int search_frac, TraceCounters &tc,
                 bool any_hit = false) {     // any_hit: occlusion query, the walk ends at the first accepted primitive
    const float o[3] = {(float)org.x, (float)org.y, (float)org.z};
    const float d[3] = {(float)dir.x, (float)dir.y, (float)dir.z};
    float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    float oi[3] = {o[0] * inv[0], o[1] * inv[1], o[2] * inv[2]};
    // a zero direction component gives inv = inf and o*inv = inf or NaN: the slab then drops out of fmin/fmax only if
    // the product is NaN, so force it (the ray is parallel to that slab; the padded box contains the origin's
    // coordinate or the other axes reject)
#pragma unroll
    for (int k = 0; k < 3; k++) if (d[k] == 0.0f) { inv[k] = __builtin_nanf(""); oi[k] = __builtin_nanf(""); }
    constexpr bool kOvf = TC::WIDE && TC::FLAT && GDPT_HBM_BVH8;
    int cur = tv.cur, sp = tv.sp;
    Hit best = tv.best;
    for (;;) {
        const int live = __popcll(__ballot(cur != kTravDone));
        if (live <= stop_below) break;
        if (TC::WW) {
            // inner nodes until at most search_frac/256 of the live lanes are still looking for their next leaf
            // (waiting for the last lane costs ~ln(64) mean search lengths); those lanes sit out the leaf tests.
            const int few = (live * search_frac) >> 8;
            if (TC::SPEC) {
                // A lane that reaches a leaf sets it aside and goes on looking for its next one while the others are still
                // searching (the nodes it visits meanwhile are tested against the hit distance it had before the leaf: a
                // few more visits, never a different hit); the leaf set aside is intersected when the node loop is left, so
                // no leaf is held across two rounds and the resumable state stays (cur, sp, best). One-sided lane machine,
                // same-box A/B: sponza +3.3 % (20.2 instead of 19.4 nodes and 7.9 instead of 6.9 triangles per ray, 33.4
                // instead of 35.0 trips per wave step), Disney metal / diffuse +2 %; the two-sided machine does not gain.
                int held = kTravDone;
                for (;;) {
                    if (cur < 0 && cur != kTravDone && held == kTravDone) { held = cur; trav_pop<kOvf>(tx, cur, sp, tv.ovf); }
                    const bool searching = cur >= 0;
                    if (__popcll(__ballot(searching)) <= few) break;
                    if (searching) {
                        if (tx.count) { tc.nodes++; if (wave_leader()) tc.node_trips++; }
                        trav_node<TC::WIDE, TC::FLAT>(tx, oi, inv, tnear, best.t, cur, sp, tv.ovf);
                    }
                }
                if (held != kTravDone) {
                    test_leaf<TC::FLAT, TC::SPHERES>(sv, tx, held, o, d, tnear, tfar, best, tc);
                    if (any_hit && best.gid >= 0) cur = kTravDone;
                }
            } else {
                for (;;) {
                    const bool searching = cur >= 0;
                    if (__popcll(__ballot(searching)) <= few) break;
                    if (searching) {
                        if (tx.count) { tc.nodes++; if (wave_leader()) tc.node_trips++; }
                        trav_node<TC::WIDE, TC::FLAT>(tx, oi, inv, tnear, best.t, cur, sp, tv.ovf);
                    }
                }
                if (cur < 0 && cur != kTravDone) {
                    test_leaf<TC::FLAT, TC::SPHERES>(sv, tx, cur, o, d, tnear, tfar, best, tc);
                    trav_pop<kOvf>(tx, cur, sp, tv.ovf);
                    if (any_hit && best.gid >= 0) cur = kTravDone;
                }
            }
        } else if (cur >= 0) {
            if (tx.count) { tc.nodes++; if (wave_leader()) tc.node_trips++; }
            trav_node<TC::WIDE, TC::FLAT>(tx, oi, inv, tnear, best.t, cur, sp, tv.ovf);
        } else if (cur != kTravDone) {
            test_leaf<TC::FLAT, TC::SPHERES>(sv, tx, cur, o, d, tnear, tfar, best, tc);
            trav_pop<kOvf>(tx, cur, sp, tv.ovf);
            if (any_hit && best.gid >= 0) cur = kTravDone;
        }
    }
    tv.cur = cur; tv.sp = sp; tv.best = best;
}

constexpr int kEagerSearchFrac = 112;     // /256 of the live lanes still searching a leaf when the node loop is left

// One ray, start to finish (eager evaluator).
template <class TC>
GD bool intersect_ctx(const DevSceneView &sv, const TraceCtx &tx, const Ray &ray, double rd_spread, Vertex &v, LaneCounters &lc, TraceCounters &tc) {
    lc.rays++;
    Trav tv;
    trav_init(sv, tv, ray.tfar);
    // run to completion, but leave the inner-node loop early like the lane machine does (kEagerSearchFrac)
    if (tv.cur != kTravDone) trav_run<TC>(sv, tx, ray.org, ray.dir, (float)ray.tnear, (float)ray.tfar, tv, 0, kEagerSearchFrac, tc);
    if (tv.best.gid < 0) return false;
    make_vertex(sv, tx.tris, tx.need_uv, ray, tv.best, 0.0, rd_spread, v);
    return true;
}

// occluded(), src/intersection.cpp:67-85: true if any primitive accepts the fp32 ray in [tnear, tfar).
template <class TC>
GD bool occluded_ctx(const DevSceneView &sv, const TraceCtx &tx, const Ray &ray, LaneCounters &lc, TraceCounters &tc) {
    lc.rays++;
    Trav tv;
    trav_init(sv, tv, ray.tfar);
    if (tv.cur != kTravDone) trav_run<TC>(sv, tx, ray.org, ray.dir, (float)ray.tnear, (float)ray.tfar, tv, 0, kEagerSearchFrac, tc, true);
    return tv.best.gid >= 0;
}

// ------------------------------------------------------------------------------------------------
// material dispatch: Lambertian-only scenes get the three-line lobe inline, others the full switch
// ------------------------------------------------------------------------------------------------
template <bool LAMBERT, bool ROUGH = false, bool TWOSIDED = false, unsigned MASK = kAllMaterials>
GD bool mat_sample(const DevSceneView &sv, const TraceCtx &tx, const Vertex &v, D3 in, D2 ruv, double rw, BsdfSample &s) {
    if (LAMBERT) { Ctx c{sv, v}; return cos_sample(c, in, ruv, 1.0, s); }
    return bsdf_sample<ROUGH, TWOSIDED, MASK>(sv, tx.materials[v.material_id], in, v, ruv, rw, s);
}
// eval (f*|cos|) and solid-angle pdf together
template <bool LAMBERT, bool ROUGH = false, bool TWOSIDED = false, unsigned MASK = kAllMaterials, int PLAIN = 0>
GD void mat_eval_pdf(const DevSceneView &sv, const TraceCtx &tx, const Vertex &v, D3 in, D3 out, D3 &f, double &pdf) {
    if (LAMBERT) {
        // src/materials/lambertian.inl:1-33 — eval and pdf share the clamped cosine
        if (below(v, in) || below(v, out)) { f = splat(0); pdf = 0; return; }
        Frame fr = oriented_frame(v, in);
        double c = fmax(dot(fr.n, out), 0.0);
        const GdptTexture &refl = tx.materials[v.material_id].tex[0];
        f = c * ((PLAIN & kPlainConstTex) ? mk(refl.v0[0], refl.v0[1], refl.v0[2]) : tex3(sv, refl, v)) / kPi;
        pdf = c / kPi;
        return;
    }
    const GdptMaterial &m = tx.materials[v.material_id];
    bsdf_eval_pdf<ROUGH, TWOSIDED, MASK>(sv, m, in, out, v, f, pdf);
}
template <bool LAMBERT>
GD double mat_pdf(const DevSceneView &sv, const TraceCtx &tx, const Vertex &v, D3 in, D3 out) {
    if (LAMBERT) { Ctx c{sv, v}; return cos_pdf(c, in, out); }
    return bsdf_pdf(sv, tx.materials[v.material_id], in, out, v);
}

// ------------------------------------------------------------------------------------------------
// one-ray-per-iteration lane machine
// ------------------------------------------------------------------------------------------------
// A lane always holds ONE pending ray. Every loop iteration each live lane traces its pending ray and rebuilds the
// hit vertex — the bulk of the instructions, executed with all lanes active — then a short state-dependent part
// consumes the hit and prepares the lane's next pending ray:
//   S_PRIMARY  base camera ray of the lane's current sample
//   S_BOUNCE   bounce ray of the base path; f and the solid-angle pdf of its direction were evaluated at the
//              vertex it leaves, so no vertex has to stay in registers across the traversal
//   S_OFFSET   camera ray of offset k of a base path that stopped where offsets are observable (lazy, see header)
enum { S_START = 0, S_PRIMARY = 1, S_BOUNCE = 2, S_OFFSET = 3, S_DONE = 4 };
enum { ACT_NONE = 0, ACT_BOUNCE = 1, ACT_OFFSETS = 2, ACT_NEXT_SAMPLE = 3, ACT_PRIMARY_RAY = 4, ACT_OFFSET_RAY = 5 };
// why the base path stopped, which decides what the offsets still do
enum { C_NO_LOOP = 0,        // the bounce loop never ran (max_depth < 2): jacobian 1
       C_AFTER_BOUNCE1 = 1,  // left the loop in bounce 1 after the updates: offsets were re-sampled in bounce 1
       C_BROKE_BOUNCE1 = 2,  // p2 <= 0 in bounce 1 (break before the offsets' re-sampling): jacobian 1
       C_BROKE_BOUNCE2 = 3 };// p2 <= 0 in bounce 2: bounce-1 jacobian (bounce 2's material test already passed)

// Cold per-lane fields live in a private LDS slot (a "manual spill" that costs an LDS access instead of a scratch
// round trip): radiance (touched on emitter hits and when a sample is finished), p2_1 and the filter cache
// (offset phase only).  Layout: slot[c * stride], c = 0..2 radiance, 3 p2_1, 4..7 filter cache (dx, dy, ox, oy).
struct LanePriv {
    double *slot; int stride;
    GD D3 radiance() const { return mk(slot[0], slot[stride], slot[2 * stride]); }
    GD void set_radiance(D3 v) { slot[0] = v.x; slot[stride] = v.y; slot[2 * stride] = v.z; }
    GD double p2_1() const { return slot[3 * stride]; }
    GD void set_p2_1(double v) { slot[3 * stride] = v; }
    GD FilterCache fc() const { FilterCache f; f.dx = slot[4 * stride]; f.dy = slot[5 * stride]; f.ox = slot[6 * stride]; f.oy = slot[7 * stride]; return f; }
    GD void set_fc(const FilterCache &f) { slot[4 * stride] = f.dx; slot[5 * stride] = f.dy; slot[6 * stride] = f.ox; slot[7 * stride] = f.oy; }
};
constexpr int kPrivDoubles = 8;

struct Lane {
    int st, s, s_end, num_vertices;
    int mats;                 // mat0 | mat1 << 12 (material ids of the base primary hit / the bounce-1 hit), 0xFFF = none
    int kc;                   // offset index k | cmode << 2
    unsigned long long rng_state, rng_inc;
    D3 org, dir;              // the pending ray
    D3 f; double pdf;         // S_BOUNCE: f*|cos| and solid-angle pdf of `dir` at the vertex the ray leaves
    D3 contrib, throughput;
    double prob;
    double rng_x, rng_y;      // serial-RNG (TILE) mode only: the sample's sub-pixel and bounce-1 numbers
    D2 ruv1; double rw1;      // (SAMPLE mode re-derives them from the sample's own PCG stream)
    GD int mat0() const { return mats & 0xFFF; }
    GD int mat1() const { return (mats >> 12) & 0xFFF; }
    GD int k() const { return kc & 3; }
    GD int cmode() const { return kc >> 2; }
};

// SERIAL_RNG: one PCG stream runs through consecutive samples (TILE scheme); otherwise each sample owns stream
// `base + s` (SAMPLE scheme) and its sub-pixel / bounce-1 numbers are re-derived from it when an offset needs them.
// PLAIN: device_trace.h (kPlainNoSpheres | kPlainConstTex): sphere / texture code is not compiled in.
template <bool LAMBERT, bool SERIAL_RNG, class ACC, class STAMPS = Stamps<false>, int PLAIN = 0>
GD int lane_consume(const DevSceneView &sv, const TraceCtx &tx, int max_depth, double spp, unsigned long long base,
                    Lane &L, Trav &tv, LanePriv &lp, ACC &acc, LaneCounters &lc, TraceCounters &tc, STAMPS *stp = nullptr) {
    STAMPS none{}; STAMPS &stamps = stp ? *stp : none;
    const DevCamera &cam = sv.cam;
    const int w = cam.width, h = cam.height;
    const int st0 = L.st;
    int act = ACT_NONE;
    Vertex nv;
    Ray ray;
    ray.org = L.org; ray.dir = L.dir; ray.tfar = __builtin_huge_val();
    ray.tnear = (st0 == S_BOUNCE) ? sv.isect_eps : 0.0;
    // ---------------- the pending ray's traversal has finished (tv): rebuild the hit vertex ----------------
    const bool tracing = (st0 == S_PRIMARY || st0 == S_BOUNCE || st0 == S_OFFSET);
    bool hit = false;
    if (tracing) {
        lc.rays++;
        hit = tv.best.gid >= 0;
        if (hit) make_vertex<PLAIN>(sv, tx.tris, tx.need_uv, ray, tv.best, 0.0, (st0 == S_BOUNCE) ? 0.0 : 0.25 / (double)max(w, h), nv);   // src/ray.h:33-35, :564
    }
    stamps.mark(SEG_VERTEX);
    // ---------------- consume the hit ----------------
    if (st0 == S_START) {
        act = ACT_PRIMARY_RAY;
    } else if (st0 == S_PRIMARY) {
        if (!hit) act = ACT_NEXT_SAMPLE;                                            // :375-379: zero record, adds nothing
        else {
            L.mats = (nv.material_id & 0xFFF) | (0xFFF << 12);
            L.contrib = splat(1.0); L.throughput = splat(1.0);
            L.prob = 1.0;
            D3 rad0 = splat(0);
            if (nv.light_id >= 0) { D3 Le = emission(tx.lights, nv, -ray.dir); rad0 = Le; L.contrib = Le; }   // :490-493
            lp.set_radiance(rad0); lp.set_p2_1(1.0);
            L.num_vertices = 3;
            if (!loop_allows(max_depth, 3)) { L.kc = C_NO_LOOP << 2; act = ACT_OFFSETS; } else act = ACT_BOUNCE;
        }
    } else if (st0 == S_BOUNCE) {
        const bool first = (L.num_vertices == 3);
        double G = 1.0;
        if (hit) { D3 dl = nv.position - ray.org; G = fabs(dot(ray.dir, nv.gn)) / dot(dl, dl); }   // :746-753
        const D3 f = L.f;
        double p2 = L.pdf * G;                                                      // :766
        L.contrib = L.contrib * f * G; L.prob *= p2;                                // :769-770
        if (first) lp.set_p2_1(p2);
        if (hit && nv.light_id >= 0) {                                              // :971-980
            D3 Le = emission(tx.lights, nv, -ray.dir);
            D3 C2 = (G * f) * Le;
            L.contrib = L.contrib * Le;
            lp.set_radiance(lp.radiance() + L.throughput * (C2 / p2));
        }
        bool stop = !hit;                                                           // :982-985
        if (!stop) {
            double rr_prob = 1;
            if (L.num_vertices - 1 >= sv.rr_depth) {                                // :992-999
                rr_prob = fmin(maxc(1.0 * L.throughput), 0.95);                    // eta_scale == 1: one-sided lobes never refract
                Pcg rr_rng; rr_rng.state = L.rng_state; rr_rng.inc = L.rng_inc;
                double u = pcg_real(rr_rng); L.rng_state = rr_rng.state;
                if (u > rr_prob) stop = true;
            }
            if (!stop) {
                L.throughput = L.throughput * (G * f) / (p2 * rr_prob);             // :1003
                L.num_vertices++;
                if (first) L.mats = (L.mats & 0xFFF) | ((nv.material_id & 0xFFF) << 12);
                if (!loop_allows(max_depth, L.num_vertices)) stop = true;
            }
        }
        if (!stop) act = ACT_BOUNCE;
        else if (first) { L.kc = C_AFTER_BOUNCE1 << 2; act = ACT_OFFSETS; }
        else { acc_no_offsets(acc, lp.radiance(), L.contrib, L.prob, spp, lc); act = ACT_NEXT_SAMPLE; }
        // (one-sided lobes: offsets alive after bounce 1 are retired by bounce 2's re-sampling, see file header)
    }
    stamps.mark(SEG_CONSUME);
    // ---------------- shared BSDF block: sample a direction at `nv` and evaluate f / pdf there ----------------
    // Needed by (a) base lanes that start a bounce iteration at nv and (b) offset lanes whose vertex nv is re-sampled
    // with the base path's bounce-1 numbers (:773-959). One copy of the code, run by both groups together.
    const bool off_valid = (st0 == S_OFFSET) && hit && nv.material_id == L.mat0();                   // :424-443
    const bool off_resample = off_valid && (L.cmode() == C_AFTER_BOUNCE1 || L.cmode() == C_BROKE_BOUNCE2);
    bool sampled = false;
    BsdfSample bs; bs.dir_out = splat(0); bs.eta = 0; bs.roughness = 0;
    D3 f = splat(0);
    double pdf = 0;
    if (act == ACT_BOUNCE || off_resample) {
        D2 ruv; double rw;
        if (act == ACT_BOUNCE) {
            lc.bounces++;
            Pcg r; r.state = L.rng_state; r.inc = L.rng_inc;
            ruv.x = pcg_real(r); ruv.y = pcg_real(r); rw = pcg_real(r);                              // :536-537
            L.rng_state = r.state;
            if (SERIAL_RNG && L.num_vertices == 3) { L.ruv1 = ruv; L.rw1 = rw; }
        } else if (SERIAL_RNG) { ruv = L.ruv1; rw = L.rw1; }
        else {
            Pcg r2 = pcg_init(base + (unsigned long long)L.s);
            (void)pcg_next(r2); (void)pcg_next(r2);
            ruv.x = pcg_real(r2); ruv.y = pcg_real(r2); rw = pcg_real(r2);
        }
        const D3 dir_view = -ray.dir;
        // (PLAIN >> kPlainSetShift: the scene's material set, if the kernel was built for a small one — render_phases_general_sets.hip)
        constexpr unsigned kSet = (PLAIN >> kPlainSetShift) ? (unsigned)(PLAIN >> kPlainSetShift) : kAllMaterials;
        sampled = mat_sample<LAMBERT, false, false, kSet>(sv, tx, nv, dir_view, ruv, rw, bs);
        if (sampled) mat_eval_pdf<LAMBERT, false, false, kSet, PLAIN>(sv, tx, nv, dir_view, bs.dir_out, f, pdf);
    }
    stamps.mark(SEG_BSDF);
    if (st0 == S_OFFSET) {
        const int k = L.k();
        D3 cX = splat(0);
        double wgt = 1.0;
        if (off_valid) {
            D3 c0 = (nv.light_id >= 0) ? emission(tx.lights, nv, -ray.dir) : splat(1.0);   // :496-508
            double jac = 1.0;
            bool alive = true;
            if (off_resample) {
                if (!sampled || pdf <= 0.0) alive = false; else jac = lp.p2_1() / pdf;  // :813
            }
            if (alive) { cX = c0 * jac; wgt = L.prob / (L.prob + 1.0 * jac); }      // :1019-1045
        }
        bool flagged = false;
        acc_offset(acc, k, L.contrib, cX, wgt, L.prob, spp, lc, flagged);
        if (k == 3) { acc_base(acc, lp.radiance(), L.prob, spp, lc); act = ACT_NEXT_SAMPLE; }
        else { L.kc = (L.kc & ~3) | (k + 1); act = ACT_OFFSET_RAY; }
    } else if (act == ACT_BOUNCE) {                                                  // bounce iteration L.num_vertices starts at `nv`
        if (!sampled) act = ACT_NEXT_SAMPLE;                                        // :545-548: GraidentPTRadiance{}
        else {
            // bs.eta == 0 for every one-sided lobe, so eta_scale (:553-558) stays 1 on this path
            if (pdf <= 0) {                                                         // :760-763: break before any update; the ray's
                if (L.num_vertices == 3) { L.kc = C_BROKE_BOUNCE1 << 2; act = ACT_OFFSETS; }           // hit is unobservable
                else if (L.num_vertices == 4 && L.mat0() == L.mat1()) { L.kc = C_BROKE_BOUNCE2 << 2; act = ACT_OFFSETS; }   // :607-612 passed
                else { acc_no_offsets(acc, lp.radiance(), L.contrib, L.prob, spp, lc); act = ACT_NEXT_SAMPLE; }
            } else {
                L.org = nv.position; L.dir = bs.dir_out; L.f = f; L.pdf = pdf; L.st = S_BOUNCE;
            }
        }
    }
    stamps.mark(SEG_FINISH);
    if (act == ACT_OFFSETS) { L.kc &= ~3; act = ACT_OFFSET_RAY; }
    if (act == ACT_NEXT_SAMPLE) {
        L.s++;
        if (L.s >= L.s_end) L.st = S_DONE; else act = ACT_PRIMARY_RAY;
    }
    if (L.st == S_BOUNCE && act == ACT_BOUNCE) trav_init(sv, tv, __builtin_huge_val());       // a bounce ray was armed
    return act;
}

// Second half of a lane's step: the camera ray of the next sample (ACT_PRIMARY_RAY) or of the next offset (ACT_OFFSET_RAY).
// Separate from lane_consume so that the persistent kernels can hand a NEW ITEM to a lane whose item has just ended in
// between: its first camera ray is then made in the same step instead of one trace phase later.
template <bool SERIAL_RNG, class STAMPS = Stamps<false>>
GD void lane_camera(const DevSceneView &sv, int act, int x, int y, unsigned long long base, Lane &L, Trav &tv, LanePriv &lp, STAMPS *stp = nullptr) {
    STAMPS none{}; STAMPS &stamps = stp ? *stp : none;
    const DevCamera &cam = sv.cam;
    if (act == ACT_PRIMARY_RAY || act == ACT_OFFSET_RAY) {
        double rx, ry;
        int ox = 0, oy = 0;
        if (act == ACT_PRIMARY_RAY) {
            Pcg r;
            if (!SERIAL_RNG) r = pcg_init(base + (unsigned long long)L.s);
            else { r.state = L.rng_state; r.inc = L.rng_inc; }
            rx = pcg_real(r); ry = pcg_real(r);                                     // :360-361
            L.rng_state = r.state; L.rng_inc = r.inc;
            if (SERIAL_RNG) { L.rng_x = rx; L.rng_y = ry; }
            L.st = S_PRIMARY;
        } else {
            if (SERIAL_RNG) { rx = L.rng_x; ry = L.rng_y; }
            else { Pcg r2 = pcg_init(base + (unsigned long long)L.s); rx = pcg_real(r2); ry = pcg_real(r2); }
            const int k = L.k();
            ox = (k == 0) ? -1 : (k == 1 ? 1 : 0); oy = (k == 2) ? 1 : (k == 3 ? -1 : 0);   // x0,x1,y0,y1 (:385-403)
            L.st = S_OFFSET;
        }
        FilterCache fc;
        if (act == ACT_OFFSET_RAY) fc = lp.fc();
        Ray r = sample_primary<true>(cam, (x + ox) + rx, (y + oy) + ry, &fc, act == ACT_PRIMARY_RAY);
        if (act == ACT_PRIMARY_RAY) lp.set_fc(fc);
        L.org = r.org; L.dir = r.dir;
        trav_init(sv, tv, __builtin_huge_val());                                    // a fresh pending ray
    }
    stamps.mark(SEG_CAMERA);
}

// Both halves back to back (serial kernels, wavefront step kernel).
template <bool LAMBERT, bool SERIAL_RNG, class ACC, class STAMPS = Stamps<false>, int PLAIN = 0>
GD void lane_step(const DevSceneView &sv, const TraceCtx &tx, int max_depth, double spp, int x, int y, unsigned long long base,
                  Lane &L, Trav &tv, LanePriv &lp, ACC &acc, LaneCounters &lc, TraceCounters &tc, STAMPS *stp = nullptr) {
    const int act = lane_consume<LAMBERT, SERIAL_RNG, ACC, STAMPS, PLAIN>(sv, tx, max_depth, spp, base, L, tv, lp, acc, lc, tc, stp);
    lane_camera<SERIAL_RNG, STAMPS>(sv, act, x, y, base, L, tv, lp, stp);
}

// The lane machine's two halves. trace_pending: traversal of the wave's unfinished pending rays, left when at most
// `keep_frac`/256 of them are still unfinished; step_ready: lane_step for the lanes whose ray is done (or that need a
// first ray). Lanes still in flight keep (tv, stack) and continue on the next call.
GD bool lane_tracing(int st) { return st == S_PRIMARY || st == S_BOUNCE || st == S_OFFSET; }
template <class TC>
GD void trace_pending(const DevSceneView &sv, const TraceCtx &tx, const Lane &L, Trav &tv, int keep_frac, int search_frac, TraceCounters &tc) {
    const bool pending = lane_tracing(L.st) && tv.cur != kTravDone;
    const unsigned long long m = __ballot(pending);
    if (m == 0ull) return;
    const int stop_below = (__popcll(m) * keep_frac) >> 8;
    if (pending) trav_run<TC>(sv, tx, L.org, L.dir, (L.st == S_BOUNCE) ? (float)sv.isect_eps : 0.0f, __builtin_huge_valf(), tv, stop_below, search_frac, tc);
}
GD bool lane_ready(const Lane &L, const Trav &tv) { return L.st == S_START || (lane_tracing(L.st) && tv.cur == kTravDone); }

GD unsigned wave_sum_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
GD unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
GD void flush_counters(const KernelArgs &a, const LaneCounters &lc, const TraceCounters &tc, bool count) {
    unsigned r = wave_sum_u32(lc.rays), b = wave_sum_u32(lc.bounces), nf = wave_sum_u32(lc.nonfinite);
    unsigned long long nn = 0, np = 0;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (count) {
        nn = wave_sum_u64((unsigned long long)tc.nodes); np = wave_sum_u64((unsigned long long)tc.prims);
        t0 = wave_sum_u64((unsigned long long)tc.node_trips); t1 = wave_sum_u64((unsigned long long)tc.leaf_trips);
        t2 = wave_sum_u64((unsigned long long)tc.wave_steps); t3 = wave_sum_u64((unsigned long long)tc.lane_steps);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&a.counters->rays, (unsigned long long)r);
        atomicAdd(&a.counters->bounces, (unsigned long long)b);
        if (nf) atomicAdd(&a.counters->nonfinite, (unsigned long long)nf);
        if (count) {
            atomicAdd(&a.counters->nodes, nn); atomicAdd(&a.counters->prims, np);
            atomicAdd(&a.counters->wave_node_trips, t0); atomicAdd(&a.counters->wave_leaf_trips, t1);
            atomicAdd(&a.counters->wave_steps, t2); atomicAdd(&a.counters->lane_steps, t3);
        }
    }
}

GD void reduce_and_store(const KernelArgs &a, Accum &acc, int K, bool writer, int x, int y, int W) {
    // fixed-order tree over the K lanes of a pixel (lanes of one pixel are contiguous and K-aligned)
    for (int o = K >> 1; o >= 1; o >>= 1) {
        acc.r.x += __shfl_xor(acc.r.x, o, 64); acc.r.y += __shfl_xor(acc.r.y, o, 64); acc.r.z += __shfl_xor(acc.r.z, o, 64);
        acc.dx0.x += __shfl_xor(acc.dx0.x, o, 64); acc.dx0.y += __shfl_xor(acc.dx0.y, o, 64); acc.dx0.z += __shfl_xor(acc.dx0.z, o, 64);
        acc.dy0.x += __shfl_xor(acc.dy0.x, o, 64); acc.dy0.y += __shfl_xor(acc.dy0.y, o, 64); acc.dy0.z += __shfl_xor(acc.dy0.z, o, 64);
        acc.dx1.x += __shfl_xor(acc.dx1.x, o, 64); acc.dx1.y += __shfl_xor(acc.dx1.y, o, 64); acc.dx1.z += __shfl_xor(acc.dx1.z, o, 64);
        acc.dy1.x += __shfl_xor(acc.dy1.x, o, 64); acc.dy1.y += __shfl_xor(acc.dy1.y, o, 64); acc.dy1.z += __shfl_xor(acc.dy1.z, o, 64);
    }
    if (writer) {
        size_t i = ((size_t)y * W + x) * 3;
        a.img[i] = acc.r.x; a.img[i + 1] = acc.r.y; a.img[i + 2] = acc.r.z;
        a.cx0[i] = acc.dx0.x; a.cx0[i + 1] = acc.dx0.y; a.cx0[i + 2] = acc.dx0.z;
        a.cy0[i] = acc.dy0.x; a.cy0[i + 1] = acc.dy0.y; a.cy0[i + 2] = acc.dy0.z;
        a.cx1[i] = acc.dx1.x; a.cx1[i + 1] = acc.dx1.y; a.cx1[i + 2] = acc.dx1.z;
        a.cy1[i] = acc.dy1.x; a.cy1[i + 1] = acc.dy1.y; a.cy1[i + 2] = acc.dy1.z;
    }
}

// dynamic LDS of the kernels that walk the scene from HBM: the traversal stack, sized from the tree's own bound
inline size_t hbm_dynamic_lds(const KernelArgs &a) { return (size_t)a.stack_levels * kBlock * sizeof(int); }

// Copies nodes (BVH2 or wide form), primitive records, the shading table and the materials into LDS (small scenes)
// and returns the trace context of this lane.
template <bool LDS_SCENE, bool WIDE>
GD TraceCtx setup_trace(const DevSceneView &sv, unsigned char *s_scene, int *s_stack, int tid, int nthreads, bool count) {
    TraceCtx tx;
    tx.count = count;
    tx.need_uv = !sv.all_textures_constant;
    tx.stack = s_stack + tid; tx.stride = nthreads;
    if (LDS_SCENE) {
        const int nw = WIDE ? sv.num_nodes4 * (int)(sizeof(DevBvh4Node) / 4) : sv.num_nodes * (int)(sizeof(DevBvhNode) / 4);
        const int pw = sv.num_prims * (int)(sizeof(DevPrim) / 4);
        const int tw = sv.num_tris * (int)(sizeof(DevTriShade) / 4), mw = sv.num_materials * (int)(sizeof(GdptMaterial) / 4);
        unsigned *dst = (unsigned *)s_scene;
        const unsigned *s0 = WIDE ? (const unsigned *)sv.nodes4 : (const unsigned *)sv.nodes;
        const unsigned *s1 = (const unsigned *)sv.prims, *s2 = (const unsigned *)sv.tris, *s3 = (const unsigned *)sv.materials;
        for (int i = tid; i < nw; i += nthreads) dst[i] = s0[i];
        for (int i = tid; i < pw; i += nthreads) dst[nw + i] = s1[i];
        for (int i = tid; i < tw; i += nthreads) dst[nw + pw + i] = s2[i];
        for (int i = tid; i < mw; i += nthreads) dst[nw + pw + tw + i] = s3[i];
        // the light table too: a global load in the loop makes its s_waitcnt vmcnt also wait for the acknowledgement of
        // every partial-sum store issued before it (one in-order counter for loads and stores on gfx9)
        const int l0 = (nw + pw + tw + mw + 1) & ~1, lw = sv.num_lights * 6;
        const unsigned *s4 = (const unsigned *)sv.light_intensity;
        for (int i = tid; i < lw; i += nthreads) dst[l0 + i] = s4[i];
        __syncthreads();
        tx.lights = (const double *)(s_scene + (size_t)l0 * 4);
        tx.nodes = (const DevBvhNode *)s_scene; tx.nodes4 = (const DevBvh4Node *)s_scene; tx.nodes8 = nullptr; tx.nodes4q = nullptr;
        tx.prims = (const DevPrim *)(s_scene + (size_t)nw * 4);
        tx.tris = (const DevTriShade *)(s_scene + (size_t)(nw + pw) * 4);
        tx.materials = (const GdptMaterial *)(s_scene + (size_t)(nw + pw + tw) * 4);
    } else {
        tx.nodes = sv.nodes; tx.nodes4 = sv.nodes4; tx.nodes8 = sv.nodes8; tx.nodes4q = sv.nodes4q; tx.prims = sv.prims; tx.tris = sv.tris; tx.materials = sv.materials;
        tx.lights = sv.light_intensity;
    }
    return tx;
}
// (A copy of the material table in LDS was tried twice and dropped: DESIGN.md 7, "Tried and dropped".)

// Work items of the persistent kernels: item = chunk * num_slots + slot, slot = tile * 256 + pixel_in_tile, tiles = the
// reference's 16x16 tiles of the band in row-major order (ragged edge tiles keep all 256 slots; their outside pixels
// are empty items). A wave's batch of 64 consecutive items covers 64 neighbouring pixels (a 16x4 quarter tile) of one
// chunk. Chunks are handed out chunk-major and get SHORTER along the queue (render_kernels.hip: make_chunk_plan), so the
// items still in flight when the queue runs dry are single samples: the kernel's drain is one short sample long
// instead of one long item (measured: 0.44 ms -> see DESIGN.md 4.1). One 32-bit division per started item for the
// chunk, one for the tile row. Returns false for an empty slot.
// `chunks`: a.chunk_begin, or the block's LDS copy of it (lane machine: no global load in the loop, see setup_trace).
GD bool item_to_pixel(const KernelArgs &a, int W, unsigned item, int &x, int &y, int &s0, int &s1, const int *chunks) {
    const unsigned c = item / (unsigned)a.num_slots, pt = item - c * (unsigned)a.num_slots;
    const unsigned pin = pt & 255u, tile = pt >> 8;
    const unsigned ty = tile / (unsigned)a.tiles_x, tx = tile - ty * (unsigned)a.tiles_x;
    x = (int)(tx * 16u + (pin & 15u)); y = a.row_begin + (int)(ty * 16u + (pin >> 4));
    s0 = chunks[c]; s1 = chunks[c + 1];
    return x < W && y < a.row_end;
}

// A wave's view of the global work queue. Every lane of the wave calls take(); idle lanes may receive an item index.
// Slices are fetched with one returning atomicAdd per refill (a round trip to the L2 atomic unit: 2-3 k cycles of the
// wave, measured with the stamped build): 64 items while at least one full slice per wave is left, i.e. for all but the
// last waves*64 items; in that tail only what the idle lanes (and half a fair share of the remainder) can start now, so
// that no wave sits on unstarted items while others have run dry. (Shrinking the slices from half-way through the
// queue, as an earlier version did, made the second half of the kernel pay the round trip on almost every step:
// 15 % of all wave cycles on cbox 512x512x16.)
struct WaveQueue {
    long long next = 0, end = 0, seen_head = 0;
    bool exhausted = false;
    GD long long take(const KernelArgs &a, bool idle, int tid) {
        long long item = -1;
        const unsigned long long m_idle = __ballot(idle);
        if (m_idle) {
            if (next >= end && !exhausted) {
                const unsigned left = (unsigned)(a.num_items - seen_head);          // num_items < 2^32
                const unsigned waves = gridDim.x * (unsigned)(kBlock / 64);
                const long long n_idle_now = __popcll(m_idle);
                long long want = 64;
                if (left < waves * 64u) { want = (long long)(left / (waves * 2u)); want = want > 64 ? 64 : (want < n_idle_now ? n_idle_now : want); }
                unsigned long long got = 0;
                if ((tid & 63) == 0) got = atomicAdd(a.queue_head, (unsigned long long)want);
                got = __shfl(got, 0, 64);
                next = (long long)got;
                end = min((long long)got + want, a.num_items);
                seen_head = end;
                if (next >= a.num_items) { exhausted = true; end = next; }
            }
            const int avail = (int)(end - next);
            const int rank = __popcll(m_idle & ((1ull << (tid & 63)) - 1ull));
            if (idle && rank < avail) item = next + rank;
            const int n_idle = __popcll(m_idle);
            next += (n_idle < avail) ? n_idle : avail;
        }
        return item;
    }
};

// SAMPLE stream, persistent threads. Every lane repeatedly takes a work item (pixel, chunk of the pixel's samples)
// from a global queue — fetched 64 at a time per wave with one atomicAdd — and runs the lane machine on it; a lane
// that finishes early picks up the next item instead of idling behind the longest path of its wave. Per-item sums go
// to `partials` ([15][items], one writer per slot) and are merged per pixel in chunk order by gdpt_reduce_partials,
// so the result does not depend on which lane processed what, or when.
template <bool LAMBERT, bool LDS_SCENE, bool WIDE, bool WW, bool STAMPED = false, int PLAIN = 0>
__global__ __launch_bounds__(kBlock, 2) void gdpt_render_phases(DevSceneView sv, KernelArgs a) {
    // LDS-resident scenes: fixed 12-slot stack + the scene copy. Scenes walked from HBM: dynamic LDS = a.stack_levels stack
    // slots per lane (the tree's own bound, not the builder's maximum of 32).
    __shared__ int s_stack_fixed[LDS_SCENE ? kLdsSceneLevels * kBlock : 1];
    __shared__ __attribute__((aligned(16))) unsigned char s_scene[LDS_SCENE ? kLdsSceneBytes : 16];
    __shared__ double s_acc[15 * kBlock];
    __shared__ double s_priv[kPrivDoubles * kBlock];
    __shared__ int s_chunks[66];
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    int *s_stack = LDS_SCENE ? s_stack_fixed : (int *)s_dyn;
    const int tid = threadIdx.x;
    if (tid <= a.num_chunks && tid < 66) s_chunks[tid] = a.chunk_begin[tid];       // LDS copy of the chunk table (item_to_pixel)
    TraceCtx tx = setup_trace<LDS_SCENE, WIDE>(sv, s_scene, s_stack, tid, kBlock, a.count != 0);
    if (!LDS_SCENE) __syncthreads();                                                // (for LDS scenes setup_trace ends with the barrier)
    const int W = sv.cam.width;
    const double spp = (double)a.spp;
    AccLds acc; acc.slot = s_acc + tid; acc.stride = kBlock;
    acc.init();
    LanePriv lp; lp.slot = s_priv + tid; lp.stride = kBlock;
    LaneCounters lc = {0, 0, 0};
    TraceCounters tc = {0, 0, 0, 0, 0, 0};
    Lane L;
    Trav tv;
    trav_init(sv, tv, __builtin_huge_val());
    L.s = 0; L.s_end = 0; L.st = S_DONE;
    L.kc = 0; L.num_vertices = 0; L.mats = 0xFFFFFF; L.rng_state = 0; L.rng_inc = 1;
    L.org = L.dir = splat(0);
    int x = 0, y = 0;
    unsigned long long base = 0;
    long long my_item = -1;
    WaveQueue wq;
    __shared__ unsigned long long s_stamps[STAMPED ? (kBlock / 64) * (SEG_COUNT + 1) : 1];
    Stamps<STAMPED> stamps;
    stamps.start(s_stamps + (STAMPED ? (tid >> 6) * (SEG_COUNT + 1) : 0));
    unsigned long long t_dry = ~0ull;        // diagnostic build: wall clock (100 MHz) when this wave first found the queue empty
    if (STAMPED && (tid & 63) == 0) atomicMin(&a.counters->stamps[SEG_T_START], __builtin_amdgcn_s_memrealtime());
    // Work is handed out BETWEEN the two halves of a step in the Lambertian kernels (QUEUE_MID): a lane whose item has just
    // ended gets the first camera ray of its next item in this step and sits out no trace phase (cbox 16 spp +2.8 %, sponza
    // +2.4 %). The kernels with the full material switch keep the queue at the head of the loop: there the same move cost
    // 3-5 % (one more block of live state across the BSDF switch; measured on disney_metal / disney_diffuse, same box).
    constexpr bool QUEUE_MID = LAMBERT;
    auto hand_out = [&](int &act) __attribute__((always_inline)) {
        const bool idle = (L.st == S_DONE);
        if (idle && my_item >= 0) {                     // item finished: publish its 15 sums, clear the slot
            Accum r = acc.result();
            // one 128-byte record per item (15 sums + pad), written as eight 16-byte stores: a single HBM line
            typedef double d2 __attribute__((ext_vector_type(2)));
            d2 *dst = (d2 *)(a.partials + (size_t)my_item * 16);
            dst[0] = d2{r.r.x, r.r.y}; dst[1] = d2{r.r.z, r.dx0.x}; dst[2] = d2{r.dx0.y, r.dx0.z}; dst[3] = d2{r.dy0.x, r.dy0.y};
            dst[4] = d2{r.dy0.z, r.dx1.x}; dst[5] = d2{r.dx1.y, r.dx1.z}; dst[6] = d2{r.dy1.x, r.dy1.y}; dst[7] = d2{r.dy1.z, 0.0};
            acc.init();
            my_item = -1;
        }
        stamps.mark(SEG_PUBLISH);
        const long long got_item = wq.take(a, idle, tid);
        stamps.mark(SEG_TAKE);
        if (STAMPED && wq.exhausted && t_dry == ~0ull) t_dry = __builtin_amdgcn_s_memrealtime();
        if (got_item >= 0) {
            my_item = got_item;
            int s0, s1;
            const bool inside = item_to_pixel(a, W, (unsigned)my_item, x, y, s0, s1, s_chunks);
            base = ((unsigned long long)y * W + x) * (unsigned long long)a.spp;
            L.s = s0; L.s_end = s1;
            if (inside && s0 < s1) act = ACT_PRIMARY_RAY;       // (an empty slot of a ragged edge tile stays S_DONE and is published as zeros)
        }
        stamps.mark(SEG_ITEM);
    };
    for (;;) {
        int act = ACT_NONE;
        if (!QUEUE_MID) {
            hand_out(act);
            if (act == ACT_PRIMARY_RAY) L.st = S_START;         // lane_consume turns it into the first camera ray
            if (!__any(L.st != S_DONE)) { if (wq.exhausted) break; else continue; }
            stamps.mark(SEG_QUEUE);
        }
        // ---- (T) the wave's unfinished pending rays, then (S, first half) the lanes whose ray is done consume their hit
        trace_pending<TraceCfg<WW, WIDE, !LDS_SCENE, !(PLAIN & kPlainNoSpheres), !LDS_SCENE>>(sv, tx, L, tv, a.thresh_a, a.thresh_c, tc);
        stamps.mark(SEG_TRACE);
        stamps.tick(SEG_STEPS);
        act = ACT_NONE;
        if (lane_ready(L, tv)) {
            if (tx.count) { tc.lane_steps++; if (wave_leader()) tc.wave_steps++; }
            act = lane_consume<LAMBERT, false, AccLds, Stamps<STAMPED>, PLAIN>(sv, tx, a.max_depth, spp, base, L, tv, lp, acc, lc, tc, &stamps);
        }
        if (QUEUE_MID) hand_out(act);
        // ---- (S, second half) camera rays: next sample, next offset, first sample of a new item
        lane_camera<false, Stamps<STAMPED>>(sv, act, x, y, base, L, tv, lp, &stamps);
        if (QUEUE_MID) {
            if (!__any(L.st != S_DONE) && wq.exhausted) break;
            stamps.mark(SEG_QUEUE);
        }
    }
    flush_counters(a, lc, tc, a.count != 0);
    if (STAMPED && (tid & 63) == 0) {
        for (int i = 0; i < SEG_COUNT; i++) atomicAdd(&a.counters->stamps[i], stamps.get(i));
        atomicMin(&a.counters->stamps[SEG_T_DRY], t_dry);
        atomicMax(&a.counters->stamps[SEG_T_END], __builtin_amdgcn_s_memrealtime());
    }
}

#ifdef GDPT_BUILD_REDUCE   // emitted by render_phases_lambert.hip only (non-template kernel)
// Sums the C per-chunk partials of every pixel in chunk order and writes the five images (one thread per pixel).
__global__ __launch_bounds__(256) void gdpt_reduce_partials(KernelArgs a, int W) {
    // 16 consecutive threads per pixel slot: thread j sums component j of the slot's records in chunk order
    const long long nslots = a.num_slots;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long idx = t >> 4;                                        // pixel slot
    const int j = (int)(t & 15);
    if (idx >= nslots || j == 15) return;
    int x, y, s0, s1;
    if (!item_to_pixel(a, W, (unsigned)idx, x, y, s0, s1, a.chunk_begin)) return;
    const double *src = a.partials + (size_t)idx * 16 + j;
    double v = 0;
    for (int c = 0; c < a.num_chunks; c++) v += src[(size_t)c * (size_t)nslots * 16];
    double *img = (j < 3) ? a.img : (j < 6) ? a.cx0 : (j < 9) ? a.cy0 : (j < 12) ? a.cx1 : a.cy1;
    img[((size_t)y * W + x) * 3 + (j % 3)] = v;
}
#endif // GDPT_BUILD_REDUCE

// TILE stream: bit-for-bit the reference's RNG order — one PCG stream per 16x16 tile, pixels y-outer / x-inner,
// samples innermost (src/render.cpp:281-309). Serial per tile => one lane per tile. For checks.
template <bool LAMBERT>
__global__ __launch_bounds__(64) void gdpt_render_tile_stream_phases(DevSceneView sv, KernelArgs a, int ntx, int nty) {
    __shared__ int s_stack[GDPT_BVH_MAX_DEPTH * 64];
    __shared__ double s_priv[kPrivDoubles * 64];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x * 64 + tid;
    TraceCtx tx = setup_trace<false, true>(sv, nullptr, s_stack, tid, 64, a.count != 0);
    LanePriv lp; lp.slot = s_priv + tid; lp.stride = 64;
    LaneCounters lc = {0, 0, 0};
    TraceCounters tc = {0, 0, 0, 0, 0, 0};
    const int W = sv.cam.width, H = sv.cam.height;
    const double spp = (double)a.spp;
    if (tile < ntx * nty) {
        const int txi = tile % ntx, tyi = tile / ntx;
        Lane L;
        { Pcg r0 = pcg_init((unsigned long long)(tyi * ntx + txi)); L.rng_state = r0.state; L.rng_inc = r0.inc; }
        L.kc = 0; L.num_vertices = 0; L.mats = 0xFFFFFF;
        L.org = L.dir = splat(0);
        Trav tv;
        trav_init(sv, tv, __builtin_huge_val());
        const int x0 = txi * 16, x1 = min(x0 + 16, W), y0 = tyi * 16, y1 = min(y0 + 16, H);
        for (int y = y0; y < y1; y++) {
            if (y < a.row_begin || y >= a.row_end) continue;   // bands are whole tile rows in this mode
            for (int x = x0; x < x1; x++) {
                AccReg acc; acc.init();
                L.s = 0; L.s_end = a.spp; L.st = S_START;
                while (L.st != S_DONE) {
                    if (lane_tracing(L.st) && tv.cur != kTravDone)
                        trav_run<TraceHbm>(sv, tx, L.org, L.dir, (L.st == S_BOUNCE) ? (float)sv.isect_eps : 0.0f, __builtin_huge_valf(), tv, 0, 0, tc);
                    lane_step<LAMBERT, true>(sv, tx, a.max_depth, spp, x, y, 0ull, L, tv, lp, acc, lc, tc);
                }
                Accum sum = acc.result();
                reduce_and_store(a, sum, 1, true, x, y, W);
            }
        }
    }
    flush_counters(a, lc, tc, a.count != 0);
}

#ifdef GDPT_BUILD_EAGER   // only render_eager.hip emits these (non-template) kernels
// ------------------------------------------------------------------------------------------------
// eager evaluator (any material set, offsets carried across bounces in private memory)
// ------------------------------------------------------------------------------------------------
struct Offset { Vertex v; D3 dir; D3 contrib; double jacob; };
struct SampleOut { D3 radiance, contrib; D3 cX[4]; double w[4]; double prob; };

GD void zero_out(SampleOut &o) {
    o.radiance = splat(0); o.contrib = splat(0);
#pragma unroll
    for (int k = 0; k < 4; k++) { o.cX[k] = splat(0); o.w[k] = 1.0; }
    o.prob = 1.0;
}

GD void grad_sample_eager(const DevSceneView &sv, const TraceCtx &tx, int max_depth, int x, int y, Pcg &rng,
                          SampleOut &out, LaneCounters &lc, TraceCounters &tc) {
    const DevCamera &cam = sv.cam;
    const int w = cam.width, h = cam.height;
    zero_out(out);
    double rng_x = pcg_real(rng), rng_y = pcg_real(rng);
    Ray ray = sample_primary(cam, (x + rng_x) / w, (y + rng_y) / h);
    const double rd_spread = 0.25 / (double)max(w, h);
    Vertex vertex;
    if (!intersect_ctx<TraceHbm>(sv, tx, ray, rd_spread, vertex, lc, tc)) return;
    Offset off[4];
    unsigned alive = 0;
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        int ox = (k == 0) ? -1 : (k == 1 ? 1 : 0), oy = (k == 2) ? 1 : (k == 3 ? -1 : 0);
        Ray r = sample_primary(cam, ((x + ox) + rng_x) / w, ((y + oy) + rng_y) / h);
        Vertex ov;
        bool ok = intersect_ctx<TraceHbm>(sv, tx, r, rd_spread, ov, lc, tc);
        if (ok && ov.material_id == vertex.material_id) {
            alive |= 1u << k;
            off[k].v = ov; off[k].dir = r.dir; off[k].jacob = 1.0;
            off[k].contrib = (ov.light_id >= 0) ? emission(sv, ov, -r.dir) : splat(1.0);
        }
    }
    D3 contrib = splat(1.0), throughput = splat(1.0), radiance = splat(0);
    double prob = 1.0, eta_scale = 1.0;
    if (vertex.light_id >= 0) { D3 L = emission(sv, vertex, -ray.dir); radiance = L; contrib = L; }
    for (int num_vertices = 3; loop_allows(max_depth, num_vertices); num_vertices++) {
        lc.bounces++;
        const GdptMaterial &mat = sv.materials[vertex.material_id];
        D3 dir_view = -ray.dir;
        D2 ruv; ruv.x = pcg_real(rng); ruv.y = pcg_real(rng);
        double rw = pcg_real(rng);
        BsdfSample bs;
        if (!bsdf_sample(sv, mat, dir_view, vertex, ruv, rw, bs)) { zero_out(out); return; }
        D3 dir_bsdf = bs.dir_out;
        if (bs.eta != 0) eta_scale /= (bs.eta * bs.eta);
        Ray bsdf_ray; bsdf_ray.org = vertex.position; bsdf_ray.dir = dir_bsdf; bsdf_ray.tnear = sv.isect_eps; bsdf_ray.tfar = __builtin_huge_val();
        Vertex bsdf_vertex;
        bool hit = intersect_ctx<TraceHbm>(sv, tx, bsdf_ray, 0.0, bsdf_vertex, lc, tc);
        if (alive) {
#pragma unroll 1
            for (int k = 0; k < 4; k++)
                if ((alive >> k & 1u) && off[k].v.material_id != vertex.material_id) alive &= ~(1u << k);
        }
        double G = 1.0;
        if (hit) { D3 dl = bsdf_vertex.position - vertex.position; G = fabs(dot(dir_bsdf, bsdf_vertex.gn)) / dot(dl, dl); }
        D3 f = bsdf_eval(sv, mat, dir_view, dir_bsdf, vertex);
        double p2 = bsdf_pdf(sv, mat, dir_view, dir_bsdf, vertex);
        if (p2 <= 0) break;
        p2 *= G;
        contrib = contrib * f * G;
        prob *= p2;
        if (alive) {
#pragma unroll 1
            for (int k = 0; k < 4; k++) {
                if (!(alive >> k & 1u)) continue;
                Offset &o = off[k];
                const GdptMaterial &omat = sv.materials[o.v.material_id];
                D3 oin = -o.dir;
                BsdfSample os;
                if (!bsdf_sample(sv, omat, oin, o.v, ruv, rw, os)) { alive &= ~(1u << k); continue; }
                double p2o = bsdf_pdf(sv, omat, oin, os.dir_out, o.v);
                if (p2o <= 0.0) { alive &= ~(1u << k); continue; }
                o.jacob *= p2 / p2o;
                o.dir = os.dir_out;
            }
        }
        if (hit && bsdf_vertex.light_id >= 0) {
            D3 L = emission(sv, bsdf_vertex, -dir_bsdf);
            D3 C2 = (G * f) * L;
            contrib = contrib * L;
            radiance = radiance + throughput * (C2 / p2);
        }
        if (!hit) break;
        double rr_prob = 1;
        if (num_vertices - 1 >= sv.rr_depth) {
            rr_prob = fmin(maxc((1 / eta_scale) * throughput), 0.95);
            if (pcg_real(rng) > rr_prob) break;
        }
        ray = bsdf_ray;
        vertex = bsdf_vertex;
        throughput = throughput * (G * f) / (p2 * rr_prob);
    }
    out.radiance = radiance; out.contrib = contrib; out.prob = prob;
    if (alive) {
#pragma unroll 1
        for (int k = 0; k < 4; k++)
            if (alive >> k & 1u) { out.cX[k] = off[k].contrib * off[k].jacob; out.w[k] = prob / (prob + 1.0 * off[k].jacob); }
    }
}

GD void accumulate_eager(AccReg &a, const SampleOut &s, double spp, LaneCounters &lc) {
    acc_base(a, s.radiance, s.prob, spp, lc);
    bool flagged = false;
#pragma unroll
    for (int k = 0; k < 4; k++) acc_offset(a, k, s.contrib, s.cX[k], s.w[k], s.prob, spp, lc, flagged);
}

__global__ __launch_bounds__(kBlock, 2) void gdpt_render_eager(DevSceneView sv, KernelArgs a) {
    __shared__ int s_stack[GDPT_BVH_MAX_DEPTH * kBlock];
    const int tid = threadIdx.x;
    TraceCtx tx = setup_trace<false, true>(sv, nullptr, s_stack, tid, kBlock, a.count != 0);
    const int K = 1 << a.log2k;
    const int c = tid & (K - 1), p = tid >> a.log2k;
    const int px = p % a.tile_w, py = p / a.tile_w;
    const int bx = blockIdx.x % a.tiles_x, by = blockIdx.x / a.tiles_x;
    const int x = bx * a.tile_w + px, y = a.row_begin + by * a.tile_h + py;
    const int W = sv.cam.width;
    const bool valid = (x < W) && (y < a.row_end);
    AccReg acc; acc.init();
    LaneCounters lc = {0, 0, 0};
    TraceCounters tc = {0, 0, 0, 0, 0, 0};
    if (valid) {
        const int s0 = (int)(((long long)c * a.spp) >> a.log2k), s1 = (int)(((long long)(c + 1) * a.spp) >> a.log2k);
        const unsigned long long base = ((unsigned long long)y * W + x) * (unsigned long long)a.spp;
        for (int s = s0; s < s1; s++) {
            Pcg rng = pcg_init(base + (unsigned long long)s);
            SampleOut so;
            grad_sample_eager(sv, tx, a.max_depth, x, y, rng, so, lc, tc);
            accumulate_eager(acc, so, (double)a.spp, lc);
        }
    }
    Accum sum = acc.result();
    reduce_and_store(a, sum, K, valid && c == 0, x, y, W);
    flush_counters(a, lc, tc, a.count != 0);
}

__global__ __launch_bounds__(64) void gdpt_render_tile_stream_eager(DevSceneView sv, KernelArgs a, int ntx, int nty) {
    __shared__ int s_stack[GDPT_BVH_MAX_DEPTH * 64];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x * 64 + tid;
    TraceCtx tx = setup_trace<false, true>(sv, nullptr, s_stack, tid, 64, a.count != 0);
    LaneCounters lc = {0, 0, 0};
    TraceCounters tc = {0, 0, 0, 0, 0, 0};
    const int W = sv.cam.width, H = sv.cam.height;
    if (tile < ntx * nty) {
        const int txi = tile % ntx, tyi = tile / ntx;
        Pcg rng = pcg_init((unsigned long long)(tyi * ntx + txi));
        const int x0 = txi * 16, x1 = min(x0 + 16, W), y0 = tyi * 16, y1 = min(y0 + 16, H);
        for (int y = y0; y < y1; y++) {
            if (y < a.row_begin || y >= a.row_end) continue;
            for (int x = x0; x < x1; x++) {
                AccReg acc; acc.init();
                for (int s = 0; s < a.spp; s++) {
                    SampleOut so;
                    grad_sample_eager(sv, tx, a.max_depth, x, y, rng, so, lc, tc);
                    accumulate_eager(acc, so, (double)a.spp, lc);
                }
                Accum sum = acc.result();
                reduce_and_store(a, sum, 1, true, x, y, W);
            }
        }
    }
    flush_counters(a, lc, tc, a.count != 0);
}

#endif // GDPT_BUILD_EAGER

} // namespace gd

namespace gdpt {
// host launchers, one translation unit per kernel family (parallel compilation)
void launch_phases_lambert(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, bool lds, bool lds_wide, hipStream_t stream);
void launch_phases_lambert_plain(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, bool lds, bool const_tex, hipStream_t stream);   // triangles only (and constant textures)
void launch_phases_lambert_stamped(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, bool lds, bool plain, hipStream_t stream);   // diagnostic build
void launch_reduce_partials(const DevSceneView &sv, const gd::KernelArgs &a, hipStream_t stream);
void launch_phases_general(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, bool lds, bool lds_wide, hipStream_t stream);
// kernels built for {Lambertian, one Disney lobe} (render_phases_general_sets_*.hip): false if the scene's set is not one of theirs
bool launch_phases_general_set_a(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, unsigned material_mask, hipStream_t stream);
bool launch_phases_general_set_b(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, unsigned material_mask, hipStream_t stream);
void launch_phases_twosided(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, bool lds, unsigned material_mask, void *bounce_log, hipStream_t stream);
void launch_tile_phases_lambert(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, int ntx, int nty, hipStream_t stream);
void launch_tile_phases_general(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, int ntx, int nty, hipStream_t stream);
void launch_eager(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, hipStream_t stream);
void launch_tile_eager(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, int ntx, int nty, hipStream_t stream);
void launch_reconnect(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, bool lds, bool lambert, hipStream_t stream);
void launch_path(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, hipStream_t stream);
void launch_path_persistent(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, bool lds, bool lambert, bool plain, hipStream_t stream);
void launch_tile_path(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, int ntx, int nty, hipStream_t stream);
} // namespace gdpt
