// render_wavefront.h — the GradPath tile loop as a wavefront pipeline, for scenes that live in HBM.
//
// The lane machine of render_device.h keeps a path's whole state in registers / LDS and therefore runs at 2 waves per
// SIMD. That is the right trade for a scene that fits LDS (cbox), but a scene walked from HBM (sponza: 19 BVH4 nodes
// and 7 triangles per ray out of L2 / Infinity Cache) then spends half of every wave's life in s_waitcnt with nothing
// to switch to. Here the same per-sample program (lane_step, unchanged: same arithmetic, same RNG streams, same
// per-item summation order, so the images are bit-identical to the lane machine's) is cut at the ray:
//
//   gdpt_wf_step     one thread per path slot: take a work item if idle, consume the slot's hit record (rebuild the
//                    vertex, run the state arm, sample the BSDF or the camera), store the state, and emit the slot's next
//                    ray as a 32-byte fp32 record (exactly what the walk consumes: src/intersection.cpp:15-24 hands
//                    Embree fp32 rays too) together with its sort key. State lives in HBM between steps, SoA
//                    ([word][slot], 8-byte words), so loads and stores coalesce.
//   gdpt_wf_scan     counting sort of the generation's rays, part 2 (part 1, the histogram with the ray's rank inside
//   gdpt_wf_scatter  its bin, rides on the step kernel): bin offsets, then slot indices into the ray queue in key order.
//                    Key = pixel tile for camera rays, (direction octant, Morton code of the origin's cell on a 16^3
//                    grid over the scene bounds) for bounce rays: a wave of the trace kernel gets 64 rays that start
//                    in the same part of the scene and point into the same octant, so their node fetches share cache
//                    lines and their loops stay together.
//   gdpt_wf_trace    one lane = one ray and NOTHING else: fp32 origin / direction, best hit, node index, stack pointer.
//                    The walk holds one BVH4 node or one triangle record at a time (<= 64 VGPRs), the stack's first
//                    16 levels sit in LDS and the (rare) rest in HBM, so 8 waves per SIMD are resident and a node fetch
//                    that goes to L2 or the Infinity Cache is covered by the other seven. Closest hit = min fp32 t,
//                    ties to the lowest primitive id (device_trace.h), i.e. independent of traversal order: the hit
//                    records equal the lane machine's bit for bit.
//
// One "generation" = step + scan + scatter + trace. Generations are enqueued back to back and the host looks at the
// live count one chunk of generations behind the launches.
//
// Reference: the tile loop src/render.cpp:277-331 over grad_path_tracing src/path_tracing.h:354-1050 (as
// render_device.h); the walk replaces rtcIntersect1 behind intersect(), src/intersection.cpp:7-65.
#pragma once
#include "render_device.h"

namespace gd {

// ---- state layout: 8-byte words, word w of slot s at state[w * N + s] ------------------------------------------
enum {
    WF_RNG = 0,                 // u64 PCG state (the increment follows from pixel + sample index)
    WF_ORG = 1, WF_DIR = 4,     // pending ray, fp64 (3 + 3): the vertex is rebuilt from these, not from the fp32 record
    WF_F = 7, WF_PDF = 10,      // f*|cos| and pdf of the pending bounce ray
    WF_CONTRIB = 11, WF_THROUGHPUT = 14, WF_PROB = 17,
    WF_I0 = 18,                 // st | s << 32
    WF_I1 = 19,                 // s_end | num_vertices << 32
    WF_I2 = 20,                 // mats | kc << 32
    WF_XY = 21,                 // x | y << 32
    WF_ITEM = 22,               // item (u32, ~0 = none)
    WF_PRIV = 23,               // LanePriv: 8 doubles (touched on emitter hits, at bounce 1 and around the offsets)
    WF_ACC = 31,                // 15 running sums of the slot's current item (touched when a sample ends)
    WF_WORDS = 46
};

// counting sort of a generation's rays
constexpr unsigned kWfCamBins = 8192;                       // camera rays: 16x16 pixel tile of the film (mod 8192)
constexpr unsigned kWfBounceBins = 8u * 4096u;              // bounce rays: octant x 16^3 origin cells
constexpr unsigned kWfBins = kWfCamBins + kWfBounceBins;    // 40960 = 1024 x 40
constexpr int kWfScanBlock = 1024;
constexpr unsigned kWfNoKey = 0xFFFFFFFFu;
enum { WF_SORT_NONE = 0, WF_SORT_OCTANT_MAJOR = 1, WF_SORT_CELL_MAJOR = 2 };

struct WfBuf {
    unsigned long long *state;      // WF_WORDS * N
    float4 *rays;                   // 2 per slot: (o.xyz, tnear), (d.xyz, -)
    float4 *hits;                   // 2 per slot: (t, u, v, gid as bits), (ng.xyz, -) — the second for sphere hits only
    uint2 *keys;                    // per slot: (sort key or kWfNoKey, rank inside the key's bin)
    unsigned *hist;                 // kWfBins counters, all zero between generations
    unsigned *offsets;              // kWfBins bin starts of the current generation
    unsigned *live;                 // N: the generation's ray queue (slot indices), written by step or scatter, read by trace
    unsigned *counters;             // [3][kWfMaxGen]: rays in the queue, (unused), active slots (hold a ray or must publish)
    uint4 *block_counts;            // per block of the step kernel: (active slots, rays consumed, bounces, non-finite samples)
    gdpt::RenderCounters *totals;   // the render's counters (the scan kernel adds the generation's block counts)
    int n;                          // slots (multiple of 256)
    int gen;                        // generation of this launch
    int sort;                       // WF_SORT_*
    int tiles_x;                    // 16-pixel tiles per film row (camera-ray key)
    float bmin[3], cell_scale[3];   // bounce-ray key: cell = (org - bmin) * cell_scale, clamped to 0..15
};
constexpr int kWfMaxGen = 8192;

struct AccMem {                     // the slot's 15 sums in HBM: one owner, plain read-modify-write
    double *slot; long long stride;
    GD void init() { for (int c = 0; c < 15; c++) slot[c * stride] = 0.0; }
    GD void add(int which, D3 v) {
        // the addend must arrive rounded, as it does at the lane machine's ds_add_f64: no contraction of the multiply
        // that produced it into an FMA with this sum (__dadd_rn is a plain + in HIP's headers and does not stop that),
        // or the two pipelines differ in the last bit wherever an item holds more than one sample
#pragma clang fp contract(off)
        double *p = slot + which * 3 * stride;
        const double a0 = p[0], a1 = p[stride], a2 = p[2 * stride];
        p[0] = a0 + v.x; p[stride] = a1 + v.y; p[2 * stride] = a2 + v.z;
    }
};

GD unsigned long long pack2(unsigned lo, unsigned hi) { return (unsigned long long)lo | ((unsigned long long)hi << 32); }
GD unsigned spread4(unsigned v) { return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6); }   // bit i -> bit 3 i

// Sort key of a slot's pending ray (any key gives the same images: the trace kernel's results do not depend on which
// lane walks which ray; the key only decides which rays share a wave).
GD unsigned wf_ray_key(const WfBuf &w, int st, int x, int y, D3 org, D3 dir) {
    if (st != S_BOUNCE) return (unsigned)((y >> 4) * w.tiles_x + (x >> 4)) & (kWfCamBins - 1u);
    unsigned c[3];
    const float o[3] = {(float)org.x, (float)org.y, (float)org.z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float f = (o[k] - w.bmin[k]) * w.cell_scale[k];
        c[k] = (unsigned)fminf(fmaxf(f, 0.0f), 15.0f);            // NaN -> 0
    }
    const unsigned morton = spread4(c[0]) | (spread4(c[1]) << 1) | (spread4(c[2]) << 2);
    const unsigned oct = (dir.x < 0 ? 1u : 0u) | (dir.y < 0 ? 2u : 0u) | (dir.z < 0 ? 4u : 0u);
    return kWfCamBins + (w.sort == WF_SORT_CELL_MAJOR ? ((morton << 3) | oct) : ((oct << 12) | morton));
}

// One step of every slot. N threads; a wave's 64 slots are neighbours, so every state access is a coalesced 512-byte row.
template <bool LAMBERT>
__global__ __launch_bounds__(kBlock) void gdpt_wf_step(DevSceneView sv, KernelArgs a, WfBuf w) {
    const int tid = threadIdx.x;
    const long long N = w.n;
    const long long slot = (long long)blockIdx.x * kBlock + tid;           // grid covers exactly N slots
    unsigned long long *S = w.state + slot;
    auto ld = [&](int word) { return S[(long long)word * N]; };
    auto st_ = [&](int word, unsigned long long v) { S[(long long)word * N] = v; };
    auto ldd = [&](int word) { return __longlong_as_double((long long)S[(long long)word * N]); };
    auto std_ = [&](int word, double v) { S[(long long)word * N] = (unsigned long long)__double_as_longlong(v); };

    const int W = sv.cam.width;
    const double spp = (double)a.spp;
    LaneCounters lc = {0, 0, 0};
    TraceCounters tc = {0, 0, 0, 0, 0, 0};
    TraceCtx tx;
    tx.count = false; tx.need_uv = !sv.all_textures_constant;
    tx.stack = nullptr; tx.stride = 0;
    tx.nodes = sv.nodes; tx.nodes4 = sv.nodes4; tx.nodes8 = sv.nodes8; tx.nodes4q = sv.nodes4q; tx.prims = sv.prims; tx.tris = sv.tris; tx.materials = sv.materials; tx.lights = sv.light_intensity;
    AccMem acc; acc.slot = (double *)(S + (long long)WF_ACC * N); acc.stride = N;
    LanePriv lp; lp.slot = (double *)(S + (long long)WF_PRIV * N); lp.stride = (int)N;

    Lane L;
    const unsigned long long i0 = ld(WF_I0);
    L.st = (int)(unsigned)i0; L.s = (int)(i0 >> 32);
    const unsigned itw = (unsigned)ld(WF_ITEM);
    long long my_item = (itw == 0xFFFFFFFFu) ? -1 : (long long)itw;
    unsigned long long i1 = 0, xy = 0;
    int x = 0, y = 0;

    // ---- idle slots publish their finished item and take the next one (one atomicAdd per block)
    const bool idle = (L.st == S_DONE);
    if (idle && my_item >= 0) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        double r[16];
        for (int c = 0; c < 15; c++) r[c] = acc.slot[c * N];
        r[15] = 0.0;
        d2 *dst = (d2 *)(a.partials + (size_t)my_item * 16);
        for (int k = 0; k < 8; k++) dst[k] = d2{r[2 * k], r[2 * k + 1]};
        my_item = -1;
    }
    bool fresh = false;
    {
        // one atomicAdd per BLOCK and generation (a single queue word serves about 90 returning atomics per
        // microsecond: one per wave would cost 32 k of them = 0.4 ms per generation), none once the queue is empty
        __shared__ unsigned s_idle[kBlock / 64];
        __shared__ unsigned long long s_base;
        const unsigned long long m_idle = __ballot(idle);
        const int wv = tid >> 6;
        if ((tid & 63) == 0) s_idle[wv] = (unsigned)__popcll(m_idle);
        __syncthreads();
        if (tid == 0) {
            const unsigned total = s_idle[0] + s_idle[1] + s_idle[2] + s_idle[3];
            unsigned long long got = (unsigned long long)a.num_items;
            if (total && __hip_atomic_load(a.queue_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)a.num_items)
                got = atomicAdd(a.queue_head, (unsigned long long)total);
            s_base = got;
        }
        __syncthreads();
        if (m_idle) {
            unsigned before = 0;
            for (int k = 0; k < wv; k++) before += s_idle[k];
            const long long mine = (long long)s_base + before + __popcll(m_idle & ((1ull << (tid & 63)) - 1ull));
            if (idle && mine < a.num_items) {
                my_item = mine;
                int s0, s1;
                const bool inside = item_to_pixel(a, W, (unsigned)my_item, x, y, s0, s1, a.chunk_begin);
                L.s = s0; L.s_end = s1;
                L.st = (inside && s0 < s1) ? S_START : S_DONE;
                L.kc = 0; L.num_vertices = 0; L.mats = 0xFFFFFF; L.rng_state = 0; L.rng_inc = 1;
                L.org = L.dir = splat(0);
                acc.init();
                fresh = true;
                // (an empty item — a slot of a ragged edge tile — is published as zeros by the next step)
            }
        }
    }
    // (no early return: the wave-wide sums and ballots at the end need every lane)
    const bool work = (L.st != S_DONE);
    if (!work) {                                             // nothing to do: remember the item (if any)
        if (idle || fresh) { st_(WF_I0, pack2((unsigned)S_DONE, (unsigned)L.s)); st_(WF_ITEM, (unsigned long long)(my_item < 0 ? 0xFFFFFFFFu : (unsigned)my_item)); }
        if (fresh) st_(WF_XY, pack2((unsigned)x, (unsigned)y));
    } else {
        // ---- load the slot (a fresh slot has nothing to load)
        Trav tv;
        tv.cur = kTravDone; tv.sp = 0;
        tv.best.gid = -1; tv.best.t = 0; tv.best.u = tv.best.v = 0; tv.best.ngx = tv.best.ngy = tv.best.ngz = 0;
        if (!fresh) {
            const float4 h0 = w.hits[2 * slot];              // written by the trace kernel for every ray of the last generation
            i1 = ld(WF_I1); L.s_end = (int)(unsigned)i1; L.num_vertices = (int)(i1 >> 32);
            const unsigned long long i2 = ld(WF_I2); L.mats = (int)(unsigned)i2; L.kc = (int)(i2 >> 32);
            xy = ld(WF_XY); x = (int)(unsigned)xy; y = (int)(xy >> 32);
            L.rng_state = ld(WF_RNG);
            L.org = mk(ldd(WF_ORG), ldd(WF_ORG + 1), ldd(WF_ORG + 2)); L.dir = mk(ldd(WF_DIR), ldd(WF_DIR + 1), ldd(WF_DIR + 2));
            if (L.st == S_BOUNCE) { L.f = mk(ldd(WF_F), ldd(WF_F + 1), ldd(WF_F + 2)); L.pdf = ldd(WF_PDF); }
            L.contrib = mk(ldd(WF_CONTRIB), ldd(WF_CONTRIB + 1), ldd(WF_CONTRIB + 2));
            L.throughput = mk(ldd(WF_THROUGHPUT), ldd(WF_THROUGHPUT + 1), ldd(WF_THROUGHPUT + 2));
            L.prob = ldd(WF_PROB);
            tv.best.t = h0.x; tv.best.u = h0.y; tv.best.v = h0.z; tv.best.gid = (int)__float_as_uint(h0.w);
            if (tv.best.gid >= sv.num_tris) { const float4 h1 = w.hits[2 * slot + 1]; tv.best.ngx = h1.x; tv.best.ngy = h1.y; tv.best.ngz = h1.z; }
        }
        const unsigned long long base = ((unsigned long long)y * W + x) * (unsigned long long)a.spp;
        L.rng_inc = ((base + (unsigned long long)L.s) << 1u) | 1u;       // pcg_init's increment of the sample in flight

        lane_step<LAMBERT, false>(sv, tx, a.max_depth, spp, x, y, base, L, tv, lp, acc, lc, tc);

        // ---- store, and emit the slot's next ray
        st_(WF_I0, pack2((unsigned)L.st, (unsigned)L.s));
        st_(WF_I1, pack2((unsigned)L.s_end, (unsigned)L.num_vertices));
        st_(WF_I2, pack2((unsigned)L.mats, (unsigned)L.kc));
        if (fresh) { st_(WF_XY, pack2((unsigned)x, (unsigned)y)); st_(WF_ITEM, (unsigned long long)(unsigned)my_item); }
        if (L.st != S_DONE) {
            st_(WF_RNG, L.rng_state);
            std_(WF_ORG, L.org.x); std_(WF_ORG + 1, L.org.y); std_(WF_ORG + 2, L.org.z);
            std_(WF_DIR, L.dir.x); std_(WF_DIR + 1, L.dir.y); std_(WF_DIR + 2, L.dir.z);
            if (L.st == S_BOUNCE) { std_(WF_F, L.f.x); std_(WF_F + 1, L.f.y); std_(WF_F + 2, L.f.z); std_(WF_PDF, L.pdf); }
            std_(WF_CONTRIB, L.contrib.x); std_(WF_CONTRIB + 1, L.contrib.y); std_(WF_CONTRIB + 2, L.contrib.z);
            std_(WF_THROUGHPUT, L.throughput.x); std_(WF_THROUGHPUT + 1, L.throughput.y); std_(WF_THROUGHPUT + 2, L.throughput.z);
            std_(WF_PROB, L.prob);
            // the fp32 ray the walk sees (trav_run converts the same way)
            w.rays[2 * slot] = make_float4((float)L.org.x, (float)L.org.y, (float)L.org.z, (L.st == S_BOUNCE) ? (float)sv.isect_eps : 0.0f);
            w.rays[2 * slot + 1] = make_float4((float)L.dir.x, (float)L.dir.y, (float)L.dir.z, 0.0f);
        }
    }
    {
        const bool live = work && (L.st != S_DONE);          // holds a ray: goes to the trace queue
        // must be stepped again: every slot that did something (a finished item still has to be published), and a fresh
        // but empty item (a slot of a ragged edge tile: published as zeros by the next step)
        const bool again = work || (fresh && my_item >= 0);
        const unsigned long long m_live = __ballot(live), m_again = __ballot(again);
        const unsigned lane = (unsigned)(tid & 63);
        __shared__ unsigned s_cnt[kBlock / 64][5];
        __shared__ unsigned s_pos;
        const unsigned n_live = (unsigned)__popcll(m_live);
        if (w.sort != WF_SORT_NONE) {
            // counting sort, part 1: the ray's bin and its rank inside the bin. The lanes that share the first live lane's
            // key (camera rays of one pixel tile: often the whole wave) take their ranks from ONE returning atomic, the
            // others one each, spread over 40 k counters.
            uint2 kr = make_uint2(kWfNoKey, 0u);
            if (live) kr.x = wf_ray_key(w, L.st, x, y, L.org, L.dir);
            if (m_live) {
                const int leader = __ffsll((unsigned long long)m_live) - 1;
                const unsigned k0 = (unsigned)__shfl((int)kr.x, leader, 64);
                const unsigned long long same = __ballot(live && kr.x == k0);
                unsigned base = 0;
                if ((int)lane == leader) base = atomicAdd(&w.hist[k0], (unsigned)__popcll(same));
                base = (unsigned)__shfl((int)base, leader, 64);
                if (live && kr.x == k0) kr.y = base + (unsigned)__popcll(same & ((1ull << lane) - 1ull));
                else if (live) kr.y = atomicAdd(&w.hist[kr.x], 1u);
            }
            w.keys[slot] = kr;
        }
        // Per-generation totals without same-address atomics (32 k waves adding to one counter line cost 0.7 ms per
        // generation): wave sums -> block sums in LDS -> one record per block, which the scan kernel adds up.
        const unsigned r = wave_sum_u32(lc.rays), bn = wave_sum_u32(lc.bounces), nf = wave_sum_u32(lc.nonfinite);
        if (lane == 0) { unsigned *c = s_cnt[tid >> 6]; c[0] = (unsigned)__popcll(m_again); c[1] = r; c[2] = bn; c[3] = nf; c[4] = n_live; }
        __syncthreads();
        if (tid == 0) {
            unsigned t[5] = {0, 0, 0, 0, 0};
            for (int k = 0; k < kBlock / 64; k++) for (int j = 0; j < 5; j++) t[j] += s_cnt[k][j];
            w.block_counts[blockIdx.x] = make_uint4(t[0], t[1], t[2], t[3]);
            // unsorted queue: the block's place in it (one returning atomic per block)
            if (w.sort == WF_SORT_NONE && t[4]) s_pos = atomicAdd(&w.counters[w.gen], t[4]);
        }
        if (w.sort == WF_SORT_NONE) {
            __syncthreads();
            unsigned before = 0;
            for (int k = 0; k < (tid >> 6); k++) before += s_cnt[k][4];
            if (live) w.live[s_pos + before + (unsigned)__popcll(m_live & ((1ull << lane) - 1ull))] = (unsigned)slot;
        }
    }
}

// ---- the trace kernel ------------------------------------------------------------------------------------------
constexpr int kWfTraceBlock = 256;
constexpr int kWfTraceWaves = 8;          // waves per SIMD the trace kernel is built for (<= 64 VGPRs)
constexpr int kWfLdsLevels = 16;          // stack levels per lane in LDS (8 waves/SIMD x 16 levels = 128 KB per CU); deeper ones in HBM

struct WfTrace {
    const DevBvh4Node *nodes4;
    const DevBvh4QNode *nodes4q;    // GDPT_HBM_Q4 builds: the walk reads these instead
    const DevPrim *prims;
    const DevSphere *spheres;
    const float4 *rays;
    float4 *hits;
    const unsigned *live;
    const unsigned *count;          // rays in the queue (device: the step or scan kernel of this generation wrote it)
    int *ovf;                       // stack levels past kWfLdsLevels: [level - kWfLdsLevels][lane of the grid]
    gdpt::RenderCounters *counters;
    unsigned ovf_stride;            // lanes the overflow area is laid out for (>= lanes of the grid)
    int num_tris, num_nodes4, num_spheres, search_frac, count_stats;
};

#ifdef GDPT_BUILD_WF_TRACE   // emitted by render_wavefront_lambert.hip only (non-template kernels)

// counting sort, part 2: exclusive scan of the histogram (one block, 40 bins per thread), which is cleared for the next
// generation on the way; the total is the generation's ray count
__global__ __launch_bounds__(kWfScanBlock) void gdpt_wf_scan(WfBuf w) {
    __shared__ unsigned s_sum[kWfScanBlock];
    constexpr int kPer = (int)(kWfBins / kWfScanBlock);
    const int tid = threadIdx.x;
    {   // the step kernel's per-block counts of this generation -> the render's counters and the host's "active slots" word
        __shared__ unsigned long long s_tot[kWfScanBlock / 64][4];
        unsigned long long c[4] = {0, 0, 0, 0};
        for (int b = tid; b < w.n / kBlock; b += kWfScanBlock) { const uint4 v = w.block_counts[b]; c[0] += v.x; c[1] += v.y; c[2] += v.z; c[3] += v.w; }
#pragma unroll
        for (int j = 0; j < 4; j++) c[j] = wave_sum_u64(c[j]);
        if ((tid & 63) == 0) for (int j = 0; j < 4; j++) s_tot[tid >> 6][j] = c[j];
        __syncthreads();
        if (tid == 0) {
            unsigned long long t[4] = {0, 0, 0, 0};
            for (int k = 0; k < kWfScanBlock / 64; k++) for (int j = 0; j < 4; j++) t[j] += s_tot[k][j];
            w.counters[2 * kWfMaxGen + w.gen] = (unsigned)t[0];
            if (t[1]) atomicAdd(&w.totals->rays, t[1]);
            if (t[2]) atomicAdd(&w.totals->bounces, t[2]);
            if (t[3]) atomicAdd(&w.totals->nonfinite, t[3]);
        }
        if (w.sort == WF_SORT_NONE) return;       // (block-uniform) the queue is already in place, in slot order
        __syncthreads();
    }
    unsigned v[kPer], total = 0;
#pragma unroll
    for (int k = 0; k < kPer; k++) { v[k] = w.hist[tid * kPer + k]; total += v[k]; }
    s_sum[tid] = total;
    __syncthreads();
    for (int o = 1; o < kWfScanBlock; o <<= 1) {                // Hillis-Steele over the 1024 thread totals
        const unsigned add = (tid >= o) ? s_sum[tid - o] : 0u;
        __syncthreads();
        s_sum[tid] += add;
        __syncthreads();
    }
    unsigned run = s_sum[tid] - total;
#pragma unroll
    for (int k = 0; k < kPer; k++) { w.offsets[tid * kPer + k] = run; run += v[k]; if (v[k]) w.hist[tid * kPer + k] = 0u; }
    if (tid == kWfScanBlock - 1) w.counters[w.gen] = s_sum[tid];
}
// counting sort, part 3: every slot that emitted a ray writes its index to its place in the queue
__global__ __launch_bounds__(kBlock) void gdpt_wf_scatter(WfBuf w) {
    const long long slot = (long long)blockIdx.x * kBlock + threadIdx.x;
    const uint2 kr = w.keys[slot];
    if (kr.x != kWfNoKey) w.live[w.offsets[kr.x] + kr.y] = (unsigned)slot;
}

// Closest hit of every ray in the queue. A wave takes 64 consecutive queue entries (neighbours in key order) and walks
// them to completion; the other seven waves of the SIMD cover its fetches.
// The walk is the while-while order of trav_run with one node or one triangle in registers at a time: box arithmetic =
// visit_wide (conservative by the host's padding, box_hit), triangle arithmetic = tri_hit / test_tri_flat, spheres =
// sphere_hit — the same accept / reject decisions and the same (t, u, v, id) as every other kernel of the library.
// Spheres (an emitter or two per scene) are tested BEFORE the walk, every sphere against every ray, and skipped inside it:
// the closest hit does not depend on the order of the tests, the fp64 quadratic stays out of the loop's register budget,
// and a hit found first shortens the walk. A ray whose line provably passes the sphere at a distance is not tested at
// all (fp32 pre-test with a margin of 170 ulp of |o - c|^2 + r^2; the fp64 test decides everything else).
// The lane's stack is ONE register: `spa` = byte address of the next free slot in the block's LDS image
// ([level][thread], 4 bytes each), so level = spa >> 10 and the stack is empty while spa < 1024.
template <bool SPHERES, bool COUNT>
__global__ __launch_bounds__(kWfTraceBlock, COUNT ? 4 : kWfTraceWaves) void gdpt_wf_trace(WfTrace t) {
    extern __shared__ int s_wf_stack[];                       // [kWfLdsLevels][kWfTraceBlock]
    static_assert(kWfTraceBlock * 4 == 1024, "spa >> 10 = stack level");
    const unsigned count = *t.count;
    const unsigned wave_base = __builtin_amdgcn_readfirstlane(blockIdx.x * (unsigned)kWfTraceBlock + (threadIdx.x & ~63u));
    const unsigned total = gridDim.x * (unsigned)kWfTraceBlock;
    unsigned n_nodes = 0, n_prims = 0, n_node_trips = 0, n_leaf_trips = 0;
    auto lds = [&](unsigned byte_addr) -> int & { return *(int *)((char *)s_wf_stack + byte_addr); };
    auto ovf = [&](unsigned spa_) -> int & {                  // levels past the LDS image: HBM, [level - kWfLdsLevels][lane of the grid]
        return t.ovf[(size_t)((spa_ >> 10) - (unsigned)kWfLdsLevels) * t.ovf_stride + blockIdx.x * (unsigned)kWfTraceBlock + threadIdx.x];
    };
    for (unsigned base = wave_base; base < count; base += total) {        // wave-uniform; normally one trip
        const unsigned idx = base + (threadIdx.x & 63u);
        float o[3] = {0, 0, 0}, d[3] = {0, 0, 1}, tnear = 0;
        int cur = kTravDone;
        unsigned spa = threadIdx.x * 4u;
        if (idx < count) {
            const unsigned slot = t.live[idx];
            const float4 r0 = t.rays[2 * (size_t)slot], r1 = t.rays[2 * (size_t)slot + 1];
            o[0] = r0.x; o[1] = r0.y; o[2] = r0.z; tnear = r0.w; d[0] = r1.x; d[1] = r1.y; d[2] = r1.z;
            cur = t.num_nodes4 ? 0 : kTravDone;
        }
        Hit best;
        best.gid = -1; best.t = __builtin_huge_valf(); best.u = best.v = 0; best.ngx = best.ngy = best.ngz = 0;
        if (SPHERES && idx < count) {
            for (int s = 0; s < t.num_spheres; s++) {
                const DevSphere &sph = t.spheres[s];
                const float cx = (float)sph.center[0], cy = (float)sph.center[1], cz = (float)sph.center[2], r = (float)sph.radius;
                const float lx = o[0] - cx, ly = o[1] - cy, lz = o[2] - cz;
                const float dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], ll = lx * lx + ly * ly + lz * lz, ld_ = lx * d[0] + ly * d[1] + lz * d[2];
                // squared distance of the ray's line from the centre = ll - ld^2 / dd; far outside the sphere: no test
                if (ll * dd - ld_ * ld_ > (r * r + 1e-5f * (ll + r * r)) * dd) continue;
                const SphereHit sh = sphere_hit(o[0], o[1], o[2], d[0], d[1], d[2], tnear, __builtin_huge_valf(), sph.center[0], sph.center[1], sph.center[2], sph.radius);
                const int sg = t.num_tris + s;
                if (sh.ok && (best.gid < 0 || sh.t < best.t || (sh.t == best.t && sg < best.gid))) {
                    best.t = sh.t; best.u = sh.u; best.v = sh.v; best.gid = sg;
                    // the sphere's normal goes to the hit record at once (read only if a sphere is still the closest hit at
                    // the end): three registers less across the walk
                    t.hits[2 * (size_t)t.live[idx] + 1] = make_float4(sh.ngx, sh.ngy, sh.ngz, 0.0f);
                }
            }
        }
        float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
        float oi[3] = {o[0] * inv[0], o[1] * inv[1], o[2] * inv[2]};
#pragma unroll
        for (int k = 0; k < 3; k++) if (d[k] == 0.0f) { inv[k] = __builtin_nanf(""); oi[k] = __builtin_nanf(""); }     // trav_run: the slab drops out
        auto pop = [&]() {
            if (spa >= 1024u) { spa -= 1024u; cur = ((spa >> 10) < (unsigned)kWfLdsLevels) ? lds(spa) : ovf(spa); }
            else cur = kTravDone;
        };
        for (;;) {
            const int live = __popcll(__ballot(cur != kTravDone));
            if (live == 0) break;
            // inner nodes until at most search_frac/256 of the live lanes are still looking for their next leaf
            const int few = (live * t.search_frac) >> 8;
            for (;;) {
                const bool searching = cur >= 0;
                if (__popcll(__ballot(searching)) <= few) break;
                if (searching) {
                    if (COUNT) { n_nodes++; if (wave_leader()) n_node_trips++; }
                    WideVisit wv;
                    // tfar of the boxes = the best hit so far (a sphere's, before the first triangle)
                    if (GDPT_HBM_Q4) visit_wide_q4(t.nodes4q[cur], oi, inv, tnear, best.t, wv);
                    else visit_wide<false>(t.nodes4[cur], oi, inv, tnear, best.t, wv);
                    if (wv.key[0] != kMissKey) {
                        if ((spa >> 10) + 3u <= (unsigned)kWfLdsLevels) {           // (the usual case: all three slots in LDS)
                            if (wv.key[3] != kMissKey) { lds(spa) = wv.ch[3]; spa += 1024u; }
                            if (wv.key[2] != kMissKey) { lds(spa) = wv.ch[2]; spa += 1024u; }
                            if (wv.key[1] != kMissKey) { lds(spa) = wv.ch[1]; spa += 1024u; }
                        } else {
#pragma unroll
                            for (int i = 3; i >= 1; i--)
                                if (wv.key[i] != kMissKey) {
                                    if ((spa >> 10) < (unsigned)kWfLdsLevels) lds(spa) = wv.ch[i]; else ovf(spa) = wv.ch[i];
                                    spa += 1024u;
                                }
                        }
                        cur = wv.ch[0];
                    } else pop();
                }
            }
            if (cur < 0 && cur != kTravDone) {
                const unsigned packed = ~(unsigned)cur;
                const unsigned first = packed >> 2, n = (packed & 3u) + 1u;
                if (COUNT) { n_prims += n; if (wave_leader()) n_leaf_trips++; }
                for (unsigned i = 0; i < n; i++) {
                    const DevPrim pr = t.prims[first + i];                  // one 48-byte record in registers at a time
                    test_tri_flat(pr, o, d, tnear, __builtin_huge_valf(), !(SPHERES && (pr.gid & GDPT_SPHERE_FLAG)), best);
                }
                pop();
            }
        }
        if (idx < count) {
            const unsigned slot = t.live[idx];
            t.hits[2 * (size_t)slot] = make_float4(best.t, best.u, best.v, __uint_as_float((unsigned)best.gid));
        }
    }
    if (COUNT) {
        const unsigned long long nn = wave_sum_u64(n_nodes), np = wave_sum_u64(n_prims), t0 = wave_sum_u64(n_node_trips), t1 = wave_sum_u64(n_leaf_trips);
        if ((threadIdx.x & 63) == 0 && (nn | np)) {
            atomicAdd(&t.counters->nodes, nn); atomicAdd(&t.counters->prims, np);
            atomicAdd(&t.counters->wave_node_trips, t0); atomicAdd(&t.counters->wave_leaf_trips, t1);
        }
    }
}
#endif // GDPT_BUILD_WF_TRACE

} // namespace gd

namespace gdpt {
void launch_wf_step_lambert(const DevSceneView &sv, const gd::KernelArgs &a, const gd::WfBuf &w, hipStream_t stream);
void launch_wf_step_general(const DevSceneView &sv, const gd::KernelArgs &a, const gd::WfBuf &w, hipStream_t stream);
void launch_wf_sort(const gd::WfBuf &w, hipStream_t stream);                       // scan + scatter
void launch_wf_trace(const gd::WfTrace &t, bool spheres, unsigned blocks, hipStream_t stream);
void launch_wf_init(const gd::WfBuf &w, hipStream_t stream);
} // namespace gdpt
