// render_wavefront.h — the GradPath tile loop as a wavefront pipeline, for scenes that live in HBM.
//
// The lane machine of render_device.h keeps a path's whole state in registers / LDS and therefore runs at 2 waves per
// SIMD. That is the right trade for a scene that fits LDS (cbox), but a scene walked from HBM (sponza: 19 BVH4 nodes
// and 7 triangles per ray out of L2 / Infinity Cache) then spends half of every wave's life in s_waitcnt with nothing
// to switch to. Here the same per-sample program (lane_step, unchanged: same arithmetic, same RNG streams, same
// per-item summation order, so the images are bit-identical to the lane machine's) is cut at the ray:
//
//   gdpt_wf_step   one thread per path slot: take a work item if idle, consume the slot's hit (rebuild the vertex,
//                  run the state arm, sample the BSDF or the camera), store the state, queue the slot's next ray.
//                  State lives in HBM between steps, SoA ([field][slot], 8-byte words), so loads and stores coalesce.
//   gdpt_wf_trace  persistent waves with NOTHING but a ray in registers (fp32 origin / direction, best hit, stack
//                  pointer): many waves per SIMD hide the node-fetch latency. A lane that finishes its ray writes the
//                  hit record and pulls the next ray index from the generation's queue (one wave-wide atomicAdd per
//                  refill, ballot + prefix popcount to hand indices out), so no lane idles behind the wave's longest
//                  walk; the walk itself is trav_run (while-while, BVH4, closest hit = min fp32 t / lowest id).
//
// One "generation" = step + trace. The queue of live slots is rebuilt by every step (ballot compaction of the slots
// that still hold a ray); generations are enqueued back to back and the host looks at the live count one chunk of
// generations behind the launches.
//
// Reference: the tile loop src/render.cpp:277-331 over grad_path_tracing src/path_tracing.h:354-1050, as render_device.h.
#pragma once
#include "render_device.h"

namespace gd {

// ---- state layout: 8-byte words, word w of slot s at state[w * N + s] ------------------------------------------
enum {
    WF_RNG = 0,                 // u64 PCG state (the increment follows from pixel + sample index)
    WF_ORG = 1, WF_DIR = 4,     // pending ray, fp64 (3 + 3)
    WF_F = 7, WF_PDF = 10,      // f*|cos| and pdf of the pending bounce ray
    WF_CONTRIB = 11, WF_THROUGHPUT = 14, WF_PROB = 17,
    WF_I0 = 18,                 // st | s << 32
    WF_I1 = 19,                 // s_end | num_vertices << 32
    WF_I2 = 20,                 // mats | kc << 32
    WF_XY = 21,                 // x | y << 32
    WF_ITEM = 22,               // item (u32, ~0 = none) | hit gid << 32
    WF_HIT0 = 23,               // hit t | u << 32 (float bits)
    WF_HIT1 = 24,               // hit v | ngx << 32
    WF_HIT2 = 25,               // hit ngy | ngz << 32   (sphere hits only)
    WF_PRIV = 26,               // LanePriv: 8 doubles
    WF_ACC = 34,                // 15 running sums of the slot's current item
    WF_WORDS = 49
};

struct WfBuf {
    unsigned long long *state;      // WF_WORDS * N
    unsigned *live;                 // N: slots holding a ray, written by step, read by trace
    unsigned *counters;             // [3][kWfMaxGen]: live count, trace queue head, active slots (live + just started)
    int n;                          // slots (multiple of 256)
    int gen;                        // generation of this launch
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
    tx.nodes = sv.nodes; tx.nodes4 = sv.nodes4; tx.nodes8 = sv.nodes8; tx.prims = sv.prims; tx.tris = sv.tris; tx.materials = sv.materials; tx.lights = sv.light_intensity;
    AccMem acc; acc.slot = (double *)(S + (long long)WF_ACC * N); acc.stride = N;
    LanePriv lp; lp.slot = (double *)(S + (long long)WF_PRIV * N); lp.stride = (int)N;

    Lane L;
    const unsigned long long i0 = ld(WF_I0);
    L.st = (int)(unsigned)i0; L.s = (int)(i0 >> 32);
    const unsigned long long itw = ld(WF_ITEM);
    long long my_item = ((unsigned)itw == 0xFFFFFFFFu) ? -1 : (long long)(unsigned)itw;
    unsigned long long i1 = 0, xy = 0;
    int x = 0, y = 0;

    // ---- idle slots publish their finished item and take the next one (one atomicAdd per wave)
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
        if (idle || fresh) { st_(WF_I0, pack2((unsigned)S_DONE, (unsigned)L.s)); st_(WF_ITEM, pack2(my_item < 0 ? 0xFFFFFFFFu : (unsigned)my_item, 0xFFFFFFFFu)); }
        if (fresh) st_(WF_XY, pack2((unsigned)x, (unsigned)y));
    } else {
        // ---- load the slot (a fresh slot has nothing to load)
        Trav tv;
        tv.cur = kTravDone; tv.sp = 0;
        tv.best.gid = -1; tv.best.t = 0; tv.best.u = tv.best.v = 0; tv.best.ngx = tv.best.ngy = tv.best.ngz = 0;
        if (!fresh) {
            i1 = ld(WF_I1); L.s_end = (int)(unsigned)i1; L.num_vertices = (int)(i1 >> 32);
            const unsigned long long i2 = ld(WF_I2); L.mats = (int)(unsigned)i2; L.kc = (int)(i2 >> 32);
            xy = ld(WF_XY); x = (int)(unsigned)xy; y = (int)(xy >> 32);
            L.rng_state = ld(WF_RNG);
            L.org = mk(ldd(WF_ORG), ldd(WF_ORG + 1), ldd(WF_ORG + 2)); L.dir = mk(ldd(WF_DIR), ldd(WF_DIR + 1), ldd(WF_DIR + 2));
            L.f = mk(ldd(WF_F), ldd(WF_F + 1), ldd(WF_F + 2)); L.pdf = ldd(WF_PDF);
            L.contrib = mk(ldd(WF_CONTRIB), ldd(WF_CONTRIB + 1), ldd(WF_CONTRIB + 2));
            L.throughput = mk(ldd(WF_THROUGHPUT), ldd(WF_THROUGHPUT + 1), ldd(WF_THROUGHPUT + 2));
            L.prob = ldd(WF_PROB);
            tv.best.gid = (int)(itw >> 32);
            const unsigned long long h0 = ld(WF_HIT0), h1 = ld(WF_HIT1);
            tv.best.t = __uint_as_float((unsigned)h0); tv.best.u = __uint_as_float((unsigned)(h0 >> 32));
            tv.best.v = __uint_as_float((unsigned)h1); tv.best.ngx = __uint_as_float((unsigned)(h1 >> 32));
            if (tv.best.gid >= sv.num_tris) { const unsigned long long h2 = ld(WF_HIT2); tv.best.ngy = __uint_as_float((unsigned)h2); tv.best.ngz = __uint_as_float((unsigned)(h2 >> 32)); }
        }
        const unsigned long long base = ((unsigned long long)y * W + x) * (unsigned long long)a.spp;
        L.rng_inc = ((base + (unsigned long long)L.s) << 1u) | 1u;       // pcg_init's increment of the sample in flight

        lane_step<LAMBERT, false>(sv, tx, a.max_depth, spp, x, y, base, L, tv, lp, acc, lc, tc);

        // ---- store, and queue the slot's next ray
        st_(WF_I0, pack2((unsigned)L.st, (unsigned)L.s));
        st_(WF_I1, pack2((unsigned)L.s_end, (unsigned)L.num_vertices));
        st_(WF_I2, pack2((unsigned)L.mats, (unsigned)L.kc));
        if (fresh) st_(WF_XY, pack2((unsigned)x, (unsigned)y));
        st_(WF_ITEM, pack2(my_item < 0 ? 0xFFFFFFFFu : (unsigned)my_item, 0xFFFFFFFFu));     // hit gid: none until traced
        if (L.st != S_DONE) {
            st_(WF_RNG, L.rng_state);
            std_(WF_ORG, L.org.x); std_(WF_ORG + 1, L.org.y); std_(WF_ORG + 2, L.org.z);
            std_(WF_DIR, L.dir.x); std_(WF_DIR + 1, L.dir.y); std_(WF_DIR + 2, L.dir.z);
            if (L.st == S_BOUNCE) { std_(WF_F, L.f.x); std_(WF_F + 1, L.f.y); std_(WF_F + 2, L.f.z); std_(WF_PDF, L.pdf); }
            std_(WF_CONTRIB, L.contrib.x); std_(WF_CONTRIB + 1, L.contrib.y); std_(WF_CONTRIB + 2, L.contrib.z);
            std_(WF_THROUGHPUT, L.throughput.x); std_(WF_THROUGHPUT + 1, L.throughput.y); std_(WF_THROUGHPUT + 2, L.throughput.z);
            std_(WF_PROB, L.prob);
        }
    }
    {
        const bool live = work && (L.st != S_DONE);          // holds a ray: goes to the trace queue
        // must be stepped again: every slot that did something (a finished item still has to be published), and a fresh
        // but empty item (a slot of a ragged edge tile: published as zeros by the next step)
        const bool again = work || (fresh && my_item >= 0);
        const unsigned long long m_live = __ballot(live), m_again = __ballot(again);
        const unsigned lane = (unsigned)(tid & 63);
        if (m_live) {
            unsigned pos = 0;
            if (lane == 0) pos = atomicAdd(&w.counters[w.gen], (unsigned)__popcll(m_live));
            pos = __shfl(pos, 0, 64);
            if (live) w.live[pos + __popcll(m_live & ((1ull << lane) - 1ull))] = (unsigned)slot;
        }
        if (m_again && lane == 0) atomicAdd(&w.counters[2 * kWfMaxGen + w.gen], (unsigned)__popcll(m_again));
        // counters of this step (rays are counted where they are consumed, as in the lane machine)
        const unsigned r = wave_sum_u32(lc.rays), bn = wave_sum_u32(lc.bounces), nf = wave_sum_u32(lc.nonfinite);
        if (lane == 0) {
            if (r) atomicAdd(&a.counters->rays, (unsigned long long)r);
            if (bn) atomicAdd(&a.counters->bounces, (unsigned long long)bn);
            if (nf) atomicAdd(&a.counters->nonfinite, (unsigned long long)nf);
        }
    }
}

constexpr int kWfTraceWaves = 3;          // waves per SIMD the trace kernel is built for (LDS: 32-slot stacks)
#ifdef GDPT_BUILD_WF_TRACE   // emitted by render_wavefront_lambert.hip only (non-template kernel)
// Traces the generation's rays. Persistent waves; a lane holds one ray (fp32) and nothing else.
__global__ __launch_bounds__(kBlock, kWfTraceWaves) void gdpt_wf_trace(DevSceneView sv, KernelArgs a, WfBuf w) {
    __shared__ int s_stack[GDPT_BVH_MAX_DEPTH * kBlock];
    const int tid = threadIdx.x;
    const long long N = w.n;
    const unsigned count = w.counters[w.gen];
    if (count == 0) return;
    unsigned *head = &w.counters[kWfMaxGen + w.gen];
    TraceCtx tx;
    tx.count = a.count != 0; tx.need_uv = false;
    tx.stack = s_stack + tid; tx.stride = kBlock;
    tx.nodes = sv.nodes; tx.nodes4 = sv.nodes4; tx.nodes8 = sv.nodes8; tx.prims = sv.prims; tx.tris = sv.tris; tx.materials = sv.materials; tx.lights = sv.light_intensity;
    TraceCounters tc = {0, 0, 0, 0, 0, 0};
    LaneCounters lc = {0, 0, 0};
    Trav tv;
    tv.cur = kTravDone; tv.sp = 0; tv.best.gid = -1; tv.best.t = 0; tv.best.u = tv.best.v = tv.best.ngx = tv.best.ngy = tv.best.ngz = 0;
    long long my = -1;                     // slot whose ray this lane walks
    D3 org = splat(0), dir = splat(0);
    float tnear = 0.0f;
    bool dry = false;                      // the generation's queue has been handed out completely
    for (;;) {
        if (my >= 0 && tv.cur == kTravDone) {              // ray finished: hit record, lane is free
            unsigned long long *S = w.state + my;
            unsigned long long itw = S[(long long)WF_ITEM * N];
            S[(long long)WF_ITEM * N] = pack2((unsigned)itw, (unsigned)tv.best.gid);
            if (tv.best.gid >= 0) {
                S[(long long)WF_HIT0 * N] = pack2(__float_as_uint(tv.best.t), __float_as_uint(tv.best.u));
                S[(long long)WF_HIT1 * N] = pack2(__float_as_uint(tv.best.v), __float_as_uint(tv.best.ngx));
                if (tv.best.gid >= sv.num_tris) S[(long long)WF_HIT2 * N] = pack2(__float_as_uint(tv.best.ngy), __float_as_uint(tv.best.ngz));
            }
            my = -1;
        }
        const bool idle = (my < 0);
        const unsigned long long m_idle = __ballot(idle);
        if (m_idle && !dry) {
            unsigned got = 0;
            const int leader = __ffsll((unsigned long long)m_idle) - 1;
            if ((tid & 63) == leader) got = atomicAdd(head, (unsigned)__popcll(m_idle));
            got = __shfl(got, leader, 64);
            const unsigned mine = got + (unsigned)__popcll(m_idle & ((1ull << (tid & 63)) - 1ull));
            if (got + (unsigned)__popcll(m_idle) >= count) dry = true;
            if (idle && mine < count) {
                my = (long long)w.live[mine];
                const unsigned long long *S = w.state + my;
                org = mk(__longlong_as_double((long long)S[(long long)WF_ORG * N]), __longlong_as_double((long long)S[(long long)(WF_ORG + 1) * N]), __longlong_as_double((long long)S[(long long)(WF_ORG + 2) * N]));
                dir = mk(__longlong_as_double((long long)S[(long long)WF_DIR * N]), __longlong_as_double((long long)S[(long long)(WF_DIR + 1) * N]), __longlong_as_double((long long)S[(long long)(WF_DIR + 2) * N]));
                tnear = ((int)(unsigned)S[(long long)WF_I0 * N] == S_BOUNCE) ? (float)sv.isect_eps : 0.0f;
                trav_init(sv, tv, __builtin_huge_val());
            }
        }
        const bool walking = (my >= 0) && tv.cur != kTravDone;
        const unsigned long long m_walk = __ballot(walking);
        if (m_walk == 0ull) { if (dry && !__any(my >= 0)) break; else continue; }
        // leave the walk when a quarter of the rays that entered it is still unfinished (or nothing is left to refill with)
        const int stop_below = dry ? 0 : ((__popcll(m_walk) * a.thresh_a) >> 8);
        if (walking) trav_run<TraceHbm>(sv, tx, org, dir, tnear, __builtin_huge_valf(), tv, stop_below, a.thresh_c, tc);
    }
    flush_counters(a, lc, tc, a.count != 0);
}
#endif // GDPT_BUILD_WF_TRACE

} // namespace gd

namespace gdpt {
void launch_wf_step_lambert(const DevSceneView &sv, const gd::KernelArgs &a, const gd::WfBuf &w, hipStream_t stream);
void launch_wf_step_general(const DevSceneView &sv, const gd::KernelArgs &a, const gd::WfBuf &w, hipStream_t stream);
void launch_wf_trace(const DevSceneView &sv, const gd::KernelArgs &a, const gd::WfBuf &w, unsigned blocks, hipStream_t stream);
void launch_wf_init(const gd::WfBuf &w, hipStream_t stream);
} // namespace gdpt
