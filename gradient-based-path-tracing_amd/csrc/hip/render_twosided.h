// render_twosided.h — the GradPath lane machine for scenes with two-sided lobes (DisneyGlass, DisneyBSDF).
//
// For one-sided lobes an offset that survives bounce 1 is retired by bounce 2, which is what lets render_device.h
// evaluate offsets lazily from two special cases. Two-sided lobes let an offset live for many bounces. But under the
// A-semantics (an offset keeps its PRIMARY vertex for the whole path) an offset never traces anything
// after its primary ray: per bounce it only needs (a) the material id of the base path's current vertex (the test at
// src/path_tracing.h:607-612), (b) the base path's p2 of that bounce (jacob *= p2 / p2_offset, :813) and (c) the
// bounce's three random numbers, which sit at a known position of the sample's PCG stream. So the base path runs first
// and logs 16 bytes per bounce iteration into a per-lane log in HBM; when it ends, the four offsets are replayed from
// the log: one primary ray each, then one BSDF re-sampling at the same primary vertex per logged iteration — through
// the same shared BSDF block the base path's bounces use. Exact for every material (no one-sidedness argument), with
// the lane machine's work queue, resumable trace phase and occupancy.
//
// Log capacity: kLogCap iterations per sample. A path longer than that (probability < 0.95^1000 with Russian
// roulette) stops being logged; an offset still alive at that point keeps the jacobian it has.
#pragma once
#include "render_device.h"

namespace gd {

constexpr int S_REPLAY = 5;            // after S_DONE = 4: an offset re-sampling its primary vertex, no pending ray
constexpr int kLogCap = 1024;
constexpr int kReplayPerStep = 4;       // replay iterations of an offset per wave step (KernelArgs::replay_per_step; test knob)
struct BounceLog { double p2; int mat; int pad; };   // p2 < 0: the iteration broke at pdf <= 0 (before any update); log[it * kBlock] = iteration `it` of this lane

// LDS slot of a lane (doubles, stride kBlock): 0..2 radiance, 3 eta_scale, 4..7 filter cache
struct LanePriv2 {
    double *slot; int stride;
    GD D3 radiance() const { return mk(slot[0], slot[stride], slot[2 * stride]); }
    GD void set_radiance(D3 v) { slot[0] = v.x; slot[stride] = v.y; slot[2 * stride] = v.z; }
    GD double eta_scale() const { return slot[3 * stride]; }
    GD void set_eta_scale(double v) { slot[3 * stride] = v; }
    GD FilterCache fc() const { FilterCache f; f.dx = slot[4 * stride]; f.dy = slot[5 * stride]; f.ox = slot[6 * stride]; f.oy = slot[7 * stride]; return f; }
    GD void set_fc(const FilterCache &f) { slot[4 * stride] = f.dx; slot[5 * stride] = f.dy; slot[6 * stride] = f.ox; slot[7 * stride] = f.oy; }
};

// Lane fields are reused: mats = mat0 | n_iter << 12 (iterations logged); kc = offset k | replay index << 2;
// during the offset phase f = the offset's current direction, pdf = its jacobian, rng_state = the replay stream.
GD int n_iter(const Lane &L) { return (int)((unsigned)L.mats >> 12); }
GD void set_n_iter(Lane &L, int n) { L.mats = (L.mats & 0xFFF) | (n << 12); }

// Material set the kernel can be built for (bit t = material type t, include/gdpt.h): the test scenes pair ONE Disney lobe
// with Lambertian walls, and a kernel that need not carry DisneyBSDF's five inlined lobes is a much smaller register
// allocation problem (42 instead of 318 spilled VGPRs; +5 % on the glass scene).
constexpr unsigned kSetGlass = (1u << GDPT_MAT_LAMBERTIAN) | (1u << GDPT_MAT_DISNEY_GLASS);
template <unsigned MASK, class ACC>
GD void lane_step2(const DevSceneView &sv, const TraceCtx &tx, int max_depth, double spp, int x, int y, unsigned long long base,
                   Lane &L, Trav &tv, LanePriv2 &lp, ACC &acc, LaneCounters &lc, BounceLog *log, int replay_per_step) {
    const DevCamera &cam = sv.cam;
    const int w = cam.width, h = cam.height;
    const int st0 = L.st;
    int act = ACT_NONE;
    Vertex nv;
    Ray ray;
    ray.org = L.org; ray.dir = L.dir; ray.tfar = __builtin_huge_val();
    ray.tnear = (st0 == S_BOUNCE) ? sv.isect_eps : 0.0;
    const bool tracing = (st0 == S_PRIMARY || st0 == S_BOUNCE || st0 == S_OFFSET);
    bool hit = false;
    if (tracing) { lc.rays++; hit = tv.best.gid >= 0; }
    if (st0 == S_REPLAY) hit = true;                       // the offset's primary hit is still in tv
    if (hit) make_vertex(sv, tx.tris, tx.need_uv, ray, tv.best, 0.0, (st0 == S_BOUNCE) ? 0.0 : 0.25 / (double)max(w, h), nv);
    // ---------------- consume the hit ----------------
    if (st0 == S_START) {
        act = ACT_PRIMARY_RAY;
    } else if (st0 == S_PRIMARY) {
        if (!hit) act = ACT_NEXT_SAMPLE;                                            // :375-379
        else {
            L.mats = nv.material_id & 0xFFF;                                        // mat0, no iteration logged yet
            L.contrib = splat(1.0); L.throughput = splat(1.0);
            L.prob = 1.0;
            D3 rad0 = splat(0);
            if (nv.light_id >= 0) { D3 Le = emission(sv, nv, -ray.dir); rad0 = Le; L.contrib = Le; }   // :490-493
            lp.set_radiance(rad0); lp.set_eta_scale(1.0);
            L.num_vertices = 3;
            act = loop_allows(max_depth, 3) ? ACT_BOUNCE : ACT_OFFSETS;
        }
    } else if (st0 == S_BOUNCE) {
        const int it = L.num_vertices - 3;
        double G = 1.0;
        if (hit) { D3 dl = nv.position - ray.org; G = fabs(dot(ray.dir, nv.gn)) / dot(dl, dl); }   // :746-753
        const D3 f = L.f;
        const double p2 = L.pdf * G;                                                // :766
        if (it < kLogCap) log[it * kBlock].p2 = p2;
        L.contrib = L.contrib * f * G; L.prob *= p2;                                // :769-770
        if (hit && nv.light_id >= 0) {                                              // :971-980
            D3 Le = emission(sv, nv, -ray.dir);
            D3 C2 = (G * f) * Le;
            L.contrib = L.contrib * Le;
            lp.set_radiance(lp.radiance() + L.throughput * (C2 / p2));
        }
        bool stop = !hit;                                                           // :982-985
        if (!stop) {
            double rr_prob = 1;
            if (L.num_vertices - 1 >= sv.rr_depth) {                                // :992-999
                rr_prob = fmin(maxc((1 / lp.eta_scale()) * L.throughput), 0.95);
                Pcg rr_rng; rr_rng.state = L.rng_state; rr_rng.inc = L.rng_inc;
                double u = pcg_real(rr_rng); L.rng_state = rr_rng.state;
                if (u > rr_prob) stop = true;
            }
            if (!stop) {
                L.throughput = L.throughput * (G * f) / (p2 * rr_prob);             // :1003
                L.num_vertices++;
                if (!loop_allows(max_depth, L.num_vertices)) stop = true;
            }
        }
        act = stop ? ACT_OFFSETS : ACT_BOUNCE;
    }
    // ---------------- offsets: validity of the primary hit, then replay of the logged iterations ----------------
    // The replay runs as a loop inside this step (at most kReplayPerStep iterations, the rest continues as S_REPLAY in the
    // next step): an iteration needs only sample + pdf of the offset's primary vertex, a fraction of a wave step, and a
    // lane that replays holds no pending ray, so every step it spends in S_REPLAY is a trace phase it sits out.
    bool off_done = false, off_alive = false;
    if (st0 == S_OFFSET) {
        if (hit && nv.material_id == L.mat0()) {                                    // :424-443
            L.f = L.dir; L.pdf = 1.0;                                               // o.dir, o.jacob
            L.kc &= 3;                                                              // replay index 0
            Pcg r2 = pcg_init(base + (unsigned long long)L.s);
            (void)pcg_next(r2); (void)pcg_next(r2);                                 // the sample's two sub-pixel numbers
            L.rng_state = r2.state; L.rng_inc = r2.inc;
            off_alive = true;
        } else off_done = true;                                                     // dead: contribX = 0, w = 1
    }
    if (st0 == S_REPLAY) off_alive = true;
    // The replay of the logged iterations and the base paths' BSDF block. SHARE: the FIRST re-sampling of a step runs through the
    // shared block, in the same instructions as the base paths' bounces (a lane is either replaying or bouncing, and it is the same
    // lobe code on the same kind of inputs: material of nv, a view direction, three numbers; the block's fused eval + pdf returns
    // bsdf_pdf's bits); further iterations, up to the step's budget, follow in a loop of their own. Without it every iteration runs in
    // that loop, and a wave executes the lobe code once per population and step. Same-box A/B
    // (profiles/r03_ab_shared_bsdf_block.txt): the kernel with the full two-sided switch +6 % on DisneyBSDF (its offsets rarely outlive
    // one iteration, so the loop is nearly emptied); the kernel built for {Lambertian, DisneyGlass} -10 % (43 -> 127 spilled VGPRs):
    // that one keeps the separate loop. The two forms are written out side by side: expressed through shared helpers they compile
    // to twice the spills of either (577 / 81 spilled VGPRs against 278 / 43).
    constexpr bool SHARE = (MASK != kSetGlass);
    bool sampled = false;
    BsdfSample bs; bs.dir_out = splat(0); bs.eta = 0; bs.roughness = 0;
    D3 f = splat(0);
    double pdf = 0;
    if (SHARE) {
        const bool replaying = off_alive;
        int r = L.kc >> 2, budget = replay_per_step;
        const int n_it = n_iter(L);
        Pcg rs; rs.state = L.rng_state; rs.inc = L.rng_inc;
        D3 odir = L.f;
        double jac = L.pdf;
        BounceLog e; e.p2 = 0; e.mat = 0; e.pad = 0;
        D2 ruv; ruv.x = ruv.y = 0;
        double rw = 0;
        // head of a replay iteration: may this lane re-sample now? (draws the base path's numbers of iteration r if so)
        auto replay_head = [&]() __attribute__((always_inline)) -> bool {
            if (r >= n_it || r >= kLogCap) { off_done = true; off_alive = (r >= n_it); return false; }      // all iterations replayed
            e = log[r * kBlock];
            if (nv.material_id != e.mat) { off_done = true; off_alive = false; return false; }             // :607-612
            if (e.p2 < 0) { off_done = true; return false; }                         // the base broke at pdf <= 0: no re-sampling
            if (budget-- == 0) { L.st = S_REPLAY; return false; }
            ruv.x = pcg_real(rs); ruv.y = pcg_real(rs); rw = pcg_real(rs);            // the base path's numbers of this iteration
            if (r + 2 >= sv.rr_depth) (void)pcg_next(rs);                             // skip its roulette draw
            return true;
        };
        const bool replay_now = replaying ? replay_head() : false;
        // ---------------- shared BSDF block: base-path bounces and the first replay iteration of the step ----------------
        if (act == ACT_BOUNCE) {
            Pcg rb; rb.state = L.rng_state; rb.inc = L.rng_inc;
            ruv.x = pcg_real(rb); ruv.y = pcg_real(rb); rw = pcg_real(rb);              // :536-537
            L.rng_state = rb.state;
            lc.bounces++;
        }
        if (act == ACT_BOUNCE || replay_now) {
            const D3 dir_view = replay_now ? -odir : -ray.dir;
            sampled = mat_sample<false, false, true, MASK>(sv, tx, nv, dir_view, ruv, rw, bs);
            if (sampled) mat_eval_pdf<false, false, true, MASK>(sv, tx, nv, dir_view, bs.dir_out, f, pdf);
        }
        if (replay_now) {
            const GdptMaterial &om = tx.materials[nv.material_id];
            bool osampled = sampled;
            double opdf = sampled ? pdf : 0.0;
            D3 onext = bs.dir_out;
            for (;;) {
                if (!osampled || opdf <= 0.0) { off_done = true; off_alive = false; break; }             // :773-959
                jac *= e.p2 / opdf; odir = onext; r++;                                  // :813, :815-816
                if (!replay_head()) break;
                BsdfSample obs; obs.dir_out = splat(0); obs.eta = 0; obs.roughness = 0;
                const D3 oview = -odir;
                osampled = bsdf_sample<false, true, MASK>(sv, om, oview, nv, ruv, rw, obs);
                opdf = osampled ? bsdf_pdf<false, true, MASK>(sv, om, oview, obs.dir_out, nv) : 0.0;
                onext = obs.dir_out;
            }
        }
        if (replaying) {
            L.f = odir; L.pdf = jac; L.kc = (L.kc & 3) | (r << 2);
            L.rng_state = rs.state;
        }
    } else {
        if (off_alive) {
            int r = L.kc >> 2, budget = replay_per_step;
            const int n_it = n_iter(L);
            Pcg rs; rs.state = L.rng_state; rs.inc = L.rng_inc;
            D3 odir = L.f;
            double jac = L.pdf;
            const GdptMaterial &om = tx.materials[nv.material_id];
            for (;;) {
                if (r >= n_it || r >= kLogCap) { off_done = true; off_alive = (r >= n_it); break; }       // all iterations replayed
                const BounceLog e = log[r * kBlock];
                if (nv.material_id != e.mat) { off_done = true; off_alive = false; break; }              // :607-612
                if (e.p2 < 0) { off_done = true; break; }                               // the base broke at pdf <= 0: no re-sampling
                if (budget-- == 0) { L.st = S_REPLAY; break; }
                D2 ruv; double rw;
                ruv.x = pcg_real(rs); ruv.y = pcg_real(rs); rw = pcg_real(rs);          // the base path's numbers of this iteration
                if (r + 2 >= sv.rr_depth) (void)pcg_next(rs);                           // skip its roulette draw
                BsdfSample obs; obs.dir_out = splat(0); obs.eta = 0; obs.roughness = 0;
                const D3 oview = -odir;
                const bool osampled = bsdf_sample<false, true, MASK>(sv, om, oview, nv, ruv, rw, obs);
                const double opdf = osampled ? bsdf_pdf<false, true, MASK>(sv, om, oview, obs.dir_out, nv) : 0.0;
                if (!osampled || opdf <= 0.0) { off_done = true; off_alive = false; break; }             // :773-959
                jac *= e.p2 / opdf; odir = obs.dir_out; r++;                            // :813, :815-816
            }
            L.f = odir; L.pdf = jac; L.kc = (L.kc & 3) | (r << 2);
            L.rng_state = rs.state;
        }
        // ---------------- shared BSDF block (base path) ----------------
        if (act == ACT_BOUNCE) {
            D2 ruv; double rw;
            Pcg r; r.state = L.rng_state; r.inc = L.rng_inc;
            ruv.x = pcg_real(r); ruv.y = pcg_real(r); rw = pcg_real(r);                 // :536-537
            L.rng_state = r.state;
            lc.bounces++;
            const D3 dir_view = -ray.dir;
            sampled = mat_sample<false, false, true, MASK>(sv, tx, nv, dir_view, ruv, rw, bs);
            if (sampled) mat_eval_pdf<false, false, true, MASK>(sv, tx, nv, dir_view, bs.dir_out, f, pdf);
        }
    }
    if (off_done) {
        const int k = L.k();
        D3 cX = splat(0);
        double wgt = 1.0;
        if (off_alive) {                                                            // :1019-1045
            const D3 c0 = (nv.light_id >= 0) ? emission(sv, nv, -ray.dir) : splat(1.0);   // :496-508
            cX = c0 * L.pdf; wgt = L.prob / (L.prob + 1.0 * L.pdf);
        }
        bool flagged = false;
        acc_offset(acc, k, L.contrib, cX, wgt, L.prob, spp, lc, flagged);
        if (k == 3) { acc_base(acc, lp.radiance(), L.prob, spp, lc); act = ACT_NEXT_SAMPLE; }
        else { L.kc = k + 1; act = ACT_OFFSET_RAY; }
    } else if (act == ACT_BOUNCE) {                                                  // iteration `it` starts at nv
        const int it = L.num_vertices - 3;
        if (!sampled) act = ACT_NEXT_SAMPLE;                                        // :545-548: GraidentPTRadiance{}
        else {
            if (it < kLogCap) { log[it * kBlock].mat = nv.material_id; if (pdf <= 0) log[it * kBlock].p2 = -1.0; }
            set_n_iter(L, min(it + 1, kLogCap));
            if (pdf <= 0) act = ACT_OFFSETS;                                        // :760-763
            else {
                if (bs.eta != 0) lp.set_eta_scale(lp.eta_scale() / (bs.eta * bs.eta));   // :553-558
                L.org = nv.position; L.dir = bs.dir_out; L.f = f; L.pdf = pdf; L.st = S_BOUNCE;
            }
        }
    }
    if (act == ACT_OFFSETS) {
        // A base path whose PRIMARY vertex carries a one-sided lobe gets the lazy rule of render_device.h: only offsets that hit
        // the same material are valid, and an offset on a one-sided lobe that survives the re-sampling of iteration 0 is retired
        // by iteration 1 (its direction now points away from the surface, so the next sample_bsdf sees dir_in below it; a
        // sampled direction below the geometric surface has pdf 0 and died in iteration 0). So once iteration 1 has run past
        // its pdf test, all four offsets are dead whatever their primary rays would hit: contribX = 0, w = 1, no ray traced.
        // (Iteration 1 broken at pdf <= 0 on a vertex of mat0 keeps them observable: C_BROKE_BOUNCE2 of the one-sided machine.)
        const int n_it = n_iter(L), t0 = tx.materials[L.mat0()].type;
        const bool one_sided0 = !(t0 == GDPT_MAT_DISNEY_GLASS || t0 == GDPT_MAT_DISNEY_BSDF || t0 == GDPT_MAT_ROUGHDIELECTRIC);
        bool unobservable = false;
        if (one_sided0 && n_it >= 2) {
            const BounceLog e1 = log[1 * kBlock];
            unobservable = (n_it >= 3) || !(e1.p2 < 0 && e1.mat == L.mat0());
        }
        if (unobservable) { acc_no_offsets(acc, lp.radiance(), L.contrib, L.prob, spp, lc); act = ACT_NEXT_SAMPLE; }
        else { L.kc = 0; act = ACT_OFFSET_RAY; }
    }
    if (act == ACT_NEXT_SAMPLE) {
        L.s++;
        if (L.s >= L.s_end) L.st = S_DONE; else act = ACT_PRIMARY_RAY;
    }
    if (act == ACT_PRIMARY_RAY || act == ACT_OFFSET_RAY) {
        Pcg r = pcg_init(base + (unsigned long long)L.s);
        const double rx = pcg_real(r), ry = pcg_real(r);                            // :360-361
        int ox = 0, oy = 0;
        if (act == ACT_PRIMARY_RAY) { L.rng_state = r.state; L.rng_inc = r.inc; L.st = S_PRIMARY; }
        else {
            const int k = L.k();
            ox = (k == 0) ? -1 : (k == 1 ? 1 : 0); oy = (k == 2) ? 1 : (k == 3 ? -1 : 0);   // x0,x1,y0,y1 (:385-403)
            L.st = S_OFFSET;
        }
        FilterCache fc;
        if (act == ACT_OFFSET_RAY) fc = lp.fc();
        Ray pr = sample_primary<true>(cam, (x + ox) + rx, (y + oy) + ry, &fc, act == ACT_PRIMARY_RAY);
        if (act == ACT_PRIMARY_RAY) lp.set_fc(fc);
        L.org = pr.org; L.dir = pr.dir;
    }
    if (lane_tracing(L.st)) trav_init(sv, tv, __builtin_huge_val());      // a fresh pending ray (S_REPLAY keeps the offset's hit)
}

template <bool LDS_SCENE, unsigned MASK = kAllMaterials>
__global__ __launch_bounds__(kBlock, 2) void gdpt_render_twosided(DevSceneView sv, KernelArgs a, BounceLog *logs) {
    __shared__ int s_stack_fixed[LDS_SCENE ? kLdsSceneLevels * kBlock : 1];
    __shared__ __attribute__((aligned(16))) unsigned char s_scene[LDS_SCENE ? kLdsSceneBytes : 16];
    __shared__ double s_acc[15 * kBlock];
    __shared__ double s_priv[kPrivDoubles * kBlock];
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];       // HBM scenes: the traversal stack (render_device.h)
    int *s_stack = LDS_SCENE ? s_stack_fixed : (int *)s_dyn;
    const int tid = threadIdx.x;
    TraceCtx tx = setup_trace<LDS_SCENE, true>(sv, s_scene, s_stack, tid, kBlock, a.count != 0);
    const int W = sv.cam.width;
    const double spp = (double)a.spp;
    AccLds acc; acc.slot = s_acc + tid; acc.stride = kBlock;
    acc.init();
    LanePriv2 lp; lp.slot = s_priv + tid; lp.stride = kBlock;
    // lane-interleaved: entry `it` of lane `tid` sits at it * kBlock + tid of the block's slab, so the 64 lanes of a wave
    // touch one contiguous kilobyte per logged iteration (a per-lane 16 KB stride made every access its own cache line)
    BounceLog *log = logs + (size_t)blockIdx.x * kBlock * kLogCap + tid;
    LaneCounters lc = {0, 0, 0};
    TraceCounters tc = {0, 0, 0, 0, 0, 0};
    Lane L;
    Trav tv;
    trav_init(sv, tv, __builtin_huge_val());
    L.s = 0; L.s_end = 0; L.st = S_DONE;
    L.kc = 0; L.num_vertices = 0; L.mats = 0; L.rng_state = 0; L.rng_inc = 1;
    L.org = L.dir = L.f = splat(0); L.pdf = 1;
    int x = 0, y = 0;
    unsigned long long base = 0;
    long long my_item = -1;
    WaveQueue wq;
    for (;;) {
        const bool idle = (L.st == S_DONE);
        if (idle && my_item >= 0) {
            Accum r = acc.result();
            typedef double d2 __attribute__((ext_vector_type(2)));
            d2 *dst = (d2 *)(a.partials + (size_t)my_item * 16);
            dst[0] = d2{r.r.x, r.r.y}; dst[1] = d2{r.r.z, r.dx0.x}; dst[2] = d2{r.dx0.y, r.dx0.z}; dst[3] = d2{r.dy0.x, r.dy0.y};
            dst[4] = d2{r.dy0.z, r.dx1.x}; dst[5] = d2{r.dx1.y, r.dx1.z}; dst[6] = d2{r.dy1.x, r.dy1.y}; dst[7] = d2{r.dy1.z, 0.0};
            acc.init();
            my_item = -1;
        }
        const long long got_item = wq.take(a, idle, tid);
        if (got_item >= 0) {
            my_item = got_item;
            int s0, s1;
            const bool inside = item_to_pixel(a, W, (unsigned)my_item, x, y, s0, s1, a.chunk_begin);
            base = ((unsigned long long)y * W + x) * (unsigned long long)a.spp;
            L.s = s0; L.s_end = s1;
            L.st = (inside && s0 < s1) ? S_START : S_DONE;
        }
        if (!__any(L.st != S_DONE)) { if (wq.exhausted) break; else continue; }
        trace_pending<TraceCfg<true, true, !LDS_SCENE>>(sv, tx, L, tv, a.thresh_a, a.thresh_c, tc);
        if (L.st == S_REPLAY || lane_ready(L, tv)) {
            if (tx.count) { tc.lane_steps++; if (wave_leader()) tc.wave_steps++; }
            lane_step2<MASK>(sv, tx, a.max_depth, spp, x, y, base, L, tv, lp, acc, lc, log, a.replay_per_step);
        }
    }
    flush_counters(a, lc, tc, a.count != 0);
}

} // namespace gd

namespace gdpt {
void launch_phases_twosided(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, bool lds, unsigned material_mask, void *bounce_log, hipStream_t stream);
void launch_phases_twosided_glass(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, void *bounce_log, hipStream_t stream);
size_t twosided_log_bytes(unsigned blocks);
} // namespace gdpt
