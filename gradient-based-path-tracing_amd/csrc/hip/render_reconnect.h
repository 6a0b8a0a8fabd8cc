// render_reconnect.h — GDPT_SHIFT_RECONNECT: gradient-domain path tracing with a reconnection shift.
//
// "a *correct* GDPT mode (smallgdpt-style reconnection) behind a flag". The parity mode
// (grad_path_tracing, src/path_tracing.h:351-560) replays random numbers on the offset paths and never brings them
// back to the base path, so its gradient buffers are not finite differences of the image. This mode follows the
// scheme of the reference's own sketch, /root/reference/small_gdpt.py:163-219 (shiftPath) and :380-420 (estimator):
//   * base path: exactly the parity mode's base path (same draws, BSDF sampling, emitter hits, Russian roulette),
//   * offset path through pixel (x±1, y), (x, y±1): same sub-pixel position, first vertex v1' from its own primary
//     ray, then reconnected to the base path's second vertex v2 (visibility ray); v2 onward is shared,
//   * Jacobian |dω1'/dω1| = (cosθ2'/d'²)/(cosθ2/d²)                                        (small_gdpt.py:186-199),
//   * per path length, gradient += w·(f − f'·J)/p with w = p/(p + p'·J), failed shifts f' = 0, w = 1  (:396-420).
// A base sample carries one path per emitter hit along it; the terms are grouped by where the shift acts:
//   length 1 (v1 emitter)            f = Le(v1),                 f' = Le(v1'),                     w = 1/2
//   length 2 (v2 emitter)            f = A1·Le(v2→v1),           f'J = (f1'/p1)·J·Le(v2→v1'),      w = 1/(1+p1'J/p1)
//   length ≥3 (shared tail S)        f = A1·B2·S,                f'J = (f1'/p1)·J·(f2'/p2)·S,      w = 1/(1+p1'J p2'/(p1 p2))
// with A1 = f1/p1, B2 = f2/p2 (LaJolla's f includes the cosine) and S = Σ_{k≥3} U_k Le_k the radiance the base
// path gathers after v2, Russian roulette included (it is part of the true density p and cancels in w: any positive
// function of the path may stand in for p there as long as both directions use the same one).
// Buffers: cx0 ← pixel (x−1), cx1 ← (x+1), cy0 ← (y−1), cy1 ← (y+1), so that gdpt_assemble's cx0[x]+cx1[x−1] and
// cy0[y]+cy1[y−1] (src/render.cpp:340-350) are both estimates of I(x)−I(x−1), I(y)−I(y−1). (The parity mode keeps the
// reference's y offsets, which in image-row order point the other way — small_gdpt.py stores rows bottom-up.)
#pragma once
#include "render_device.h"

namespace gd {

struct ShiftV1 { Vertex v; D3 dir_view; bool ok; };

// Lambert-only scenes inline the three-line lobe; everything else goes through the full material switch (inlined at each
// of the eight call sites: DESIGN.md 4.4 records why the switch is not an out-of-line function on this toolchain).
template <bool LAMBERT> struct Mat {
    static GD bool sample(const DevSceneView &sv, const TraceCtx &tx, const Vertex &v, D3 in, D2 ruv, double rw, BsdfSample &s) {
        return mat_sample<LAMBERT, true, true>(sv, tx, v, in, ruv, rw, s);
    }
    static GD void eval_pdf(const DevSceneView &sv, const TraceCtx &tx, const Vertex &v, D3 in, D3 out, D3 &f, double &pdf) {
        mat_eval_pdf<LAMBERT, true, true>(sv, tx, v, in, out, f, pdf);
    }
};

template <class TC, bool LAMBERT>
GD void grad_sample_reconnect(const DevSceneView &sv, const TraceCtx &tx, int max_depth, int x, int y, Pcg &rng, double spp,
                              AccReg &acc, LaneCounters &lc, TraceCounters &tc) {
    const DevCamera &cam = sv.cam;
    const int w = cam.width, h = cam.height;
    const double rng_x = pcg_real(rng), rng_y = pcg_real(rng);
    Ray ray = sample_primary(cam, (x + rng_x) / w, (y + rng_y) / h);
    const double rd_spread = 0.25 / (double)max(w, h);
    Vertex v1;
    if (!intersect_ctx<TC>(sv, tx, ray, rd_spread, v1, lc, tc)) return;
    ShiftV1 sh[4];
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        const int ox = (k == 0) ? -1 : (k == 1 ? 1 : 0), oy = (k == 2) ? -1 : (k == 3 ? 1 : 0);
        Ray r = sample_primary(cam, ((x + ox) + rng_x) / w, ((y + oy) + rng_y) / h);
        sh[k].ok = intersect_ctx<TC>(sv, tx, r, rd_spread, sh[k].v, lc, tc);
        sh[k].dir_view = -r.dir;
    }
    // signs of (f - f'J) per buffer: cx0, cx1, cy0, cy1 = acc slots 1, 3, 2, 4
    const int slot[4] = {1, 3, 2, 4};
    const double sgn[4] = {1.0, -1.0, 1.0, -1.0};
    bool flagged = false;
    auto add_term = [&](int k, D3 f_over_p, D3 fo_over_p, double wgt) {
        D3 g = (f_over_p - fo_over_p) * (sgn[k] * wgt / spp);
        if (!isfinite(g.x + g.y + g.z)) { if (!flagged) lc.nonfinite++; flagged = true; }
        acc.add(slot[k], g);
    };

    // ---- length 1 ----
    const D3 dir_view1 = -ray.dir;
    D3 radiance = splat(0);
    {
        const D3 Le1 = v1.light_id >= 0 ? emission(sv, v1, dir_view1) : splat(0);
        radiance = Le1;
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
            if (!sh[k].ok) { add_term(k, Le1, splat(0), 1.0); continue; }
            const D3 Lo = sh[k].v.light_id >= 0 ? emission(sv, sh[k].v, sh[k].dir_view) : splat(0);
            add_term(k, Le1, Lo, 0.5);
        }
    }
    if (!loop_allows(max_depth, 3)) { acc.add(0, radiance / spp); return; }

    // ---- bounce at v1 ----
    lc.bounces++;
    D2 ruv; ruv.x = pcg_real(rng); ruv.y = pcg_real(rng);
    double rw = pcg_real(rng);
    BsdfSample bs;
    double eta_scale = 1.0;
    if (!Mat<LAMBERT>::sample(sv, tx, v1, dir_view1, ruv, rw, bs)) { acc.add(0, radiance / spp); return; }
    const D3 w1 = bs.dir_out;
    if (bs.eta != 0) eta_scale /= (bs.eta * bs.eta);
    D3 f1; double p1;
    Mat<LAMBERT>::eval_pdf(sv, tx, v1, dir_view1, w1, f1, p1);
    Ray r1; r1.org = v1.position; r1.dir = w1; r1.tnear = sv.isect_eps; r1.tfar = __builtin_huge_val();
    Vertex v2;
    const bool hit2 = intersect_ctx<TC>(sv, tx, r1, 0.0, v2, lc, tc);
    if (!(p1 > 0) || !hit2) { acc.add(0, radiance / spp); return; }
    const D3 A1 = f1 / p1;
    // Russian roulette exactly as the parity mode places it: after the emitter term of the vertex just reached, from
    // the throughput *before* this bounce's factor (src/path_tracing.h:540-552)
    D3 throughput = splat(1.0);            // the base path's throughput
    double rr_all = 1.0;                   // Russian roulette factors met before the tail starts (0: path ended)
    if (3 - 1 >= sv.rr_depth) {
        const double rr_prob = fmin(maxc((1 / eta_scale) * throughput), 0.95);
        if (pcg_real(rng) > rr_prob) rr_all = 0.0; else rr_all /= rr_prob;
    }
    throughput = A1 * rr_all;

    // ---- reconnect the four offsets to v2 ----
    const D3 d12 = v2.position - v1.position;
    const double dist2 = dot(d12, d12), cos2 = fabs(dot(w1, v2.gn));
    D3 fo1[4], wo1[4];                     // f1'/p1 * J  and the offset's direction into v2
    double ro1[4];                         // p1' J / p1
    bool rec[4];
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        rec[k] = false; fo1[k] = splat(0); ro1[k] = 0; wo1[k] = w1;
        if (!sh[k].ok || !(cos2 > 0)) continue;
        const Vertex &o = sh[k].v;
        D3 d = v2.position - o.position;
        const double od2 = dot(d, d);
        if (!(od2 > 0)) continue;
        const double od = sqrt(od2);
        const D3 wo = d / od;
        const double ocos2 = fabs(dot(wo, v2.gn));
        // the offset must see v2 on the side the base path arrived at
        if (!(ocos2 > 0) || dot(wo, v2.gn) * dot(w1, v2.gn) <= 0) continue;
        D3 f; double p;
        Mat<LAMBERT>::eval_pdf(sv, tx, o, sh[k].dir_view, wo, f, p);
        if (!(p > 0)) continue;
        Ray sr; sr.org = o.position; sr.dir = wo; sr.tnear = sv.isect_eps; sr.tfar = (1 - sv.isect_eps) * od;
        if (occluded_ctx<TC>(sv, tx, sr, lc, tc)) continue;
        const double J = (ocos2 / od2) / (cos2 / dist2);
        rec[k] = true; wo1[k] = wo;
        fo1[k] = f * (J / p1); ro1[k] = p * J / p1;
    }

    // ---- length 2 ----
    if (v2.light_id >= 0) {
        const D3 Le2 = emission(sv, v2, -w1);
        radiance = radiance + A1 * Le2;
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
            if (!rec[k]) { add_term(k, A1 * Le2, splat(0), 1.0); continue; }
            add_term(k, A1 * Le2, fo1[k] * emission(sv, v2, -wo1[k]), 1.0 / (1.0 + ro1[k]));
        }
    }
    if (rr_all == 0.0 || !loop_allows(max_depth, 4)) { acc.add(0, radiance / spp); return; }

    // ---- bounce at v2 ----
    lc.bounces++;
    ruv.x = pcg_real(rng); ruv.y = pcg_real(rng); rw = pcg_real(rng);
    const D3 dir_view2 = -w1;
    if (!Mat<LAMBERT>::sample(sv, tx, v2, dir_view2, ruv, rw, bs)) { acc.add(0, radiance / spp); return; }
    const D3 w2 = bs.dir_out;
    if (bs.eta != 0) eta_scale /= (bs.eta * bs.eta);
    D3 f2; double p2;
    Mat<LAMBERT>::eval_pdf(sv, tx, v2, dir_view2, w2, f2, p2);
    if (!(p2 > 0)) { acc.add(0, radiance / spp); return; }
    const D3 B2 = f2 / p2;
    D3 fo2[4]; double ro2[4];              // offset factors through v2: (f1'/p1) J (f2'/p2)  and  p1' J p2' / (p1 p2)
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        fo2[k] = splat(0); ro2[k] = 0;
        if (!rec[k]) continue;
        D3 f; double p;
        Mat<LAMBERT>::eval_pdf(sv, tx, v2, -wo1[k], w2, f, p);
        if (!(p > 0)) { rec[k] = false; continue; }
        fo2[k] = fo1[k] * (f / p2); ro2[k] = ro1[k] * (p / p2);
    }

    // ---- the shared tail: the base path from v2 on, gathering S = sum U_k Le_k (U relative to A1*B2) ----
    D3 U = splat(rr_all), S = splat(0);
    D3 pendT = B2, pendU = splat(1.0);     // factor of the bounce just taken: enters after its Russian roulette
    Ray cur; cur.org = v2.position; cur.dir = w2; cur.tnear = sv.isect_eps; cur.tfar = __builtin_huge_val();
    Vertex vertex;
    bool hit = intersect_ctx<TC>(sv, tx, cur, 0.0, vertex, lc, tc);
    for (int num_vertices = 4; hit;) {
        if (vertex.light_id >= 0) S = S + U * pendU * emission(sv, vertex, -cur.dir);
        double rr_prob = 1;
        if (num_vertices - 1 >= sv.rr_depth) {
            rr_prob = fmin(maxc((1 / eta_scale) * throughput), 0.95);
            if (pcg_real(rng) > rr_prob) break;
        }
        throughput = throughput * pendT / rr_prob; U = U * pendU / rr_prob;
        num_vertices++;
        if (!loop_allows(max_depth, num_vertices)) break;
        lc.bounces++;
        const D3 dir_view = -cur.dir;
        ruv.x = pcg_real(rng); ruv.y = pcg_real(rng); rw = pcg_real(rng);
        if (!Mat<LAMBERT>::sample(sv, tx, vertex, dir_view, ruv, rw, bs)) break;
        if (bs.eta != 0) eta_scale /= (bs.eta * bs.eta);
        D3 f; double p;
        Mat<LAMBERT>::eval_pdf(sv, tx, vertex, dir_view, bs.dir_out, f, p);
        if (!(p > 0)) break;
        pendT = f / p; pendU = pendT;
        cur.org = vertex.position; cur.dir = bs.dir_out;
        hit = intersect_ctx<TC>(sv, tx, cur, 0.0, vertex, lc, tc);
    }
    const D3 base3 = A1 * B2 * S;
    radiance = radiance + base3;
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        if (!rec[k]) { add_term(k, base3, splat(0), 1.0); continue; }
        add_term(k, base3, fo2[k] * S, 1.0 / (1.0 + ro2[k]));
    }
    if (!isfinite(radiance.x + radiance.y + radiance.z) && !flagged) lc.nonfinite++;
    acc.add(0, radiance / spp);
}

// LDS_SCENE: nodes (BVH4 form), primitive records, shading table and materials are copied into the block's LDS first
// (scenes of cbox size), exactly as the lane machine does.
template <bool LAMBERT, bool LDS_SCENE>
__global__ __launch_bounds__(kBlock, 2) void gdpt_render_reconnect(DevSceneView sv, KernelArgs a) {
    __shared__ int s_stack[GDPT_BVH_MAX_DEPTH * kBlock];
    __shared__ __attribute__((aligned(16))) unsigned char s_scene[LDS_SCENE ? kLdsSceneBytes : 16];
    using TC = TraceCfg<true, true, !LDS_SCENE>;
    const int tid = threadIdx.x;
    TraceCtx tx = setup_trace<LDS_SCENE, true>(sv, s_scene, s_stack, tid, kBlock, a.count != 0);
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
            grad_sample_reconnect<TC, LAMBERT>(sv, tx, a.max_depth, x, y, rng, (double)a.spp, acc, lc, tc);
        }
    }
    Accum sum = acc.result();
    reduce_and_store(a, sum, K, valid && c == 0, x, y, W);
    flush_counters(a, lc, tc, a.count != 0);
}

} // namespace gd

