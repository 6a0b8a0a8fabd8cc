// sbvh.cpp — SAH BVH2 with spatial splits (the split-BVH idea of Stich, Friedrich and Dietrich, "Spatial splits in bounding volume
// hierarchies", HPG 2009, which is also what the reference asks Embree for with RTC_BUILD_QUALITY_HIGH, src/scene.cpp:20-31):
// at every node the best object split (binned SAH over reference centroids, as host/bvh.cpp) competes with the best spatial split
// (the node's box chopped into bins along each axis, every reference clipped to the bins it spans); a spatial split sends a
// straddling triangle to BOTH children, each with the box of its part on that side. A reference never changes what a ray hits —
// every reference of a triangle points at the same primitive record, and the references' boxes together cover the triangle — only
// how many boxes it visits (closest hit = min fp32 t, ties to the lowest id: device_trace.h). Same output form, depth bound and
// leaf policy as build_bvh; the extra references are limited by a budget.
#include "bvh.h"
#include "../capi_common.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <exception>
#include <stdexcept>
#include <thread>
#include <utility>

namespace gdpt {
namespace {

constexpr float kInf = std::numeric_limits<float>::infinity();

struct SRef { float lo[3], hi[3]; uint32_t prim; };

struct SBox {
    float lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; k++) { lo[k] = kInf; hi[k] = -kInf; } }
    void grow(const float *a, const float *b) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], a[k]); hi[k] = std::max(hi[k], b[k]); } }
    void grow(const SBox &o) { grow(o.lo, o.hi); }
    void grow(const SRef &r) { grow(r.lo, r.hi); }
    bool empty() const { return !(lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]); }
    float half_area() const {
        if (empty()) return 0.f;
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

float down(double x) { float f = (float)x; if ((double)f > x) f = std::nextafterf(f, -kInf); return std::nextafterf(f, -kInf); }   // below x, one ulp to spare
float up(double x) { float f = (float)x; if ((double)f < x) f = std::nextafterf(f, kInf); return std::nextafterf(f, kInf); }

struct SBuilder {
    const std::vector<float> &tv;          // 9 floats per triangle; primitives from tv.size()/9 on (spheres) have a box only
    size_t ntri;
    BvhBuildResult out;
    std::vector<uint32_t> *ref_prim;
    float root_area = 0.f;
    double alpha = 1e-5;
    long long spare = 0;                   // references that may still be added
    uint32_t leaf_max = GDPT_LEAF_MAX_PRIMS;
    float leaf_factor = 0.7f;              // (build_bvh: 0.8; the flat leaf test of the kernels that walk these trees pays for four slots per visit: +1 % on five scenes)
#ifndef GDPT_SBVH_BINS
#define GDPT_SBVH_BINS 32
#endif
    static constexpr int NB = GDPT_SBVH_BINS;

    SBuilder(const std::vector<float> &tri_verts, std::vector<uint32_t> *rp) : tv(tri_verts), ntri(tri_verts.size() / 9), ref_prim(rp) {}

    static int ceil_log2(uint32_t n) { int l = 0; while ((1u << l) < n) l++; return l; }
    static SBox box_of(const std::vector<SRef> &r) { SBox b; b.reset(); for (const SRef &x : r) b.grow(x); return b; }

    // The two parts of reference r on either side of the plane x[axis] = pos: boxes of the triangle's vertices on that side and of
    // the points where its edges cross the plane (the part of the triangle on that side lies in their hull), rounded outward, cut
    // back to r's box and to the plane. A primitive without vertices (sphere) keeps its box cut at the plane. Returns which exist.
    void split_ref(const SRef &r, int axis, float pos, SRef &L, SRef &R, bool &hasL, bool &hasR) const {
        L = r; R = r;
        if (r.prim < ntri) {
            double lo[2][3], hi[2][3];
            for (int s = 0; s < 2; s++) for (int k = 0; k < 3; k++) { lo[s][k] = std::numeric_limits<double>::infinity(); hi[s][k] = -lo[s][k]; }
            auto add = [&](int s, const double *p) { for (int k = 0; k < 3; k++) { lo[s][k] = std::min(lo[s][k], p[k]); hi[s][k] = std::max(hi[s][k], p[k]); } };
            const float *v = &tv[9 * (size_t)r.prim];
            for (int e = 0; e < 3; e++) {
                const double a[3] = {v[3 * e], v[3 * e + 1], v[3 * e + 2]};
                const int f = (e + 1) % 3;
                const double b[3] = {v[3 * f], v[3 * f + 1], v[3 * f + 2]};
                if (a[axis] <= (double)pos) add(0, a);
                if (a[axis] >= (double)pos) add(1, a);
                if ((a[axis] < (double)pos && b[axis] > (double)pos) || (a[axis] > (double)pos && b[axis] < (double)pos)) {
                    const double t = ((double)pos - a[axis]) / (b[axis] - a[axis]);
                    double p[3];
                    for (int k = 0; k < 3; k++) {       // between a and b whatever the rounding of t
                        p[k] = a[k] + t * (b[k] - a[k]);
                        p[k] = std::min(std::max(p[k], std::min(a[k], b[k])), std::max(a[k], b[k]));
                    }
                    p[axis] = (double)pos;
                    // the crossing point carries the rounding of t: widen its contribution by a relative 1e-12 of the edge's extent
                    double q0[3], q1[3];
                    for (int k = 0; k < 3; k++) { const double w = 1e-12 * std::fabs(b[k] - a[k]); q0[k] = p[k] - w; q1[k] = p[k] + w; }
                    q0[axis] = q1[axis] = (double)pos;
                    add(0, q0); add(0, q1); add(1, q0); add(1, q1);
                }
            }
            for (int k = 0; k < 3; k++) {
                L.lo[k] = std::max(r.lo[k], lo[0][k] <= hi[0][k] ? down(lo[0][k]) : kInf); L.hi[k] = std::min(r.hi[k], lo[0][k] <= hi[0][k] ? up(hi[0][k]) : -kInf);
                R.lo[k] = std::max(r.lo[k], lo[1][k] <= hi[1][k] ? down(lo[1][k]) : kInf); R.hi[k] = std::min(r.hi[k], lo[1][k] <= hi[1][k] ? up(hi[1][k]) : -kInf);
            }
        }
        L.hi[axis] = std::min(L.hi[axis], pos);
        R.lo[axis] = std::max(R.lo[axis], pos);
        hasL = L.lo[0] <= L.hi[0] && L.lo[1] <= L.hi[1] && L.lo[2] <= L.hi[2];
        hasR = R.lo[0] <= R.hi[0] && R.lo[1] <= R.hi[1] && R.lo[2] <= R.hi[2];
        if (!hasL && !hasR) { hasL = true; L = r; }      // (cannot happen for a reference that straddles the plane; keep it whole if it does)
    }

    struct ObjSplit { int axis = -1, bin = -1; float cost = kInf, lo = 0, scale = 0; SBox lb, rb; };
    ObjSplit best_object_split(const std::vector<SRef> &refs) const {
        ObjSplit best;
        SBox cb; cb.reset();
        for (const SRef &r : refs) { const float c[3] = {0.5f * r.lo[0] + 0.5f * r.hi[0], 0.5f * r.lo[1] + 0.5f * r.hi[1], 0.5f * r.lo[2] + 0.5f * r.hi[2]}; cb.grow(c, c); }
        for (int ax = 0; ax < 3; ax++) {
            const float lo = cb.lo[ax], hi = cb.hi[ax];
            if (!(hi > lo)) continue;
            SBox bb[NB]; uint32_t cnt[NB];
            for (int k = 0; k < NB; k++) { bb[k].reset(); cnt[k] = 0; }
            const float scale = NB / (hi - lo);
            for (const SRef &r : refs) {
                int k = (int)(((0.5f * r.lo[ax] + 0.5f * r.hi[ax]) - lo) * scale);
                k = std::min(std::max(k, 0), NB - 1);
                bb[k].grow(r); cnt[k]++;
            }
            SBox rbox[NB]; uint32_t rcnt[NB];
            SBox acc; acc.reset(); uint32_t c = 0;
            for (int k = NB - 1; k > 0; k--) { acc.grow(bb[k]); c += cnt[k]; rbox[k] = acc; rcnt[k] = c; }
            acc.reset(); c = 0;
            for (int k = 0; k < NB - 1; k++) {
                acc.grow(bb[k]); c += cnt[k];
                if (c == 0 || rcnt[k + 1] == 0) continue;
                const float cost = acc.half_area() * (float)c + rbox[k + 1].half_area() * (float)rcnt[k + 1];
                if (cost < best.cost) { best.cost = cost; best.axis = ax; best.bin = k; best.lo = lo; best.scale = scale; best.lb = acc; best.rb = rbox[k + 1]; }
            }
        }
        return best;
    }
    static bool obj_left(const ObjSplit &s, const SRef &r) {
        int k = (int)(((0.5f * r.lo[s.axis] + 0.5f * r.hi[s.axis]) - s.lo) * s.scale);
        k = std::min(std::max(k, 0), NB - 1);
        return k <= s.bin;
    }

    struct SpatialSplit { int axis = -1; float pos = 0, cost = kInf; uint32_t nl = 0, nr = 0; };
    SpatialSplit best_spatial_split(const std::vector<SRef> &refs, const SBox &nb) const {
        SpatialSplit best;
        for (int ax = 0; ax < 3; ax++) {
            const float lo = nb.lo[ax], hi = nb.hi[ax];
            if (!(hi > lo)) continue;
            const float width = (hi - lo) / NB;
            if (!(width > 0.f) || !std::isfinite(width)) continue;
            float plane[NB + 1];
            for (int k = 0; k <= NB; k++) plane[k] = (k == NB) ? hi : lo + width * (float)k;
            bool mono = true;
            for (int k = 0; k < NB; k++) if (!(plane[k] < plane[k + 1])) mono = false;
            if (!mono) continue;
            SBox bb[NB]; uint32_t enter[NB], leave[NB];
            for (int k = 0; k < NB; k++) { bb[k].reset(); enter[k] = leave[k] = 0; }
            auto bin_of = [&](float x) {           // the bin whose [plane[k], plane[k+1]) holds x, by the planes themselves
                int k = (int)((x - lo) / width);
                k = std::min(std::max(k, 0), NB - 1);
                while (k > 0 && x < plane[k]) k--;
                while (k < NB - 1 && x >= plane[k + 1]) k++;
                return k;
            };
            for (const SRef &r : refs) {
                const int b0 = bin_of(r.lo[ax]);
                int b1 = bin_of(r.hi[ax]);
                if (b1 > b0 && r.hi[ax] <= plane[b1]) b1--;            // touches the plane from below: belongs to the bin below
                enter[b0]++; leave[b1]++;
                if (b0 == b1) { bb[b0].grow(r); continue; }
                SRef cur = r;
                for (int b = b0; b < b1; b++) {
                    SRef L, R; bool hl, hr;
                    split_ref(cur, ax, plane[b + 1], L, R, hl, hr);
                    if (hl) bb[b].grow(L);
                    if (!hr) { cur.lo[0] = kInf; break; }
                    cur = R;
                }
                if (cur.lo[0] != kInf) bb[b1].grow(cur);
            }
            SBox rbox[NB]; uint32_t rcnt[NB];
            SBox acc; acc.reset(); uint32_t c = 0;
            for (int k = NB - 1; k > 0; k--) { acc.grow(bb[k]); c += leave[k]; rbox[k] = acc; rcnt[k] = c; }
            acc.reset(); c = 0;
            for (int k = 0; k < NB - 1; k++) {
                acc.grow(bb[k]); c += enter[k];
                if (c == 0 || rcnt[k + 1] == 0) continue;
                const float cost = acc.half_area() * (float)c + rbox[k + 1].half_area() * (float)rcnt[k + 1];
                if (cost < best.cost) { best.cost = cost; best.axis = ax; best.pos = plane[k + 1]; best.nl = c; best.nr = rcnt[k + 1]; }
            }
        }
        return best;
    }

    int32_t make_leaf(const std::vector<SRef> &refs) {
        const uint32_t first = (uint32_t)ref_prim->size();
        for (const SRef &r : refs) { out.order.push_back((uint32_t)ref_prim->size()); ref_prim->push_back(r.prim); }
        const uint32_t packed = (first << 2) | ((uint32_t)refs.size() - 1u);
        return ~(int32_t)packed;
    }

    // Splits refs into l and r (both non-empty, both smaller than refs unless a spatial split was chosen with duplicates).
    void split(std::vector<SRef> &refs, bool force_median, std::vector<SRef> &l, std::vector<SRef> &r) {
        const size_t n = refs.size();
        l.clear(); r.clear();
        if (!force_median) {
            const ObjSplit os = best_object_split(refs);
            bool try_spatial = spare > 0 && root_area > 0.f;
            if (try_spatial && os.axis >= 0) {      // only where the object split's children overlap noticeably
                SBox ov;
                for (int k = 0; k < 3; k++) { ov.lo[k] = std::max(os.lb.lo[k], os.rb.lo[k]); ov.hi[k] = std::min(os.lb.hi[k], os.rb.hi[k]); }
                try_spatial = !ov.empty() && (double)ov.half_area() / (double)root_area > alpha;
            }
            if (try_spatial) {
                const SBox nb = box_of(refs);
                const SpatialSplit ss = best_spatial_split(refs, nb);
                const long long extra = (long long)ss.nl + (long long)ss.nr - (long long)n;
                if (ss.axis >= 0 && ss.cost < os.cost && ss.nl < n && ss.nr < n && extra <= spare) {
                    // references on one side first; a straddling one is then either split or, where that is cheaper by the SAH,
                    // handed whole to one side ("reference unsplitting" of the paper, section 4.4: no duplicate for it)
                    std::vector<const SRef *> straddle;
                    SBox lb, rb; lb.reset(); rb.reset();
                    for (const SRef &x : refs) {
                        if (x.hi[ss.axis] <= ss.pos) { l.push_back(x); lb.grow(x); }
                        else if (x.lo[ss.axis] >= ss.pos) { r.push_back(x); rb.grow(x); }
                        else straddle.push_back(&x);
                    }
                    std::vector<SRef> pl(straddle.size()), pr(straddle.size());
                    std::vector<char> hl(straddle.size()), hr(straddle.size());
                    for (size_t i = 0; i < straddle.size(); i++) {
                        bool a, b;
                        split_ref(*straddle[i], ss.axis, ss.pos, pl[i], pr[i], a, b);
                        hl[i] = a; hr[i] = b;
                        if (a) lb.grow(pl[i]);
                        if (b) rb.grow(pr[i]);
                    }
                    float nl = (float)l.size(), nr = (float)r.size();
                    for (size_t i = 0; i < straddle.size(); i++) { nl += hl[i] ? 1.f : 0.f; nr += hr[i] ? 1.f : 0.f; }
                    for (size_t i = 0; i < straddle.size(); i++) {
                        if (!hl[i]) { r.push_back(pr[i]); continue; }
                        if (!hr[i]) { l.push_back(pl[i]); continue; }
                        SBox lw = lb, rw = rb; lw.grow(*straddle[i]); rw.grow(*straddle[i]);
                        const float c_split = lb.half_area() * nl + rb.half_area() * nr;
                        const float c_left = lw.half_area() * nl + rb.half_area() * (nr - 1.f);
                        const float c_right = lb.half_area() * (nl - 1.f) + rw.half_area() * nr;
                        if (c_left < c_split && c_left <= c_right && nr > 1.f) { l.push_back(*straddle[i]); lb = lw; nr -= 1.f; }
                        else if (c_right < c_split && nl > 1.f) { r.push_back(*straddle[i]); rb = rw; nl -= 1.f; }
                        else { l.push_back(pl[i]); r.push_back(pr[i]); }
                    }
                    if (!l.empty() && !r.empty() && l.size() < n && r.size() < n) {
                        spare -= (long long)(l.size() + r.size()) - (long long)n;
                        return;
                    }
                    l.clear(); r.clear();
                }
            }
            if (os.axis >= 0) {
                for (const SRef &x : refs) (obj_left(os, x) ? l : r).push_back(x);
                if (!l.empty() && !r.empty()) return;
                l.clear(); r.clear();
            }
        }
        // median split along the widest centroid axis (also the depth-bounding fallback)
        SBox cb; cb.reset();
        for (const SRef &x : refs) { const float c[3] = {0.5f * x.lo[0] + 0.5f * x.hi[0], 0.5f * x.lo[1] + 0.5f * x.hi[1], 0.5f * x.lo[2] + 0.5f * x.hi[2]}; cb.grow(c, c); }
        int ax = 0; float ext = -1;
        for (int k = 0; k < 3; k++) { const float d = cb.hi[k] - cb.lo[k]; if (d > ext) { ext = d; ax = k; } }
        const size_t m = n / 2;
        std::nth_element(refs.begin(), refs.begin() + (std::ptrdiff_t)m, refs.end(),
                         [&](const SRef &p, const SRef &q) { return 0.5f * p.lo[ax] + 0.5f * p.hi[ax] < 0.5f * q.lo[ax] + 0.5f * q.hi[ax]; });
        l.assign(refs.begin(), refs.begin() + (std::ptrdiff_t)m); r.assign(refs.begin() + (std::ptrdiff_t)m, refs.end());
    }

    static constexpr int kForkLevels = 3;          // the top three levels fork: up to eight subtrees in flight
    static constexpr size_t kForkMin = 4096;
    // Appends a sub-builder's nodes and leaf slots to this builder's; returns the sub-root's reference in this builder's numbering.
    int32_t adopt(SBuilder &sb, const std::vector<uint32_t> &rp, int32_t code) {
        const int32_t node_off = (int32_t)out.nodes.size();
        const uint32_t slot_off = (uint32_t)ref_prim->size();
        auto fix = [&](int32_t c) -> int32_t {
            if (c == GDPT_CHILD_EMPTY) return c;
            if (c >= 0) return c + node_off;
            const uint32_t packed = ~(uint32_t)c;
            return ~(int32_t)((((packed >> 2) + slot_off) << 2) | (packed & 3u));
        };
        for (DevBvhNode nd : sb.out.nodes) { nd.left = fix(nd.left); nd.right = fix(nd.right); out.nodes.push_back(nd); }
        for (uint32_t p : rp) { out.order.push_back((uint32_t)ref_prim->size()); ref_prim->push_back(p); }
        return fix(code);
    }
    int32_t build_inner(std::vector<SRef> &refs, int level, int *depth_out) {
        const int32_t me = (int32_t)out.nodes.size();
        out.nodes.emplace_back();
        const uint32_t n = (uint32_t)refs.size();
        const int remaining = GDPT_BVH_MAX_DEPTH - (level + 1);
        const int need_balanced = std::max(0, ceil_log2((n + leaf_max - 1) / leaf_max));
        // as in build_bvh: every split leaves both children fewer references than their parent, so medians forced from here on
        // finish within the levels that remain
        const bool force_median = (need_balanced >= remaining);
        std::vector<SRef> l, r;
        split(refs, force_median, l, r);
        std::vector<SRef>().swap(refs);
        const SBox lb = box_of(l), rb = box_of(r);
        int dl = 0, dr = 0;
        int32_t cl, cr;
        if (level < kForkLevels && l.size() >= kForkMin && r.size() >= kForkMin) {
            // The two subtrees are built side by side by builders of their own (the reference's Embree build is parallel too), each
            // with its share of the reference budget, and appended left then right — the node and leaf numbering a depth-first
            // build would give, whatever the threads' timing.
            SBuilder sub[2] = {SBuilder(tv, nullptr), SBuilder(tv, nullptr)};
            std::vector<uint32_t> rp[2];
            std::vector<SRef> *part[2] = {&l, &r};
            int32_t code[2] = {0, 0};
            int depth[2] = {0, 0};
            const long long share_l = (long long)((double)spare * (double)l.size() / (double)(l.size() + r.size()));
            for (int c = 0; c < 2; c++) {
                sub[c].ref_prim = &rp[c]; sub[c].root_area = root_area; sub[c].alpha = alpha; sub[c].leaf_max = leaf_max; sub[c].leaf_factor = leaf_factor;
                sub[c].spare = c == 0 ? share_l : spare - share_l;
            }
            std::exception_ptr err;
            std::thread other([&]() { try { code[1] = sub[1].build_child(*part[1], level + 1, &depth[1]); } catch (...) { err = std::current_exception(); } });
            try { code[0] = sub[0].build_child(*part[0], level + 1, &depth[0]); } catch (...) { other.join(); throw; }
            other.join();
            if (err) std::rethrow_exception(err);
            spare = sub[0].spare + sub[1].spare;
            cl = adopt(sub[0], rp[0], code[0]); cr = adopt(sub[1], rp[1], code[1]);
            dl = depth[0]; dr = depth[1];
        } else {
            cl = build_child(l, level + 1, &dl);
            cr = build_child(r, level + 1, &dr);
        }
        DevBvhNode &nd = out.nodes[(size_t)me];
        for (int k = 0; k < 3; k++) { nd.lmin[k] = lb.lo[k]; nd.lmax[k] = lb.hi[k]; nd.rmin[k] = rb.lo[k]; nd.rmax[k] = rb.hi[k]; }
        nd.left = cl; nd.right = cr; nd.pad[0] = nd.pad[1] = 0;
        *depth_out = 1 + std::max(dl, dr);
        return me;
    }

    int32_t build_child(std::vector<SRef> &refs, int level, int *depth_out) {
        const uint32_t n = (uint32_t)refs.size();
        if (n <= leaf_max) {
            bool make = true;
            if (n > 1 && level < GDPT_BVH_MAX_DEPTH) {       // the leaf test of build_bvh: split on only if it pays (object splits)
                const float fa = box_of(refs).half_area();
                const ObjSplit os = best_object_split(refs);
                if (fa > 0 && os.axis >= 0 && os.cost / fa + 1.0f < (float)n * leaf_factor) make = false;
            }
            if (make) { *depth_out = 0; return make_leaf(refs); }
        }
        return build_inner(refs, level, depth_out);
    }
};

} // namespace

BvhBuildResult build_sbvh(const std::vector<PrimBounds> &bounds, const std::vector<float> &tri_verts, double budget, std::vector<uint32_t> *ref_prim) {
    if (tri_verts.size() / 9 > bounds.size()) throw std::runtime_error("build_sbvh: more triangles than primitive boxes");
    ref_prim->clear();
    SBuilder bld(tri_verts, ref_prim);
    bld.leaf_max = (uint32_t)std::min(std::max(debug_knob_int("bvh_leaf_max", (int)bld.leaf_max), 1), GDPT_LEAF_MAX_PRIMS);
    bld.leaf_factor = (float)debug_knob("bvh_leaf_factor", (double)bld.leaf_factor);
    bld.alpha = debug_knob("sbvh_alpha", 1e-5);
    const uint32_t n = (uint32_t)bounds.size();
    if (n == 0) return std::move(bld.out);
    bld.spare = (long long)std::floor(std::max(0.0, budget) * (double)n);
    std::vector<SRef> refs(n);
    for (uint32_t i = 0; i < n; i++) { for (int k = 0; k < 3; k++) { refs[i].lo[k] = bounds[i].bmin[k]; refs[i].hi[k] = bounds[i].bmax[k]; } refs[i].prim = i; }
    bld.root_area = SBuilder::box_of(refs).half_area();
    ref_prim->reserve((size_t)n + (size_t)bld.spare);
    bld.out.order.reserve((size_t)n + (size_t)bld.spare);
    bld.out.nodes.reserve(n);
    if (n <= 1) {
        bld.out.nodes.emplace_back();
        DevBvhNode &nd = bld.out.nodes[0];
        for (int k = 0; k < 3; k++) { nd.lmin[k] = bounds[0].bmin[k]; nd.lmax[k] = bounds[0].bmax[k]; nd.rmin[k] = kInf; nd.rmax[k] = -kInf; }
        nd.left = bld.make_leaf(refs); nd.right = GDPT_CHILD_EMPTY; nd.pad[0] = nd.pad[1] = 0;
        bld.out.depth = 1;
        return std::move(bld.out);
    }
    int depth = 0;
    bld.build_inner(refs, 0, &depth);
    bld.out.depth = depth;
    return std::move(bld.out);
}

} // namespace gdpt
