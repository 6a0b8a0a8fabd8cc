// presplit.cpp — early split clipping of large triangles ahead of the SAH build (see bvh.h).
#include "bvh.h"
#include "../capi_common.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <queue>

namespace gdpt {
namespace {

struct Poly { double p[10][3]; int n; };

struct Ref {
    Poly poly;
    float lo[3], hi[3];
    uint32_t prim;
    float key;              // half area of the box: the largest boxes are split first (a priority by the box area
                            // the triangle cannot fill was measured too: slower on sponza, where the win is in the
                            // large axis-aligned floor and wall triangles)
};

float half_area(const float *lo, const float *hi) {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// Sutherland-Hodgman against one axis-aligned half space: keeps sign*(x[axis] - pos) <= 0.
Poly clip(const Poly &in, int axis, double pos, double sign) {
    Poly out; out.n = 0;
    for (int i = 0; i < in.n; i++) {
        const double *a = in.p[i], *b = in.p[(i + 1) % in.n];
        double da = sign * (a[axis] - pos), db = sign * (b[axis] - pos);
        if (da <= 0) { if (out.n < 10) { for (int k = 0; k < 3; k++) out.p[out.n][k] = a[k]; out.n++; } }
        if ((da < 0 && db > 0) || (da > 0 && db < 0)) {
            double t = da / (da - db);
            if (out.n < 10) {
                for (int k = 0; k < 3; k++) out.p[out.n][k] = a[k] + t * (b[k] - a[k]);
                out.p[out.n][axis] = pos;
                out.n++;
            }
        }
    }
    return out;
}

// Box of a clipped piece: polygon extent rounded outward to float, then cut back to the parent's box (the piece lies
// inside it by construction; the rounding must not grow past it).
void piece_box(const Poly &pl, const float *plo, const float *phi, float *lo, float *hi) {
    for (int k = 0; k < 3; k++) {
        double mn = std::numeric_limits<double>::infinity(), mx = -mn;
        for (int i = 0; i < pl.n; i++) { mn = std::min(mn, pl.p[i][k]); mx = std::max(mx, pl.p[i][k]); }
        float l = (float)mn, h = (float)mx;
        if ((double)l > mn) l = std::nextafterf(l, -std::numeric_limits<float>::infinity());
        if ((double)h < mx) h = std::nextafterf(h, std::numeric_limits<float>::infinity());
        // one more ulp: the interpolated clip vertices carry rounding of their own
        l = std::nextafterf(l, -std::numeric_limits<float>::infinity());
        h = std::nextafterf(h, std::numeric_limits<float>::infinity());
        lo[k] = std::max(l, plo[k]); hi[k] = std::min(h, phi[k]);
    }
}

struct ByKey { bool operator()(const Ref *a, const Ref *b) const { return a->key < b->key; } };

} // namespace

void presplit_triangles(const std::vector<PrimBounds> &bounds, const std::vector<float> &tri_verts, double budget,
                        std::vector<PrimBounds> *refs, std::vector<uint32_t> *ref_prim) {
    const size_t n = bounds.size(), ntri = tri_verts.size() / 9;
    refs->assign(bounds.begin(), bounds.end());
    ref_prim->resize(n);
    for (size_t i = 0; i < n; i++) (*ref_prim)[i] = (uint32_t)i;
    if (budget <= 0 || ntri == 0) return;
    size_t extra = (size_t)std::floor(budget * (double)n);
    if (extra == 0) return;

    float slo[3], shi[3];
    for (int k = 0; k < 3; k++) { slo[k] = std::numeric_limits<float>::infinity(); shi[k] = -slo[k]; }
    for (const PrimBounds &b : bounds) for (int k = 0; k < 3; k++) { slo[k] = std::min(slo[k], b.bmin[k]); shi[k] = std::max(shi[k], b.bmax[k]); }
    // boxes below this size are left alone however much budget is left
    const float floor_key = half_area(slo, shi) * (float)debug_knob("presplit_floor", 1e-6);

    // candidates: the (budget-limited) largest triangle boxes
    std::vector<Ref *> pool;
    std::priority_queue<Ref *, std::vector<Ref *>, ByKey> heap;
    std::vector<size_t> cand;
    for (size_t i = 0; i < ntri; i++) if (half_area(bounds[i].bmin, bounds[i].bmax) > floor_key) cand.push_back(i);
    if (cand.size() > extra) {
        std::nth_element(cand.begin(), cand.begin() + (std::ptrdiff_t)extra, cand.end(), [&](size_t a, size_t b) {
            return half_area(bounds[a].bmin, bounds[a].bmax) > half_area(bounds[b].bmin, bounds[b].bmax); });
        cand.resize(extra);
    }
    std::vector<char> taken(n, 0);
    for (size_t i : cand) {
        Ref *r = new Ref; pool.push_back(r);
        r->poly.n = 3;
        for (int v = 0; v < 3; v++) for (int k = 0; k < 3; k++) r->poly.p[v][k] = tri_verts[9 * i + 3 * v + k];
        for (int k = 0; k < 3; k++) { r->lo[k] = bounds[i].bmin[k]; r->hi[k] = bounds[i].bmax[k]; }
        r->prim = (uint32_t)i; r->key = half_area(r->lo, r->hi);
        heap.push(r); taken[i] = 1;
    }
    std::vector<Ref *> done;
    while (!heap.empty() && extra > 0) {
        Ref *r = heap.top(); heap.pop();
        if (r->key <= floor_key) { done.push_back(r); continue; }
        int axis = 0;
        for (int k = 1; k < 3; k++) if (r->hi[k] - r->lo[k] > r->hi[axis] - r->lo[axis]) axis = k;
        const double pos = 0.5 * ((double)r->lo[axis] + (double)r->hi[axis]);
        Poly a = clip(r->poly, axis, pos, 1.0), b = clip(r->poly, axis, pos, -1.0);
        if (a.n < 3 || b.n < 3 || !(pos > r->lo[axis] && pos < r->hi[axis])) { r->key = 0; done.push_back(r); continue; }
        Ref *ra = new Ref, *rb = new Ref; pool.push_back(ra); pool.push_back(rb);
        ra->poly = a; rb->poly = b; ra->prim = rb->prim = r->prim;
        piece_box(a, r->lo, r->hi, ra->lo, ra->hi); piece_box(b, r->lo, r->hi, rb->lo, rb->hi);
        ra->key = half_area(ra->lo, ra->hi); rb->key = half_area(rb->lo, rb->hi);
        heap.push(ra); heap.push(rb);
        r->prim = UINT32_MAX;                  // replaced by its two pieces
        extra--;
    }
    while (!heap.empty()) { done.push_back(heap.top()); heap.pop(); }

    refs->clear(); ref_prim->clear();
    for (size_t i = 0; i < n; i++) if (!taken[i]) { refs->push_back(bounds[i]); ref_prim->push_back((uint32_t)i); }
    for (Ref *r : done) {
        if (r->prim == UINT32_MAX) continue;
        PrimBounds pb;
        for (int k = 0; k < 3; k++) { pb.bmin[k] = r->lo[k]; pb.bmax[k] = r->hi[k]; }
        refs->push_back(pb); ref_prim->push_back(r->prim);
    }
    for (Ref *r : pool) delete r;
}

} // namespace gdpt
