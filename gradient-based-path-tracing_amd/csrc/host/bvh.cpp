// bvh.cpp — binned-SAH BVH2 with a hard depth bound (so the GPU traversal stack, one LDS slot
// per level per lane, can never overflow).
#include "bvh.h"
#include "../capi_common.h"

#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <algorithm>
#include <cmath>
#include <limits>
#include <utility>

namespace gdpt {

namespace {

struct Box {
    float mn[3], mx[3];
    void reset() {
        for (int k = 0; k < 3; k++) { mn[k] = std::numeric_limits<float>::infinity(); mx[k] = -std::numeric_limits<float>::infinity(); }
    }
    void grow(const float *a, const float *b) {
        for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], a[k]); mx[k] = std::max(mx[k], b[k]); }
    }
    void grow(const Box &o) { grow(o.mn, o.mx); }
    void grow_pt(const float *p) { grow(p, p); }
    float half_area() const {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (!(dx >= 0 && dy >= 0 && dz >= 0)) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Builder {
    const std::vector<PrimBounds> &pb;
    std::vector<uint32_t> idx;       // working permutation
    std::vector<float> cent;         // 3 per prim
    BvhBuildResult out;

    explicit Builder(const std::vector<PrimBounds> &b) : pb(b) {}

    static int ceil_log2(uint32_t n) { int l = 0; while ((1u << l) < n) l++; return l; }

    Box range_box(uint32_t b, uint32_t e) const {
        Box bx; bx.reset();
        for (uint32_t i = b; i < e; i++) bx.grow(pb[idx[i]].bmin, pb[idx[i]].bmax);
        return bx;
    }

    // Leaf policy (build-time tuning knobs; the closest hit does not depend on the tree's shape).
    uint32_t leaf_max = GDPT_LEAF_MAX_PRIMS;
    float leaf_factor = 0.8f;

    int32_t make_leaf(uint32_t b, uint32_t e) {
        uint32_t first = (uint32_t)out.order.size();
        for (uint32_t i = b; i < e; i++) out.order.push_back(idx[i]);
        uint32_t packed = (first << 2) | (e - b - 1);
        return ~(int32_t)packed;
    }

    // Returns the split position in [b+1, e-1]; reorders idx[b,e).
    uint32_t split(uint32_t b, uint32_t e, bool force_median) {
        uint32_t n = e - b;
        Box cb; cb.reset();
        for (uint32_t i = b; i < e; i++) cb.grow_pt(&cent[3 * idx[i]]);
        int best_axis = -1; int best_bin = -1; float best_cost = std::numeric_limits<float>::infinity();
        constexpr int NB = 32;
        if (!force_median) {
            for (int ax = 0; ax < 3; ax++) {
                float lo = cb.mn[ax], hi = cb.mx[ax];
                if (!(hi > lo)) continue;
                Box bb[NB]; uint32_t cnt[NB];
                for (int k = 0; k < NB; k++) { bb[k].reset(); cnt[k] = 0; }
                float scale = NB / (hi - lo);
                for (uint32_t i = b; i < e; i++) {
                    int k = (int)((cent[3 * idx[i] + ax] - lo) * scale);
                    k = std::min(std::max(k, 0), NB - 1);
                    bb[k].grow(pb[idx[i]].bmin, pb[idx[i]].bmax); cnt[k]++;
                }
                float right_area[NB]; uint32_t right_cnt[NB];
                Box acc; acc.reset(); uint32_t c = 0;
                for (int k = NB - 1; k > 0; k--) { acc.grow(bb[k]); c += cnt[k]; right_area[k] = acc.half_area(); right_cnt[k] = c; }
                acc.reset(); c = 0;
                for (int k = 0; k < NB - 1; k++) {
                    acc.grow(bb[k]); c += cnt[k];
                    if (c == 0 || right_cnt[k + 1] == 0) continue;
                    float cost = acc.half_area() * (float)c + right_area[k + 1] * (float)right_cnt[k + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = k; }
                }
            }
        }
        if (best_axis >= 0) {
            float lo = cb.mn[best_axis], hi = cb.mx[best_axis];
            float scale = NB / (hi - lo);
            auto mid = std::partition(idx.begin() + b, idx.begin() + e, [&](uint32_t p) {
                int k = (int)((cent[3 * p + best_axis] - lo) * scale);
                k = std::min(std::max(k, 0), NB - 1);
                return k <= best_bin;
            });
            uint32_t m = (uint32_t)(mid - idx.begin());
            if (m > b && m < e) return m;
        }
        // median split along the widest centroid axis (also the depth-bounding fallback)
        int ax = 0;
        float ext = -1;
        for (int k = 0; k < 3; k++) { float d = cb.mx[k] - cb.mn[k]; if (d > ext) { ext = d; ax = k; } }
        uint32_t m = b + n / 2;
        std::nth_element(idx.begin() + b, idx.begin() + m, idx.begin() + e,
                         [&](uint32_t p, uint32_t q) { return cent[3 * p + ax] < cent[3 * q + ax]; });
        return m;
    }

    // Builds the subtree over idx[b,e) (n > LEAF_MAX or forced inner) and returns its node index.
    // `level` = number of inner nodes above this one.
    int32_t build_inner(uint32_t b, uint32_t e, int level, int *depth_out) {
        int32_t me = (int32_t)out.nodes.size();
        out.nodes.emplace_back();
        uint32_t n = e - b;
        // Levels still available below this node, and levels a balanced tree over n prims needs.
        int remaining = GDPT_BVH_MAX_DEPTH - (level + 1);
        int need_balanced = std::max(0, ceil_log2((n + leaf_max - 1) / leaf_max));
        bool force_median = (need_balanced >= remaining);
        uint32_t m = split(b, e, force_median);
        int dl = 0, dr = 0;
        int32_t cl = build_child(b, m, level + 1, &dl);
        int32_t cr = build_child(m, e, level + 1, &dr);
        Box lb = range_box(b, m), rb = range_box(m, e);
        DevBvhNode &nd = out.nodes[(size_t)me];
        for (int k = 0; k < 3; k++) { nd.lmin[k] = lb.mn[k]; nd.lmax[k] = lb.mx[k]; nd.rmin[k] = rb.mn[k]; nd.rmax[k] = rb.mx[k]; }
        nd.left = cl; nd.right = cr; nd.pad[0] = nd.pad[1] = 0;
        *depth_out = 1 + std::max(dl, dr);
        return me;
    }

    int32_t build_child(uint32_t b, uint32_t e, int level, int *depth_out) {
        uint32_t n = e - b;
        if (n <= leaf_max) {
            // SAH leaf test for small ranges: keep splitting only if it pays (cost model: 1 per prim, 1 per node)
            bool make = true;
            if (n > 1 && level < GDPT_BVH_MAX_DEPTH) {
                Box full = range_box(b, e);
                float fa = full.half_area();
                if (fa > 0) {
                    // try the best object split along the widest axis
                    std::vector<uint32_t> save(idx.begin() + b, idx.begin() + e);
                    uint32_t m = split(b, e, false);
                    float cost = (range_box(b, m).half_area() * (float)(m - b) + range_box(m, e).half_area() * (float)(e - m)) / fa + 1.0f;
                    if (cost < (float)n * leaf_factor) {
                        make = false;
                    } else {
                        std::copy(save.begin(), save.end(), idx.begin() + b);
                    }
                }
            }
            if (make) { *depth_out = 0; return make_leaf(b, e); }
        }
        return build_inner(b, e, level, depth_out);
    }
};

} // namespace

BvhBuildResult build_bvh(const std::vector<PrimBounds> &bounds) {
    Builder bld(bounds);
    // leaf policy overrides of the tree-independence test (include/gdpt_debug.h); defaults = the product path
    bld.leaf_max = (uint32_t)std::min(std::max(debug_knob_int("bvh_leaf_max", (int)bld.leaf_max), 1), GDPT_LEAF_MAX_PRIMS);
    bld.leaf_factor = (float)debug_knob("bvh_leaf_factor", (double)bld.leaf_factor);
    uint32_t n = (uint32_t)bounds.size();
    if (n == 0) return std::move(bld.out);
    bld.idx.resize(n);
    bld.cent.resize(3 * (size_t)n);
    for (uint32_t i = 0; i < n; i++) {
        bld.idx[i] = i;
        for (int k = 0; k < 3; k++) bld.cent[3 * (size_t)i + k] = 0.5f * bounds[i].bmin[k] + 0.5f * bounds[i].bmax[k];
    }
    bld.out.order.reserve(n);
    bld.out.nodes.reserve(n);
    if (n <= 1) {
        // degenerate root: one leaf on the left, nothing on the right
        bld.out.nodes.emplace_back();
        DevBvhNode &nd = bld.out.nodes[0];
        for (int k = 0; k < 3; k++) {
            nd.lmin[k] = bounds[0].bmin[k]; nd.lmax[k] = bounds[0].bmax[k];
            nd.rmin[k] = std::numeric_limits<float>::infinity(); nd.rmax[k] = -std::numeric_limits<float>::infinity();
        }
        nd.left = bld.make_leaf(0, 1); nd.right = GDPT_CHILD_EMPTY; nd.pad[0] = nd.pad[1] = 0;
        bld.out.depth = 1;
        return std::move(bld.out);
    }
    int depth = 0;
    bld.build_inner(0, n, 0, &depth);
    bld.out.depth = depth;
    return std::move(bld.out);
}

namespace {
// A node of the collapsed tree before it is written in a device format: up to 8 child boxes + child references.
struct WideTmp { float lo[3][8], hi[3][8]; int32_t child[8]; int cnt; };

std::vector<WideTmp> collapse_generic(const std::vector<DevBvhNode> &nodes, int max_children, int stack_slots, int *stack_need) {
    std::vector<WideTmp> out;
    if (stack_need) *stack_need = 0;
    if (nodes.empty()) return out;
    max_children = std::min(8, std::max(2, max_children));
    struct Ref { float lo[3], hi[3]; int32_t child; };
    auto area = [](const Ref &r) {
        float dx = r.hi[0] - r.lo[0], dy = r.hi[1] - r.lo[1], dz = r.hi[2] - r.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    auto refs_of = [&](int32_t ni, Ref *dst) {       // child refs of BVH2 node ni (EMPTY slots skipped)
        const DevBvhNode &n = nodes[(size_t)ni];
        int c = 0;
        if (n.left != GDPT_CHILD_EMPTY) { for (int k = 0; k < 3; k++) { dst[c].lo[k] = n.lmin[k]; dst[c].hi[k] = n.lmax[k]; } dst[c].child = n.left; c++; }
        if (n.right != GDPT_CHILD_EMPTY) { for (int k = 0; k < 3; k++) { dst[c].lo[k] = n.rmin[k]; dst[c].hi[k] = n.rmax[k]; } dst[c].child = n.right; c++; }
        return c;
    };
    // d2[i]: inner-node levels below and including BVH2 node i = the stack a purely binary walk of that subtree needs.
    std::vector<int> d2(nodes.size(), 0);
    {
        std::vector<std::pair<int32_t, int>> st{{0, 0}};
        while (!st.empty()) {
            auto [ni, phase] = st.back(); st.pop_back();
            const DevBvhNode &n = nodes[(size_t)ni];
            if (phase == 0) {
                st.push_back({ni, 1});
                if (n.left >= 0) st.push_back({n.left, 0});
                if (n.right >= 0) st.push_back({n.right, 0});
            } else d2[(size_t)ni] = 1 + std::max(n.left >= 0 ? d2[(size_t)n.left] : 0, n.right >= 0 ? d2[(size_t)n.right] : 0);
        }
    }
    // A node that holds cnt children pushes up to cnt-1 entries before it descends, so each child inherits the node's
    // slot budget minus cnt-1. A subtree whose binary depth fits its budget can always be finished with arity 2, so the
    // widest arity whose children all satisfy d2(child) <= budget-(cnt-1) is taken: the tree is as wide as asked wherever
    // the SAH tree is reasonably balanced and narrows only along unusually deep paths.
    struct Item { int32_t node; int budget; };
    std::vector<Item> queue{{0, stack_slots}};
    out.emplace_back();
    for (size_t qi = 0; qi < queue.size(); qi++) {
        Ref refs[8];
        int cnt = 0;
        for (int arity = max_children; arity >= 2; arity--) {
            cnt = refs_of(queue[qi].node, refs);
            while (cnt < arity) {
                int best = -1; float best_area = -1.f;
                for (int i = 0; i < cnt; i++) if (refs[i].child >= 0) { float a = area(refs[i]); if (a > best_area) { best_area = a; best = i; } }
                if (best < 0) break;
                Ref sub[2];
                int sc = refs_of(refs[best].child, sub);
                if (sc == 0) { refs[best] = refs[cnt - 1]; cnt--; continue; }
                refs[best] = sub[0];
                if (sc == 2) refs[cnt++] = sub[1];
            }
            bool fits = (cnt - 1 <= queue[qi].budget);
            for (int i = 0; i < cnt; i++) if (refs[i].child >= 0 && d2[(size_t)refs[i].child] > queue[qi].budget - (cnt - 1)) fits = false;
            if (fits) break;      // arity 2 of a subtree with d2 <= budget always fits
        }
        WideTmp nd;
        for (int c = 0; c < 8; c++) {
            for (int k = 0; k < 3; k++) { nd.lo[k][c] = std::numeric_limits<float>::infinity(); nd.hi[k][c] = -std::numeric_limits<float>::infinity(); }
            nd.child[c] = GDPT_CHILD_EMPTY;
        }
        nd.cnt = cnt;
        for (int c = 0; c < cnt; c++) {
            for (int k = 0; k < 3; k++) { nd.lo[k][c] = refs[c].lo[k]; nd.hi[k][c] = refs[c].hi[k]; }
            if (refs[c].child >= 0) {
                nd.child[c] = (int32_t)out.size();
                queue.push_back({refs[c].child, queue[qi].budget - (cnt - 1)});
                out.emplace_back();
            } else nd.child[c] = refs[c].child;
        }
        out[qi] = nd;
    }
    if (stack_need) {
        // need(node) = (children - 1) + max over inner children of need(child); children come after parents
        std::vector<int> need(out.size(), 0);
        for (size_t i = out.size(); i-- > 0;) {
            int cnt = 0, sub = 0;
            for (int c = 0; c < 8; c++) if (out[i].child[c] != GDPT_CHILD_EMPTY) { cnt++; if (out[i].child[c] >= 0) sub = std::max(sub, need[(size_t)out[i].child[c]]); }
            need[i] = std::max(0, cnt - 1) + sub;
        }
        *stack_need = need[0];
    }
    return out;
}

// Cost-optimal collapse to four children (test knob bvh_collapse_dp): of all ways to pick the BVH2 nodes that survive as BVH4 nodes,
// the one with the least summed surface area of BVH4 nodes — the expected number of node visits of a random line — among those whose
// traversal stack stays within `stack_slots`. Dynamic programme over the BVH2, bottom-up (after Ylitie, Karras and Laine, "Efficient
// incoherent ray traversal on GPUs through compressed wide BVHs", HPG 2017, section 3.1, with the stack budget as an extra index):
//   W(n, b)    = cost of subtree n as a BVH4 node that may push b entries in total below itself
//              = A(n) + min over c = 2..4 of Dist(n, c, b - (c - 1))         (c children cost c - 1 pushes; each child inherits the rest)
//   D(n, j, b) = cost of subtree n as AT MOST j children of one BVH4 node, each with budget b:  D(n, 1, b) = W(n, b),
//                D(n, j, b) = min(D(n, j - 1, b), Dist(n, j, b)),  Dist(n, j, b) = min over k of D(left, k, b) + D(right, j - k, b)
// leaves cost nothing here (their count and boxes are fixed by the BVH2).
std::vector<WideTmp> collapse_optimal4(const std::vector<DevBvhNode> &nodes, int stack_slots, int *stack_need) {
    std::vector<WideTmp> out;
    if (stack_need) *stack_need = 0;
    if (nodes.empty()) return out;
    const size_t N = nodes.size();
    const int B = stack_slots + 1;                                   // budgets 0..stack_slots
    for (const DevBvhNode &d : nodes) if (d.left == GDPT_CHILD_EMPTY || d.right == GDPT_CHILD_EMPTY) return out;   // (single-leaf tree: caller falls back)
    const float INF = std::numeric_limits<float>::infinity();
    std::vector<float> area(N, 0.f);
    std::vector<int32_t> order;                                      // post-order of the inner nodes
    {
        std::vector<std::pair<int32_t, int>> st{{0, 0}};
        while (!st.empty()) {
            auto [ni, phase] = st.back(); st.pop_back();
            const DevBvhNode &n = nodes[(size_t)ni];
            if (phase == 0) { st.push_back({ni, 1}); if (n.left >= 0) st.push_back({n.left, 0}); if (n.right >= 0) st.push_back({n.right, 0}); }
            else {
                float lo[3], hi[3];
                for (int k = 0; k < 3; k++) { lo[k] = std::min(n.lmin[k], n.rmin[k]); hi[k] = std::max(n.lmax[k], n.rmax[k]); }
                const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
                area[(size_t)ni] = dx * dy + dy * dz + dz * dx;
                order.push_back(ni);
            }
        }
    }
    // D[(n * 4 + (j - 1)) * B + b]; choice: kD = split k of Dist (0 = "use j - 1 roots"), cW = children count of W, kW = its split
    std::vector<float> D(N * 4 * (size_t)B, INF);
    std::vector<uint8_t> kD(N * 4 * (size_t)B, 0), cW(N * (size_t)B, 0), kW(N * (size_t)B, 0);
    auto Dv = [&](int32_t child, int j, int b) -> float { return child < 0 ? 0.f : D[((size_t)child * 4 + (size_t)(j - 1)) * B + b]; };
    for (int32_t ni : order) {
        const DevBvhNode &n = nodes[(size_t)ni];
        auto dist = [&](int j, int b, int *kbest) {
            float best = INF; *kbest = 0;
            for (int k = 1; k < j; k++) { const float v = Dv(n.left, k, b) + Dv(n.right, j - k, b); if (v < best) { best = v; *kbest = k; } }
            return best;
        };
        // Dist(n, j, b) for j = 2..4 depends on the children only; W(n, b) needs Dist at smaller budgets: one ascending sweep does both
        std::vector<float> distv(3 * (size_t)B, INF); std::vector<uint8_t> distk(3 * (size_t)B, 0);
        for (int b = 0; b < B; b++)
            for (int j = 2; j <= 4; j++) { int k; distv[(size_t)(j - 2) * B + b] = dist(j, b, &k); distk[(size_t)(j - 2) * B + b] = (uint8_t)k; }
        for (int b = 0; b < B; b++) {
            float w = INF; int wc = 0, wk = 0;
            for (int c = 2; c <= 4; c++) {
                const int bc = b - (c - 1);
                if (bc < 0) break;
                const float v = distv[(size_t)(c - 2) * B + bc];
                if (v < w) { w = v; wc = c; wk = distk[(size_t)(c - 2) * B + bc]; }
            }
            const size_t base = ((size_t)ni * 4) * B;
            D[base + b] = (w < INF) ? area[(size_t)ni] + w : INF;
            cW[(size_t)ni * B + b] = (uint8_t)wc; kW[(size_t)ni * B + b] = (uint8_t)wk;
            for (int j = 2; j <= 4; j++) {
                const float prev = D[base + (size_t)(j - 2) * B + b], dv = distv[(size_t)(j - 2) * B + b];
                if (dv < prev) { D[base + (size_t)(j - 1) * B + b] = dv; kD[base + (size_t)(j - 1) * B + b] = distk[(size_t)(j - 2) * B + b]; }
                else { D[base + (size_t)(j - 1) * B + b] = prev; kD[base + (size_t)(j - 1) * B + b] = 0; }
            }
        }
    }
    if (!(D[(size_t)stack_slots] < INF)) return out;                 // (the BVH2 is deeper than the stack: caller falls back)
    // top-down: a queue of (BVH2 node, budget) that become BVH4 nodes, breadth-first
    struct Item { int32_t node; int budget; };
    std::vector<Item> queue{{0, stack_slots}};
    out.emplace_back();
    struct Ref { float lo[3], hi[3]; int32_t child; };
    for (size_t qi = 0; qi < queue.size(); qi++) {
        const int32_t ni = queue[qi].node; const int b = queue[qi].budget;
        const int c = cW[(size_t)ni * B + b], bc = b - (c - 1);
        Ref refs[4]; int cnt = 0;
        // expand (subtree, j roots) pairs; `box` of a subtree comes from its parent's record
        struct Todo { int32_t sub; int j; float lo[3], hi[3]; };
        std::vector<Todo> todo;
        auto push_children = [&](int32_t p, int k, int j) {        // children of BVH2 node p: left gets k roots, right j - k
            const DevBvhNode &n = nodes[(size_t)p];
            Todo r; r.sub = n.right; r.j = j - k; for (int x = 0; x < 3; x++) { r.lo[x] = n.rmin[x]; r.hi[x] = n.rmax[x]; }
            Todo l; l.sub = n.left; l.j = k; for (int x = 0; x < 3; x++) { l.lo[x] = n.lmin[x]; l.hi[x] = n.lmax[x]; }
            todo.push_back(r); todo.push_back(l);
        };
        push_children(ni, kW[(size_t)ni * B + b], c);
        while (!todo.empty()) {
            Todo t = todo.back(); todo.pop_back();
            if (t.sub < 0) { Ref r; r.child = t.sub; for (int x = 0; x < 3; x++) { r.lo[x] = t.lo[x]; r.hi[x] = t.hi[x]; } refs[cnt++] = r; continue; }
            int j = t.j;
            while (j > 1 && kD[((size_t)t.sub * 4 + (size_t)(j - 1)) * B + bc] == 0) j--;      // "at most j": fewer roots were as cheap
            if (j == 1) { Ref r; r.child = t.sub; for (int x = 0; x < 3; x++) { r.lo[x] = t.lo[x]; r.hi[x] = t.hi[x]; } refs[cnt++] = r; continue; }
            push_children(t.sub, kD[((size_t)t.sub * 4 + (size_t)(j - 1)) * B + bc], j);
        }
        WideTmp nd;
        for (int cc = 0; cc < 8; cc++) {
            for (int k = 0; k < 3; k++) { nd.lo[k][cc] = std::numeric_limits<float>::infinity(); nd.hi[k][cc] = -std::numeric_limits<float>::infinity(); }
            nd.child[cc] = GDPT_CHILD_EMPTY;
        }
        nd.cnt = cnt;
        for (int cc = 0; cc < cnt; cc++) {
            for (int k = 0; k < 3; k++) { nd.lo[k][cc] = refs[cc].lo[k]; nd.hi[k][cc] = refs[cc].hi[k]; }
            if (refs[cc].child >= 0) { nd.child[cc] = (int32_t)out.size(); queue.push_back({refs[cc].child, bc}); out.emplace_back(); }
            else nd.child[cc] = refs[cc].child;
        }
        out[qi] = nd;
    }
    if (stack_need) {
        std::vector<int> need(out.size(), 0);
        for (size_t i = out.size(); i-- > 0;) {
            int cnt = 0, sub = 0;
            for (int c = 0; c < 8; c++) if (out[i].child[c] != GDPT_CHILD_EMPTY) { cnt++; if (out[i].child[c] >= 0) sub = std::max(sub, need[(size_t)out[i].child[c]]); }
            need[i] = std::max(0, cnt - 1) + sub;
        }
        *stack_need = need[0];
    }
    return out;
}
} // namespace

std::vector<DevBvh4Node> collapse_bvh4(const std::vector<DevBvhNode> &nodes, int max_children, int stack_slots, int *stack_need) {
    std::vector<WideTmp> tmp;
    if (max_children >= 4 && debug_knob_int("bvh_collapse_dp", 0) != 0) tmp = collapse_optimal4(nodes, stack_slots, stack_need);
    if (tmp.empty()) tmp = collapse_generic(nodes, std::min(4, max_children), stack_slots, stack_need);
    std::vector<DevBvh4Node> out(tmp.size());
    for (size_t i = 0; i < tmp.size(); i++) {
        DevBvh4Node &nd = out[i];
        for (int c = 0; c < 4; c++) {
            for (int k = 0; k < 3; k++) { nd.lo[k][c] = tmp[i].lo[k][c]; nd.hi[k][c] = tmp[i].hi[k][c]; }
            nd.child[c] = tmp[i].child[c]; nd.pad[c] = 0;
        }
    }
    return out;
}

// Quantised 8-wide form. Per node and axis a grid `org + q * scale`, q = 0..255, scale = extent / 255 of the union of
// the child boxes (rounded up to fp32); a child's lower bounds are rounded down and its upper bounds up on that grid,
// checked in double precision, so the grid box always contains the fp32 box it replaces. Boxes grow by less than
// extent / 255 per side.
std::vector<DevBvh8Node> collapse_bvh8(const std::vector<DevBvhNode> &nodes, int stack_slots, int *stack_need) {
    std::vector<WideTmp> tmp = collapse_generic(nodes, 8, stack_slots, stack_need);
    std::vector<DevBvh8Node> out(tmp.size());
    for (size_t i = 0; i < tmp.size(); i++) {
        const WideTmp &w = tmp[i];
        DevBvh8Node &nd = out[i];
        std::memset(&nd, 0, sizeof(nd));
        for (int c = 0; c < 8; c++) nd.child[c] = w.child[c];
        for (int k = 0; k < 3; k++) {
            float lo = std::numeric_limits<float>::infinity(), hi = -std::numeric_limits<float>::infinity();
            for (int c = 0; c < w.cnt; c++) if (w.lo[k][c] <= w.hi[k][c]) { lo = std::min(lo, w.lo[k][c]); hi = std::max(hi, w.hi[k][c]); }
            if (!(lo <= hi)) { lo = 0.f; hi = 0.f; }
            const double ext = (double)hi - (double)lo;
            // grid step: extent / 255 rounded up to fp32 (floor 2^-60: keeps step / d a normal number)
            float step = (float)(ext / 255.0);
            while ((double)step * 255.0 < ext) step = std::nextafterf(step, std::numeric_limits<float>::infinity());
            step = std::max(step, 0x1p-60f);
            for (int tries = 0;; tries++) {
                const double sc = (double)step;
                bool ok = true;
                for (int c = 0; c < 8 && ok; c++) {
                    if (c >= w.cnt || !(w.lo[k][c] <= w.hi[k][c])) { nd.qlo[k][c] = 255; nd.qhi[k][c] = 0; continue; }   // (empty slot: never read as a box, the child id decides)
                    double ql = std::floor(((double)w.lo[k][c] - (double)lo) / sc), qh = std::ceil(((double)w.hi[k][c] - (double)lo) / sc);
                    while (ql > 0 && (double)lo + ql * sc > (double)w.lo[k][c]) ql -= 1;
                    while ((double)lo + qh * sc < (double)w.hi[k][c]) qh += 1;
                    if (ql < 0) ql = 0;
                    if (qh > 255) { ok = false; break; }
                    nd.qlo[k][c] = (uint8_t)ql; nd.qhi[k][c] = (uint8_t)qh;
                }
                if (ok) { nd.org[k] = lo; nd.scale[k] = step; break; }
                if (tries > 64 || !std::isfinite(step)) throw std::runtime_error("collapse_bvh8: box extent out of range");
                step *= 1.0001f;
            }
        }
    }
    return out;
}

std::vector<DevBvh4QNode> quantise_bvh4(const std::vector<DevBvh4Node> &nodes) {
    static_assert(sizeof(DevBvh4QNode) == 64, "quantised BVH4 node is half a 128-byte line");
    std::vector<DevBvh4QNode> out(nodes.size());
    for (size_t i = 0; i < nodes.size(); i++) {
        const DevBvh4Node &w = nodes[i];
        DevBvh4QNode &nd = out[i];
        std::memset(&nd, 0, sizeof(nd));
        for (int c = 0; c < 4; c++) nd.child[c] = w.child[c];
        for (int k = 0; k < 3; k++) {
            float lo = std::numeric_limits<float>::infinity(), hi = -std::numeric_limits<float>::infinity();
            for (int c = 0; c < 4; c++) if (w.child[c] != GDPT_CHILD_EMPTY && w.lo[k][c] <= w.hi[k][c]) { lo = std::min(lo, w.lo[k][c]); hi = std::max(hi, w.hi[k][c]); }
            if (!(lo <= hi)) { lo = 0.f; hi = 0.f; }
            const double ext = (double)hi - (double)lo;
            float step = (float)(ext / 255.0);        // grid step: extent / 255 rounded up to fp32 (floor 2^-60: keeps step / d a normal number)
            while ((double)step * 255.0 < ext) step = std::nextafterf(step, std::numeric_limits<float>::infinity());
            step = std::max(step, 0x1p-60f);
            for (int tries = 0;; tries++) {
                const double sc = (double)step;
                bool ok = true;
                for (int c = 0; c < 4 && ok; c++) {
                    if (w.child[c] == GDPT_CHILD_EMPTY || !(w.lo[k][c] <= w.hi[k][c])) { nd.qlo[k][c] = 255; nd.qhi[k][c] = 0; continue; }   // (never read as a box: the child id decides)
                    double ql = std::floor(((double)w.lo[k][c] - (double)lo) / sc), qh = std::ceil(((double)w.hi[k][c] - (double)lo) / sc);
                    while (ql > 0 && (double)lo + ql * sc > (double)w.lo[k][c]) ql -= 1;
                    while ((double)lo + qh * sc < (double)w.hi[k][c]) qh += 1;
                    if (ql < 0) ql = 0;
                    if (qh > 255) { ok = false; break; }
                    nd.qlo[k][c] = (uint8_t)ql; nd.qhi[k][c] = (uint8_t)qh;
                }
                if (ok) { nd.org[k] = lo; nd.scale[k] = step; break; }
                if (tries > 64 || !std::isfinite(step)) throw std::runtime_error("quantise_bvh4: box extent out of range");
                step *= 1.0001f;
            }
        }
    }
    return out;
}

WideBvh collapse_for_traversal(const std::vector<DevBvhNode> &nodes, bool with_bvh8) {
    WideBvh w;
    w.nodes = collapse_bvh4(nodes, 4, GDPT_BVH_MAX_DEPTH, &w.stack_need);
    if (with_bvh8) w.nodes8 = collapse_bvh8(nodes, GDPT_BVH_MAX_DEPTH + GDPT_STACK_OVERFLOW, &w.stack_need8);
    for (const DevBvh4Node &n : w.nodes) {
        int cnt = 0;
        for (int c = 0; c < 4; c++) cnt += (n.child[c] != GDPT_CHILD_EMPTY);
        w.arity = std::max(w.arity, cnt);
    }
    return w;
}

} // namespace gdpt
