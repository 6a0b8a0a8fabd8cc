// capi_host.cpp — host-only entry points of include/gdpt.h (scene ingest, image output, errors).
#include "../../include/gdpt.h"
#include "../../include/gdpt_debug.h"
#include "capi_common.h"
#include "host/bvh.h"
#include "host/image_io.h"
#include "host/scene_loader.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <map>
#include <mutex>
#include <string>

namespace gdpt {
static thread_local std::string g_last_error;
void set_last_error(const std::string &msg) { g_last_error = msg; }
} // namespace gdpt

namespace gdpt {
namespace {
std::mutex g_knob_mu;
std::map<std::string, double> g_knobs;     // test-only overrides, include/gdpt_debug.h
const char *const kKnobNames[] = {"force_eager", "log2k", "keep_frac", "search_frac", "blocks_per_cu", "no_lds_scene", "lds_wide",
                                  "no_twosided_machine", "presplit", "presplit_floor", "bvh_leaf_max", "bvh_leaf_factor", "sbvh", "sbvh_alpha", "plan_rounds", "plan_shrink", "plan_digits", "bvh_collapse_dp", "stamps", "wavefront", "wf_slots", "wf_sort", "multi_fail_band", "multi_fail_stage", "dct_bk", "dct_bm", "full_material_switch", "no_plain_kernel", "replay_per_step"};
double g_stamps[16] = {0};
} // namespace
void debug_store_stamps(const unsigned long long *v, int n) {
    std::lock_guard<std::mutex> lk(g_knob_mu);
    for (int i = 0; i < 16; i++) g_stamps[i] = i < n ? (double)v[i] : 0.0;
}
double debug_knob(const char *name, double def) {
    std::lock_guard<std::mutex> lk(g_knob_mu);
    if (g_knobs.empty()) return def;
    auto it = g_knobs.find(name);
    return it == g_knobs.end() ? def : it->second;
}
} // namespace gdpt

namespace {
std::mutex g_mu;
std::map<GdptSceneDesc *, std::unique_ptr<gdpt::HostScene>> g_descs; // desc pointer -> owner
} // namespace

extern "C" {

const char *gdpt_last_error(void) { return gdpt::g_last_error.c_str(); }

int gdpt_debug_knob_set(const char *name, double value) {
    return gdpt::guarded([&]() {
        if (!name) throw std::runtime_error("gdpt_debug_knob_set: null name");
        bool known = false;
        for (const char *k : gdpt::kKnobNames) if (std::strcmp(k, name) == 0) known = true;
        if (!known) throw std::runtime_error(std::string("gdpt_debug_knob_set: unknown knob '") + name + "'");
        std::lock_guard<std::mutex> lk(gdpt::g_knob_mu);
        gdpt::g_knobs[name] = value;
    });
}
void gdpt_debug_get_stamps(double out[16]) {
    std::lock_guard<std::mutex> lk(gdpt::g_knob_mu);
    for (int i = 0; i < 16; i++) out[i] = gdpt::g_stamps[i];
}
void gdpt_debug_knobs_reset(void) {
    std::lock_guard<std::mutex> lk(gdpt::g_knob_mu);
    gdpt::g_knobs.clear();
}

int gdpt_parse_scene(const char *xml_path, GdptSceneDesc **out_desc) { return gdpt_parse_scene_film(xml_path, 0, 0, out_desc); }

int gdpt_parse_scene_film(const char *xml_path, int film_width, int film_height, GdptSceneDesc **out_desc) {
    return gdpt::guarded([&]() {
        if (!xml_path || !out_desc) throw std::runtime_error("gdpt_parse_scene: null argument");
        if (film_width < 0 || film_height < 0 || film_width > 65536 || film_height > 65536) throw std::runtime_error("gdpt_parse_scene_film: bad film extent");
        std::unique_ptr<gdpt::HostScene> hs = gdpt::load_scene_xml(xml_path, film_width, film_height);
        GdptSceneDesc *d = &hs->desc;
        std::lock_guard<std::mutex> lk(g_mu);
        g_descs[d] = std::move(hs);
        *out_desc = d;
    });
}

void gdpt_free_scene_desc(GdptSceneDesc *desc) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_descs.erase(desc);
}

int gdpt_imwrite(const char *filename, int width, int height, const double *rgb) {
    return gdpt::guarded([&]() {
        if (!filename || !rgb || width <= 0 || height <= 0) throw std::runtime_error("gdpt_imwrite: bad argument");
        gdpt::write_image(filename, width, height, rgb);
    });
}

int gdpt_imread(const char *filename, int channels, int *width, int *height, double **texels) {
    return gdpt::guarded([&]() {
        if (!filename || !width || !height || !texels || (channels != 1 && channels != 3)) throw std::runtime_error("gdpt_imread: bad argument");
        std::vector<double> t;
        gdpt::load_texture_file(filename, channels, width, height, &t);
        double *out = (double *)std::malloc(t.size() * sizeof(double));
        if (!out) throw std::runtime_error("gdpt_imread: out of memory");
        std::memcpy(out, t.data(), t.size() * sizeof(double));
        *texels = out;
    });
}

void gdpt_image_free(double *texels) { std::free(texels); }

int gdpt_sbvh_check(const float *tri_verts9, int n, double budget, int samples_per_tri, int32_t stats[8]) {
    return gdpt::guarded([&]() {
        if (!tri_verts9 || n < 0 || !stats || samples_per_tri < 0) throw std::runtime_error("gdpt_sbvh_check: bad argument");
        std::memset(stats, 0, 8 * sizeof(int32_t));
        std::vector<float> tv(tri_verts9, tri_verts9 + 9 * (size_t)n);
        std::vector<gdpt::PrimBounds> b((size_t)n);
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 3; k++) {
                b[i].bmin[k] = std::min(tv[9 * i + k], std::min(tv[9 * i + 3 + k], tv[9 * i + 6 + k]));
                b[i].bmax[k] = std::max(tv[9 * i + k], std::max(tv[9 * i + 3 + k], tv[9 * i + 6 + k]));
            }
        std::vector<uint32_t> ref_prim;
        gdpt::BvhBuildResult r = gdpt::build_sbvh(b, tv, budget, &ref_prim);
        gdpt::WideBvh wide = gdpt::collapse_for_traversal(r.nodes, false);
        stats[0] = (int32_t)r.nodes.size(); stats[1] = r.depth; stats[2] = (int32_t)ref_prim.size();
        stats[3] = (int32_t)wide.nodes.size(); stats[4] = wide.stack_need;
        if (n == 0) return;
        if (r.depth > GDPT_BVH_MAX_DEPTH || wide.stack_need > GDPT_BVH_MAX_DEPTH) throw std::runtime_error("gdpt_sbvh_check: depth or stack bound exceeded");
        if (r.order.size() != ref_prim.size() || ref_prim.size() < (size_t)n || (double)ref_prim.size() > (1.0 + budget) * (double)n + 1.0)
            throw std::runtime_error("gdpt_sbvh_check: reference count outside [n, (1 + budget) n]");
        std::vector<int> refs_of((size_t)n, 0);
        for (uint32_t o : r.order) { if (o >= ref_prim.size() || ref_prim[o] >= (uint32_t)n) throw std::runtime_error("gdpt_sbvh_check: bad reference"); refs_of[ref_prim[o]]++; }
        for (int i = 0; i < n; i++) if (refs_of[i] < 1) throw std::runtime_error("gdpt_sbvh_check: a triangle is not referenced");
        // structure: every leaf slot used once, children's stored boxes nest (a child's box lies inside the box its parent stores for it)
        int leaves = 0;
        {
            std::vector<char> used(r.order.size(), 0);
            struct It { int32_t node; float lo[3], hi[3]; bool root; };
            std::vector<It> st; st.push_back({0, {0, 0, 0}, {0, 0, 0}, true});
            while (!st.empty()) {
                const It it = st.back(); st.pop_back();
                const DevBvhNode &nd = r.nodes[(size_t)it.node];
                const int32_t ch[2] = {nd.left, nd.right};
                const float *lo[2] = {nd.lmin, nd.rmin}, *hi[2] = {nd.lmax, nd.rmax};
                for (int c = 0; c < 2; c++) {
                    if (ch[c] == GDPT_CHILD_EMPTY) continue;
                    if (!it.root) for (int k = 0; k < 3; k++) if (!(it.lo[k] <= lo[c][k] && hi[c][k] <= it.hi[k])) throw std::runtime_error("gdpt_sbvh_check: a child box sticks out of its parent's");
                    if (ch[c] >= 0) { It nx; nx.node = ch[c]; nx.root = false; for (int k = 0; k < 3; k++) { nx.lo[k] = lo[c][k]; nx.hi[k] = hi[c][k]; } st.push_back(nx); }
                    else {
                        const unsigned packed = ~(unsigned)ch[c], first = packed >> 2, cnt = (packed & 3u) + 1u;
                        if (first + cnt > r.order.size()) throw std::runtime_error("gdpt_sbvh_check: leaf range out of bounds");
                        for (unsigned i = 0; i < cnt; i++) { if (used[first + i]) throw std::runtime_error("gdpt_sbvh_check: leaf slot used twice"); used[first + i] = 1; }
                        leaves++;
                    }
                }
            }
            for (char u : used) if (!u) throw std::runtime_error("gdpt_sbvh_check: leaf slot not reachable");
        }
        stats[5] = leaves;
        {   // surface-area cost of the collapsed tree: expected child boxes entered / triangles tested by a random line through the root box
            auto ha = [](const float *lo, const float *hi) { const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2]; return dx * dy + dy * dz + dz * dx; };
            float rlo[3] = {INFINITY, INFINITY, INFINITY}, rhi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int c = 0; c < 4; c++) if (wide.nodes[0].child[c] != GDPT_CHILD_EMPTY) for (int k = 0; k < 3; k++) { rlo[k] = std::min(rlo[k], wide.nodes[0].lo[k][c]); rhi[k] = std::max(rhi[k], wide.nodes[0].hi[k][c]); }
            const double ra = ha(rlo, rhi);
            double cn = 1.0, cp = 0.0;
            for (const DevBvh4Node &nd : wide.nodes)
                for (int c = 0; c < 4; c++) {
                    if (nd.child[c] == GDPT_CHILD_EMPTY) continue;
                    const float lo[3] = {nd.lo[0][c], nd.lo[1][c], nd.lo[2][c]}, hi[3] = {nd.hi[0][c], nd.hi[1][c], nd.hi[2][c]};
                    const double a = ra > 0 ? ha(lo, hi) / ra : 0.0;
                    if (nd.child[c] >= 0) cn += a; else cp += a * (double)(((~(unsigned)nd.child[c]) & 3u) + 1u);
                }
            stats[6] = (int32_t)std::lround(std::min(cn, 2e6) * 1000.0); stats[7] = (int32_t)std::lround(std::min(cp, 2e6) * 1000.0);
        }
        // coverage by point sampling, in the collapsed tree the kernels walk (same boxes)
        auto covered = [&](const double p[3], uint32_t prim) {
            std::vector<int32_t> st{0};
            while (!st.empty()) {
                const DevBvh4Node &nd = wide.nodes[(size_t)st.back()]; st.pop_back();
                for (int c = 0; c < 4; c++) {
                    if (nd.child[c] == GDPT_CHILD_EMPTY) continue;
                    bool in = true;
                    for (int k = 0; k < 3; k++) if (!((double)nd.lo[k][c] <= p[k] && p[k] <= (double)nd.hi[k][c])) in = false;
                    if (!in) continue;
                    if (nd.child[c] >= 0) { st.push_back(nd.child[c]); continue; }
                    const unsigned packed = ~(unsigned)nd.child[c], first = packed >> 2, cnt = (packed & 3u) + 1u;
                    for (unsigned i = 0; i < cnt; i++) if (ref_prim[r.order[first + i]] == prim) return true;
                }
            }
            return false;
        };
        for (int i = 0; i < n; i++) {
            double v[3][3];
            for (int a = 0; a < 3; a++) for (int k = 0; k < 3; k++) v[a][k] = tv[9 * (size_t)i + 3 * a + k];
            for (int s = 0; s < samples_per_tri; s++) {
                double w[3];
                if (s < 3) { w[0] = s == 0; w[1] = s == 1; w[2] = s == 2; }
                else if (s < 6) { w[0] = s == 3 ? 0 : 0.5; w[1] = s == 4 ? 0 : 0.5; w[2] = s == 5 ? 0 : 0.5; }
                else if (s == 6) { w[0] = w[1] = w[2] = 1.0 / 3.0; }
                else {      // additive-recurrence lattice folded into the triangle
                    double a = std::fmod(0.7548776662466927 * (s - 6) + 0.31 * i, 1.0), bb = std::fmod(0.5698402909980532 * (s - 6) + 0.17 * i, 1.0);
                    if (a + bb > 1.0) { a = 1.0 - a; bb = 1.0 - bb; }
                    w[0] = a; w[1] = bb; w[2] = 1.0 - a - bb;
                }
                double p[3];
                for (int k = 0; k < 3; k++) {       // inside the vertices' extent whatever the rounding of the weights
                    p[k] = w[0] * v[0][k] + w[1] * v[1][k] + w[2] * v[2][k];
                    p[k] = std::min(std::max(p[k], (double)b[i].bmin[k]), (double)b[i].bmax[k]);
                }
                if (!covered(p, (uint32_t)i)) throw std::runtime_error("gdpt_sbvh_check: a point of a triangle is in no leaf box that references the triangle");
            }
        }
    });
}

int gdpt_bvh_check(const float *bounds6, int n, int32_t stats[8]) {
    return gdpt::guarded([&]() {
        if (!bounds6 || n < 0 || !stats) throw std::runtime_error("gdpt_bvh_check: bad argument");
        std::vector<gdpt::PrimBounds> b((size_t)n);
        for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) { b[i].bmin[k] = bounds6[6 * i + k]; b[i].bmax[k] = bounds6[6 * i + 3 + k]; }
        gdpt::BvhBuildResult r = gdpt::build_bvh(b);
        gdpt::WideBvh wide = gdpt::collapse_for_traversal(r.nodes, true);
        std::memset(stats, 0, 8 * sizeof(int32_t));
        stats[0] = (int32_t)r.nodes.size(); stats[1] = r.depth;
        stats[2] = (int32_t)wide.nodes.size(); stats[3] = wide.arity; stats[4] = wide.stack_need;
        if (n == 0) return;
        if ((int)r.order.size() != n) throw std::runtime_error("gdpt_bvh_check: leaf order does not cover the primitives");
        std::vector<int> seen((size_t)n, 0);
        int leaves = 0, max_leaf = 0;
        // box of a leaf / subtree, checked against the box stored in the parent
        struct Box { float lo[3], hi[3]; };
        auto leaf_box = [&](int32_t child, std::vector<int> *mark) {
            unsigned packed = ~(unsigned)child, first = packed >> 2, cnt = (packed & 3u) + 1u;
            Box bx; for (int k = 0; k < 3; k++) { bx.lo[k] = INFINITY; bx.hi[k] = -INFINITY; }
            if (first + cnt > (unsigned)n) throw std::runtime_error("gdpt_bvh_check: leaf range out of bounds");
            for (unsigned i = 0; i < cnt; i++) {
                uint32_t p = r.order[first + i];
                if (mark) (*mark)[p]++;
                for (int k = 0; k < 3; k++) { bx.lo[k] = std::min(bx.lo[k], b[p].bmin[k]); bx.hi[k] = std::max(bx.hi[k], b[p].bmax[k]); }
            }
            if (mark) { leaves++; max_leaf = std::max(max_leaf, (int)cnt); }
            return bx;
        };
        auto inside = [](const Box &in, const float *lo, const float *hi) {
            for (int k = 0; k < 3; k++) if (!(lo[k] <= in.lo[k] && in.hi[k] <= hi[k])) return false;
            return true;
        };
        // BVH2: children have larger indices than parents? not guaranteed -> explicit post-order with a stack
        std::vector<Box> box2(r.nodes.size());
        {
            std::vector<std::pair<int32_t, int>> st{{0, 0}};
            while (!st.empty()) {
                auto [ni, phase] = st.back(); st.pop_back();
                const DevBvhNode &nd = r.nodes[(size_t)ni];
                if (phase == 0) {
                    st.push_back({ni, 1});
                    if (nd.left >= 0) st.push_back({nd.left, 0});
                    if (nd.right >= 0) st.push_back({nd.right, 0});
                } else {
                    Box acc; for (int k = 0; k < 3; k++) { acc.lo[k] = INFINITY; acc.hi[k] = -INFINITY; }
                    const int32_t ch[2] = {nd.left, nd.right};
                    const float *lo[2] = {nd.lmin, nd.rmin}, *hi[2] = {nd.lmax, nd.rmax};
                    for (int c = 0; c < 2; c++) {
                        if (ch[c] == GDPT_CHILD_EMPTY) continue;
                        Box cb = ch[c] >= 0 ? box2[(size_t)ch[c]] : leaf_box(ch[c], &seen);
                        if (!inside(cb, lo[c], hi[c])) throw std::runtime_error("gdpt_bvh_check: BVH2 child box does not enclose its subtree");
                        for (int k = 0; k < 3; k++) { acc.lo[k] = std::min(acc.lo[k], lo[c][k]); acc.hi[k] = std::max(acc.hi[k], hi[c][k]); }
                    }
                    box2[(size_t)ni] = acc;
                }
            }
        }
        for (int i = 0; i < n; i++) if (seen[i] != 1) throw std::runtime_error("gdpt_bvh_check: a primitive is not in exactly one BVH2 leaf");
        // wide form: children are emitted after their parents, so a reverse sweep sees children first
        std::vector<Box> boxw(wide.nodes.size());
        std::vector<int> seenw((size_t)n, 0);
        int leaves_keep = leaves, max_keep = max_leaf;
        for (size_t i = wide.nodes.size(); i-- > 0;) {
            const DevBvh4Node &nd = wide.nodes[i];
            Box acc; for (int k = 0; k < 3; k++) { acc.lo[k] = INFINITY; acc.hi[k] = -INFINITY; }
            int cnt = 0;
            for (int c = 0; c < 4; c++) {
                if (nd.child[c] == GDPT_CHILD_EMPTY) continue;
                cnt++;
                if (nd.child[c] >= 0 && (size_t)nd.child[c] <= i) throw std::runtime_error("gdpt_bvh_check: wide child precedes its parent");
                Box cb = nd.child[c] >= 0 ? boxw[(size_t)nd.child[c]] : leaf_box(nd.child[c], &seenw);
                float lo[3] = {nd.lo[0][c], nd.lo[1][c], nd.lo[2][c]}, hi[3] = {nd.hi[0][c], nd.hi[1][c], nd.hi[2][c]};
                if (!inside(cb, lo, hi)) throw std::runtime_error("gdpt_bvh_check: wide child box does not enclose its subtree");
                for (int k = 0; k < 3; k++) { acc.lo[k] = std::min(acc.lo[k], lo[k]); acc.hi[k] = std::max(acc.hi[k], hi[k]); }
            }
            if (cnt > wide.arity) throw std::runtime_error("gdpt_bvh_check: wide node has more children than its arity");
            boxw[i] = acc;
        }
        for (int i = 0; i < n; i++) if (seenw[i] != 1) throw std::runtime_error("gdpt_bvh_check: a primitive is not in exactly one wide leaf");
        stats[5] = leaves_keep; stats[6] = max_keep;
        // quantised BVH4 (DevBvh4QNode, what the kernels walk from HBM): same topology; the grid box of a child, evaluated exactly
        // in double, encloses the fp32 box it replaces (and with it the whole subtree)
        {
            const std::vector<DevBvh4QNode> q4 = gdpt::quantise_bvh4(wide.nodes);
            if (q4.size() != wide.nodes.size()) throw std::runtime_error("gdpt_bvh_check: quantised BVH4 has another node count");
            for (size_t i = 0; i < q4.size(); i++) {
                const DevBvh4QNode &nd = q4[i];
                for (int k = 0; k < 3; k++) if (!(nd.scale[k] >= 0x1p-60f) || !std::isfinite(nd.scale[k])) throw std::runtime_error("gdpt_bvh_check: quantised BVH4 grid step out of range");
                for (int c = 0; c < 4; c++) {
                    if (nd.child[c] != wide.nodes[i].child[c]) throw std::runtime_error("gdpt_bvh_check: quantised BVH4 child differs");
                    if (nd.child[c] == GDPT_CHILD_EMPTY) continue;
                    for (int k = 0; k < 3; k++) {
                        if (!(wide.nodes[i].lo[k][c] <= wide.nodes[i].hi[k][c])) continue;
                        const double lo = (double)nd.org[k] + (double)nd.qlo[k][c] * (double)nd.scale[k], hi = (double)nd.org[k] + (double)nd.qhi[k][c] * (double)nd.scale[k];
                        if (!(lo <= (double)wide.nodes[i].lo[k][c] && (double)wide.nodes[i].hi[k][c] <= hi)) throw std::runtime_error("gdpt_bvh_check: quantised BVH4 grid box does not enclose the fp32 box");
                    }
                }
            }
        }
        // 8-wide quantised form: the grid box of a child (evaluated exactly, in double) encloses its whole subtree
        {
            const std::vector<DevBvh8Node> &n8 = wide.nodes8;
            stats[7] = (int32_t)n8.size();
            if (wide.stack_need8 > GDPT_BVH_MAX_DEPTH + GDPT_STACK_OVERFLOW) throw std::runtime_error("gdpt_bvh_check: BVH8 stack bound exceeds the builder's maximum");
            std::vector<Box> box8(n8.size());
            std::vector<int> seen8((size_t)n, 0), need8(n8.size(), 0);
            for (size_t i = n8.size(); i-- > 0;) {
                const DevBvh8Node &nd = n8[i];
                Box acc; for (int k = 0; k < 3; k++) { acc.lo[k] = INFINITY; acc.hi[k] = -INFINITY; }
                int cnt = 0, sub = 0;
                for (int k = 0; k < 3; k++) if (!(nd.scale[k] >= 0x1p-60f) || !std::isfinite(nd.scale[k])) throw std::runtime_error("gdpt_bvh_check: BVH8 grid step out of range");
                for (int c = 0; c < 8; c++) {
                    if (nd.child[c] == GDPT_CHILD_EMPTY) continue;
                    cnt++;
                    if (nd.child[c] >= 0 && (size_t)nd.child[c] <= i) throw std::runtime_error("gdpt_bvh_check: BVH8 child precedes its parent");
                    if (nd.child[c] >= 0) sub = std::max(sub, need8[(size_t)nd.child[c]]);
                    Box cb = nd.child[c] >= 0 ? box8[(size_t)nd.child[c]] : leaf_box(nd.child[c], &seen8);
                    for (int k = 0; k < 3; k++) {
                        const double lo = (double)nd.org[k] + (double)nd.qlo[k][c] * (double)nd.scale[k], hi = (double)nd.org[k] + (double)nd.qhi[k][c] * (double)nd.scale[k];
                        if (!(lo <= (double)cb.lo[k] && (double)cb.hi[k] <= hi)) throw std::runtime_error("gdpt_bvh_check: BVH8 grid box does not enclose its subtree");
                    }
                }
                need8[i] = std::max(0, cnt - 1) + sub;
                // (a parent only has to enclose the primitives below it, not its children's grid boxes)
                for (int c = 0; c < 8; c++) {
                    if (nd.child[c] == GDPT_CHILD_EMPTY) continue;
                    Box cb = nd.child[c] >= 0 ? box8[(size_t)nd.child[c]] : leaf_box(nd.child[c], nullptr);
                    for (int k = 0; k < 3; k++) { acc.lo[k] = std::min(acc.lo[k], cb.lo[k]); acc.hi[k] = std::max(acc.hi[k], cb.hi[k]); }
                }
                box8[i] = acc;
            }
            if (!n8.empty() && need8[0] != wide.stack_need8) throw std::runtime_error("gdpt_bvh_check: BVH8 stack bound differs from the reported one");
            for (int i = 0; i < n; i++) if (seen8[i] != 1) throw std::runtime_error("gdpt_bvh_check: a primitive is not in exactly one BVH8 leaf");
            leaves = leaves_keep; max_leaf = max_keep;
        }
    });
}

} // extern "C"
