// oracle.cpp — CPU restatement of LaJolla's Integrator::GradPath path.
//
// TEST INFRASTRUCTURE ONLY (see oracle.h). Plain C++17, fp64 shading, fp32 ray/triangle arithmetic,
// compiled with -ffp-contract=off so every operation rounds once, as the reference built for x86-64 does.
// Each block cites the reference lines it restates (paths relative to /root/reference).
//
// Pinning (details in oracle/README.md and DESIGN.md):
//   * PCG32, sample_primary + filters, every BSDF's eval/pdf/sample, triangle compute_shading_info, texture
//     lookups: checked against vectors produced by the reference's OWN sources compiled here
//     (oracle/ref_kat.cpp -> tests/golden/ref_kat.json).
//   * Ray/primitive intersection: Embree 4.3.0 is absent from the reference tree (.MISSING_LARGE_BLOBS:10-11),
//     so closest-hit arithmetic is this file's own definition (fp32 Moller-Trumbore, ties -> lowest
//     primitive id); only src/tests/intersection.cpp:28-38 pins it (one hit, 1e-3).
//   * grad_path_tracing's control flow cannot be executed from the reference (it calls Embree): restated from
//     source with the "A-semantics" of SURVEY.md §8(a) G2 for the four undefined reads at
//     src/path_tracing.h:1007-1010, and cross-checked against the integer statistics the survey recorded
//     from a shim-linked reference run.
#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

using Real = double;
constexpr Real c_PI = 3.14159265358979323846;   // src/lajolla.h:25
constexpr Real c_TWOPI = 2.0 * c_PI;

struct V2 { Real x, y; };
struct V3 {
    Real x, y, z;
    Real &operator[](int i) { return (&x)[i]; }
    const Real &operator[](int i) const { return (&x)[i]; }
};
inline V3 operator+(const V3 &a, const V3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(const V3 &a, const V3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(const V3 &a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(const V3 &a, Real s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(Real s, const V3 &a) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(const V3 &a, const V3 &b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator/(const V3 &a, Real s) { Real inv = Real(1) / s; return {a.x * inv, a.y * inv, a.z * inv}; } // src/vector.h:194-197
inline V3 operator+(const V3 &a, Real s) { return {a.x + s, a.y + s, a.z + s}; }
inline V3 operator+(Real s, const V3 &a) { return {a.x + s, a.y + s, a.z + s}; }
inline V3 operator-(Real s, const V3 &a) { return {s - a.x, s - a.y, s - a.z}; }
inline V3 operator-(const V3 &a, Real s) { return {a.x - s, a.y - s, a.z - s}; }
inline Real dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(const V3 &a, const V3 &b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline Real length(const V3 &a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(const V3 &a) { Real l = length(a); if (l <= 0) return {0, 0, 0}; return a / l; } // src/vector.h:250-257
inline Real distance_squared(const V3 &a, const V3 &b) { return dot(a - b, a - b); }
inline Real maxc(const V3 &a) { return std::max(std::max(a.x, a.y), a.z); }
inline V3 splat(Real v) { return {v, v, v}; }
inline Real luminance(const V3 &s) { return s.x * 0.212671 + s.y * 0.715160 + s.z * 0.072169; } // src/spectrum.h:33-35
template <class T> inline T rmax(T a, T b) { return a > b ? a : b; } // src/lajolla.h:57-65
template <class T> inline T rmin(T a, T b) { return a < b ? a : b; }
inline Real modulo(Real a, Real b) { Real r = std::fmod(a, b); return (r < 0.0) ? r + b : r; } // src/lajolla.h:52-55
inline int modulo(int a, int b) { int r = a % b; return (r < 0) ? r + b : r; }

// ---- frame, src/frame.h ------------------------------------------------------------------------
struct Frame { V3 x, y, n; };
inline void coordinate_system(const V3 &n, V3 &a, V3 &b) { // src/frame.h:11-22
    if (n.z < Real(-1 + 1e-6)) { a = V3{0, -1, 0}; b = V3{-1, 0, 0}; }
    else {
        Real aa = 1 / (1 + n.z);
        Real bb = -n.x * n.y * aa;
        a = V3{1 - n.x * n.x * aa, bb, -n.x};
        b = V3{bb, 1 - n.y * n.y * aa, -n.y};
    }
}
inline Frame make_frame(const V3 &n) { Frame f; f.n = n; coordinate_system(n, f.x, f.y); return f; }
inline Frame neg(const Frame &f) { return Frame{-f.x, -f.y, -f.n}; }
inline V3 to_local(const Frame &f, const V3 &v) { return {dot(v, f.x), dot(v, f.y), dot(v, f.n)}; }
inline V3 to_world(const Frame &f, const V3 &v) { return f.x * v.x + f.y * v.y + f.n * v.z; }

// ---- PCG32, src/pcg.h:18-66 ---------------------------------------------------------------------
struct Pcg { uint64_t state, inc; };
inline uint32_t pcg_next(Pcg &r) {
    uint64_t old = r.state;
    r.state = old * 6364136223846793005ULL + (r.inc | 1);
    uint32_t xorshifted = uint32_t(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = uint32_t(old >> 59u);
    return uint32_t((xorshifted >> rot) | (xorshifted << ((-rot) & 31)));
}
inline Pcg pcg_init(uint64_t stream, uint64_t seed = 0x31e241f862a1fb5eULL) {
    Pcg s; s.state = 0U; s.inc = (stream << 1u) | 1u;
    pcg_next(s); s.state += seed; pcg_next(s);
    return s;
}
inline double pcg_real(Pcg &r) {
    union { uint64_t u; double d; } x;
    x.u = ((uint64_t)pcg_next(r) << 20) | 0x3ff0000000000000ULL;
    return x.d - 1.0;
}

// ---- scene --------------------------------------------------------------------------------------
struct Vertex { // PathVertex, src/intersection.h:15-37
    V3 position, geometric_normal;
    Frame shading_frame;
    V2 st, uv;
    Real uv_screen_size = 0, mean_curvature = 0, ray_radius = 0;
    int shape_id = -1, primitive_id = -1, material_id = -1, gid = -1;
    Real t = 0;
};

struct Tri {            // fp32 traversal copy (src/shapes/triangle_mesh.inl:11-14) + ids
    float v0[3], e1[3], e2[3];
    int shape_id, prim_id;
};
struct Mip { int channels = 0; std::vector<int> w, h; std::vector<std::vector<double>> lv; };

struct BNode { float mn[3], mx[3]; int left, right, first, count; };

} // namespace

struct OracleScene {
    GdptSceneDesc desc;
    std::vector<Tri> tris;           // gid order: shapes in order, triangles in order
    std::vector<int> sphere_shapes;  // gid = tris.size() + k
    std::vector<Mip> mips;
    Real isect_eps = 0;
    bool use_bvh = false;
    std::vector<BNode> nodes;        // oracle's own BVH (median split) over prim ids
    std::vector<int> bvh_prims;
    mutable std::atomic<uint64_t> nodes_visited{0}, tris_tested{0};
    // Integrator::Path: light selection table (src/scene.cpp:44-53) and per-mesh triangle tables of the emitters
    // (init_sampling_dist, src/shapes/triangle_mesh.inl:60-75)
    std::vector<Real> light_pmf, light_cdf;
    struct MeshTable { std::vector<Real> pmf, cdf; Real total_area = 0; };
    std::vector<MeshTable> mesh_tables;     // per shape (empty for non-emitters and spheres)
    // environment map (src/lights/envmap.inl): TableDist2D over luminance * sin(elevation) of the level-0 image
    struct Table2D { std::vector<Real> cdf_rows, pdf_rows, cdf_marginals, pdf_marginals; Real total_values = 0; int width = 0, height = 0; };
    Table2D env_table;
    Real bounds_radius = 0;                 // scene.bounds.radius (src/scene.cpp:29-34)
};

namespace {

// ---- fp32 ray/triangle (normative definition shared with the HIP kernel, bit for bit) -----------
// Moller-Trumbore, two-sided, every product and sum rounded separately (no FMA), left-to-right sums.
inline bool tri_hit(const float o[3], const float d[3], float tnear, float tfar, const Tri &tr, float *t_out, float *u_out, float *v_out) {
    const float *e1 = tr.e1, *e2 = tr.e2;
    float px = d[1] * e2[2] - d[2] * e2[1];
    float py = d[2] * e2[0] - d[0] * e2[2];
    float pz = d[0] * e2[1] - d[1] * e2[0];
    float det = e1[0] * px + e1[1] * py + e1[2] * pz;
    if (!(det != 0.0f)) return false;
    float inv = 1.0f / det;
    float sx = o[0] - tr.v0[0], sy = o[1] - tr.v0[1], sz = o[2] - tr.v0[2];
    float u = (sx * px + sy * py + sz * pz) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    float qx = sy * e1[2] - sz * e1[1];
    float qy = sz * e1[0] - sx * e1[2];
    float qz = sx * e1[1] - sy * e1[0];
    float v = (d[0] * qx + d[1] * qy + d[2] * qz) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    float t = (e2[0] * qx + e2[1] * qy + e2[2] * qz) * inv;
    if (!(t >= tnear && t < tfar)) return false;
    *t_out = t; *u_out = u; *v_out = v;
    return true;
}

// Sphere primitive: fp64 quadratic on the fp32 ray, src/shapes/sphere.inl:15-106.
inline bool solve_quadratic(Real a, Real b, Real c, Real *t0, Real *t1) {
    if (a == 0) { if (b == 0) return false; *t0 = *t1 = -c / b; return true; }
    Real disc = b * b - 4 * a * c;
    if (disc < 0) return false;
    Real rd = std::sqrt(disc);
    if (b >= 0) { *t0 = (-b - rd) / (2 * a); *t1 = 2 * c / (-b - rd); }
    else { *t0 = 2 * c / (-b + rd); *t1 = (-b + rd) / (2 * a); }
    return true;
}
struct SphereHit { float t, u, v, ng[3]; };
inline bool sphere_hit(const float o[3], const float d[3], float tnear, float tfar, const GdptShape &sp, SphereHit *h) {
    V3 org{o[0], o[1], o[2]}, dir{d[0], d[1], d[2]}, c{sp.center[0], sp.center[1], sp.center[2]};
    Real rtnear = tnear, rtfar = tfar;
    V3 v = org - c;
    Real A = dot(dir, dir), B = 2 * dot(dir, v), C = dot(v, v) - sp.radius * sp.radius;
    Real t0, t1;
    if (!solve_quadratic(A, B, C, &t0, &t1)) return false;
    if (t0 > t1) std::swap(t0, t1);
    Real t = -1;
    if (t0 >= rtnear && t0 < rtfar) t = t0;
    if (t1 >= rtnear && t1 < rtfar && t < 0) t = t1;
    if (!(t >= rtnear && t < rtfar)) return false;
    V3 p = org + t * dir;
    V3 gn = p - c;
    V3 cart = gn / sp.radius;
    Real elevation = std::acos(std::clamp(cart.y, Real(-1), Real(1)));
    Real azimuth = std::atan2(cart.z, cart.x);
    h->ng[0] = (float)gn.x; h->ng[1] = (float)gn.y; h->ng[2] = (float)gn.z;
    h->u = (float)(azimuth / c_TWOPI); h->v = (float)(elevation / c_PI);
    h->t = (float)t;
    return true;
}

struct Hit { bool valid = false; float t = 0, u = 0, v = 0, ng[3] = {0, 0, 0}; int gid = -1; };

inline void test_prim(const OracleScene &sc, int gid, const float o[3], const float d[3], float tnear, float tfar, Hit &best) {
    int ntri = (int)sc.tris.size();
    if (gid < ntri) {
        float t, u, v;
        // candidates beyond the current best are irrelevant; equal t is kept for the id tie-break
        if (!tri_hit(o, d, tnear, tfar, sc.tris[gid], &t, &u, &v)) return;
        if (best.valid && !(t < best.t || (t == best.t && gid < best.gid))) return;
        const Tri &tr = sc.tris[gid];
        best.valid = true; best.t = t; best.u = u; best.v = v; best.gid = gid;
        // Ng = e1 x e2 in fp32 (Embree reports an unnormalised fp32 Ng), one rounding per op
        best.ng[0] = tr.e1[1] * tr.e2[2] - tr.e1[2] * tr.e2[1];
        best.ng[1] = tr.e1[2] * tr.e2[0] - tr.e1[0] * tr.e2[2];
        best.ng[2] = tr.e1[0] * tr.e2[1] - tr.e1[1] * tr.e2[0];
    } else {
        const GdptShape &sp = sc.desc.shapes[sc.sphere_shapes[gid - ntri]];
        SphereHit h;
        // the callback sees the ray's current tfar (src/shapes/sphere.inl:55-57); using the best hit so far
        // or the original tfar selects the same minimum
        if (!sphere_hit(o, d, tnear, tfar, sp, &h)) return;
        if (best.valid && !(h.t < best.t || (h.t == best.t && gid < best.gid))) return;
        best.valid = true; best.t = h.t; best.u = h.u; best.v = h.v; best.gid = gid;
        best.ng[0] = h.ng[0]; best.ng[1] = h.ng[1]; best.ng[2] = h.ng[2];
    }
}

// conservative slab test (never rejects a box whose primitive can produce t <= tbest)
inline bool box_hit(const float mn[3], const float mx[3], const float o[3], const float inv[3], float tnear, float tbest) {
    float t0 = tnear, t1 = tbest;
    for (int k = 0; k < 3; k++) {
        float a = (mn[k] - o[k]) * inv[k], b = (mx[k] - o[k]) * inv[k];
        float lo = std::fmin(a, b), hi = std::fmax(a, b); // fmin/fmax drop NaNs (0*inf on flat boxes)
        lo = lo - std::fabs(lo) * 4e-7f; hi = hi + std::fabs(hi) * 4e-7f; // widen by ~3 ulp
        t0 = std::fmax(t0, lo); t1 = std::fmin(t1, hi);
    }
    return t0 <= t1;
}

Hit closest_hit(const OracleScene &sc, const float o[3], const float d[3], float tnear, float tfar) {
    Hit best;
    int nprim = (int)sc.tris.size() + (int)sc.sphere_shapes.size();
    uint64_t nv = 0, nt = 0;
    if (!sc.use_bvh) {
        for (int g = 0; g < nprim; g++) test_prim(sc, g, o, d, tnear, tfar, best);
        nt = (uint64_t)nprim;
    } else if (!sc.nodes.empty()) {
        float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
        int stack[128]; int sp = 0; stack[sp++] = 0;
        while (sp) {
            const BNode &n = sc.nodes[stack[--sp]];
            nv++;
            float tb = best.valid ? best.t : tfar;
            if (!box_hit(n.mn, n.mx, o, inv, tnear, tb)) continue;
            if (n.count > 0) {
                for (int i = 0; i < n.count; i++) { test_prim(sc, sc.bvh_prims[n.first + i], o, d, tnear, tfar, best); nt++; }
            } else { stack[sp++] = n.left; stack[sp++] = n.right; }
        }
    }
    sc.nodes_visited.fetch_add(nv, std::memory_order_relaxed);
    sc.tris_tested.fetch_add(nt, std::memory_order_relaxed);
    return best;
}

// ---- textures, src/texture.h:112-159, src/mipmap.h ----------------------------------------------
V3 mip_lookup_level(const Mip &m, Real u, Real v, int level) { // src/mipmap.h:51-72
    int w = m.w[level], h = m.h[level];
    const std::vector<double> &img = m.lv[level];
    u = u * w - Real(0.5);
    v = v * h - Real(0.5);
    int ufi = modulo(int(u), w), vfi = modulo(int(v), h);
    int uci = modulo(ufi + 1, w), vci = modulo(vfi + 1, h);
    Real u_off = u - ufi, v_off = v - vfi;
    auto at = [&](int x, int y) {
        if (m.channels == 1) { Real t = img[(size_t)y * w + x]; return V3{t, t, t}; }
        const double *p = &img[((size_t)y * w + x) * 3];
        return V3{p[0], p[1], p[2]};
    };
    V3 ff = at(ufi, vfi), fc = at(ufi, vci), cf = at(uci, vfi), cc = at(uci, vci);
    return ff * (1 - u_off) * (1 - v_off) + fc * (1 - u_off) * v_off + cf * u_off * (1 - v_off) + cc * u_off * v_off;
}
V3 mip_lookup(const Mip &m, Real u, Real v, Real level) { // src/mipmap.h:74-88
    int n = (int)m.lv.size();
    if (level <= 0) return mip_lookup_level(m, u, v, 0);
    if (level < Real(n - 1)) {
        int fl = std::clamp((int)std::floor(level), 0, n - 1);
        int cl = std::clamp(fl + 1, 0, n - 1);
        Real off = level - fl;
        return mip_lookup_level(m, u, v, fl) * (1 - off) + mip_lookup_level(m, u, v, cl) * off;
    }
    return mip_lookup_level(m, u, v, n - 1);
}
Mip make_mip(const GdptImage &im) { // src/mipmap.h:27-48
    Mip m; m.channels = im.channels;
    int size = std::max(im.width, im.height);
    int num_levels = std::min((int)std::ceil(std::log2(Real(size)) + 1), 8);
    m.w.push_back(im.width); m.h.push_back(im.height);
    m.lv.emplace_back(im.texels, im.texels + (size_t)im.width * im.height * im.channels);
    for (int i = 1; i < num_levels; i++) {
        int pw = m.w.back(), ph = m.h.back();
        int nw = std::max(pw / 2, 1), nh = std::max(ph / 2, 1);
        const std::vector<double> &prev = m.lv.back();
        std::vector<double> next((size_t)nw * nh * im.channels);
        // note: like the reference, reads (2x+1, 2y+1) without clamping; sizes that are not powers of two
        // with an odd dimension of 1 would read out of range there — guard by clamping (never hit by the scenes)
        auto P = [&](int x, int y, int c) { x = std::min(x, pw - 1); y = std::min(y, ph - 1); return prev[((size_t)y * pw + x) * im.channels + c]; };
        for (int y = 0; y < nh; y++) for (int x = 0; x < nw; x++) for (int c = 0; c < im.channels; c++)
            next[((size_t)y * nw + x) * im.channels + c] =
                (P(2 * x, 2 * y, c) + P(2 * x + 1, 2 * y, c) + P(2 * x, 2 * y + 1, c) + P(2 * x + 1, 2 * y + 1, c)) / Real(4);
        m.w.push_back(nw); m.h.push_back(nh); m.lv.push_back(std::move(next));
    }
    return m;
}

V3 tex_eval3(const OracleScene &sc, const GdptTexture &t, const V2 &uv, Real footprint) {
    if (t.type == GDPT_TEX_CONSTANT) return V3{t.v0[0], t.v0[1], t.v0[2]};
    V2 luv{modulo(uv.x * t.uscale + t.uoffset, Real(1)), modulo(uv.y * t.vscale + t.voffset, Real(1))};
    if (t.type == GDPT_TEX_IMAGE) {
        const Mip &m = sc.mips[t.image_id];
        Real scaled = rmax(m.w[0], m.h[0]) * rmax(t.uscale, t.vscale) * footprint;
        Real level = std::log2(rmax(scaled, Real(1e-8f)));
        return mip_lookup(m, luv.x, luv.y, level);
    }
    int x = 2 * modulo((int)(luv.x * 2), 2) - 1, y = 2 * modulo((int)(luv.y * 2), 2) - 1;
    if (x * y == 1) return V3{t.v0[0], t.v0[1], t.v0[2]};
    return V3{t.v1[0], t.v1[1], t.v1[2]};
}
inline Real tex_eval1(const OracleScene &sc, const GdptTexture &t, const V2 &uv, Real footprint) { return tex_eval3(sc, t, uv, footprint).x; }

// ---- camera + filters, src/camera.cpp:23-47, src/filters/*.inl ----------------------------------
V2 filter_sample(int type, Real param, const V2 &r) {
    if (type == GDPT_FILTER_BOX) return V2{(2 * r.x - 1) * (param / 2), (2 * r.y - 1) * (param / 2)};
    if (type == GDPT_FILTER_GAUSSIAN) {
        Real rr = param * std::sqrt(-2 * std::log(rmax(r.x, Real(1e-8))));
        return V2{rr * std::cos(2 * c_PI * r.y), rr * std::sin(2 * c_PI * r.y)};
    }
    Real h = param / 2;
    Real x = r.x < 0.5 ? h * (std::sqrt(2 * r.x) - 1) : h * (1 - std::sqrt(1 - 2 * (r.x - Real(0.5))));
    Real y = r.y < 0.5 ? h * (std::sqrt(2 * r.y) - 1) : h * (1 - std::sqrt(1 - 2 * (r.y - Real(0.5))));
    return V2{x, y};
}
inline V3 xform_point(const double *m, const V3 &p) { // src/transform.cpp:82-90
    Real tx = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    Real ty = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    Real tz = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    Real tw = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    Real inv_w = Real(1) / tw;
    return V3{tx * inv_w, ty * inv_w, tz * inv_w};
}
inline V3 xform_vector(const double *m, const V3 &v) {
    return V3{m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z};
}
struct Ray { V3 org, dir; Real tnear, tfar; };
Ray sample_primary(const GdptCamera &cam, const V2 &screen_pos) {
    V2 pixel_pos{screen_pos.x * cam.width, screen_pos.y * cam.height};
    Real dx = pixel_pos.x - std::floor(pixel_pos.x), dy = pixel_pos.y - std::floor(pixel_pos.y);
    V2 off = filter_sample(cam.filter_type, cam.filter_param, V2{dx, dy});
    V2 rp{(std::floor(pixel_pos.x) + Real(0.5) + off.x) / cam.width, (std::floor(pixel_pos.y) + Real(0.5) + off.y) / cam.height};
    V3 pt = xform_point(cam.sample_to_cam, V3{rp.x, rp.y, Real(0)});
    V3 dir = normalize(pt);
    return Ray{xform_point(cam.cam_to_world, V3{0, 0, 0}), normalize(xform_vector(cam.cam_to_world, dir)), Real(0),
               std::numeric_limits<Real>::infinity()};
}

// ---- shading info, src/shapes/triangle_mesh.inl:77-169, src/shapes/sphere.inl:243-268 ------------
struct ShadingInfo { V2 uv; Frame frame; Real mean_curvature, inv_uv_size; };

ShadingInfo shading_info_tri(const GdptShape &mesh, int prim, const V2 &st, const V3 &gn) {
    const int32_t *idx = &mesh.indices[3 * prim];
    auto P = [&](int i) { return V3{mesh.positions[3 * idx[i]], mesh.positions[3 * idx[i] + 1], mesh.positions[3 * idx[i] + 2]}; };
    V2 uvs[3];
    if (mesh.uvs) for (int i = 0; i < 3; i++) uvs[i] = V2{mesh.uvs[2 * idx[i]], mesh.uvs[2 * idx[i] + 1]};
    else { uvs[0] = V2{0, 0}; uvs[1] = V2{1, 0}; uvs[2] = V2{1, 1}; }
    Real b0 = 1 - st.x - st.y;
    V2 uv{b0 * uvs[0].x + st.x * uvs[1].x + st.y * uvs[2].x, b0 * uvs[0].y + st.x * uvs[1].y + st.y * uvs[2].y};
    V3 p0 = P(0), p1 = P(1), p2 = P(2);
    V2 duvds{uvs[2].x - uvs[0].x, uvs[2].y - uvs[0].y};
    V2 duvdt{uvs[2].x - uvs[1].x, uvs[2].y - uvs[1].y};
    Real det = duvds.x * duvdt.y - duvdt.x * duvds.y;
    Real dsdu = duvdt.y / det, dtdu = -duvds.y / det, dsdv = duvdt.x / det, dtdv = -duvds.x / det;
    V3 dpdu, dpdv;
    if (std::fabs(det) > 1e-8f) {
        V3 dpds = p2 - p0, dpdt = p2 - p1;
        dpdu = dpds * dsdu + dpdt * dtdu;
        dpdv = dpds * dsdv + dpdt * dtdv;
    } else {
        coordinate_system(gn, dpdu, dpdv);
    }
    V3 shading_normal = gn;
    Real mean_curvature = 0;
    V3 tangent, bitangent;
    if (mesh.normals) {
        auto N = [&](int i) { return V3{mesh.normals[3 * idx[i]], mesh.normals[3 * idx[i] + 1], mesh.normals[3 * idx[i] + 2]}; };
        V3 n0 = N(0), n1 = N(1), n2 = N(2);
        shading_normal = normalize(b0 * n0 + st.x * n1 + st.y * n2);
        tangent = normalize(dpdu - shading_normal * dot(shading_normal, dpdu));
        V3 dnds = n2 - n0, dndt = n2 - n1;
        V3 dndu = dnds * dsdu + dndt * dtdu, dndv = dnds * dsdv + dndt * dtdv;
        bitangent = normalize(cross(shading_normal, tangent));
        mean_curvature = (dot(dndu, tangent) + dot(dndv, bitangent)) / Real(2);
    } else {
        tangent = normalize(dpdu - shading_normal * dot(shading_normal, dpdu));
        bitangent = normalize(cross(shading_normal, tangent));
    }
    return ShadingInfo{uv, Frame{tangent, bitangent, shading_normal}, mean_curvature, rmax(length(dpdu), length(dpdv))};
}
ShadingInfo shading_info_sphere(const GdptShape &sp, const V2 &st, const V3 &gn) {
    Real r = sp.radius;
    V3 dpdu{-r * std::sin(st.x) * std::sin(st.y), r * std::cos(st.x) * std::sin(st.y), Real(0)};
    V3 dpdv{r * std::cos(st.x) * std::cos(st.y), r * std::sin(st.x) * std::cos(st.y), -r * std::sin(st.y)};
    V3 tangent = normalize(dpdu - gn * dot(gn, dpdu));
    Frame f{tangent, normalize(cross(gn, tangent)), gn};
    return ShadingInfo{st, f, 1 / r, (length(dpdu) + length(dpdv)) / 2};
}

// intersect(), src/intersection.cpp:7-65
bool intersect(const OracleScene &sc, const Ray &ray, Real rd_radius, Real rd_spread, Vertex *out) {
    float o[3] = {(float)ray.org.x, (float)ray.org.y, (float)ray.org.z};
    float d[3] = {(float)ray.dir.x, (float)ray.dir.y, (float)ray.dir.z};
    Hit h = closest_hit(sc, o, d, (float)ray.tnear, (float)ray.tfar);
    if (!h.valid) return false;
    Vertex v;
    v.t = Real(h.t);
    v.position = ray.org + ray.dir * Real(h.t);
    v.geometric_normal = normalize(V3{h.ng[0], h.ng[1], h.ng[2]});
    v.gid = h.gid;
    int ntri = (int)sc.tris.size();
    ShadingInfo si;
    v.st = V2{h.u, h.v};
    if (h.gid < ntri) {
        v.shape_id = sc.tris[h.gid].shape_id; v.primitive_id = sc.tris[h.gid].prim_id;
        si = shading_info_tri(sc.desc.shapes[v.shape_id], v.primitive_id, v.st, v.geometric_normal);
    } else {
        v.shape_id = sc.sphere_shapes[h.gid - ntri]; v.primitive_id = 0;
        si = shading_info_sphere(sc.desc.shapes[v.shape_id], v.st, v.geometric_normal);
    }
    v.material_id = sc.desc.shapes[v.shape_id].material_id;
    v.shading_frame = si.frame; v.uv = si.uv; v.mean_curvature = si.mean_curvature;
    Real dist = std::sqrt(distance_squared(ray.org, v.position));
    v.ray_radius = rd_radius + rd_spread * dist;           // transfer(), src/ray.h:38-40
    v.uv_screen_size = v.ray_radius / si.inv_uv_size;
    if (dot(v.geometric_normal, v.shading_frame.n) < 0) v.geometric_normal = -v.geometric_normal;
    *out = v;
    return true;
}

inline bool is_light(const OracleScene &sc, int shape_id) { return sc.desc.shapes[shape_id].area_light_id >= 0; }
// emission(), src/intersection.cpp:87-98 + src/lights/diffuse_area_light.inl:15-20
inline V3 emission(const OracleScene &sc, const Vertex &v, const V3 &view_dir) {
    const GdptLight &l = sc.desc.lights[sc.desc.shapes[v.shape_id].area_light_id];
    if (dot(v.geometric_normal, view_dir) <= 0) return V3{0, 0, 0};
    return V3{l.intensity[0], l.intensity[1], l.intensity[2]};
}

// ---- BSDFs --------------------------------------------------------------------------------------
struct BsdfSample { V3 dir_out; Real eta, roughness; };

inline V3 sample_cos_hemisphere(const V2 &r) { // src/material.cpp:4-11
    Real phi = c_TWOPI * r.x;
    Real tmp = std::sqrt(std::clamp(1 - r.y, Real(0), Real(1)));
    return V3{std::cos(phi) * tmp, std::sin(phi) * tmp, std::sqrt(std::clamp(r.y, Real(0), Real(1)))};
}
// src/microfacet.h:34-56
inline Real fresnel_dielectric(Real n_dot_i, Real n_dot_t, Real eta) {
    Real rs = (n_dot_i - eta * n_dot_t) / (n_dot_i + eta * n_dot_t);
    Real rp = (eta * n_dot_i - n_dot_t) / (eta * n_dot_i + n_dot_t);
    return (rs * rs + rp * rp) / 2;
}
inline Real fresnel_dielectric(Real n_dot_i, Real eta) {
    Real n_dot_t_sq = 1 - (1 - n_dot_i * n_dot_i) / (eta * eta);
    if (n_dot_t_sq < 0) return 1;
    Real n_dot_t = std::sqrt(n_dot_t_sq);
    return fresnel_dielectric(std::fabs(n_dot_i), n_dot_t, eta);
}
// src/microfacet.h:96-128 (isotropic) and :131-161 (anisotropic "custom")
V3 sample_visible_normals(const V3 &local_dir_in, Real alpha_x, Real alpha_y, const V2 &rnd) {
    if (local_dir_in.z < 0) return -sample_visible_normals(-local_dir_in, alpha_x, alpha_y, rnd);
    V3 hemi = normalize(V3{alpha_x * local_dir_in.x, alpha_y * local_dir_in.y, local_dir_in.z});
    Real r = std::sqrt(rnd.x);
    Real phi = 2 * c_PI * rnd.y;
    Real t1 = r * std::cos(phi), t2 = r * std::sin(phi);
    Real s = (1 + hemi.z) / 2;
    t2 = (1 - s) * std::sqrt(1 - t1 * t1) + s * t2;
    V3 disk_N{t1, t2, std::sqrt(rmax(Real(0), 1 - t1 * t1 - t2 * t2))};
    Frame hf = make_frame(hemi);
    V3 hemi_N = to_world(hf, disk_N);
    return normalize(V3{alpha_x * hemi_N.x, alpha_y * hemi_N.y, rmax(Real(0), hemi_N.z)});
}
V3 sample_clearcoat_normal(Real alpha, const V2 &rnd) { // src/microfacet.h:164-177
    Real u0 = rnd.x, u1 = rnd.y;
    Real h_azim = 2 * c_PI * u1;
    Real a2 = alpha * alpha;
    Real sin_e = std::pow(((std::pow(a2, 1 - u0) - a2) / (1 - a2)), 0.5);
    Real cos_e = std::pow(((1 - std::pow(a2, 1 - u0)) / (1 - a2)), 0.5);
    return normalize(V3{sin_e * std::cos(h_azim), sin_e * std::sin(h_azim), cos_e});
}

struct Ctx { const OracleScene &sc; const Vertex &v; };
inline V3 T3(const Ctx &c, const GdptTexture &t) { return tex_eval3(c.sc, t, c.v.uv, c.v.uv_screen_size); }
inline Real T1(const Ctx &c, const GdptTexture &t) { return tex_eval1(c.sc, t, c.v.uv, c.v.uv_screen_size); }

inline bool below(const Vertex &v, const V3 &d) { return dot(v.geometric_normal, d) < 0; }
inline Frame oriented_frame(const Vertex &v, const V3 &dir_in) {        // one-sided lobes
    Frame f = v.shading_frame;
    if (dot(f.n, dir_in) < 0) f = neg(f);
    return f;
}
inline Frame oriented_frame_2s(const Vertex &v, const V3 &dir_in) {     // glass / DisneyBSDF
    Frame f = v.shading_frame;
    if (dot(f.n, dir_in) * dot(v.geometric_normal, dir_in) < 0) f = neg(f);
    return f;
}

// Lambertian, src/materials/lambertian.inl
V3 lambert_eval(const Ctx &c, const GdptTexture &refl, const V3 &in, const V3 &out) {
    if (below(c.v, in) || below(c.v, out)) return V3{0, 0, 0};
    Frame f = oriented_frame(c.v, in);
    return std::fmax(dot(f.n, out), Real(0)) * T3(c, refl) / c_PI;
}
Real cos_pdf(const Ctx &c, const V3 &in, const V3 &out) {
    if (below(c.v, in) || below(c.v, out)) return 0;
    Frame f = oriented_frame(c.v, in);
    return std::fmax(dot(f.n, out), Real(0)) / c_PI;
}
bool cos_sample(const Ctx &c, const V3 &in, const V2 &ruv, Real roughness, BsdfSample *s) {
    if (below(c.v, in)) return false;
    Frame f = oriented_frame(c.v, in);
    *s = BsdfSample{to_world(f, sample_cos_hemisphere(ruv)), Real(0), roughness};
    return true;
}

// DisneyDiffuse, src/materials/disney_diffuse.inl
V3 dd_eval(const Ctx &c, const GdptTexture &base, const GdptTexture &rough, const GdptTexture &subs, const V3 &in, const V3 &out) {
    if (below(c.v, in) || below(c.v, out)) return V3{0, 0, 0};
    Frame f = oriented_frame(c.v, in);
    V3 h = normalize(in + out);
    Real n_in = dot(f.n, in), n_out = dot(f.n, out), h_out = dot(h, out);
    Real roughness = T1(c, rough);
    V3 bc = T3(c, base);
    roughness = std::clamp(roughness, Real(0.01), Real(1));
    Real f_d_90 = 0.5 + 2 * roughness * std::pow(std::fabs(h_out), 2);
    Real p5o = std::pow((1 - std::fabs(n_out)), 5), p5i = std::pow((1 - std::fabs(n_in)), 5);
    Real f_d_out = 1.0 + (f_d_90 - 1.0) * p5o, f_d_in = 1.0 + (f_d_90 - 1.0) * p5i;
    V3 f_base = (bc * f_d_in * f_d_out * std::fabs(n_out)) / c_PI;
    Real f_ss_90 = roughness * std::pow(std::fabs(h_out), 2);
    Real f_ss_in = 1.0 + (f_ss_90 - 1.0) * p5i, f_ss_out = 1.0 + (f_ss_90 - 1.0) * p5o;
    V3 f_ss = (1.25 * bc / c_PI) * ((f_ss_in * f_ss_out) * (splat(1 / (std::fabs(n_in) + std::fabs(n_out))) - splat(0.5)) + splat(0.5)) * std::fabs(n_out);
    Real sv = T1(c, subs);
    return ((1 - sv) * f_base + sv * f_ss);
}
bool dd_sample(const Ctx &c, const GdptTexture &rough, const V3 &in, const V2 &ruv, BsdfSample *s) {
    if (below(c.v, in)) return false;
    Real roughness = std::clamp(T1(c, rough), Real(0.01), Real(1));
    return cos_sample(c, in, ruv, roughness, s);
}

// DisneyMetal, src/materials/disney_metal.inl (base colour passed as a value: DisneyBSDF feeds a constant c_0)
V3 dm_eval(const Ctx &c, const V3 &bc, const GdptTexture &rough, const GdptTexture &aniso, const V3 &in, const V3 &out) {
    if (below(c.v, in) || below(c.v, out)) return V3{0, 0, 0};
    Frame f = oriented_frame(c.v, in);
    V3 h = normalize(in + out);
    Real n_in = dot(f.n, in), h_out = dot(h, out);
    Real roughness = T1(c, rough), anisotropic = T1(c, aniso);
    Real denom = 4 * std::fabs(n_in);
    roughness = std::clamp(roughness, Real(0.01), Real(1));
    V3 f_m = bc + (splat(1.0) - bc) * std::pow(1.0 - std::fabs(h_out), 5);
    V3 pv = to_local(f, h);
    Real aspect = std::sqrt(1 - 0.9 * anisotropic);
    Real ax = rmax(0.0001, std::pow(roughness, 2) / aspect), ay = rmax(0.0001, std::pow(roughness, 2) * aspect);
    Real dc = c_PI * ax * ay;
    Real D = 1 / (dc * std::pow((std::pow(pv.x / ax, 2) + std::pow(pv.y / ay, 2) + std::pow(pv.z, 2)), 2));
    V3 li = to_local(f, in), lo = to_local(f, out);
    Real io = (std::pow(lo.x * ax, 2) + std::pow(lo.y * ay, 2)) / (std::pow(lo.z, 2));
    Real ii = (std::pow(li.x * ax, 2) + std::pow(li.y * ay, 2)) / (std::pow(li.z, 2));
    Real d_out = (std::sqrt(1 + io) - 1) / 2, d_in = (std::sqrt(1 + ii) - 1) / 2;
    Real G = (1 / (1 + d_in)) * (1 / (1 + d_out));
    return (f_m * D * G) / denom;
}
Real dm_pdf(const Ctx &c, const GdptTexture &rough, const GdptTexture &aniso, const V3 &in, const V3 &out) {
    if (below(c.v, in) || below(c.v, out)) return 0;
    Frame f = oriented_frame(c.v, in);
    V3 h = normalize(in + out);
    Real n_in = dot(f.n, in);
    Real denom = 4 * std::fabs(n_in);
    Real roughness = T1(c, rough), anisotropic = T1(c, aniso);   // NOT clamped here (src/materials/disney_metal.inl:107-125)
    V3 pv = to_local(f, h);
    Real aspect = std::pow((1 - 0.9 * anisotropic), 0.5);
    Real ax = rmax(0.0001, std::pow(roughness, 2) / aspect), ay = rmax(0.0001, std::pow(roughness, 2) * aspect);
    Real dc = c_PI * ax * ay;
    Real D = 1 / (dc * std::pow((std::pow(pv.x / ax, 2) + std::pow(pv.y / ay, 2) + std::pow(pv.z, 2)), 2));
    V3 li = to_local(f, in);
    Real ii = (std::pow(li.x * ax, 2) + std::pow(li.y * ay, 2)) / (std::pow(li.z, 2));
    Real d_in = (std::sqrt(1 + ii) - 1) / 2;
    Real G = (1 / (1 + d_in));
    return (G * D) / denom;
}
bool dm_sample(const Ctx &c, const GdptTexture &rough, const GdptTexture &aniso, const V3 &in, const V2 &ruv, BsdfSample *s) {
    if (below(c.v, in)) return false;
    Frame f = oriented_frame(c.v, in);
    V3 li = to_local(f, in);
    Real roughness = T1(c, rough), anisotropic = T1(c, aniso);
    roughness = std::clamp(roughness, Real(0.01), Real(1));
    Real aspect = std::sqrt(1 - 0.9 * anisotropic);
    Real ax = rmax(0.0001, std::pow(roughness, 2) / aspect), ay = rmax(0.0001, std::pow(roughness, 2) * aspect);
    V3 lm = sample_visible_normals(li, ax, ay, ruv);
    V3 h = to_world(f, lm);
    V3 refl = normalize(-in + 2 * dot(in, h) * h);
    *s = BsdfSample{refl, Real(0), roughness};
    return true;
}

// DisneyClearcoat, src/materials/disney_clearcoat.inl
inline Real cc_D(Real alpha_g, Real hz) {
    return (std::pow(alpha_g, 2) - 1) / (c_PI * std::log(std::pow(alpha_g, 2)) * (1 + (std::pow(alpha_g, 2) - 1) * (std::pow(hz, 2))));
}
V3 cc_eval(const Ctx &c, const GdptTexture &gloss, const V3 &in, const V3 &out) {
    if (below(c.v, in) || below(c.v, out)) return V3{0, 0, 0};
    Frame f = oriented_frame(c.v, in);
    V3 h = normalize(in + out);
    Real n_in = dot(f.n, in), h_out = dot(h, out);
    Real r_0 = std::pow((1.5 - 1.0), 2) / std::pow((1.5 + 1.0), 2);
    Real f_c = r_0 + (1 - r_0) * std::pow(1 - std::fabs(h_out), 5);
    Real g = T1(c, gloss);
    Real alpha_g = (1 - g) * 0.1 + g * 0.001;
    V3 pv = to_local(f, h);
    Real d_c = cc_D(alpha_g, pv.z);
    V3 li = to_local(f, in), lo = to_local(f, out);
    Real dno = std::pow((1 + (std::pow(lo.x * 0.25, 2) + std::pow(lo.y * 0.25, 2)) / (std::pow(lo.z, 2))), 0.5) - 1.0;
    Real dni = std::pow((1 + (std::pow(li.x * 0.25, 2) + std::pow(li.y * 0.25, 2)) / (std::pow(li.z, 2))), 0.5) - 1.0;
    Real g_c = (1 / (1 + dni / 2)) * (1 / (1 + dno / 2));
    Real denom = 4 * (std::fabs(n_in));
    return splat((f_c * d_c * g_c) / denom);
}
Real cc_pdf(const Ctx &c, const GdptTexture &gloss, const V3 &in, const V3 &out) {
    if (below(c.v, in) || below(c.v, out)) return 0;
    Frame f = oriented_frame(c.v, in);
    V3 h = normalize(in + out);
    Real n_out = dot(f.n, out), n_h = dot(f.n, h);
    Real denom = 4 * (std::fabs(n_out));                        // sic: |n.out| (src/materials/disney_clearcoat.inl:64)
    Real g = T1(c, gloss);
    Real alpha_g = (1 - g) * 0.1 + g * 0.001;
    V3 pv = to_local(f, h);
    Real d_c = cc_D(alpha_g, pv.z);
    return (d_c * std::fabs(n_h)) / denom;
}
bool cc_sample(const Ctx &c, const GdptTexture &gloss, const V3 &in, const V2 &ruv, BsdfSample *s) {
    if (below(c.v, in)) return false;
    Frame f = oriented_frame(c.v, in);
    Real g = T1(c, gloss);
    Real alpha_g = (1 - g) * 0.1 + g * 0.001;
    V3 lm = sample_clearcoat_normal(alpha_g, ruv);
    V3 h = to_world(f, lm);
    V3 refl = normalize(-in + 2 * dot(in, h) * h);
    *s = BsdfSample{refl, Real(0), alpha_g};
    return true;
}

// DisneySheen, src/materials/disney_sheen.inl
V3 sh_eval(const Ctx &c, const GdptTexture &base, const GdptTexture &tint, const V3 &in, const V3 &out) {
    if (below(c.v, in) || below(c.v, out)) return V3{0, 0, 0};
    Frame f = oriented_frame(c.v, in);
    V3 h = normalize(in + out);
    Real h_out = dot(h, out), n_out = dot(f.n, out);
    V3 bc = T3(c, base);
    Real st = T1(c, tint);
    V3 c_tint = splat(1.0);
    if (luminance(bc) > 0) c_tint = bc / luminance(bc);
    V3 c_sheen = splat(1.0 - st) + st * c_tint;
    return c_sheen * (std::pow(1 - std::fabs(h_out), 5)) * std::fabs(n_out);
}

// DisneyGlass, src/materials/disney_glass.inl
struct GlassTerms { Frame f; V3 h; Real eta, F, d_m, g_in, g_out, h_dot_in; bool reflect; };
GlassTerms glass_terms(const Ctx &c, const GdptTexture &rough, const GdptTexture &aniso, Real bsdf_eta, const V3 &in, const V3 &out) {
    GlassTerms g;
    g.reflect = dot(c.v.geometric_normal, in) * dot(c.v.geometric_normal, out) > 0;
    g.f = oriented_frame_2s(c.v, in);
    g.eta = dot(c.v.geometric_normal, in) > 0 ? bsdf_eta : 1 / bsdf_eta;
    Real roughness = T1(c, rough);
    g.h = g.reflect ? normalize(in + out) : normalize(in + out * g.eta);
    if (dot(g.h, g.f.n) < 0) g.h = -g.h;
    roughness = std::clamp(roughness, Real(0.01), Real(1));
    Real anisotropic = T1(c, aniso);
    g.h_dot_in = dot(g.h, in);
    g.F = fresnel_dielectric(g.h_dot_in, g.eta);
    V3 pv = to_local(g.f, g.h);
    Real aspect = std::pow((1 - 0.9 * anisotropic), 0.5);
    Real ax = rmax(0.0001, std::pow(roughness, 2) / aspect), ay = rmax(0.0001, std::pow(roughness, 2) * aspect);
    Real dc = c_PI * ax * ay;
    g.d_m = 1 / (dc * std::pow((std::pow(pv.x, 2) / (std::pow(ax, 2)) + std::pow(pv.y, 2) / (std::pow(ay, 2)) + std::pow(pv.z, 2)), 2));
    V3 li = to_local(g.f, in), lo = to_local(g.f, out);
    Real io = (std::pow(lo.x * ax, 2) + std::pow(lo.y * ay, 2)) / (std::pow(lo.z, 2));
    Real ii = (std::pow(li.x * ax, 2) + std::pow(li.y * ay, 2)) / (std::pow(li.z, 2));
    Real d_out = (std::pow(1 + io, 0.5) - 1) / 2, d_in = (std::pow(1 + ii, 0.5) - 1) / 2;
    g.g_in = 1 / (1 + d_in); g.g_out = 1 / (1 + d_out);
    return g;
}
V3 dg_eval(const Ctx &c, const GdptTexture &base, const GdptTexture &rough, const GdptTexture &aniso, Real eta, const V3 &in, const V3 &out) {
    V3 bc = T3(c, base);
    GlassTerms g = glass_terms(c, rough, aniso, eta, in, out);
    Real g_m = g.g_in * g.g_out;
    if (g.reflect) return bc * (g.F * g.d_m * g_m) / (4 * std::fabs(dot(g.f.n, in)));
    Real h_dot_out = dot(g.h, out);
    Real sd = g.h_dot_in + g.eta * h_dot_out;
    V3 csq{std::pow(bc.x, 0.5), std::pow(bc.y, 0.5), std::pow(bc.z, 0.5)};
    return csq * ((1 - g.F) * g.d_m * g_m * std::fabs(h_dot_out * g.h_dot_in)) / (std::fabs(dot(g.f.n, in)) * sd * sd);
}
Real dg_pdf(const Ctx &c, const GdptTexture &rough, const GdptTexture &aniso, Real eta, const V3 &in, const V3 &out) {
    GlassTerms g = glass_terms(c, rough, aniso, eta, in, out);
    Real g_m = g.g_in;
    if (g.reflect) return (g.F * g.d_m * g_m) / (4 * std::fabs(dot(g.f.n, in)));
    Real h_dot_out = dot(g.h, out);
    Real prod = std::fabs(h_dot_out * g.h_dot_in);
    Real sd = g.h_dot_in + g.eta * h_dot_out;
    Real frame_in = dot(g.f.n, in);
    return ((1 - g.F) * g.d_m * g_m * std::fabs(prod)) / (std::fabs(frame_in) * sd * sd);
}
bool dg_sample(const Ctx &c, const GdptTexture &rough, Real bsdf_eta, const V3 &in, const V2 &ruv, Real rw, BsdfSample *s) {
    Real eta = dot(c.v.geometric_normal, in) > 0 ? bsdf_eta : 1 / bsdf_eta;
    Frame f = oriented_frame_2s(c.v, in);
    Real roughness = std::clamp(T1(c, rough), Real(0.01), Real(1));
    Real alpha = roughness * roughness;
    V3 li = to_local(f, in);
    V3 lm = sample_visible_normals(li, alpha, alpha, ruv);    // isotropic VNDF (src/materials/disney_glass.inl:195-198)
    V3 h = to_world(f, lm);
    if (dot(h, f.n) < 0) h = -h;
    Real h_dot_in = dot(h, in);
    Real F = fresnel_dielectric(h_dot_in, eta);
    if (rw <= F) {
        V3 refl = normalize(-in + 2 * dot(in, h) * h);
        *s = BsdfSample{refl, Real(0), roughness};
        return true;
    }
    Real h_dot_out_sq = 1 - (1 - h_dot_in * h_dot_in) / (eta * eta);
    if (h_dot_out_sq <= 0) return false;
    if (h_dot_in < 0) h = -h;
    Real h_dot_out = std::sqrt(h_dot_out_sq);
    V3 refr = -in / eta + (std::fabs(h_dot_in) / eta - h_dot_out) * h;
    *s = BsdfSample{refr, eta, roughness};
    return true;
}

// DisneyBSDF, src/materials/disney_bsdf.inl
struct DisneyParams { V3 base_color, c_0; Real spec_trans, metallic, clearcoat, sheen, eta; };
DisneyParams disney_params(const Ctx &c, const GdptMaterial &m, const V3 &in) {
    DisneyParams p;
    p.base_color = T3(c, m.tex[0]);
    p.spec_trans = T1(c, m.tex[1]); p.metallic = T1(c, m.tex[2]);
    Real specular = T1(c, m.tex[4]), specular_tint = T1(c, m.tex[6]);
    p.sheen = T1(c, m.tex[8]); p.clearcoat = T1(c, m.tex[10]);
    V3 c_tint = splat(1.0);
    if (luminance(p.base_color) > 0) c_tint = p.base_color / luminance(p.base_color);
    p.eta = dot(c.v.geometric_normal, in) > 0 ? m.eta : 1 / m.eta;
    V3 K_s = (1 - specular_tint) + specular_tint * c_tint;
    Real r_0 = std::pow((p.eta - 1), 2) / std::pow((p.eta + 1), 2);
    p.c_0 = specular * r_0 * (1 - p.metallic) * K_s + p.metallic * p.base_color;
    return p;
}
V3 db_eval(const Ctx &c, const GdptMaterial &m, const V3 &in, const V3 &out) {
    bool inside = dot(c.v.geometric_normal, in) <= 0;
    DisneyParams p = disney_params(c, m, in);
    // inner glass is built with the already-flipped eta and flips it again (src/materials/disney_bsdf.inl:29,38)
    V3 glass = dg_eval(c, m.tex[0], m.tex[5], m.tex[7], p.eta, in, out);
    Real wd = (1 - p.spec_trans) * (1 - p.metallic), wm = (1 - p.spec_trans * (1 - p.metallic));
    Real wc = 0.25 * p.clearcoat, wg = (1 - p.metallic) * p.spec_trans, ws = (1 - p.metallic) * p.sheen;
    if (inside) return wg * glass;
    V3 fd = dd_eval(c, m.tex[0], m.tex[5], m.tex[3], in, out);
    V3 fm = dm_eval(c, p.c_0, m.tex[5], m.tex[7], in, out);
    V3 fs = sh_eval(c, m.tex[0], m.tex[9], in, out);
    V3 fc = cc_eval(c, m.tex[11], in, out);
    return wd * fd + wm * fm + wc * fc + wg * glass + ws * fs;
}
Real db_pdf(const Ctx &c, const GdptMaterial &m, const V3 &in, const V3 &out) {
    bool inside = dot(c.v.geometric_normal, in) <= 0;
    DisneyParams p = disney_params(c, m, in);
    Real wd = (1 - p.spec_trans) * (1 - p.metallic), wm = (1 - p.spec_trans * (1 - p.metallic));
    Real wc = 0.25 * p.clearcoat, wg = (1 - p.metallic) * p.spec_trans;
    Real net = wd + wm + wc + wg;
    if (inside) return dg_pdf(c, m.tex[5], m.tex[7], p.eta, in, out);
    return (wd / net) * cos_pdf(c, in, out) + (wm / net) * dm_pdf(c, m.tex[5], m.tex[7], in, out) +
           (wc / net) * cc_pdf(c, m.tex[11], in, out) + (wg / net) * dg_pdf(c, m.tex[5], m.tex[7], p.eta, in, out);
}
bool db_sample(const Ctx &c, const GdptMaterial &m, const V3 &in, const V2 &ruv, Real rw, BsdfSample *s) {
    DisneyParams p = disney_params(c, m, in);
    Real r = ruv.x; // fixed thresholds, number reused unrescaled (src/materials/disney_bsdf.inl:173-191)
    if (r < 0.25) return dd_sample(c, m.tex[5], in, ruv, s);
    if (r >= 0.25 && r < 0.5) return dm_sample(c, m.tex[5], m.tex[7], in, ruv, s);
    if (r >= 0.5 && r < 0.75) return cc_sample(c, m.tex[11], in, ruv, s);
    return dg_sample(c, m.tex[5], p.eta, in, ruv, rw, s);
}

[[noreturn]] void unsupported(int type) {
    std::fprintf(stderr, "oracle: material type %d is outside the restated subset\n", type);
    std::abort();
}
// ---- RoughPlastic / RoughDielectric: src/materials/roughplastic.inl, roughdielectric.inl ----
inline Real gtr2_iso(Real n_dot_h, Real roughness) {                 // GTR2, src/microfacet.h:58-63
    Real alpha = roughness * roughness;
    Real a2 = alpha * alpha;
    Real t = 1 + (a2 - 1) * n_dot_h * n_dot_h;
    return a2 / (c_PI * t * t);
}
inline Real smith_gtr2_iso(const V3 &v_local, Real roughness) {      // smith_masking_gtr2, src/microfacet.h:72-78
    Real alpha = roughness * roughness;
    Real a2 = alpha * alpha;
    V3 v2 = v_local * v_local;
    Real Lambda = (-1 + std::sqrt(1 + (v2.x * a2 + v2.y * a2) / v2.z)) / 2;
    return 1 / (1 + Lambda);
}
V3 rp_eval(const Ctx &c, const GdptMaterial &m, const V3 &in, const V3 &out) {           // roughplastic.inl:3-43
    if (dot(c.v.geometric_normal, in) < 0 || dot(c.v.geometric_normal, out) < 0) return {0, 0, 0};
    Frame f = oriented_frame(c.v, in);
    V3 h = normalize(in + out);
    Real n_dot_h = dot(f.n, h), n_dot_in = dot(f.n, in), n_dot_out = dot(f.n, out);
    if (n_dot_out <= 0 || n_dot_h <= 0) return {0, 0, 0};
    V3 Kd = T3(c, m.tex[0]), Ks = T3(c, m.tex[1]);
    Real roughness = std::clamp(T1(c, m.tex[2]), Real(0.01), Real(1));
    Real F_o = fresnel_dielectric(dot(h, out), m.eta);
    Real D = gtr2_iso(n_dot_h, roughness);
    Real G = smith_gtr2_iso(to_local(f, in), roughness) * smith_gtr2_iso(to_local(f, out), roughness);
    V3 spec = Ks * (G * F_o * D) / (4 * n_dot_in * n_dot_out);
    Real F_i = fresnel_dielectric(dot(h, in), m.eta);
    V3 diff = Kd * (Real(1) - F_o) * (Real(1) - F_i) / c_PI;
    return (spec + diff) * n_dot_out;
}
Real rp_pdf(const Ctx &c, const GdptMaterial &m, const V3 &in, const V3 &out) {          // roughplastic.inl:45-88
    if (dot(c.v.geometric_normal, in) < 0 || dot(c.v.geometric_normal, out) < 0) return 0;
    Frame f = oriented_frame(c.v, in);
    V3 h = normalize(in + out);
    Real n_dot_in = dot(f.n, in), n_dot_out = dot(f.n, out), n_dot_h = dot(f.n, h);
    if (n_dot_out <= 0 || n_dot_h <= 0) return 0;
    Real lS = luminance(T3(c, m.tex[1])), lR = luminance(T3(c, m.tex[0]));
    if (lS + lR <= 0) return 0;
    Real roughness = std::clamp(T1(c, m.tex[2]), Real(0.01), Real(1));
    Real spec_prob = lS / (lS + lR);
    Real diff_prob = 1 - spec_prob;
    Real G = smith_gtr2_iso(to_local(f, in), roughness);
    Real D = gtr2_iso(n_dot_h, roughness);
    spec_prob *= (G * D) / (4 * n_dot_in);
    diff_prob *= n_dot_out / c_PI;
    return spec_prob + diff_prob;
}
bool rp_sample(const Ctx &c, const GdptMaterial &m, const V3 &in, const V2 &ruv, Real rw, BsdfSample *s) {   // :90-137
    if (dot(c.v.geometric_normal, in) < 0) return false;
    Frame f = oriented_frame(c.v, in);
    Real lS = luminance(T3(c, m.tex[1])), lR = luminance(T3(c, m.tex[0]));
    if (lS + lR <= 0) return false;
    Real spec_prob = lS / (lS + lR);
    if (rw < spec_prob) {
        V3 li = to_local(f, in);
        Real roughness = std::clamp(T1(c, m.tex[2]), Real(0.01), Real(1));
        Real alpha = roughness * roughness;
        V3 h = to_world(f, sample_visible_normals(li, alpha, alpha, ruv));
        s->dir_out = normalize(-in + 2 * dot(in, h) * h); s->eta = 0; s->roughness = roughness;
    } else {
        s->dir_out = to_world(f, sample_cos_hemisphere(ruv)); s->eta = 0; s->roughness = 1;
    }
    return true;
}
struct RdTerms { bool reflect; Frame f; Real eta, roughness, h_dot_in, F, D; V3 h; };
RdTerms rd_terms(const Ctx &c, const GdptMaterial &m, const V3 &in, const V3 &out) {     // roughdielectric.inl:4-37,52-78
    RdTerms t;
    t.reflect = dot(c.v.geometric_normal, in) * dot(c.v.geometric_normal, out) > 0;
    t.f = oriented_frame_2s(c.v, in);
    t.eta = dot(c.v.geometric_normal, in) > 0 ? m.eta : 1 / m.eta;
    t.h = t.reflect ? normalize(in + out) : normalize(in + out * t.eta);
    if (dot(t.h, t.f.n) < 0) t.h = -t.h;
    t.roughness = std::clamp(T1(c, m.tex[2]), Real(0.01), Real(1));
    t.h_dot_in = dot(t.h, in);
    t.F = fresnel_dielectric(t.h_dot_in, t.eta);
    t.D = gtr2_iso(dot(t.f.n, t.h), t.roughness);
    return t;
}
V3 rd_eval(const Ctx &c, const GdptMaterial &m, const V3 &in, const V3 &out) {           // roughdielectric.inl:3-49
    RdTerms t = rd_terms(c, m, in, out);
    V3 Ks = T3(c, m.tex[0]), Kt = T3(c, m.tex[1]);
    Real G = smith_gtr2_iso(to_local(t.f, in), t.roughness) * smith_gtr2_iso(to_local(t.f, out), t.roughness);
    if (t.reflect) return Ks * (t.F * t.D * G) / (4 * std::fabs(dot(t.f.n, in)));
    Real eta_factor = 1 / (t.eta * t.eta);                    // TransportDirection::TO_LIGHT, the default of eval() (src/material.h)
    Real h_dot_out = dot(t.h, out);
    Real sd = t.h_dot_in + t.eta * h_dot_out;
    return Kt * (eta_factor * (1 - t.F) * t.D * G * t.eta * t.eta * std::fabs(h_dot_out * t.h_dot_in)) / (std::fabs(dot(t.f.n, in)) * sd * sd);
}
Real rd_pdf(const Ctx &c, const GdptMaterial &m, const V3 &in, const V3 &out) {          // roughdielectric.inl:51-93
    RdTerms t = rd_terms(c, m, in, out);
    Real G_in = smith_gtr2_iso(to_local(t.f, in), t.roughness);
    if (t.reflect) return (t.F * t.D * G_in) / (4 * std::fabs(dot(t.f.n, in)));
    Real h_dot_out = dot(t.h, out);
    Real sd = t.h_dot_in + t.eta * h_dot_out;
    Real dh_dout = t.eta * t.eta * h_dot_out / (sd * sd);
    return (1 - t.F) * t.D * G_in * std::fabs(dh_dout * t.h_dot_in / dot(t.f.n, in));
}
bool rd_sample(const Ctx &c, const GdptMaterial &m, const V3 &in, const V2 &ruv, Real rw, BsdfSample *s) {   // :95-139
    Real eta = dot(c.v.geometric_normal, in) > 0 ? m.eta : 1 / m.eta;
    Frame f = oriented_frame_2s(c.v, in);
    Real roughness = std::clamp(T1(c, m.tex[2]), Real(0.01), Real(1));
    Real alpha = roughness * roughness;
    V3 h = to_world(f, sample_visible_normals(to_local(f, in), alpha, alpha, ruv));
    if (dot(h, f.n) < 0) h = -h;
    Real h_dot_in = dot(h, in);
    Real F = fresnel_dielectric(h_dot_in, eta);
    if (rw <= F) {
        s->dir_out = normalize(-in + 2 * dot(in, h) * h); s->eta = 0; s->roughness = roughness;
        return true;
    }
    Real h_dot_out_sq = 1 - (1 - h_dot_in * h_dot_in) / (eta * eta);
    if (h_dot_out_sq <= 0) return false;
    if (h_dot_in < 0) h = -h;
    Real h_dot_out = std::sqrt(h_dot_out_sq);
    s->dir_out = -in / eta + (std::fabs(h_dot_in) / eta - h_dot_out) * h; s->eta = eta; s->roughness = roughness;
    return true;
}

V3 bsdf_eval(const OracleScene &sc, const GdptMaterial &m, const V3 &in, const V3 &out, const Vertex &v) {
    Ctx c{sc, v};
    switch (m.type) {
        case GDPT_MAT_ROUGHPLASTIC: return rp_eval(c, m, in, out);
        case GDPT_MAT_ROUGHDIELECTRIC: return rd_eval(c, m, in, out);
        case GDPT_MAT_LAMBERTIAN: return lambert_eval(c, m.tex[0], in, out);
        case GDPT_MAT_DISNEY_DIFFUSE: return dd_eval(c, m.tex[0], m.tex[1], m.tex[2], in, out);
        case GDPT_MAT_DISNEY_METAL: return dm_eval(c, T3(c, m.tex[0]), m.tex[1], m.tex[2], in, out);
        case GDPT_MAT_DISNEY_GLASS: return dg_eval(c, m.tex[0], m.tex[1], m.tex[2], m.eta, in, out);
        case GDPT_MAT_DISNEY_CLEARCOAT: return cc_eval(c, m.tex[0], in, out);
        case GDPT_MAT_DISNEY_SHEEN: return sh_eval(c, m.tex[0], m.tex[1], in, out);
        case GDPT_MAT_DISNEY_BSDF: return db_eval(c, m, in, out);
        default: unsupported(m.type);
    }
}
Real bsdf_pdf(const OracleScene &sc, const GdptMaterial &m, const V3 &in, const V3 &out, const Vertex &v) {
    Ctx c{sc, v};
    switch (m.type) {
        case GDPT_MAT_ROUGHPLASTIC: return rp_pdf(c, m, in, out);
        case GDPT_MAT_ROUGHDIELECTRIC: return rd_pdf(c, m, in, out);
        case GDPT_MAT_LAMBERTIAN: case GDPT_MAT_DISNEY_DIFFUSE: case GDPT_MAT_DISNEY_SHEEN: return cos_pdf(c, in, out);
        case GDPT_MAT_DISNEY_METAL: return dm_pdf(c, m.tex[1], m.tex[2], in, out);
        case GDPT_MAT_DISNEY_GLASS: return dg_pdf(c, m.tex[1], m.tex[2], m.eta, in, out);
        case GDPT_MAT_DISNEY_CLEARCOAT: return cc_pdf(c, m.tex[0], in, out);
        case GDPT_MAT_DISNEY_BSDF: return db_pdf(c, m, in, out);
        default: unsupported(m.type);
    }
}
bool bsdf_sample(const OracleScene &sc, const GdptMaterial &m, const V3 &in, const Vertex &v, const V2 &ruv, Real rw, BsdfSample *s) {
    Ctx c{sc, v};
    switch (m.type) {
        case GDPT_MAT_ROUGHPLASTIC: return rp_sample(c, m, in, ruv, rw, s);
        case GDPT_MAT_ROUGHDIELECTRIC: return rd_sample(c, m, in, ruv, rw, s);
        case GDPT_MAT_LAMBERTIAN: case GDPT_MAT_DISNEY_SHEEN: return cos_sample(c, in, ruv, Real(1), s);
        case GDPT_MAT_DISNEY_DIFFUSE: return dd_sample(c, m.tex[1], in, ruv, s);
        case GDPT_MAT_DISNEY_METAL: return dm_sample(c, m.tex[1], m.tex[2], in, ruv, s);
        case GDPT_MAT_DISNEY_GLASS: return dg_sample(c, m.tex[1], m.eta, in, ruv, rw, s);
        case GDPT_MAT_DISNEY_CLEARCOAT: return cc_sample(c, m.tex[0], in, ruv, s);
        case GDPT_MAT_DISNEY_BSDF: return db_sample(c, m, in, ruv, rw, s);
        default: unsupported(m.type);
    }
}

// ---- grad_path_tracing, src/path_tracing.h:354-1050 (A-semantics for :1007-1010) ------------------
struct Offset {
    bool check = false;     // check_path_*
    Vertex vertex;          // offset's PRIMARY hit for the whole path (A-semantics)
    V3 ray_dir;             // ray_*.dir
    V3 contrib{1, 1, 1};
    Real jacob = 1.0;
};

void grad_sample(const OracleScene &sc, int x, int y, Pcg &rng, OracleSampleRecord *rec) {
    const GdptCamera &cam = sc.desc.camera;
    int w = cam.width, h = cam.height;
    OracleSampleRecord R;
    std::memset(&R, 0, sizeof(R));
    R.prob = 1.0; R.wX0 = R.wY0 = R.wX1 = R.wY1 = 1.0; // struct defaults, src/intersection.h:65-77
    auto done = [&]() { *rec = R; };

    double rng_x = pcg_real(rng), rng_y = pcg_real(rng);                       // :360-361
    R.rng_draws = 2;
    Ray ray = sample_primary(cam, V2{(x + rng_x) / w, (y + rng_y) / h});
    Real rd_radius = 0, rd_spread = Real(0.25) / rmax(w, h);                   // init_ray_differential, src/ray.h:33-35
    Vertex vertex;
    if (!intersect(sc, ray, rd_radius, rd_spread, &vertex)) { R.primary_miss = 1; done(); return; }  // :375-379

    // offsets in the reference's order x0,y0,x1,y1 = (x-1,y),(x,y+1),(x+1,y),(x,y-1)  (:385-403)
    // stored here as index 0:x0 1:x1 2:y0 3:y1
    const int ox[4] = {-1, +1, 0, 0}, oy[4] = {0, 0, +1, -1};
    Offset off[4];
    for (int k = 0; k < 4; k++) {
        Ray r = sample_primary(cam, V2{((x + ox[k]) + rng_x) / w, ((y + oy[k]) + rng_y) / h});
        off[k].ray_dir = r.dir;
        off[k].check = intersect(sc, r, rd_radius, rd_spread, &off[k].vertex);
        if (off[k].check && off[k].vertex.material_id != vertex.material_id) off[k].check = false; // :424-443
        R.valid0[k] = off[k].check;
    }

    V3 contrib{1, 1, 1};
    Real prob = 1.0;
    V3 throughput{1, 1, 1};
    Real eta_scale = 1;
    V3 radiance{0, 0, 0};
    if (is_light(sc, vertex.shape_id)) {                                       // :490-493
        radiance = radiance + throughput * emission(sc, vertex, -ray.dir);
        contrib = emission(sc, vertex, -ray.dir);
    }
    for (int k = 0; k < 4; k++)                                                // :496-508
        if (off[k].check && is_light(sc, off[k].vertex.shape_id)) off[k].contrib = emission(sc, off[k].vertex, -off[k].ray_dir);

    int max_depth = sc.desc.max_depth;
    for (int num_vertices = 3; max_depth == -1 || num_vertices <= max_depth + 1; num_vertices++) {
        R.bounces++;
        const GdptMaterial &mat = sc.desc.materials[vertex.material_id];
        V3 dir_view = -ray.dir;
        V2 ruv; ruv.x = pcg_real(rng); ruv.y = pcg_real(rng);                  // brace-init: ordered, :536
        Real rw = pcg_real(rng);
        R.rng_draws += 3;
        BsdfSample bs;
        if (!bsdf_sample(sc, mat, dir_view, vertex, ruv, rw, &bs)) {           // :545-548 -> GraidentPTRadiance{}
            int b = R.bounces, d = R.rng_draws; int v0[4]; std::memcpy(v0, R.valid0, sizeof(v0));
            std::memset(&R, 0, sizeof(R));
            R.prob = 1.0; R.wX0 = R.wY0 = R.wX1 = R.wY1 = 1.0; R.bounces = b; R.rng_draws = d; std::memcpy(R.valid0, v0, sizeof(v0));
            done(); return;
        }
        V3 dir_bsdf = bs.dir_out;
        if (bs.eta != 0) eta_scale /= (bs.eta * bs.eta);                       // :553-558 (ray_diff.spread is never read again)
        Ray bsdf_ray{vertex.position, dir_bsdf, sc.isect_eps, std::numeric_limits<Real>::infinity()};
        Vertex bsdf_vertex;
        bool hit = intersect(sc, bsdf_ray, 0, 0, &bsdf_vertex);                // default RayDifferential{}, :564
        // :565-568 four tnear=tfar=0 rays: always miss, unobservable -> omitted

        // :571-740: CHECK_TYPE is always false -> only the material test survives
        for (int k = 0; k < 4; k++)
            if (off[k].check && off[k].vertex.material_id != vertex.material_id) {
                off[k].check = false; off[k].contrib = V3{0, 0, 0}; off[k].jacob = 1.0;
            }

        Real G;
        if (hit) G = std::fabs(dot(dir_bsdf, bsdf_vertex.geometric_normal)) / distance_squared(bsdf_vertex.position, vertex.position);
        else G = 1;
        V3 f = bsdf_eval(sc, mat, dir_view, dir_bsdf, vertex);
        Real p2 = bsdf_pdf(sc, mat, dir_view, dir_bsdf, vertex);
        if (p2 <= 0) break;                                                     // :760-763
        p2 *= G;
        contrib = contrib * f * G;                                              // :769
        prob *= p2;

        for (int k = 0; k < 4; k++) {                                           // :773-959 (merge_flag never set)
            Offset &o = off[k];
            if (!o.check) continue;
            const GdptMaterial &omat = sc.desc.materials[o.vertex.material_id];
            BsdfSample os;
            V3 oin = -o.ray_dir;
            if (!bsdf_sample(sc, omat, oin, o.vertex, ruv, rw, &os)) {
                o.check = false; o.contrib = V3{0, 0, 0}; o.jacob = 1.0;
            } else {
                Real p2o = bsdf_pdf(sc, omat, oin, os.dir_out, o.vertex);
                if (p2o <= 0.0) { o.check = false; o.contrib = V3{0, 0, 0}; o.jacob = 1.0; }
                else o.jacob *= p2 / p2o;                                       // p2 includes G, p2o does not (:813)
                o.ray_dir = os.dir_out;                                         // :815-816
            }
        }

        if (hit && is_light(sc, bsdf_vertex.shape_id)) {                        // :971-980
            V3 L = emission(sc, bsdf_vertex, -dir_bsdf);
            V3 C2 = G * f * L;
            contrib = contrib * L;
            C2 = C2 / p2;
            radiance = radiance + throughput * C2;
        }
        if (!hit) break;                                                         // :982-985
        Real rr_prob = 1;
        if (num_vertices - 1 >= sc.desc.rr_depth) {                             // :992-999
            rr_prob = rmin(maxc((1 / eta_scale) * throughput), Real(0.95));
            R.rng_draws++;
            if (pcg_real(rng) > rr_prob) break;
        }
        ray = bsdf_ray;
        vertex = bsdf_vertex;
        throughput = throughput * (G * f) / (p2 * rr_prob);                     // :1003
        // :1007-1010 (undefined reads) -> A-semantics: offset vertices keep their primary hit
    }

    for (int c = 0; c < 3; c++) { R.radiance[c] = radiance[c]; R.contrib[c] = contrib[c]; }
    R.prob = prob;
    double *cx[4] = {R.contribX0, R.contribX1, R.contribY0, R.contribY1};
    double *wx[4] = {&R.wX0, &R.wX1, &R.wY0, &R.wY1};
    for (int k = 0; k < 4; k++)
        if (off[k].check) {                                                      // :1019-1045 (prob_x* stays 1)
            V3 cj = off[k].contrib * off[k].jacob;
            for (int c = 0; c < 3; c++) cx[k][c] = cj[c];
            *wx[k] = prob / (prob + 1.0 * off[k].jacob);
        }
    done();
}

// ---- oracle-side BVH (median split; only has to return the same closest hit as brute force) ------
void prim_bounds(const OracleScene &sc, int gid, float mn[3], float mx[3]) {
    int ntri = (int)sc.tris.size();
    if (gid < ntri) {
        const Tri &t = sc.tris[gid];
        // bounds from the fp32 vertices the traversal copy was made of: v0, v0+e1, v0+e2 are not exactly v1,v2,
        // so widen by the true vertices as well
        const GdptShape &sh = sc.desc.shapes[t.shape_id];
        for (int k = 0; k < 3; k++) { mn[k] = std::numeric_limits<float>::infinity(); mx[k] = -mn[k]; }
        for (int i = 0; i < 3; i++) {
            int vi = sh.indices[3 * t.prim_id + i];
            for (int k = 0; k < 3; k++) { float p = (float)sh.positions[3 * vi + k]; mn[k] = std::min(mn[k], p); mx[k] = std::max(mx[k], p); }
        }
    } else {
        const GdptShape &sp = sc.desc.shapes[sc.sphere_shapes[gid - ntri]];
        for (int k = 0; k < 3; k++) {
            mn[k] = std::nextafterf((float)(sp.center[k] - sp.radius), -std::numeric_limits<float>::infinity());
            mx[k] = std::nextafterf((float)(sp.center[k] + sp.radius), std::numeric_limits<float>::infinity());
        }
    }
}
int build_node(OracleScene &sc, std::vector<int> &ids, int b, int e, const std::vector<float> &cent, const std::vector<float> &bb) {
    int me = (int)sc.nodes.size();
    sc.nodes.emplace_back();
    BNode n;
    for (int k = 0; k < 3; k++) { n.mn[k] = std::numeric_limits<float>::infinity(); n.mx[k] = -n.mn[k]; }
    for (int i = b; i < e; i++) for (int k = 0; k < 3; k++) { n.mn[k] = std::min(n.mn[k], bb[6 * ids[i] + k]); n.mx[k] = std::max(n.mx[k], bb[6 * ids[i] + 3 + k]); }
    n.left = n.right = -1; n.first = 0; n.count = 0;
    if (e - b <= 4) {
        n.first = (int)sc.bvh_prims.size(); n.count = e - b;
        for (int i = b; i < e; i++) sc.bvh_prims.push_back(ids[i]);
        sc.nodes[me] = n;
        return me;
    }
    int ax = 0; float ext = -1;
    for (int k = 0; k < 3; k++) { float d = n.mx[k] - n.mn[k]; if (d > ext) { ext = d; ax = k; } }
    int m = (b + e) / 2;
    std::nth_element(ids.begin() + b, ids.begin() + m, ids.begin() + e, [&](int p, int q) { return cent[3 * p + ax] < cent[3 * q + ax]; });
    int l = build_node(sc, ids, b, m, cent, bb);
    int r = build_node(sc, ids, m, e, cent, bb);
    n.left = l; n.right = r;
    sc.nodes[me] = n;
    return me;
}

// naive DCT-I (FFTW REDFT00): Y[k] = X[0] + (-1)^k X[n-1] + 2 sum_{j=1}^{n-2} X[j] cos(pi j k/(n-1))
void dct1(const std::vector<double> &tab, int n, const double *in, int stride, double *outp) {
    for (int k = 0; k < n; k++) {
        double s = in[0] + ((k & 1) ? -in[(size_t)(n - 1) * stride] : in[(size_t)(n - 1) * stride]);
        for (int j = 1; j < n - 1; j++) s += 2.0 * in[(size_t)j * stride] * tab[((size_t)j * k) % (2 * (size_t)(n - 1))];
        outp[k] = s;
    }
}

} // namespace

// ================================= C interface ====================================================
namespace {

// ---- Integrator::Path pieces -------------------------------------------------------------------
void make_table_1d(const std::vector<Real> &f, std::vector<Real> &pmf, std::vector<Real> &cdf) { // src/table_dist.cpp:3-25
    pmf = f;
    cdf.assign(f.size() + 1, 0);
    for (size_t i = 0; i < f.size(); i++) cdf[i + 1] = cdf[i] + pmf[i];
    Real total = cdf.back();
    if (total > 0) {
        for (size_t i = 0; i < pmf.size(); i++) { pmf[i] /= total; cdf[i] /= total; }   // cdf.back() keeps the total
    } else {
        for (size_t i = 0; i < pmf.size(); i++) { pmf[i] = Real(1) / Real(pmf.size()); cdf[i] = Real(i) / Real(pmf.size()); }
        cdf.back() = 1;
    }
}
int table_sample(const std::vector<Real> &pmf, const std::vector<Real> &cdf, Real u) { // src/table_dist.cpp:27-33
    int size = (int)pmf.size();
    const Real *ptr = std::upper_bound(cdf.data(), cdf.data() + size + 1, u);
    return std::min(std::max(int(ptr - cdf.data() - 1), 0), size - 1);
}

// make_table_dist_2d / sample / pdf, src/table_dist.cpp:40-150
void make_table_2d(const std::vector<Real> &f, int width, int height, OracleScene::Table2D &t) {
    t.width = width; t.height = height;
    t.cdf_rows.assign((size_t)height * (width + 1), 0); t.pdf_rows.assign((size_t)height * width, 0);
    for (int y = 0; y < height; y++) {
        Real *cdf = &t.cdf_rows[(size_t)y * (width + 1)];
        cdf[0] = 0;
        for (int x = 0; x < width; x++) cdf[x + 1] = cdf[x] + f[(size_t)y * width + x];
        Real integral = cdf[width];
        if (integral > 0) {
            for (int x = 0; x < width; x++) cdf[x] /= integral;
            for (int x = 0; x < width; x++) t.pdf_rows[(size_t)y * width + x] = f[(size_t)y * width + x] / integral;
        } else {
            for (int x = 0; x < width; x++) { t.pdf_rows[(size_t)y * width + x] = Real(1) / Real(width); cdf[x] = Real(x) / Real(width); }
            cdf[width] = 1;
        }
    }
    t.cdf_marginals.assign((size_t)height + 1, 0); t.pdf_marginals.assign((size_t)height, 0);
    for (int y = 0; y < height; y++) t.cdf_marginals[(size_t)y + 1] = t.cdf_marginals[(size_t)y] + t.cdf_rows[(size_t)y * (width + 1) + width];
    t.total_values = t.cdf_marginals.back();
    if (t.total_values > 0) {
        for (int y = 0; y < height; y++) t.cdf_marginals[(size_t)y] /= t.total_values;
        t.cdf_marginals[(size_t)height] = 1;
        for (int y = 0; y < height; y++) t.pdf_marginals[(size_t)y] = t.cdf_rows[(size_t)y * (width + 1) + width] / t.total_values;
    } else {
        for (int y = 0; y < height; y++) { t.pdf_marginals[(size_t)y] = Real(1) / Real(height); t.cdf_marginals[(size_t)y] = Real(y) / Real(height); }
        t.cdf_marginals[(size_t)height] = 1;
    }
    for (int y = 0; y < height; y++) t.cdf_rows[(size_t)y * (width + 1) + width] = 1;
}
V2 table2d_sample(const OracleScene::Table2D &t, const V2 &rnd) {
    int w = t.width, h = t.height;
    const Real *y_ptr = std::upper_bound(t.cdf_marginals.data(), t.cdf_marginals.data() + h + 1, rnd.y);
    int y_offset = std::min(std::max(int(y_ptr - t.cdf_marginals.data() - 1), 0), h - 1);
    Real dy = rnd.y - t.cdf_marginals[(size_t)y_offset];
    if ((t.cdf_marginals[(size_t)y_offset + 1] - t.cdf_marginals[(size_t)y_offset]) > 0) dy /= (t.cdf_marginals[(size_t)y_offset + 1] - t.cdf_marginals[(size_t)y_offset]);
    const Real *cdf = &t.cdf_rows[(size_t)y_offset * (w + 1)];
    const Real *x_ptr = std::upper_bound(cdf, cdf + w + 1, rnd.x);
    int x_offset = std::min(std::max(int(x_ptr - cdf - 1), 0), w - 1);
    Real dx = rnd.x - cdf[x_offset];
    if (cdf[x_offset + 1] - cdf[x_offset] > 0) dx /= (cdf[x_offset + 1] - cdf[x_offset]);
    return V2{(x_offset + dx) / w, (y_offset + dy) / h};
}
Real table2d_pdf(const OracleScene::Table2D &t, const V2 &xy) {
    int w = t.width, h = t.height;
    int x = (int)std::min(std::max(xy.x * w, Real(0)), Real(w - 1));
    int y = (int)std::min(std::max(xy.y * h, Real(0)), Real(h - 1));
    return t.pdf_marginals[(size_t)y] * t.pdf_rows[(size_t)y * w + x] * w * h;
}

// Envmap, src/lights/envmap.inl
inline V2 envmap_uv(const V3 &local_dir) {
    const Real c_INVTWOPI = Real(1) / c_TWOPI, c_INVPI = Real(1) / c_PI;   // src/lajolla.h
    V2 uv{std::atan2(local_dir.x, -local_dir.z) * c_INVTWOPI, std::acos(std::min(std::max(local_dir.y, Real(-1)), Real(1))) * c_INVPI};
    if (uv.x < 0) uv.x += 1;
    return uv;
}
V3 envmap_emission(const OracleScene &sc, const V3 &view_dir) {                     // :49-64; view_dir points away from the light
    const GdptEnvmap &e = sc.desc.envmap;
    V3 local_dir = xform_vector(e.to_local, -view_dir);
    V2 uv = envmap_uv(local_dir);
    V3 w = local_dir;
    Real dudwx = -w.z / (w.x * w.x + w.z * w.z);
    Real dudwz = w.x / (w.x * w.x + w.z * w.z);
    Real dvdwy = -1 / std::sqrt(rmax(1 - w.y * w.y, Real(0)));
    Real footprint = rmin(std::sqrt(dudwx * dudwx + dudwz * dudwz), dvdwy);
    GdptTexture t{};
    t.type = GDPT_TEX_IMAGE; t.image_id = e.image_id; t.uscale = t.vscale = 1; t.uoffset = t.voffset = 0;
    return tex_eval3(sc, t, uv, footprint) * e.scale;
}
V3 envmap_sample_dir(const OracleScene &sc, const V2 &rnd_uv) {                      // :7-18, returns world_dir (normal = -world_dir)
    V2 uv = table2d_sample(sc.env_table, rnd_uv);
    Real azimuth = uv.x * (2 * c_PI);
    Real elevation = uv.y * c_PI;
    V3 local_dir{std::sin(azimuth) * std::sin(elevation), std::cos(elevation), -std::cos(azimuth) * std::sin(elevation)};
    return xform_vector(sc.desc.envmap.to_world, local_dir);
}
Real envmap_pdf(const OracleScene &sc, const V3 &normal) {                           // :20-37 (point_on_light.normal)
    V3 world_dir = -normal;
    V3 local_dir = xform_vector(sc.desc.envmap.to_local, world_dir);
    V2 uv = envmap_uv(local_dir);
    Real cos_elevation = local_dir.y;
    Real sin_elevation = std::sqrt(std::min(std::max(1 - cos_elevation * cos_elevation, Real(0)), Real(1)));
    if (sin_elevation <= 0) return 0;
    return table2d_pdf(sc.env_table, uv) / (2 * c_PI * c_PI * sin_elevation);
}

struct PointNormal { V3 position, normal; };

PointNormal sample_point_on_shape(const OracleScene &sc, int shape_id, const V3 &ref_point, const V2 &uv, Real w) {
    const GdptShape &sh = sc.desc.shapes[shape_id];
    if (sh.type == GDPT_SHAPE_TRIMESH) {                                   // src/shapes/triangle_mesh.inl:24-50
        const OracleScene::MeshTable &tb = sc.mesh_tables[(size_t)shape_id];
        int tri = table_sample(tb.pmf, tb.cdf, w);
        const int *ix = sh.indices + 3 * tri;
        auto P = [&](int i) { return V3{sh.positions[3 * ix[i]], sh.positions[3 * ix[i] + 1], sh.positions[3 * ix[i] + 2]}; };
        V3 v0 = P(0), v1 = P(1), v2 = P(2);
        V3 e1 = v1 - v0, e2 = v2 - v0;
        Real a = std::sqrt(std::min(std::max(uv.x, Real(0)), Real(1)));
        Real b1 = 1 - a, b2 = a * uv.y;
        V3 gn = normalize(cross(e1, e2));
        if (sh.normals) {
            auto N = [&](int i) { return V3{sh.normals[3 * ix[i]], sh.normals[3 * ix[i] + 1], sh.normals[3 * ix[i] + 2]}; };
            V3 sn = normalize((1 - b1 - b2) * N(0) + b1 * N(1) + b2 * N(2));
            if (dot(gn, sn) < 0) gn = -gn;
        }
        return {v0 + (e1 * b1) + (e2 * b2), gn};
    }
    // sphere, src/shapes/sphere.inl:161-205
    V3 center{sh.center[0], sh.center[1], sh.center[2]};
    Real r = sh.radius;
    if (distance_squared(ref_point, center) < r * r) {
        Real z = 1 - 2 * uv.x;
        Real r_ = std::sqrt(std::fmax(Real(0), 1 - z * z));
        Real phi = 2 * c_PI * uv.y;
        V3 offset{r_ * std::cos(phi), r_ * std::sin(phi), z};
        return {center + r * offset, offset};
    }
    V3 dir_to_center = normalize(center - ref_point);
    Frame frame = make_frame(dir_to_center);
    Real sin_elevation_max_sq = r * r / distance_squared(ref_point, center);
    Real cos_elevation_max = std::sqrt(rmax(Real(0), 1 - sin_elevation_max_sq));
    Real cos_elevation = (1 - uv.x) + uv.x * cos_elevation_max;
    Real sin_elevation = std::sqrt(rmax(Real(0), 1 - cos_elevation * cos_elevation));
    Real azimuth = uv.y * 2 * c_PI;
    Real dc = std::sqrt(distance_squared(ref_point, center));
    Real ds = dc * cos_elevation - std::sqrt(rmax(Real(0), r * r - dc * dc * sin_elevation * sin_elevation));
    Real cos_alpha = (dc * dc + r * r - ds * ds) / (2 * dc * r);
    Real sin_alpha = std::sqrt(rmax(Real(0), 1 - cos_alpha * cos_alpha));
    V3 n_on_sphere = -to_world(frame, V3{sin_alpha * std::cos(azimuth), sin_alpha * std::sin(azimuth), cos_alpha});
    return {r * n_on_sphere + center, n_on_sphere};
}

Real surface_area(const OracleScene &sc, int shape_id) {
    const GdptShape &sh = sc.desc.shapes[shape_id];
    if (sh.type == GDPT_SHAPE_TRIMESH) return sc.mesh_tables[(size_t)shape_id].total_area;
    return 4 * c_PI * sh.radius * sh.radius;
}

Real pdf_point_on_shape(const OracleScene &sc, int shape_id, const PointNormal &pt, const V3 &ref_point) {
    const GdptShape &sh = sc.desc.shapes[shape_id];
    if (sh.type == GDPT_SHAPE_TRIMESH) return 1 / surface_area(sc, shape_id);      // triangle_mesh.inl:56-58
    V3 center{sh.center[0], sh.center[1], sh.center[2]};                            // sphere.inl:211-228
    Real r = sh.radius;
    if (distance_squared(ref_point, center) < r * r) return 1 / surface_area(sc, shape_id);
    Real sin_elevation_max_sq = r * r / distance_squared(ref_point, center);
    Real cos_elevation_max = std::sqrt(rmax(Real(0), 1 - sin_elevation_max_sq));
    Real pdf_solid_angle = 1 / (2 * c_PI * (1 - cos_elevation_max));
    V3 dir = normalize(pt.position - ref_point);
    return pdf_solid_angle * std::fabs(dot(pt.normal, dir)) / distance_squared(ref_point, pt.position);
}

bool occluded(const OracleScene &sc, const Ray &ray) {                              // src/intersection.cpp:67-85
    float o[3] = {(float)ray.org.x, (float)ray.org.y, (float)ray.org.z};
    float d[3] = {(float)ray.dir.x, (float)ray.dir.y, (float)ray.dir.z};
    return closest_hit(sc, o, d, (float)ray.tnear, (float)ray.tfar).valid;
}

// path_tracing, src/path_tracing.h:13-348, scenes without an environment map.
V3 path_sample(const OracleScene &sc, int x, int y, Pcg &rng, int *bounces_out, int *shadow_out, int *rays_out = nullptr) {
    const GdptSceneDesc &D = sc.desc;
    int w = D.camera.width, h = D.camera.height;
    int bounces = 0, shadows = 0, rays = 1;      // rays: closest-hit + occlusion queries actually issued
    if (bounces_out) *bounces_out = 0;
    if (shadow_out) *shadow_out = 0;
    if (rays_out) *rays_out = 1;
    // :21-22 — constructor arguments; evaluated left to right by the compiler the reference was developed with
    Real rx = pcg_real(rng);
    Real ry = pcg_real(rng);
    V2 screen_pos{(x + rx) / w, (y + ry) / h};
    Ray ray = sample_primary(D.camera, screen_pos);
    Real rd_spread = Real(0.25) / Real(std::max(w, h));                              // init_ray_differential, src/ray.h:33-35
    Vertex vertex;
    if (!intersect(sc, ray, 0, rd_spread, &vertex))                                 // :31-43
        return D.has_envmap ? envmap_emission(sc, -ray.dir) : V3{0, 0, 0};
    V3 radiance{0, 0, 0};
    V3 throughput{1, 1, 1};
    Real eta_scale = 1;
    if (is_light(sc, vertex.shape_id)) radiance = radiance + throughput * emission(sc, vertex, -ray.dir);   // :76-79
    int max_depth = D.max_depth;
    Real shadow_eps = sc.isect_eps;                                                 // get_shadow_epsilon == get_intersection_epsilon (src/scene.h:100-106)
    for (int num_vertices = 3; max_depth == -1 || num_vertices <= max_depth + 1; num_vertices++) {
        bounces++;
        const GdptMaterial &mat = D.materials[vertex.material_id];
        // ---- next-event estimation, :116-175
        V2 light_uv; light_uv.x = pcg_real(rng); light_uv.y = pcg_real(rng);
        Real light_w = pcg_real(rng);
        Real shape_w = pcg_real(rng);
        int light_id = table_sample(sc.light_pmf, sc.light_cdf, light_w);
        const GdptLight &light = D.lights[light_id];
        const bool env_light = D.has_envmap && light_id == D.envmap.light_id;
        V3 C1{0, 0, 0};
        Real w1 = 0;
        if (!env_light) {
            PointNormal pl = sample_point_on_shape(sc, light.shape_id, vertex.position, light_uv, shape_w);
            Real G = 0;
            V3 dir_light = normalize(pl.position - vertex.position);
            Real dist = std::sqrt(distance_squared(pl.position, vertex.position));
            Ray shadow_ray{vertex.position, dir_light, shadow_eps, (1 - shadow_eps) * dist};
            shadows++; rays++;
            if (!occluded(sc, shadow_ray))
                G = rmax(-dot(dir_light, pl.normal), Real(0)) / distance_squared(pl.position, vertex.position);
            Real p1 = sc.light_pmf[(size_t)light_id] * pdf_point_on_shape(sc, light.shape_id, pl, vertex.position);
            if (G > 0 && p1 > 0) {
                V3 dir_view = -ray.dir;
                V3 f = bsdf_eval(sc, mat, dir_view, dir_light, vertex);
                V3 L = (dot(pl.normal, -dir_light) <= 0) ? V3{0, 0, 0} : V3{light.intensity[0], light.intensity[1], light.intensity[2]};
                C1 = G * f * L;
                Real p2 = bsdf_pdf(sc, mat, dir_view, dir_light, vertex);
                p2 *= G;
                w1 = (p1 * p1) / (p1 * p1 + p2 * p2);
                C1 = C1 / p1;
            }
        } else {                                                                    // :151-160: directional, G = 1 if unoccluded
            V3 world_dir = envmap_sample_dir(sc, light_uv);
            V3 normal = -world_dir;                                                 // point_on_light.normal
            V3 dir_light = -normal;
            Real G = 0;
            Ray shadow_ray{vertex.position, dir_light, shadow_eps, std::numeric_limits<Real>::infinity()};
            shadows++; rays++;
            if (!occluded(sc, shadow_ray)) G = 1;
            Real p1 = sc.light_pmf[(size_t)light_id] * envmap_pdf(sc, normal);
            if (G > 0 && p1 > 0) {
                V3 dir_view = -ray.dir;
                V3 f = bsdf_eval(sc, mat, dir_view, dir_light, vertex);
                V3 L = envmap_emission(sc, -dir_light);
                C1 = G * f * L;
                Real p2 = bsdf_pdf(sc, mat, dir_view, dir_light, vertex);
                p2 *= G;
                w1 = (p1 * p1) / (p1 * p1 + p2 * p2);
                C1 = C1 / p1;
            }
        }
        radiance = radiance + throughput * C1 * w1;
        // ---- BSDF sampling, :186-230
        V3 dir_view = -ray.dir;
        V2 ruv; ruv.x = pcg_real(rng); ruv.y = pcg_real(rng);
        Real rw = pcg_real(rng);
        BsdfSample bs;
        if (!bsdf_sample(sc, mat, dir_view, vertex, ruv, rw, &bs)) break;           // :200-203
        V3 dir_bsdf = bs.dir_out;
        if (bs.eta != 0) eta_scale /= (bs.eta * bs.eta);                            // ray_diff.spread only feeds envmap lookups
        Ray bsdf_ray{vertex.position, dir_bsdf, sc.isect_eps, std::numeric_limits<Real>::infinity()};
        Vertex bsdf_vertex;
        bool hit = intersect(sc, bsdf_ray, 0, 0, &bsdf_vertex);
        rays++;
        Real G = 1;
        if (hit) G = std::fabs(dot(dir_bsdf, bsdf_vertex.geometric_normal)) / distance_squared(bsdf_vertex.position, vertex.position);
        V3 f = bsdf_eval(sc, mat, dir_view, dir_bsdf, vertex);
        Real p2 = bsdf_pdf(sc, mat, dir_view, dir_bsdf, vertex);
        if (p2 <= 0) break;                                                         // :263-266
        p2 *= G;
        if (hit && is_light(sc, bsdf_vertex.shape_id)) {                            // :286-306: added WITHOUT the MIS weight w2
            V3 L = emission(sc, bsdf_vertex, -dir_bsdf);
            V3 C2 = G * f * L;
            C2 = C2 / p2;
            radiance = radiance + throughput * C2;
        }
        else if (!hit && D.has_envmap) {                                            // :307-325: WITH the MIS weight
            V3 L = envmap_emission(sc, -dir_bsdf);
            V3 C2 = G * f * L;
            Real p1 = sc.light_pmf[(size_t)D.envmap.light_id] * envmap_pdf(sc, -dir_bsdf);
            Real w2 = (p2 * p2) / (p1 * p1 + p2 * p2);
            C2 = C2 / p2;
            radiance = radiance + throughput * C2 * w2;
        }
        if (!hit) break;                                                            // :327-329
        Real rr_prob = 1;
        if (num_vertices - 1 >= D.rr_depth) {                                       // :333-340
            rr_prob = rmin(maxc((1 / eta_scale) * throughput), Real(0.95));
            if (pcg_real(rng) > rr_prob) break;
        }
        ray = bsdf_ray;
        vertex = bsdf_vertex;
        throughput = throughput * (G * f) / (p2 * rr_prob);                          // :344
    }
    if (bounces_out) *bounces_out = bounces;
    if (shadow_out) *shadow_out = shadows;
    if (rays_out) *rays_out = rays;
    return radiance;
}

} // namespace

extern "C" {

OracleScene *oracle_scene_create(const GdptSceneDesc *desc, int use_bvh) {
    OracleScene *sc = new OracleScene();
    sc->desc = *desc;
    sc->use_bvh = use_bvh != 0;
    float lb[3], ub[3];
    for (int k = 0; k < 3; k++) { lb[k] = std::numeric_limits<float>::infinity(); ub[k] = -lb[k]; }
    for (int s = 0; s < desc->num_shapes; s++) {
        const GdptShape &sh = desc->shapes[s];
        if (sh.type == GDPT_SHAPE_TRIMESH) {
            for (int t = 0; t < sh.num_triangles; t++) {
                Tri tr; tr.shape_id = s; tr.prim_id = t;
                float v[3][3];
                for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) v[i][k] = (float)sh.positions[3 * sh.indices[3 * t + i] + k];
                for (int k = 0; k < 3; k++) { tr.v0[k] = v[0][k]; tr.e1[k] = v[1][k] - v[0][k]; tr.e2[k] = v[2][k] - v[0][k]; }
                sc->tris.push_back(tr);
            }
            // Embree's scene bounds cover the vertex buffer (fp32)
            for (int i = 0; i < sh.num_vertices; i++) for (int k = 0; k < 3; k++) {
                float p = (float)sh.positions[3 * i + k]; lb[k] = std::min(lb[k], p); ub[k] = std::max(ub[k], p);
            }
        }
    }
    for (int s = 0; s < desc->num_shapes; s++) {
        const GdptShape &sh = desc->shapes[s];
        if (sh.type == GDPT_SHAPE_SPHERE) {
            sc->sphere_shapes.push_back(s);
            for (int k = 0; k < 3; k++) { // sphere_bounds_func, src/shapes/sphere.inl:1-10 (double -> float store)
                lb[k] = std::min(lb[k], (float)(sh.center[k] - sh.radius)); ub[k] = std::max(ub[k], (float)(sh.center[k] + sh.radius));
            }
        }
    }
    // bounds sphere + epsilon, src/scene.cpp:29-33, src/scene.h:100-102
    V3 l{lb[0], lb[1], lb[2]}, u{ub[0], ub[1], ub[2]};
    Real radius = std::sqrt(distance_squared(u, l)) / 2;
    sc->bounds_radius = radius;
    sc->isect_eps = rmin(radius * Real(1e-5), Real(0.01));
    for (int i = 0; i < desc->num_images; i++) sc->mips.push_back(make_mip(desc->images[i]));
    if (sc->use_bvh) {
        int n = (int)sc->tris.size() + (int)sc->sphere_shapes.size();
        std::vector<int> ids(n);
        std::vector<float> cent(3 * (size_t)n), bb(6 * (size_t)n);
        for (int g = 0; g < n; g++) {
            ids[g] = g;
            prim_bounds(*sc, g, &bb[6 * g], &bb[6 * g + 3]);
            for (int k = 0; k < 3; k++) cent[3 * g + k] = 0.5f * (bb[6 * g + k] + bb[6 * g + 3 + k]);
        }
        if (n > 0) build_node(*sc, ids, 0, n, cent, bb);
    }
    // emitter sampling tables (Integrator::Path): init_sampling_dist per emitter mesh, then the light power table
    sc->mesh_tables.resize((size_t)desc->num_shapes);
    for (int sidx = 0; sidx < desc->num_shapes; sidx++) {
        const GdptShape &sh = desc->shapes[sidx];
        if (sh.type != GDPT_SHAPE_TRIMESH || sh.area_light_id < 0) continue;
        std::vector<Real> areas((size_t)sh.num_triangles, Real(0));
        Real total = 0;
        for (int t = 0; t < sh.num_triangles; t++) {
            const int *ix = sh.indices + 3 * t;
            auto P = [&](int i) { return V3{sh.positions[3 * ix[i]], sh.positions[3 * ix[i] + 1], sh.positions[3 * ix[i] + 2]}; };
            V3 v0 = P(0), e1 = P(1) - v0, e2 = P(2) - v0;
            areas[(size_t)t] = length(cross(e1, e2)) / 2;
            total += areas[(size_t)t];
        }
        make_table_1d(areas, sc->mesh_tables[(size_t)sidx].pmf, sc->mesh_tables[(size_t)sidx].cdf);
        sc->mesh_tables[(size_t)sidx].total_area = total;
    }
    if (desc->has_envmap) {                                                // init_sampling_dist, src/lights/envmap.inl:66-83
        const Mip &m = sc->mips[(size_t)desc->envmap.image_id];
        int w = m.w[0], h = m.h[0];
        std::vector<Real> f((size_t)w * h);
        size_t i = 0;
        for (int y = 0; y < h; y++) {
            Real v = (y + Real(0.5)) / Real(h);
            Real sin_elevation = std::sin(c_PI * v);
            for (int x = 0; x < w; x++) {
                Real uu = (x + Real(0.5)) / Real(w);
                f[i++] = luminance(mip_lookup(m, uu, v, 0)) * sin_elevation;
            }
        }
        make_table_2d(f, w, h, sc->env_table);
    }
    {
        std::vector<Real> power((size_t)desc->num_lights);
        for (int l = 0; l < desc->num_lights; l++) {                       // light_power, src/lights/diffuse_area_light.inl:1-3
            const GdptLight &lt = desc->lights[l];
            if (lt.shape_id < 0) {                                         // envmap.inl:1-5
                power[(size_t)l] = c_PI * sc->bounds_radius * sc->bounds_radius * sc->env_table.total_values / (sc->env_table.width * sc->env_table.height);
                continue;
            }
            power[(size_t)l] = luminance(V3{lt.intensity[0], lt.intensity[1], lt.intensity[2]}) * surface_area(*sc, lt.shape_id) * c_PI;
        }
        if (!power.empty()) make_table_1d(power, sc->light_pmf, sc->light_cdf);
    }
    return sc;
}
void oracle_scene_free(OracleScene *s) { delete s; }
double oracle_intersection_epsilon(const OracleScene *s) { return s->isect_eps; }

void oracle_pcg_init(uint64_t stream, uint64_t *state, uint64_t *inc) { Pcg p = pcg_init(stream); *state = p.state; *inc = p.inc; }
uint32_t oracle_pcg_next(uint64_t *state, uint64_t inc) { Pcg p{*state, inc}; uint32_t r = pcg_next(p); *state = p.state; return r; }
double oracle_pcg_next_double(uint64_t *state, uint64_t inc) { Pcg p{*state, inc}; double r = pcg_real(p); *state = p.state; return r; }

void oracle_sample_primary(const OracleScene *s, double sx, double sy, double org[3], double dir[3]) {
    Ray r = sample_primary(s->desc.camera, V2{sx, sy});
    for (int k = 0; k < 3; k++) { org[k] = r.org[k]; dir[k] = r.dir[k]; }
}
void oracle_filter_sample(int filter_type, double param, double u0, double u1, double out[2]) {
    V2 o = filter_sample(filter_type, param, V2{u0, u1}); out[0] = o.x; out[1] = o.y;
}

static void to_c(const Vertex &v, OracleVertex *o) {
    for (int k = 0; k < 3; k++) {
        o->position[k] = v.position[k]; o->geometric_normal[k] = v.geometric_normal[k];
        o->frame_x[k] = v.shading_frame.x[k]; o->frame_y[k] = v.shading_frame.y[k]; o->frame_n[k] = v.shading_frame.n[k];
    }
    o->st[0] = v.st.x; o->st[1] = v.st.y; o->uv[0] = v.uv.x; o->uv[1] = v.uv.y;
    o->uv_screen_size = v.uv_screen_size; o->mean_curvature = v.mean_curvature; o->ray_radius = v.ray_radius;
    o->shape_id = v.shape_id; o->primitive_id = v.primitive_id; o->material_id = v.material_id; o->gid = v.gid; o->t = v.t;
}
static Vertex from_c(const OracleVertex *o) {
    Vertex v;
    v.position = V3{o->position[0], o->position[1], o->position[2]};
    v.geometric_normal = V3{o->geometric_normal[0], o->geometric_normal[1], o->geometric_normal[2]};
    v.shading_frame = Frame{V3{o->frame_x[0], o->frame_x[1], o->frame_x[2]}, V3{o->frame_y[0], o->frame_y[1], o->frame_y[2]},
                            V3{o->frame_n[0], o->frame_n[1], o->frame_n[2]}};
    v.st = V2{o->st[0], o->st[1]}; v.uv = V2{o->uv[0], o->uv[1]};
    v.uv_screen_size = o->uv_screen_size; v.mean_curvature = o->mean_curvature; v.ray_radius = o->ray_radius;
    v.shape_id = o->shape_id; v.primitive_id = o->primitive_id; v.material_id = o->material_id; v.gid = o->gid; v.t = o->t;
    return v;
}

int oracle_intersect(const OracleScene *s, const double org[3], const double dir[3], double tnear, double tfar,
                     const double ray_diff[2], OracleVertex *out) {
    Ray r{V3{org[0], org[1], org[2]}, V3{dir[0], dir[1], dir[2]}, tnear, tfar};
    Vertex v;
    if (!intersect(*s, r, ray_diff ? ray_diff[0] : 0, ray_diff ? ray_diff[1] : 0, &v)) return 0;
    to_c(v, out);
    return 1;
}
void oracle_shading_info_tri(const OracleScene *s, int gid, const double st[2], const double gn[3], double out[13]) {
    const Tri &t = s->tris[gid];
    ShadingInfo si = shading_info_tri(s->desc.shapes[t.shape_id], t.prim_id, V2{st[0], st[1]}, V3{gn[0], gn[1], gn[2]});
    out[0] = si.uv.x; out[1] = si.uv.y;
    for (int k = 0; k < 3; k++) { out[2 + k] = si.frame.x[k]; out[5 + k] = si.frame.y[k]; out[8 + k] = si.frame.n[k]; }
    out[11] = si.mean_curvature; out[12] = si.inv_uv_size;
}

void oracle_bsdf_eval(const OracleScene *s, const GdptMaterial *m, const double dir_in[3], const double dir_out[3],
                      const OracleVertex *v, double f[3]) {
    V3 r = bsdf_eval(*s, *m, V3{dir_in[0], dir_in[1], dir_in[2]}, V3{dir_out[0], dir_out[1], dir_out[2]}, from_c(v));
    f[0] = r.x; f[1] = r.y; f[2] = r.z;
}
double oracle_bsdf_pdf(const OracleScene *s, const GdptMaterial *m, const double dir_in[3], const double dir_out[3], const OracleVertex *v) {
    return bsdf_pdf(*s, *m, V3{dir_in[0], dir_in[1], dir_in[2]}, V3{dir_out[0], dir_out[1], dir_out[2]}, from_c(v));
}
int oracle_bsdf_sample(const OracleScene *s, const GdptMaterial *m, const double dir_in[3], const OracleVertex *v,
                       const double rnd_uv[2], double rnd_w, double dir_out[3], double *eta, double *roughness) {
    BsdfSample bs;
    if (!bsdf_sample(*s, *m, V3{dir_in[0], dir_in[1], dir_in[2]}, from_c(v), V2{rnd_uv[0], rnd_uv[1]}, rnd_w, &bs)) return 0;
    for (int k = 0; k < 3; k++) dir_out[k] = bs.dir_out[k];
    *eta = bs.eta; *roughness = bs.roughness;
    return 1;
}
void oracle_texture_eval(const OracleScene *s, const GdptTexture *t, int channels, const double uv[2], double footprint, double out[3]) {
    (void)channels;
    V3 r = tex_eval3(*s, *t, V2{uv[0], uv[1]}, footprint);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

void oracle_grad_sample(const OracleScene *s, int x, int y, uint64_t *state, uint64_t inc, OracleSampleRecord *rec) {
    Pcg p{*state, inc};
    grad_sample(*s, x, y, p, rec);
    *state = p.state;
}

int oracle_render(const OracleScene *s, int spp, int rng_scheme, int row_begin, int row_end, int threads,
                  double *img, double *cx0, double *cy0, double *cx1, double *cy1, OracleStats *stats) {
    const OracleScene &sc = *s;
    int w = sc.desc.camera.width, h = sc.desc.camera.height;
    if (spp <= 0) spp = sc.desc.samples_per_pixel;
    if (row_begin == 0 && row_end == 0) row_end = h;
    if (rng_scheme != GDPT_RNG_TILE && rng_scheme != GDPT_RNG_SAMPLE) return 1;
    constexpr int tile_size = 16;                                              // src/render.cpp:271
    int ntx = (w + tile_size - 1) / tile_size, nty = (h + tile_size - 1) / tile_size;
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads <= 0) threads = 1;
    sc.nodes_visited = 0; sc.tris_tested = 0;
    std::atomic<int> next_tile{0};
    std::atomic<uint64_t> a_samples{0}, a_rays{0}, a_bounces{0}, a_miss{0}, a_x0{0}, a_nonfinite{0};
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&]() {
        uint64_t n_samples = 0, n_rays = 0, n_bounces = 0, n_miss = 0, n_x0 = 0, n_nf = 0;
        for (;;) {
            int tile = next_tile.fetch_add(1);
            if (tile >= ntx * nty) break;
            int tx = tile % ntx, ty = tile / ntx;
            Pcg rng = pcg_init((uint64_t)(ty * ntx + tx));                     // src/render.cpp:281
            int x0 = tx * tile_size, x1 = std::min(x0 + tile_size, w);
            int y0 = ty * tile_size, y1 = std::min(y0 + tile_size, h);
            for (int y = y0; y < y1; y++) {
                // a band render skips rows outside [row_begin,row_end); in TILE mode that would desynchronise the
                // stream, so TILE mode requires bands aligned to whole tile rows (checked by the caller/tests)
                if (y < row_begin || y >= row_end) continue;
                for (int x = x0; x < x1; x++) {
                    V3 r{0, 0, 0}, rdX0{0, 0, 0}, rdX1{0, 0, 0}, rdY0{0, 0, 0}, rdY1{0, 0, 0};
                    for (int sidx = 0; sidx < spp; sidx++) {
                        if (rng_scheme == GDPT_RNG_SAMPLE) rng = pcg_init(((uint64_t)y * w + x) * (uint64_t)spp + (uint64_t)sidx);
                        OracleSampleRecord R;
                        grad_sample(sc, x, y, rng, &R);
                        n_samples++; n_bounces += R.bounces; n_miss += R.primary_miss; n_x0 += R.valid0[0];
                        n_rays += (R.primary_miss ? 1 : 5) + R.bounces;
                        bool finite = std::isfinite(R.prob);
                        for (int c = 0; c < 3; c++) finite = finite && std::isfinite(R.radiance[c]) && std::isfinite(R.contrib[c]) &&
                            std::isfinite(R.contribX0[c]) && std::isfinite(R.contribX1[c]) && std::isfinite(R.contribY0[c]) && std::isfinite(R.contribY1[c]);
                        if (!finite) n_nf++;
                        if (R.prob > 0.0) {                                    // src/render.cpp:311-318
                            V3 rad{R.radiance[0], R.radiance[1], R.radiance[2]}, C{R.contrib[0], R.contrib[1], R.contrib[2]};
                            V3 CX0{R.contribX0[0], R.contribX0[1], R.contribX0[2]}, CX1{R.contribX1[0], R.contribX1[1], R.contribX1[2]};
                            V3 CY0{R.contribY0[0], R.contribY0[1], R.contribY0[2]}, CY1{R.contribY1[0], R.contribY1[1], R.contribY1[2]};
                            r = r + rad / Real(spp);
                            rdX0 = rdX0 + (C - CX0) * (R.wX0 / (R.prob * Real(spp)));
                            rdY0 = rdY0 + (C - CY0) * (R.wY0 / (R.prob * Real(spp)));
                            rdX1 = rdX1 + (CX1 - C) * (R.wX1 / (R.prob * Real(spp)));
                            rdY1 = rdY1 + (CY1 - C) * (R.wY1 / (R.prob * Real(spp)));
                        }
                    }
                    size_t i = ((size_t)y * w + x) * 3;
                    for (int c = 0; c < 3; c++) { img[i + c] += r[c]; cx0[i + c] += rdX0[c]; cy0[i + c] += rdY0[c]; cx1[i + c] += rdX1[c]; cy1[i + c] += rdY1[c]; }
                }
            }
        }
        a_samples += n_samples; a_rays += n_rays; a_bounces += n_bounces; a_miss += n_miss; a_x0 += n_x0; a_nonfinite += n_nf;
    };
    std::vector<std::thread> pool;
    for (int i = 1; i < threads; i++) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    if (stats) {
        stats->samples = a_samples; stats->rays = a_rays; stats->bounces = a_bounces; stats->primary_misses = a_miss;
        stats->x0_valid_initial = a_x0; stats->nonfinite_samples = a_nonfinite;
        stats->nodes_visited = sc.nodes_visited; stats->tris_tested = sc.tris_tested;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

void oracle_light_table(const OracleScene *s, double *pmf, double *cdf) {
    for (size_t i = 0; i < s->light_pmf.size(); i++) pmf[i] = s->light_pmf[i];
    for (size_t i = 0; i < s->light_cdf.size(); i++) cdf[i] = s->light_cdf[i];
}
void oracle_sample_point_on_shape(const OracleScene *s, int shape_id, const double ref_point[3], const double uv[2], double w, double out[6]) {
    PointNormal p = sample_point_on_shape(*s, shape_id, V3{ref_point[0], ref_point[1], ref_point[2]}, V2{uv[0], uv[1]}, w);
    for (int k = 0; k < 3; k++) { out[k] = p.position[k]; out[3 + k] = p.normal[k]; }
}
double oracle_pdf_point_on_shape(const OracleScene *s, int shape_id, const double point[3], const double normal[3], const double ref_point[3]) {
    PointNormal p{V3{point[0], point[1], point[2]}, V3{normal[0], normal[1], normal[2]}};
    return pdf_point_on_shape(*s, shape_id, p, V3{ref_point[0], ref_point[1], ref_point[2]});
}
int oracle_occluded(const OracleScene *s, const double org[3], const double dir[3], double tnear, double tfar) {
    Ray r{V3{org[0], org[1], org[2]}, V3{dir[0], dir[1], dir[2]}, tnear, tfar};
    return occluded(*s, r) ? 1 : 0;
}
void oracle_path_sample(const OracleScene *s, int x, int y, uint64_t *state, uint64_t inc, double radiance[3], int32_t *bounces, int32_t *shadow_rays) {
    Pcg rng{*state, inc};
    int b = 0, sh = 0;
    V3 r = path_sample(*s, x, y, rng, &b, &sh);
    *state = rng.state;
    radiance[0] = r.x; radiance[1] = r.y; radiance[2] = r.z;
    if (bounces) *bounces = b;
    if (shadow_rays) *shadow_rays = sh;
}

void oracle_table2d(const double *f, int width, int height, const double *rnd, int n_rnd, double *uv_out, double *pdf_out, double *total) {
    OracleScene::Table2D t;
    make_table_2d(std::vector<Real>(f, f + (size_t)width * height), width, height, t);
    *total = t.total_values;
    for (int i = 0; i < n_rnd; i++) {
        V2 uv = table2d_sample(t, V2{rnd[2 * i], rnd[2 * i + 1]});
        uv_out[2 * i] = uv.x; uv_out[2 * i + 1] = uv.y;
        pdf_out[i] = table2d_pdf(t, uv);
    }
}

int oracle_path_render(const OracleScene *s, int spp, int rng_scheme, int row_begin, int row_end, int threads,
                       double *img, OracleStats *stats) {
    const OracleScene &sc = *s;
    if (sc.desc.num_lights <= 0) return 2;
    int w = sc.desc.camera.width, h = sc.desc.camera.height;
    if (spp <= 0) spp = sc.desc.samples_per_pixel;
    if (row_begin == 0 && row_end == 0) row_end = h;
    if (rng_scheme != GDPT_RNG_TILE && rng_scheme != GDPT_RNG_SAMPLE) return 1;
    constexpr int tile_size = 16;                                              // src/render.cpp:83
    int ntx = (w + tile_size - 1) / tile_size, nty = (h + tile_size - 1) / tile_size;
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads <= 0) threads = 1;
    sc.nodes_visited = 0; sc.tris_tested = 0;
    std::atomic<int> next_tile{0};
    std::atomic<uint64_t> a_samples{0}, a_rays{0}, a_bounces{0}, a_nonfinite{0};
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&]() {
        uint64_t n_samples = 0, n_rays = 0, n_bounces = 0, n_nf = 0;
        for (;;) {
            int tile = next_tile.fetch_add(1);
            if (tile >= ntx * nty) break;
            int tx = tile % ntx, ty = tile / ntx;
            Pcg rng = pcg_init((uint64_t)(ty * ntx + tx));                     // src/render.cpp:94
            int x0 = tx * tile_size, x1 = std::min(x0 + tile_size, w);
            int y0 = ty * tile_size, y1 = std::min(y0 + tile_size, h);
            for (int y = y0; y < y1; y++) {
                if (y < row_begin || y >= row_end) continue;
                for (int x = x0; x < x1; x++) {
                    V3 radiance{0, 0, 0};
                    for (int sidx = 0; sidx < spp; sidx++) {
                        if (rng_scheme == GDPT_RNG_SAMPLE) rng = pcg_init(((uint64_t)y * w + x) * (uint64_t)spp + (uint64_t)sidx);
                        int b = 0, sh = 0, nr = 0;
                        V3 r = path_sample(sc, x, y, rng, &b, &sh, &nr);
                        radiance = radiance + r;                               // src/render.cpp:107-109
                        n_samples++; n_bounces += (uint64_t)b; n_rays += (uint64_t)nr;
                        if (!(std::isfinite(r.x) && std::isfinite(r.y) && std::isfinite(r.z))) n_nf++;
                    }
                    V3 px = radiance / Real(spp);                              // :110
                    size_t i = ((size_t)y * w + x) * 3;
                    for (int c = 0; c < 3; c++) img[i + c] += px[c];
                }
            }
        }
        a_samples += n_samples; a_rays += n_rays; a_bounces += n_bounces; a_nonfinite += n_nf;
    };
    std::vector<std::thread> pool;
    for (int i = 1; i < threads; i++) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    if (stats) {
        stats->samples = a_samples; stats->rays = a_rays; stats->bounces = a_bounces; stats->primary_misses = 0;
        stats->x0_valid_initial = 0; stats->nonfinite_samples = a_nonfinite;
        stats->nodes_visited = sc.nodes_visited; stats->tris_tested = sc.tris_tested;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

// ---- GDPT_SHIFT_RECONNECT (include/gdpt.h): the reconnection shift of the reference's sketch small_gdpt.py:163-219 with
// its estimator (:380-420), on LaJolla's scenes and base path. CPU restatement of csrc/hip/render_reconnect.hip, term by
// term (see the derivation in that file's header). PARITY UNPINNED against the reference: the reference never produces
// this output; the mode is pinned by its defining property instead (tests: gradient buffers converge to the finite
// differences of the converged image; the primal equals the reference mode's primal).
static void reconnect_sample(const OracleScene &sc, int x, int y, Pcg &rng, Real spp, V3 acc[5], uint64_t *rays, uint64_t *bounces) {
    const GdptCamera &cam = sc.desc.camera;
    const int w = cam.width, h = cam.height, max_depth = sc.desc.max_depth;
    auto allows = [&](int nv) { return max_depth == -1 || nv <= max_depth + 1; };
    const double rng_x = pcg_real(rng), rng_y = pcg_real(rng);
    Ray ray = sample_primary(cam, V2{(x + rng_x) / w, (y + rng_y) / h});
    const Real rd_spread = Real(0.25) / rmax(w, h);
    Vertex v1;
    (*rays)++;
    if (!intersect(sc, ray, 0, rd_spread, &v1)) return;
    const int ox[4] = {-1, +1, 0, 0}, oy[4] = {0, 0, -1, +1};
    Vertex ov[4]; V3 oview[4]; bool ok[4];
    for (int k = 0; k < 4; k++) {
        Ray r = sample_primary(cam, V2{((x + ox[k]) + rng_x) / w, ((y + oy[k]) + rng_y) / h});
        (*rays)++;
        ok[k] = intersect(sc, r, 0, rd_spread, &ov[k]);
        oview[k] = -r.dir;
    }
    const int slot[4] = {1, 3, 2, 4};
    const Real sgn[4] = {1, -1, 1, -1};
    auto add_term = [&](int k, const V3 &f, const V3 &fo, Real wgt) { acc[slot[k]] = acc[slot[k]] + (f - fo) * (sgn[k] * wgt / spp); };
    auto Le = [&](const Vertex &v, const V3 &view) { return is_light(sc, v.shape_id) ? emission(sc, v, view) : V3{0, 0, 0}; };

    const V3 view1 = -ray.dir;
    V3 radiance = Le(v1, view1);
    for (int k = 0; k < 4; k++) {
        if (!ok[k]) add_term(k, radiance, V3{0, 0, 0}, 1);
        else add_term(k, radiance, Le(ov[k], oview[k]), Real(0.5));
    }
    auto finish = [&]() { acc[0] = acc[0] + radiance / spp; };
    if (!allows(3)) { finish(); return; }

    (*bounces)++;
    V2 ruv; ruv.x = pcg_real(rng); ruv.y = pcg_real(rng);
    Real rw = pcg_real(rng);
    BsdfSample bs;
    Real eta_scale = 1;
    const GdptMaterial &mat1 = sc.desc.materials[v1.material_id];
    if (!bsdf_sample(sc, mat1, view1, v1, ruv, rw, &bs)) { finish(); return; }
    const V3 w1 = bs.dir_out;
    if (bs.eta != 0) eta_scale /= (bs.eta * bs.eta);
    const V3 f1 = bsdf_eval(sc, mat1, view1, w1, v1);
    const Real p1 = bsdf_pdf(sc, mat1, view1, w1, v1);
    Vertex v2;
    (*rays)++;
    const bool hit2 = intersect(sc, Ray{v1.position, w1, sc.isect_eps, std::numeric_limits<Real>::infinity()}, 0, 0, &v2);
    if (!(p1 > 0) || !hit2) { finish(); return; }
    const V3 A1 = f1 / p1;
    V3 throughput{1, 1, 1};
    Real rr_all = 1;
    if (3 - 1 >= sc.desc.rr_depth) {
        const Real rr_prob = rmin(maxc((1 / eta_scale) * throughput), Real(0.95));
        if (pcg_real(rng) > rr_prob) rr_all = 0; else rr_all /= rr_prob;
    }
    throughput = A1 * rr_all;

    const V3 d12 = v2.position - v1.position;
    const Real dist2 = dot(d12, d12), cos2 = std::fabs(dot(w1, v2.geometric_normal));
    V3 fo1[4], wo1[4]; Real ro1[4]; bool rec[4];
    for (int k = 0; k < 4; k++) {
        rec[k] = false; fo1[k] = V3{0, 0, 0}; ro1[k] = 0; wo1[k] = w1;
        if (!ok[k] || !(cos2 > 0)) continue;
        const Vertex &o = ov[k];
        const V3 d = v2.position - o.position;
        const Real od2 = dot(d, d);
        if (!(od2 > 0)) continue;
        const Real od = std::sqrt(od2);
        const V3 wo = d / od;
        const Real ocos2 = std::fabs(dot(wo, v2.geometric_normal));
        if (!(ocos2 > 0) || dot(wo, v2.geometric_normal) * dot(w1, v2.geometric_normal) <= 0) continue;
        const GdptMaterial &omat = sc.desc.materials[o.material_id];
        const V3 f = bsdf_eval(sc, omat, oview[k], wo, o);
        const Real p = bsdf_pdf(sc, omat, oview[k], wo, o);
        if (!(p > 0)) continue;
        (*rays)++;
        if (occluded(sc, Ray{o.position, wo, sc.isect_eps, (1 - sc.isect_eps) * od})) continue;
        const Real J = (ocos2 / od2) / (cos2 / dist2);
        rec[k] = true; wo1[k] = wo;
        fo1[k] = f * (J / p1); ro1[k] = p * J / p1;
    }
    if (is_light(sc, v2.shape_id)) {
        const V3 Le2 = emission(sc, v2, -w1);
        radiance = radiance + A1 * Le2;
        for (int k = 0; k < 4; k++) {
            if (!rec[k]) add_term(k, A1 * Le2, V3{0, 0, 0}, 1);
            else add_term(k, A1 * Le2, fo1[k] * emission(sc, v2, -wo1[k]), 1 / (1 + ro1[k]));
        }
    }
    if (rr_all == 0 || !allows(4)) { finish(); return; }

    (*bounces)++;
    ruv.x = pcg_real(rng); ruv.y = pcg_real(rng); rw = pcg_real(rng);
    const GdptMaterial &mat2 = sc.desc.materials[v2.material_id];
    const V3 view2 = -w1;
    if (!bsdf_sample(sc, mat2, view2, v2, ruv, rw, &bs)) { finish(); return; }
    const V3 w2 = bs.dir_out;
    if (bs.eta != 0) eta_scale /= (bs.eta * bs.eta);
    const V3 f2 = bsdf_eval(sc, mat2, view2, w2, v2);
    const Real p2 = bsdf_pdf(sc, mat2, view2, w2, v2);
    if (!(p2 > 0)) { finish(); return; }
    const V3 B2 = f2 / p2;
    V3 fo2[4]; Real ro2[4];
    for (int k = 0; k < 4; k++) {
        fo2[k] = V3{0, 0, 0}; ro2[k] = 0;
        if (!rec[k]) continue;
        const V3 f = bsdf_eval(sc, mat2, -wo1[k], w2, v2);
        const Real p = bsdf_pdf(sc, mat2, -wo1[k], w2, v2);
        if (!(p > 0)) { rec[k] = false; continue; }
        fo2[k] = fo1[k] * (f / p2); ro2[k] = ro1[k] * (p / p2);
    }
    V3 U = splat(rr_all), S{0, 0, 0}, pendT = B2, pendU{1, 1, 1};
    Ray cur{v2.position, w2, sc.isect_eps, std::numeric_limits<Real>::infinity()};
    Vertex vertex;
    (*rays)++;
    bool hit = intersect(sc, cur, 0, 0, &vertex);
    for (int nv = 4; hit;) {
        if (is_light(sc, vertex.shape_id)) S = S + U * pendU * emission(sc, vertex, -cur.dir);
        Real rr_prob = 1;
        if (nv - 1 >= sc.desc.rr_depth) {
            rr_prob = rmin(maxc((1 / eta_scale) * throughput), Real(0.95));
            if (pcg_real(rng) > rr_prob) break;
        }
        throughput = throughput * pendT / rr_prob; U = U * pendU / rr_prob;
        nv++;
        if (!allows(nv)) break;
        (*bounces)++;
        const GdptMaterial &mat = sc.desc.materials[vertex.material_id];
        const V3 view = -cur.dir;
        ruv.x = pcg_real(rng); ruv.y = pcg_real(rng); rw = pcg_real(rng);
        if (!bsdf_sample(sc, mat, view, vertex, ruv, rw, &bs)) break;
        if (bs.eta != 0) eta_scale /= (bs.eta * bs.eta);
        const V3 f = bsdf_eval(sc, mat, view, bs.dir_out, vertex);
        const Real p = bsdf_pdf(sc, mat, view, bs.dir_out, vertex);
        if (!(p > 0)) break;
        pendT = f / p; pendU = pendT;
        cur = Ray{vertex.position, bs.dir_out, sc.isect_eps, std::numeric_limits<Real>::infinity()};
        (*rays)++;
        hit = intersect(sc, cur, 0, 0, &vertex);
    }
    const V3 base3 = A1 * B2 * S;
    radiance = radiance + base3;
    for (int k = 0; k < 4; k++) {
        if (!rec[k]) add_term(k, base3, V3{0, 0, 0}, 1);
        else add_term(k, base3, fo2[k] * S, 1 / (1 + ro2[k]));
    }
    finish();
}

int oracle_reconnect_render(const OracleScene *s, int spp, int row_begin, int row_end, int threads,
                            double *img, double *cx0, double *cy0, double *cx1, double *cy1, OracleStats *stats) {
    const OracleScene &sc = *s;
    int w = sc.desc.camera.width, h = sc.desc.camera.height;
    if (spp <= 0) spp = sc.desc.samples_per_pixel;
    if (row_begin == 0 && row_end == 0) row_end = h;
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads <= 0) threads = 1;
    std::atomic<int> next_row{row_begin};
    std::atomic<uint64_t> a_samples{0}, a_rays{0}, a_bounces{0};
    auto t0 = std::chrono::steady_clock::now();
    double *out[5] = {img, cx0, cy0, cx1, cy1};
    auto worker = [&]() {
        uint64_t n_samples = 0, n_rays = 0, n_bounces = 0;
        for (;;) {
            int y = next_row.fetch_add(1);
            if (y >= row_end) break;
            for (int x = 0; x < w; x++) {
                V3 acc[5] = {};
                for (int sidx = 0; sidx < spp; sidx++) {
                    Pcg rng = pcg_init(((uint64_t)y * w + x) * (uint64_t)spp + (uint64_t)sidx);
                    reconnect_sample(sc, x, y, rng, Real(spp), acc, &n_rays, &n_bounces);
                    n_samples++;
                }
                size_t i = ((size_t)y * w + x) * 3;
                for (int b = 0; b < 5; b++) for (int c = 0; c < 3; c++) out[b][i + c] += acc[b][c];
            }
        }
        a_samples += n_samples; a_rays += n_rays; a_bounces += n_bounces;
    };
    std::vector<std::thread> pool;
    for (int i = 1; i < threads; i++) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->samples = a_samples; stats->rays = a_rays; stats->bounces = a_bounces;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

void oracle_assemble(int w, int h, const double *img, const double *cx0, const double *cy0,
                     const double *cx1, const double *cy1, double *c, double *cx, double *cy) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int ch = 0; ch < 3; ch++) {
                size_t i = ((size_t)y * w + x) * 3 + ch;
                c[i] = img[i];
                cx[i] = (x == 0) ? cx0[i] : cx0[i] + cx1[i - 3];
                cy[i] = (y == 0) ? cy0[i] : cy0[i] + cy1[i - (size_t)w * 3];
            }
}

void oracle_poisson_dct(int width, int height, const double *imgData, const double *imgGradX, const double *imgGradY, double dataCost, double *imgOut) {
    int nodeCount = width * height;
    std::vector<double> buf(nodeCount), tmp(nodeCount), ftLapY(height), ftLapX(width), line(std::max(width, height));
    for (int x = 0; x < width; x++) ftLapX[x] = 2.0 * std::cos(M_PI * x / (width - 1));
    for (int y = 0; y < height; y++) ftLapY[y] = -4.0 + (2.0 * std::cos(M_PI * y / (height - 1)));
    std::vector<double> tabx(2 * (size_t)(width - 1)), taby(2 * (size_t)(height - 1));
    for (size_t i = 0; i < tabx.size(); i++) tabx[i] = std::cos(M_PI * (double)i / (width - 1));
    for (size_t i = 0; i < taby.size(); i++) taby[i] = std::cos(M_PI * (double)i / (height - 1));
    auto dct2d = [&]() {
        for (int y = 0; y < height; y++) { dct1(tabx, width, &buf[(size_t)y * width], 1, line.data()); std::copy(line.begin(), line.begin() + width, &tmp[(size_t)y * width]); }
        for (int x = 0; x < width; x++) { dct1(taby, height, &tmp[x], width, line.data()); for (int y = 0; y < height; y++) buf[(size_t)y * width + x] = line[y]; }
    };
    for (int ch = 0; ch < 3; ch++) {
        double dcSum = 0.0;
        for (int y = 0; y < height; y++)
            for (int x = 0; x < width; x++) {
                size_t node = (size_t)y * width + x, p = node * 3 + ch, right = p + 3, top = p + (size_t)width * 3;
                double dcMult = 1.0;
                if ((x > 0) && (x < width - 1)) dcMult *= 2.0;
                if ((y > 0) && (y < height - 1)) dcMult *= 2.0;
                dcSum += dcMult * imgData[p];
                buf[node] = dataCost * imgData[p];
                if ((x > 0) && (x < width - 1)) buf[node] -= (imgGradX[right] - imgGradX[p]); else buf[node] -= (-2.0 * imgGradX[p]);
                if ((y > 0) && (y < height - 1)) buf[node] -= (imgGradY[top] - imgGradY[p]); else buf[node] -= (-2.0 * imgGradY[p]);
            }
        dct2d();
        for (int y = 0; y < height; y++)
            for (int x = 0; x < width; x++) {
                float ftLapResponse = ftLapY[y] + ftLapX[x];                    // fp32 rounding, src/render.cpp:233
                buf[(size_t)y * width + x] /= (dataCost - ftLapResponse);
            }
        buf[0] = dcSum;
        dct2d();
        double fftDenom = 4.0 * (width - 1) * (height - 1);
        for (int i = 0; i < nodeCount; i++) imgOut[(size_t)i * 3 + ch] = buf[i] / fftDenom;
    }
}

} // extern "C"
