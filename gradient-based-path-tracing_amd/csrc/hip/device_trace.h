// device_trace.h — closest-hit BVH2 traversal with an LDS stack, shading-info reconstruction and the
// camera for the HIP kernels. Replaces Embree behind intersect() (src/intersection.cpp:7-65).
//
// Closest hit is DEFINED as: minimum fp32 t over all primitives whose test accepts the ray
// (tnear <= t < tfar), ties broken by the lowest global primitive id. That makes the result
// independent of traversal order, so this BVH walk, the oracle's brute force and the oracle's own
// BVH must all return the same hit; box tests only have to be conservative.
#pragma once
#include "device_bsdf.h"

namespace gd {

struct Hit { float t, u, v, ngx, ngy, ngz; int gid; };   // gid < 0: miss

struct TraceCounters { unsigned nodes, prims, node_trips, leaf_trips, wave_steps, lane_steps; };   // per lane; summed in 64 bits across the wave
// counting builds: one lane of the currently active set ticks a per-wave event
GD bool wave_leader() { return (int)(threadIdx.x & 63) == __ffsll((unsigned long long)__ballot(1)) - 1; }

// fp32 Moller-Trumbore, two-sided; every operation rounds once (no FMA contraction), sums left to right.
// The CPU checker used by the tests restates exactly this arithmetic.
GD bool tri_hit(const float o[3], const float d[3], float tnear, float tfar, const DevPrim &tr, float &t, float &u, float &v) {
#pragma clang fp contract(off)
    const float *e1 = tr.e1, *e2 = tr.e2;
    float px = d[1] * e2[2] - d[2] * e2[1];
    float py = d[2] * e2[0] - d[0] * e2[2];
    float pz = d[0] * e2[1] - d[1] * e2[0];
    float det = e1[0] * px + e1[1] * py + e1[2] * pz;
    if (!(det != 0.0f)) return false;
    float inv = 1.0f / det;
    float sx = o[0] - tr.v0[0], sy = o[1] - tr.v0[1], sz = o[2] - tr.v0[2];
    u = (sx * px + sy * py + sz * pz) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    float qx = sy * e1[2] - sz * e1[1];
    float qy = sz * e1[0] - sx * e1[2];
    float qz = sx * e1[1] - sy * e1[0];
    v = (d[0] * qx + d[1] * qy + d[2] * qz) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    t = (e2[0] * qx + e2[1] * qy + e2[2] * qz) * inv;
    return (t >= tnear && t < tfar);
}

// Sphere primitive: fp64 quadratic on the fp32 ray (src/shapes/sphere.inl:15-106).
// Kept out of line: spheres are rare (an emitter or two), and inlining this fp64 block into the traversal loop would
// raise the register pressure of every triangle-only ray. Arguments and result travel BY VALUE (registers): pointer
// parameters would force the caller's ray arrays into scratch memory, with a scratch store per traced ray in every
// scene, spheres or not, and a vmcnt(0) wait behind them wherever the compiler has to protect the stored registers.
struct SphereHit { float t, u, v, ngx, ngy, ngz; int ok; };
__device__ __noinline__ SphereHit sphere_hit(float o0, float o1, float o2, float d0, float d1, float d2, float tnear, float tfar,
                                             double cx, double cyc, double cz, double radius) {
#pragma clang fp contract(off)
    SphereHit h; h.ok = 0; h.t = h.u = h.v = h.ngx = h.ngy = h.ngz = 0.0f;
    double ox = o0 - cx, oy = o1 - cyc, oz = o2 - cz;
    double dx = d0, dy = d1, dz = d2;
    double A = dx * dx + dy * dy + dz * dz;
    double B = 2 * (dx * ox + dy * oy + dz * oz);
    double C = (ox * ox + oy * oy + oz * oz) - radius * radius;
    double t0, t1;
    if (A == 0) {
        if (B == 0) return h;
        t0 = t1 = -C / B;
    } else {
        double disc = B * B - 4 * A * C;
        if (disc < 0) return h;
        double rd = sqrt(disc);
        if (B >= 0) { t0 = (-B - rd) / (2 * A); t1 = 2 * C / (-B - rd); }
        else { t0 = 2 * C / (-B + rd); t1 = (-B + rd) / (2 * A); }
    }
    if (t0 > t1) { double tmp = t0; t0 = t1; t1 = tmp; }
    double rn = tnear, rf = tfar, t = -1;
    if (t0 >= rn && t0 < rf) t = t0;
    if (t1 >= rn && t1 < rf && t < 0) t = t1;
    if (!(t >= rn && t < rf)) return h;
    double gx = ((double)o0 + t * dx) - cx, gy = ((double)o1 + t * dy) - cyc, gz = ((double)o2 + t * dz) - cz;
    double inv_r = 1.0 / radius;
    double cy = fmin(fmax(gy * inv_r, -1.0), 1.0);
    double elevation = acos(cy);
    double azimuth = atan2(gz * inv_r, gx * inv_r);
    h.ngx = (float)gx; h.ngy = (float)gy; h.ngz = (float)gz;
    h.u = (float)(azimuth / kTwoPi); h.v = (float)(elevation / kPi);
    h.t = (float)t;
    h.ok = 1;
    return h;
}

// SPHERES = false: the scene holds triangles only (decided at upload), so the sphere arm is not even compiled in.
template <bool SPHERES = true>
GD void test_prim(const DevSceneView &sv, const DevPrim &pr, const float o[3], const float d[3], float tnear, float tfar, Hit &best) {
    unsigned gid = pr.gid;
    if (!SPHERES || !(gid & GDPT_SPHERE_FLAG)) {
        float t, u, v;
        if (!tri_hit(o, d, tnear, tfar, pr, t, u, v)) return;
        if (best.gid >= 0 && !(t < best.t || (t == best.t && (int)gid < best.gid))) return;
        best.t = t; best.u = u; best.v = v; best.gid = (int)gid;   // Ng: per-triangle constant, see DevTriShade::gn
    } else {
        const DevSphere &sp = sv.spheres[gid & ~GDPT_SPHERE_FLAG];
        const SphereHit sh = sphere_hit(o[0], o[1], o[2], d[0], d[1], d[2], tnear, tfar, sp.center[0], sp.center[1], sp.center[2], sp.radius);
        if (!sh.ok) return;
        int sg = sv.num_tris + (int)(gid & ~GDPT_SPHERE_FLAG);   // spheres order after all triangles
        if (best.gid >= 0 && !(sh.t < best.t || (sh.t == best.t && sg < best.gid))) return;
        best.t = sh.t; best.u = sh.u; best.v = sh.v; best.ngx = sh.ngx; best.ngy = sh.ngy; best.ngz = sh.ngz; best.gid = sg;
    }
}

// The same test without control flow (the arithmetic of tri_hit, operation for operation): in a 64-wide wave some lane
// passes every early-out anyway, so the branches only cost exec-mask bookkeeping. `valid`: the record belongs to the
// lane's leaf. Rejections by NaN (det = 0 gives inv = inf) fall out of the comparisons exactly as in tri_hit.
GD void test_tri_flat(const DevPrim &tr, const float o[3], const float d[3], float tnear, float tfar, bool valid, Hit &best) {
#pragma clang fp contract(off)
    const float *e1 = tr.e1, *e2 = tr.e2;
    float px = d[1] * e2[2] - d[2] * e2[1];
    float py = d[2] * e2[0] - d[0] * e2[2];
    float pz = d[0] * e2[1] - d[1] * e2[0];
    float det = e1[0] * px + e1[1] * py + e1[2] * pz;
    float inv = 1.0f / det;
    float sx = o[0] - tr.v0[0], sy = o[1] - tr.v0[1], sz = o[2] - tr.v0[2];
    float u = (sx * px + sy * py + sz * pz) * inv;
    float qx = sy * e1[2] - sz * e1[1];
    float qy = sz * e1[0] - sx * e1[2];
    float qz = sx * e1[1] - sy * e1[0];
    float v = (d[0] * qx + d[1] * qy + d[2] * qz) * inv;
    float t = (e2[0] * qx + e2[1] * qy + e2[2] * qz) * inv;
    const int gid = (int)tr.gid;
    bool ok = valid & (det != 0.0f) & (u >= 0.0f) & (u <= 1.0f) & (v >= 0.0f) & (u + v <= 1.0f) & (t >= tnear) & (t < tfar);
    ok = ok & ((best.gid < 0) | (t < best.t) | ((t == best.t) & (gid < best.gid)));
    best.t = ok ? t : best.t; best.u = ok ? u : best.u; best.v = ok ? v : best.v; best.gid = ok ? gid : best.gid;
}

// Slab test against boxes the host has already widened (gdpt_scene_upload pads every child box by P = 1e-6 E, E = the
// largest absolute coordinate of the scene bounds and the camera position, i.e. of every possible ray origin).
// Distances are formed as fma(bound, 1/d, -o/d). With e = 2^-24: 1/d and o/d round once each and the fma once, so a
// computed distance is off by at most e (2|t| + |o/d|) = e (2|bound-o| + |o|)/|d| <= 5 e E/|d| = 3e-7 E/|d|, less than
// the P/|d| by which the padding moved the plane. The computed interval therefore always contains the true interval
// of the unpadded box, and no relative slack is needed: a box holding a primitive that reports t <= tbest is never
// rejected. Axes with d = 0 (or 1/d overflowing) produce NaN and drop out of fmin/fmax.
GD bool box_hit(const float *mn, const float *mx, const float oi[3], const float inv[3], float tnear, float tbest, float &tin) {
    float t0 = tnear, t1 = tbest;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float a = fmaf(mn[k], inv[k], -oi[k]), b = fmaf(mx[k], inv[k], -oi[k]);
        t0 = fmaxf(t0, fminf(a, b)); t1 = fminf(t1, fmaxf(a, b));       // NaN (inf - inf, 0*inf) is dropped by fmin/fmax
    }
    tin = t0;
    return t0 <= t1;
}

// ---- intersect() post-processing: src/intersection.cpp:37-63 + compute_shading_info ----------------
struct Ray { D3 org, dir; double tnear, tfar; };

GD void shading_info_tri(const DevTriShade &ts, D2 st, D3 gn, bool need_uv, D2 &uv, Frame &frame, double &inv_uv_size) {
    // src/shapes/triangle_mesh.inl:77-169. dp/du, dp/dv and max(|dpdu|,|dpdv|) do not depend on the hit point and
    // come precomputed (host, same formulas); mean curvature is never read on the GradPath path: skipped.
    double b0 = 1 - st.x - st.y;
    if (need_uv) {
        uv.x = b0 * ts.uv[0][0] + st.x * ts.uv[1][0] + st.y * ts.uv[2][0];
        uv.y = b0 * ts.uv[0][1] + st.x * ts.uv[1][1] + st.y * ts.uv[2][1];
    } else { uv.x = uv.y = 0; }
    inv_uv_size = ts.inv_uv_size;
    if (ts.flat_frame) {
        // identical vertex normals: the frame is a per-triangle constant (host-evaluated at the barycentre; the
        // reference's per-hit value differs from it only by the rounding of (1-s-t)*n + s*n + t*n)
        frame.x = mk(ts.n[0][0], ts.n[0][1], ts.n[0][2]); frame.y = mk(ts.n[1][0], ts.n[1][1], ts.n[1][2]); frame.n = mk(ts.n[2][0], ts.n[2][1], ts.n[2][2]);
        return;
    }
    D3 dpdu = mk(ts.dpdu[0], ts.dpdu[1], ts.dpdu[2]);
    D3 sn = gn;
    if (ts.has_normals) {
        D3 n0 = mk(ts.n[0][0], ts.n[0][1], ts.n[0][2]), n1 = mk(ts.n[1][0], ts.n[1][1], ts.n[1][2]), n2 = mk(ts.n[2][0], ts.n[2][1], ts.n[2][2]);
        sn = normalize(b0 * n0 + st.x * n1 + st.y * n2);
    }
    D3 tangent = normalize(dpdu - sn * dot(sn, dpdu));
    D3 bitangent = normalize(cross(sn, tangent));
    frame.x = tangent; frame.y = bitangent; frame.n = sn;
}

GD void shading_info_sphere(const DevSphere &sp, D2 st, D3 gn, D2 &uv, Frame &frame, double &inv_uv_size) {
    // src/shapes/sphere.inl:243-268 (st used as radians although it is in [0,1] units — kept)
    double r = sp.radius;
    double su, cu, sv_, cv;
    sincos(st.x, &su, &cu);
    sincos(st.y, &sv_, &cv);
    D3 dpdu = mk(-r * su * sv_, r * cu * sv_, 0.0);
    D3 dpdv = mk(r * cu * cv, r * su * cv, -r * sv_);
    D3 tangent = normalize(dpdu - gn * dot(gn, dpdu));
    frame.x = tangent; frame.y = normalize(cross(gn, tangent)); frame.n = gn;
    uv = st;
    inv_uv_size = (length(dpdu) + length(dpdv)) / 2;
}

// Builds the PathVertex of a hit. rd_spread/rd_radius: RayDifferential of the query (src/ray.h:26-40).
// `tris` is the shading table (HBM, or the block's LDS copy); `need_uv`: some texture is not constant.
// PLAIN (what the scene does not contain, decided at upload): bit 0 = no spheres, bit 1 = every texture constant (uv and
// the footprint are unobservable). The code for what is absent is not compiled in.
constexpr int kPlainNoSpheres = 1, kPlainConstTex = 2, kPlainBoth = 3;
constexpr int kPlainSetShift = 4;      // bits above: the material set a one-sided lane machine was built for (0 = every lobe)
template <int PLAIN = 0>
GD void make_vertex(const DevSceneView &sv, const DevTriShade *tris, bool need_uv, const Ray &ray, const Hit &h,
                    double rd_radius, double rd_spread, Vertex &v) {
    v.position = ray.org + ray.dir * (double)h.t;
    D2 st; st.x = (double)h.u; st.y = (double)h.v;
    double inv_uv_size;
    D3 gn;
    if ((PLAIN & kPlainNoSpheres) || h.gid < sv.num_tris) {
        const DevTriShade &ts = tris[h.gid];
        v.material_id = ts.material_id; v.light_id = ts.light_id;
        gn = mk(ts.gn[0], ts.gn[1], ts.gn[2]);
        shading_info_tri(ts, st, gn, !(PLAIN & kPlainConstTex) && need_uv, v.uv, v.frame, inv_uv_size);
    } else {
        const DevSphere &sp = sv.spheres[h.gid - sv.num_tris];
        v.material_id = sp.material_id; v.light_id = sp.light_id;
        gn = normalize(mk((double)h.ngx, (double)h.ngy, (double)h.ngz));
        shading_info_sphere(sp, st, gn, v.uv, v.frame, inv_uv_size);
    }
    if (!(PLAIN & kPlainConstTex) && need_uv && rd_spread != 0.0) {
        D3 dlt = ray.org - v.position;
        double dist = sqrt(dot(dlt, dlt));
        double ray_radius = rd_radius + rd_spread * dist;
        v.uv_screen_size = ray_radius / inv_uv_size;
    } else {
        v.uv_screen_size = 0.0;     // secondary rays carry the default RayDifferential{0,0} (src/path_tracing.h:564)
    }
    if (dot(gn, v.frame.n) < 0) gn = -gn;
    v.gn = gn;
}

// emission(), src/intersection.cpp:87-98 + src/lights/diffuse_area_light.inl:15-20
// `lights`: the intensity table (3 per area light) — HBM, or the block's LDS copy (TraceCtx::lights)
GD D3 emission(const double *lights, const Vertex &v, D3 view_dir) {
    if (dot(v.gn, view_dir) <= 0) return splat(0);
    const double *L = lights + 3 * v.light_id;
    return mk(L[0], L[1], L[2]);
}
GD D3 emission(const DevSceneView &sv, const Vertex &v, D3 view_dir) { return emission(sv.light_intensity, v, view_dir); }

// ---- camera, src/camera.cpp:23-47 + src/filters/*.inl ----------------------------------------------
GD D2 filter_sample(int type, double param, double rx, double ry) {
    D2 o;
    if (type == GDPT_FILTER_BOX) { o.x = (2 * rx - 1) * (param / 2); o.y = (2 * ry - 1) * (param / 2); }
    else if (type == GDPT_FILTER_GAUSSIAN) {
        double r = param * sqrt(-2 * log(fmax(rx, 1e-8)));
        double s, c;
        sincospi(2.0 * ry, &s, &c);
        o.x = r * c; o.y = r * s;
    } else {
        double h = param / 2;
        o.x = rx < 0.5 ? h * (sqrt(2 * rx) - 1) : h * (1 - sqrt(1 - 2 * (rx - 0.5)));
        o.y = ry < 0.5 ? h * (sqrt(2 * ry) - 1) : h * (1 - sqrt(1 - 2 * (ry - 0.5)));
    }
    return o;
}
GD D3 xform_point(const double *m, D3 p) {
    double tx = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    double ty = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    double tz = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    double tw = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    double inv_w = 1.0 / tw;
    return mk(tx * inv_w, ty * inv_w, tz * inv_w);
}
GD D3 xform_vector(const double *m, D3 v) {
    return mk(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
// `cache` (optional): the sub-pixel numbers (dx,dy) and filter offset of the lane's base ray. An offset ray whose
// (dx,dy) are bit-identical (always, for power-of-two film sizes) reuses the offset instead of re-evaluating the filter.
struct FilterCache { double dx, dy, ox, oy; };
// PIXEL_SPACE: (sx, sy) are the pixel-space numbers (x + u, y + v) the caller would otherwise divide by the film extent.
// For power-of-two films that division, the multiplication back (src/camera.cpp:26-27) and the division of the filtered
// position are exact scalings by 2^-k: the fast path skips them and gets the reference's bits with four fp64 divisions less.
template <bool PIXEL_SPACE = false>
GD Ray sample_primary(const DevCamera &cam, double sx, double sy, FilterCache *cache = nullptr, bool store = false) {
    const bool exact = PIXEL_SPACE && cam.pow2_film != 0;        // wave-uniform
    if (PIXEL_SPACE && !exact) { sx = sx / cam.width; sy = sy / cam.height; }
    double ppx = exact ? sx : sx * cam.width, ppy = exact ? sy : sy * cam.height;
    double fx = floor(ppx), fy = floor(ppy);
    D2 off;
    if (cache && !store && cache->dx == ppx - fx && cache->dy == ppy - fy) { off.x = cache->ox; off.y = cache->oy; }
    else {
        off = filter_sample(cam.filter_type, cam.filter_param, ppx - fx, ppy - fy);
        if (cache && store) { cache->dx = ppx - fx; cache->dy = ppy - fy; cache->ox = off.x; cache->oy = off.y; }
    }
    double rx, ry;
    if (exact) { rx = (fx + 0.5 + off.x) * cam.inv_width; ry = (fy + 0.5 + off.y) * cam.inv_height; }
    else { rx = (fx + 0.5 + off.x) / cam.width; ry = (fy + 0.5 + off.y) / cam.height; }
    D3 pt = xform_point(cam.sample_to_cam, mk(rx, ry, 0.0));
    D3 dir = normalize(pt);
    Ray r;
    r.org = mk(cam.org[0], cam.org[1], cam.org[2]);
    r.dir = normalize(xform_vector(cam.cam_to_world, dir));
    r.tnear = 0; r.tfar = __builtin_huge_val();
    return r;
}

} // namespace gd
