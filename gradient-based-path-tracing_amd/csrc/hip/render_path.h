// render_path.h — Integrator::Path on gfx950: path_tracing (src/path_tracing.h:13-348) and the
// path_render tile loop (src/render.cpp:74-117) for scenes lit by area emitters (triangle meshes, spheres).
//
// Unidirectional path tracing with next-event estimation and power-heuristic MIS. Reproduced as the reference computes
// it, including: the emitter-hit term of the BSDF-sampled ray is added WITHOUT its MIS weight w2 (:303-306 computes w2
// and drops it — only the environment-map branch applies it); Russian roulette from rr_depth on throughput/eta_scale;
// the sub-pixel numbers are drawn x first (left-to-right evaluation of the constructor arguments at :21-22, the order
// of the compiler the reference was developed with). Environment maps are not restated: gdpt_path_render refuses such
// scenes. Shares traversal, vertex reconstruction, BSDFs and textures with the GradPath kernels (render_device.h);
// shadow rays are any-hit walks of the same BVH4.
#pragma once
#include "render_device.h"

namespace gd {

// sample(TableDist1D), src/table_dist.cpp:27-33: std::upper_bound over cdf[0..size], minus one, clamped.
GD int table_sample(const double *cdf, int size, double u) {
    int lo = 0, hi = size + 1;              // first index in [0, size+1) whose entry is > u
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] > u) hi = mid; else lo = mid + 1;
    }
    const int off = lo - 1;
    return off < 0 ? 0 : (off > size - 1 ? size - 1 : off);
}

struct PointNormal { D3 position, normal; };

// sample_point_on_shape: src/shapes/triangle_mesh.inl:24-50, src/shapes/sphere.inl:161-205
GD PointNormal sample_point_on_light(const DevSceneView &sv, const DevLight &lt, D3 ref_point, D2 uv, double w) {
    PointNormal out;
    if (!lt.is_sphere) {
        const int tri = table_sample(sv.light_tri_cdf + lt.cdf_first, lt.tri_count, w);
        const double *p = sv.light_tri_pos + (size_t)(lt.tri_first + tri) * 9;
        const D3 v0 = mk(p[0], p[1], p[2]), v1 = mk(p[3], p[4], p[5]), v2 = mk(p[6], p[7], p[8]);
        const D3 e1 = v1 - v0, e2 = v2 - v0;
        const double a = sqrt(fmin(fmax(uv.x, 0.0), 1.0));
        const double b1 = 1 - a, b2 = a * uv.y;
        D3 gn = normalize(cross(e1, e2));
        if (lt.has_normals) {
            const double *n = sv.light_tri_nrm + (size_t)(lt.tri_first + tri) * 9;
            const D3 sn = normalize((1 - b1 - b2) * mk(n[0], n[1], n[2]) + b1 * mk(n[3], n[4], n[5]) + b2 * mk(n[6], n[7], n[8]));
            if (dot(gn, sn) < 0) gn = -gn;
        }
        out.position = v0 + (e1 * b1) + (e2 * b2); out.normal = gn;
        return out;
    }
    const DevSphere &sp = sv.spheres[lt.sphere_index];
    const D3 center = mk(sp.center[0], sp.center[1], sp.center[2]);
    const double r = sp.radius;
    const D3 dc_ = ref_point - center;
    if (dot(dc_, dc_) < r * r) {
        const double z = 1 - 2 * uv.x;
        const double r_ = sqrt(fmax(0.0, 1 - z * z));
        const double phi = 2 * kPi * uv.y;
        const D3 offset = mk(r_ * cos(phi), r_ * sin(phi), z);
        out.position = center + r * offset; out.normal = offset;
        return out;
    }
    const D3 dir_to_center = normalize(center - ref_point);
    const Frame frame = make_frame(dir_to_center);
    const double d2 = dot(dc_, dc_);
    const double sin_elevation_max_sq = r * r / d2;
    const double cos_elevation_max = sqrt(fmax(0.0, 1 - sin_elevation_max_sq));
    const double cos_elevation = (1 - uv.x) + uv.x * cos_elevation_max;
    const double sin_elevation = sqrt(fmax(0.0, 1 - cos_elevation * cos_elevation));
    const double azimuth = uv.y * 2 * kPi;
    const double dc = sqrt(d2);
    const double ds = dc * cos_elevation - sqrt(fmax(0.0, r * r - dc * dc * sin_elevation * sin_elevation));
    const double cos_alpha = (dc * dc + r * r - ds * ds) / (2 * dc * r);
    const double sin_alpha = sqrt(fmax(0.0, 1 - cos_alpha * cos_alpha));
    const D3 n_on_sphere = -to_world(frame, mk(sin_alpha * cos(azimuth), sin_alpha * sin(azimuth), cos_alpha));
    out.position = r * n_on_sphere + center; out.normal = n_on_sphere;
    return out;
}

// pdf_point_on_shape: triangle_mesh.inl:56-58, sphere.inl:211-228
GD double pdf_point_on_light(const DevSceneView &sv, const DevLight &lt, const PointNormal &pt, D3 ref_point) {
    if (!lt.is_sphere) return 1 / lt.area;
    const DevSphere &sp = sv.spheres[lt.sphere_index];
    const D3 center = mk(sp.center[0], sp.center[1], sp.center[2]);
    const double r = sp.radius;
    const D3 dc_ = ref_point - center;
    const double d2 = dot(dc_, dc_);
    if (d2 < r * r) return 1 / lt.area;
    const double sin_elevation_max_sq = r * r / d2;
    const double cos_elevation_max = sqrt(fmax(0.0, 1 - sin_elevation_max_sq));
    const double pdf_solid_angle = 1 / (2 * kPi * (1 - cos_elevation_max));
    const D3 dir = normalize(pt.position - ref_point);
    const D3 dl = ref_point - pt.position;
    return pdf_solid_angle * fabs(dot(pt.normal, dir)) / dot(dl, dl);
}

// ---- environment map (src/lights/envmap.inl) + TableDist2D (src/table_dist.cpp:114-150) ----
GD int upper_bound_index(const double *a, int n, double u) {       // first index in [0, n) with a[i] > u, n if none
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] > u) hi = mid; else lo = mid + 1; }
    return lo;
}
GD D2 table2d_sample(const DevSceneView &sv, D2 rnd) {
    const int w = sv.env_w, h = sv.env_h;
    int y_offset = upper_bound_index(sv.env_cdf_marginals, h + 1, rnd.y) - 1;
    y_offset = min(max(y_offset, 0), h - 1);
    double dy = rnd.y - sv.env_cdf_marginals[y_offset];
    const double hy = sv.env_cdf_marginals[y_offset + 1] - sv.env_cdf_marginals[y_offset];
    if (hy > 0) dy /= hy;
    const double *cdf = sv.env_cdf_rows + (size_t)y_offset * (w + 1);
    int x_offset = upper_bound_index(cdf, w + 1, rnd.x) - 1;
    x_offset = min(max(x_offset, 0), w - 1);
    double dx = rnd.x - cdf[x_offset];
    const double hx = cdf[x_offset + 1] - cdf[x_offset];
    if (hx > 0) dx /= hx;
    D2 uv; uv.x = (x_offset + dx) / w; uv.y = (y_offset + dy) / h;
    return uv;
}
GD double table2d_pdf(const DevSceneView &sv, D2 xy) {
    const int w = sv.env_w, h = sv.env_h;
    const int x = (int)fmin(fmax(xy.x * w, 0.0), (double)(w - 1));
    const int y = (int)fmin(fmax(xy.y * h, 0.0), (double)(h - 1));
    return sv.env_pdf_marginals[y] * sv.env_pdf_rows[(size_t)y * w + x] * w * h;
}
GD D3 xform_vector16(const double *m, D3 v) {
    return mk(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
GD D2 envmap_uv(D3 local_dir) {
    const double inv_two_pi = 1.0 / kTwoPi, inv_pi = 1.0 / kPi;
    D2 uv; uv.x = atan2(local_dir.x, -local_dir.z) * inv_two_pi; uv.y = acos(fmin(fmax(local_dir.y, -1.0), 1.0)) * inv_pi;
    if (uv.x < 0) uv.x += 1;
    return uv;
}
// emission(envmap, view_dir, ...): view_dir points away from the light. The footprint the reference derives
// (min(|du/dw|, dv/dw_y) with dv/dw_y < 0) is always negative, i.e. the lookup is always level 0 (envmap.inl:49-64).
GD D3 envmap_emission(const DevSceneView &sv, D3 view_dir) {
    const D3 local_dir = xform_vector16(sv.env_to_local, -view_dir);
    const D2 uv = envmap_uv(local_dir);
    const double lu = modulo_d(uv.x, 1.0), lv = modulo_d(uv.y, 1.0);
    return mip_lookup_level(sv, sv.images[sv.env_image_id], lu, lv, 0) * sv.env_scale;
}
GD D3 envmap_sample_dir(const DevSceneView &sv, D2 rnd_uv) {                 // world_dir; point_on_light.normal = -world_dir
    const D2 uv = table2d_sample(sv, rnd_uv);
    const double azimuth = uv.x * (2 * kPi), elevation = uv.y * kPi;
    const D3 local_dir = mk(sin(azimuth) * sin(elevation), cos(elevation), -cos(azimuth) * sin(elevation));
    return xform_vector16(sv.env_to_world, local_dir);
}
GD double envmap_pdf(const DevSceneView &sv, D3 normal) {
    const D3 local_dir = xform_vector16(sv.env_to_local, -normal);
    const D2 uv = envmap_uv(local_dir);
    const double cos_elevation = local_dir.y;
    const double sin_elevation = sqrt(fmin(fmax(1 - cos_elevation * cos_elevation, 0.0), 1.0));
    if (sin_elevation <= 0) return 0;
    return table2d_pdf(sv, uv) / (2 * kPi * kPi * sin_elevation);
}

// One path_tracing call. Returns the sample's radiance.
GD D3 path_sample(const DevSceneView &sv, const TraceCtx &tx, int max_depth, int x, int y, Pcg &rng, LaneCounters &lc, TraceCounters &tc) {
    const DevCamera &cam = sv.cam;
    const int w = cam.width, h = cam.height;
    const double rx = pcg_real(rng);                                            // :21-22, x first
    const double ry = pcg_real(rng);
    Ray ray = sample_primary(cam, (x + rx) / w, (y + ry) / h);
    const double rd_spread = 0.25 / (double)max(w, h);                          // init_ray_differential, src/ray.h:33-35
    Vertex vertex;
    if (!intersect_ctx<TraceHbm>(sv, tx, ray, rd_spread, vertex, lc, tc))                       // :31-43
        return sv.has_envmap ? envmap_emission(sv, -ray.dir) : splat(0);
    D3 radiance = splat(0), throughput = splat(1.0);
    double eta_scale = 1.0;
    if (vertex.light_id >= 0) radiance = radiance + throughput * emission(sv, vertex, -ray.dir);   // :76-79
    const double shadow_eps = sv.isect_eps;                                     // get_shadow_epsilon, src/scene.h:100-102
    for (int num_vertices = 3; loop_allows(max_depth, num_vertices); num_vertices++) {
        lc.bounces++;
        const GdptMaterial &mat = sv.materials[vertex.material_id];
        // ---- next-event estimation, :116-175
        D2 light_uv; light_uv.x = pcg_real(rng); light_uv.y = pcg_real(rng);
        const double light_w = pcg_real(rng);
        const double shape_w = pcg_real(rng);
        const int light_id = table_sample(sv.light_cdf, sv.num_lights, light_w);
        const DevLight &light = sv.lights[light_id];
        const bool env_light = sv.has_envmap && light_id == sv.env_light_id;
        D3 C1 = splat(0);
        double w1 = 0;
        if (!env_light) {
            const PointNormal pl = sample_point_on_light(sv, light, vertex.position, light_uv, shape_w);
            double G = 0;
            const D3 dir_light = normalize(pl.position - vertex.position);
            const D3 dl = pl.position - vertex.position;
            const double dist2 = dot(dl, dl);
            Ray shadow_ray; shadow_ray.org = vertex.position; shadow_ray.dir = dir_light;
            shadow_ray.tnear = shadow_eps; shadow_ray.tfar = (1 - shadow_eps) * sqrt(dist2);
            if (!occluded_ctx<TraceHbm>(sv, tx, shadow_ray, lc, tc)) G = fmax(-dot(dir_light, pl.normal), 0.0) / dist2;
            const double p1 = sv.light_pmf[light_id] * pdf_point_on_light(sv, light, pl, vertex.position);
            if (G > 0 && p1 > 0) {
                const D3 dir_view = -ray.dir;
                const D3 f = bsdf_eval(sv, mat, dir_view, dir_light, vertex);
                const D3 L = (dot(pl.normal, -dir_light) <= 0) ? splat(0) : mk(light.intensity[0], light.intensity[1], light.intensity[2]);
                C1 = G * f * L;
                double p2 = bsdf_pdf(sv, mat, dir_view, dir_light, vertex);
                p2 *= G;
                w1 = (p1 * p1) / (p1 * p1 + p2 * p2);
                C1 = C1 / p1;
            }
        } else {                                                                // :151-160
            const D3 dir_light = envmap_sample_dir(sv, light_uv);               // = -point_on_light.normal
            double G = 0;
            Ray shadow_ray; shadow_ray.org = vertex.position; shadow_ray.dir = dir_light;
            shadow_ray.tnear = shadow_eps; shadow_ray.tfar = __builtin_huge_val();
            if (!occluded_ctx<TraceHbm>(sv, tx, shadow_ray, lc, tc)) G = 1;
            const double p1 = sv.light_pmf[light_id] * envmap_pdf(sv, -dir_light);
            if (G > 0 && p1 > 0) {
                const D3 dir_view = -ray.dir;
                const D3 f = bsdf_eval(sv, mat, dir_view, dir_light, vertex);
                const D3 L = envmap_emission(sv, -dir_light);
                C1 = G * f * L;
                double p2 = bsdf_pdf(sv, mat, dir_view, dir_light, vertex);
                p2 *= G;
                w1 = (p1 * p1) / (p1 * p1 + p2 * p2);
                C1 = C1 / p1;
            }
        }
        radiance = radiance + throughput * C1 * w1;
        // ---- BSDF sampling, :186-230
        const D3 dir_view = -ray.dir;
        D2 ruv; ruv.x = pcg_real(rng); ruv.y = pcg_real(rng);
        const double rw = pcg_real(rng);
        BsdfSample bs;
        if (!bsdf_sample(sv, mat, dir_view, vertex, ruv, rw, bs)) break;         // :200-203
        const D3 dir_bsdf = bs.dir_out;
        if (bs.eta != 0) eta_scale /= (bs.eta * bs.eta);
        Ray bsdf_ray; bsdf_ray.org = vertex.position; bsdf_ray.dir = dir_bsdf; bsdf_ray.tnear = sv.isect_eps; bsdf_ray.tfar = __builtin_huge_val();
        Vertex bsdf_vertex;
        const bool hit = intersect_ctx<TraceHbm>(sv, tx, bsdf_ray, 0.0, bsdf_vertex, lc, tc);
        double G = 1.0;
        if (hit) { const D3 dl = bsdf_vertex.position - vertex.position; G = fabs(dot(dir_bsdf, bsdf_vertex.gn)) / dot(dl, dl); }
        const D3 f = bsdf_eval(sv, mat, dir_view, dir_bsdf, vertex);
        double p2 = bsdf_pdf(sv, mat, dir_view, dir_bsdf, vertex);
        if (p2 <= 0) break;                                                     // :263-266
        p2 *= G;
        if (hit && bsdf_vertex.light_id >= 0) {                                 // :286-306, no MIS weight (see header)
            const D3 L = emission(sv, bsdf_vertex, -dir_bsdf);
            D3 C2 = G * f * L;
            C2 = C2 / p2;
            radiance = radiance + throughput * C2;
        }
        else if (!hit && sv.has_envmap) {                                       // :307-325, WITH the MIS weight
            const D3 L = envmap_emission(sv, -dir_bsdf);
            D3 C2 = G * f * L;
            const double p1 = sv.light_pmf[sv.env_light_id] * envmap_pdf(sv, -dir_bsdf);
            const double w2 = (p2 * p2) / (p1 * p1 + p2 * p2);
            C2 = C2 / p2;
            radiance = radiance + throughput * C2 * w2;
        }
        if (!hit) break;                                                        // :327-329
        double rr_prob = 1;
        if (num_vertices - 1 >= sv.rr_depth) {                                  // :333-340
            rr_prob = fmin(maxc((1 / eta_scale) * throughput), 0.95);
            if (pcg_real(rng) > rr_prob) break;
        }
        ray = bsdf_ray;
        vertex = bsdf_vertex;
        throughput = throughput * (G * f) / (p2 * rr_prob);                      // :344
    }
    return radiance;
}

GD void path_count_nonfinite(D3 r, LaneCounters &lc) { if (!isfinite(r.x + r.y + r.z)) lc.nonfinite++; }

#ifdef GDPT_BUILD_PATH_MISC   // non-template kernels: emitted by render_path.hip only
// SAMPLE streams: K = 2^log2k lanes per pixel, each sums a contiguous chunk of the pixel's samples; the K partial sums
// are combined in a fixed-order tree and divided by spp (src/render.cpp:107-110).
__global__ __launch_bounds__(kBlock, 2) void gdpt_path_eager(DevSceneView sv, KernelArgs a) {
    __shared__ int s_stack[GDPT_BVH_MAX_DEPTH * kBlock];
    const int tid = threadIdx.x;
    TraceCtx tx = setup_trace<false, true>(sv, nullptr, s_stack, tid, kBlock, a.count != 0);
    const int K = 1 << a.log2k;
    const int c = tid & (K - 1), p = tid >> a.log2k;
    const int px = p % a.tile_w, py = p / a.tile_w;
    const int bx = blockIdx.x % a.tiles_x, by = blockIdx.x / a.tiles_x;
    const int x = bx * a.tile_w + px, y = a.row_begin + by * a.tile_h + py;
    const int W = sv.cam.width;
    const bool valid = (x < W) && (y < a.row_end);
    LaneCounters lc = {0, 0, 0};
    TraceCounters tc = {0, 0, 0, 0, 0, 0};
    D3 sum = splat(0);
    if (valid) {
        const int s0 = (int)(((long long)c * a.spp) >> a.log2k), s1 = (int)(((long long)(c + 1) * a.spp) >> a.log2k);
        const unsigned long long base = ((unsigned long long)y * W + x) * (unsigned long long)a.spp;
        for (int s = s0; s < s1; s++) {
            Pcg rng = pcg_init(base + (unsigned long long)s);
            const D3 r = path_sample(sv, tx, a.max_depth, x, y, rng, lc, tc);
            path_count_nonfinite(r, lc);
            sum = sum + r;
        }
    }
    for (int o = K >> 1; o >= 1; o >>= 1) { sum.x += __shfl_xor(sum.x, o, 64); sum.y += __shfl_xor(sum.y, o, 64); sum.z += __shfl_xor(sum.z, o, 64); }
    if (valid && c == 0) {
        const D3 px_val = sum / (double)a.spp;
        const size_t i = ((size_t)y * W + x) * 3;
        a.img[i] = px_val.x; a.img[i + 1] = px_val.y; a.img[i + 2] = px_val.z;
    }
    flush_counters(a, lc, tc, a.count != 0);
}

// TILE streams: the reference's RNG order, one PCG stream per 16x16 tile (src/render.cpp:94-112), one lane per tile.
__global__ __launch_bounds__(64) void gdpt_path_tile_stream(DevSceneView sv, KernelArgs a, int ntx, int nty) {
    __shared__ int s_stack[GDPT_BVH_MAX_DEPTH * 64];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x * 64 + tid;
    TraceCtx tx = setup_trace<false, true>(sv, nullptr, s_stack, tid, 64, a.count != 0);
    LaneCounters lc = {0, 0, 0};
    TraceCounters tc = {0, 0, 0, 0, 0, 0};
    const int W = sv.cam.width, H = sv.cam.height;
    if (tile < ntx * nty) {
        const int txi = tile % ntx, tyi = tile / ntx;
        Pcg rng = pcg_init((unsigned long long)(tyi * ntx + txi));
        const int x0 = txi * 16, x1 = min(x0 + 16, W), y0 = tyi * 16, y1 = min(y0 + 16, H);
        for (int y = y0; y < y1; y++) {
            if (y < a.row_begin || y >= a.row_end) continue;
            for (int x = x0; x < x1; x++) {
                D3 sum = splat(0);
                for (int s = 0; s < a.spp; s++) {
                    const D3 r = path_sample(sv, tx, a.max_depth, x, y, rng, lc, tc);
                    path_count_nonfinite(r, lc);
                    sum = sum + r;
                }
                const D3 px_val = sum / (double)a.spp;
                const size_t i = ((size_t)y * W + x) * 3;
                a.img[i] = px_val.x; a.img[i + 1] = px_val.y; a.img[i + 2] = px_val.z;
            }
        }
    }
    flush_counters(a, lc, tc, a.count != 0);
}

#endif // GDPT_BUILD_PATH_MISC

// ------------------------------------------------------------------------------------------------
// persistent lane machine for Integrator::Path (same structure as gdpt_render_phases, render_device.h §"lane machine")
// ------------------------------------------------------------------------------------------------
// A lane holds one pending ray: the camera ray of its current sample, a shadow ray towards the light sample of the
// vertex it stands on, or the BSDF-sampled ray leaving that vertex. All shading of a vertex — light sample, BSDF
// evaluation towards it, MIS weight, BSDF sample with f and pdf — happens when the vertex is reached, so nothing but
// (origin, two directions, f, pdf, throughput) stays in registers across the traversals; radiance and the pending
// next-event contribution live in the lane's LDS slot.
enum { P_START = 0, P_PRIMARY = 1, P_SHADOW = 2, P_BOUNCE = 3, P_DONE = 4 };

struct PathLane {
    int st, s, s_end, num_vertices;
    int bounce_valid;                 // a BSDF-sampled ray follows the shadow ray
    unsigned long long rng_state, rng_inc;
    D3 org;                           // the vertex both pending rays leave (camera position for P_PRIMARY)
    D3 dir_b;                         // closest-hit ray: camera ray or BSDF-sampled direction
    D3 dir_s; double tfar_s;          // shadow ray
    double eta_scale;
};
// LDS slot layout (doubles, stride kBlock): 0..2 radiance of the current sample, 3..5 pending next-event contribution,
// 6..8 f*|cos| and 9 the solid-angle pdf of dir_b at org, 10..12 throughput — everything that is cold while a ray is
// in flight ("manual spills" that cost an LDS access instead of a scratch round trip).
constexpr int kPathPrivDoubles = 13;
struct PathPriv {
    double *slot; int stride;
    GD D3 radiance() const { return mk(slot[0], slot[stride], slot[2 * stride]); }
    GD void set_radiance(D3 v) { slot[0] = v.x; slot[stride] = v.y; slot[2 * stride] = v.z; }
    GD D3 nee() const { return mk(slot[3 * stride], slot[4 * stride], slot[5 * stride]); }
    GD void set_nee(D3 v) { slot[3 * stride] = v.x; slot[4 * stride] = v.y; slot[5 * stride] = v.z; }
    GD D3 f() const { return mk(slot[6 * stride], slot[7 * stride], slot[8 * stride]); }
    GD double pdf() const { return slot[9 * stride]; }
    GD void set_f_pdf(D3 v, double p) { slot[6 * stride] = v.x; slot[7 * stride] = v.y; slot[8 * stride] = v.z; slot[9 * stride] = p; }
    GD D3 throughput() const { return mk(slot[10 * stride], slot[11 * stride], slot[12 * stride]); }
    GD void set_throughput(D3 v) { slot[10 * stride] = v.x; slot[11 * stride] = v.y; slot[12 * stride] = v.z; }
};

GD bool path_lane_tracing(int st) { return st == P_PRIMARY || st == P_SHADOW || st == P_BOUNCE; }

template <class TC>
GD void path_trace_pending(const DevSceneView &sv, const TraceCtx &tx, const PathLane &L, Trav &tv, int keep_frac, int search_frac, TraceCounters &tc) {
    const bool pending = path_lane_tracing(L.st) && tv.cur != kTravDone;
    const unsigned long long m = __ballot(pending);
    if (m == 0ull) return;
    const int stop_below = (__popcll(m) * keep_frac) >> 8;
    if (pending) {
        const bool shadow = (L.st == P_SHADOW);
        const float tnear = (L.st == P_PRIMARY) ? 0.0f : (float)sv.isect_eps;        // shadow epsilon == intersection epsilon
        const float tfar = shadow ? (float)L.tfar_s : __builtin_huge_valf();
        trav_run<TC>(sv, tx, L.org, shadow ? L.dir_s : L.dir_b, tnear, tfar, tv, stop_below, search_frac, tc, shadow);
    }
}

// One step of a lane whose pending ray is finished (or that needs its first ray). ENV: the scene has an environment map
// (compile-time so that scenes without one do not carry its code and registers).
// PLAIN: device_trace.h (scene without spheres / with constant textures only: that code is not compiled in).
template <bool LAMBERT, bool ENV, int PLAIN = 0>
GD void path_lane_step(const DevSceneView &sv, const TraceCtx &tx, int max_depth, int x, int y, unsigned long long base,
                       PathLane &L, Trav &tv, PathPriv &lp, double *acc_slot, int acc_stride, LaneCounters &lc) {
    const DevCamera &cam = sv.cam;
    const int w = cam.width, h = cam.height;
    const int st0 = L.st;
    bool shade = false, finish = false, new_sample = (st0 == P_START);
    Vertex nv;
    D3 arriving = L.dir_b;                       // direction of the closest-hit ray that reached nv
    if (st0 == P_PRIMARY || st0 == P_BOUNCE) {
        lc.rays++;
        const bool hit = tv.best.gid >= 0;
        Ray ray; ray.org = L.org; ray.dir = L.dir_b; ray.tnear = 0; ray.tfar = __builtin_huge_val();
        if (hit) make_vertex<PLAIN>(sv, tx.tris, tx.need_uv, ray, tv.best, 0.0, (st0 == P_PRIMARY) ? 0.25 / (double)max(w, h) : 0.0, nv);
        if (st0 == P_PRIMARY) {
            if (!hit) { lp.set_radiance(ENV ? envmap_emission(sv, -L.dir_b) : splat(0)); finish = true; }   // :31-43
            else {
                lp.set_throughput(splat(1.0)); L.eta_scale = 1.0; L.num_vertices = 3;
                lp.set_radiance((nv.light_id >= 0) ? emission(tx.lights, nv, -L.dir_b) : splat(0));       // :76-79
                if (loop_allows(max_depth, 3)) shade = true; else finish = true;
            }
        } else {
            double G = 1.0;
            if (hit) { const D3 dl = nv.position - L.org; G = fabs(dot(L.dir_b, nv.gn)) / dot(dl, dl); }
            const D3 f_b = lp.f();
            const D3 T = lp.throughput();
            const double p2 = lp.pdf() * G;                                                        // :268 (pdf > 0 was checked at the vertex)
            if (hit && nv.light_id >= 0) {                                                         // :286-306, no MIS weight
                const D3 Le = emission(tx.lights, nv, -L.dir_b);
                D3 C2 = G * f_b * Le;
                C2 = C2 / p2;
                lp.set_radiance(lp.radiance() + T * C2);
            }
            else if (ENV && !hit) {                                                                // :307-325, WITH the MIS weight
                const D3 Le = envmap_emission(sv, -L.dir_b);
                D3 C2 = G * f_b * Le;
                const double p1 = sv.light_pmf[sv.env_light_id] * envmap_pdf(sv, -L.dir_b);
                const double w2 = (p2 * p2) / (p1 * p1 + p2 * p2);
                C2 = C2 / p2;
                lp.set_radiance(lp.radiance() + T * C2 * w2);
            }
            if (!hit) finish = true;                                                               // :327-329
            else {
                double rr_prob = 1;
                bool stop = false;
                if (L.num_vertices - 1 >= sv.rr_depth) {                                           // :333-340
                    rr_prob = fmin(maxc((1 / L.eta_scale) * T), 0.95);
                    Pcg r; r.state = L.rng_state; r.inc = L.rng_inc;
                    const double u = pcg_real(r); L.rng_state = r.state;
                    if (u > rr_prob) stop = true;
                }
                if (stop) finish = true;
                else {
                    lp.set_throughput(T * (G * f_b) / (p2 * rr_prob));                             // :344
                    L.num_vertices++;
                    if (loop_allows(max_depth, L.num_vertices)) shade = true; else finish = true;
                }
            }
        }
    } else if (st0 == P_SHADOW) {
        lc.rays++;
        if (!(tv.best.gid >= 0)) lp.set_radiance(lp.radiance() + lp.nee());                       // unoccluded: :178
        if (L.bounce_valid) { L.st = P_BOUNCE; trav_init(sv, tv, __builtin_huge_val()); }
        else finish = true;
    }
    if (shade) {                                  // all the work of one loop iteration at vertex nv (:113-266)
        lc.bounces++;
        Pcg rng; rng.state = L.rng_state; rng.inc = L.rng_inc;
        const D3 dir_view = -arriving;
        // ---- next-event estimation
        D2 light_uv; light_uv.x = pcg_real(rng); light_uv.y = pcg_real(rng);
        const double light_w = pcg_real(rng);
        const double shape_w = pcg_real(rng);
        const int light_id = table_sample(sv.light_cdf, sv.num_lights, light_w);
        const DevLight &light = sv.lights[light_id];
        const bool env_light = ENV && light_id == sv.env_light_id;
        D3 dir_light;
        double shadow_tfar;
        D3 nee = splat(0);
        if (!env_light) {
            const PointNormal pl = sample_point_on_light(sv, light, nv.position, light_uv, shape_w);
            const D3 dl = pl.position - nv.position;
            const double dist2 = dot(dl, dl);
            dir_light = normalize(dl);
            shadow_tfar = (1 - sv.isect_eps) * sqrt(dist2);
            // the contribution assuming the shadow ray is unoccluded; the ray is traced only if it is non-zero
            const double G = fmax(-dot(dir_light, pl.normal), 0.0) / dist2;
            const double p1 = sv.light_pmf[light_id] * pdf_point_on_light(sv, light, pl, nv.position);
            if (G > 0 && p1 > 0) {
                D3 f; double p2;
                mat_eval_pdf<LAMBERT, true, true, kAllMaterials, PLAIN>(sv, tx, nv, dir_view, dir_light, f, p2);
                const D3 Le = (dot(pl.normal, -dir_light) <= 0) ? splat(0) : mk(light.intensity[0], light.intensity[1], light.intensity[2]);
                D3 C1 = G * f * Le;
                p2 *= G;
                const double w1 = (p1 * p1) / (p1 * p1 + p2 * p2);
                C1 = C1 / p1;
                nee = lp.throughput() * C1 * w1;
            }
        } else {                                                                                   // :151-160: G = 1 if unoccluded
            dir_light = envmap_sample_dir(sv, light_uv);
            shadow_tfar = __builtin_huge_val();
            const double p1 = sv.light_pmf[light_id] * envmap_pdf(sv, -dir_light);
            if (p1 > 0) {
                D3 f; double p2;
                mat_eval_pdf<LAMBERT, true, true, kAllMaterials, PLAIN>(sv, tx, nv, dir_view, dir_light, f, p2);
                D3 C1 = 1.0 * f * envmap_emission(sv, -dir_light);
                p2 *= 1.0;
                const double w1 = (p1 * p1) / (p1 * p1 + p2 * p2);
                C1 = C1 / p1;
                nee = lp.throughput() * C1 * w1;
            }
        }
        // ---- BSDF sampling
        D2 ruv; ruv.x = pcg_real(rng); ruv.y = pcg_real(rng);
        const double rw = pcg_real(rng);
        L.rng_state = rng.state;
        BsdfSample bs;
        L.bounce_valid = 0;
        if (mat_sample<LAMBERT, true, true>(sv, tx, nv, dir_view, ruv, rw, bs)) {                              // :200-203
            if (bs.eta != 0) L.eta_scale /= (bs.eta * bs.eta);
            D3 f; double pdf;
            mat_eval_pdf<LAMBERT, true, true, kAllMaterials, PLAIN>(sv, tx, nv, dir_view, bs.dir_out, f, pdf);
            if (pdf > 0) { L.bounce_valid = 1; L.dir_b = bs.dir_out; lp.set_f_pdf(f, pdf); }      // :263-266
        }
        L.org = nv.position;
        const bool want_shadow = (nee.x != 0 || nee.y != 0 || nee.z != 0);
        if (want_shadow) {
            lp.set_nee(nee);
            L.dir_s = dir_light; L.tfar_s = shadow_tfar;
            L.st = P_SHADOW; trav_init(sv, tv, L.tfar_s);
        } else if (L.bounce_valid) { L.st = P_BOUNCE; trav_init(sv, tv, __builtin_huge_val()); }
        else finish = true;
    }
    if (finish) {                                 // src/render.cpp:107-109
        const D3 r = lp.radiance();
        path_count_nonfinite(r, lc);
        __hip_atomic_fetch_add(acc_slot, r.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(acc_slot + acc_stride, r.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(acc_slot + 2 * acc_stride, r.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        L.s++;
        if (L.s >= L.s_end) L.st = P_DONE; else new_sample = true;
    }
    if (new_sample) {
        Pcg r = pcg_init(base + (unsigned long long)L.s);
        const double rx = pcg_real(r);                                                             // :21-22, x first
        const double ry = pcg_real(r);
        L.rng_state = r.state; L.rng_inc = r.inc;
        const Ray pr = sample_primary(cam, (x + rx) / w, (y + ry) / h);
        L.org = pr.org; L.dir_b = pr.dir;
        L.st = P_PRIMARY; trav_init(sv, tv, __builtin_huge_val());
    }
}

template <bool LAMBERT, bool LDS_SCENE, bool ENV, int PLAIN = 0>
__global__ __launch_bounds__(kBlock, 2) void gdpt_path_persistent(DevSceneView sv, KernelArgs a) {
    constexpr int kLevels = LDS_SCENE ? kLdsSceneLevels : GDPT_BVH_MAX_DEPTH;
    __shared__ int s_stack[kLevels * kBlock];
    __shared__ __attribute__((aligned(16))) unsigned char s_scene[LDS_SCENE ? kLdsSceneBytes : 16];
    __shared__ double s_acc[3 * kBlock];
    __shared__ double s_priv[kPathPrivDoubles * kBlock];
    const int tid = threadIdx.x;
    TraceCtx tx = setup_trace<LDS_SCENE, true>(sv, s_scene, s_stack, tid, kBlock, a.count != 0);
    const int W = sv.cam.width;
    double *acc_slot = s_acc + tid;
    acc_slot[0] = 0; acc_slot[kBlock] = 0; acc_slot[2 * kBlock] = 0;
    PathPriv lp; lp.slot = s_priv + tid; lp.stride = kBlock;
    LaneCounters lc = {0, 0, 0};
    TraceCounters tc = {0, 0, 0, 0, 0, 0};
    PathLane L;
    Trav tv;
    trav_init(sv, tv, __builtin_huge_val());
    L.s = 0; L.s_end = 0; L.st = P_DONE; L.num_vertices = 0; L.bounce_valid = 0; L.rng_state = 0; L.rng_inc = 1;
    L.org = L.dir_b = L.dir_s = splat(0); L.tfar_s = 0; L.eta_scale = 1;
    int x = 0, y = 0;
    unsigned long long base = 0;
    long long my_item = -1;
    WaveQueue wq;
    for (;;) {
        const bool idle = (L.st == P_DONE);
        if (idle && my_item >= 0) {                     // item finished: publish its three sums (32-byte record)
            typedef double d2 __attribute__((ext_vector_type(2)));
            d2 *dst = (d2 *)(a.partials + (size_t)my_item * 4);
            dst[0] = d2{acc_slot[0], acc_slot[kBlock]}; dst[1] = d2{acc_slot[2 * kBlock], 0.0};
            acc_slot[0] = 0; acc_slot[kBlock] = 0; acc_slot[2 * kBlock] = 0;
            my_item = -1;
        }
        const long long got_item = wq.take(a, idle, tid);
        if (got_item >= 0) {
            my_item = got_item;
            int s0, s1;
            const bool inside = item_to_pixel(a, W, (unsigned)my_item, x, y, s0, s1, a.chunk_begin);
            base = ((unsigned long long)y * W + x) * (unsigned long long)a.spp;
            L.s = s0; L.s_end = s1;
            L.st = (inside && s0 < s1) ? P_START : P_DONE;
        }
        if (!__any(L.st != P_DONE)) { if (wq.exhausted) break; else continue; }
        path_trace_pending<TraceCfg<true, true, !LDS_SCENE, !(PLAIN & kPlainNoSpheres)>>(sv, tx, L, tv, a.thresh_a, a.thresh_c, tc);
        if (L.st == P_START || (path_lane_tracing(L.st) && tv.cur == kTravDone)) {
            if (tx.count) { tc.lane_steps++; if (wave_leader()) tc.wave_steps++; }
            path_lane_step<LAMBERT, ENV, PLAIN>(sv, tx, a.max_depth, x, y, base, L, tv, lp, acc_slot, kBlock, lc);
        }
    }
    flush_counters(a, lc, tc, a.count != 0);
}

#ifdef GDPT_BUILD_PATH_MISC
// Sums the per-chunk records of every pixel in chunk order and divides by spp (src/render.cpp:110).
__global__ __launch_bounds__(256) void gdpt_path_reduce(KernelArgs a, int W) {
    const long long nslots = a.num_slots;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long idx = t >> 2;
    const int j = (int)(t & 3);
    if (idx >= nslots || j == 3) return;
    int x, y, s0, s1;
    if (!item_to_pixel(a, W, (unsigned)idx, x, y, s0, s1, a.chunk_begin)) return;
    const double *src = a.partials + (size_t)idx * 4 + j;
    double v = 0;
    for (int c = 0; c < a.num_chunks; c++) v += src[(size_t)c * (size_t)nslots * 4];
    a.img[((size_t)y * W + x) * 3 + j] = v / (double)a.spp;
}

#endif // GDPT_BUILD_PATH_MISC

} // namespace gd

namespace gdpt {
// one translation unit per kernel family (parallel compilation)
void launch_path_persistent_lambert(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, bool lds, bool plain, hipStream_t stream);
void launch_path_persistent_general(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, bool lds, hipStream_t stream);
} // namespace gdpt
