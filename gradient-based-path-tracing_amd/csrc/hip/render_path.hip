// render_path.hip — Integrator::Path on gfx950: path_tracing (src/path_tracing.h:13-348) and the
// path_render tile loop (src/render.cpp:74-117) for scenes lit by area emitters (triangle meshes, spheres).
//
// Unidirectional path tracing with next-event estimation and power-heuristic MIS. Reproduced as the reference computes
// it, including: the emitter-hit term of the BSDF-sampled ray is added WITHOUT its MIS weight w2 (:303-306 computes w2
// and drops it — only the environment-map branch applies it); Russian roulette from rr_depth on throughput/eta_scale;
// the sub-pixel numbers are drawn x first (left-to-right evaluation of the constructor arguments at :21-22, the order
// of the compiler the reference was developed with). Environment maps are not restated: gdpt_path_render refuses such
// scenes. Shares traversal, vertex reconstruction, BSDFs and textures with the GradPath kernels (render_device.h);
// shadow rays are any-hit walks of the same BVH4.
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

// One path_tracing call. Returns the sample's radiance.
GD D3 path_sample(const DevSceneView &sv, const TraceCtx &tx, int max_depth, int x, int y, Pcg &rng, LaneCounters &lc, TraceCounters &tc) {
    const DevCamera &cam = sv.cam;
    const int w = cam.width, h = cam.height;
    const double rx = pcg_real(rng);                                            // :21-22, x first
    const double ry = pcg_real(rng);
    Ray ray = sample_primary(cam, (x + rx) / w, (y + ry) / h);
    const double rd_spread = 0.25 / (double)max(w, h);                          // init_ray_differential, src/ray.h:33-35
    Vertex vertex;
    if (!intersect_ctx<TraceHbm>(sv, tx, ray, rd_spread, vertex, lc, tc)) return splat(0);   // :31-43 (no environment map)
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
        const PointNormal pl = sample_point_on_light(sv, light, vertex.position, light_uv, shape_w);
        D3 C1 = splat(0);
        double w1 = 0;
        {
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

// SAMPLE streams: K = 2^log2k lanes per pixel, each sums a contiguous chunk of the pixel's samples; the K partial sums
// are combined in a fixed-order tree and divided by spp (src/render.cpp:107-110).
__global__ __launch_bounds__(kBlock) void gdpt_path_eager(DevSceneView sv, KernelArgs a) {
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

} // namespace gd

namespace gdpt {
void launch_path(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, hipStream_t stream) {
    hipLaunchKernelGGL(gd::gdpt_path_eager, grid, dim3(gd::kBlock), 0, stream, sv, a);
}
void launch_tile_path(const DevSceneView &sv, const gd::KernelArgs &a, dim3 grid, int ntx, int nty, hipStream_t stream) {
    hipLaunchKernelGGL(gd::gdpt_path_tile_stream, grid, dim3(64), 0, stream, sv, a, ntx, nty);
}
} // namespace gdpt
