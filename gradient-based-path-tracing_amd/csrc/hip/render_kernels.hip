// render_kernels.hip — the GDPT sample evaluator and tile loop for gfx950.
//
// Restates gradient_path_render's tile loop (src/render.cpp:277-331) and grad_path_tracing
// (src/path_tracing.h:354-1050): one base path plus four pixel-offset paths per sample, BSDF sampling
// only (no NEE/MIS on this integrator), Russian roulette from rr_depth, per-pixel accumulation of
// img, cx0, cy0, cx1, cy1. The four undefined end-of-bounce reads (src/path_tracing.h:1007-1010) use
// the "A-semantics" of SURVEY.md §8(a) G2: an offset keeps its primary hit for the whole path.
//
// Mapping to the hardware (v1): one lane owns (pixel, chunk of the pixel's samples); K lanes per pixel
// sit next to each other in a wave and are combined with a fixed-order xor-shuffle tree, so every
// pixel has exactly one writer (no atomics, run-to-run deterministic). Traversal is fp32 with a
// per-lane stack in LDS; shading is fp64 with per-lane PCG32 state in registers.
#include "render_kernels.h"
#include "device_trace.h"

#include <stdexcept>
#include <string>

namespace gd {

constexpr int kBlock = 256;

struct Offset {           // per-offset path state (lives in private memory; touched once per bounce)
    Vertex v;             // the offset's PRIMARY hit (A-semantics)
    D3 dir;               // ray_*.dir
    D3 contrib;
    double jacob;
};

struct SampleOut {        // what the tile loop needs from one GraidentPTRadiance (src/intersection.h:65-77)
    D3 radiance, contrib;
    D3 cX[4];             // contribX0, contribX1, contribY0, contribY1
    double w[4];          // wX0, wX1, wY0, wY1
    double prob;
};

struct LaneCounters { unsigned rays, bounces, nonfinite; };

GD void zero_out(SampleOut &o) {
    o.radiance = splat(0); o.contrib = splat(0);
#pragma unroll
    for (int k = 0; k < 4; k++) { o.cX[k] = splat(0); o.w[k] = 1.0; }
    o.prob = 1.0;
}

template <bool COUNT>
GD void grad_sample(const DevSceneView &sv, int max_depth, int x, int y, Pcg &rng, int *stack, int stride,
                    SampleOut &out, LaneCounters &lc, TraceCounters &tc) {
    const DevCamera &cam = sv.cam;
    const int w = cam.width, h = cam.height;
    zero_out(out);
    double rng_x = pcg_real(rng), rng_y = pcg_real(rng);                          // :360-361
    Ray ray = sample_primary(cam, (x + rng_x) / w, (y + rng_y) / h);
    const double rd_spread = 0.25 / (double)max(w, h);                            // init_ray_differential, src/ray.h:33-35
    Vertex vertex;
    lc.rays++;
    if (!intersect<COUNT>(sv, ray, 0.0, rd_spread, vertex, stack, stride, tc)) return;   // :375-379

    // offsets x0,x1,y0,y1 = (x-1,y),(x+1,y),(x,y+1),(x,y-1) with the same sub-pixel numbers (:385-403)
    Offset off[4];
    unsigned alive = 0;
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        int ox = (k == 0) ? -1 : (k == 1 ? 1 : 0), oy = (k == 2) ? 1 : (k == 3 ? -1 : 0);
        Ray r = sample_primary(cam, ((x + ox) + rng_x) / w, ((y + oy) + rng_y) / h);
        lc.rays++;
        Vertex ov;
        bool ok = intersect<COUNT>(sv, r, 0.0, rd_spread, ov, stack, stride, tc);
        if (ok && ov.material_id == vertex.material_id) {                         // :424-443
            alive |= 1u << k;
            off[k].v = ov; off[k].dir = r.dir; off[k].jacob = 1.0;
            off[k].contrib = (ov.light_id >= 0) ? emission(sv, ov, -r.dir) : splat(1.0);   // :496-508
        }
    }

    D3 contrib = splat(1.0), throughput = splat(1.0), radiance = splat(0);
    double prob = 1.0, eta_scale = 1.0;
    if (vertex.light_id >= 0) {                                                   // :490-493
        D3 L = emission(sv, vertex, -ray.dir);
        radiance = radiance + throughput * L;
        contrib = L;
    }

    for (int num_vertices = 3; max_depth == -1 || num_vertices <= max_depth + 1; num_vertices++) {
        lc.bounces++;
        const GdptMaterial &mat = sv.materials[vertex.material_id];
        D3 dir_view = -ray.dir;
        D2 ruv; ruv.x = pcg_real(rng); ruv.y = pcg_real(rng);                     // :536 (brace-init: ordered)
        double rw = pcg_real(rng);
        BsdfSample bs;
        if (!bsdf_sample(sv, mat, dir_view, vertex, ruv, rw, bs)) { zero_out(out); return; }   // :545-548
        D3 dir_bsdf = bs.dir_out;
        if (bs.eta != 0) eta_scale /= (bs.eta * bs.eta);                          // :553-558
        Ray bsdf_ray; bsdf_ray.org = vertex.position; bsdf_ray.dir = dir_bsdf; bsdf_ray.tnear = sv.isect_eps; bsdf_ray.tfar = __builtin_huge_val();
        Vertex bsdf_vertex;
        lc.rays++;
        bool hit = intersect<COUNT>(sv, bsdf_ray, 0.0, 0.0, bsdf_vertex, stack, stride, tc);   // :564 (default RayDifferential)
        // :565-568: four rays with tnear = tfar = 0 can never hit; their results are unobservable -> not traced.

        // :571-740: CHECK_TYPE(...) is always false, so only the material test remains
        if (alive) {
#pragma unroll 1
            for (int k = 0; k < 4; k++)
                if ((alive >> k & 1u) && off[k].v.material_id != vertex.material_id) alive &= ~(1u << k);
        }

        double G = 1.0;
        if (hit) {
            D3 dl = bsdf_vertex.position - vertex.position;
            G = fabs(dot(dir_bsdf, bsdf_vertex.gn)) / dot(dl, dl);
        }
        D3 f = bsdf_eval(sv, mat, dir_view, dir_bsdf, vertex);
        double p2 = bsdf_pdf(sv, mat, dir_view, dir_bsdf, vertex);
        if (p2 <= 0) break;                                                        // :760-763
        p2 *= G;
        contrib = contrib * f * G;                                                 // :769
        prob *= p2;

        if (alive) {                                                               // :773-959 (merge_flag never set)
#pragma unroll 1
            for (int k = 0; k < 4; k++) {
                if (!(alive >> k & 1u)) continue;
                Offset &o = off[k];
                const GdptMaterial &omat = sv.materials[o.v.material_id];
                D3 oin = -o.dir;
                BsdfSample os;
                if (!bsdf_sample(sv, omat, oin, o.v, ruv, rw, os)) { alive &= ~(1u << k); continue; }
                double p2o = bsdf_pdf(sv, omat, oin, os.dir_out, o.v);
                if (p2o <= 0.0) { alive &= ~(1u << k); continue; }
                o.jacob *= p2 / p2o;                                               // p2 holds G, p2o does not (:813)
                o.dir = os.dir_out;                                                // :815-816
            }
        }

        if (hit && bsdf_vertex.light_id >= 0) {                                    // :971-980
            D3 L = emission(sv, bsdf_vertex, -dir_bsdf);
            D3 C2 = (G * f) * L;
            contrib = contrib * L;
            C2 = C2 / p2;
            radiance = radiance + throughput * C2;
        }
        if (!hit) break;                                                           // :982-985
        double rr_prob = 1;
        if (num_vertices - 1 >= sv.rr_depth) {                                     // :992-999
            rr_prob = fmin(maxc((1 / eta_scale) * throughput), 0.95);
            if (pcg_real(rng) > rr_prob) break;
        }
        ray = bsdf_ray;
        vertex = bsdf_vertex;
        throughput = throughput * (G * f) / (p2 * rr_prob);                        // :1003
        // :1007-1010 -> A-semantics: offsets keep their primary vertex
    }

    out.radiance = radiance; out.contrib = contrib; out.prob = prob;
    if (alive) {                                                                   // :1016-1045 (prob_x* stays 1)
#pragma unroll 1
        for (int k = 0; k < 4; k++)
            if (alive >> k & 1u) {
                out.cX[k] = off[k].contrib * off[k].jacob;
                out.w[k] = prob / (prob + 1.0 * off[k].jacob);
            }
    }
}

struct Accum { D3 r, dx0, dy0, dx1, dy1; };

GD void accumulate(Accum &a, const SampleOut &s, double spp, LaneCounters &lc) {     // src/render.cpp:311-318
    bool finite = isfinite(s.prob) && isfinite(s.radiance.x + s.radiance.y + s.radiance.z) && isfinite(s.contrib.x + s.contrib.y + s.contrib.z);
#pragma unroll
    for (int k = 0; k < 4; k++) finite = finite && isfinite(s.cX[k].x + s.cX[k].y + s.cX[k].z) && isfinite(s.w[k]);
    if (!finite) lc.nonfinite++;
    if (s.prob > 0.0) {
        a.r = a.r + s.radiance / spp;
        double ps = s.prob * spp;
        a.dx0 = a.dx0 + (s.contrib - s.cX[0]) * (s.w[0] / ps);
        a.dy0 = a.dy0 + (s.contrib - s.cX[2]) * (s.w[2] / ps);
        a.dx1 = a.dx1 + (s.cX[1] - s.contrib) * (s.w[1] / ps);
        a.dy1 = a.dy1 + (s.cX[3] - s.contrib) * (s.w[3] / ps);
    }
}

GD unsigned wave_sum_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
GD unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct KernelArgs {
    int spp, log2k, tile_w, tile_h, tiles_x;
    int row_begin, row_end, max_depth;
    double *img, *cx0, *cy0, *cx1, *cy1;
    gdpt::RenderCounters *counters;
};

GD void flush_counters(const KernelArgs &a, const LaneCounters &lc, const TraceCounters &tc, bool count) {
    unsigned r = wave_sum_u32(lc.rays), b = wave_sum_u32(lc.bounces), nf = wave_sum_u32(lc.nonfinite);
    unsigned long long nn = 0, np = 0;
    if (count) { nn = wave_sum_u64(tc.nodes); np = wave_sum_u64(tc.prims); }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&a.counters->rays, (unsigned long long)r);
        atomicAdd(&a.counters->bounces, (unsigned long long)b);
        if (nf) atomicAdd(&a.counters->nonfinite, (unsigned long long)nf);
        if (count) { atomicAdd(&a.counters->nodes, nn); atomicAdd(&a.counters->prims, np); }
    }
}

// SAMPLE stream: init_pcg32((y*W+x)*spp + s) per sample. K = 2^log2k lanes share a pixel.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void gdpt_render_sample_stream(DevSceneView sv, KernelArgs a) {
    __shared__ int s_stack[GDPT_BVH_MAX_DEPTH * kBlock];
    const int tid = threadIdx.x;
    const int K = 1 << a.log2k;
    const int c = tid & (K - 1), p = tid >> a.log2k;
    const int px = p % a.tile_w, py = p / a.tile_w;
    const int bx = blockIdx.x % a.tiles_x, by = blockIdx.x / a.tiles_x;
    const int x = bx * a.tile_w + px, y = a.row_begin + by * a.tile_h + py;
    const int W = sv.cam.width;
    const bool valid = (x < W) && (y < a.row_end);
    Accum acc; acc.r = acc.dx0 = acc.dy0 = acc.dx1 = acc.dy1 = splat(0);
    LaneCounters lc = {0, 0, 0};
    TraceCounters tc = {0, 0};
    if (valid) {
        const int s0 = (int)(((long long)c * a.spp) >> a.log2k), s1 = (int)(((long long)(c + 1) * a.spp) >> a.log2k);
        const unsigned long long base = ((unsigned long long)y * W + x) * (unsigned long long)a.spp;
        for (int s = s0; s < s1; s++) {
            Pcg rng = pcg_init(base + (unsigned long long)s);
            SampleOut so;
            grad_sample<COUNT>(sv, a.max_depth, x, y, rng, s_stack + tid, kBlock, so, lc, tc);
            accumulate(acc, so, (double)a.spp, lc);
        }
    }
    // fixed-order tree over the K lanes of a pixel (lanes of one pixel are contiguous and K-aligned)
    for (int o = K >> 1; o >= 1; o >>= 1) {
        acc.r.x += __shfl_xor(acc.r.x, o, 64); acc.r.y += __shfl_xor(acc.r.y, o, 64); acc.r.z += __shfl_xor(acc.r.z, o, 64);
        acc.dx0.x += __shfl_xor(acc.dx0.x, o, 64); acc.dx0.y += __shfl_xor(acc.dx0.y, o, 64); acc.dx0.z += __shfl_xor(acc.dx0.z, o, 64);
        acc.dy0.x += __shfl_xor(acc.dy0.x, o, 64); acc.dy0.y += __shfl_xor(acc.dy0.y, o, 64); acc.dy0.z += __shfl_xor(acc.dy0.z, o, 64);
        acc.dx1.x += __shfl_xor(acc.dx1.x, o, 64); acc.dx1.y += __shfl_xor(acc.dx1.y, o, 64); acc.dx1.z += __shfl_xor(acc.dx1.z, o, 64);
        acc.dy1.x += __shfl_xor(acc.dy1.x, o, 64); acc.dy1.y += __shfl_xor(acc.dy1.y, o, 64); acc.dy1.z += __shfl_xor(acc.dy1.z, o, 64);
    }
    if (valid && c == 0) {
        size_t i = ((size_t)y * W + x) * 3;
        a.img[i] = acc.r.x; a.img[i + 1] = acc.r.y; a.img[i + 2] = acc.r.z;
        a.cx0[i] = acc.dx0.x; a.cx0[i + 1] = acc.dx0.y; a.cx0[i + 2] = acc.dx0.z;
        a.cy0[i] = acc.dy0.x; a.cy0[i + 1] = acc.dy0.y; a.cy0[i + 2] = acc.dy0.z;
        a.cx1[i] = acc.dx1.x; a.cx1[i + 1] = acc.dx1.y; a.cx1[i + 2] = acc.dx1.z;
        a.cy1[i] = acc.dy1.x; a.cy1[i + 1] = acc.dy1.y; a.cy1[i + 2] = acc.dy1.z;
    }
    flush_counters(a, lc, tc, COUNT);
}

// TILE stream: bit-for-bit the reference's RNG order — one PCG stream per 16x16 tile, pixels y-outer /
// x-inner, samples innermost (src/render.cpp:281-309). Serial per tile => one lane per tile. For checks.
template <bool COUNT>
__global__ __launch_bounds__(64) void gdpt_render_tile_stream(DevSceneView sv, KernelArgs a, int ntx, int nty) {
    __shared__ int s_stack[GDPT_BVH_MAX_DEPTH * 64];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x * 64 + tid;
    LaneCounters lc = {0, 0, 0};
    TraceCounters tc = {0, 0};
    const int W = sv.cam.width, H = sv.cam.height;
    if (tile < ntx * nty) {
        const int tx = tile % ntx, ty = tile / ntx;
        Pcg rng = pcg_init((unsigned long long)(ty * ntx + tx));
        const int x0 = tx * 16, x1 = min(x0 + 16, W), y0 = ty * 16, y1 = min(y0 + 16, H);
        for (int y = y0; y < y1; y++) {
            if (y < a.row_begin || y >= a.row_end) continue;   // bands must be whole tile rows in this mode
            for (int x = x0; x < x1; x++) {
                Accum acc; acc.r = acc.dx0 = acc.dy0 = acc.dx1 = acc.dy1 = splat(0);
                for (int s = 0; s < a.spp; s++) {
                    SampleOut so;
                    grad_sample<COUNT>(sv, a.max_depth, x, y, rng, s_stack + tid, 64, so, lc, tc);
                    accumulate(acc, so, (double)a.spp, lc);
                }
                size_t i = ((size_t)y * W + x) * 3;
                a.img[i] = acc.r.x; a.img[i + 1] = acc.r.y; a.img[i + 2] = acc.r.z;
                a.cx0[i] = acc.dx0.x; a.cx0[i + 1] = acc.dx0.y; a.cx0[i + 2] = acc.dx0.z;
                a.cy0[i] = acc.dy0.x; a.cy0[i + 1] = acc.dy0.y; a.cy0[i + 2] = acc.dy0.z;
                a.cx1[i] = acc.dx1.x; a.cx1[i + 1] = acc.dx1.y; a.cx1[i + 2] = acc.dx1.z;
                a.cy1[i] = acc.dy1.x; a.cy1[i + 1] = acc.dy1.y; a.cy1[i + 2] = acc.dy1.z;
            }
        }
    }
    flush_counters(a, lc, tc, COUNT);
}

} // namespace gd

namespace gdpt {

const char *render_kernel_name(int rng_scheme) {
    return rng_scheme == GDPT_RNG_TILE ? "gdpt_render_tile_stream" : "gdpt_render_sample_stream";
}

void launch_render(const DevSceneView &sv, const RenderLaunch &rl, hipStream_t stream) {
    gd::KernelArgs a{};
    a.spp = rl.spp; a.row_begin = rl.row_begin; a.row_end = rl.row_end; a.max_depth = rl.max_depth;
    a.img = rl.img; a.cx0 = rl.cx0; a.cy0 = rl.cy0; a.cx1 = rl.cx1; a.cy1 = rl.cy1; a.counters = rl.counters;
    const int W = sv.cam.width, rows = rl.row_end - rl.row_begin;
    if (W <= 0 || rows <= 0 || rl.spp <= 0) throw std::runtime_error("launch_render: empty image band or spp <= 0");
    if (rl.rng_scheme == GDPT_RNG_TILE) {
        int ntx = (W + 15) / 16, nty = (sv.cam.height + 15) / 16;
        int blocks = (ntx * nty + 63) / 64;
        if (rl.count_traversal) hipLaunchKernelGGL(gd::gdpt_render_tile_stream<true>, dim3(blocks), dim3(64), 0, stream, sv, a, ntx, nty);
        else hipLaunchKernelGGL(gd::gdpt_render_tile_stream<false>, dim3(blocks), dim3(64), 0, stream, sv, a, ntx, nty);
    } else if (rl.rng_scheme == GDPT_RNG_SAMPLE) {
        // lanes per pixel: enough lanes to fill the chip (>= ~4 waves per SIMD over 256 CUs) but never more than spp
        long long pixels = (long long)W * rows;
        int log2k = 0;
        while ((1 << (log2k + 1)) <= rl.spp && log2k < 6 && (pixels << log2k) < (1LL << 19)) log2k++;
        a.log2k = log2k;
        int ppb = gd::kBlock >> log2k;                 // pixels per block
        a.tile_w = ppb >= 16 ? 16 : ppb;
        a.tile_h = ppb / a.tile_w;
        a.tiles_x = (W + a.tile_w - 1) / a.tile_w;
        int tiles_y = (rows + a.tile_h - 1) / a.tile_h;
        dim3 grid((unsigned)(a.tiles_x * tiles_y));
        if (rl.count_traversal) hipLaunchKernelGGL(gd::gdpt_render_sample_stream<true>, grid, dim3(gd::kBlock), 0, stream, sv, a);
        else hipLaunchKernelGGL(gd::gdpt_render_sample_stream<false>, grid, dim3(gd::kBlock), 0, stream, sv, a);
    } else {
        throw std::runtime_error("launch_render: unknown rng_scheme");
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) throw std::runtime_error(std::string("render kernel launch failed: ") + hipGetErrorString(e));
}

} // namespace gdpt
