/* gdpt.h — C ABI of the MI355X-native gradient-domain path tracing hot path.
 *
 * This is the drop-in boundary for LaJolla's `Integrator::GradPath` path
 * (reference: vedrocks15/Gradient-Based-Path-Tracing). The reference has no FFI
 * layer; its seams are C++ functions linked statically. Each entry point below
 * names the reference interface it replaces (file:line under /root/reference).
 *
 *   gdpt_parse_scene        <- parse_scene()            src/parsers/parse_scene.h:9, parse_scene.cpp:1615-1630
 *   gdpt_scene_upload       <- Scene::Scene()           src/scene.cpp:4-53   (Embree BVH build -> own BVH2 + HBM upload)
 *   gdpt_render             <- gradient_path_render()   src/render.cpp:257-333 (tile loop + grad_path_tracing, src/path_tracing.h:354-1050)
 *   gdpt_assemble           <- gradient assembly        src/render.cpp:336-350
 *   gdpt_poisson_solve      <- fourierSolve()           src/render.cpp:172-254 (argument-for-argument)
 *   gdpt_gradient_path_render <- gradient_path_render() src/render.cpp:257-370 (whole: render + assemble + solve)
 *   gdpt_imwrite            <- imwrite()                src/image.cpp:135-173
 *   gdpt_multi_*            <- parallel_for tile pool   src/parallel.cpp:183-256, src/render.cpp:271-277 (bands over several GPUs)
 *
 * Plain pointers and sizes only; no C++/torch types. All images are row-major
 * `data[(y*W+x)*3+c]` fp64, exactly `Image3::data` (src/image.h:13-39).
 * Every function returns 0 on success, non-zero on failure; the message is
 * available from gdpt_last_error() (thread-local).
 */
#ifndef GDPT_H
#define GDPT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scene description (host memory, flattened `Scene`, src/scene.h:40-80) ---- */

enum { GDPT_TEX_CONSTANT = 0, GDPT_TEX_IMAGE = 1, GDPT_TEX_CHECKERBOARD = 2 }; /* src/texture.h:83-102 */

/* Texture<Real> uses v0[0]/v1[0]; Texture<Spectrum> uses all three channels. */
typedef struct GdptTexture {
    int32_t type;
    int32_t image_id;         /* IMAGE: index into GdptSceneDesc::images (1- or 3-channel) */
    double v0[3];             /* CONSTANT: value; CHECKERBOARD: color0 */
    double v1[3];             /* CHECKERBOARD: color1 */
    double uscale, vscale, uoffset, voffset;
} GdptTexture;

/* Order follows the std::variant in src/material.h:82-90. */
enum {
    GDPT_MAT_LAMBERTIAN = 0, GDPT_MAT_ROUGHPLASTIC = 1, GDPT_MAT_ROUGHDIELECTRIC = 2,
    GDPT_MAT_DISNEY_DIFFUSE = 3, GDPT_MAT_DISNEY_METAL = 4, GDPT_MAT_DISNEY_GLASS = 5,
    GDPT_MAT_DISNEY_CLEARCOAT = 6, GDPT_MAT_DISNEY_SHEEN = 7, GDPT_MAT_DISNEY_BSDF = 8
};

/* Texture slots per material type (member order of the structs in src/material.h:12-80):
 *   LAMBERTIAN       0 reflectance
 *   DISNEY_DIFFUSE   0 base_color 1 roughness 2 subsurface
 *   DISNEY_METAL     0 base_color 1 roughness 2 anisotropic
 *   DISNEY_GLASS     0 base_color 1 roughness 2 anisotropic            (+eta)
 *   DISNEY_CLEARCOAT 0 clearcoat_gloss
 *   DISNEY_SHEEN     0 base_color 1 sheen_tint
 *   DISNEY_BSDF      0 base_color 1 specular_transmission 2 metallic 3 subsurface 4 specular
 *                    5 roughness 6 specular_tint 7 anisotropic 8 sheen 9 sheen_tint
 *                    10 clearcoat 11 clearcoat_gloss                    (+eta)
 */
#define GDPT_MAT_MAX_TEX 12
typedef struct GdptMaterial {
    int32_t type;
    int32_t _pad;
    double eta;
    GdptTexture tex[GDPT_MAT_MAX_TEX];
} GdptMaterial;

typedef struct GdptImage {     /* level 0 of a Mipmap1/Mipmap3 (src/mipmap.h:9-12); mips are built by the library */
    int32_t width, height, channels; /* channels 1 or 3 */
    int32_t _pad;
    const double *texels;      /* width*height*channels, row-major */
} GdptImage;

enum { GDPT_SHAPE_SPHERE = 0, GDPT_SHAPE_TRIMESH = 1 }; /* src/shape.h:44 */

typedef struct GdptShape {
    int32_t type;
    int32_t material_id;
    int32_t area_light_id;     /* -1 if not an emitter */
    int32_t num_vertices, num_triangles;
    int32_t _pad;
    double center[3];          /* sphere */
    double radius;             /* sphere */
    const double *positions;   /* 3*num_vertices (world space) */
    const int32_t *indices;    /* 3*num_triangles */
    const double *normals;     /* 3*num_vertices or NULL */
    const double *uvs;         /* 2*num_vertices or NULL */
} GdptShape;

typedef struct GdptLight {     /* DiffuseAreaLight (src/light.h), or the placeholder of the environment map (shape_id = -1,
                                  its data is GdptSceneDesc::envmap) so that light ids keep the reference's parse order */
    int32_t shape_id;
    int32_t _pad;
    double intensity[3];
} GdptLight;

enum { GDPT_FILTER_BOX = 0, GDPT_FILTER_TENT = 1, GDPT_FILTER_GAUSSIAN = 2 }; /* src/filter.h:31-46 */

typedef struct GdptCamera {    /* src/camera.h:10-26 */
    double sample_to_cam[16];  /* row-major 4x4 */
    double cam_to_world[16];
    int32_t width, height;
    int32_t filter_type;
    int32_t _pad;
    double filter_param;       /* Box/Tent: width; Gaussian: stddev */
} GdptCamera;

enum { GDPT_INTEGRATOR_PATH = 5, GDPT_INTEGRATOR_GRADPATH = 7, GDPT_INTEGRATOR_OTHER = -1 }; /* src/scene.h:14-23 */

typedef struct GdptEnvmap {    /* Envmap, src/light.h + src/lights/envmap.inl: lat-long image, uv = (azimuth/2pi, elevation/pi) */
    int32_t light_id;          /* index into lights[] */
    int32_t image_id;          /* 3-channel image */
    double scale;
    double to_world[16], to_local[16];   /* row-major; to_local = inverse(to_world) */
} GdptEnvmap;

typedef struct GdptSceneDesc {
    GdptCamera camera;
    int32_t integrator;
    int32_t samples_per_pixel; /* <sampler sampleCount>; the reference ignores it (src/render.cpp:293) */
    int32_t max_depth;         /* -1 = unbounded (RR only) */
    int32_t rr_depth;
    int32_t num_materials, num_shapes, num_lights, num_images;
    const GdptMaterial *materials;
    const GdptShape *shapes;
    const GdptLight *lights;
    const GdptImage *images;
    char output_filename[256]; /* film `filename`, default "image.exr" (src/parsers/parse_scene.cpp:15) */
    int32_t has_envmap;        /* the XML holds an <emitter type="envmap">: ignored by GradPath (src/path_tracing.h:982-985),
                                  sampled and looked up by the Path entry points */
    int32_t _pad;
    GdptEnvmap envmap;         /* valid when has_envmap */
} GdptSceneDesc;

/* ---- render parameters ---- */

/* How PCG32 streams (src/pcg.h:33-41) are assigned.
 *   TILE   : init_pcg32(tile_y*ntx+tile_x), pixels y-outer/x-inner, samples innermost — bit-for-bit the
 *            reference order (src/render.cpp:281-309). Serial per tile: one GPU lane per tile (slow; for checks).
 *   SAMPLE : init_pcg32((y*W+x)*spp + s) per sample — every sample independent (the throughput mode). */
enum { GDPT_RNG_TILE = 0, GDPT_RNG_SAMPLE = 2 };
/* How the four offset paths follow the base path.
 * GDPT_SHIFT_REFERENCE: what grad_path_tracing does (src/path_tracing.h:351-560): offsets replay the base path's random
 *   numbers and never rejoin it. The drop-in, parity-tested behaviour.
 * GDPT_SHIFT_RECONNECT: the offset path's first vertex is reconnected to the base path's second vertex, with the
 *   Jacobian and the balance-heuristic weights of the reference's own sketch (small_gdpt.py:163-219, :380-420); the
 *   gradient buffers then are unbiased estimates of I(x)-I(x-1), I(y)-I(y-1) and the Poisson solve denoises. Same
 *   base path (and primal image) as the reference mode. GDPT_RNG_SAMPLE only. Not part of the reference's output. */
enum { GDPT_SHIFT_REFERENCE = 0, GDPT_SHIFT_RECONNECT = 1 };

typedef struct GdptRenderParams {
    int32_t spp;               /* <=0: use scene samples_per_pixel; the reference hard-codes 1000 (src/render.cpp:293) */
    int32_t rng_scheme;        /* GDPT_RNG_* */
    int32_t row_begin, row_end;/* render rows [row_begin,row_end) only (multi-GPU bands); 0,0 = whole image.
                                  Rows outside the band are left untouched in the output buffers. */
    int32_t max_depth_override;/* 0 = use scene; else value */
    int32_t shift_mode;        /* GDPT_SHIFT_* (gdpt_render* only; 0 = the reference's behaviour) */
    int32_t plan_rows;         /* 0 = film height. A pixel's samples are cut into work items ("chunks") whose partial sums are
                                  merged in chunk order; the cut is made for a band of this many rows, so that a small band
                                  rendered alone (one device of N) still consists of many short items instead of a few long
                                  ones. Renders of the same pixel with the same plan_rows are bit-identical whatever band
                                  they are part of: the multi-device hosts pass the rows of their largest band, and a
                                  single-device render given the same value reproduces their images bit for bit. */
    int32_t reserved;          /* 0 */
} GdptRenderParams;

typedef struct GdptRenderStats {
    uint64_t samples;          /* grad_path_tracing calls */
    uint64_t rays;             /* closest-hit queries (5 primaries + 1 per bounce; the reference's 4 tfar=0 rays are omitted) */
    uint64_t bounces;          /* bounce-loop iterations */
    uint64_t nodes_visited;    /* BVH nodes fetched (counting builds only, else 0) */
    uint64_t tris_tested;      /* triangles tested (counting builds only, else 0) */
    uint64_t nonfinite_samples;/* samples whose record held a NaN/Inf (propagated, as the reference does) */
    double render_ms;          /* device time of the render kernel(s), HIP events */
    uint64_t node_bytes;       /* size of one fetched BVH node in the form this render walked (64: BVH2, 128: BVH4) */
    /* SIMT utilisation of the persistent kernel (counting builds only, else 0): loop trips counted once per wave */
    uint64_t wave_node_trips;  /* node-visit trips;  nodes_visited / (64 * this) = lane utilisation of the box tests */
    uint64_t wave_leaf_trips;  /* leaf-test trips */
    uint64_t wave_steps;       /* lane-machine steps (one pending ray per live lane each) */
    uint64_t lane_steps;       /* live lanes summed over those steps */
} GdptRenderStats;

typedef struct GdptPoissonStats {
    int32_t iterations;        /* CG iterations (0 for the direct DCT solver) */
    int32_t solver;            /* GDPT_SOLVER_* actually used */
    double rel_residual;       /* ||W(h - A f)|| / ||W h|| at exit (CG) */
    double solve_ms;           /* device time, HIP events */
} GdptPoissonStats;

/* CG: conjugate gradients on W(alpha I - L) f = W h + DC shift (matches the DCT solve to the CG tolerance; differs
 *     from the reference by its fp32-lambda quirk, ~3e-9).
 * DCT_MFMA (the default): direct solve, exact reference operator. The two 1-D DCT-I passes are fp64 GEMMs against fixed
 *     cosine matrices, run by hand-written fp64 MFMA kernels on HALF the flops (even / odd fold of the DCT-I matrix); their
 *     epilogues carry the spectral division, the DC override and the final scaling (5 launches).
 * DCT: the same solve with the passes as rocBLAS dgemm_strided_batched on the unfolded matrices (8 launches): kept as the
 *     measured alternative (tests/time_poisson.py). */
enum { GDPT_SOLVER_CG = 0, GDPT_SOLVER_DCT = 1, GDPT_SOLVER_DCT_MFMA = 2 };
#define GDPT_SOLVER_DEFAULT GDPT_SOLVER_DCT_MFMA   /* what gdpt_poisson_solve, gdpt_gradient_path_render and gdpt_multi_* use */

typedef struct GdptScene GdptScene;   /* opaque: device-resident scene (BVH2 + BVH4, triangles, materials, textures) */

/* ---- host-side scene ingest (Mitsuba-0.x XML subset) ---- */
int gdpt_parse_scene(const char *xml_path, GdptSceneDesc **out_desc);
/* Same with the <film> extent replaced (0 = keep the scene's): the benchmark configurations quote their own film
 * sizes (sponza at 1280x720, cbox at 1024x1024); the camera is built for the new aspect ratio, as if the XML said so. */
int gdpt_parse_scene_film(const char *xml_path, int film_width, int film_height, GdptSceneDesc **out_desc);
void gdpt_free_scene_desc(GdptSceneDesc *desc);

/* ---- device scene ---- */
int gdpt_scene_upload(const GdptSceneDesc *desc, int device, GdptScene **out_scene);
void gdpt_scene_free(GdptScene *scene);
/* Copies the BVH the library built (for inspection/tests): node count, triangle count, depth. */
int gdpt_scene_info(const GdptScene *scene, int32_t *num_nodes, int32_t *num_tris, int32_t *num_spheres, int32_t *bvh_depth);

/* ---- hot path, host buffers (caller-owned W*H*3 doubles each, as Image3::data) ---- */
int gdpt_render(GdptScene *scene, const GdptRenderParams *params,
                double *img, double *cx0, double *cy0, double *cx1, double *cy1,
                GdptRenderStats *stats /* nullable */);

/* ---- hot path, device buffers (hipMalloc'd / torch CUDA tensors; data stays in HBM) ----
 * `stream` is a hipStream_t passed as void* (NULL = default stream). Asynchronous w.r.t. the host
 * unless `stats` is non-NULL (then it synchronises to read the counters).
 * A GdptScene owns ONE set of launch scratch (work queue, per-item partial sums, counters, bounce log): at most one
 * render of a given scene handle may be in flight at a time. Renders issued on the same stream are ordered by it;
 * to overlap renders on different streams (or devices) upload one scene handle per stream — the scene tables are
 * a few MB. The reference's render() is likewise called once, synchronously (src/main.cpp:40). */
int gdpt_render_device(GdptScene *scene, const GdptRenderParams *params,
                       double *d_img, double *d_cx0, double *d_cy0, double *d_cx1, double *d_cy1,
                       void *stream, GdptRenderStats *stats /* nullable */);

/* ---- Integrator::Path (path_render, src/render.cpp:74-117 over path_tracing, src/path_tracing.h:13-348) ----
 * Unidirectional path tracing with next-event estimation + MIS for scenes lit by area emitters (meshes, spheres)
 * and / or an environment map (src/lights/envmap.inl): img = mean over spp of path_tracing(x, y). Same parameters,
 * RNG schemes and stats as gdpt_render; scenes without any emitter are refused (error status). */
int gdpt_path_render(GdptScene *scene, const GdptRenderParams *params, double *img, GdptRenderStats *stats /* nullable */);
int gdpt_path_render_device(GdptScene *scene, const GdptRenderParams *params, double *d_img,
                            void *stream, GdptRenderStats *stats /* nullable */);

/* c=img; cx=cx0(x,y)+cx1(x-1,y); cy=cy0(x,y)+cy1(x,y-1)  (src/render.cpp:340-350). Device pointers. */
int gdpt_assemble_device(int width, int height,
                         const double *d_img, const double *d_cx0, const double *d_cy0,
                         const double *d_cx1, const double *d_cy1,
                         double *d_c, double *d_cx, double *d_cy, void *stream);

/* Screened Poisson reconstruction; same arguments as fourierSolve (src/render.cpp:172-175). Host pointers.
 * Uses GDPT_SOLVER_DEFAULT: the reference's algorithm itself (DCT-I, fp32-rounded eigenvalue, DC override), the
 * transform evaluated as fp64 MFMA GEMMs on the GPU. */
int gdpt_poisson_solve(int width, int height,
                       const double *imgData, const double *imgGradX, const double *imgGradY,
                       double dataCost, double *imgOut);
/* Same with solver choice, tolerance and stats. solver: GDPT_SOLVER_*. tol<=0 selects the default 1e-10. */
int gdpt_poisson_solve_ex(int width, int height,
                          const double *imgData, const double *imgGradX, const double *imgGradY,
                          double dataCost, double *imgOut,
                          int solver, double tol, int max_iters, GdptPoissonStats *stats /* nullable */);
/* Device-pointer variant (inputs/outputs in HBM). The library keeps its scratch per (device, stream). With
 * GDPT_SOLVER_DCT_MFMA / GDPT_SOLVER_DCT and stats == NULL the call only enqueues work on `stream` (no event, no host wait); with stats it
 * brackets the solve with HIP events and waits for it. GDPT_SOLVER_CG always waits (host-side convergence check). */
int gdpt_poisson_solve_device(int width, int height,
                              const double *d_c, const double *d_gx, const double *d_gy,
                              double dataCost, double *d_out,
                              int solver, double tol, int max_iters,
                              void *stream, GdptPoissonStats *stats /* nullable */);

/* gdpt_assemble_device followed by gdpt_poisson_solve_device on its outputs — what gradient_path_render does after the tile loop
 * (src/render.cpp:340-353) — as one call for a film that sits whole on one device: with the DCT solvers the assembly and the
 * right-hand side of the solve are one pass over the film. d_c / d_cx / d_cy are written as gdpt_assemble_device writes them; every
 * result has the bits of the two separate calls. Enqueue-only unless `stats` (as gdpt_poisson_solve_device). */
int gdpt_assemble_solve_device(int width, int height,
                               const double *d_img, const double *d_cx0, const double *d_cy0,
                               const double *d_cx1, const double *d_cy1,
                               double *d_c, double *d_cx, double *d_cy,
                               double dataCost, double *d_out,
                               int solver, double tol, int max_iters,
                               void *stream, GdptPoissonStats *stats /* nullable */);

/* Drops the solver scratch the library keeps for `stream` on the current device (buffers, handles, events). Owners of a
 * stream call it before destroying the stream, with no solve in flight on it; a stream the library never saw is fine. */
int gdpt_poisson_forget_stream(void *stream);

/* Whole Integrator::GradPath: render -> assemble -> solve -> final image (host, W*H*3 doubles).
 * Optionally returns the five raw buffers too (any of them may be NULL). */
int gdpt_gradient_path_render(GdptScene *scene, const GdptRenderParams *params, double dataCost,
                              double *out_image,
                              double *img, double *cx0, double *cy0, double *cx1, double *cy1,
                              GdptRenderStats *rstats, GdptPoissonStats *pstats);

/* ---- several devices of one node: the tile loop sharded into row bands ----
 * Replaces the reference's only parallelism, parallel_for over 16x16 tiles on a std::thread pool
 * (src/render.cpp:271-277, src/parallel.cpp:183-256): contiguous bands of whole tile rows go to the devices, one host
 * thread per device; between render and solve each device sends the last cy1 row of its band to the next one
 * (src/render.cpp:345-349 needs it), assembles c, cx, cy for its own band, and the three images are all-gathered in
 * place; the global solve runs on devices[0]. Results equal the single-device entry points bit for bit (same per-sample
 * RNG streams, same per-pixel summation order). */
#define GDPT_MULTI_MAX_DEVICES 16
/* Transport of the halo row and the all-gather.
 *   GDPT_EXCHANGE_RCCL:      ncclSend/ncclRecv + grouped in-place ncclAllGather (ncclBroadcast per band when the bands are
 *                            ragged) on one communicator per device; devices must be distinct. NOT YET RUN BETWEEN TWO
 *                            DEVICES (every box this library was developed on had one GPU): correct by construction
 *                            and by the one-device tests only. A failure behind the first collective aborts the
 *                            communicators (ncclCommAbort) and spends the handle: create a new one.
 *   GDPT_EXCHANGE_PEER_COPY: direct device-to-device copies (hipMemcpyPeerAsync over xGMI, ordered by events): every device
 *                            pushes its band to every other one, N-1 links at once. Accepts a device twice. */
enum { GDPT_EXCHANGE_RCCL = 0, GDPT_EXCHANGE_PEER_COPY = 1 };
typedef struct GdptMultiConfig {
    int32_t num_devices;       /* 1..GDPT_MULTI_MAX_DEVICES */
    int32_t exchange;          /* GDPT_EXCHANGE_* */
    int32_t devices[GDPT_MULTI_MAX_DEVICES];   /* HIP device ordinals, band order */
    int32_t balance;           /* 0: bands of equal tile-row count (gdpt_band_rows). 1: bands of equal measured cost — a pilot
                                  render (1 spp, first device, at create time) counts the rays of every tile row and the
                                  bands are cut by gdpt_band_rows_weighted */
    int32_t reserved;          /* 0 */
} GdptMultiConfig;
typedef struct GdptMultiStats {
    int32_t num_devices, exchange;
    int32_t row_begin[GDPT_MULTI_MAX_DEVICES], row_end[GDPT_MULTI_MAX_DEVICES];
    double render_ms[GDPT_MULTI_MAX_DEVICES];  /* device time of each band's render (HIP events) */
    double render_ms_max;      /* slowest band */
    double exchange_ms;        /* halo + assembly + all-gather, slowest device (includes waiting for the slowest band) */
    double solve_ms;           /* Poisson solve on devices[0] */
    double wall_ms;            /* host wall time of the call, D2H of the results included */
} GdptMultiStats;
typedef struct GdptMulti GdptMulti;   /* opaque: one uploaded scene, stream and image set per device (+ communicators) */

/* Rows [row_begin,row_end) of band `band` of `num_bands`: whole 16-pixel tile rows (src/render.cpp:271), balanced by
 * tile-row count, in band order. Host only. */
int gdpt_band_rows(int height, int num_bands, int band, int32_t *row_begin, int32_t *row_end);
/* The same with bands of (nearly) equal COST: tile_row_cost[t] >= 0 is the cost of tile row t (num_tile_rows =
 * ceil(height / 16) of them); the split minimises the cost of the most expensive band among all contiguous splits into
 * whole tile rows, rank-ordered. Host only. */
int gdpt_band_rows_weighted(int height, int num_bands, int band, const double *tile_row_cost, int num_tile_rows,
                            int32_t *row_begin, int32_t *row_end);
/* Cost of every 16-pixel tile row of the film: rays traced by a pilot render of that tile row alone at `spp` samples per
 * pixel (GDPT_RNG_SAMPLE, reference shift). Exact counts, so every caller — every rank of a sharded run — gets the same
 * numbers. `capacity` >= ceil(height / 16). Blocking; what the multi-device hosts balance their bands by. */
int gdpt_tile_row_costs(GdptScene *scene, int spp, double *cost, int capacity);
/* The same partition on a cost per pixel ROW (`row_cost[height]`), cuts on multiples of `granularity_rows` (1, 2, 4, 8 or 16): the
 * persistent kernels anchor their 16x16 work items at a band's first row, so a band of the GDPT_RNG_SAMPLE streams need not consist
 * of whole tile rows (GDPT_RNG_TILE does: granularity 16). Host only. */
int gdpt_band_rows_from_row_costs(int height, int num_bands, int band, const double *row_cost, int granularity_rows,
                                  int32_t *row_begin, int32_t *row_end);
int gdpt_multi_create(const GdptSceneDesc *desc, const GdptMultiConfig *config, GdptMulti **out);
/* Feedback between frames: `band_ms[i]` = measured render time of band i in the last call (GdptMultiStats.render_ms). The handle
 * keeps a cost model of the film's rows (uniform, or the pilot's with config.balance); every band's rows are rescaled so that the
 * band's modelled share equals its measured share, and the bands are cut again, on multiples of `granularity_rows`. Call between
 * two renders, from the thread that renders. The images of later calls differ from earlier ones in the last bits of their sums
 * (the work items follow the largest band, GdptRenderParams.plan_rows); GDPT_RNG_TILE renders need granularity 16. */
int gdpt_multi_rebalance(GdptMulti *multi, const double *band_ms, int granularity_rows);
void gdpt_multi_free(GdptMulti *multi);
/* Whole Integrator::GradPath on the device set; arguments as gdpt_gradient_path_render (params->row_begin/row_end
 * must be 0: the library chooses the bands). Blocking. */
int gdpt_multi_gradient_path_render(GdptMulti *multi, const GdptRenderParams *params, double dataCost, double *out_image,
                                    double *img, double *cx0, double *cy0, double *cx1, double *cy1,
                                    GdptRenderStats *rstats /* nullable */, GdptMultiStats *mstats /* nullable */);
/* gdpt_assemble_device restricted to rows [row_begin,row_end) (0,0 = all): what one band owner runs. */
int gdpt_assemble_rows_device(int width, int height, int row_begin, int row_end,
                              const double *d_img, const double *d_cx0, const double *d_cy0,
                              const double *d_cx1, const double *d_cy1,
                              double *d_c, double *d_cx, double *d_cy, void *stream);

/* ---- output ---- */
/* By suffix: ".pfm" (fp32, header "PF\nW H\n-1\n", rows as stored) or ".exr" (fp16 RGB scanline). */
int gdpt_imwrite(const char *filename, int width, int height, const double *rgb);

/* ---- input images (host only) ----
 * imread1 / imread3 of the reference (src/image.cpp:26-133) for channels = 1 / 3: fp64 texels, row-major, top row
 * first. ".jpg"/".jpeg": own baseline decoder returning stb_image's samples, widened like stbi_loadf; ".pfm"; other
 * suffixes through a pre-decoded "<file>.gdtex" companion. Free with gdpt_image_free. */
int gdpt_imread(const char *filename, int channels, int *width, int *height, double **texels);
void gdpt_image_free(double *texels);

/* ---- diagnostics (host only, no GPU) ----
 * Builds the acceleration structure that gdpt_scene_upload would build over `n` primitive boxes (bounds6 = n x
 * {min xyz, max xyz}, fp32) — the replacement for the reference's Embree commit, src/scene.cpp:20-31 — and verifies it:
 * every primitive sits in exactly one leaf, every child box encloses the boxes below it, in the BVH2 and in the
 * collapsed wide forms (4-wide fp32 boxes; 8-wide boxes quantised on a per-node grid, whose exactly evaluated grid
 * boxes must enclose their subtrees and whose stack bound must hold). stats: [0] BVH2 nodes, [1] BVH2 depth, [2] BVH4
 * nodes, [3] BVH4 node arity used (2..4), [4] traversal-stack bound of the BVH4, [5] leaves, [6] max primitives per
 * leaf, [7] BVH8 nodes. */
int gdpt_bvh_check(const float *bounds6, int n, int32_t stats[8]);
/* The same for the build with spatial splits (host/sbvh.cpp; what gdpt_scene_upload uses for meshes walked from HBM): over `n`
 * fp32 triangles (tri_verts9 = n x 3 vertices x xyz) with `budget` extra references per primitive allowed. Verifies that child
 * boxes enclose the reference boxes below them, that every triangle is referenced, the depth and stack bounds of the collapsed
 * form, and COVERAGE: for `samples_per_tri` points on every triangle (its vertices, edge midpoints, centroid, then points from a
 * fixed lattice), descending from the root through every child box that contains the point reaches a leaf that references the
 * triangle — a ray that hits the triangle there cannot miss it in the tree. stats: [0] BVH2 nodes, [1] BVH2 depth, [2] references
 * (>= n), [3] BVH4 nodes, [4] BVH4 stack bound, [5] leaves, [6] / [7] 1000 x the surface-area cost of the BVH4: inner nodes entered /
 * triangles tested per random line through the root box. */
int gdpt_sbvh_check(const float *tri_verts9, int n, double budget, int samples_per_tri, int32_t stats[8]);

const char *gdpt_last_error(void);
/* "gfx950" etc. of the device the library's kernels were built for, and the running device name. */
const char *gdpt_build_arch(void);

#ifdef __cplusplus
}
#endif
#endif /* GDPT_H */
