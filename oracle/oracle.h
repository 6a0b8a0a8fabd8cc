/* oracle.h — C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY. This is a CPU restatement (fp64 shading, fp32 ray/triangle arithmetic)
 * of the reference's Integrator::GradPath path, used by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg as the checker. Nothing in the product path may include, link or
 * call it. See oracle/README.md for how it is pinned.
 */
#ifndef GDPT_ORACLE_H
#define GDPT_ORACLE_H
#include "../include/gdpt.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct OracleScene OracleScene;

typedef struct OracleVertex {       /* PathVertex, src/intersection.h:15-37 */
    double position[3];
    double geometric_normal[3];
    double frame_x[3], frame_y[3], frame_n[3];
    double st[2], uv[2];
    double uv_screen_size, mean_curvature, ray_radius;
    int32_t shape_id, primitive_id, material_id, gid;
    double t;                       /* fp32 hit distance widened */
} OracleVertex;

typedef struct OracleSampleRecord { /* GraidentPTRadiance, src/intersection.h:65-77 */
    double radiance[3], contrib[3], contribX0[3], contribX1[3], contribY0[3], contribY1[3];
    double prob, wX0, wY0, wX1, wY1;
    int32_t bounces;                /* bounce-loop iterations executed */
    int32_t primary_miss;
    int32_t valid0[4];              /* offsets x0,x1,y0,y1 valid after the material test (path_tracing.h:424-443) */
    int32_t rng_draws;
} OracleSampleRecord;

typedef struct OracleStats {
    uint64_t samples, rays, bounces, primary_misses, x0_valid_initial, nonfinite_samples;
    uint64_t nodes_visited, tris_tested;
    double seconds;
} OracleStats;

/* use_bvh: 0 = brute force over all primitives, 1 = the oracle's own median-split BVH (same closest hit). */
OracleScene *oracle_scene_create(const GdptSceneDesc *desc, int use_bvh);
void oracle_scene_free(OracleScene *s);
double oracle_intersection_epsilon(const OracleScene *s);

/* PCG32, src/pcg.h */
void oracle_pcg_init(uint64_t stream, uint64_t *state, uint64_t *inc);
uint32_t oracle_pcg_next(uint64_t *state, uint64_t inc);
double oracle_pcg_next_double(uint64_t *state, uint64_t inc);

/* sample_primary, src/camera.cpp:23-47 */
void oracle_sample_primary(const OracleScene *s, double sx, double sy, double org[3], double dir[3]);
/* filter sample, src/filters/*.inl */
void oracle_filter_sample(int filter_type, double param, double u0, double u1, double out[2]);

/* intersect, src/intersection.cpp:7-65. ray_diff: {radius, spread}. Returns 1 on hit. */
int oracle_intersect(const OracleScene *s, const double org[3], const double dir[3], double tnear, double tfar,
                     const double ray_diff[2], OracleVertex *out);
/* compute_shading_info for triangle `gid` at barycentrics st given a (normalised) geometric normal.
 * out: uv[2], frame_x[3], frame_y[3], frame_n[3], mean_curvature, inv_uv_size  (12 doubles) */
void oracle_shading_info_tri(const OracleScene *s, int gid, const double st[2], const double gn[3], double out[13]);

/* BSDF trio on an explicit material (src/material.cpp:90-119). Return value of sample: 1 = record valid. */
void oracle_bsdf_eval(const OracleScene *s, const GdptMaterial *m, const double dir_in[3], const double dir_out[3],
                      const OracleVertex *v, double f[3]);
double oracle_bsdf_pdf(const OracleScene *s, const GdptMaterial *m, const double dir_in[3], const double dir_out[3],
                       const OracleVertex *v);
int oracle_bsdf_sample(const OracleScene *s, const GdptMaterial *m, const double dir_in[3], const OracleVertex *v,
                       const double rnd_uv[2], double rnd_w, double dir_out[3], double *eta, double *roughness);
/* Texture eval (src/texture.h:112-159): channels 1 or 3 */
void oracle_texture_eval(const OracleScene *s, const GdptTexture *t, int channels, const double uv[2], double footprint, double out[3]);

/* One grad_path_tracing call (src/path_tracing.h:354-1050, A-semantics) on pixel (x,y); advances the RNG. */
void oracle_grad_sample(const OracleScene *s, int x, int y, uint64_t *state, uint64_t inc, OracleSampleRecord *rec);

/* gradient_path_render tile loop (src/render.cpp:257-333). Buffers W*H*3 doubles, zero-initialised by the caller
 * (accumulated with +=, as the reference does). threads<=0: hardware concurrency. */
int oracle_render(const OracleScene *s, int spp, int rng_scheme, int row_begin, int row_end, int threads,
                  double *img, double *cx0, double *cy0, double *cx1, double *cy1, OracleStats *stats);

/* ---- Integrator::Path: path_tracing with next-event estimation + MIS, area lights only ---- */
/* Light selection table (src/scene.cpp:44-66): pmf[num_lights], cdf[num_lights+1]. */
void oracle_light_table(const OracleScene *s, double *pmf, double *cdf);
/* sample_point_on_shape / pdf_point_on_shape (src/shapes/triangle_mesh.inl:24-58, src/shapes/sphere.inl:161-230).
 * out: position[3], normal[3]. */
void oracle_sample_point_on_shape(const OracleScene *s, int shape_id, const double ref_point[3], const double uv[2], double w, double out[6]);
double oracle_pdf_point_on_shape(const OracleScene *s, int shape_id, const double point[3], const double normal[3], const double ref_point[3]);
/* occluded(), src/intersection.cpp:67-85: any primitive accepting the fp32 ray in [tnear, tfar). */
int oracle_occluded(const OracleScene *s, const double org[3], const double dir[3], double tnear, double tfar);
/* One path_tracing call (src/path_tracing.h:13-348) on pixel (x,y); advances the RNG. The two sub-pixel numbers are
 * drawn x first (the left-to-right order of the compiler the reference was developed with). */
void oracle_path_sample(const OracleScene *s, int x, int y, uint64_t *state, uint64_t inc, double radiance[3], int32_t *bounces, int32_t *shadow_rays);
/* path_render tile loop (src/render.cpp:74-117): img = mean over spp of path_tracing. W*H*3 doubles, caller-zeroed.
 * Returns non-zero for scenes without any emitter. Environment maps: src/lights/envmap.inl. */
/* TableDist2D (src/table_dist.cpp:40-150) built over f (row-major, height rows): sample(rnd) -> uv, pdf(uv). */
void oracle_table2d(const double *f, int width, int height, const double *rnd, int n_rnd, double *uv_out, double *pdf_out, double *total);
/* GDPT_SHIFT_RECONNECT (sample-stream RNG only): restatement of csrc/hip/render_reconnect.hip; parity unpinned against the
 * reference (it has no such output), pinned by the mode's defining properties in tests/test_reconnect_shift.py. */
int oracle_reconnect_render(const OracleScene *s, int spp, int row_begin, int row_end, int threads,
                            double *img, double *cx0, double *cy0, double *cx1, double *cy1, OracleStats *stats);
int oracle_path_render(const OracleScene *s, int spp, int rng_scheme, int row_begin, int row_end, int threads,
                       double *img, OracleStats *stats);

/* gradient assembly, src/render.cpp:340-350 */
void oracle_assemble(int w, int h, const double *img, const double *cx0, const double *cy0,
                     const double *cx1, const double *cy1, double *c, double *cx, double *cy);

/* fourierSolve with a direct O(N^2)-per-line DCT-I in place of FFTW (REDFT00 definition), incl. the fp32 lambda
 * rounding and the DC override (src/render.cpp:172-254). Slow; small sizes only. */
void oracle_poisson_dct(int w, int h, const double *c, const double *gx, const double *gy, double alpha, double *out);

#ifdef __cplusplus
}
#endif
#endif
