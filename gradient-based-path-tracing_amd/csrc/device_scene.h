// device_scene.h — HBM layout of the flattened scene (POD, shared by host upload code and HIP kernels).
//
// Traversal data is fp32 (what Embree receives from the reference, src/intersection.cpp:15-24,
// src/shapes/triangle_mesh.inl:11-14); everything a shading computation touches is fp64
// (Real = double, src/lajolla.h:23).
#pragma once
#include <stdint.h>

#define GDPT_BVH_MAX_DEPTH 32      // builder guarantee; traversal stack has this many slots per lane
#define GDPT_LEAF_MAX_PRIMS 4
#define GDPT_SPHERE_FLAG 0x80000000u
#ifndef GDPT_HBM_BVH8
#define GDPT_HBM_BVH8 0            // 1: scenes walked from HBM use the quantised 8-wide tree (A/B builds; measured slower)
#endif
// The BVH8's traversal-stack bound may exceed the LDS stack by this many slots: a lane whose stack grows past
// GDPT_BVH_MAX_DEPTH entries (seven pushes per level make the bound of a full 8-wide tree 7 x its depth, although a ray
// that hits every box of every level does not occur in practice) continues in a private array.
#define GDPT_STACK_OVERFLOW 64
#ifndef GDPT_BVH8_SORT
#define GDPT_BVH8_SORT 1           // 1: hit children pushed far to near (sorted); 0: nearest first, the others in slot order
#endif
#ifndef GDPT_SPEC_LEAF
#define GDPT_SPEC_LEAF 0           // 1: every while-while walk sets leaves aside (A/B; the one-sided GradPath lane machine always does)
#endif
#define GDPT_CHILD_EMPTY INT32_MIN // child slot with no primitives (only in degenerate roots)

// BVH2 node, 64 B, 64-B aligned: both children's boxes live in the parent so one fetch decides both.
// child >= 0: index of an inner node. child < 0 (and != EMPTY): leaf, ~child = (first_prim << 2) | (count-1).
struct DevBvhNode {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    int32_t left, right;
    int32_t pad[2];
};

// BVH4 node, 128 B, 128-B aligned (scenes walked from HBM): four child boxes in SoA order, so one fetch decides four
// subtrees and a ray needs about half as many dependent fetches as with DevBvhNode. Made by collapsing the BVH2
// (host/bvh.cpp: collapse_bvh4); child encoding as above, unused slots hold GDPT_CHILD_EMPTY.
struct DevBvh4Node {
    float lo[3][4];      // lo[axis][child]
    float hi[3][4];
    int32_t child[4];
    int32_t pad[4];
};

// BVH8 node, 128 B, 128-B aligned (scenes walked from HBM): eight child boxes, each bound one byte on a per-node grid
// `org[axis] + q * scale[axis]`, lower bounds rounded down and upper bounds up, so a grid box
// contains the fp32 box of the BVH2 it was made from (host/bvh.cpp: collapse_bvh8). One 128-byte fetch decides eight
// subtrees. Child encoding as above; unused slots hold GDPT_CHILD_EMPTY.
struct DevBvh8Node {
    int32_t child[8];
    uint8_t qlo[3][8];   // qlo[axis][child]  (offset 32: the 48 quantised bytes are three aligned 16-byte loads)
    uint8_t qhi[3][8];
    float org[3];
    float scale[3];
    int32_t pad[6];
};

// BVH4 node with quantised child boxes, 64 B, 64-B aligned: the SAME tree as DevBvh4Node (node i here is node i there, same
// child references), each bound one byte on the node's own grid `org[axis] + q * scale[axis]`, lower bounds rounded down and
// upper bounds up (verified in double on the host), so a grid box contains the fp32 box it replaces. A lane fetches a node
// with four 16-byte loads instead of seven: the walk of a scene that lives in HBM is bound by the number of per-lane
// requests the texture-address unit has to serialise, not by bytes.
struct DevBvh4QNode {
    float org[3];
    float scale[3];
    uint8_t qlo[3][4];   // qlo[axis][child]
    uint8_t qhi[3][4];
    int32_t child[4];
};
#ifndef GDPT_HBM_Q4
#define GDPT_HBM_Q4 1              // scenes walked from HBM use DevBvh4QNode (0: the fp32 DevBvh4Node, A/B builds)
#endif

// Traversal record of one primitive, 48 B, in BVH leaf order.
// Triangle: v0, e1 = fl(v1-v0), e2 = fl(v2-v0) in fp32; gid = global triangle id (index into DevTriShade).
// Sphere:   gid = GDPT_SPHERE_FLAG | sphere index; the fp32 fields are unused (fp64 data in DevSphere).
struct DevPrim {
    float v0[3], e1[3], e2[3];
    uint32_t gid;
    uint32_t pad[2];
};

// fp64 per-triangle shading inputs (compute_shading_info, src/shapes/triangle_mesh.inl:77-169), 232 B.
// Everything that does not depend on the hit point is evaluated once on the host with the reference's formulas
// (same operation order, no FMA contraction): dp/du, dp/dv (:108-131), max(|dpdu|,|dpdv|) (:168) and the
// normalised fp32-cross-product geometric normal that intersect() derives from Embree's Ng
// (src/intersection.cpp:41). `degenerate_uv` marks |det| <= 1e-8f, where dpdu/dpdv hold coordinate_system(gn).
struct DevTriShade {
    double n[3][3];      // vertex normals (has_normals)
    double uv[3][2];     // vertex uvs (has_uvs), else the reference's default (0,0),(1,0),(1,1)
    double dpdu[3], dpdv[3];
    double gn[3];        // normalize(fp64(e1 x e2 in fp32))
    double inv_uv_size;
    int32_t shape_id, prim_id, material_id, light_id; // light_id < 0: not an emitter
    int32_t has_normals, has_uvs;
    // flat_frame: the three vertex normals are identical (or absent), so the shading frame does not depend on the hit
    // point beyond rounding: n[0] = tangent, n[1] = bitangent, n[2] = shading normal, evaluated at the barycentre.
    int32_t flat_frame, pad;
};

struct DevSphere {
    double center[3];
    double radius;
    int32_t shape_id, material_id, light_id, pad;
};

// One emitter of Integrator::Path (DiffuseAreaLight, src/lights/diffuse_area_light.inl). Mesh emitters carry their
// triangle sampling table (init_sampling_dist, src/shapes/triangle_mesh.inl:60-75) in DevSceneView::light_tri_*.
struct DevLight {
    double intensity[3];
    double area;             // surface_area of the shape
    int32_t is_sphere;       // 0: triangle mesh, 1: sphere (index into DevSceneView::spheres)
    int32_t sphere_index;
    int32_t tri_first;       // first triangle of this emitter in light_tri_pos / light_tri_nrm
    int32_t tri_count;
    int32_t cdf_first;       // first entry of this emitter's cdf (tri_count + 1 entries) in light_tri_cdf
    int32_t has_normals;
};

struct DevImage {        // mip chain of one Mipmap1/Mipmap3 (src/mipmap.h:27-48), texels fp64 in one pool
    int32_t channels, num_levels;
    int32_t width[8], height[8];
    int64_t offset[8];   // in doubles, into the texel pool
};

struct DevCamera {
    double sample_to_cam[16];
    double cam_to_world[16];
    double org[3];       // xform_point(cam_to_world, 0) — constant per render (src/camera.cpp:42)
    int32_t width, height;
    int32_t filter_type;
    int32_t pow2_film;   // width and height are powers of two: x / width is an exact scaling (camera fast path)
    double filter_param;
    double inv_width, inv_height;   // 1 / width, 1 / height (exact when pow2_film)
};

// Everything a kernel needs, passed by value as one kernel argument (pointers are device pointers).
struct DevSceneView {
    DevCamera cam;
    const DevBvhNode *nodes;
    const DevBvh4Node *nodes4;              // wide form of the same tree (LDS-resident scenes)
    const DevBvh8Node *nodes8;              // 8-wide quantised form (scenes walked from HBM)
    const DevBvh4QNode *nodes4q;            // nodes4 with quantised boxes (scenes walked from HBM, GDPT_HBM_Q4 builds)
    const DevPrim *prims;
    const DevTriShade *tris;
    const DevSphere *spheres;
    const struct GdptMaterial *materials;   // same layout as the C ABI struct
    const double *light_intensity;          // 3 per area light
    const DevImage *images;
    const double *texels;
    // Integrator::Path: emitter selection table (src/scene.cpp:44-53; cdf has num_lights + 1 entries, the last one keeps
    // the total like make_table_dist_1d leaves it) and the emitters' own tables / fp64 triangles
    const DevLight *lights;
    const double *light_pmf, *light_cdf;
    const double *light_tri_cdf;
    const double *light_tri_pos;            // 9 doubles per emitter triangle: v0, v1, v2
    const double *light_tri_nrm;            // 9 doubles per emitter triangle: n0, n1, n2 (zeros when the mesh has none)
    // environment map (Envmap, src/lights/envmap.inl): lat-long image + its TableDist2D (src/table_dist.cpp:40-150)
    int32_t has_envmap, env_light_id, env_image_id, env_pad;
    int32_t env_w, env_h;
    double env_scale;
    double env_to_world[16], env_to_local[16];
    const double *env_cdf_rows, *env_pdf_rows, *env_cdf_marginals, *env_pdf_marginals;
    int32_t num_nodes, num_nodes4, num_nodes8, num_prims, num_tris, num_spheres;
    int32_t num_materials, num_lights, num_images;
    int32_t max_depth, rr_depth;
    int32_t all_textures_constant;          // no image / checkerboard texture anywhere: uv and footprints are unobservable
    double isect_eps;                       // get_intersection_epsilon, src/scene.h:100-102
};
